"""End-to-end pin of the oracle against REAL third-party arithmetic.

Run in the build container (needs scipy; writes tests/golden/e2e_lapack.npz and prints the report that DESIGN.md section 2
quotes; about ten minutes on eight cores):

    python tests/golden/make_e2e_goldens.py

For every sample of a case the local frame is computed the way the reference computes it, with the third-party pieces the
oracle had to *interpret* replaced by real library code:

  * FLANN radius search (hand_search.cpp:85): candidates from a kd-tree with a padded radius, then FLANN's own test in
    float32, d2 = ((dx*dx)+dy*dy)+dz*dz kept iff d2 < (float)(r*r), ascending (d2, index)  [variants: `<=`, ties by
    descending index -- the sensitivity runs]
  * M, N exactly as quadric.cpp:24-141 (sequential sums in neighbour order, one numpy cumsum per entry)
  * LAPACK dggev through scipy.linalg.lapack.dggev -- the routine quadric.cpp:353,359 calls -- then quadric.cpp:149-153:
    eigen_values = alphar / beta, argmin over THE FIRST NINE, that column, entries 3..5 halved
  * normals as quadric.cpp:238-247; sum n n^T (quadric.cpp:266) and the (n_i . n_j)^6 column sums (quadric.cpp:283) as plain
    sequential sums, std::pow(v, 6) through libm (numpy.power(v, 6.0))
  * Eigen::EigenSolver (general, non-symmetric solver; quadric.cpp:268) -> numpy.linalg.eig (LAPACK dgeev, likewise a
    general Hessenberg-QR solver)
  * the rest of quadric.cpp:278-304 literally.

Those frames are then handed to the oracle's hand search (orc_hands_from_frames: rotating_hand.cpp / finger_hand.cpp /
antipodal.cpp are plain loops over doubles whose order the source fixes) and to its HOG + SVM, and the result is stored.

Round 3 adds
  * the AXIS-ALIGNED scenes (`C2u`: SURVEY section 8d to the letter; `boxu`: an ideal table + box), where most Taubin
    neighbourhoods are exactly planar and the pencil of quadric.cpp:143-153 is singular.  A sample is called DEGENERATE
    (an EXACT FIT) when dggev itself reports an eigenvalue at the noise level (|lambda| < 1e-9; fits with a residual have
    lambda >= 1e-7 on these clouds) among the first nine: a quadric then passes through every neighbour exactly -- one
    lattice plane (three such eigenvalues: the minimiser is not even unique), or two (a table and a box face) -- and
    what the reference goes on to compute is decided by rounding noise: the eigenvector inside the null space, the
    in-plane direction of the curvature axis (sum n n^T has a double zero eigenvalue), the argmax among hundreds of
    exactly tied columns of (n_i . n_j)^6, the normals of the points where the fitted quadric's gradient vanishes.  For
    those samples the script also runs the LAPACK path a second time with one unit in the last place added to or
    subtracted from every entry of M: the reference against ITSELF is the yardstick the oracle's distance from the
    reference is to be read against;
  * the ANTIPODAL pass (hand_search.cpp:17-26: findQuadrics over ALL points with r = 0.01, then antipodal.cpp:12-86):
    LAPACK frames for every cloud point inside the 8 cm ball of a tested sample -> cloud_normals_ -> the oracle's hand
    search with those normals;
  * all 2000 samples of C2, all 500 of C1, 256 of C4.

Nothing here is imported by the product; tests/test_e2e_lapack.py compares the oracle (CPU) and the HIP path (GPU) with
the stored lists under the tolerance this script measured.
"""
from __future__ import annotations

import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from scipy.linalg import lapack  # noqa: E402
from scipy.spatial import cKDTree  # noqa: E402

from agile_grasp_amd import synthetic  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

# (scene, samples, production-mode normals, antipodal pass)
CASES = [
    ("tiny", 64, False, False), ("small", 200, False, False), ("C1", 500, False, False), ("C2", 2000, False, False),
    ("C4", 256, False, False),
    # the reference's production mode (HandSearch hard-wires uses_determinstic_normal_estimation_ = false): 50 draws of
    # rand() % n per neighbourhood of more than 50 points, one stream in sample order, glibc's default seed 1
    ("small", 200, True, False), ("C2", 128, True, False),
    # axis-aligned scenes: exactly planar neighbourhoods, singular pencils
    ("C2u", 2000, False, False), ("boxu", 400, False, False),
    # calculates_antipodal = true (the training path and the antipodal labels of north_star)
    ("small", 200, False, True), ("smallu", 200, False, True), ("C2", 2000, False, True),
]
R_TAUBIN, R_HANDS, R_NORMALS = 0.03, 0.08, 0.01
ZERO_EV = 1e-9  # |eigenvalue| below this (the regular ones are >= 1e-7 on these clouds) counts as "at the noise level"

# compact per-sample record of the fixture (the full 200-byte frame is only needed inside this script)
FRAME_GOLD = np.dtype([("normal", "<f8", 3), ("axis", "<f8", 3), ("params", "<f8", 10), ("n_nb", "<i4"), ("max_index", "<i4"),
                       ("majority_cam", "i1"), ("degenerate", "i1")])


def key_of(name, rand50, antipodal):
    return name + ("_rand50" if rand50 else "") + ("_antipodal" if antipodal else "")


def seqsum(a: np.ndarray) -> np.ndarray:
    """Strictly sequential (left-to-right) sum down axis 0 -- numpy's add.reduce is pairwise, its cumsum is not."""
    if a.shape[0] == 0:
        return np.zeros(a.shape[1:], a.dtype)
    return np.cumsum(a, axis=0)[-1]


_G = {}  # per-process scene: xyz, cam, cam_origins, tree


def set_scene(sc):
    _G["xyz"], _G["cam"], _G["co"] = sc.xyz, sc.cam, sc.cam_origins
    _G["tree"] = cKDTree(sc.xyz.astype(np.float64))


def radius_search(q_index: int, r: float, inclusive: bool = False, ties_descending: bool = False):
    """FLANN L2_Simple<float>: float32 arithmetic in FLANN's accumulation order (exact: the kd-tree only pre-selects)."""
    xyz = _G["xyz"]
    q = xyz[q_index]
    cand = np.asarray(_G["tree"].query_ball_point(q.astype(np.float64), r * 1.001 + 1e-6), dtype=np.int64)
    d = q[None, :].astype(np.float32) - xyz[cand]
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    r2 = np.float32(r * r)
    m = d2 <= r2 if inclusive else d2 < r2
    keep, d2k = cand[m], d2[m]
    order = np.lexsort((-keep if ties_descending else keep, d2k))
    return keep[order], int((d2 == r2).sum())


def build_MN(p: np.ndarray):
    """quadric.cpp:24-141 for the neighbours p (n x 3 float32, search order)."""
    n = p.shape[0]
    x, y, z = (p[:, k].astype(np.float64) for k in range(3))
    x2, y2, z2, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    M = np.zeros((10, 10))
    N = np.zeros((10, 10))
    s = seqsum
    M[0, 0], M[0, 1], M[0, 2], M[0, 3], M[0, 4] = s(x2 * x2), s(x2 * y2), s(x2 * z2), s(x2 * xy), s(x2 * yz)
    M[0, 5], M[0, 6], M[0, 7], M[0, 8], M[0, 9] = s(x2 * xz), s(x2 * x), s(x2 * y), s(x2 * z), s(x2)
    M[1, 1], M[1, 2], M[1, 3], M[1, 4], M[1, 5] = s(y2 * y2), s(y2 * z2), s(y2 * xy), s(y2 * yz), s(y2 * xz)
    M[1, 6], M[1, 7], M[1, 8], M[1, 9] = s(y2 * x), s(y2 * y), s(y2 * z), s(y2)
    M[2, 2], M[2, 3], M[2, 4], M[2, 5] = s(z2 * z2), s(z2 * xy), s(z2 * yz), s(z2 * xz)
    M[2, 6], M[2, 7], M[2, 8], M[2, 9] = s(z2 * x), s(z2 * y), s(z2 * z), s(z2)
    M[3, 8], M[3, 9], M[4, 9], M[5, 9], M[6, 9], M[7, 9], M[8, 9] = s(x * yz), s(xy), s(yz), s(xz), s(x), s(y), s(z)
    if n > 0:  # the "repeating elements" are assigned inside the loop: with no neighbour they stay zero
        M[3, 3], M[5, 5], M[3, 5], M[3, 6], M[5, 6], M[6, 6] = M[0, 1], M[0, 2], M[0, 4], M[0, 7], M[0, 8], M[0, 9]
        M[4, 4], M[3, 4], M[3, 7], M[4, 7], M[7, 7] = M[1, 2], M[1, 5], M[1, 6], M[1, 8], M[1, 9]
        M[4, 5], M[5, 8], M[4, 8], M[8, 8] = M[2, 3], M[2, 6], M[2, 7], M[2, 9]
        M[4, 6], M[5, 7], M[6, 7], M[7, 8], M[6, 8] = M[3, 8], M[3, 8], M[3, 9], M[4, 9], M[5, 9]
    N[0, 0], N[0, 3], N[0, 5], N[0, 6] = s(4.0 * x2), s(2.0 * xy), s(2.0 * xz), s(2.0 * x)
    N[1, 1], N[1, 3], N[1, 4], N[1, 7] = s(4.0 * y2), s(2.0 * xy), s(2.0 * yz), s(2.0 * y)
    N[2, 2], N[2, 4], N[2, 5], N[2, 8] = s(4.0 * z2), s(2.0 * yz), s(2.0 * xz), s(2.0 * z)
    N[3, 3], N[3, 4], N[3, 5], N[3, 6], N[3, 7] = s(x2 + y2), s(xz), s(yz), s(y), s(x)
    N[4, 4], N[4, 5], N[4, 7], N[4, 8] = s(y2 + z2), s(xy), s(z), s(y)
    N[5, 5], N[5, 6], N[5, 8] = s(x2 + z2), s(z), s(x)
    M[9, 9] = n
    N[6, 6] = N[7, 7] = N[8, 8] = n
    iu = np.triu_indices(10, 1)
    M[(iu[1], iu[0])] = M[iu]
    N[(iu[1], iu[0])] = N[iu]
    return M, N


_LIBC = None


def libc_rand():
    """The host's real rand() (quadric.cpp:184)."""
    global _LIBC
    if _LIBC is None:
        import ctypes

        _LIBC = ctypes.CDLL("libc.so.6")
        _LIBC.rand.restype = ctypes.c_int
    return _LIBC.rand()


def dggev_min(M, N):
    """solveGeneralizedEigenProblem (quadric.cpp:330-363) + quadric.cpp:149-151: dggev("N", "V"), alphar / beta, the first
    minimum over the first nine (Eigen's minCoeff visitor: first element, then strict `<`)."""
    alphar, alphai, beta, _vl, vr, _work, info = lapack.dggev(np.asfortranarray(M), np.asfortranarray(N), compute_vl=0, compute_vr=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        ev = alphar / beta
    mi = 0
    for k in range(1, 9):
        if ev[k] < ev[mi]:
            mi = k
    return ev, vr, mi, alphai, beta, info


def frame_lapack(sample_index, radius=R_TAUBIN, rand50=False, perturb=False, light=False, **search_kw):
    """One Quadric (quadric.cpp:14-305) with real LAPACK / a general eigen-solver.  `perturb`: every entry of M moved by
    one unit in the last place (random sign, symmetric) before dggev sees it.  `light`: skip the diagnostics."""
    xyz, cam, cam_origins = _G["xyz"], _G["cam"], _G["co"]
    q = xyz[sample_index]
    idx, n_on_boundary = radius_search(sample_index, radius, **search_kw)
    pts = xyz[idx]
    n = len(idx)
    M, N = build_MN(pts)
    if perturb:
        rng = np.random.default_rng(sample_index + 77)
        M = M * (1.0 + (rng.integers(0, 2, M.shape) * 2 - 1) * 2.0 ** -52)
        M = np.triu(M) + np.triu(M, 1).T
    ev, vr, mi, alphai, beta, info = dggev_min(M, N)
    params = vr[:, mi].copy()
    params[3:6] *= 0.5
    a, b, c = params[0], params[1], params[2]
    d, e, f = 2.0 * params[3], 2.0 * params[4], 2.0 * params[5]
    g, h, i = params[6], params[7], params[8]
    sub_cams = cam[idx]
    if rand50 and n > 50:  # quadric.cpp:177-193 (the clouds hold no NaN, so no re-draw happens)
        draws = np.array([libc_rand() % n for _ in range(50)])
        pts_n, sub_cams = pts[draws], cam[idx][draws]
    else:
        pts_n = pts
    X, Y, Z = (pts_n[:, k].astype(np.float64) for k in range(3))
    fx = (((2.0 * a) * X + d * Y) + f * Z) + g  # quadric.cpp:238
    fy = (((2.0 * b) * Y + d * X) + e * Z) + h
    fz = (((2.0 * c) * Z + e * Y) + f * X) + i
    with np.errstate(divide="ignore", invalid="ignore"):
        mag = np.sqrt((fx * fx + fy * fy) + fz * fz)
        nrm = np.stack([fx / mag, fy / mag, fz / mag])  # 3 x n
    cams = sub_cams
    counts = [int((cams == 0).sum()), int((cams == 1).sum())]
    majority = 0 if counts[0] >= counts[1] else 1  # maxCoeff: first maximum
    fr = np.zeros(1, O.FRAME_DTYPE)[0]
    sample = q.astype(np.float64)
    fr["sample"], fr["params"], fr["eigenvalue"], fr["n_nb"], fr["majority_cam"] = sample, params, ev[mi], n, majority
    evs = ev[:9][np.isfinite(ev[:9])]
    n_zero = int((np.abs(evs) < ZERO_EV).sum())
    extra = {"info": int(info), "n_zero_ev": n_zero, "degenerate": n_zero >= 1, "nan_frame": False}
    if not np.isfinite(nrm).all():  # the reference would carry NaN through the frame: no points pass its crop, no hands
        fr["valid"] = 1
        fr["normal"] = fr["axis"] = fr["binormal"] = np.nan
        extra["nan_frame"] = True
        return fr, extra
    M3 = np.zeros((3, 3))
    for r in range(3):
        for cc in range(3):
            M3[r, cc] = seqsum(nrm[r] * nrm[cc])  # quadric.cpp:266
    w, V = np.linalg.eig(M3)  # general solver, like Eigen::EigenSolver (quadric.cpp:268-270)
    w, V = np.real(w), np.real(V)
    ai = 0
    for k in range(1, 3):
        if w[k] < w[ai]:
            ai = k
    axis = V[:, ai].copy()
    G = (nrm[0][:, None] * nrm[0][None, :] + nrm[1][:, None] * nrm[1][None, :]) + nrm[2][:, None] * nrm[2][None, :]
    col = seqsum(np.power(G, 6.0))  # quadric.cpp:283: .array().pow(6).colwise().sum()
    mx = 0
    for k in range(1, nrm.shape[1]):
        if col[k] > col[mx]:
            mx = k
    P = np.eye(3) - np.outer(axis, axis)
    nm = nrm[:, mx]
    normpartial = np.array([(P[r, 0] * nm[0] + P[r, 1] * nm[1]) + P[r, 2] * nm[2] for r in range(3)])
    normal = normpartial / np.sqrt((normpartial[0] ** 2 + normpartial[1] ** 2) + normpartial[2] ** 2)
    cross = lambda u, v: np.array([u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]])
    dot = lambda u, v: (u[0] * v[0] + u[1] * v[1]) + u[2] * v[2]
    binormal = cross(axis, normal)
    s2s = sample - cam_origins[majority]
    if dot(normal, s2s) > 0:
        normal = normal * -1.0
    if dot(binormal, s2s) > 0:
        binormal = binormal * -1.0
    axis = cross(normal, binormal)
    fr["normal"], fr["axis"], fr["binormal"], fr["max_index"], fr["valid"] = normal, axis, binormal, mx, 1
    if light:
        return fr, extra
    # how far dggev's own answer moves when its input moves by one unit in the last place: the conditioning of the
    # reference's un-centred formulation, against which the oracle's distance from LAPACK is to be read
    rng = np.random.default_rng(sample_index)
    Mp = M * (1.0 + (rng.integers(0, 2, M.shape) * 2 - 1) * 2.0 ** -52)
    Mp = np.triu(Mp) + np.triu(Mp, 1).T
    _ev2, vr2, m2, _a2, _b2, _i2 = dggev_min(Mp, N)
    extra.update({"self_angle": float(angle(vr[:, mi], vr2[:, m2])), "beta_zero_at": [int(k) for k in np.nonzero(beta == 0)[0]],
                  "complex": bool(np.any(alphai != 0)), "min_index": mi, "on_boundary": n_on_boundary,
                  "cond_N9": float(np.linalg.cond(N[:9, :9]))})
    return fr, extra


def _frame_job(args):
    idx, kw = args
    return frame_lapack(idx, **kw)


def angle(u, v):
    c = np.abs((u * v).sum(-1)) / (np.linalg.norm(u, axis=-1) * np.linalg.norm(v, axis=-1))
    return np.arccos(np.clip(c, -1, 1))


def compare_lists(a, b, only_samples=None):
    """Flip counts and max |delta| per field between two hypothesis lists, aligned by (sample, orientation); with
    `only_samples` (a boolean mask over the sample positions) restricted to the hypotheses of those samples."""
    if only_samples is not None:
        a = a[only_samples[a["sample"]]]
        b = b[only_samples[b["sample"]]]
    ka = {(int(h["sample"]), int(h["orientation"])): h for h in a}
    kb = {(int(h["sample"]), int(h["orientation"])): h for h in b}
    common = sorted(set(ka) & set(kb))
    rep = {"n_a": len(a), "n_b": len(b), "only_a": len(set(ka) - set(kb)), "only_b": len(set(kb) - set(ka))}
    for f in ("finger_index", "depth_index", "n_in_box", "cam_source", "half_antipodal", "full_antipodal"):
        rep["flips_" + f] = int(sum(int(ka[k][f]) != int(kb[k][f]) for k in common))
    rep["max_abs_n_in_box"] = int(max([abs(int(ka[k]["n_in_box"]) - int(kb[k]["n_in_box"])) for k in common] or [0]))
    rep["median_abs_axis"] = float(np.median([np.abs(np.asarray(ka[k]["axis"]) - np.asarray(kb[k]["axis"])).max()
                                              for k in common] or [0.0]))
    for f in ("axis", "approach", "binormal", "bottom", "surface", "width"):
        rep["max_abs_" + f] = float(max([np.abs(np.asarray(ka[k][f]) - np.asarray(kb[k][f])).max() for k in common] or [0.0]))
    return rep, common, ka, kb


def lapack_frames(pool, indices, **kw):
    """LAPACK frames of the given cloud points; sequential (one rand() stream) in the production mode."""
    jobs = [(int(s), kw) for s in indices]
    res = [_frame_job(j) for j in jobs] if kw.get("rand50") else pool.map(_frame_job, jobs, chunksize=64)
    frames = np.zeros(len(jobs), O.FRAME_DTYPE)
    for k, (fr, _ex) in enumerate(res):
        frames[k] = fr
    return frames, [ex for _fr, ex in res]


def lapack_normals(pool, sc, samples, frames):
    """cloud_normals_ the way hand_search.cpp:13-26,102 leaves it, for every point a tested sample's hand ball can read:
    zero, then the r = 0.01 frame normal of every point, then the r = 0.03 frame normal of every sample."""
    tree = _G["tree"]
    need = np.zeros(sc.n, bool)
    for s in samples:
        need[tree.query_ball_point(sc.xyz[s].astype(np.float64), R_HANDS * 1.01)] = True
    pts = np.nonzero(need)[0]
    fr, ex = lapack_frames(pool, pts, radius=R_NORMALS, light=True)
    normals = np.zeros((sc.n, 3))
    normals[pts] = fr["normal"]
    normals[samples] = frames["normal"]
    return normals, pts, fr, ex


def run_case(pool, name, n_samples, w, rho, rand50=False, antipodal=False, self_all=False, **search_kw):
    sc = synthetic.config(name)
    set_scene(sc)
    samples = sc.samples[:n_samples]
    p = O.default_params(sc.cam_origins, normals_mode=O.NORMALS_RAND50 if rand50 else O.NORMALS_DETERMINISTIC, rand_seed=1)
    if rand50:
        import ctypes

        libc_rand()
        _LIBC.srand(ctypes.c_uint(1))
    with mp.get_context("fork").Pool(8) as pool:  # (forked after set_scene: the workers see this case's cloud)
        frames, extras = lapack_frames(pool, samples, rand50=rand50, **search_kw)
        degenerate = np.array([e["degenerate"] for e in extras])
        normals = None
        rep_n = {}
        if antipodal:
            normals, pts, nfr, nex = lapack_normals(pool, sc, samples, frames)
        frames_p = None
        if (degenerate.any() or self_all) and not search_kw:  # the reference against itself, one input bit flipped
            frames_p, _ = lapack_frames(pool, samples, perturb=True, light=True)
    res = O.hands_from_frames(p, sc.xyz, sc.cam, samples, frames, normals=normals, want_images=True)
    keep, sums = O.classify(res["images"], w, rho)
    own = O.find_hands(p, sc.xyz, sc.cam, samples, calculates_antipodal=antipodal, want_images=True)
    okeep, osums = O.classify(own["images"], w, rho)
    of = own["frames"]
    regular = ~degenerate
    if antipodal:  # the oracle's own r = 0.01 normals against LAPACK's, on the points the hand search reads
        onf = O.fit_frames(p, sc.xyz, sc.cam, pts.astype(np.int32), R_NORMALS)
        ndeg = np.array([e["degenerate"] for e in nex])
        okn = np.isfinite(nfr["normal"]).all(1) & np.isfinite(onf["normal"]).all(1)
        dn = np.abs(nfr["normal"] - onf["normal"]).max(1)
        rep_n = {"normals_points": int(len(pts)), "normals_degenerate": int(ndeg.sum()),
                 "normals_nan_lapack": int((~np.isfinite(nfr["normal"]).all(1)).sum()),
                 "normals_nan_oracle": int((~np.isfinite(onf["normal"]).all(1)).sum()),
                 "normals_max_abs_regular": float(dn[okn & ~ndeg].max()) if (okn & ~ndeg).any() else 0.0,
                 "normals_median_abs_regular": float(np.median(dn[okn & ~ndeg])) if (okn & ~ndeg).any() else 0.0,
                 "normals_max_abs_degenerate": float(dn[okn & ndeg].max()) if (okn & ndeg).any() else 0.0,
                 "normals_above_1e-3_degenerate": int((dn[okn & ndeg] > 1e-3).sum()),
                 "normals_above_1e-3_regular": int((dn[okn & ~ndeg] > 1e-3).sum())}
    rep, common, ka, kb = compare_lists(res["hyps"], own["hyps"], regular)
    rep_all, _c, _ka, _kb = compare_lists(res["hyps"], own["hyps"])
    pos_a = {(int(h["sample"]), int(h["orientation"])): i for i, h in enumerate(res["hyps"])}
    pos_b = {(int(h["sample"]), int(h["orientation"])): i for i, h in enumerate(own["hyps"])}
    rg = lambda key, default=0.0: [e[key] for e, r in zip(extras, regular) if r and key in e] or [default]
    rep.update({
        "case": key_of(name, rand50, antipodal), "samples": int(len(samples)), "degenerate_samples": int(degenerate.sum()),
        "hyps_lapack_all": int(len(res["hyps"])), "hyps_oracle_all": int(len(own["hyps"])),
        "only_lapack_all": rep_all["only_a"], "only_oracle_all": rep_all["only_b"],
        "nan_frames_lapack": int(sum(e["nan_frame"] for e in extras)),
        "oracle_invalid_frames": int((of["valid"] == 0).sum()),
        "dggev_info_nonzero": int(sum(e["info"] != 0 for e in extras)),
        "dggev_one_ulp_input_max_angle_rad": float(max(rg("self_angle"))),
        "dggev_one_ulp_input_median_angle_rad": float(np.median(rg("self_angle"))),
        "median_angle_params_rad": float(np.median(angle(frames["params"][regular], of["params"][regular]))) if regular.any() else 0.0,
        "max_cond_N9": float(max(rg("cond_N9"))),
        "dggev_complex_spectrum": int(sum(e.get("complex", False) for e in extras)),
        "beta_zero_not_last": int(sum(e.get("beta_zero_at", [9]) != [9] for e in extras)),
        "points_exactly_on_taubin_radius": int(sum(e.get("on_boundary", 0) for e in extras)),
        "n_nb_mismatch": int((frames["n_nb"] != of["n_nb"]).sum()),
        "max_index_mismatch": int((frames["max_index"] != of["max_index"])[regular].sum()),
        "majority_cam_mismatch": int((frames["majority_cam"] != of["majority_cam"]).sum()),
        "max_angle_params_rad": float(angle(frames["params"][regular], of["params"][regular]).max()) if regular.any() else 0.0,
        "max_abs_normal": float(np.abs(frames["normal"] - of["normal"])[regular].max()) if regular.any() else 0.0,
        "max_abs_axis_frame": float(np.abs(frames["axis"] - of["axis"])[regular].max()) if regular.any() else 0.0,
        "svm_label_flips": int(sum(int(keep[pos_a[k]]) != int(okeep[pos_b[k]]) for k in common)),
        "max_abs_svm_sum": float(max([abs(sums[pos_a[k]] - osums[pos_b[k]]) for k in common] or [0.0])),
        "image_pixels_differing": int(sum(int((res["images"][pos_a[k]] != own["images"][pos_b[k]]).sum()) for k in common)),
        "svm_kept": int(keep.sum()), "min_abs_svm_sum": float(np.abs(sums).min()) if len(sums) else None,
    })
    rep.update(rep_n)
    if self_all and frames_p is not None:  # LAPACK against LAPACK with one ulp on every entry of M, regular samples: the yardstick
        resp = O.hands_from_frames(p, sc.xyz, sc.cam, samples, frames_p, normals=normals, want_images=True)
        keep_p, sums_p = O.classify(resp["images"], w, rho)
        rs, common_s, ka_s, kb_s = compare_lists(res["hyps"], resp["hyps"], regular)
        pa = {(int(h["sample"]), int(h["orientation"])): i for i, h in enumerate(res["hyps"])}
        pb = {(int(h["sample"]), int(h["orientation"])): i for i, h in enumerate(resp["hyps"])}
        rep["self"] = {k: rs[k] for k in ("n_a", "only_a", "only_b", "flips_finger_index", "flips_depth_index", "flips_cam_source",
                                            "flips_n_in_box", "max_abs_n_in_box", "max_abs_axis", "max_abs_bottom", "max_abs_surface",
                                            "max_abs_width")}
        rep["self"]["svm_label_flips"] = int(sum(int(keep[pa[k]]) != int(keep_p[pb[k]]) for k in common_s))
        rep["self"]["max_abs_svm_sum"] = float(max([abs(sums[pa[k]] - sums_p[pb[k]]) for k in common_s] or [0.0]))
        rep["self"]["max_index_mismatch"] = int((frames["max_index"] != frames_p["max_index"])[regular].sum())
    if degenerate.any():
        # degenerate samples: the surface normal is what every minimiser agrees on; the in-plane axis is not
        fin = np.isfinite(frames["normal"]).all(1) & np.isfinite(of["normal"]).all(1) & degenerate
        rep["degenerate_normal_max_abs"] = float(np.abs(frames["normal"] - of["normal"])[fin].max()) if fin.any() else 0.0
        rep["degenerate_normal_above_1e-3"] = int((np.abs(frames["normal"] - of["normal"])[fin].max(1) > 1e-3).sum())
        rep["degenerate_axis_median_abs"] = float(np.median(np.abs(frames["axis"] - of["axis"])[fin].max(1))) if fin.any() else 0.0
        rd, _c2, _a2, _b2 = compare_lists(res["hyps"], own["hyps"], degenerate)
        rep.update({"degenerate_hyps_lapack": rd["n_a"], "degenerate_hyps_oracle": rd["n_b"],
                    "degenerate_only_lapack": rd["only_a"], "degenerate_only_oracle": rd["only_b"]})
        if frames_p is not None:
            resp = O.hands_from_frames(p, sc.xyz, sc.cam, samples, frames_p, normals=normals, want_images=False)
            rs, _c3, _a3, _b3 = compare_lists(res["hyps"], resp["hyps"], degenerate)
            rr, _c4, _a4, _b4 = compare_lists(res["hyps"], resp["hyps"], regular)
            finp = np.isfinite(frames["normal"]).all(1) & np.isfinite(frames_p["normal"]).all(1) & degenerate
            rep.update({"self_degenerate_hyps": rs["n_a"], "self_degenerate_hyps_perturbed": rs["n_b"],
                        "self_degenerate_only_plain": rs["only_a"], "self_degenerate_only_perturbed": rs["only_b"],
                        "self_regular_only_plain": rr["only_a"], "self_regular_only_perturbed": rr["only_b"],
                        "self_degenerate_axis_median_abs": float(np.median(np.abs(frames["axis"] - frames_p["axis"])[finp].max(1))) if finp.any() else 0.0,
                        "self_degenerate_normal_max_abs": float(np.abs(frames["normal"] - frames_p["normal"])[finp].max()) if finp.any() else 0.0})
    gold = np.zeros(len(samples), FRAME_GOLD)
    for f in ("normal", "axis", "params", "n_nb", "max_index", "majority_cam"):
        gold[f] = frames[f]
    gold["degenerate"] = degenerate
    return sc, samples, gold, res["hyps"], keep, sums, rep


def main():
    w, rho = O.load_svm(os.path.join(ROOT, "tests", "golden", "svm_032015_linear_20_20_same"))
    store, report = {}, []
    only = set(sys.argv[1:])
    for name, ns, r50, anti in CASES:
        key = key_of(name, r50, anti)
        if only and key not in only:
            continue
        sc, samples, gold, hyps, keep, sums, rep = run_case(None, name, ns, w, rho, rand50=r50, antipodal=anti)
        report.append(rep)
        print(json.dumps(rep), flush=True)
        store[f"{key}_samples"] = samples.astype(np.int32)
        store[f"{key}_frames"] = gold
        store[f"{key}_hyps"] = hyps
        store[f"{key}_keep"] = keep
        store[f"{key}_sums"] = sums.astype(np.float64)
    # FLANN sensitivity (the radius criterion and the order of equal distances are the builder's reading of FLANN)
    sens = []
    for variant, kw in (("inclusive_radius", dict(inclusive=True)), ("ties_descending", dict(ties_descending=True))):
        for name, ns in (("tiny", 64), ("small", 200), ("C2", 256)):
            if only:
                continue
            _, _, gold_v, hyps_v, keep_v, _, _ = run_case(None, name, ns, w, rho, **kw)
            base_h, base_k = store[f"{name}_hyps"], store[f"{name}_keep"]
            sel = base_h["sample"] < ns
            base_h, base_k = base_h[sel], base_k[sel]
            rep, common, ka, kb = compare_lists(hyps_v, base_h)
            pa = {(int(h["sample"]), int(h["orientation"])): i for i, h in enumerate(hyps_v)}
            pb = {(int(h["sample"]), int(h["orientation"])): i for i, h in enumerate(base_h)}
            rep.update({"variant": variant, "case": name,
                        "n_nb_changed": int((gold_v["n_nb"] != store[f"{name}_frames"]["n_nb"][:ns]).sum()),
                        "max_index_changed": int((gold_v["max_index"] != store[f"{name}_frames"]["max_index"][:ns]).sum()),
                        "svm_label_flips": int(sum(int(keep_v[pa[k]]) != int(base_k[pb[k]]) for k in common))})
            sens.append(rep)
            print(json.dumps(rep), flush=True)
    if only:
        return
    store["report_json"] = np.frombuffer(json.dumps({"cases": report, "flann_sensitivity": sens}).encode(), np.uint8)
    out = os.path.join(ROOT, "tests", "golden", "e2e_lapack.npz")
    np.savez_compressed(out, **store)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
