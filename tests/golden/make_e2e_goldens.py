"""End-to-end pin of the oracle against REAL third-party arithmetic.

Run in the build container (needs scipy; writes tests/golden/e2e_lapack.npz and prints the report that DESIGN.md section 2
quotes):

    python tests/golden/make_e2e_goldens.py

For every sample of `tiny` (64), `small` (200) and the first 256 samples of C2, the local frame is computed the way the
reference computes it, with the third-party pieces the oracle had to *interpret* replaced by real library code:

  * FLANN radius search (hand_search.cpp:85): brute force over the whole cloud in float32, d2 = ((dx*dx)+dy*dy)+dz*dz,
    kept iff d2 < (float)(r*r), ascending (d2, index)  [variants: `<=`, ties by descending index -- the sensitivity runs]
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
Nothing here is imported by the product; tests/test_e2e_lapack.py compares the oracle (CPU) and the HIP path (GPU) with
the stored lists under the tolerance this script measured.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from scipy.linalg import lapack  # noqa: E402

from agile_grasp_amd import synthetic  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

CASES = [("tiny", 64, False), ("small", 200, False), ("C2", 256, False),
         # the reference's production mode (HandSearch hard-wires uses_determinstic_normal_estimation_ = false): 50 draws of
         # rand() % n per neighbourhood of more than 50 points, one stream in sample order, glibc's default seed 1
         ("small", 200, True), ("C2", 128, True)]
R_TAUBIN = 0.03


def seqsum(a: np.ndarray) -> np.ndarray:
    """Strictly sequential (left-to-right) sum down axis 0 -- numpy's add.reduce is pairwise, its cumsum is not."""
    if a.shape[0] == 0:
        return np.zeros(a.shape[1:], a.dtype)
    return np.cumsum(a, axis=0)[-1]


def radius_search(xyz: np.ndarray, q: np.ndarray, r: float, inclusive: bool = False, ties_descending: bool = False):
    """FLANN L2_Simple<float> brute force: float32 arithmetic in FLANN's accumulation order."""
    d = q[None, :].astype(np.float32) - xyz
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    r2 = np.float32(r * r)
    keep = np.nonzero(d2 <= r2 if inclusive else d2 < r2)[0]
    order = np.lexsort((-keep if ties_descending else keep, d2[keep]))
    idx = keep[order]
    return idx, int((d2 == r2).sum())


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


def frame_lapack(xyz, cam, cam_origins, sample_index, rand50=False, **search_kw):
    """One Quadric (quadric.cpp:14-305, deterministic normals) with real LAPACK / a general eigen-solver."""
    q = xyz[sample_index]
    idx, n_on_boundary = radius_search(xyz, q, R_TAUBIN, **search_kw)
    pts = xyz[idx]
    n = len(idx)
    M, N = build_MN(pts)
    # solveGeneralizedEigenProblem (quadric.cpp:330-363): dggev("N", "V"); Eigen matrices are column-major, M and N symmetric
    alphar, alphai, beta, _vl, vr, _work, info = lapack.dggev(np.asfortranarray(M), np.asfortranarray(N), compute_vl=0,
                                                               compute_vr=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        ev = alphar / beta  # quadric.cpp:149
    mi = 0  # Eigen's minCoeff visitor over segment(0, 9): first element, then strict `<`
    for k in range(1, 9):
        if ev[k] < ev[mi]:
            mi = k
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
    mag = np.sqrt((fx * fx + fy * fy) + fz * fz)
    nrm = np.stack([fx / mag, fy / mag, fz / mag])  # 3 x n
    cams = sub_cams
    counts = [int((cams == 0).sum()), int((cams == 1).sum())]
    majority = 0 if counts[0] >= counts[1] else 1  # maxCoeff: first maximum
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
    sample = q.astype(np.float64)
    s2s = sample - cam_origins[majority]
    if dot(normal, s2s) > 0:
        normal = normal * -1.0
    if dot(binormal, s2s) > 0:
        binormal = binormal * -1.0
    axis = cross(normal, binormal)
    fr = np.zeros(1, O.FRAME_DTYPE)[0]
    fr["sample"], fr["normal"], fr["axis"], fr["binormal"] = sample, normal, axis, binormal
    fr["params"], fr["eigenvalue"], fr["n_nb"], fr["majority_cam"], fr["max_index"], fr["valid"] = params, ev[mi], n, majority, mx, 1
    # how far dggev's own answer moves when its input moves by one unit in the last place: the conditioning of the
    # reference's un-centred formulation, against which the oracle's distance from LAPACK is to be read
    rng = np.random.default_rng(sample_index)
    Mp = M * (1.0 + (rng.integers(0, 2, M.shape) * 2 - 1) * 2.0 ** -52)
    Mp = np.triu(Mp) + np.triu(Mp, 1).T
    ar2, _ai2, be2, _vl2, vr2, _w2, _i2 = lapack.dggev(np.asfortranarray(Mp), np.asfortranarray(N), compute_vl=0, compute_vr=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        ev2 = ar2 / be2
    m2 = 0
    for k in range(1, 9):
        if ev2[k] < ev2[m2]:
            m2 = k
    self_angle = float(angle(vr[:, mi], vr2[:, m2]))
    extra = {"self_angle": self_angle, "info": int(info), "beta_zero_at": [int(k) for k in np.nonzero(beta == 0)[0]], "complex": bool(np.any(alphai != 0)),
             "min_index": mi, "on_boundary": n_on_boundary, "cond_N9": float(np.linalg.cond(N[:9, :9]))}
    return fr, extra


def angle(u, v):
    c = np.abs((u * v).sum(-1)) / (np.linalg.norm(u, axis=-1) * np.linalg.norm(v, axis=-1))
    return np.arccos(np.clip(c, -1, 1))


def compare_lists(a, b):
    """Flip counts and max |delta| per field between two hypothesis lists, aligned by (sample, orientation)."""
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


def run_case(name, n_samples, w, rho, rand50=False, **search_kw):
    sc = synthetic.config(name)
    samples = sc.samples[:n_samples]
    p = O.default_params(sc.cam_origins, normals_mode=O.NORMALS_RAND50 if rand50 else O.NORMALS_DETERMINISTIC, rand_seed=1)
    frames = np.zeros(len(samples), O.FRAME_DTYPE)
    extras = []
    if rand50:
        import ctypes

        libc_rand()
        _LIBC.srand(ctypes.c_uint(1))
    for k, s in enumerate(samples):
        frames[k], ex = frame_lapack(sc.xyz, sc.cam, sc.cam_origins, int(s), rand50=rand50, **search_kw)
        extras.append(ex)
    res = O.hands_from_frames(p, sc.xyz, sc.cam, samples, frames, want_images=True)
    keep, sums = O.classify(res["images"], w, rho)
    own = O.find_hands(p, sc.xyz, sc.cam, samples, want_images=True)
    okeep, osums = O.classify(own["images"], w, rho)
    of = own["frames"]
    valid = of["valid"] != 0
    rep, common, ka, kb = compare_lists(res["hyps"], own["hyps"])
    pos_a = {(int(h["sample"]), int(h["orientation"])): i for i, h in enumerate(res["hyps"])}
    pos_b = {(int(h["sample"]), int(h["orientation"])): i for i, h in enumerate(own["hyps"])}
    rep.update({
        "case": name + ("_rand50" if rand50 else ""), "samples": int(len(samples)), "oracle_invalid_frames": int((~valid).sum()),
        "dggev_info_nonzero": int(sum(e["info"] != 0 for e in extras)),
        "dggev_one_ulp_input_max_angle_rad": float(max(e["self_angle"] for e in extras)),
        "dggev_one_ulp_input_median_angle_rad": float(np.median([e["self_angle"] for e in extras])),
        "median_angle_params_rad": float(np.median(angle(frames["params"][valid], of["params"][valid]))),
        "max_cond_N9": float(max(e["cond_N9"] for e in extras)),
        "dggev_complex_spectrum": int(sum(e["complex"] for e in extras)),
        "beta_zero_not_last": int(sum(e["beta_zero_at"] != [9] for e in extras)),
        "points_exactly_on_taubin_radius": int(sum(e["on_boundary"] for e in extras)),
        "n_nb_mismatch": int((frames["n_nb"] != of["n_nb"])[valid].sum()),
        "max_index_mismatch": int((frames["max_index"] != of["max_index"])[valid].sum()),
        "majority_cam_mismatch": int((frames["majority_cam"] != of["majority_cam"])[valid].sum()),
        "max_angle_params_rad": float(angle(frames["params"][valid], of["params"][valid]).max()),
        "max_abs_normal": float(np.abs(frames["normal"] - of["normal"])[valid].max()),
        "max_abs_axis_frame": float(np.abs(frames["axis"] - of["axis"])[valid].max()),
        "max_abs_binormal_frame": float(np.abs(frames["binormal"] - of["binormal"])[valid].max()),
        "max_rel_eigenvalue": float((np.abs(frames["eigenvalue"] - of["eigenvalue"]) / np.abs(of["eigenvalue"]))[valid].max()),
        "svm_label_flips": int(sum(int(keep[pos_a[k]]) != int(okeep[pos_b[k]]) for k in common)),
        "max_abs_svm_sum": float(max([abs(sums[pos_a[k]] - osums[pos_b[k]]) for k in common] or [0.0])),
        "image_pixels_differing": int(sum(int((res["images"][pos_a[k]] != own["images"][pos_b[k]]).sum()) for k in common)),
        "svm_kept": int(keep.sum()), "min_abs_svm_sum": float(np.abs(sums).min()) if len(sums) else None,
    })
    return sc, samples, frames, res["hyps"], keep, sums, rep


def main():
    w, rho = O.load_svm(os.path.join(ROOT, "tests", "golden", "svm_032015_linear_20_20_same"))
    store, report = {}, []
    for name, ns, r50 in CASES:
        sc, samples, frames, hyps, keep, sums, rep = run_case(name, ns, w, rho, rand50=r50)
        report.append(rep)
        print(json.dumps(rep))
        key = name + ("_rand50" if r50 else "")
        store[f"{key}_samples"] = samples.astype(np.int32)
        store[f"{key}_frames"] = frames
        store[f"{key}_hyps"] = hyps
        store[f"{key}_keep"] = keep
        store[f"{key}_sums"] = sums
    # FLANN sensitivity (the radius criterion and the order of equal distances are the builder's reading of FLANN)
    sens = []
    for variant, kw in (("inclusive_radius", dict(inclusive=True)), ("ties_descending", dict(ties_descending=True))):
        for name, ns, r50 in CASES:
            if r50:
                continue
            _, _, frames_v, hyps_v, keep_v, _, _ = run_case(name, ns, w, rho, **kw)
            base_h, base_k = store[f"{name}_hyps"], store[f"{name}_keep"]
            rep, common, ka, kb = compare_lists(hyps_v, base_h)
            pa = {(int(h["sample"]), int(h["orientation"])): i for i, h in enumerate(hyps_v)}
            pb = {(int(h["sample"]), int(h["orientation"])): i for i, h in enumerate(base_h)}
            rep.update({"variant": variant, "case": name,
                        "n_nb_changed": int((frames_v["n_nb"] != store[f"{name}_frames"]["n_nb"]).sum()),
                        "max_index_changed": int((frames_v["max_index"] != store[f"{name}_frames"]["max_index"]).sum()),
                        "svm_label_flips": int(sum(int(keep_v[pa[k]]) != int(base_k[pb[k]]) for k in common))})
            sens.append(rep)
            print(json.dumps(rep))
    store["report_json"] = np.frombuffer(json.dumps({"cases": report, "flann_sensitivity": sens}).encode(), np.uint8)
    out = os.path.join(ROOT, "tests", "golden", "e2e_lapack.npz")
    np.savez_compressed(out, **store)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
