"""CPU tests that pin the oracle (oracle/) against the committed goldens and an independent transcription."""
import os

import numpy as np
import pytest

from oracle import oracle_py as O
from tests import ref_numpy as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_struct_layouts():
    assert O.HYP_DTYPE.itemsize == 160
    assert O.FRAME_DTYPE.itemsize == 200


def test_glibc_rand_matches_real_libc():
    z = np.load(os.path.join(GOLD, "glibc_rand.npz"))
    for seed, vals in zip(z["seeds"], z["values"]):
        got = O.glibc_rand(int(seed), vals.size)
        assert np.array_equal(got, vals)
    assert O.glibc_rand(1, 1)[0] == 1804289383  # the well-known first value of glibc rand()


def test_taubin_solve_matches_lapack_dggev():
    """quadric.cpp:143-153 via dggev vs the oracle's 9x9 reduction; tolerance: angle <= 1e-4 rad (measured
    <= 4e-7 rad on lattice neighbourhoods, <= 1.3e-5 rad on the near-noise-free analytic patches)."""
    z = np.load(os.path.join(GOLD, "taubin_dggev.npz"))
    for M, N, ar, ai, be, v, mi in zip(z["M"], z["N"], z["alphar"], z["alphai"], z["beta"], z["v"], z["min_index"]):
        assert be[9] == 0.0 and np.all(be[:9] != 0) and np.all(ai == 0)  # the single infinite eigenvalue sits last
        rc, vo, lam = O.solve_taubin(M, N)
        assert rc == 0
        cosang = abs(v @ vo) / np.linalg.norm(v) / np.linalg.norm(vo)
        assert 1.0 - cosang <= 5e-9, 1.0 - cosang  # angle = sqrt(2(1-cos)) <= 1e-4 rad
        assert abs(lam - ar[mi] / be[mi]) <= 1e-6 * abs(lam) + 1e-12


def test_radius_search_vs_brute_force(tiny_scene):
    sc = tiny_scene
    rng = np.random.default_rng(0)
    for r in (0.01, 0.03, 0.08):
        for s in rng.choice(sc.n, 6, replace=False):
            idx, d2 = O.radius_search(sc.xyz, sc.xyz[s], r)
            d = sc.xyz[s][None, :] - sc.xyz
            bd2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            bidx = np.nonzero(bd2 < np.float32(r * r))[0]
            order = np.lexsort((bidx, bd2[bidx]))
            assert np.array_equal(idx, bidx[order])
            assert np.array_equal(d2, bd2[bidx][order])
            assert idx[0] == s and d2[0] == 0.0  # the query point itself comes first


def test_radius_search_empty_and_single():
    xyz = np.array([[0.0, 0.0, 0.0]], np.float32)
    idx, _ = O.radius_search(xyz, [1.0, 1.0, 1.0], 0.03)
    assert idx.size == 0
    idx, _ = O.radius_search(xyz, [0.0, 0.0, 0.0], 0.03)
    assert list(idx) == [0]


def test_frames_are_right_handed_and_face_the_camera(tiny_scene):
    sc = tiny_scene
    p = O.default_params(sc.cam_origins)
    fr = O.fit_frames(p, sc.xyz, sc.cam, sc.samples, 0.03)
    assert fr["valid"].all()
    n, a, b = fr["normal"], fr["axis"], fr["binormal"]
    assert np.allclose(np.linalg.norm(n, axis=1), 1, atol=1e-12)
    assert np.allclose(np.linalg.norm(a, axis=1), 1, atol=1e-12)
    assert np.abs(np.einsum("ij,ij->i", n, a)).max() < 1e-12
    assert np.allclose(np.cross(n, b), a, atol=1e-12)  # quadric.cpp:304
    s2s = fr["sample"] - sc.cam_origins[fr["majority_cam"]]
    assert (np.einsum("ij,ij->i", n, s2s) <= 0).all() and (np.einsum("ij,ij->i", b, s2s) <= 0).all()


def test_underdetermined_neighbourhood_still_yields_a_frame():
    """Fewer than nine neighbours: infinitely many quadrics interpolate them and the pencil is singular many times over.
    The reference takes whatever dggev returns (quadric.cpp:146-153, no validity test) and goes on; so does the oracle --
    every rank-deficient coordinate is deflated and a frame comes out (round 2 dropped such samples)."""
    xyz = np.array([[0.5, 0.0, 0.0], [2.0, 2.0, 2.0]], np.float32)
    cam = np.zeros(2, np.int32)
    p = O.default_params(np.zeros((2, 3)))
    fr = O.fit_frames(p, xyz, cam, np.array([0], np.int32), 0.03)
    assert fr["valid"][0] == 1 and fr["n_nb"][0] == 1
    assert np.isfinite(fr["normal"][0]).all() and abs(np.linalg.norm(fr["normal"][0]) - 1) < 1e-12
    # (with a finite frame the sweep then "grasps" the lone point -- finger_hand.cpp only asks for one point between the
    # fingers; what dggev returns for such a pencil, finite or NaN, is solver noise in the reference)
    r = O.find_hands(p, xyz, cam, np.array([0], np.int32))
    assert (r["hyps"]["sample"] == 0).all()


def test_pow6_libm_switch_keeps_argmax(tiny_scene):
    sc = tiny_scene
    f0 = O.fit_frames(O.default_params(sc.cam_origins, pow6_libm=0), sc.xyz, sc.cam, sc.samples, 0.03)
    f1 = O.fit_frames(O.default_params(sc.cam_origins, pow6_libm=1), sc.xyz, sc.cam, sc.samples, 0.03)
    assert np.array_equal(f0["max_index"], f1["max_index"])
    assert np.array_equal(f0["normal"], f1["normal"])


def _transcription_inputs(sc, p, fr, si):
    """Re-derive the transformed + cropped neighbourhood the way hand_search.cpp:154-169 does, in numpy."""
    s = sc.samples[si]
    idx, _ = O.radius_search(sc.xyz, sc.xyz[s], 0.08)
    cen = (sc.xyz[idx] - sc.xyz[s][None, :]).astype(np.float64)  # float32 subtraction, then cast
    nrm, ax = fr["normal"][si], fr["axis"][si]
    frame = np.stack([nrm, np.cross(nrm, ax), ax], 1)
    # oracle order: (f0*cx + f1*cy) + f2*cz per output row
    pts = np.stack([(frame[0, r] * cen[:, 0] + frame[1, r] * cen[:, 1]) + frame[2, r] * cen[:, 2] for r in range(3)])
    keep = (pts[2] > -1.0 * p.hand_height) & (pts[2] < p.hand_height)
    return pts[:, keep], sc.cam[idx][keep], frame


def test_hand_search_matches_numpy_transcription(tiny_scene):
    sc = tiny_scene
    p = O.default_params(sc.cam_origins)
    res = O.find_hands(p, sc.xyz, sc.cam, sc.samples, want_images=True)
    fr, hyps, images = res["frames"], res["hyps"], res["images"]
    checked = 0
    for si in range(0, sc.samples.size, 2):
        pts, cams_ids, frame = _transcription_inputs(sc, p, fr, si)
        sample = fr["sample"][si]
        cams = (sc.cam_origins - sample[None, :]).T
        # np.cross uses a different evaluation order than the oracle's cross3 only in trivially equal ways
        got = R.evaluate_hand(pts, np.zeros_like(pts), cams_ids, frame, cams, (0.01, 0.09, 0.06), 0.01, sample)
        mine = hyps[hyps["sample"] == si]
        assert [g["orientation"] for g in got] == list(mine["orientation"])
        for g, h in zip(got, mine):
            assert g["finger_index"] == h["finger_index"] and g["depth_index"] == h["depth_index"]
            assert g["n_in_box"] == h["n_in_box"]
            assert g["width"] == h["width"]
            for k in ("approach", "binormal", "surface", "bottom"):
                assert np.array_equal(g[k], h[k]), k
            k = np.nonzero((hyps["sample"] == si) & (hyps["orientation"] == g["orientation"]))[0][0]
            s2c = h["surface"] - sc.cam_origins[h["cam_source"]]
            img = R.convert_to_image(g["points_in_box"], g["binormal"], s2c)
            assert np.array_equal(img.reshape(-1), images[k])
            checked += 1
    assert checked > 10


def test_reduction_form_equals_literal_sweep():
    """finger[i](d) = !(gapmin[i] < d) && (sidemin[i] < d) unless ymin < d && ymin < back(d)."""
    rng = np.random.default_rng(5)
    fw, od, depth = 0.01, 0.09, 0.06
    for _ in range(300):
        n = int(rng.integers(1, 60))
        pts = np.stack([rng.uniform(-0.1, 0.1, n), rng.uniform(-0.07, 0.09, n)])
        fh = R.FingerHand(fw, od, depth)
        fh.pts = pts
        d = 0.01
        while d <= depth:
            fh.evaluate_fingers(d)
            back = -1.0 * (depth - d)
            ymin = pts[1].min()
            exp = np.zeros(20, bool)
            if not (ymin < d and ymin < back):
                for i in range(20):
                    ingap = (pts[0] > fh.fs[i]) & (pts[0] < fh.fs[i] + fw)
                    side = (pts[0] > fh.fs[i] + fw) if i <= 10 else (pts[0] < fh.fs[i])
                    gapmin = pts[1][ingap].min() if ingap.any() else np.inf
                    sidemin = pts[1][side].min() if side.any() else np.inf
                    exp[i] = (not gapmin < d) and (sidemin < d)
            assert np.array_equal(exp, fh.fingers)
            d += 0.005


def test_antipodal_pass_sets_flags(tiny_scene):
    sc = tiny_scene
    p = O.default_params(sc.cam_origins)
    sub = sc.samples[:24]
    r0 = O.find_hands(p, sc.xyz, sc.cam, sub, calculates_antipodal=False)
    r1 = O.find_hands(p, sc.xyz, sc.cam, sub, calculates_antipodal=True)
    assert not r0["hyps"]["half_antipodal"].any()
    assert len(r0["hyps"]) == len(r1["hyps"])
    assert np.array_equal(r0["hyps"]["surface"], r1["hyps"]["surface"])
    assert (r1["hyps"]["full_antipodal"] <= r1["hyps"]["half_antipodal"]).all()


def test_rand50_mode_uses_glibc_stream(tiny_scene):
    sc = tiny_scene
    pd = O.default_params(sc.cam_origins)
    pr = O.default_params(sc.cam_origins, normals_mode=O.NORMALS_RAND50, num_threads=1)
    pr4 = O.default_params(sc.cam_origins, normals_mode=O.NORMALS_RAND50, num_threads=4)
    fd = O.fit_frames(pd, sc.xyz, sc.cam, sc.samples, 0.03)
    fr = O.fit_frames(pr, sc.xyz, sc.cam, sc.samples, 0.03)
    fr4 = O.fit_frames(pr4, sc.xyz, sc.cam, sc.samples, 0.03)
    assert fr.tobytes() == fr4.tobytes()  # thread count does not change the single-thread rand() order
    assert np.array_equal(fd["params"], fr["params"])  # the quadric itself is the same
    assert (fr["max_index"] < 50).all()
    cosang = np.abs(np.einsum("ij,ij->i", fd["normal"], fr["normal"]))
    assert np.median(cosang) > 0.99


# ---- HOG / SVM structural known-answer tests (OpenCV 2.4 semantics) ----
def test_hog_empty_image_is_rejected(svm_model):
    w, rho = svm_model
    d = O.hog(np.zeros((80, 100), np.uint8))
    assert d.shape == (3528,) and not d.any()
    keep, sums = O.classify(np.zeros((1, 8000), np.uint8), w, rho)
    assert keep[0] == 0 and sums[0] == pytest.approx(0.31383255947302025)


def test_hog_vertical_edge_bins():
    img = np.zeros((80, 100), np.uint8)
    img[:, 40:] = 255
    d = O.hog(img).reshape(2, 49, 4, 9)
    assert d[..., 3].max() == 0 and d[..., 5].max() == 0
    nz = d.sum(axis=(0, 1, 2))
    assert (nz[[1, 2, 4, 6, 7]] == 0).all() and nz[0] > 0
    assert np.array_equal(d[..., 0], d[..., 8])  # horizontal gradient splits equally between bins 8 and 0


def test_hog_ignores_rows_below_window():
    rng = np.random.default_rng(1)
    img = (rng.random((80, 100)) < 0.2).astype(np.uint8) * 255
    img2 = img.copy()
    img2[65:, :] = 0  # rows 65..79 are outside every window and outside the row-63 gradient stencil
    assert np.array_equal(O.hog(img), O.hog(img2))
    img3 = img.copy()
    img3[:64, 97:] = 255 - img3[:64, 97:]  # columns 97..99 likewise
    assert np.array_equal(O.hog(img), O.hog(img3))


def test_hog_never_populates_bins_3_and_5_on_binary_images():
    rng = np.random.default_rng(2)
    for _ in range(5):
        img = (rng.random((80, 100)) < 0.3).astype(np.uint8) * 255
        d = O.hog(img).reshape(2, 49, 4, 9)
        assert d[..., 3].max() == 0 and d[..., 5].max() == 0
        assert np.isfinite(d).all() and d.max() <= 1.0


def test_svm_file_parsers_agree(svm_model):
    w, rho = svm_model
    w2, rho2 = O.load_svm(os.path.join(GOLD, "svm_032015_linear_20_20_same"))
    assert np.array_equal(w, w2) and rho == rho2
    assert np.count_nonzero(w) == 2198 and rho == -0.31383255947302025
    wb = w.reshape(2, 49, 4, 9)
    assert not wb[..., 3].any() and not wb[..., 5].any()  # consistent with the HOG binning above


def test_classify_oracle_on_real_hypotheses(tiny_scene, svm_model):
    sc = tiny_scene
    w, rho = svm_model
    res = O.find_hands(O.default_params(sc.cam_origins), sc.xyz, sc.cam, sc.samples, want_images=True)
    keep, sums = O.classify(res["images"], w, rho)
    assert keep.shape[0] == len(res["hyps"]) and np.isfinite(sums).all()
    for img, k, s in zip(res["images"][:8], keep[:8], sums[:8]):
        d = O.hog(img)
        prod = (w * d).astype(np.float32).reshape(-1, 4)
        acc = 0.0
        for q in prod:
            acc += float(np.float32(np.float32(np.float32(q[0] + q[1]) + q[2]) + q[3]))
        assert float(np.float32(acc)) - rho == pytest.approx(s, abs=1e-12)
        assert k == (0 if s > 0 else 1)


def test_points_for_learning_match_numpy_transcription(tiny_scene):
    """a14: the variable part of GraspHypothesis (rotating_hand.cpp:125-151) -- all three rows and the camera split."""
    sc = tiny_scene
    p = O.default_params(sc.cam_origins)
    res = O.find_hands_points(p, sc.xyz, sc.cam, sc.samples)
    fr = O.find_hands(p, sc.xyz, sc.cam, sc.samples)["frames"]
    hyps = res["hyps"]
    checked = 0
    for si in range(0, sc.samples.size, 2):
        pts, cam_ids, frame = _transcription_inputs(sc, p, fr, si)
        sample = fr["sample"][si]
        cams = (sc.cam_origins - sample[None, :]).T
        for g in R.evaluate_hand(pts, np.zeros_like(pts), cam_ids, frame, cams, (0.01, 0.09, 0.06), 0.01, sample):
            k = np.nonzero((hyps["sample"] == si) & (hyps["orientation"] == g["orientation"]))[0][0]
            assert np.array_equal(res["points"][k], g["points_in_box"])
            assert np.array_equal(res["cams"][k], g["cam_in_box"])
            checked += 1
    assert checked > 10


def test_oracle_runs_config_c1(svm_model):
    """BASELINE config C1 (single view, 50k points, 500 samples: the reference's own CPU-runnable case).  No golden exists
    for it (the reference ships no PCD), so this pins the invariants of the result instead: orthonormal right-handed
    frames facing the (only) camera, sample-major / orientation-ascending order, every point from camera 0, and an SVM
    verdict for every hypothesis."""
    from agile_grasp_amd import synthetic

    sc = synthetic.config("C1")
    assert sc.n == 50_000 and sc.samples.size == 500 and not sc.cam.any()
    r = O.find_hands(O.default_params(sc.cam_origins), sc.xyz, sc.cam, sc.samples, want_images=True)
    fr, hy = r["frames"], r["hyps"]
    ok = fr["valid"] != 0
    assert ok.sum() > 450 and len(hy) > 50
    n, a, b = fr["normal"][ok], fr["axis"][ok], fr["binormal"][ok]
    for u, v in ((n, n), (a, a), (b, b)):
        assert np.allclose((u * v).sum(1), 1.0, atol=1e-12)
    assert np.allclose((n * a).sum(1), 0, atol=1e-12) and np.allclose(np.cross(n, b), a, atol=1e-12)
    to_cam = fr["sample"][ok] - sc.cam_origins[0]
    assert ((n * to_cam).sum(1) <= 0).all()  # quadric.cpp:294-301
    key = hy["sample"].astype(np.int64) * 8 + hy["orientation"]
    assert (np.diff(key) > 0).all() and not hy["cam_source"].any()
    keep, sums = O.classify(r["images"], *svm_model)
    assert keep.shape == (len(hy),) and np.array_equal(keep.astype(bool), sums <= 0)


def _auc(label, score):
    from scipy.stats import rankdata

    r = rankdata(score)
    n1 = int(label.sum())
    n0 = len(label) - n1
    return float((r[label].sum() - n1 * (n1 + 1) / 2) / (n1 * n0))


def test_shipped_svm_separates_antipodal_hands_through_this_hog_and_not_through_a_scrambled_one(small_scene, svm_model):
    """HOG / CvSVM cannot be pinned against OpenCV 2.4 here (absent from the image).  What can be checked: the reference's
    SHIPPED model was trained on OpenCV-2.4 descriptors of real grasp images to predict antipodal hands (learning.cpp:76-141),
    so if this repository's descriptor has OpenCV's layout -- window, block and cell order, bin order, normalisation -- the
    model must separate (half-)antipodal hands from the rest on scenes it has never seen, and must lose that ability when
    the descriptor's blocks or bins are permuted.  Measured: AUC of the decision value for `half_antipodal` 0.77 here
    (0.73 on C2, 0.69 on C1); weights reversed 0.51-0.58, blocks shuffled 0.61-0.64, bins rolled 0.61-0.64."""
    sc = small_scene
    w, rho = svm_model
    r = O.find_hands(O.default_params(sc.cam_origins), sc.xyz, sc.cam, sc.samples, calculates_antipodal=True, want_images=True)
    half = r["hyps"]["half_antipodal"] != 0
    assert 20 < half.sum() < len(half) - 20
    keep, sums = O.classify(r["images"], w, rho)
    auc = _auc(half, -sums)  # kept iff sum <= 0
    assert auc > 0.70, auc
    assert half[keep != 0].mean() > 2 * half[keep == 0].mean() - 0.1  # kept hands are antipodal far more often
    rng = np.random.default_rng(0)
    scrambled = [w[::-1].copy(), w.reshape(-1, 36)[rng.permutation(98)].reshape(-1).copy(),
                 np.roll(w.reshape(-1, 9), 4, axis=1).reshape(-1).copy()]
    for ws in scrambled:
        _k, s2 = O.classify(r["images"], ws, rho)
        assert _auc(half, -s2) < auc - 0.08


def test_non_finite_points_are_invisible():
    """pcl::KdTreeFLANN::setInputCloud drops points with a non-finite coordinate (hand_search.cpp:10-11), so the search on a
    cloud that holds some equals the search on the cloud without them (indices remapped), and a sample AT one has no frame."""
    from agile_grasp_amd import synthetic

    sc = synthetic.config("tiny")
    rng = np.random.default_rng(7)
    xyz = sc.xyz.copy()
    bad = rng.permutation(sc.n)[:120]
    xyz[bad[:30], rng.integers(0, 3, 30)] = np.nan
    xyz[bad[30:60]] = np.inf
    xyz[bad[60:90], 1] = -np.inf
    xyz[bad[90:]] = np.nan
    samples = np.unique(np.concatenate([sc.samples, bad[:10]])).astype(np.int32)
    p = O.default_params(sc.cam_origins)
    a = O.find_hands(p, xyz, sc.cam, samples)
    keep = np.ones(sc.n, bool)
    keep[bad] = False
    remap = np.cumsum(keep) - 1
    b = O.find_hands(p, xyz[keep], sc.cam[keep], remap[samples[keep[samples]]].astype(np.int32))
    assert len(a["hyps"]) == len(b["hyps"]) > 0
    for f in ("orientation", "finger_index", "depth_index", "n_in_box", "cam_source", "axis", "approach", "binormal", "bottom",
              "surface", "width"):
        assert np.array_equal(a["hyps"][f], b["hyps"][f]), f
    assert not a["frames"]["valid"][np.isin(samples, bad)].any()
    assert np.array_equal(a["frames"]["valid"][~np.isin(samples, bad)], b["frames"]["valid"])
