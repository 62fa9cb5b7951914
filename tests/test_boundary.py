"""The drop-in boundary as the reference's nodes use it (grasp_localizer.cpp:21, 95-103; nodes/test.cpp:72):
Localization(filters_boundaries = true), predictAntipodalHands on a FILTERED list, lists that outlive their search,
GraspHypothesis::getPointsForLearning as a lazy accessor, the empty-`indices` sampling path (pcl::RandomSample)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from tests.test_cpp_adapter import GOLD, ROOT, _dump_raw, _preprocess_numpy, _raw_cloud

SVM = os.path.join(GOLD, "svm_032015_linear_20_20_same")


def _build(tmp_path):
    from agile_grasp_amd import build

    build.build()
    exe = str(tmp_path / "boundary_test")
    libdir = os.path.join(ROOT, "agile_grasp_amd", "lib")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "boundary_test.cpp"), "-o", exe, "-L" + libdir,
                           "-lagile_grasp_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _algorithm_a(n_points, num_samples, seed):
    """pcl::RandomSample::applyFilter(indices) of PCL 1.7 (filters/impl/random_sample.hpp) on this host's libc rand()."""
    libc = ctypes.CDLL("libc.so.6")
    libc.rand.restype = ctypes.c_int
    rand_max = 2147483647
    N = n_points
    if num_samples >= N:
        return list(range(N))
    if num_samples <= 0:
        return []
    libc.srand(ctypes.c_uint(seed))
    unif = lambda: np.float32(libc.rand() / float(rand_max))
    out = []
    top, index = N - num_samples, 0
    for n in range(num_samples, 1, -1):
        V = unif()
        S = 0
        quot = np.float32(top) / np.float32(N)
        while quot > V:
            S += 1
            top -= 1
            N -= 1
            quot = np.float32(quot * np.float32(top)) / np.float32(N)
        index += S
        out.append(index)
        index += 1
        N -= 1
    index += N * int(unif())  # (PCL casts the variate, not the product)
    out.append(index)
    return out


@pytest.mark.parametrize("n_points,num_samples,seed", [(1000, 10, 7), (50000, 500, 1), (300000, 2000, 12345), (64, 64, 3),
                                                        (64, 100, 3), (10, 1, 5), (10, 0, 5), (2, 1, 9)])
def test_random_sample_is_pcl_algorithm_a(tmp_path, n_points, num_samples, seed):
    """hand_search.cpp:31-44 with no indices: the adapter's sampler is PCL 1.7's selection sampling on the host's rand()."""
    exe = _build(tmp_path)
    out = subprocess.run([exe, "sampler", str(n_points), str(num_samples), str(seed)], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0
    got = [int(l.split()[1]) for l in out.stdout.splitlines() if l.startswith("S ")]
    exp = _algorithm_a(n_points, num_samples, seed)
    assert got == exp
    assert len(got) == min(max(num_samples, 0), n_points)
    assert all(0 <= a < b < n_points for a, b in zip(got, got[1:])) or len(got) <= 1  # ascending, distinct, in range


def _surf(lines, tag):
    return [[float(v) for v in l.split()[1:5]] + [int(l.split()[5])] for l in lines if l.startswith(tag + " ")]


@pytest.mark.gpu
def test_filtered_list_classifies_like_the_c_abi(tmp_path, svm_model):
    """with filters_boundaries = true (the ROS node's configuration) the list handed to
    predictAntipodalHands is shorter than the device-side list; the result must be the C ABI's labels, filtered."""
    from agile_grasp_amd import binding

    exe = _build(tmp_path)
    xyz, size_left, ws, cams = _raw_cloud()
    ws = ws.copy()
    ws[1] = 0.8  # cuts through the scene: a dozen hands end up within 2 cm of the boundary
    vox, vcam = _preprocess_numpy(xyz, size_left, ws)
    idx = np.sort(np.random.default_rng(0).permutation(len(vox))[:96]).astype(np.int32)
    path = str(tmp_path / "raw.bin")
    _dump_raw(path, xyz, size_left, idx, ws, cams)
    out = subprocess.run([exe, "filtered", path, SVM], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.splitlines()
    ctx = binding.Context(cams)
    ctx.set_cloud(vox, vcam)
    hyps = ctx.find_hands(idx)
    ctx.load_svm(*svm_model)
    keep = ctx.classify().astype(bool)
    s = hyps["surface"]
    near = np.zeros(len(hyps), bool)
    for k in range(6):  # Localization::filterHands, localization.cpp:364-388
        near |= np.abs(s[:, k // 2] - ws[k]) < 0.02
    assert near.sum() >= 2 and (~near).sum() >= 2 and (keep & ~near).sum() >= 1
    res = [l for l in lines if l.startswith("RESULT")][0].split()
    assert [int(v) for v in res[1:]] == [int((~near).sum()), int((keep & ~near).sum())]
    exp_h = [[float(h["surface"][0]), float(h["surface"][1]), float(h["surface"][2]), float(h["width"]), 0]
             for h in hyps[~near]]
    exp_k = [[float(h["surface"][0]), float(h["surface"][1]), float(h["surface"][2]), float(h["width"]), 1]
             for h in hyps[keep & ~near]]
    assert _surf(lines, "H") == exp_h and _surf(lines, "K") == exp_k
    cl = [l for l in lines if l.startswith("CLOUD")][0].split()
    assert int(cl[1]) == len(xyz) and int(cl[2]) == int(np.isfinite(xyz).all(1).sum())  # NaN points removed in place
    assert "CLUSTERING 0" in lines and any("uses_clustering" in l and "Error" in l for l in lines)
    hd, _ = ctx.find_handles(hyps[keep & ~near], 3, 0.005)
    assert f"HANDLES {len(hd)}" in lines


@pytest.mark.gpu
def test_lists_that_outlive_their_search(tmp_path, svm_model):
    """hypotheses of an earlier localizeHands must never be matched against the device state of a
    later one.  They classify through the images they carry; their lazy point accessor refuses loudly."""
    from agile_grasp_amd import binding

    exe = _build(tmp_path)
    xyz, size_left, ws, cams = _raw_cloud()
    vox, vcam = _preprocess_numpy(xyz, size_left, ws)
    idx = np.sort(np.random.default_rng(0).permutation(len(vox))[:96]).astype(np.int32)
    path = str(tmp_path / "raw.bin")
    _dump_raw(path, xyz, size_left, idx, ws, cams)
    out = subprocess.run([exe, "stale", path, SVM], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.splitlines()
    ctx = binding.Context(cams)
    ctx.set_cloud(vox, vcam)
    ctx.load_svm(*svm_model)
    h1 = ctx.find_hands(idx[:48])
    k1 = ctx.classify().astype(bool)
    p1 = ctx.learning_points(0)
    h2 = ctx.find_hands(idx[48:])
    k2 = ctx.classify().astype(bool)
    p2 = ctx.learning_points(0)
    assert len(h1) >= 2 and len(h2) >= 1 and k1.sum() + k2.sum() >= 1
    assert len(set(h1["epoch"])) == 1 and len(set(h2["epoch"])) == 1 and h1["epoch"][0] != h2["epoch"][0] != 0
    rec = lambda hs, lab: [[float(h["surface"][0]), float(h["surface"][1]), float(h["surface"][2]), float(h["width"]), lab]
                           for h in hs]
    res = [int(v) for v in [l for l in lines if l.startswith("RESULT")][0].split()[1:]]
    assert res == [len(h1), len(h2), int(k1.sum()), int(k2.sum()), int(k1.sum() + k2.sum())]
    assert _surf(lines, "H1") == rec(h1, 0) and _surf(lines, "H2") == rec(h2, 0)
    assert _surf(lines, "K1") == rec(h1[k1], 1) and _surf(lines, "K2") == rec(h2[k2], 1)
    assert _surf(lines, "K12") == rec(h2[k2], 1) + rec(h1[k1], 1)  # input order preserved (learning.cpp:236-243)
    assert f"AGAIN {int(k1.sum())}" in lines  # Learning(int) with every search gone

    def points_line(tag):
        t = [l for l in lines if l.startswith(tag + " ")][0].split()
        return [int(t[1]), int(t[2]), int(t[3]), int(t[4])], [float(v) for v in t[5:8]]

    def expect(p):
        pts, cam = p
        sums = [0.0, 0.0, 0.0]
        for k in range(pts.shape[1]):
            for q in range(3):
                sums[q] += float(pts[q, k])
        return [0, pts.shape[1], int((cam == 0).sum()), int((cam == 1).sum())], sums

    assert points_line("P1") == tuple(expect(p1)) and points_line("P1AGAIN") == tuple(expect(p1))
    assert points_line("P2") == tuple(expect(p2))
    assert points_line("P1STALE") == ([1, 0, 0, 0], [0.0, 0.0, 0.0])
    assert any("getPointsForLearning" in l and "Error" in l for l in lines)


@pytest.mark.gpu
def test_empty_indices_draws_the_samples(tmp_path):
    """hand_search.cpp:31-44: an empty `indices` draws num_samples indices (pcl::RandomSample) and
    hands_cam_source(i) = pts_cam_source(index i); the search then equals the C ABI's on those indices."""
    from agile_grasp_amd import binding

    exe = _build(tmp_path)
    xyz, size_left, ws, cams = _raw_cloud()
    vox, vcam = _preprocess_numpy(xyz, size_left, ws)
    n_left = int((vcam == 0).sum())
    assert np.array_equal(vcam, (np.arange(len(vox)) >= n_left).astype(np.int32))
    path = str(tmp_path / "vox.bin")
    _dump_raw(path, vox, n_left, [], ws, cams)
    out = subprocess.run([exe, "drawn", path, SVM], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.splitlines()
    drawn = [int(l.split()[1]) for l in lines if l.startswith("S ")]
    assert drawn == _algorithm_a(len(vox), 40, 7)
    ctx = binding.Context(cams)
    ctx.set_cloud(vox, vcam)
    hyps = ctx.find_hands(np.asarray(drawn, np.int32))
    assert len(hyps) > 0 and f"RESULT {len(hyps)}" in lines
    assert _surf(lines, "H") == [[float(h["surface"][0]), float(h["surface"][1]), float(h["surface"][2]),
                                  float(h["width"]), 0] for h in hyps]
