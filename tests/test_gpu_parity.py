"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Tolerance: none.  Integer fields, labels, images and every floating-point field (fp64 frames / hypotheses, fp32 HOG
descriptors, fp64 SVM sums) are compared with exact equality -- the kernels evaluate the oracle's operation order
without FMA contraction, and agh_selftest_math checks the IEEE assumptions on the device.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FLOAT_FIELDS = ("axis", "approach", "binormal", "bottom", "surface", "width")
INT_FIELDS = ("sample", "orientation", "cam_source", "n_in_box", "half_antipodal", "full_antipodal", "valid",
              "finger_index", "depth_index")


def _ctx(sc, **kw):
    from agile_grasp_amd import binding

    return binding.Context(sc.cam_origins, **kw)


def assert_hyps_equal(got, ref):
    assert len(got) == len(ref)
    for f in INT_FIELDS + FLOAT_FIELDS:
        assert np.array_equal(got[f], ref[f]), f


def assert_frames_equal(got, ref):
    for f in ("valid", "n_nb", "majority_cam", "max_index", "sample", "params", "eigenvalue", "normal", "axis",
              "binormal"):
        assert np.array_equal(got[f], ref[f]), f


def test_device_arithmetic_is_ieee_identical(tiny_scene):
    ctx = _ctx(tiny_scene)
    assert ctx.selftest_math(1 << 21, seed=7) == 0


@pytest.mark.parametrize("scene_name", ["tiny", "small"])
def test_full_path_bit_exact(scene_name, svm_model):
    from agile_grasp_amd import synthetic
    from oracle import oracle_py as O

    sc = synthetic.config(scene_name)
    w, rho = svm_model
    ctx = _ctx(sc)
    ctx.set_cloud(sc.xyz, sc.cam)
    hyps = ctx.find_hands(sc.samples)
    ref = O.find_hands(O.default_params(sc.cam_origins), sc.xyz, sc.cam, sc.samples, want_images=True)
    assert_frames_equal(ctx.frames(), ref["frames"])
    nt, nh = ctx.neighbor_counts()
    assert np.array_equal(nt, ref["frames"]["n_nb"]) and np.array_equal(nh, ref["nh"])
    assert len(hyps) > 10
    assert_hyps_equal(hyps, ref["hyps"])
    assert np.array_equal(ctx.images(), ref["images"])
    ctx.load_svm(w, rho)
    keep = ctx.classify()
    desc, sums = ctx.hog()
    okeep, osums = O.classify(ref["images"], w, rho)
    assert np.array_equal(desc, np.stack([O.hog(i) for i in ref["images"]]))
    assert np.array_equal(sums, osums) and np.array_equal(keep, okeep)
    assert 0 < keep.sum() < keep.size


@pytest.mark.parametrize("geom", [
    dict(finger_width=0.012, hand_outer_diameter=0.11, hand_depth=0.07, hand_height=0.025, init_bite=0.015),
    dict(finger_width=0.008, hand_outer_diameter=0.075, hand_depth=0.045, hand_height=0.015, init_bite=0.005),
    dict(nn_radius_taubin=0.025, nn_radius_hands=0.07),
    # radii that change the grid (cell = max(0.02, r_hands / 4)) and the number of rows a ball touches
    dict(nn_radius_taubin=0.02, nn_radius_hands=0.05),
    dict(nn_radius_taubin=0.04, nn_radius_hands=0.1, hand_outer_diameter=0.12, hand_depth=0.08),
    # finger width = slot spacing: coincident thresholds share a look-up cell (the 4-probe kernel instantiation)
    dict(finger_width=0.01, hand_outer_diameter=0.1),
    dict(finger_width=0.015, hand_outer_diameter=0.105, hand_depth=0.05, init_bite=0.02, hand_height=0.03),
])
def test_other_hand_geometries_bit_exact(tiny_scene, geom):
    """Finger-slot thresholds, bite depths, look-up tables and radii all derive from the parameters: check parity for
    geometries other than the node defaults (learning_test.cpp uses init_bite 0.015, for instance)."""
    from oracle import oracle_py as O

    sc = tiny_scene
    ctx = _ctx(sc, **geom)
    ctx.set_cloud(sc.xyz, sc.cam)
    hyps = ctx.find_hands(sc.samples)
    ref = O.find_hands(O.default_params(sc.cam_origins, **geom), sc.xyz, sc.cam, sc.samples, want_images=True)
    assert_frames_equal(ctx.frames(), ref["frames"])
    assert len(hyps) > 5
    assert_hyps_equal(hyps, ref["hyps"])
    assert np.array_equal(ctx.images(), ref["images"])


def test_svm_file_loader_matches_memory_loader(tiny_scene, svm_model):
    sc = tiny_scene
    w, rho = svm_model
    a, b = _ctx(sc), _ctx(sc)
    for c in (a, b):
        c.set_cloud(sc.xyz, sc.cam)
        c.find_hands(sc.samples)
    a.load_svm(w, rho)
    b.load_svm_file(os.path.join(GOLD, "svm_032015_linear_20_20_same"))
    assert np.array_equal(a.hog()[1], b.hog()[1])


def test_antipodal_pass_bit_exact(tiny_scene):
    """calculates_antipodal: normals for ALL points (r = 0.01) then Antipodal::evaluateGrasp labels."""
    from oracle import oracle_py as O

    sc = tiny_scene
    ctx = _ctx(sc)
    ctx.set_cloud(sc.xyz, sc.cam)
    sub = sc.samples[:40]
    hyps = ctx.find_hands(sub, calculates_antipodal=True)
    p = O.default_params(sc.cam_origins)
    ref = O.find_hands(p, sc.xyz, sc.cam, sub, calculates_antipodal=True)
    assert_hyps_equal(hyps, ref["hyps"])
    assert hyps["half_antipodal"].any()
    # the per-point normals themselves
    allp = np.arange(sc.n, dtype=np.int32)
    fr = O.fit_frames(p, sc.xyz, sc.cam, allp, 0.01)
    exp = np.where(fr["valid"][:, None] != 0, fr["normal"], 0.0)
    f2 = O.fit_frames(p, sc.xyz, sc.cam, sub, 0.03)
    exp[sub] = np.where(f2["valid"][:, None] != 0, f2["normal"], exp[sub])  # hand_search.cpp:102 in the sample pass
    assert np.array_equal(ctx.normals(), exp)


def test_rand50_mode_bit_exact(tiny_scene):
    """HandSearch's default (uses_determinstic_normal_estimation_ = false): 50 glibc rand() draws per sample."""
    from agile_grasp_amd import binding
    from oracle import oracle_py as O

    sc = tiny_scene
    for seed in (1, 42):
        ctx = _ctx(sc, normals_mode=binding.NORMALS_RAND50, rand_seed=seed)
        ctx.set_cloud(sc.xyz, sc.cam)
        hyps = ctx.find_hands(sc.samples)
        p = O.default_params(sc.cam_origins, normals_mode=O.NORMALS_RAND50, rand_seed=seed)
        ref = O.find_hands(p, sc.xyz, sc.cam, sc.samples)
        assert_frames_equal(ctx.frames(), ref["frames"])
        assert_hyps_equal(hyps, ref["hyps"])


def _unstamped(h):
    """Records without their per-call stamp (agh_hypothesis::epoch differs between any two calls by design)."""
    h = h.copy()
    assert (h["epoch"] != 0).all() and len(set(h["epoch"].tolist())) <= 1
    h["epoch"] = 0
    return h.tobytes()


def test_pointxyzrgba_stride(tiny_scene):
    """pcl::PointXYZRGBA layout: 32-byte points, xyz at offset 0 (agh_set_cloud stride_bytes = 32)."""
    sc = tiny_scene
    a, b = _ctx(sc), _ctx(sc)
    a.set_cloud(sc.xyz, sc.cam)
    wide = np.zeros((sc.n, 8), np.float32)
    wide[:, :3] = sc.xyz
    wide[:, 3:] = 7.0
    b.set_cloud(wide, sc.cam)
    assert _unstamped(a.find_hands(sc.samples)) == _unstamped(b.find_hands(sc.samples))


def test_sharding_and_order_properties(small_scene):
    """Samples are independent: a slice of the samples gives the slice of the results (basis of the multi-GPU
    sharding), a permutation of the samples permutes them, and a repeat call is byte-identical."""
    sc = small_scene
    ctx = _ctx(sc)
    ctx.set_cloud(sc.xyz, sc.cam)
    full = ctx.find_hands(sc.samples)
    again = ctx.find_hands(sc.samples)
    assert _unstamped(full) == _unstamped(again) and full["epoch"][0] != again["epoch"][0]
    half = sc.samples.size // 2
    a = ctx.find_hands(sc.samples[:half])
    b = ctx.find_hands(sc.samples[half:])
    b["sample"] += half
    assert _unstamped(a) + _unstamped(b) == _unstamped(full)
    perm = np.random.default_rng(0).permutation(sc.samples.size)
    shuf = ctx.find_hands(sc.samples[perm])
    shuf["sample"] = perm[shuf["sample"]]
    order = np.lexsort((shuf["orientation"], shuf["sample"]))
    assert _unstamped(shuf[order]) == _unstamped(full)


def test_edge_cases(tiny_scene):
    from agile_grasp_amd import binding

    sc = tiny_scene
    ctx = _ctx(sc)
    with pytest.raises(binding.AghError) as e:
        ctx.find_hands(sc.samples)
    assert e.value.code == -5  # AGH_ERR_NO_CLOUD
    ctx.set_cloud(sc.xyz, sc.cam)
    assert len(ctx.find_hands(np.zeros(0, np.int32))) == 0  # empty sample list
    with pytest.raises(binding.AghError) as e:
        ctx.find_hands(np.array([sc.n], np.int32))
    assert e.value.code == -1
    ctx.find_hands(sc.samples[:4])
    with pytest.raises(binding.AghError) as e:
        ctx.classify()
    assert e.value.code == -6  # AGH_ERR_NO_SVM
    # empty cloud, single isolated point, two far points: no hypotheses, no crash
    ctx.set_cloud(np.zeros((0, 3), np.float32), np.zeros(0, np.int32))
    assert len(ctx.find_hands(np.zeros(0, np.int32))) == 0
    two = np.array([[0.5, 0.0, 0.0], [2.0, 2.0, 2.0]], np.float32)
    ctx.set_cloud(two, np.zeros(2, np.int32))
    from oracle import oracle_py as O

    lone = ctx.find_hands(np.array([0, 1], np.int32))
    ref = O.find_hands(O.default_params(sc.cam_origins), two, np.zeros(2, np.int32), np.array([0, 1], np.int32))["hyps"]
    assert len(lone) == len(ref)
    for f in INT_FIELDS + FLOAT_FIELDS:
        assert np.array_equal(lone[f], ref[f]), f
    fr = ctx.frames()
    assert (fr["valid"] == 1).all() and list(fr["n_nb"]) == [1, 1]  # under-determined fits still yield a frame
    # duplicate samples are independent work items
    ctx.set_cloud(sc.xyz, sc.cam)
    s = np.array([sc.samples[3]] * 3, np.int32)
    h = ctx.find_hands(s)
    one = ctx.find_hands(s[:1])
    assert len(h) == 3 * len(one)


@pytest.mark.parametrize("kind", ["duplicates", "lattice_plane", "diagonal_plane", "collinear", "random_blob", "two_sheets"])
def test_degenerate_and_ragged_clouds_bit_exact(kind, tiny_scene):
    """Ties in the neighbour order (duplicate points, lattice symmetry), singular fits (exact planes, lines), tiny and
    ragged neighbourhoods: whatever the oracle does with them, the kernels must do the same, bit for bit."""
    from oracle import oracle_py as O

    rng = np.random.default_rng(sum(map(ord, kind)))
    cams = tiny_scene.cam_origins
    c0 = np.array([0.75, 0.05, -0.05])
    if kind == "duplicates":
        base = c0 + rng.uniform(-0.05, 0.05, (1500, 3)) * np.array([1, 1, 0.15])
        xyz = np.concatenate([base, base[:400], base[:100]])  # exact duplicates: d2 ties broken by index
    elif kind == "lattice_plane":
        g = np.arange(-0.06, 0.06, 0.003)
        u, v = np.meshgrid(g, g, indexing="ij")
        xyz = c0 + np.stack([u.ravel(), v.ravel(), np.zeros(u.size)], 1)  # exactly planar: singular pencil, deflated -> frames with the plane's normal
    elif kind == "diagonal_plane":
        # x + y + z = const on a 2^-8 m lattice (every coordinate and the plane equation exact in float32): a singular pencil
        # whose Cholesky fails at the LAST pivot, where the axis-aligned plane's fails at pivot 8 of a sparser matrix
        k = np.arange(-18, 19)
        i, j = np.meshgrid(k, k, indexing="ij")
        xyz = np.stack([i.ravel(), j.ravel(), -(i + j).ravel()], 1) * 2.0 ** -8 + np.array([192, 13, -12]) * 2.0 ** -8
    elif kind == "collinear":
        t = np.linspace(-0.08, 0.08, 300)
        xyz = np.concatenate([c0 + np.stack([t, 0 * t, 0 * t], 1), c0 + rng.uniform(-0.02, 0.02, (40, 3))])
    elif kind == "random_blob":
        xyz = c0 + rng.normal(0, 0.03, (4000, 3))
    else:  # two parallel sheets 4 mm apart (two camera lattices that do not agree)
        g = np.arange(-0.07, 0.07, 0.003)
        u, v = np.meshgrid(g, g, indexing="ij")
        a = np.stack([u.ravel(), v.ravel(), 0.02 * np.sin(20 * u.ravel())], 1)
        xyz = c0 + np.concatenate([a, a + np.array([0.0011, 0.0007, 0.004])])
    xyz = xyz.astype(np.float32)
    cam = (rng.random(len(xyz)) < 0.4).astype(np.int32)
    samples = np.sort(rng.choice(len(xyz), min(40, len(xyz)), replace=False)).astype(np.int32)
    from agile_grasp_amd import binding

    ctx = binding.Context(cams)
    ctx.set_cloud(xyz, cam)
    hyps = ctx.find_hands(samples)
    ref = O.find_hands(O.default_params(cams), xyz, cam, samples, want_images=True)
    fr, rf = ctx.frames(), ref["frames"]
    assert np.array_equal(fr["valid"], rf["valid"]) and np.array_equal(fr["n_nb"], rf["n_nb"])
    ok = rf["valid"] != 0
    for f in ("majority_cam", "max_index", "params", "eigenvalue", "normal", "axis", "binormal"):
        assert np.array_equal(fr[f][ok], rf[f][ok], equal_nan=True), (kind, f)
    nt, nh = ctx.neighbor_counts()
    assert np.array_equal(nh[ok], ref["nh"][ok])
    assert len(hyps) == len(ref["hyps"])
    for f in INT_FIELDS + FLOAT_FIELDS:
        assert np.array_equal(hyps[f], ref["hyps"][f]), (kind, f)
    if len(hyps):
        assert np.array_equal(ctx.images(), ref["images"])


@pytest.mark.parametrize("n_points", [6000, 12000])
@pytest.mark.parametrize("normals_mode", ["det", "rand50"])
def test_unvoxelised_blob_beyond_4096_neighbours_bit_exact(normals_mode, n_points):
    """kdtree.radiusSearch has max_nn = 0 (hand_search.cpp:85): the reference's test mains run on raw, un-voxelised captures.
    A 6000-point blob puts ~5800 points into the Taubin ball -- beyond the LDS-resident classes of K1a (4096): K1a through
    global scratch, K1c's 6144 class in LDS; a 12000-point blob ~11 600 -- beyond every LDS class: K1c on normals in global
    memory, every column summed.  Switched on by the call that needs them; frames and hypotheses must be the oracle's."""
    from agile_grasp_amd import binding
    from oracle import oracle_py as O

    rng = np.random.default_rng(0)
    xyz = (rng.normal(0, 0.01, (n_points, 3)) + np.array([0.7, 0.0, 0.0])).astype(np.float32)
    cam = (rng.random(n_points) < 0.4).astype(np.int32)
    cams = np.array([[0.0, 0.3, 0.5], [0.0, -0.3, 0.5]])
    mode = binding.NORMALS_RAND50 if normals_mode == "rand50" else binding.NORMALS_DETERMINISTIC
    ctx = binding.Context(cams, normals_mode=mode, rand_seed=3)
    ctx.set_cloud(xyz, cam)
    samples = np.array([5, 77, 1234, n_points - 1], np.int32)
    hyps = ctx.find_hands(samples)  # (the host entry point repeats by itself: larger classes, then the classes beyond LDS)
    omode = O.NORMALS_RAND50 if normals_mode == "rand50" else O.NORMALS_DETERMINISTIC
    ref = O.find_hands(O.default_params(cams, normals_mode=omode, rand_seed=3), xyz, cam, samples)
    assert ref["frames"]["n_nb"].max() > 4096 and (ref["frames"]["n_nb"].max() <= 6144) == (n_points == 6000)
    assert_frames_equal(ctx.frames(), ref["frames"])
    assert_hyps_equal(hyps, ref["hyps"])
    # ... and the context keeps working on ordinary clouds afterwards (its per-sample scratch was re-sized)
    from agile_grasp_amd import synthetic

    sc = synthetic.config("tiny")
    ctx2 = binding.Context(sc.cam_origins)
    ctx2.set_cloud(xyz, cam)
    ctx2.find_hands(samples)
    ctx2.set_cloud(sc.xyz, sc.cam)
    got = ctx2.find_hands(sc.samples)
    ref2 = O.find_hands(O.default_params(sc.cam_origins), sc.xyz, sc.cam, sc.samples)
    assert_hyps_equal(got, ref2["hyps"])


def test_capacity_overflow_is_loud():
    """The neighbourhoods beyond 4096 points of ONE launch share a pool of 2^21 points: 400 samples of a 12000-point blob
    (~11 600 neighbours each) exceed it (400 samples) -- the call must raise, never return partial results."""
    from agile_grasp_amd import binding

    rng = np.random.default_rng(0)
    xyz = (rng.normal(0, 0.01, (12000, 3)) + np.array([0.7, 0.0, 0.0])).astype(np.float32)
    ctx = binding.Context(np.zeros((2, 3)), normals_mode=binding.NORMALS_RAND50)
    ctx.set_cloud(xyz, np.zeros(12000, np.int32))
    with pytest.raises(binding.AghError) as e:
        ctx.find_hands(np.arange(400, dtype=np.int32))
    assert e.value.code == -4  # AGH_ERR_CAPACITY


def test_large_neighbourhood_classes(tiny_scene):
    """Neighbourhoods between 1024 and 2048 points go through the second LDS capacity class."""
    from oracle import oracle_py as O

    rng = np.random.default_rng(3)
    n = 30000
    # a dense curved sheet: ~1500 neighbours within 3 cm
    u, v = rng.uniform(-0.08, 0.08, n), rng.uniform(-0.08, 0.08, n)
    xyz = np.stack([0.7 + u, v, 0.1 * u * u - 0.2 * v * v + rng.normal(0, 3e-4, n)], 1).astype(np.float32)
    cam = (rng.random(n) < 0.5).astype(np.int32)
    cams = tiny_scene.cam_origins
    from agile_grasp_amd import binding

    ctx = binding.Context(cams)
    ctx.set_cloud(xyz, cam)
    d = np.linalg.norm(xyz - np.array([0.7, 0, 0], np.float32), axis=1)
    samples = np.sort(np.argsort(d)[:6]).astype(np.int32)
    hyps = ctx.find_hands(samples)
    ref = O.find_hands(O.default_params(cams), xyz, cam, samples)
    assert ref["frames"]["n_nb"].max() > 1024
    assert_frames_equal(ctx.frames(), ref["frames"])
    assert_hyps_equal(hyps, ref["hyps"])


def test_device_resident_api_matches_host_api(tiny_scene, svm_model):
    import torch

    sc = tiny_scene
    w, rho = svm_model
    host = _ctx(sc)
    host.set_cloud(sc.xyz, sc.cam)
    ref = host.find_hands(sc.samples)
    host.load_svm(w, rho)
    ref_keep = host.classify()
    dev = _ctx(sc)
    dev.load_svm(w, rho)
    xyz_t = torch.from_numpy(sc.xyz).cuda()
    cam_t = torch.from_numpy(sc.cam).cuda()
    s_t = torch.from_numpy(sc.samples).cuda()
    out_t = torch.zeros(8 * sc.samples.size * 160, dtype=torch.uint8, device="cuda")
    n_t = torch.zeros(1, dtype=torch.int64, device="cuda")
    keep_t = torch.zeros(8 * sc.samples.size, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ts = torch.cuda.Stream()  # an explicit stream, as bench.py uses (NULL would select the context's own stream)
    st = ts.cuda_stream
    from agile_grasp_amd import binding

    retried = 0
    for attempt in range(3):
        dev.set_cloud_torch(xyz_t, cam_t, stream=st)
        dev.find_hands_torch(s_t, out_t, n_t, stream=st)
        dev.classify_torch(keep_t, stream=st)
        try:
            dev.synchronize()  # also reports device-side errors of the asynchronous calls
        except binding.AghError as e:
            # the dense test scene has Taubin balls beyond the first capacity class: the first asynchronous call says so
            # once, the context switches the larger classes on, and the repeated call is complete
            assert e.code == binding.AGH_ERR_RETRY and attempt == 0
            retried += 1
    assert retried == (1 if host.neighbor_counts()[0].max() > 1152 else 0)
    n = int(n_t.item())
    bad = s_t.clone()
    bad[5] = sc.n + 7  # device-resident sample lists are validated on the device: loud error, no out-of-bounds read
    torch.cuda.synchronize()  # (bad was written on torch's stream, the search runs on ts)
    dev.find_hands_torch(bad, out_t, n_t, stream=st)
    with pytest.raises(binding.AghError) as e:
        dev.synchronize()
    assert e.value.code == -1
    dev.find_hands_torch(s_t, out_t, n_t, stream=st)
    dev.classify_torch(keep_t, stream=st)
    dev.synchronize()

    got = np.frombuffer(out_t.cpu().numpy().tobytes(), dtype=binding.HYP_DTYPE)[:n]
    assert n == len(ref)
    for f in INT_FIELDS + FLOAT_FIELDS:
        assert np.array_equal(got[f], ref[f]), f
    assert np.array_equal(keep_t.cpu().numpy()[:n], ref_keep)
    assert np.array_equal(got["svm_keep"], ref_keep)


def test_full_size_c2_against_oracle(svm_model):
    """BASELINE config C2/C3 (300k points, 2000 samples): whole result list bit-identical to the oracle."""
    from agile_grasp_amd import synthetic
    from oracle import oracle_py as O

    sc = synthetic.config("C2")
    w, rho = svm_model
    ctx = _ctx(sc)
    ctx.set_cloud(sc.xyz, sc.cam)
    hyps = ctx.find_hands(sc.samples)
    ctx.load_svm(w, rho)
    keep = ctx.classify()
    ref = O.find_hands(O.default_params(sc.cam_origins), sc.xyz, sc.cam, sc.samples, want_images=True)
    assert_frames_equal(ctx.frames(), ref["frames"])
    assert_hyps_equal(hyps, ref["hyps"])
    okeep, _ = O.classify(ref["images"], w, rho)
    assert np.array_equal(keep, okeep)
    assert len(hyps) > 300


@pytest.mark.parametrize("name", ["C1", "C4", "C5_0", "C2u"])
def test_full_size_other_configs_against_oracle(svm_model, name):
    """The BASELINE configs besides C2/C3, whole result list (frames, hypotheses, SVM labels) bit-identical to the oracle:
    C1 (single view, 50k points, 500 samples: every point from camera 0), C4 (1M points, 8000 samples: multi-tile sweeps
    with the parked-points path, the 4096-point moments class, the large scheduling sort), C5_0 (first cloud of the
    batch, seed 10), C2u (SURVEY 8d's generator taken literally: the axis-aligned C2 scene, two thirds of whose Taubin
    neighbourhoods are exactly planar -- the exhaustive argmax of quadric.cpp:283-284, deterministic normals)."""
    from agile_grasp_amd import synthetic
    from oracle import oracle_py as O

    sc = synthetic.config(name)
    w, rho = svm_model
    ctx = _ctx(sc)
    ctx.set_cloud(sc.xyz, sc.cam)
    hyps = ctx.find_hands(sc.samples)
    ctx.load_svm(w, rho)
    keep = ctx.classify()
    ref = O.find_hands(O.default_params(sc.cam_origins), sc.xyz, sc.cam, sc.samples, want_images=True)
    assert_frames_equal(ctx.frames(), ref["frames"])
    assert_hyps_equal(hyps, ref["hyps"])
    okeep, _ = O.classify(ref["images"], w, rho)
    assert np.array_equal(keep, okeep)
    assert len(hyps) > sc.samples.size // 10
    nt, nh = ctx.neighbor_counts()
    if name == "C4":
        assert nh.max() > 2176  # at least one sample streams several tiles through the sweep
    # the stateless classification path (images carried by the hypotheses) gives the same labels and decision values
    packed = ctx.packed_images()
    assert np.array_equal(binding_unpack(packed), ref["images"].reshape(len(hyps), -1))
    keep2, sums2 = ctx.classify_images(packed)
    _, osums = O.classify(ref["images"], w, rho)
    assert np.array_equal(keep2, okeep) and np.array_equal(sums2, osums)


def binding_unpack(words):
    from agile_grasp_amd import binding

    return binding.unpack_images(words)


def test_large_sample_list_matches_small_one(tiny_scene):
    """70 000 samples (the list of 64 repeated) take the large-S paths -- three-kernel concatenation, many tiles in the
    scheduling sort, thousands of eigen work-groups -- and must give, position by position, what the 64 give."""
    sc = tiny_scene
    ctx = _ctx(sc)
    ctx.set_cloud(sc.xyz, sc.cam)
    base = ctx.find_hands(sc.samples)
    reps = 70_000 // sc.samples.size + 1
    big_idx = np.tile(sc.samples, reps)[:70_000].astype(np.int32)
    big = ctx.find_hands(big_idx)
    per = {}
    for h in base:
        per.setdefault(int(h["sample"]), []).append(h)
    exp_n = sum(len(per.get(p % sc.samples.size, [])) for p in range(len(big_idx)))
    assert len(big) == exp_n and len(base) > 0
    # sample-major order, orientation ascending inside a sample
    assert (np.diff(big["sample"]) >= 0).all()
    pos = 0
    for p in range(0, len(big_idx), 997):  # spot-check every 997th position in full
        rows = big[big["sample"] == p]
        ref = per.get(p % sc.samples.size, [])
        assert len(rows) == len(ref)
        for r, q in zip(rows, ref):
            for f in FLOAT_FIELDS + ("orientation", "cam_source", "n_in_box", "finger_index", "depth_index"):
                assert np.array_equal(r[f], q[f]), (p, f)


def test_points_for_learning_on_demand(tiny_scene):
    """GraspHypothesis::getPointsForLearning + the camera index lists, recomputed per hypothesis (agh_get_learning_points)."""
    from agile_grasp_amd import binding
    from oracle import oracle_py as O

    sc = tiny_scene
    ctx = _ctx(sc)
    with pytest.raises(binding.AghError):
        ctx.learning_points(0)  # nothing searched yet
    ctx.set_cloud(sc.xyz, sc.cam)
    hyps = ctx.find_hands(sc.samples)
    ref = O.find_hands_points(O.default_params(sc.cam_origins), sc.xyz, sc.cam, sc.samples)
    assert_hyps_equal(hyps, ref["hyps"])
    for k in range(len(hyps)):
        pts, cam = ctx.learning_points(k)
        assert pts.shape == (3, hyps["n_in_box"][k])
        assert np.array_equal(pts, ref["points"][k]) and np.array_equal(cam, ref["cams"][k])
    with pytest.raises(binding.AghError):
        ctx.learning_points(len(hyps))
    # the image Learning::convertToImage draws from these points is the one the sweep rasterised
    from tests import ref_numpy as R

    images = ctx.images()
    for k in (0, len(hyps) // 2, len(hyps) - 1):
        pts, _ = ctx.learning_points(k)
        s2c = hyps["surface"][k] - sc.cam_origins[hyps["cam_source"][k]]
        assert np.array_equal(R.convert_to_image(pts, hyps["binormal"][k], s2c).reshape(-1), images[k])


@pytest.mark.parametrize("name,expect_retry", [("small", True), ("C2", False)])
def test_larger_capacity_classes_are_switched_on_by_the_first_cloud_that_needs_them(name, expect_retry):
    """The launches of the capacity classes beyond the first are skipped until a cloud needs them (they are empty for
    voxelised clouds, ~5 us each).  Asynchronous calls report the switch once as AGH_ERR_RETRY at agh_synchronize; the repeated
    call is complete and equal to what the host entry point (which repeats by itself) returns."""
    import torch

    from agile_grasp_amd import binding, synthetic

    sc = synthetic.config(name)
    host = binding.Context(sc.cam_origins)
    host.set_cloud(sc.xyz, sc.cam)
    ref = host.find_hands(sc.samples)
    assert (host.neighbor_counts()[0].max() > 1152) == expect_retry  # the scenes are what the test takes them for
    dev = binding.Context(sc.cam_origins)
    xyz_t, cam_t, s_t = (torch.from_numpy(a).cuda() for a in (sc.xyz, sc.cam, sc.samples))
    out_t = torch.zeros(8 * sc.samples.size * 160, dtype=torch.uint8, device="cuda")
    n_t = torch.zeros(1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    codes = []
    for _ in range(3):
        dev.set_cloud_torch(xyz_t, cam_t)
        dev.find_hands_torch(s_t, out_t, n_t)
        try:
            dev.synchronize()
            codes.append(0)
        except binding.AghError as e:
            codes.append(e.code)
    assert codes == ([binding.AGH_ERR_RETRY, 0, 0] if expect_retry else [0, 0, 0])
    got = np.frombuffer(out_t.cpu().numpy().tobytes(), dtype=binding.HYP_DTYPE)[: int(n_t.item())]
    assert len(got) == len(ref) > 0
    assert _unstamped(got) == _unstamped(ref)


def _with_non_finite_points(sc, seed=7):
    """The scene with 1 % of its points made non-finite (a NaN in one coordinate, all NaN, +-Inf) and twenty of them added to the
    sample list: what HandSearch::findHands sees when it is called on a raw capture (hands_test.cpp:21-40)."""
    rng = np.random.default_rng(seed)
    xyz = sc.xyz.copy()
    bad = rng.permutation(sc.n)[:max(40, sc.n // 100)]
    q = bad.size // 4
    xyz[bad[:q], rng.integers(0, 3, q)] = np.nan
    xyz[bad[q:2 * q]] = np.inf
    xyz[bad[2 * q:3 * q], 1] = -np.inf
    xyz[bad[3 * q:]] = np.nan
    samples = np.unique(np.concatenate([sc.samples, bad[:20]])).astype(np.int32)
    return xyz, bad, samples


@pytest.mark.parametrize("scene_name", ["tiny", "small"])
def test_non_finite_points_bit_exact(scene_name):
    """agh_set_cloud with NaN / Inf points.  Defined as PCL's kd-tree defines it (setInputCloud drops them:
    they are nobody's neighbour; indices do not move); a sample AT such a point has no frame and no hypotheses.  GPU == oracle."""
    from agile_grasp_amd import synthetic
    from oracle import oracle_py as O

    sc = synthetic.config(scene_name)
    xyz, bad, samples = _with_non_finite_points(sc)
    ctx = _ctx(sc)
    ctx.set_cloud(xyz, sc.cam)
    hyps = ctx.find_hands(samples)
    ref = O.find_hands(O.default_params(sc.cam_origins), xyz, sc.cam, samples, want_images=True)
    fr = ctx.frames()
    for f in ("valid", "n_nb", "majority_cam", "max_index", "params", "eigenvalue", "normal", "axis", "binormal"):
        assert np.array_equal(fr[f], ref["frames"][f]), f
    at_bad = np.isin(samples, bad)
    assert at_bad.sum() >= 20 and not fr["valid"][at_bad].any() and fr["valid"][~at_bad].all()
    nt, nh = ctx.neighbor_counts()
    assert np.array_equal(nt, ref["frames"]["n_nb"]) and np.array_equal(nh, ref["nh"])
    assert len(hyps) > 5
    assert_hyps_equal(hyps, ref["hyps"])
    assert np.array_equal(ctx.images(), ref["images"])
    # ... and the device-resident entry points, and a cloud without one finite point
    import torch

    ctx.set_cloud_torch(torch.from_numpy(xyz).cuda(), torch.from_numpy(sc.cam).cuda())
    assert_hyps_equal(ctx.find_hands(samples), ref["hyps"])
    nothing = np.full((64, 3), np.nan, np.float32)
    ctx.set_cloud(nothing, np.zeros(64, np.int32))
    assert len(ctx.find_hands(np.arange(8, dtype=np.int32))) == 0


def test_big_class_list_walk_with_several_samples_per_work_group():
    """The 4096 class of the Taubin stage runs as 512 work-groups that walk the list of the samples the 1152 class handed on
    (round 6).  3000 samples of the over-dense `small` scene put ~1500 samples on that list: every work-group takes three of them
    in turn (the tiles are reused from one sample to the next).  Frames and hypotheses against the oracle, bit for bit."""
    from agile_grasp_amd import synthetic
    from oracle import oracle_py as O

    sc = synthetic.config("small")
    samples = np.sort(np.random.default_rng(3).permutation(sc.n)[:3000]).astype(np.int32)
    ctx = _ctx(sc)
    ctx.set_cloud(sc.xyz, sc.cam)
    hyps = ctx.find_hands(samples)
    ref = O.find_hands(O.default_params(sc.cam_origins), sc.xyz, sc.cam, samples)
    n_nb = ref["frames"]["n_nb"]
    assert (n_nb > 1152).sum() > 1024 and n_nb.max() <= 4096
    assert_frames_equal(ctx.frames(), ref["frames"])
    assert_hyps_equal(hyps, ref["hyps"])
    # ... and through the all-points pass (batches of 16 384 points at r = 0.01: the same kernels' other classes)
    hyps = ctx.find_hands(samples[:400], calculates_antipodal=True)
    ref = O.find_hands(O.default_params(sc.cam_origins), sc.xyz, sc.cam, samples[:400], calculates_antipodal=True)
    assert_hyps_equal(hyps, ref["hyps"])
