"""NaN removal + workspace box + per-camera voxelisation (localization.cpp:17-45,216-355).

CPU part: the C++ oracle against an independent numpy transcription.  GPU part: the HIP bitmap voxeliser (through the
C ABI) against the oracle -- same points, same order, same float32 bits, same camera ids -- and the chain
preprocess -> find_hands against set_cloud(oracle voxels) -> find_hands.
"""
import numpy as np
import pytest

from tests import ref_numpy

BIG = [-1e3, 1e3, -1e3, 1e3, -1e3, 1e3]


def _cases():
    from agile_grasp_amd import synthetic

    rng = np.random.default_rng(11)
    out = {}
    rc = synthetic.make_raw_cloud(60_000, 7)
    out["scene_nan_ws"] = (rc.xyz, rc.size_left, rc.workspace, False)
    out["scene_dense_flag"] = (rc.xyz, rc.size_left, rc.workspace, True)  # is_dense clouds skip the NaN compaction
    out["scene_no_ws"] = (rc.xyz, rc.size_left, BIG, False)
    out["left_only"] = (rc.xyz[:rc.size_left], rc.size_left, rc.workspace, False)
    out["right_only"] = (rc.xyz, 0, rc.workspace, False)
    out["size_left_beyond"] = (rc.xyz[:5000], 10_000, BIG, False)
    out["empty"] = (np.zeros((0, 3), np.float32), 0, BIG, False)
    out["all_nan"] = (np.full((300, 3), np.nan, np.float32), 100, BIG, False)
    out["all_outside"] = (rng.uniform(2, 3, (500, 3)).astype(np.float32), 250, [0, 1, 0, 1, 0, 1], False)
    out["single_point"] = (np.array([[0.1, -0.2, 0.3]], np.float32), 1, BIG, False)
    # coordinates exactly on lattice planes, on the workspace faces, negative, +-0 and +-inf
    g = (np.arange(-40, 40)[:, None] * 0.003).astype(np.float32)
    lat = np.concatenate([np.concatenate([g, g * 0, g * 0 + 0.5], 1), np.concatenate([g * 0 - 0.0, g, -g], 1)])
    lat = np.concatenate([lat, [[np.inf, 0, 0], [0, -np.inf, 0], [-0.12, 0.0, 0.5], [0.117, -0.0, 0.5]]]).astype(np.float32)
    out["lattice_planes"] = (lat, 80, [-0.12, 0.117, -0.12, 0.117, -0.2, 0.5], False)
    dup = np.repeat(rng.uniform(-0.05, 0.05, (40, 3)).astype(np.float32), 50, 0)
    out["duplicates"] = (dup[rng.permutation(len(dup))], 1000, BIG, False)
    far = rng.uniform(10_000.2, 10_001.0, (400, 3)).astype(np.float32)  # beyond the initial minimum 10000: it stays
    out["beyond_10000"] = (far, 200, [0, 2e4, 0, 2e4, 0, 2e4], False)
    sparse = rng.uniform(-2, 2, (20_000, 3)).astype(np.float32)     # 1334^3 lattice per camera, almost empty bitmap
    out["sparse_wide"] = (sparse, 9000, BIG, False)
    return out


CASES = _cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_numpy_transcription(name):
    from oracle import oracle_py as orc

    xyz, size_left, ws, dense = CASES[name]
    v, cam = orc.preprocess(xyz, size_left, ws, 0.003, dense)
    v2, cam2 = ref_numpy.preprocess(xyz, size_left, ws, 0.003, dense)
    assert np.array_equal(v.view(np.uint32), v2.view(np.uint32))
    assert np.array_equal(cam, cam2)
    if len(v):  # lexicographic order inside each camera block, camera 0 first, no duplicates
        assert (np.diff(cam) >= 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_voxeliser_bit_exact(name):
    from agile_grasp_amd import binding, synthetic
    from oracle import oracle_py as orc

    xyz, size_left, ws, dense = CASES[name]
    ctx = binding.Context(synthetic.camera_origins())
    for stride_pad in (0, 5):  # packed xyz and pcl::PointXYZRGBA-like 32-byte points
        x = xyz if stride_pad == 0 else np.concatenate([xyz, np.full((len(xyz), stride_pad), 7.0, np.float32)], 1)
        nv = ctx.preprocess(x, size_left, ws, 0.003, dense)
        v, cam = orc.preprocess(xyz, size_left, ws, 0.003, dense)
        assert nv == len(v)
        gv, gcam = ctx.cloud()
        assert np.array_equal(gv.view(np.uint32), v.view(np.uint32))
        assert np.array_equal(gcam, cam)


@pytest.mark.gpu
def test_gpu_voxeliser_other_cell_sizes_and_reuse():
    from agile_grasp_amd import binding, synthetic
    from oracle import oracle_py as orc

    rc = synthetic.make_raw_cloud(40_000, 9)
    ctx = binding.Context(rc.cam_origins)
    for cell in (0.003, 0.01, 0.0007, 0.05):   # the context's buffers grow and are reused across calls
        nv = ctx.preprocess(rc.xyz, rc.size_left, rc.workspace, cell)
        v, cam = orc.preprocess(rc.xyz, rc.size_left, rc.workspace, cell)
        gv, gcam = ctx.cloud()
        assert nv == len(v) and np.array_equal(gv.view(np.uint32), v.view(np.uint32)) and np.array_equal(gcam, cam)


@pytest.mark.gpu
def test_lattice_too_large_is_loud():
    from agile_grasp_amd import binding, synthetic

    ctx = binding.Context(synthetic.camera_origins())
    pts = np.array([[0, 0, 0], [900, 900, 900]], np.float32)  # 300000^3 cells
    with pytest.raises(binding.AghError) as e:
        ctx.preprocess(pts, 2, BIG, 0.003)
    assert e.value.code == -4 and "lattice" in str(e.value)


@pytest.mark.gpu
def test_preprocess_then_search_equals_oracle_chain():
    from agile_grasp_amd import binding, synthetic
    from oracle import oracle_py as orc

    rc = synthetic.make_raw_cloud(60_000, 7)
    v, cam = orc.preprocess(rc.xyz, rc.size_left, rc.workspace)
    samples = np.sort(np.random.default_rng(3).permutation(len(v))[:96]).astype(np.int32)
    p = orc.default_params(rc.cam_origins)
    ref = orc.find_hands(p, v, cam, samples)
    ctx = binding.Context(rc.cam_origins)
    ctx.preprocess(rc.xyz, rc.size_left, rc.workspace)
    got = ctx.find_hands(samples)
    rh = ref["hyps"]
    assert len(got) == len(rh) and len(got) > 0
    for f in ("sample", "orientation", "cam_source", "n_in_box", "axis", "approach", "binormal", "bottom", "surface",
              "width"):
        assert np.array_equal(got[f], rh[f]), f


HYP_FIELDS = ("sample", "orientation", "cam_source", "n_in_box", "half_antipodal", "full_antipodal", "finger_index",
              "depth_index", "svm_keep", "axis", "approach", "binormal", "bottom", "surface", "width")
HANDLE_FIELDS = ("axis", "center", "approach", "binormal", "hands_center", "width", "n_inliers", "first_inlier")


def _four_calls(ctx, rc, samples, classify, min_inliers=2):
    ctx.preprocess(rc.xyz, rc.size_left, rc.workspace)
    h = ctx.find_hands(samples)
    if classify:
        k = ctx.classify().astype(bool)
        h = h.copy()
        h["svm_keep"] = k
        h = h[k]
    hd, idx = ctx.find_handles(h, min_inliers, 0.005)
    return h, hd, idx


@pytest.mark.gpu
@pytest.mark.parametrize("classify", [True, False])
def test_localize_one_call_equals_the_four_call_chain_and_the_oracle(svm_model, classify):
    """agh_localize (grasp_localizer.cpp:95-103 as one call, one synchronisation, counts kept on the device) against the four
    separate entry points AND against the oracle's chain preprocess -> find_hands -> classify -> find_handles, through the
    handles, every field exact.  Clouds of different sizes alternate through ONE context (the speculative voxel bitmap, the
    bound-sized launches and the device-side cloud count all change from call to call); explicit and device-drawn samples."""
    from agile_grasp_amd import binding, synthetic
    from oracle import oracle_py as orc

    w, rho = svm_model
    # (few objects: the small tables of these sizes, crowded with the default fourteen, pile surfaces into neighbourhoods
    # beyond the Taubin kernels' capacity)
    raws = [synthetic.make_raw_cloud(120_000, 7, n_objects=4), synthetic.make_raw_cloud(60_000, 8, n_objects=2),
            synthetic.make_raw_cloud(200_000, 9, n_objects=6)]
    one = binding.Context(raws[0].cam_origins)
    chain = binding.Context(raws[0].cam_origins)
    for c in (one, chain):
        c.load_svm(w, rho)
    n_handles = 0
    for it in range(7):
        rc = raws[it % 3]
        nv_ref = chain.preprocess(rc.xyz, rc.size_left, rc.workspace)
        if it % 2 == 0:
            samples = np.sort(np.random.default_rng(it).permutation(nv_ref)[:600]).astype(np.int32)
            got = one.localize(rc.xyz, rc.size_left, rc.workspace, samples=samples, classify=classify, min_inliers=2)
        else:
            got = one.localize(rc.xyz, rc.size_left, rc.workspace, n_samples=600, sample_seed=40 + it, classify=classify,
                               min_inliers=2)
            samples = got["samples"]
            assert np.array_equal(samples, binding.draw_samples(nv_ref, 600, 40 + it))
            assert np.all(np.diff(samples) > 0) and samples[0] >= 0 and samples[-1] < nv_ref
        assert got["n_voxels"] == nv_ref
        h, hd, idx = _four_calls(chain, rc, samples, classify)
        assert got["n_hypotheses"] == chain.last_n and len(got["hands"]) == len(h) and len(h) > 0
        for f in HYP_FIELDS:
            assert np.array_equal(got["hands"][f], h[f]), f
        assert len(got["handles"]) == len(hd) and np.array_equal(got["inlier_idx"], idx)
        n_handles += len(hd)
        for f in HANDLE_FIELDS:
            assert np.array_equal(got["handles"][f], hd[f]), f
        if it == 0:  # ... and the chain itself against the oracle
            v, cam = orc.preprocess(rc.xyz, rc.size_left, rc.workspace)
            r = orc.find_hands(orc.default_params(rc.cam_origins), v, cam, samples, want_images=True)
            oh = r["hyps"]
            if classify:
                keep, _ = orc.classify(r["images"], w, rho)
                oh = oh[np.asarray(keep, bool)]
            ohd, oidx = orc.find_handles(oh, 2, 0.005)
            for f in ("sample", "orientation", "axis", "approach", "binormal", "bottom", "surface", "width"):
                assert np.array_equal(got["hands"][f], oh[f]), f
            assert np.array_equal(got["inlier_idx"], oidx)
            for f in HANDLE_FIELDS:
                assert np.array_equal(got["handles"][f], ohd[f]), f
        if it in (2, 5):  # the raw capture already on the device (agh_localize_device), packed and with pcl's 32-byte stride
            import torch

            raw = rc.xyz if it == 2 else np.concatenate([rc.xyz, np.zeros((len(rc.xyz), 5), np.float32)], axis=1)
            dev = one.localize(torch.from_numpy(np.ascontiguousarray(raw)).cuda(), rc.size_left, rc.workspace, samples=samples,
                               classify=classify, min_inliers=2)
            assert dev["n_voxels"] == nv_ref and np.array_equal(dev["inlier_idx"], idx)
            for f in HYP_FIELDS:
                assert np.array_equal(dev["hands"][f], h[f]), f
            for f in HANDLE_FIELDS:
                assert np.array_equal(dev["handles"][f], hd[f]), f
        # the context is left as after the separate calls
        xyz_v, cam_v = one.cloud()
        xyz_c, cam_c = chain.cloud()
        assert np.array_equal(xyz_v, xyz_c) and np.array_equal(cam_v, cam_c)
    assert n_handles > 0


@pytest.mark.gpu
def test_localize_edge_cases(svm_model):
    from agile_grasp_amd import binding, synthetic

    rc = synthetic.make_raw_cloud(60_000, 7)
    ctx = binding.Context(rc.cam_origins)
    with pytest.raises(binding.AghError) as e:  # classification asked for, no model
        ctx.localize(rc.xyz, rc.size_left, rc.workspace, n_samples=10)
    assert e.value.code == binding.AGH_ERR_NO_SVM
    ctx.load_svm(*svm_model)
    nv = ctx.localize(rc.xyz, rc.size_left, rc.workspace, n_samples=0)["n_voxels"]  # no samples: an empty result, the cloud is set
    assert nv > 0 and ctx.cloud()[0].shape[0] == nv
    for bad in (nv, -1):  # a sample outside the voxelised cloud is found on the device (the host does not know the count yet)
        with pytest.raises(binding.AghError) as e:
            ctx.localize(rc.xyz, rc.size_left, rc.workspace, samples=np.array([3, bad], np.int32))
        assert e.value.code == binding.AGH_ERR_INVALID_ARGUMENT
    got = ctx.localize(rc.xyz, rc.size_left, rc.workspace, n_samples=64, sample_seed=5)  # ... and the context still works
    assert got["n_hypotheses"] > 0
    # fewer points than samples: every point is a sample, the rest of the list is skipped without an error
    tiny = rc.xyz[:40]
    got = ctx.localize(tiny, min(rc.size_left, 40), rc.workspace, n_samples=64, classify=False)
    n = got["n_voxels"]
    assert 0 < n <= 40 and np.array_equal(got["samples"][:n], np.arange(n)) and np.all(got["samples"][n:] == -(1 << 31))
    # an empty capture
    got = ctx.localize(np.zeros((0, 3), np.float32), 0, rc.workspace, n_samples=16)
    assert got["n_voxels"] == 0 and got["n_hypotheses"] == 0 and len(got["handles"]) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("pinned", [False, True])
def test_localize_begin_stage_end_equals_the_one_call(svm_model, pinned):
    """the chain as agh_localize_begin / agh_localize_end with the NEXT capture staged in between
    (agh_localize_stage: a second raw buffer, a second stream, under the kernels of the chain in flight).  A stream of captures of
    different sizes through the split calls equals the same stream through agh_localize, every field; a staged capture that the
    next begin does not name is dropped; the state errors are loud.  Once with pageable captures, once with page-locked ones."""
    from agile_grasp_amd import binding, synthetic

    w, rho = svm_model
    raws = [synthetic.make_raw_cloud(120_000, 7, n_objects=4), synthetic.make_raw_cloud(60_000, 8, n_objects=2),
            synthetic.make_raw_cloud(200_000, 9, n_objects=6)]
    one, two = binding.Context(raws[0].cam_origins), binding.Context(raws[0].cam_origins)
    for c in (one, two):
        c.load_svm(w, rho)
    with pytest.raises(binding.AghError) as e:
        two.localize_end()
    assert e.value.code == binding.AGH_ERR_STATE
    order = [0, 1, 2, 1, 0, 2, 2]
    clouds = [np.ascontiguousarray(raws[k].xyz) for k in order]  # (one array object per capture: staged captures are named by it)
    if pinned:
        # page-locked captures: agh_localize_stage's copy is then truly asynchronous -- it is still running when the call returns,
        # and only the event the adopting begin makes the chain wait for orders it in front of the voxeliser
        import torch

        keep_alive = [torch.from_numpy(c.copy()).pin_memory() for c in clouds]
        clouds = [t.numpy() for t in keep_alive]
    kw = [dict(n_samples=500, sample_seed=11 + i, classify=True, min_inliers=2) for i in range(len(order))]
    got = []
    two.localize_begin(clouds[0], raws[order[0]].size_left, raws[order[0]].workspace, **kw[0])
    with pytest.raises(binding.AghError) as e:  # one chain in flight
        two.localize_begin(clouds[0], raws[order[0]].size_left, raws[order[0]].workspace, **kw[0])
    assert e.value.code == binding.AGH_ERR_STATE
    for i in range(len(order)):
        if i + 1 < len(order):
            # every second time the staged capture is NOT the one the next begin names: it must be dropped, not searched
            two.localize_stage(clouds[i + 1] if i % 2 == 0 else clouds[(i + 2) % len(order)])
        got.append(two.localize_end())
        if i + 1 < len(order):
            rc = raws[order[i + 1]]
            two.localize_begin(clouds[i + 1], rc.size_left, rc.workspace, **kw[i + 1])
    n_handles = 0
    for i, k in enumerate(order):
        ref = one.localize(raws[k].xyz, raws[k].size_left, raws[k].workspace, **kw[i])
        g = got[i]
        assert g["n_voxels"] == ref["n_voxels"] and g["n_hypotheses"] == ref["n_hypotheses"] > 0
        assert np.array_equal(g["samples"], ref["samples"]) and np.array_equal(g["inlier_idx"], ref["inlier_idx"])
        for f in HYP_FIELDS:
            assert np.array_equal(g["hands"][f], ref["hands"][f]), f
        for f in HANDLE_FIELDS:
            assert np.array_equal(g["handles"][f], ref["handles"][f]), f
        n_handles += len(ref["handles"])
    assert n_handles > 0
    # the context is an ordinary one afterwards
    assert two.cloud()[0].shape[0] == got[-1]["n_voxels"]


@pytest.mark.gpu
def test_two_contexts_taking_turns_equal_the_one_call(svm_model):
    """Two chains in flight: capture k + 1 begins on the other context before capture k is collected (the two chains' kernels run
    side by side on the GPU; the contexts share nothing).  Every capture's result equals agh_localize's."""
    from agile_grasp_amd import binding, synthetic

    w, rho = svm_model
    raws = [synthetic.make_raw_cloud(120_000, 7, n_objects=4), synthetic.make_raw_cloud(60_000, 8, n_objects=2),
            synthetic.make_raw_cloud(200_000, 9, n_objects=6)]
    one = binding.Context(raws[0].cam_origins)
    lanes = [binding.Context(raws[0].cam_origins), binding.Context(raws[0].cam_origins)]
    for c in [one] + lanes:
        c.load_svm(w, rho)
    order = [0, 1, 2, 2, 1, 0, 1]
    clouds = [np.ascontiguousarray(raws[k].xyz) for k in order]
    kw = [dict(n_samples=400, sample_seed=21 + i, classify=True, min_inliers=2) for i in range(len(order))]
    got = []
    lanes[0].localize_begin(clouds[0], raws[order[0]].size_left, raws[order[0]].workspace, **kw[0])
    for i in range(len(order)):
        if i + 1 < len(order):
            rc = raws[order[i + 1]]
            lanes[(i + 1) & 1].localize_begin(clouds[i + 1], rc.size_left, rc.workspace, **kw[i + 1])
        got.append(lanes[i & 1].localize_end())
    for i, k in enumerate(order):
        ref = one.localize(raws[k].xyz, raws[k].size_left, raws[k].workspace, **kw[i])
        g = got[i]
        assert g["n_voxels"] == ref["n_voxels"] and g["n_hypotheses"] == ref["n_hypotheses"] > 0
        assert np.array_equal(g["samples"], ref["samples"]) and np.array_equal(g["inlier_idx"], ref["inlier_idx"])
        for f in HYP_FIELDS:
            assert np.array_equal(g["hands"][f], ref["hands"][f]), f
        for f in HANDLE_FIELDS:
            assert np.array_equal(g["handles"][f], ref["handles"][f]), f
