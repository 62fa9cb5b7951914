"""The host-buffer entry points after round 4's changes to their synchronisation: agh_set_cloud / agh_preprocess return with the
grid build still queued, the camera ids travel on a second stream, results are written into pinned host memory by the kernels and
a call waits for one synchronisation.  These tests interleave calls in the orders a caller may use -- clouds of changing size through
ONE context, host and device entry points mixed, another stream for the search, buffers that grow -- and compare every result
with what a fresh context returns for the same input (whose parity with the oracle the other GPU tests establish)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("sample", "orientation", "cam_source", "n_in_box", "half_antipodal", "full_antipodal", "finger_index", "depth_index",
          "axis", "approach", "binormal", "bottom", "surface", "width", "valid")


def _same(a, b):
    assert len(a) == len(b)
    for f in FIELDS:
        assert np.array_equal(a[f], b[f]), f


def _reference(sc, samples, svm):
    from agile_grasp_amd import binding

    ctx = binding.Context(sc.cam_origins)
    ctx.set_cloud(sc.xyz, sc.cam)
    h = ctx.find_hands(samples)
    ctx.load_svm(*svm)
    k = ctx.classify()
    hd, idx = ctx.find_handles(h[k.astype(bool)], 3, 0.005)
    ctx.close()
    return h, k, hd, idx


def test_one_context_many_clouds_in_any_order(tiny_scene, small_scene, svm_model):
    from agile_grasp_amd import binding, synthetic

    scenes = [tiny_scene, small_scene, synthetic.config("C1"),
              synthetic.make_scene(30_000, 300, seed=91, two_view=True, n_objects=5, name="hp_a"),
              synthetic.make_scene(120_000, 700, seed=92, two_view=True, n_objects=9, name="hp_b")]
    rng = np.random.default_rng(7)
    subsets = [sc.samples[np.sort(rng.permutation(sc.samples.size)[:max(8, sc.samples.size // 2)])] for sc in scenes]
    refs = [(_reference(sc, sc.samples, svm_model), _reference(sc, sub, svm_model)) for sc, sub in zip(scenes, subsets)]
    ctx = binding.Context(scenes[0].cam_origins)   # (all scenes of a kind share the camera origins)
    assert all(np.array_equal(sc.cam_origins, scenes[0].cam_origins) for sc in scenes)
    ctx.load_svm(*svm_model)
    for it in range(60):
        k = int(rng.integers(len(scenes)))
        sc = scenes[k]
        which = int(rng.integers(2))
        samples = sc.samples if which == 0 else subsets[k]
        ref_h, ref_k, ref_hd, ref_idx = refs[k][which]
        # the caller's buffers may be reused the moment agh_set_cloud returns: hand it copies and scribble over them
        xyz, cam = sc.xyz.copy(), sc.cam.copy()
        ctx.set_cloud(xyz, cam)
        xyz[:] = np.nan
        cam[:] = 7
        if it % 3 == 1:
            ctx.set_cloud(sc.xyz, sc.cam)  # twice in a row: the first build is overtaken by the second
        s = samples.copy()
        h = ctx.find_hands(s)
        s[:] = -1
        _same(h, ref_h)
        if it % 2 == 0:
            keep = ctx.classify()
            assert np.array_equal(keep, ref_k)
            hd, idx = ctx.find_handles(h[keep.astype(bool)], 3, 0.005)
            assert np.array_equal(idx, ref_idx) and len(hd) == len(ref_hd)
            for f in hd.dtype.names:
                assert np.array_equal(hd[f], ref_hd[f]), f
        if it % 5 == 4:  # the lazy getters still describe the last search
            assert int((ctx.frames()["valid"] != 0).sum()) > 0
    ctx.close()


def test_host_set_cloud_then_device_search_on_another_stream(small_scene):
    """agh_set_cloud leaves the grid build on the context's stream; a search issued on a caller's NON-BLOCKING stream must wait
    for it (order_after_cloud)."""
    import torch

    from agile_grasp_amd import binding

    sc = small_scene
    ref = _reference(sc, sc.samples, (np.zeros(3528, np.float32), 0.0))[0]
    dev = torch.device("cuda", 0)
    ctx = binding.Context(sc.cam_origins)
    s_t = torch.from_numpy(sc.samples).to(dev)
    S = sc.samples.size
    out_t = torch.zeros(8 * S * 160, dtype=torch.uint8, device=dev)
    nout_t = torch.zeros(1, dtype=torch.int64, device=dev)
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")
    st = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0  # hipStreamNonBlocking
    torch.cuda.synchronize()
    for it in range(20):
        ctx.set_cloud(sc.xyz, sc.cam)
        ctx.find_hands_torch(s_t, out_t, nout_t, stream=st.value)
        assert hip.hipStreamSynchronize(st) == 0
        try:
            ctx.synchronize()
        except binding.AghError as e:  # the over-dense `small` scene needs the larger capacity classes: reported once, then on
            assert e.code == binding.AGH_ERR_RETRY and it == 0
            continue
        n = int(nout_t.item())
        got = np.frombuffer(out_t.cpu().numpy().tobytes(), dtype=binding.HYP_DTYPE)[:n]
        _same(got, ref)
    hip.hipStreamDestroy(st)
    ctx.close()


def test_preprocess_then_everything_twice_with_growing_lattice(svm_model):
    """The voxeliser keeps the bitmap of the previous cloud and launches stage 2 for it at once; a larger lattice must be
    noticed (error 2), the bitmap enlarged and the pass repeated -- same voxels either way."""
    from agile_grasp_amd import binding, synthetic

    # (the generator's scene gets denser with the point count: 60k and 700k raw points stay below the 4096-point capacity of a
    # Taubin ball, sizes in between do not)
    raws = [synthetic.make_raw_cloud(60_000, 31), synthetic.make_raw_cloud(700_000, 21), synthetic.make_raw_cloud(60_000, 31)]
    ctx = binding.Context(raws[0].cam_origins)
    ctx.load_svm(*svm_model)
    results = []
    for rc in raws:
        ws = np.asarray(rc.workspace, np.float64).copy()
        if rc is raws[1]:
            ws[0::2] -= 0.2  # a larger workspace: a lattice the kept bitmap cannot hold
            ws[1::2] += 0.2
        nv = ctx.preprocess(rc.xyz, rc.size_left, ws)
        fresh = binding.Context(rc.cam_origins)
        assert fresh.preprocess(rc.xyz, rc.size_left, ws) == nv
        a, b = ctx.cloud(), fresh.cloud()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        samples = np.sort(np.random.default_rng(3).permutation(nv)[:200]).astype(np.int32)
        _same(ctx.find_hands(samples), fresh.find_hands(samples))
        results.append(nv)
        fresh.close()
    assert results[0] == results[2]
    ctx.close()


def test_set_cloud_from_page_locked_memory_may_be_overwritten_at_once(tiny_scene, small_scene):
    """with a page-locked source (hipHostMalloc) the upload copies are truly asynchronous -- agh_set_cloud still
    promises that xyz and cam_source may be reused as soon as it returns, so it must wait for the COPIES.  Two clouds alternate
    through one pair of page-locked buffers that are scribbled over right after every call; the camera-id copy (a stream of its
    own) must also not overtake the previous cloud's search."""
    import torch

    from agile_grasp_amd import binding

    scenes = [small_scene, tiny_scene]
    refs = [_reference(sc, sc.samples, (np.zeros(3528, np.float32), 0.0))[0] for sc in scenes]
    nmax = max(sc.xyz.shape[0] for sc in scenes)
    xyz_pin = torch.empty((nmax, 3), dtype=torch.float32).pin_memory()
    cam_pin = torch.empty((nmax,), dtype=torch.int32).pin_memory()
    assert xyz_pin.is_pinned() and cam_pin.is_pinned()
    xyz_np, cam_np = xyz_pin.numpy(), cam_pin.numpy()
    ctx = binding.Context(scenes[0].cam_origins)
    for it in range(40):
        k = it & 1
        sc = scenes[k]
        n = sc.xyz.shape[0]
        xyz_np[:n] = sc.xyz
        cam_np[:n] = sc.cam
        ctx.set_cloud(xyz_np[:n], cam_np[:n])
        xyz_np[:] = np.float32(1e3) + np.float32(it)  # the frame buffer is reused at once
        cam_np[:] = 1 - (it & 1)
        try:
            h = ctx.find_hands(sc.samples)
        except binding.AghError as e:  # (the over-dense `small` scene: the larger capacity classes, reported once)
            assert e.code == binding.AGH_ERR_RETRY and it == 0
            h = ctx.find_hands(sc.samples)
        _same(h, refs[k])
    ctx.close()


def test_second_search_on_another_stream_still_waits_for_the_build(small_scene):
    """the first search after a host-buffer agh_set_cloud runs on the context's own stream (in order behind the
    build) -- a SECOND search, on a caller's non-blocking stream, must still be ordered behind that build."""
    import ctypes as C

    import torch

    from agile_grasp_amd import binding

    sc = small_scene
    ref = _reference(sc, sc.samples, (np.zeros(3528, np.float32), 0.0))[0]
    dev = torch.device("cuda", 0)
    ctx = binding.Context(sc.cam_origins)
    s_t = torch.from_numpy(sc.samples).to(dev)
    S = sc.samples.size
    out_a = torch.zeros(8 * S * 160, dtype=torch.uint8, device=dev)
    out_b = torch.zeros(8 * S * 160, dtype=torch.uint8, device=dev)
    n_a = torch.zeros(1, dtype=torch.int64, device=dev)
    n_b = torch.zeros(1, dtype=torch.int64, device=dev)
    hip = C.CDLL("libamdhip64.so")
    st = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0  # hipStreamNonBlocking
    torch.cuda.synchronize()
    for it in range(12):
        ctx.set_cloud(sc.xyz, sc.cam)
        ctx.find_hands_torch(s_t, out_a, n_a, stream=None)       # the context's stream: no wait needed, none taken
        ctx.find_hands_torch(s_t, out_b, n_b, stream=st.value)   # another stream, right behind
        assert hip.hipStreamSynchronize(st) == 0
        try:
            ctx.synchronize()
        except binding.AghError as e:
            assert e.code == binding.AGH_ERR_RETRY and it == 0
            continue
        got = np.frombuffer(out_b.cpu().numpy().tobytes(), dtype=binding.HYP_DTYPE)[:int(n_b.item())]
        _same(got, ref)
    hip.hipStreamDestroy(st)
    ctx.close()
