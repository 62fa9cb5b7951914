"""A batch of clouds in one context (BASELINE config C5): one launch set searches samples of every cloud, each radius
search confined to its own cloud's grid.  The batched result must be, cloud by cloud, what the single-cloud calls give --
the clouds of the test overlap in space, so any leak between grids would show."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("orientation", "cam_source", "n_in_box", "half_antipodal", "full_antipodal", "finger_index", "depth_index", "axis",
          "approach", "binormal", "bottom", "surface", "width", "valid")


def _scenes():
    from agile_grasp_amd import synthetic

    a = synthetic.config("tiny")
    b = synthetic.config("small")
    c = synthetic.make_scene(12_000, 40, seed=33, two_view=True, n_objects=3, name="tiny33")
    return [a, b, c]


def _single(sc, svm, **kw):
    from agile_grasp_amd import binding

    ctx = binding.Context(sc.cam_origins, **kw)
    ctx.set_cloud(sc.xyz, sc.cam)
    return ctx


def _compare(batch_hyps, per_cloud, sample_counts):
    pos = 0
    base = 0
    for hyps, S in zip(per_cloud, sample_counts):
        part = batch_hyps[pos:pos + len(hyps)]
        assert len(part) == len(hyps)
        assert np.array_equal(part["sample"], hyps["sample"] + base)
        for f in FIELDS:
            assert np.array_equal(part[f], hyps[f]), f
        pos += len(hyps)
        base += S
    assert pos == len(batch_hyps)


@pytest.mark.parametrize("mode", ["det", "rand50", "antipodal"])
def test_batch_equals_the_single_cloud_searches(svm_model, mode):
    from agile_grasp_amd import binding

    scs = _scenes()
    kw = dict(normals_mode=binding.NORMALS_RAND50, rand_seed=3) if mode == "rand50" else {}
    anti = mode == "antipodal"
    ctx = binding.Context(scs[0].cam_origins, **kw)
    off = ctx.set_cloud_batch([s.xyz for s in scs], [s.cam for s in scs])
    samples = np.concatenate([s.samples + off[k] for k, s in enumerate(scs)]).astype(np.int32)
    hyps = ctx.find_hands(samples, calculates_antipodal=anti)
    ctx.load_svm(*svm_model)
    keep = ctx.classify()
    if mode == "rand50":
        # ONE rand() stream runs through the whole sample list of a call (quadric.cpp:184), so only the first cloud's part
        # has a single-cloud call with the same stream position; the later parts are checked for structure
        one = _single(scs[0], svm_model, **kw)
        h0 = one.find_hands(scs[0].samples)
        _compare(hyps[hyps["sample"] < scs[0].samples.size], [h0], [scs[0].samples.size])
        assert (np.diff(hyps["sample"]) >= 0).all() and len(hyps) > len(h0)
        return
    per, per_keep = [], []
    for s in scs:
        one = _single(s, svm_model, **kw)
        h = one.find_hands(s.samples, calculates_antipodal=anti)
        one.load_svm(*svm_model)
        per.append(h)
        per_keep.append(one.classify())
    _compare(hyps, per, [s.samples.size for s in scs])
    assert np.array_equal(keep, np.concatenate(per_keep)) and keep.sum() > 0
    if anti:
        assert hyps["half_antipodal"].sum() > 0
    # GraspHypothesis::getPointsForLearning of a hypothesis of the LAST cloud: only its own cloud's points
    last = len(hyps) - 1
    pts, cam = ctx.learning_points(last)
    one = _single(scs[-1], svm_model)
    h = one.find_hands(scs[-1].samples, calculates_antipodal=anti)
    pts1, cam1 = one.learning_points(len(h) - 1)
    assert np.array_equal(pts, pts1) and np.array_equal(cam, cam1) and pts.shape[1] == hyps["n_in_box"][last]


def test_batch_edge_cases(tiny_scene):
    from agile_grasp_amd import binding

    sc = tiny_scene
    ctx = binding.Context(sc.cam_origins)
    # an empty cloud in the middle of the batch, and the same cloud twice
    empty = np.zeros((0, 3), np.float32)
    off = ctx.set_cloud_batch([sc.xyz, empty, sc.xyz], [sc.cam, np.zeros(0, np.int32), sc.cam])
    assert list(off) == [0, sc.n, sc.n, 2 * sc.n]
    samples = np.concatenate([sc.samples, sc.samples + off[2]]).astype(np.int32)
    hyps = ctx.find_hands(samples)
    one = binding.Context(sc.cam_origins)
    one.set_cloud(sc.xyz, sc.cam)
    h = one.find_hands(sc.samples)
    _compare(hyps, [h, h], [sc.samples.size, sc.samples.size])
    # back to a single cloud on the same context
    ctx.set_cloud(sc.xyz, sc.cam)
    again = ctx.find_hands(sc.samples)
    _compare(again, [h], [sc.samples.size])
    with pytest.raises(binding.AghError):
        ctx.set_cloud_batch([sc.xyz] * 65, [sc.cam] * 65)


def test_full_size_c5_batch_against_oracle(svm_model):
    """BASELINE config C5 at full size on one GPU: the eight 300k-point clouds (seeds 10..17) in ONE context, 16 000 samples in
    one launch set, against the ORACLE cloud by cloud (frames, hypotheses, SVM labels bit-identical)."""
    from agile_grasp_amd import binding, synthetic
    from oracle import oracle_py as O

    w, rho = svm_model
    scs = [synthetic.config(f"C5_{k}") for k in range(8)]
    ctx = binding.Context(scs[0].cam_origins)
    off = ctx.set_cloud_batch([s.xyz for s in scs], [s.cam for s in scs])
    samples = np.concatenate([s.samples + off[k] for k, s in enumerate(scs)]).astype(np.int32)
    hyps = ctx.find_hands(samples)
    ctx.load_svm(w, rho)
    keep = ctx.classify()
    frames = ctx.frames()
    pos = base = 0
    for k, sc in enumerate(scs):
        ref = O.find_hands(O.default_params(sc.cam_origins), sc.xyz, sc.cam, sc.samples, want_images=True)
        okeep, _ = O.classify(ref["images"], w, rho)
        n = len(ref["hyps"])
        part = hyps[pos:pos + n]
        assert np.array_equal(part["sample"], ref["hyps"]["sample"] + base), k
        for f in FIELDS:
            assert np.array_equal(part[f], ref["hyps"][f]), (k, f)
        assert np.array_equal(keep[pos:pos + n], okeep), k
        fr = frames[base:base + sc.samples.size]
        for f in ("normal", "axis", "binormal", "params", "n_nb", "max_index", "majority_cam", "valid"):
            assert np.array_equal(fr[f], ref["frames"][f]), (k, f)
        pos += n
        base += sc.samples.size
    assert pos == len(hyps) > 5000
