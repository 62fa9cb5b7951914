"""HandleSearch::findHandles + Handle (handle_search.cpp:4-128, handle.cpp:3-74).

CPU part: the C++ oracle against an independent numpy transcription (inlier sets and order).  GPU part: the HIP path
(through the C ABI) against the oracle, every field with exact equality.
"""
import numpy as np
import pytest

from tests import ref_numpy

FIELDS = ("axis", "center", "approach", "binormal", "hands_center", "width", "n_inliers", "first_inlier")


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _crafted(seed, n_handles=5, per=14, clutter=60, ties=False, dead=0, nonunit=0):
    """Hands along a few straight handles (common axis and approach up to small noise, bottoms spread along the axis,
    one 3 cm hole in every second handle so that shortenHandle has something to cut) plus clutter."""
    from oracle import oracle_py as orc

    rng = np.random.default_rng(seed)
    recs = []
    for h in range(n_handles):
        axis = _unit(rng.normal(size=3))
        appr = _unit(np.cross(axis, rng.normal(size=3)))
        origin = rng.uniform(-0.3, 0.3, 3)
        t = np.sort(rng.uniform(0.0, 0.12, per))
        if ties:
            t = np.round(t / 0.01) * 0.01  # equal distances along the axis: the sort's tie rule matters
        if h % 2 == 1:
            t[per // 2:] += 0.03
        for k in range(per):
            r = np.zeros(1, orc.HYP_DTYPE)[0]
            r["axis"] = _unit(axis + rng.normal(scale=0.02, size=3)) * (1 if rng.random() < 0.8 else -1)
            r["approach"] = _unit(appr + rng.normal(scale=0.03, size=3))
            r["bottom"] = origin + t[k] * axis + rng.normal(scale=0.001, size=3)
            r["surface"] = r["bottom"] - 0.03 * r["approach"]
            r["binormal"] = np.cross(r["approach"], r["axis"])
            r["width"] = rng.uniform(0.01, 0.08)
            r["valid"] = 1
            recs.append(r)
    for _ in range(clutter):
        r = np.zeros(1, orc.HYP_DTYPE)[0]
        r["axis"] = _unit(rng.normal(size=3))
        r["approach"] = _unit(rng.normal(size=3))
        r["bottom"] = rng.uniform(-0.3, 0.3, 3)
        r["surface"] = r["bottom"]
        r["width"] = rng.uniform(0.01, 0.08)
        recs.append(r)
    hands = np.array(recs, orc.HYP_DTYPE)[rng.permutation(len(recs))]
    if dead:
        hands["width"][rng.choice(len(hands), dead, replace=False)] = -1.0  # already retired (handle_search.cpp:13)
    if nonunit:
        # hands a caller did not normalise: axis . axis = 0.81 fails the alignment test against ITSELF (handle_search.cpp:19-28
        # makes no exception for j == i), so such a seed is not among its own inliers
        sel = rng.choice(len(hands), nonunit, replace=False)
        hands["axis"][sel] *= 0.9
        hands["approach"][sel] *= 0.9
    return hands


def _cases():
    return {
        "crafted": (_crafted(1), 3, 0.005),
        "crafted_ties": (_crafted(2, ties=True), 3, 0.005),
        "crafted_dead": (_crafted(3, dead=25), 3, 0.005),
        "crafted_min5": (_crafted(4, per=9), 5, 0.02),
        "crafted_long": (_crafted(5, n_handles=2, per=120, clutter=10), 10, 0.005),
        "crafted_nonunit": (_crafted(9, n_handles=6, per=16, nonunit=40), 3, 0.005),
        "none_found": (_crafted(6, n_handles=0, clutter=80), 3, 0.005),
        "single": (_crafted(7, n_handles=0, clutter=1), 1, 0.005),
        "empty": (_crafted(8, n_handles=0, clutter=0), 3, 0.005),
    }


@pytest.mark.parametrize("name", sorted(_cases()))
def test_oracle_matches_numpy_transcription(name):
    from oracle import oracle_py as orc

    hands, mi, ml = _cases()[name]
    hd, idx = orc.find_handles(hands, mi, ml)
    got = [list(idx[h["first_inlier"]:h["first_inlier"] + h["n_inliers"]]) for h in hd]
    assert got == ref_numpy.find_handles(hands, mi, ml)
    if name.startswith("crafted"):
        assert len(hd) >= 1
    for h in hd:  # Handle invariants (handle.cpp): unit axis, binormal = approach x axis, width = mean
        sel = idx[h["first_inlier"]:h["first_inlier"] + h["n_inliers"]]
        assert abs(np.linalg.norm(h["axis"]) - 1.0) < 1e-12
        assert np.allclose(h["binormal"], np.cross(h["approach"], h["axis"]), atol=1e-15)
        assert abs(h["width"] - hands["width"][sel].mean()) < 1e-15
        assert np.dot(h["axis"], hands["axis"][sel[0]]) >= 0


def test_oracle_on_search_output(small_scene):
    from oracle import oracle_py as orc

    sc = small_scene
    hyps = orc.find_hands(orc.default_params(sc.cam_origins), sc.xyz, sc.cam, sc.samples)["hyps"]
    hd, idx = orc.find_handles(hyps, 3, 0.005)
    got = [list(idx[h["first_inlier"]:h["first_inlier"] + h["n_inliers"]]) for h in hd]
    assert got == ref_numpy.find_handles(hyps, 3, 0.005) and len(hd) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(_cases()))
def test_gpu_handle_search_bit_exact(name):
    from agile_grasp_amd import binding, synthetic
    from oracle import oracle_py as orc

    hands, mi, ml = _cases()[name]
    ctx = binding.Context(synthetic.camera_origins())
    for _ in range(2):  # (buffers are reused by the second call)
        ghd, gidx = ctx.find_handles(hands, mi, ml)
        hd, idx = orc.find_handles(hands, mi, ml)
        assert len(ghd) == len(hd) and np.array_equal(gidx, idx)
        for f in FIELDS:
            assert np.array_equal(ghd[f], hd[f]), f


@pytest.mark.gpu
def test_gpu_handle_search_on_search_output(small_scene, svm_model):
    from agile_grasp_amd import binding
    from oracle import oracle_py as orc

    sc = small_scene
    ctx = binding.Context(sc.cam_origins)
    ctx.set_cloud(sc.xyz, sc.cam)
    hyps = ctx.find_hands(sc.samples)
    ghd, gidx = ctx.find_handles(hyps, 3, 0.005)
    hd, idx = orc.find_handles(hyps, 3, 0.005)
    assert len(hd) > 0 and len(ghd) == len(hd) and np.array_equal(gidx, idx)
    for f in FIELDS:
        assert np.array_equal(ghd[f], hd[f]), f


@pytest.mark.gpu
def test_too_many_hands_is_loud():
    from agile_grasp_amd import binding, synthetic

    ctx = binding.Context(synthetic.camera_origins())
    with pytest.raises(binding.AghError) as e:
        ctx.find_handles(np.zeros(8193, binding.HYP_DTYPE), 3, 0.005)
    assert e.value.code == -4
