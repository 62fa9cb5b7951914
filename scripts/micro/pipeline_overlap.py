"""grasp_localizer.cpp:95-103 over a STREAM of captures: agh_localize per capture against agh_localize_begin / _stage / _end with the next
capture's upload under this one's kernels.  Host buffers in and out; per-capture times, median / min / max.
    python scripts/micro/pipeline_overlap.py [captures]"""
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from agile_grasp_amd import binding, synthetic  # noqa: E402


def main(n_caps=40):
    rc = synthetic.make_raw_cloud(700_000, 21)
    z = np.load(os.path.join(ROOT, "tests", "golden", "svm_weights.npz"))
    ctx = binding.Context(rc.cam_origins)
    ctx.load_svm(z["w"], float(z["rho"]))
    nv = ctx.preprocess(rc.xyz, rc.size_left, rc.workspace)
    samples = np.sort(np.random.default_rng(5).permutation(nv)[:2000]).astype(np.int32)
    # a ring of captures (distinct array objects, the same content: what a driver's buffer pool hands over)
    caps = [np.ascontiguousarray(rc.xyz.copy()) for _ in range(4)]
    kw = dict(samples=samples, classify=True, min_inliers=3, min_length=0.005)

    def serial():
        t = []
        for i in range(n_caps):
            t0 = time.perf_counter()
            r = ctx.localize(caps[i % 4], rc.size_left, rc.workspace, **kw)
            t.append(time.perf_counter() - t0)
        return t, r

    def overlapped():
        t = []
        ctx.localize_begin(caps[0], rc.size_left, rc.workspace, **kw)
        t0 = time.perf_counter()
        for i in range(n_caps):
            if i + 1 < n_caps:
                ctx.localize_stage(caps[(i + 1) % 4])
            r = ctx.localize_end()
            if i + 1 < n_caps:
                ctx.localize_begin(caps[(i + 1) % 4], rc.size_left, rc.workspace, **kw)
            t1 = time.perf_counter()
            t.append(t1 - t0)
            t0 = t1
        return t, r

    # two contexts taking turns: capture k + 1 begins (upload and all) on the other context before capture k is collected -- the two
    # chains' kernels run side by side (bench.py's `two_streams`, for the online chain); no staging needed
    ctx2 = binding.Context(rc.cam_origins)
    ctx2.load_svm(z["w"], float(z["rho"]))
    ctx2.preprocess(rc.xyz, rc.size_left, rc.workspace)
    lanes = [ctx, ctx2]

    def two_contexts():
        t = []
        lanes[0].localize_begin(caps[0], rc.size_left, rc.workspace, **kw)
        t0 = time.perf_counter()
        for i in range(n_caps):
            if i + 1 < n_caps:
                lanes[(i + 1) & 1].localize_begin(caps[(i + 1) % 4], rc.size_left, rc.workspace, **kw)
            r = lanes[i & 1].localize_end()
            t1 = time.perf_counter()
            t.append(t1 - t0)
            t0 = t1
        return t, r

    for _ in range(2):
        serial()
        overlapped()
        two_contexts()
    ts, rs = serial()
    to, ro = overlapped()
    t2, r2 = two_contexts()
    assert r2["n_hypotheses"] == rs["n_hypotheses"] and np.array_equal(r2["inlier_idx"], rs["inlier_idx"])
    assert rs["n_hypotheses"] == ro["n_hypotheses"] and np.array_equal(rs["inlier_idx"], ro["inlier_idx"])
    for f in ("axis", "center", "width"):
        assert np.array_equal(rs["handles"][f], ro["handles"][f]), f

    def st(t):
        t = t[2:]
        return {"median_ms": statistics.median(t) * 1e3, "min_ms": min(t) * 1e3, "max_ms": max(t) * 1e3, "mean_ms": sum(t) / len(t) * 1e3}

    print(json.dumps({"workload": "stream of raw two-view captures, 699999 points each -> 3 mm voxels -> 2000-sample search -> HOG + SVM -> "
                                  "handle search, host buffers in and out", "captures": n_caps, "hypotheses": int(rs["n_hypotheses"]),
                      "handles": int(len(rs["handles"])), "agh_localize_per_capture": st(ts),
                      "begin_stage_end_per_capture": st(to), "two_contexts_taking_turns_per_capture": st(t2),
                      "note": "steady state of a caller that stages capture k + 1 (agh_localize_stage) between agh_localize_begin and "
                              "agh_localize_end of capture k; results equal agh_localize's bit for bit"}))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
