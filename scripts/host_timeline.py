"""The workload of the host-path timeline: K calls of what the C++ adapter's HandSearch::findHands issues --
agh_set_cloud + agh_find_hands with host buffers -- and K runs of the facade chain (preprocess -> find_hands -> classify ->
find_handles), each phase fenced by a marker the trace can find: a hipDriverGetVersion call (the library never makes one) before every call.
    rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d DIR -o ht -- python scripts/host_timeline.py [api|pipeline] [K]
"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from agile_grasp_amd import binding, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1] if len(sys.argv) > 1 else "api"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20

hip = C.CDLL("libamdhip64.so")
ver = C.c_int(0)


def marker():
    hip.hipDriverGetVersion(C.byref(ver))  # a HIP API call the library never makes: the call boundary in the trace


z = np.load(os.path.join(ROOT, "tests", "golden", "svm_weights.npz"))
if mode == "api":
    sc = synthetic.config("C2")
    ctx = binding.Context(sc.cam_origins)
    for _ in range(5):
        ctx.set_cloud(sc.xyz, sc.cam)
        h = ctx.find_hands(sc.samples)
    t = np.zeros(2)
    for _ in range(K):
        marker()
        t0 = time.perf_counter()
        ctx.set_cloud(sc.xyz, sc.cam)
        t1 = time.perf_counter()
        h = ctx.find_hands(sc.samples)
        t2 = time.perf_counter()
        t += (t1 - t0, t2 - t1)
    marker()
    print(json.dumps({"mode": mode, "calls": K, "ms_set_cloud": t[0] / K * 1e3, "ms_find_hands": t[1] / K * 1e3,
                      "ms_per_call": t.sum() / K * 1e3, "hypotheses": int(len(h))}))
else:
    rc = synthetic.make_raw_cloud(700_000, 21)
    ctx = binding.Context(rc.cam_origins)
    ctx.load_svm(z["w"], float(z["rho"]))
    nv = ctx.preprocess(rc.xyz, rc.size_left, rc.workspace)
    samples = np.sort(np.random.default_rng(5).permutation(nv)[:2000]).astype(np.int32)

    def chain(mark):
        ts = [time.perf_counter()]
        if mark:
            marker()
        ctx.preprocess(rc.xyz, rc.size_left, rc.workspace); ts.append(time.perf_counter())
        if mark:
            marker()
        h = ctx.find_hands(samples); ts.append(time.perf_counter())
        if mark:
            marker()
        k = ctx.classify().astype(bool); ts.append(time.perf_counter())
        if mark:
            marker()
        hd, idx = ctx.find_handles(h[k], 3, 0.005); ts.append(time.perf_counter())
        return np.diff(ts), len(h), int(k.sum()), len(hd)

    for _ in range(3):
        chain(False)
    acc = np.zeros(4)
    for _ in range(K):
        dt, nh, nk, nd = chain(True)
        acc += dt
    marker()
    acc /= K
    print(json.dumps({"mode": mode, "calls": K, "voxels": int(nv), "hypotheses": nh, "svm_kept": nk, "handles": nd,
                      "ms": {"preprocess": acc[0] * 1e3, "find_hands": acc[1] * 1e3, "classify": acc[2] * 1e3,
                             "find_handles": acc[3] * 1e3, "total": acc.sum() * 1e3,
                             "note": "under rocprofv3 the HIP calls are slower than in an untraced run"}}))
