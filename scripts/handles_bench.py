"""Row f2 measurement: agh_find_handles (host records in, host records out: H2D + 3 kernels + D2H) against the oracle's
restatement of HandleSearch::findHandles on the host, on the hypotheses of the C2 cloud.  One JSON line per input."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from agile_grasp_amd import binding, synthetic
from oracle import oracle_py as orc

sc = synthetic.config("C2")
ctx = binding.Context(sc.cam_origins, profile=1)
ctx.set_cloud(sc.xyz, sc.cam)
hyps = ctx.find_hands(sc.samples)
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "svm_weights.npz"))
ctx.load_svm(z["w"], float(z["rho"]))
keep = ctx.classify().astype(bool)
for name, hands in (("all hypotheses of C2", hyps), ("SVM-positive hypotheses of C2", hyps[keep]),
                    ("C2 hypotheses x4 (jittered copies)", np.concatenate([hyps] * 4))):
    if "x4" in name:
        hands = hands.copy()
        hands["bottom"] += np.random.default_rng(0).normal(scale=1e-4, size=hands["bottom"].shape)
    for _ in range(3):
        hd, idx = ctx.find_handles(hands, 3, 0.005)
    ctx.timing()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        hd, idx = ctx.find_handles(hands, 3, 0.005)
    wall = (time.perf_counter() - t0) / K
    kern = ctx.timing().get("handle_search", 0.0) / K
    t0 = time.perf_counter()
    ohd, oidx = orc.find_handles(hands, 3, 0.005)
    cpu = time.perf_counter() - t0
    ok = bool(len(hd) == len(ohd) and np.array_equal(idx, oidx) and all(np.array_equal(hd[f], ohd[f]) for f in hd.dtype.names))
    print(json.dumps({"input": name, "hands": int(len(hands)), "handles": int(len(hd)), "inliers": int(len(idx)),
                      "bit_exact_vs_oracle": ok, "gpu_kernels_ms": kern, "gpu_wall_ms_host_to_host": wall * 1e3,
                      "cpu_oracle_ms_1_thread": cpu * 1e3, "pair_tests_per_s_gpu_kernels": len(hands) ** 2 / (kern * 1e-3) if kern else None}))
