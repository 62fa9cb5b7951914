"""Wall time of the HOST-buffer entry points (what the C++ adapter calls): agh_set_cloud + agh_find_hands (+ classify)
with numpy arrays in and out, against the device-resident path bench.py measures."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from agile_grasp_amd import binding, synthetic

sc = synthetic.config("C2")
ctx = binding.Context(sc.cam_origins)
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "svm_weights.npz"))
ctx.load_svm(z["w"], float(z["rho"]))
for _ in range(3):
    ctx.set_cloud(sc.xyz, sc.cam); h = ctx.find_hands(sc.samples); k = ctx.classify()
K = 30
t = {"set_cloud": 0.0, "find_hands": 0.0, "classify": 0.0}
for _ in range(K):
    t0 = time.perf_counter(); ctx.set_cloud(sc.xyz, sc.cam); t1 = time.perf_counter()
    h = ctx.find_hands(sc.samples); t2 = time.perf_counter()
    k = ctx.classify(); t3 = time.perf_counter()
    t["set_cloud"] += t1 - t0; t["find_hands"] += t2 - t1; t["classify"] += t3 - t2
print({k: round(v / K * 1e3, 3) for k, v in t.items()}, "ms; hypotheses", len(h), "kept", int(k.sum()))
