import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from agile_grasp_amd import binding, synthetic
sc = synthetic.config("C2")
ctx = binding.Context(sc.cam_origins, profile=1)
ctx.set_cloud(sc.xyz, sc.cam)
for _ in range(2):
    h = ctx.find_hands(sc.samples, calculates_antipodal=True)
ctx.timing()
t0 = time.perf_counter()
for _ in range(5):
    h = ctx.find_hands(sc.samples, calculates_antipodal=True)
dt = (time.perf_counter() - t0) / 5
print("C2 calculates_antipodal: wall ms", round(dt * 1e3, 2), "hyps", len(h), "half", int(h["half_antipodal"].sum()), "full", int(h["full_antipodal"].sum()), {k: round(v / 5, 3) for k, v in ctx.timing().items()})
