import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from agile_grasp_amd import binding, synthetic
sc = synthetic.config(sys.argv[1] if len(sys.argv) > 1 else "C2")
ctx = binding.Context(sc.cam_origins)
ctx.set_cloud(sc.xyz, sc.cam)
for i in range(4):
    t = time.time(); h = ctx.find_hands(sc.samples); print("call", i, len(h), round(time.time() - t, 4), flush=True)
