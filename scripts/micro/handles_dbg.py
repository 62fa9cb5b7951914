import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from agile_grasp_amd import binding, synthetic
sc = synthetic.config("C2")
ctx = binding.Context(sc.cam_origins)
ctx.set_cloud(sc.xyz, sc.cam)
hyps = ctx.find_hands(sc.samples)
z = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden", "svm_weights.npz"))
ctx.load_svm(z["w"], float(z["rho"]))
keep = ctx.classify().astype(bool)
for hands in (hyps[keep], hyps):
    for _ in range(3):
        hd, idx = ctx.find_handles(hands, 3, 0.005)
    ctx.synchronize()
