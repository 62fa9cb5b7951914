"""agh_localize in a loop on the pipeline bench's capture (for rocprofv3 --kernel-trace: which kernels a call is made of)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from agile_grasp_amd import binding, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rc = synthetic.make_raw_cloud(700_000, 21)
z = np.load(os.path.join(ROOT, "tests", "golden", "svm_weights.npz"))
ctx = binding.Context(rc.cam_origins)
ctx.load_svm(z["w"], float(z["rho"]))
nv = ctx.preprocess(rc.xyz, rc.size_left, rc.workspace)
samples = np.sort(np.random.default_rng(5).permutation(nv)[:2000]).astype(np.int32)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for _ in range(3):
    ctx.localize(rc.xyz, rc.size_left, rc.workspace, samples=samples)
t0 = time.perf_counter()
for _ in range(n):
    r = ctx.localize(rc.xyz, rc.size_left, rc.workspace, samples=samples)
print("ms per call", (time.perf_counter() - t0) / n * 1e3, len(r["handles"]))
