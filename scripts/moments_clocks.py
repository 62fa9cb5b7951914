# Per-work-group phase clocks of k_taubin_moments (debug build: AGH_DEBUG_BUILD=1, see sweep_clocks.py).
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AGH_DEBUG_CLOCKS"] = "/tmp/agh_mclocks.bin"
os.environ["AGH_DEBUG_CLOCKS_KERNEL"] = "moments"
import numpy as np
from agile_grasp_amd import binding, synthetic
sc = synthetic.config(sys.argv[1] if len(sys.argv) > 1 else "C2")
ctx = binding.Context(sc.cam_origins)
ctx.set_cloud(sc.xyz, sc.cam)
for _ in range(3):
    ctx.find_hands(sc.samples)
ctx.synchronize()
d = np.fromfile("/tmp/agh_mclocks.bin", np.int64).reshape(-1, 8).astype(np.float64) / 100.0  # us
n = ctx.frames()["n_nb"]
t0 = d[:, 0].min()
dur = d[:, 4] - d[:, 0]
print("span %.1f us; work-group duration median %.1f p90 %.1f max %.1f; sum/1024 slots %.1f" % (d[:, 4].max() - t0, np.median(dur), np.percentile(dur, 90), dur.max(), dur.sum() / 1024))
for k, v in {"rows": d[:, 1] - d[:, 0], "gather": d[:, 2] - d[:, 1], "sort": d[:, 3] - d[:, 2], "list write + sums": d[:, 4] - d[:, 3],
             "  forming products (to barrier)": d[:, 5], "  chain (wave 0)": d[:, 6], "  second barrier": d[:, 7]}.items():
    print("%-34s median %.2f  p90 %.2f  max %.2f" % (k, np.median(v), np.percentile(v, 90), v.max()))
chunks = np.ceil(n / 56.0)
print("per chunk: products %.3f us, chain %.3f us, barrier %.3f us (median n %d, %d chunks)" % (
    np.median(d[:, 5] / chunks), np.median(d[:, 6] / chunks), np.median(d[:, 7] / chunks), np.median(n), np.median(chunks)))
