# Per-work-group phase clocks of k_taubin_frame (debug build: AGH_DEBUG_BUILD=1, see sweep_clocks.py).
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AGH_DEBUG_CLOCKS"] = "/tmp/agh_fclocks.bin"
os.environ["AGH_DEBUG_CLOCKS_KERNEL"] = "frame"
import numpy as np
from agile_grasp_amd import binding, synthetic
sc = synthetic.config(sys.argv[1] if len(sys.argv) > 1 else "C2")
ctx = binding.Context(sc.cam_origins, normals_mode=binding.NORMALS_RAND50 if "rand50" in sys.argv else binding.NORMALS_DETERMINISTIC)
ctx.set_cloud(sc.xyz, sc.cam)
for _ in range(3):
    ctx.find_hands(sc.samples)
ctx.synchronize()
d = np.fromfile("/tmp/agh_fclocks.bin", np.int64).reshape(-1, 8).astype(np.float64) / 100.0  # us (100 MHz)
fr = ctx.frames()
n = fr["n_nb"]
t0 = d[:, 0].min()
names = ["loads (eigenpair, n)", "normals", "moments T (waves 1-3)", "candidates written (w1-3)", "M3 + axis (wave 0)", "exact sums", "tail"]
print("span %.1f us; work-group duration median %.1f p90 %.1f max %.1f; sum/1280 slots %.1f us" % (
    d[:, 7].max() - t0, np.median(d[:, 7] - d[:, 0]), np.percentile(d[:, 7] - d[:, 0], 90), (d[:, 7] - d[:, 0]).max(), (d[:, 7] - d[:, 0]).sum() / 1280))
print("start spread: median %.1f p90 %.1f max %.1f" % tuple(np.percentile(d[:, 0] - t0, [50, 90, 100])))
seg = {"loads": d[:, 1] - d[:, 0], "normals": d[:, 2] - d[:, 1], "T moments (w1)": d[:, 3] - d[:, 2], "weights+estimates+list (w1)": d[:, 4] - d[:, 3],
       "M3+axis (w0, from normals)": d[:, 5] - d[:, 2], "join -> exact sums done": d[:, 6] - np.maximum(d[:, 4], d[:, 5]), "tail": d[:, 7] - d[:, 6]}
for k, v in seg.items():
    print("%-32s median %.2f  p90 %.2f  max %.2f" % (k, np.median(v), np.percentile(v, 90), v.max()))
for lo, hi in ((0, 500), (500, 600), (600, 800), (800, 1200)):
    m = (n >= lo) & (n < hi)
    if m.any():
        print("n in [%d,%d): %d work-groups, duration %.1f" % (lo, hi, m.sum(), (d[m, 7] - d[m, 0]).mean()))
