"""The step and its kernels on a few scenes, one line each (device-resident cloud, HIP events):
    python scripts/micro/quick_bench.py C2 C2u C4 [--rand50]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import argparse  # noqa: E402

import torch  # noqa: E402

import bench  # noqa: E402
from agile_grasp_amd import binding  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("scenes", nargs="+")
ap.add_argument("--rand50", action="store_true")
ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
args = argparse.Namespace(warmup=5, steps=a.steps, no_events=False, spin_seconds=0.3)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ts = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(ts)
for sc in a.scenes:
    r = bench.single_cloud_extra(args, dev, ts.cuda_stream, sc, binding.NORMALS_RAND50 if a.rand50 else binding.NORMALS_DETERMINISTIC, sc,
                                 steps=a.steps)
    k = r["kernel_ms_per_step"]
    print("%-5s step %.4f ms [%.4f .. %.4f]  hyp %d  " % (sc, r["ms_per_step"], r["ms_per_step_spread"]["min_ms"],
                                                        r["ms_per_step_spread"]["max_ms"], r["hypotheses"]) +
          "  ".join("%s %.1f" % (n.replace("taubin_", "t_"), v * 1e3) for n, v in k.items()), flush=True)
