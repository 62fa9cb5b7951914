"""Kernel times of a C2 search restricted to the first S samples (what one rank of an N-GPU sample-sharded run executes):
    python scripts/slice_timing.py 2000 1000 500 250"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from agile_grasp_amd import binding, synthetic  # noqa: E402

sc = synthetic.config("C2")
dev = torch.device("cuda", 0)
xyz_t = torch.from_numpy(sc.xyz).to(dev)
cam_t = torch.from_numpy(sc.cam).to(dev)
for S in [int(v) for v in sys.argv[1:]] or [2000, 1000, 500, 250]:
    ctx = binding.Context(sc.cam_origins, profile=0)
    # a rank's slice is contiguous in the (spatially sorted) sample list: take the middle one
    lo = (sc.samples.size - S) // 2
    s_t = torch.from_numpy(sc.samples[lo:lo + S].copy()).to(dev)
    out_t = torch.zeros(8 * S * 160, dtype=torch.uint8, device=dev)
    nout_t = torch.zeros(1, dtype=torch.int64, device=dev)
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)

    def step():
        ctx.set_cloud_torch(xyz_t, cam_t, stream=ts.cuda_stream)
        ctx.find_hands_torch(s_t, out_t, nout_t, stream=ts.cuda_stream)

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    ctx.set_profile(1)
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    k = {n: round(v / 50 * 1e3, 1) for n, v in ctx.timing().items()}
    print(json.dumps({"samples": S, "us_per_step": round(dt * 1e6, 1), "hypotheses": int(nout_t.item()), "kernel_us": k}))
    ctx.close()
