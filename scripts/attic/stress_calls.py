"""Soak test of the asynchronous path (not a pytest: it runs for a while): thousands of grid builds and searches on
clouds of varying size through one context, watching for hangs (run it under `timeout`) and for results that change
between repetitions.  Usage: python scripts/stress_calls.py [iterations]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from agile_grasp_amd import binding, synthetic

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda", 0)
ts = torch.cuda.Stream()
sc = synthetic.config("C2")
ctx = binding.Context(sc.cam_origins)
_sync = ctx.synchronize
def _synchronize():  # (AGH_ERR_RETRY: the context switched its larger capacity classes on; the caller's next call is complete)
    try:
        _sync()
    except binding.AghError as e:
        if e.code != binding.AGH_ERR_RETRY:
            raise
        print("AGH_ERR_RETRY once (expected at most once per context)", flush=True)
ctx.synchronize = _synchronize
xyz_t = torch.from_numpy(sc.xyz).to(dev); cam_t = torch.from_numpy(sc.cam).to(dev)
out_t = torch.zeros(8 * 2000 * 160, dtype=torch.uint8, device=dev); n_t = torch.zeros(1, dtype=torch.int64, device=dev)
rng = np.random.default_rng(0)
sizes = [300000, 1000, 257, 123456, 64, 299999, 5000, 77777]
ref = {}
t0 = time.time()
for it in range(iters):
    n = sizes[it % len(sizes)]
    S = int(rng.integers(1, 2000))
    s_np = np.sort(rng.permutation(n)[:min(S, n)]).astype(np.int32)
    s_t = torch.from_numpy(s_np).to(dev)
    torch.cuda.synchronize()
    ctx.set_cloud_torch(xyz_t[:n], cam_t[:n], stream=ts.cuda_stream)
    ctx.find_hands_torch(s_t, out_t, n_t, stream=ts.cuda_stream)
    if it % 8 < 2:  # the first two sizes of every round are checked against a repetition with the same inputs
        ctx.synchronize()
        k = int(n_t.item())
        def records():  # without the per-call stamp (agh_hypothesis::epoch, the last word of every record)
            r = out_t[:k * 160].cpu().numpy().copy().reshape(k, 160)
            r[:, 156:160] = 0
            return r

        a = records()
        ctx.set_cloud_torch(xyz_t[:n], cam_t[:n], stream=ts.cuda_stream)
        ctx.find_hands_torch(s_t, out_t, n_t, stream=ts.cuda_stream)
        ctx.synchronize()
        assert int(n_t.item()) == k and np.array_equal(records(), a), ("result changed", it, n, S)
    if it % 500 == 499:
        ctx.synchronize()
        print("iteration", it + 1, "ok,", round(time.time() - t0, 1), "s", flush=True)
ctx.synchronize()
print("DONE", iters, "iterations in", round(time.time() - t0, 1), "s")
