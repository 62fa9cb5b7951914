"""Does running two half-batches on two HIP streams overlap usefully? (experiment, not a test)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from agile_grasp_amd import binding, synthetic
sc = synthetic.config("C2"); S = sc.samples.size
xyz_t = torch.from_numpy(sc.xyz).cuda(); cam_t = torch.from_numpy(sc.cam).cuda()
def mk(samples):
    c = binding.Context(sc.cam_origins)
    s_t = torch.from_numpy(np.ascontiguousarray(samples)).cuda()
    out = torch.zeros(8 * len(samples) * 160, dtype=torch.uint8, device="cuda"); n = torch.zeros(1, dtype=torch.int64, device="cuda")
    return c, s_t, out, n
def run(parts, iters=30):
    ctxs = [mk(p) for p in parts]
    streams = [torch.cuda.Stream() for _ in parts]
    for c, s_t, out, n in ctxs:
        c.set_cloud_torch(xyz_t, cam_t)
    torch.cuda.synchronize()
    for it in range(iters + 3):
        if it == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        for (c, s_t, out, n), st in zip(ctxs, streams):
            c.find_hands_torch(s_t, out, n, stream=st.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3
print("1 x 2000 :", round(run([sc.samples]), 4), "ms")
print("2 x 1000 :", round(run([sc.samples[:S // 2], sc.samples[S // 2:]]), 4), "ms")
print("4 x 500  :", round(run([sc.samples[i * S // 4:(i + 1) * S // 4] for i in range(4)]), 4), "ms")
print("2 x 2000 (two full jobs):", round(run([sc.samples, sc.samples]), 4), "ms")
