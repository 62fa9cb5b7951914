"""What would ONE cloud's step cost if its sample list ran as two (three, four) parts side by side?  Emulated with a context per part on
streams of their own, every part building the grid for itself (an in-library split would build it once): an upper bound of the time
a split inside agh_find_hands_device could reach.    python scripts/micro/split_halves_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from agile_grasp_amd import binding, synthetic  # noqa: E402

dev = torch.device("cuda", 0)
sc = synthetic.config("C2")
xyz_t, cam_t = torch.from_numpy(sc.xyz).to(dev), torch.from_numpy(sc.cam).to(dev)
S = sc.samples.size
for parts in (1, 2, 3, 4):
    lanes = []
    for p in range(parts):
        lo, hi = S * p // parts, S * (p + 1) // parts
        lanes.append((binding.Context(sc.cam_origins, profile=0), torch.cuda.Stream(device=dev),
                      torch.from_numpy(sc.samples[lo:hi].copy()).to(dev), torch.zeros(8 * S * 160, dtype=torch.uint8, device=dev),
                      torch.zeros(1, dtype=torch.int64, device=dev)))
    torch.cuda.synchronize()

    def step():
        for c, st, s_t, out_t, nout_t in lanes:
            c.set_cloud_torch(xyz_t, cam_t, stream=st.cuda_stream)
            c.find_hands_torch(s_t, out_t, nout_t, stream=st.cuda_stream)

    def step_sync():  # one cloud at a time: the next cloud's parts start when this one's are all done
        step()
        torch.cuda.synchronize()

    for _ in range(20):
        step_sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(50):
            step_sync()
        ts.append((time.perf_counter() - t0) / 50 * 1e3)
    n = sum(int(l[4].item()) for l in lanes)
    print("parts %d: %.4f ms per cloud (latency, host sync per cloud included)  [%s]  hyp %d" % (parts, sorted(ts)[2], ", ".join("%.4f" % t for t in ts), n), flush=True)
    for l in lanes:
        l[0].close()
