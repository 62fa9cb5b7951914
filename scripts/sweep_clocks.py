# NOTE: needs a library built with the phase-timing hooks: AGH_DEBUG_BUILD=1 python -c "from agile_grasp_amd import build; build.build(force=True)"
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AGH_DEBUG_CLOCKS"] = "/tmp/agh_clocks.bin"
if len(sys.argv) > 1:
    os.environ["AGH_DEBUG_STOP_SWEEP"] = sys.argv[1]
import numpy as np, torch
from agile_grasp_amd import binding, synthetic
sc = synthetic.config("C2")
ctx = binding.Context(sc.cam_origins)
ctx.set_cloud(sc.xyz, sc.cam)
for _ in range(3):
    h = ctx.find_hands(sc.samples)
ctx.synchronize()
d = np.fromfile("/tmp/agh_clocks.bin", np.int64).reshape(-1, 8)
fr = ctx.frames()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "sweep_clocks.npz"), clocks=d, axis=fr["axis"], normal=fr["normal"],
                    sample=fr["sample"], valid=fr["valid"], hyp_sample=h["sample"], hyp_orientation=h["orientation"],
                    cam_origins=sc.cam_origins)
t = d[:, :7].astype(np.float64)
ph = np.diff(t, axis=1)  # wall_clock64 ticks: 100 MHz on MI300-class
names = ["setup", "gather", "passA", "finger", "passB", "write"]
print("ticks are wall_clock64 units; per-WG phase durations (median / p90 / max):")
for i, n in enumerate(names):
    print(n, np.median(ph[:, i]), np.percentile(ph[:, i], 90), ph[:, i].max())
tot = t[:, 6] - t[:, 0]
print("total per WG median", np.median(tot), "p90", np.percentile(tot, 90), "max", tot.max())
print("kernel span", t[:, 6].max() - t[:, 0].min(), "first start spread", np.percentile(t[:, 0] - t[:, 0].min(), [50, 90, 100]))
print('guard trips', (d[:, 7] < 0).sum(), d[d[:, 7] < 0][:5])
cand = (d[:, 7] & 0xffffff); ball = d[:, 7] >> 32  # (since the slab clip: "ball" = cropped points, cand = slab candidates)
print("candidates mean", cand.mean(), "ball mean", ball.mean())
pa = ph[:, 2]
order = np.argsort(ball)
for q in (0.1, 0.5, 0.8, 0.9, 0.95, 0.99):
    i = order[int(q * len(order))]
    print("ball quantile", q, "ball", ball[i], "passA", pa[i], "gather", ph[i, 1], "passB", ph[i, 4])
big = ball > 7000
print("ball>7000:", big.sum(), "passA mean", pa[big].mean() if big.any() else 0, "; ball<5000 passA mean", pa[ball < 5000].mean())
print("corr(ball, passA)", np.corrcoef(ball, pa)[0, 1])
slow = pa > 5000
if slow.any():
    print("slow passA count", slow.sum(), "their ball mean", ball[slow].mean(), "min", ball[slow].min())
# schedule view: when do work-groups start and end relative to the kernel (ticks of 10 ns)
t0 = t[:, 0].min()
start, end = t[:, 0] - t0, t[:, 6] - t0
span = end.max()
print("kernel span us", span / 100.0)
for q in (0.5, 0.75, 0.9, 0.95, 0.99, 1.0):
    print("  fraction of WGs finished by q", q, "->", np.quantile(end, q) / 100.0, "us")
last = np.argsort(end)[-8:]
for i in last:
    print("  late WG", i, "start", start[i] / 100.0, "end", end[i] / 100.0, "dur", (end[i] - start[i]) / 100.0, "ball", ball[i])
busy = (end - start).sum() / 100.0
print("sum of WG durations us", busy, "-> /768 slots =", busy / 768.0, "us (perfect packing)")
for i in np.argsort(end)[-3:]:
    print("  phases of late WG", i, dict(zip(names, (ph[i] / 100.0).round(1))), "tiles", (int(d[i, 7]) >> 24) & 0xff, "cand", int(d[i, 7]) & 0xffffff, "ball", int(d[i, 7] >> 32))
