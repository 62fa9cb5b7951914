# What could a better blockIdx -> sample order buy k_hand_sweep?  (debug build: AGH_DEBUG_BUILD=1, see sweep_clocks.py)
# Measures the per-work-group durations once, then times the kernel with: sample order, a random permutation (what the
# spatial coherence of the sample order is worth), longest-first by the MEASURED durations (the bound no predictor beats),
# longest-first by cropped point count and by slab candidates (what a perfect count would give), and longest-first in
# blocks of 64 neighbouring samples (order kept inside a block).
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
env = dict(os.environ)
env.pop("AGH_DEBUG_SWEEP_ORDER", None)
subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "sweep_clocks.py")], env=env, stdout=subprocess.DEVNULL, check=True)
d = np.fromfile("/tmp/agh_clocks.bin", np.int64).reshape(-1, 8)
dur = (d[:, 6] - d[:, 0]).astype(np.float64)
ball = d[:, 7] >> 32
cand = d[:, 7] & 0xffffff
S = len(d)
rng = np.random.default_rng(1)
blk = np.arange(S) // 64
blk_w = np.bincount(blk, weights=dur)
# what is known BEFORE the launch for free: the Taubin neighbour count n_t (K1a's result)
from agile_grasp_amd import binding, synthetic
_sc = synthetic.config(cfg)
_ctx = binding.Context(_sc.cam_origins)
_ctx.set_cloud(_sc.xyz, _sc.cam)
_ctx.find_hands(_sc.samples)
nt = _ctx.frames()["n_nb"].astype(np.float64)
_ctx.close()
print("correlation of a work-group's duration with n_t: %.2f, with slab candidates: %.2f; of 64-blocks' sums: %.2f / %.2f" % (
    np.corrcoef(dur, nt)[0, 1], np.corrcoef(dur, cand)[0, 1], np.corrcoef(blk_w, np.bincount(blk, weights=nt))[0, 1],
    np.corrcoef(blk_w, np.bincount(blk, weights=cand.astype(np.float64)))[0, 1]))


def blocks_by(weight, size):
    b = np.arange(S) // size
    w = np.bincount(b, weights=weight)
    return np.concatenate([np.where(b == k)[0] for k in np.argsort(-w, kind="stable")])
orders = {
    "sample order": np.arange(S),
    "random": rng.permutation(S),
    "longest first, measured durations": np.argsort(-dur, kind="stable"),
    "longest first, cropped points": np.argsort(-ball, kind="stable"),
    "longest first, slab candidates": np.argsort(-cand, kind="stable"),
    "two-tile work-groups first, then sample order": np.concatenate([np.where(ball > 2176)[0], np.where(ball <= 2176)[0]]),
    "blocks of 64 samples, heaviest block first": np.concatenate([np.where(blk == b)[0] for b in np.argsort(-blk_w, kind="stable")]),
    "longest first by n_t (per sample)": np.argsort(-nt, kind="stable"),
    "blocks of 4 by sum of n_t": blocks_by(nt, 4),
    "blocks of 8 by sum of n_t": blocks_by(nt, 8),
    "blocks of 16 by sum of n_t": blocks_by(nt, 16),
    "blocks of 32 by sum of slab candidates": blocks_by(cand.astype(np.float64), 32),
    "blocks of 16 by sum of slab candidates": blocks_by(cand.astype(np.float64), 16),
    "blocks of 64 by sum of n_t": blocks_by(nt, 64),
    "blocks of 32 by sum of n_t": blocks_by(nt, 32),
    "blocks of 128 by sum of n_t": blocks_by(nt, 128),
    "blocks of 64 by sum of n_t^2": blocks_by(nt * nt, 64),
    "blocks of 64 by sum of slab candidates": blocks_by(cand.astype(np.float64), 64),
}
print("perfect packing: %.1f us" % (dur.sum() / 100.0 / 768))
if len(sys.argv) > 2 and sys.argv[2] == "quick":
    orders = {k: v for k, v in orders.items() if k in ("sample order", "longest first, measured durations", "blocks of 16 by sum of n_t",
                                                        "blocks of 32 by sum of n_t", "blocks of 64 by sum of n_t")}
for name, o in orders.items():
    o.astype(np.int32).tofile("/tmp/agh_order.bin")
    e = dict(env)
    e["AGH_DEBUG_SWEEP_ORDER"] = "/tmp/agh_order.bin"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", "100", "--warmup", "10",
                          "--no-cpu-baseline", "--no-extras"], env=e, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    r = json.loads(out)
    print("%-48s k_hand_sweep %.1f us   step %.4f ms" % (name, r["kernel_ms_per_step"]["hand_sweep"] * 1e3, r["ms_per_step"]))
