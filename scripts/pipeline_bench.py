"""End-to-end figure for the facade's call chain on one raw two-view capture (what grasp_localizer.cpp:95-103 runs per
cloud): preprocess -> findHands (2000 samples) -> classify -> findHandles, host buffers in and out, against the same
chain through the oracle on the host (all cores for the search, one core for the rest).  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from agile_grasp_amd import binding, synthetic
from oracle import oracle_py as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rc = synthetic.make_raw_cloud(700_000, 21)
z = np.load(os.path.join(ROOT, "tests", "golden", "svm_weights.npz"))
w, rho = z["w"], float(z["rho"])
ctx = binding.Context(rc.cam_origins)
ctx.load_svm(w, rho)
nv = ctx.preprocess(rc.xyz, rc.size_left, rc.workspace)
samples = np.sort(np.random.default_rng(5).permutation(nv)[:2000]).astype(np.int32)

def gpu_chain():
    t = [time.perf_counter()]
    ctx.preprocess(rc.xyz, rc.size_left, rc.workspace); t.append(time.perf_counter())
    h = ctx.find_hands(samples); t.append(time.perf_counter())
    k = ctx.classify().astype(bool); t.append(time.perf_counter())
    hd, idx = ctx.find_handles(h[k], 3, 0.005); t.append(time.perf_counter())
    return h, k, hd, idx, np.diff(t)

def gpu_one_call():
    t0 = time.perf_counter()
    r = ctx.localize(rc.xyz, rc.size_left, rc.workspace, samples=samples, classify=True, min_inliers=3, min_length=0.005)
    return r, time.perf_counter() - t0

for _ in range(3):
    gpu_chain()
K = 20
acc = np.zeros(4)
for _ in range(K):
    h, k, hd, idx, dt = gpu_chain()
    acc += dt
acc /= K
for _ in range(3):
    gpu_one_call()
one = []
for _ in range(K):
    r1, dt1 = gpu_one_call()
    one.append(dt1)
one_ms = float(np.median(one)) * 1e3
assert len(r1["handles"]) == len(hd) and np.array_equal(r1["inlier_idx"], idx) and r1["n_hypotheses"] == len(h)
import torch
xyz_dev = torch.from_numpy(rc.xyz).cuda()
ondev = []
for _ in range(K + 3):  # the raw capture already in device memory: agh_localize_device (no 8.4 MB upload)
    t0 = time.perf_counter()
    rd = ctx.localize(xyz_dev, rc.size_left, rc.workspace, samples=samples, classify=True, min_inliers=3, min_length=0.005)
    ondev.append(time.perf_counter() - t0)
ondev = ondev[3:]
assert len(rd["handles"]) == len(hd) and np.array_equal(rd["inlier_idx"], idx)
drawn = []
for _ in range(K):  # the sample list drawn on the device instead of uploaded
    t0 = time.perf_counter()
    ctx.localize(rc.xyz, rc.size_left, rc.workspace, n_samples=2000, sample_seed=5, classify=True)
    drawn.append(time.perf_counter() - t0)
t0 = time.perf_counter()
v, cam = orc.preprocess(rc.xyz, rc.size_left, rc.workspace); t1_pre = time.perf_counter()
THREADS = int(os.environ.get("PIPELINE_BENCH_CPU_THREADS", "32"))  # the port scales to a few dozen threads, not to 256
p = orc.default_params(rc.cam_origins, num_threads=THREADS)
orc.find_hands(p, v, cam, samples[:8])  # warm-up
t1 = time.perf_counter()
r = orc.find_hands(p, v, cam, samples, want_images=True); t2 = time.perf_counter()
keep, _sums = orc.classify(r["images"], w, rho, num_threads=THREADS) if len(r["hyps"]) else (np.zeros(0, np.uint8), None); t3 = time.perf_counter()
ohd, oidx = orc.find_handles(r["hyps"][np.asarray(keep, bool)], 3, 0.005); t4 = time.perf_counter()
t_pre = t1_pre - t0
print(json.dumps({
    "raw_points": int(rc.xyz.shape[0]), "voxels": int(nv), "samples": int(len(samples)), "hypotheses": int(len(h)),
    "svm_kept": int(k.sum()), "handles": int(len(hd)),
    "gpu_ms": {"preprocess": acc[0] * 1e3, "find_hands": acc[1] * 1e3, "classify": acc[2] * 1e3, "find_handles": acc[3] * 1e3,
               "total": acc.sum() * 1e3},
    "gpu_one_call_ms": {"agh_localize": one_ms, "min": float(np.min(one)) * 1e3, "device_drawn_samples": float(np.median(drawn)) * 1e3,
                        "agh_localize_device": float(np.median(ondev)) * 1e3,
                        "note": "the same chain as ONE call with one synchronisation (agh_localize), same samples, same handles"},
    "cpu_oracle_ms": {"preprocess": t_pre * 1e3, "find_hands": (t2 - t1) * 1e3, "classify": (t3 - t2) * 1e3,
                      "find_handles": (t4 - t3) * 1e3, "total": (t_pre + (t4 - t1)) * 1e3,
                      "note": f"the whole chain on the host: preprocessing and handle search on one thread, search and "
                              f"classification with {THREADS} OpenMP threads"},
}))
