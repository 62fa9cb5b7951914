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

for _ in range(3):
    gpu_chain()
K = 20
acc = np.zeros(4)
for _ in range(K):
    h, k, hd, idx, dt = gpu_chain()
    acc += dt
acc /= K
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
    "cpu_oracle_ms": {"preprocess": t_pre * 1e3, "find_hands": (t2 - t1) * 1e3, "classify": (t3 - t2) * 1e3,
                      "find_handles": (t4 - t3) * 1e3, "total": (t_pre + (t4 - t1)) * 1e3,
                      "note": f"the whole chain on the host: preprocessing and handle search on one thread, search and "
                              f"classification with {THREADS} OpenMP threads"},
}))
