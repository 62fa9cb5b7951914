"""Row f4 measurement: the training side on the GPU against the oracle's restatement on the host cores.

Training set as Learning::train(hands_list, file, cam_pos) builds it (learning.cpp:143-163): the hands of K synthetic
two-view clouds (C5's clouds: 300k points, 2000 samples, calculates_antipodal = 1), every hand that is not merely
half-antipodal, three instances each (all points / camera 0 / camera 1), label = full antipodal.
One JSON line: instance count, GPU wall time of agh_train_svm (images in, model out: upload + HOG + transpose + <= 1000
solver steps + compaction), per-step time, the oracle's time on a bounded prefix, and whether the models agree bit for bit."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from agile_grasp_amd import binding, synthetic
from oracle import oracle_py as orc

K = int(os.environ.get("TRAIN_BENCH_CLOUDS", "4"))
CPU_N = int(os.environ.get("TRAIN_BENCH_CPU_N", "3000"))
packed, labels = [], []
t_search = 0.0
ctx = None
for k in range(K):
    sc = synthetic.config(f"C5_{k}")
    if ctx is None:
        ctx = binding.Context(sc.cam_origins)
        ctx.set_training_images(True)
    ctx.set_cloud(sc.xyz, sc.cam)
    t0 = time.perf_counter()
    hyps = ctx.find_hands(sc.samples, calculates_antipodal=True)
    t_search += time.perf_counter() - t0
    im = ctx.training_images()
    use = (hyps["half_antipodal"] == 0) | (hyps["full_antipodal"] == 1)
    packed.append(im[use].reshape(-1, 250))
    labels.append(np.repeat(hyps["full_antipodal"][use].astype(np.int8), 3))
packed = np.concatenate(packed)
labels = np.concatenate(labels)
n = int(labels.size)
ctx.train_svm(packed[:64], np.r_[np.ones(32), -np.ones(32)])  # warm-up (module load)
t0 = time.perf_counter()
got = ctx.train_svm(packed, labels)
gpu_s = time.perf_counter() - t0
t0 = time.perf_counter()
got10 = ctx.train_svm(packed, labels, max_iter=100)
gpu100_s = time.perf_counter() - t0
step_ms = (gpu_s - gpu100_s) / max(got["iterations"] - got10["iterations"], 1) * 1e3
t0 = time.perf_counter()
long_run = ctx.train_svm(packed, labels, max_iter=20000)
long_s = time.perf_counter() - t0
# descriptors alone
t0 = time.perf_counter()
desc = ctx.hog_images(packed)
hog_s = time.perf_counter() - t0
# oracle on a bounded prefix that keeps both classes (same code path; its cost per step is linear in n)
m = min(n, CPU_N)
sub = np.r_[np.nonzero(labels > 0)[0][: m // 4], np.nonzero(labels <= 0)[0][: m - m // 4]]
sub.sort()
images = binding.unpack_images(packed[sub])
t0 = time.perf_counter()
feats = orc.hog_many(images)
cpu_hog_s = time.perf_counter() - t0
CPU_T = int(os.environ.get("TRAIN_BENCH_CPU_THREADS", "16"))
t0 = time.perf_counter()
ref = orc.train_svm(feats, labels[sub], num_threads=CPU_T)
cpu_s = time.perf_counter() - t0
gsub = ctx.train_svm(packed[sub], labels[sub])
same = bool(gsub["rho"] == ref["rho"] and np.array_equal(gsub["w"], ref["w"]) and gsub["iterations"] == ref["iterations"])
# the quadratic kernel (what Learning::train* asks convertData for by default): train, then classify one cloud with it
t0 = time.perf_counter()
poly = ctx.train_svm(packed, labels, kernel=binding.SVM_POLY2)
poly_s = time.perf_counter() - t0
ctx.load_svm_model(binding.SVM_POLY2, poly["sv"], poly["alpha"], poly["rho"])
ctx.set_profile(1)
for _ in range(3):
    keep = ctx.classify()
ctx.timing()
for _ in range(10):
    keep = ctx.classify()
poly_cls_ms = ctx.timing().get("hog_svm", 0.0) / 10
ctx.load_svm(got["w"], got["rho"])
for _ in range(3):
    ctx.classify()
ctx.timing()
for _ in range(10):
    ctx.classify()
lin_cls_ms = ctx.timing().get("hog_svm", 0.0) / 10
rp = orc.train_svm(feats, labels[sub], kernel=1, num_threads=CPU_T)
gp = ctx.train_svm(packed[sub], labels[sub], kernel=binding.SVM_POLY2)
same_poly = bool(gp["rho"] == rp["rho"] and np.array_equal(gp["alpha"], rp["model"][2]) and np.array_equal(gp["sv"], rp["model"][1]))
dec = desc.astype(np.float64) @ got["w"].astype(np.float64) - got["rho"]
acc = float((np.where(dec > 0, -1, 1) == np.where(labels > 0, 1, -1)).mean())
print(json.dumps({"clouds": K, "instances": n, "positives": int((labels > 0).sum()), "search_s_total": t_search,
                  "gpu_train_s_images_to_model": gpu_s, "solver_steps": got["iterations"], "support_vectors": got["n_sv"],
                  "kernel_rows_computed": got["rows_computed"], "kernel_rows_from_cache": got["rows_reused"],
                  "gpu_ms_per_solver_step": step_ms,
                  "long_solve": {"max_iter": 20000, "steps": long_run["iterations"], "gpu_s": long_s, "rows_computed": long_run["rows_computed"], "rows_from_cache": long_run["rows_reused"]}, "gpu_hog_s_host_to_host": hog_s,
                  "algorithmic_GB_per_step": n * 3528 * 4 / 1e9, "achieved_GBps_per_step": n * 3528 * 4 / 1e9 / (step_ms * 1e-3),
                  "training_set_accuracy": acc,
                  "quadratic_kernel": {"gpu_train_s": poly_s, "support_vectors": int(len(poly["alpha"])),
                                       "steps": poly["iterations"], "classify_last_cloud_ms": poly_cls_ms,
                                       "hypotheses_classified": int(len(keep)), "bit_exact_vs_oracle_on_that_prefix": same_poly},
                  "linear_classify_last_cloud_ms": lin_cls_ms,
                  "cpu_oracle": {"instances": int(sub.size), "hog_s_1_thread": cpu_hog_s, "train_s": cpu_s,
                                 "threads": CPU_T, "steps": ref["iterations"]},
                  "bit_exact_vs_oracle_on_that_prefix": same}))
