#!/usr/bin/env python
"""bench.py -- grasp hypotheses/sec of the MI355X-native HandSearch::findHands path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config C2|C3|C4] [--normals det|rand50]

One "step" = one pass of the hot path over one cloud whose points and sample indices are already resident in
HBM: uniform-grid build (the reference's kd-tree build, hand_search.cpp:10-11) -> Taubin moments / eigen / frame
(findQuadrics) -> hand sweep (findHands) -> compaction [-> HOG + linear SVM for C3].  N = 1 runs BASELINE config C2
(two-view 300k-point cloud, 2000 samples; quadric fit + hand sweep) -- the configuration the metric is quoted on.
For N > 1 every rank owns one cloud of the C5 batch (seeds 10..), runs the same per-GPU work, and the fixed-slot
hypothesis records are all-gathered over RCCL/xGMI (one collective per step): weak scaling, value = hypotheses
of all ranks per second.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(sc, n_sub: int, normals_mode: int, classify: bool, svm):
    """The oracle (a CPU port of the reference's OpenMP path: static schedule over the samples, hand_search.cpp:77-79,
    135-137) timed on this box's host cores on the same cloud.  The call includes what the reference's call includes (the
    search-structure build); the port does not scale to every core of a large host (fork/join and allocator contention
    over a few thousand samples), so a few thread counts are tried and the best one is reported with its count."""
    from oracle import oracle_py as O

    cores = os.cpu_count() or 1
    sub = sc.samples[:n_sub]
    best = None
    for threads in sorted({min(cores, t) for t in (16, 32, 64, cores)}):
        p = O.default_params(sc.cam_origins, normals_mode=normals_mode, num_threads=threads)
        O.find_hands(p, sc.xyz, sc.cam, sub[:8])  # warm-up (page-in, OpenMP pool)
        t0 = time.perf_counter()
        r = O.find_hands(p, sc.xyz, sc.cam, sub, want_images=classify)
        if classify:
            O.classify(r["images"], svm[0], svm[1], num_threads=threads)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, threads, len(r["hyps"]))
    dt, threads, n_hyp = best
    return {"value": n_hyp / dt, "unit": "hypotheses/s", "cores": threads, "kind": "port",
            "sample": f"{'all' if n_sub == sc.samples.size else 'first ' + str(n_sub) + ' of the'} {sc.samples.size} samples of the "
                      f"same cloud, {'rand50' if normals_mode else 'deterministic'} normals, best of OpenMP x16/32/64/{cores}: "
                      f"x{threads}, {dt:.2f} s",
            "samples_per_s": n_sub / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2", choices=["C2", "C3", "C4", "small"])
    ap.add_argument("--normals", default="det", choices=["det", "rand50"])
    ap.add_argument("--cpu-samples", type=int, default=1 << 30,
                    help="samples of the cloud the CPU baseline is timed on (default: all of them)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="do not time kernels with HIP events (for rocprofv3 runs)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("AGH_BENCH_FORCE_DIST") == "1"  # (the override exercises the RCCL path on 1 GPU)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if distributed else 0)

    from agile_grasp_amd import binding, sharding, synthetic

    classify = args.config == "C3"
    base = "C2" if args.config == "C3" else args.config
    # N = 1: the C2 cloud (seed 2).  N > 1: rank r owns cloud r of the C5 batch (seeds 10 + r), same size.
    sc = synthetic.config(base) if not distributed else synthetic.config(f"C5_{rank}") if base == "C2" else \
        synthetic.make_scene(1_000_000, 8000, seed=40 + rank, two_view=True, n_objects=48, name=f"C4_{rank}")
    normals_mode = binding.NORMALS_RAND50 if args.normals == "rand50" else binding.NORMALS_DETERMINISTIC
    ctx = binding.Context(sc.cam_origins, normals_mode=normals_mode, device=dev.index, profile=0 if args.no_events else 2)
    svm = None
    if classify:
        z = np.load(os.path.join(ROOT, "tests", "golden", "svm_weights.npz"))
        svm = (z["w"], float(z["rho"]))
        ctx.load_svm(*svm)

    S = sc.samples.size
    xyz_t = torch.from_numpy(sc.xyz).to(dev)
    cam_t = torch.from_numpy(sc.cam).to(dev)
    s_t = torch.from_numpy(sc.samples).to(dev)
    # exchange buffer = [160-byte header whose first int64 is the record count | 8*S records of 160 B]
    buf_t = torch.zeros(sharding.buffer_bytes(S), dtype=torch.uint8, device=dev)
    nout_t = buf_t[:8].view(torch.int64)
    out_t = buf_t[160:]
    keep_t = torch.zeros(8 * S, dtype=torch.uint8, device=dev)
    # The exchange sends a PREFIX of the buffer: header + S record slots (one per sample; a cloud yields ~0.4
    # hypotheses per sample), 320 KB instead of 2.5 MB per rank -- xGMI all-gathers of this size are latency bound.
    # The header carries the true count, so a rank that produced more is detected (checked after the timed region)
    # and the run is repeated with the full 8*S slots.
    xch_records = [min(S, 8 * S)]
    gather_full = torch.zeros(world * buf_t.numel(), dtype=torch.uint8, device=dev) if distributed else None
    # An explicit (non-null) stream: work on the legacy null stream serialises against every other blocking stream
    # (the context's own one included), which costs ~10 us per launch as soon as any torch op is interleaved.
    torch.cuda.synchronize()
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream

    def step():
        ctx.set_cloud_torch(xyz_t, cam_t, stream=stream)          # grid build (kd-tree build in the reference)
        ctx.find_hands_torch(s_t, out_t, nout_t, stream=stream)   # findQuadrics + findHands + concatenation
        if classify:
            ctx.classify_torch(keep_t, stream=stream)             # Learning::classify
        if distributed:
            nb = sharding.buffer_bytes_records(xch_records[0])
            sharding.all_gather_records(buf_t[:nb], gather_full[:world * nb])   # ONE RCCL all-gather per step

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    while True:
        for _ in range(args.warmup):
            step()
        fence()
        ctx.timing()  # drop the warm-up kernel times
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        if not distributed:
            break
        nb = sharding.buffer_bytes_records(xch_records[0])
        counts = gather_full[:world * nb].view(world, nb)[:, :8].contiguous().view(torch.int64)
        if int(counts.max().item()) <= xch_records[0]:
            break
        xch_records[0] = 8 * S  # some rank overflowed the compact slots: measure again with the full exchange
    ctx.synchronize()  # raises if any neighbourhood overflowed the kernels' capacity
    # HIP events on the launch stream bracket k_hand_sweep inside the timed region (2 events per step; bracketing all
    # six phases costs ~35 us per step, so the full breakdown comes from a second, untimed pass of K steps).
    kern = ctx.timing()
    if not args.no_events:
        ctx.set_profile(1)
        for _ in range(args.steps):
            step()
        fence()
        kern_all = ctx.timing()
        ctx.set_profile(2)
    else:
        kern_all = {}
    n_hyp = int(nout_t.item())
    n_kept = int(keep_t[:n_hyp].sum().item()) if classify else None
    nt, nh = ctx.neighbor_counts()

    tvals = torch.tensor([dt, float(n_hyp)], dtype=torch.float64, device=dev)
    if distributed:
        tmax = tvals.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tvals.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0].item())
        total_hyp = float(tsum[1].item())
    else:
        total_hyp = float(n_hyp)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_hyp * args.steps / dt
        # ---- roofline of the kernel that moves the bytes (B_alg): the hand sweep reads 16 B per
        # r = 0.08 neighbour (12 B xyz + 4 B id/cam) and writes 160 B + a 1000 B image per hypothesis slot kept.
        k_ms = {k: v / args.steps for k, v in kern_all.items()}
        k_ms["hand_sweep"] = kern.get("hand_sweep", 0.0) / args.steps   # the one measured inside the timed region
        sweep_bytes = 16.0 * float(nh.sum()) + 200.0 * S + (160.0 + 1000.0) * n_hyp
        sweep_s = k_ms.get("hand_sweep", 0.0) * 1e-3
        achieved = sweep_bytes / sweep_s / 1e9 if sweep_s > 0 else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(f"{args.config}:{args.normals}", {}).get("hand_sweep_bytes_per_launch")
            except Exception:
                traffic = None
        # whole-path algorithmic bytes (B_alg of BASELINE.md section 4)
        b_alg = 16.0 * sc.n + 16.0 * float(nt.sum() + nh.sum()) + 160.0 * n_hyp + (24.0 * 0 + 14112 if classify else 0)
        # fp64 VALU work of the (n_i . n_j)^6 stage, the time-dominant kernel in deterministic mode
        ks = np.where((normals_mode == 1) & (nt > 50), 50, nt).astype(np.float64)
        frame_flops = float((ks * ks * 9.0).sum())
        frame_s = k_ms.get("taubin_frame", 0.0) * 1e-3
        res = {
            "metric": "grasp hypotheses/sec on 300k-pt cloud @2000 samples" if base == "C2" else
                      "grasp hypotheses/sec (config %s)" % args.config,
            "value": value,
            "unit": "hypotheses/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: two-view {sc.n}-point tabletop cloud, {S} samples per GPU, "
                                   f"{'rand50' if normals_mode else 'deterministic'} normals"
                                   f"{', + HOG/linear SVM' if classify else ''}",
                       "points": sc.n, "samples": S, "hypotheses_per_cloud": n_hyp,
                       "parallelism": f"cloud-per-gpu x{world}" + (" + all-gather" if distributed else "")},
            "samples_per_s": S * world * args.steps / dt,
            "roofline": {"bound": "hbm", "kernel": "k_hand_sweep", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": sweep_bytes, "launch_ms": k_ms.get("hand_sweep", 0.0)},
            "kernel_ms_per_step": k_ms,
            "path_algorithmic_bytes": b_alg,
            "path_GBps": b_alg / (dt / args.steps) / 1e9,
            "taubin_frame_fp64": {"gflops": frame_flops / frame_s / 1e9 if frame_s > 0 else 0.0, "peak_gflops": 78600.0,
                                  "note": "fp64 VALU mul/add of the n x n (n_i.n_j)^6 column sums, 9 flop per pair"},
        }
        if classify:
            res["config"]["svm_kept"] = n_kept
        if not args.no_cpu_baseline and not distributed:
            cb = cpu_baseline(sc, min(args.cpu_samples, S), normals_mode, classify, svm)
            res["cpu_baseline"] = cb
        elif not args.no_cpu_baseline:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
