#!/usr/bin/env python
"""bench.py -- grasp hypotheses/sec of the MI355X-native HandSearch::findHands path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config C2|C3|C4] [--normals det|rand50] [--shard samples|clouds] [--dist]

`--gpus N` with N > 1 and no launcher around it starts its own N ranks (one process per GPU: python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...); under a launcher WORLD_SIZE must equal N.  `--dist` runs the
sharded entry points on a communicator of one rank at N = 1 (RCCL with one rank).  `--launch-only` lets the ranks meet over
gloo and report who came (CPU test of the launcher).

One "step" = one pass of the hot path over one cloud whose points and sample indices are already resident in HBM:
uniform-grid build (the reference's kd-tree build, hand_search.cpp:10-11) -> Taubin moments / eigen / frame
(findQuadrics) -> hand sweep (findHands) -> compaction [-> HOG + linear SVM for C3].

N = 1 runs BASELINE config C2 (two-view 300k-point cloud, 2000 samples; quadric fit + hand sweep) -- the configuration the
metric is quoted on -- on the TILTED variant of the synthetic scene (agile_grasp_amd/synthetic.py; config.workload says so);
the axis-aligned variant of SURVEY 8d, the production normals mode, configs C3 and C4, the batch of eight clouds in one
context and the host-buffer entry points (Python binding and plain C++) ride along as extra keys of the same line.

N > 1, default (--shard clouds) = BASELINE config C5, "batch of 300k-point clouds, samples sharded across the GPUs": the
sample list of the batch is sharded in cloud order, i.e. GPU g owns cloud g of the batch (seeds 10 + g; the same size and
make as C2) and searches all 2000 of its samples; the ranks' hypothesis lists are exchanged by ONE RCCL all-gather per step
issued by the library itself (agh_find_hands_sharded_device, C++ -> ncclAllGather on the search's stream over xGMI).
Per-GPU work is fixed: "scaling": "weak"; value = hypotheses of the merged list per second.
--shard samples: ONE cloud (C2 / C4), the SAME sample list whatever N is, sharded over the GPUs (rank g searches samples
[g S/N, (g+1) S/N)); every rank builds the search grid of the cloud; the same all-gather.  Total work is fixed: "strong".
It pays only when the sample count warrants it (2000 samples leave 250 work-groups per GPU at N = 8: DESIGN.md section 6).
A default N > 1 run reports both: the C5 line, and c2_sample_sharded / c4_sample_sharded as extra keys.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench_extras import (HBM_ACHIEVABLE_GBS, HBM_PEAK_GBS, batched_throughput, c5_batch_sharded_secondary,  # noqa: E402
                          cloud_per_gpu_secondary, host_api_c_extra, host_api_extra, pipeline_extra, sample_sharded_secondary, settle,
                          single_cloud_extra, sq_issue_figures, static_traffic, taubin_stage_rooflines, timed_intervals, two_streams_extra)


def cpu_baseline(sc, n_sub: int, normals_mode: int, classify: bool, svm):
    """The oracle (a CPU port of the reference's OpenMP path: static schedule over the samples, hand_search.cpp:77-79,
    135-137) timed on this box's host cores on the same cloud.  The call includes what the reference's call includes (the
    search-structure build).  The port does not scale to every core of a large host (fork/join and allocator contention
    over a few thousand samples), so a few thread counts are tried first; the reported value is the MEDIAN of five runs at
    the best count, with the 4-thread (the reference's launch files: num_threads 4) and 1-thread figures beside it."""
    from oracle import oracle_py as O

    cores = os.cpu_count() or 1
    sub = sc.samples[:n_sub]

    def once(threads):
        p = O.default_params(sc.cam_origins, normals_mode=normals_mode, num_threads=threads)
        t0 = time.perf_counter()
        r = O.find_hands(p, sc.xyz, sc.cam, sub, want_images=classify)
        if classify:
            O.classify(r["images"], svm[0], svm[1], num_threads=threads)
        return time.perf_counter() - t0, len(r["hyps"])

    once(min(cores, 16))  # warm-up (page-in, OpenMP pool)
    scan = {t: once(t)[0] for t in sorted({min(cores, t) for t in (8, 16, 32, 64, cores)})}
    best = min(scan, key=scan.get)
    runs = [once(best) for _ in range(5)]
    dt = statistics.median(r[0] for r in runs)
    n_hyp = runs[0][1]
    t4 = statistics.median(once(min(4, cores))[0] for _ in range(3))
    t1 = once(1)[0]
    return {"value": n_hyp / dt, "unit": "hypotheses/s", "cores": best, "kind": "port",
            "sample": f"{'all' if n_sub == sc.samples.size else 'first ' + str(n_sub) + ' of the'} {sc.samples.size} samples of the "
                      f"same cloud, {'rand50' if normals_mode else 'deterministic'} normals; median of 5 runs at the best of "
                      f"OpenMP x{'/'.join(str(t) for t in scan)} (x{best}: {dt:.3f} s)",
            "samples_per_s": n_sub / dt,
            "threads_4": {"value": n_hyp / t4, "seconds": t4}, "threads_1": {"value": n_hyp / t1, "seconds": t1},
            "host_cores": cores}


def free_port() -> int:
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` with no launcher around it: start the N ranks -- one process per GPU, the way the contract's
    own command does (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...) -- with
    this command line, and hand their output (rank 0's JSON line) and exit code through."""
    import subprocess

    if not args.launch_only:
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} asked for, {have} visible on this node", file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def launch_only_report(args, rank: int, world: int) -> None:
    """--launch-only: the ranks the launcher started meet over gloo and rank 0 prints who came (one JSON line).  What the
    CPU test of the launcher drives: no GPU, no library, nothing measured."""
    import socket

    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_PORT=str(free_port()))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = [None] * world
    dist.all_gather_object(seen, {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "pid": os.getpid(),
                                  "host": socket.gethostname(), "gpus_arg": args.gpus})
    dist.barrier()
    if rank == 0:
        print(json.dumps({"launch_only": True, "n_gpus": world, "ranks": seen,
                          "launcher": "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "external/none"}))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C2", choices=["C2", "C3", "C4", "small"])
    ap.add_argument("--normals", default="det", choices=["det", "rand50"])
    ap.add_argument("--shard", default="clouds", choices=["samples", "clouds"],
                    help="N > 1: shard one cloud's samples over the GPUs (default) or give every GPU its own cloud")
    ap.add_argument("--cpu-samples", type=int, default=1 << 30,
                    help="samples of the cloud the CPU baseline is timed on (default: all of them)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--event-level", type=int, default=3, choices=[2, 3],
                    help="HIP events around k_hand_sweep in the timed region: 2 = every launch, 3 = every fourth")
    ap.add_argument("--no-events", action="store_true", help="do not time kernels with HIP events (for rocprofv3 runs)")
    ap.add_argument("--batch-clouds", type=int, default=8,
                    help="N = 1: also time a batch of this many C5 clouds in one context (extra key 'batched'); 0 = skip")
    ap.add_argument("--no-extras", action="store_true", help="N = 1: skip the extra keys untilted / rand50 / host_api")
    ap.add_argument("--spin-seconds", type=float, default=0.5,
                    help="untimed run of the same step before the warm-up steps, so that the clocks are at their steady state")
    ap.add_argument("--dist", action="store_true",
                    help="N = 1: run the sharded entry points on a communicator of ONE rank (RCCL with one rank), the same code "
                         "path as N > 1")
    ap.add_argument("--launch-only", action="store_true",
                    help="start the N ranks, let them meet (gloo, no GPU needed) and report who came; measures nothing")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if not launched and args.gpus > 1:
        sys.exit(launch_ranks(args))  # start the N ranks (torch.distributed.run) and pass their line and exit code through
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if launched and world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; the two must agree "
                 f"(python bench.py --gpus N starts its own N ranks when no launcher did)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_only:
        return launch_only_report(args, rank, world)
    if torch.cuda.device_count() <= local_rank:
        print(f"bench.py: rank {rank} needs HIP device {local_rank}, {torch.cuda.device_count()} visible -- there is no CPU path to "
              f"fall back to (--gpus {args.gpus})", file=sys.stderr)
        sys.exit(2)
    distributed = world > 1 or args.dist
    if distributed and not launched:  # --dist at N = 1 without a launcher: the rendezvous of one rank, in this process
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
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
    by_cloud = distributed and args.shard == "clouds"
    if by_cloud:  # rank r owns cloud r of the C5 batch (seeds 10 + r), same size as C2 (or a C4-sized one)
        sc = synthetic.config(f"C5_{rank}") if base == "C2" else \
            synthetic.make_scene(1_000_000, 8000, seed=40 + rank, two_view=True, n_objects=48, name=f"C4_{rank}")
    else:  # one cloud: the whole of it on one GPU, or its samples sharded over the GPUs
        sc = synthetic.config(base)
    normals_mode = binding.NORMALS_RAND50 if args.normals == "rand50" else binding.NORMALS_DETERMINISTIC
    ctx = binding.Context(sc.cam_origins, normals_mode=normals_mode, device=dev.index, profile=0 if args.no_events else args.event_level)
    svm = None
    if classify:
        z = np.load(os.path.join(ROOT, "tests", "golden", "svm_weights.npz"))
        svm = (z["w"], float(z["rho"]))
        ctx.load_svm(*svm)

    # ---- the communicator of the sharded search: created by the library (C++ -> RCCL), id handed over by torch ----
    exchange = None
    if distributed:
        # every rank takes part in the same sequence of collectives whatever fails locally: the id (with a validity byte) is
        # broadcast in any case, and the outcome of comm_init is agreed on by an all-reduce
        idt = torch.zeros(129, dtype=torch.uint8, device=dev)
        err = None
        if rank == 0:
            try:
                idt[:128] = torch.frombuffer(bytearray(binding.comm_unique_id()), dtype=torch.uint8).to(dev)
                idt[128] = 1
            except Exception as e:
                err = f"agh_comm_unique_id: {e}"
        dist.broadcast(idt, 0)
        mine = 0
        if int(idt[128].item()) == 1:
            try:
                ctx.comm_init(rank, world, bytes(idt[:128].cpu().numpy().tobytes()))
                mine = 1
            except Exception as e:
                err = f"agh_comm_init: {e}"
        ok = torch.tensor([mine], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            exchange = "library: ncclAllGather issued from C++ on the search's stream"
        else:
            if mine:
                ctx.comm_destroy()
            exchange = f"torch.distributed all_gather_into_tensor (the library's communicator failed: {err or 'on another rank'})"
    lib_comm = exchange is not None and exchange.startswith("library")

    S = sc.samples.size
    xyz_t = torch.from_numpy(sc.xyz).to(dev)
    cam_t = torch.from_numpy(sc.cam).to(dev)
    s_t = torch.from_numpy(sc.samples).to(dev)
    out_t = torch.zeros(8 * S * 160 * (world if by_cloud else 1), dtype=torch.uint8, device=dev)
    nout_t = torch.zeros(1, dtype=torch.int64, device=dev)
    keep_t = torch.zeros(8 * S * (world if by_cloud else 1), dtype=torch.uint8, device=dev)
    sl = sharding.shard_slice(S, rank, world) if distributed and not by_cloud else slice(0, S)
    # fall-back exchange (torch): the same segments the library would gather
    seg = [sharding.segment_records(S if not by_cloud else S * world, world)]
    fb_local = fb_gather = None
    if distributed and not lib_comm:
        fb_local = torch.zeros(sharding.segment_bytes(8 * S), dtype=torch.uint8, device=dev)
        fb_gather = torch.zeros(world * fb_local.numel(), dtype=torch.uint8, device=dev)
    # An explicit (non-null) stream: work on the legacy null stream serialises against every other blocking stream
    # (the context's own one included), which costs ~10 us per launch as soon as any torch op is interleaved.
    torch.cuda.synchronize()
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    # cloud-per-GPU through the library: the "list" is the concatenation of the ranks' sample lists, rank r's slice of it
    # is its own samples -- the same sharded call, every rank searching its own cloud
    s_all_t = None
    if by_cloud and lib_comm:
        s_all_t = torch.zeros(world * S, dtype=torch.int32, device=dev)
        s_all_t[rank * S:(rank + 1) * S] = s_t

    def step():
        ctx.set_cloud_torch(xyz_t, cam_t, stream=stream)          # grid build (kd-tree build in the reference)
        if not distributed:
            ctx.find_hands_torch(s_t, out_t, nout_t, stream=stream)   # findQuadrics + findHands + concatenation
            if classify:
                ctx.classify_torch(keep_t, stream=stream)             # Learning::classify
        elif lib_comm:
            ctx.find_hands_sharded_torch(s_all_t if by_cloud else s_t, out_t, nout_t, stream=stream)
            if classify:
                ctx.classify_sharded_torch(keep_t, stream=stream)
        else:
            n_loc = fb_local[:8].view(torch.int64)
            ctx.find_hands_torch(s_t[sl], fb_local[160:], n_loc, stream=stream)
            if classify:
                ctx.classify_torch(keep_t, stream=stream)
            nb = sharding.segment_bytes(seg[0])
            sharding.all_gather_records(fb_local[:nb], fb_gather[:world * nb])

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    settle(ctx, step, fence)
    gc.collect()  # (here, before the clock spin -- see the timed region)
    # Clocks: a cold GPU runs the first milliseconds below its sustained clock and these kernels are issue bound, so the
    # same step runs untimed for a moment first (the W warm-up steps and the K timed steps follow unchanged).
    if distributed:
        # every step is a collective: all ranks must run the SAME number of them, so the spin is a step count here, not a clock
        # (a clock would let one rank run a batch more than another and leave it waiting in an all-gather nobody joins)
        for _ in range(max(1, int(args.spin_seconds * 4000 / 20))):
            for _ in range(20):
                step()
            torch.cuda.synchronize()
    else:
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < args.spin_seconds:
            for _ in range(20):
                step()
            torch.cuda.synchronize()
    fence()
    while True:
        for _ in range(args.warmup):
            step()
        fence()
        ctx.timing()  # drop the warm-up kernel times
        # (the timed region is 20 steps = 4 ms: a collector pause is a measurable fraction of it.  No gc.collect() anywhere between
        # the clock spin and the timed steps: tens of milliseconds of host work let the GPU's clocks fall -- with one in front of
        # the timed steps the interval read 0.215 ms, with one in front of the warm-up steps 0.206, the intervals after it 0.199)
        gc.disable()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        gc.enable()
        if not distributed:
            break
        if lib_comm:
            try:
                ctx.synchronize()  # raises AGH_ERR_CAPACITY if a rank overflowed its segment; the context then uses full ones
                break
            except binding.AghError as e:
                if e.code != binding.AGH_ERR_RETRY:
                    raise
                continue
        nb = sharding.segment_bytes(seg[0])
        counts = fb_gather[:world * nb].view(world, nb)[:, :8].contiguous().view(torch.int64)
        if int(counts.max().item()) <= seg[0]:
            break
        seg[0] = 8 * S
    ctx.synchronize()  # raises if any neighbourhood overflowed the kernels' capacity
    # the spread of the figure above: five more intervals of the same K steps (the contract's `value` stays the ONE interval above)
    spread = timed_intervals(step, fence, args.steps, 5) if not distributed else None
    # HIP events on the launch stream bracket k_hand_sweep inside the timed region -- on every fourth step (profile level 3):
    # an event record between two dependent kernels costs ~3 us, two per step 6.3 us of a 0.24 ms step (measured: 0.2436
    # against 0.2373 ms), and bracketing all six phases ~35 us, so the full breakdown comes from a second, untimed pass.
    kern, kern_n = ctx.timing(counts=True)
    if not args.no_events:
        ctx.set_profile(1)
        for _ in range(args.steps):
            step()
        fence()
        kern_all = ctx.timing()
        ctx.set_profile(args.event_level)
    else:
        kern_all = {}
    if not distributed or lib_comm:
        n_hyp = int(nout_t.item())       # the complete (merged) list
        total_hyp = float(n_hyp)
    else:
        nb = sharding.segment_bytes(seg[0])
        counts = fb_gather[:world * nb].view(world, nb)[:, :8].contiguous().view(torch.int64)
        n_hyp = int(counts[rank].item())
        total_hyp = float(counts.sum().item())
    n_kept = int(keep_t[:n_hyp].sum().item()) if classify else None
    nt, nh = ctx.neighbor_counts()       # of this rank's own samples

    tvals = torch.tensor([dt], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(tvals, op=dist.ReduceOp.MAX)
        dt = float(tvals[0].item())

    # what the communicator itself says it is, and what its hypothesis all-gather moved (HIP events of the untimed pass above)
    comm_info = None
    if distributed:
        comm_info = {"torch_world_size": world, "launcher": "torch.distributed.run" if launched else "in-process (--dist)"}
        if lib_comm:
            seg_b, n_r, via = ctx.comm_last_exchange()
            ag_ms = kern_all.get("shard_allgather")
            comm_info.update({"library_comm_ranks": n_r, "library_comm_rank0": ctx.comm_rank()[0], "via_rccl": via,
                              "rccl_origin": binding.comm_rccl_origin(), "allgather_bytes_per_rank": seg_b,
                              "allgather_bytes_total": seg_b * n_r,
                              "allgather_us": (ag_ms / args.steps * 1e3) if ag_ms is not None else None,
                              "merge_us": (kern_all["shard_merge"] / args.steps * 1e3) if "shard_merge" in kern_all else None})
            if n_r != world:
                raise RuntimeError(f"the library's communicator has {n_r} ranks, torch.distributed {world}")

    secondary, secondary_c4, secondary_c5, hung = None, None, None, False
    if distributed and lib_comm and base == "C2" and not classify:
        # never at the price of the headline line: the extra measurements run on a watched thread
        import threading

        box = {}

        def work():
            try:
                torch.cuda.set_device(dev)
                torch.cuda.set_stream(tstream)
                if by_cloud:  # the other way of using the node: ONE cloud, its samples sharded (strong scaling)
                    box["res"] = sample_sharded_secondary(args, dev, stream, rank, world, normals_mode, "C2")
                else:
                    box["res"] = cloud_per_gpu_secondary(args, dev, stream, rank, world, normals_mode)
            except Exception as e:
                box["res"] = {"error": str(e)}
            if "error" not in box["res"] and not args.no_extras:
                try:
                    box["c4"] = sample_sharded_secondary(args, dev, stream, rank, world, normals_mode, "C4")
                except Exception as e:
                    box["c4"] = {"error": str(e)}
                if 8 % world == 0:  # (the same decision on every rank: the call is a collective)
                    try:
                        box["c5"] = c5_batch_sharded_secondary(args, dev, stream, rank, world, normals_mode)
                    except Exception as e:
                        box["c5"] = {"error": str(e)}

        th = threading.Thread(target=work, daemon=True)
        th.start()
        th.join(timeout=600)
        hung = th.is_alive()
        secondary = {"error": "timed out"} if hung and "res" not in box else box.get("res")
        secondary_c4 = {"error": "timed out"} if hung else box.get("c4")
        secondary_c5 = {"error": "timed out"} if hung else box.get("c5")

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_hyp * args.steps / dt
        # ---- roofline of the kernel that moves the bytes (B_alg): the hand sweep reads 16 B per
        # r = 0.08 neighbour (12 B xyz + 4 B id/cam) and writes 160 B + a 1000 B image per hypothesis slot kept.
        k_ms = {k: v / args.steps for k, v in kern_all.items()}
        sweep_timed = kern_n.get("hand_sweep", 0)
        if sweep_timed:
            k_ms["hand_sweep"] = kern["hand_sweep"] / sweep_timed   # the one measured inside the timed region
        n_local_hyp = n_hyp if not distributed else (int(ctx.epoch()[1]) if ctx.epoch()[1] >= 0 else n_hyp // world)
        n_local_samples = sl.stop - sl.start
        sweep_bytes = 16.0 * float(nh.sum()) + 200.0 * n_local_samples + (160.0 + 1000.0) * n_local_hyp
        sweep_s = k_ms.get("hand_sweep", 0.0) * 1e-3
        achieved = sweep_bytes / sweep_s / 1e9 if sweep_s > 0 else 0.0
        traffic, traffic_src = None, None
        if not distributed:
            tmap, traffic_src = static_traffic(f"{args.config}:{args.normals}")
            hs = [b for k, b in tmap.items() if k.startswith("k_hand_sweep")]
            traffic = hs[0] if hs else None
        issue = sq_issue_figures("k_hand_sweep") if not distributed else None
        stage_roof = taubin_stage_rooflines(k_ms, float(nt.sum()), n_local_samples, args, distributed)
        # whole-path algorithmic bytes (B_alg of BASELINE.md section 4), of this rank's share
        b_alg = 16.0 * sc.n + 16.0 * float(nt.sum() + nh.sum()) + 160.0 * n_local_hyp + (14112 if classify else 0)
        # --shard clouds (the default; N = 1 is its first member: one cloud on one GPU): one more cloud per GPU -- per-GPU work
        # fixed; --shard samples: the SAME cloud and sample list whatever N is -- total work fixed
        scaling = "weak" if (by_cloud or not distributed) else "strong"
        par = "single GPU" if not distributed else (
            f"cloud-per-gpu x{world}: every rank searches its own cloud, one all-gather of the lists" if by_cloud else
            f"sample-sharded x{world}: rank g searches samples [g S/N, (g+1) S/N) of the same cloud, one all-gather of the lists")
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
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: two-view {sc.n}-point tabletop cloud, TILTED scene variant (the scene and both "
                                   f"camera origins rotated 33 / -19 degrees about the scene pivot before the 3 mm voxel snap, as in a camera "
                                   f"optical frame: no exactly planar neighbourhoods; the axis-aligned variant of SURVEY 8d is the extra key "
                                   f"'untilted'), {S} samples{' per GPU' if by_cloud else ''}, "
                                   f"{'rand50' if normals_mode else 'deterministic'} normals{', + HOG/linear SVM' if classify else ''}",
                       "scene_variant": "tilted", "points": sc.n, "samples": S, "hypotheses": int(total_hyp), "parallelism": par},
            "samples_per_s": S * (world if by_cloud else 1) * args.steps / dt,
            # "bound": what the counters say limits the kernel -- its SIMDs' instruction issue, not HBM (traffic is BELOW the
            # algorithmic bytes: the cloud is cache resident).  achieved / peak / frac stay the contract's HBM figures; the
            # issue-slot figures beside them show what an instruction cut can still buy.
            "roofline": {"bound": "hbm", "limited_by": "wave occupancy and VALU time, not bandwidth: see `issue` (the SIMD's vector ALU "
                         "busy fraction and the mean resident waves per SIMD of the committed SQ pass) -- the cloud is cache resident and "
                         "the kernel's HBM-side traffic is BELOW its algorithmic bytes", "kernel": "k_hand_sweep", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "issue": issue,
                         "algorithmic_bytes_per_launch": sweep_bytes, "launch_ms": k_ms.get("hand_sweep", 0.0),
                         "launches_timed": sweep_timed, "launches_timed_note": "HIP events around every fourth k_hand_sweep "
                         "launch of the timed region (two event records per step cost 6 us of a 0.24 ms step)",
                         "launch_samples": n_local_samples},
            "roofline_stages": stage_roof,
            "ms_per_step_spread": spread,
            "kernel_ms_per_step": k_ms,
            "path_algorithmic_bytes": b_alg,
            "path_GBps": b_alg / (dt / args.steps) / 1e9,
        }
        if secondary is not None:
            res["c2_sample_sharded" if by_cloud else "cloud_per_gpu"] = secondary
        if secondary_c4 is not None:
            res["c4_sample_sharded"] = secondary_c4
        if secondary_c5 is not None:
            res["c5_batch_sharded"] = secondary_c5
        if distributed and base == "C2":
            res["config"]["note"] = (
                "BASELINE config C5: the batch of 300k-point clouds with its samples sharded over the GPUs in cloud order -- GPU g "
                "searches cloud g (seed 10 + g; N = 1: the C2 cloud), per-GPU work fixed; ONE cloud's samples sharded is reported "
                "beside it: c4_sample_sharded (8000 samples: strong scaling pays) and c2_sample_sharded (2000 samples do not "
                "warrant it: every rank still builds the whole grid and runs the same latency chains, DESIGN.md section 6)"
                if by_cloud else
                "C2's 2000 samples do not warrant sharding (every rank still builds the whole grid and runs the same latency "
                "chains: DESIGN.md section 6); the node's GPUs are used by cloud_per_gpu (weak) and c4_sample_sharded (strong), "
                "reported beside this line")
        if not distributed:
            res["config"]["scaling_note"] = ("N = 1 member of the default --gpus N series, which is weak: every GPU searches one cloud "
                                             "of this size (BASELINE config C5, the batch's samples sharded in cloud order)")
        if distributed:
            res["config"]["exchange"] = exchange
            res["config"]["segment_records"] = seg[0] if not lib_comm else None
            res["comm"] = comm_info
        if classify:
            res["config"]["svm_kept"] = n_kept
        if not distributed and args.batch_clouds > 1 and base == "C2":
            res["batched"] = batched_throughput(args, dev, stream, normals_mode, classify, svm)
        if not distributed and base == "C2" and not classify and not args.no_extras:
            # SURVEY 8d's scene to the letter (axis-aligned: 60 % of the samples are exactly planar neighbourhoods, which the
            # reference does not drop and, since round 3, neither does this path), the reference's production normals mode
            # (HandSearch hard-wires it, hand_search.h:84), and the host-buffer entry points
            res["untilted"] = single_cloud_extra(args, dev, stream, "C2u", normals_mode,
                                                 "C2u: the same scene axis-aligned (SURVEY 8d literally), deterministic normals")
            if args.normals == "det":
                res["rand50"] = single_cloud_extra(args, dev, stream, "C2", binding.NORMALS_RAND50,
                                                   "C2, the reference's production mode: 50 x rand() % n normals per sample")
                res["untilted_rand50"] = single_cloud_extra(args, dev, stream, "C2u", binding.NORMALS_RAND50,
                                                            "C2u (axis-aligned) in the reference's production mode")
            res["two_streams"] = two_streams_extra(args, dev, "C2", normals_mode)
            res["three_streams"] = two_streams_extra(args, dev, "C2", normals_mode, n_lanes=3)
            res["host_api"] = host_api_extra(args, dev, sc, normals_mode)
            res["host_api_c"] = host_api_c_extra(sc, max(20, args.steps))
            res["pipeline"] = pipeline_extra(max(20, args.steps))
            # the other single-GPU BASELINE configs, on the same clock as the headline
            z = np.load(os.path.join(ROOT, "tests", "golden", "svm_weights.npz"))
            res["c3"] = single_cloud_extra(args, dev, stream, "C2", normals_mode,
                                           "C3: the C2 cloud (tilted variant) + HOG descriptor + linear SVM "
                                           "(svm_032015_linear_20_20_same) on every hypothesis", svm=(z["w"], float(z["rho"])))
            res["c4"] = single_cloud_extra(args, dev, stream, "C4", normals_mode,
                                           "C4: dense two-view 1000000-point cloud (tilted variant), 8000 samples",
                                           steps=max(5, args.steps // 4))
        if not args.no_cpu_baseline and not distributed:
            res["cpu_baseline"] = cpu_baseline(sc, min(args.cpu_samples, S), normals_mode, classify, svm)
        elif not args.no_cpu_baseline:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if hung:  # a rank is stuck in the extra measurement's collective: the line is out, leave without the tidy shutdown
        sys.stdout.flush()
        os._exit(0)
    if distributed:
        if lib_comm:
            ctx.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
