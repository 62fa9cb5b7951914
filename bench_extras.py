"""bench_extras.py -- everything bench.py measures BESIDE its headline (the extra keys of the N = 1 line, the secondary measurements of
an N > 1 run) and the timing helpers they share: medians of five intervals, the clock spin, the static PMC / SQ figures, the Taubin
stage's rooflines.  bench.py keeps the contract: the launcher, the headline step, the JSON line, and the CPU baseline (the one place
outside tests/ and smoke() that touches oracle/)."""
from __future__ import annotations

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

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0  # the same guide's achievable streaming rate (SURVEY 8d asks for both fractions)


def timed_intervals(step, fence, steps: int, reps: int = 5):
    """`reps` back-to-back intervals of `steps` steps each, every interval bracketed by `fence` (a device synchronisation): the
    spread of ONE measurement.  A figure that is a single interval cannot tell a stall of the box (a clock ramp after seconds of
    host work, a driver query from another process, a page fault of the first use) from a regression (the driver's
    `batched` line, 4.13 ms against 1.20 in every other run); the median of several can, and min / max / first say which it was."""
    ms = []
    for _ in range(reps):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        ms.append((time.perf_counter() - t0) / steps * 1e3)
    return {"median_ms": statistics.median(ms), "min_ms": min(ms), "max_ms": max(ms), "first_ms": ms[0],
            "intervals": reps, "steps_per_interval": steps, "all_ms": [round(x, 5) for x in ms]}


def spin(step, fence, seconds: float):
    """The same step untimed for `seconds`: brings the clocks to their steady state after host-side work (scene generation takes
    seconds per cloud, the GPU idles meanwhile) and takes every first-use allocation out of what follows."""
    t = time.perf_counter()
    while time.perf_counter() - t < seconds:
        for _ in range(10):
            step()
        fence()


def static_traffic(key: str):
    """Per-kernel HBM-side bytes per launch from the newest committed PMC pass of this workload (profiles/rNN_pmc_traffic.json:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 corrections applied by scripts/pmc_traffic.py).  Static: counters
    cannot be collected inside the timed run.  Returns ({kernel: bytes}, source string)."""
    import hashlib

    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json"):
        tf = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(tf):
            continue
        try:
            blob = open(tf, "rb").read()
            ks = json.loads(blob).get(key, {}).get("kernels")
        except Exception:  # noqa: BLE001
            ks = None
        if ks:
            return ({k: v["hbm_bytes_per_launch"] for k, v in ks.items()},
                    f"static: profiles/{name} sha256 {hashlib.sha256(blob).hexdigest()[:16]} (separate rocprofv3 --pmc FETCH_SIZE / "
                    "WRITE_SIZE passes over this workload, gfx950 corrections applied); not measured in this run")
    return {}, None


def taubin_stage_rooflines(k_ms, sum_nt: float, n_samples: int, args, distributed: bool):
    """The Taubin stage (K1a + K1b + K1c: longer than the sweep) on the roofline like the sweep.  Algorithmic
    bytes per launch, DESIGN.md section 4: K1a k_taubin_moments reads 16 B per r = 0.03 neighbour and writes the sorted list back
    (16 * sum n_t each way); K1c k_taubin_frame reads the list and writes a 200-byte frame per sample; K1b k_taubin_eigen reads 296 B
    and writes 96 B per sample (a latency chain: listed for completeness).  Times: HIP events of the untimed all-phase pass."""
    traffic, src = static_traffic(f"{args.config}:{args.normals}") if not distributed else ({}, None)

    def tr(prefix):
        v = [b for k, b in traffic.items() if k.startswith(prefix)]
        return sum(v) if v else None

    out = []
    for phase, kernel, nbytes, why in (
            ("taubin_moments", "k_taubin_moments", 32.0 * sum_nt,
             "gather latency + the reference's sequential summation order (37 dependent fp64 add chains per sample)"),
            ("taubin_eigen", "k_taubin_eigen", 392.0 * n_samples, "one lane's ~5000-instruction dependent chain per sample"),
            ("taubin_frame", "k_taubin_frame", 16.0 * sum_nt + 200.0 * n_samples, "fp64 VALU + LDS (n_t^2 pow6 terms when exhaustive)")):
        ms = k_ms.get(phase, 0.0)
        if ms <= 0:
            continue
        ach = nbytes / (ms * 1e-3) / 1e9
        out.append({"kernel": kernel, "bound": "hbm", "limited_by": why, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes, "launch_ms": ms, "traffic": tr(kernel),
                    "traffic_source": src})
    return out


def batched_throughput(args, dev, stream, normals_mode, classify, svm):
    """BASELINE config C5's batch on ONE GPU: the clouds with seeds 10.. (each as large as C2, 2000 samples each) laid end to
    end in one context (agh_set_cloud_batch_device), one launch set per step for all of them -- grid builds, Taubin
    stages, hand sweep and compaction each see 8 x 2000 work-groups instead of 2000.  Reported beside the single-cloud
    headline, never instead of it: it is the throughput of a stream of clouds, not the latency of one."""
    from agile_grasp_amd import binding, synthetic

    C = args.batch_clouds
    scs = [synthetic.config(f"C5_{k}") for k in range(C)]
    ctx = binding.Context(scs[0].cam_origins, normals_mode=normals_mode, device=dev.index, profile=0)
    if classify:
        ctx.load_svm(*svm)
    off = np.zeros(C + 1, np.int64)
    off[1:] = np.cumsum([s.n for s in scs])
    xyz_t = torch.from_numpy(np.concatenate([s.xyz for s in scs])).to(dev)
    cam_t = torch.from_numpy(np.concatenate([s.cam for s in scs])).to(dev)
    samples = np.concatenate([s.samples + off[k] for k, s in enumerate(scs)]).astype(np.int32)
    s_t = torch.from_numpy(samples).to(dev)
    S = samples.size
    out_t = torch.zeros(8 * S * 160, dtype=torch.uint8, device=dev)
    nout_t = torch.zeros(1, dtype=torch.int64, device=dev)
    keep_t = torch.zeros(8 * S, dtype=torch.uint8, device=dev)

    def step():
        ctx.set_cloud_batch_torch(xyz_t, cam_t, off, stream=stream)
        ctx.find_hands_torch(s_t, out_t, nout_t, stream=stream)
        if classify:
            ctx.classify_torch(keep_t, stream=stream)

    settle(ctx, step, torch.cuda.synchronize)
    spin(step, torch.cuda.synchronize, min(args.spin_seconds, 0.3))
    for _ in range(max(args.warmup, 3)):
        step()
    steps = max(5, args.steps // 2)
    spread = timed_intervals(step, torch.cuda.synchronize, steps, 5)
    dt = spread["median_ms"] * 1e-3
    ctx.synchronize()
    n_hyp = int(nout_t.item())
    ctx.set_profile(1)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    k_ms = {k: v / steps for k, v in ctx.timing().items()}
    nt, nh = ctx.neighbor_counts()
    sweep_bytes = 16.0 * float(nh.sum()) + 200.0 * S + (160.0 + 1000.0) * n_hyp
    sweep_s = k_ms.get("hand_sweep", 0.0) * 1e-3
    ctx.close()
    return {"workload": f"C5 batch: {C} two-view 300000-point clouds (seeds 10..{9 + C}), 2000 samples each, one context, one launch "
                        "set per step", "clouds": C, "samples": S, "hypotheses": n_hyp, "steps": steps, "ms_per_batch": dt * 1e3,
            "ms_per_batch_spread": spread, "kernel_ms_sum": sum(v for k, v in k_ms.items() if not k.startswith("total")),
            "ms_per_cloud": dt * 1e3 / C, "value": n_hyp / dt, "unit": "hypotheses/s", "kernel_ms_per_batch": k_ms,
            "roofline": {"kernel": "k_hand_sweep", "achieved": sweep_bytes / sweep_s / 1e9 if sweep_s > 0 else 0.0,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (sweep_bytes / sweep_s / 1e9 / HBM_PEAK_GBS) if sweep_s > 0 else 0.0,
                         "algorithmic_bytes_per_launch": sweep_bytes, "launch_ms": k_ms.get("hand_sweep", 0.0),
                         "note": "HIP events of an untimed pass of the same steps"}}


def single_cloud_extra(args, dev, stream, scene_name, normals_mode, label, svm=None, steps=None):
    """One more single-GPU measurement of the same step on another scene / normals mode / BASELINE config (extra keys of the
    N = 1 line): device-resident cloud and samples, `steps` // 2 timed steps after the same settling and warm-up; with `svm`
    the step ends with Learning::classify (config C3)."""
    from agile_grasp_amd import binding, synthetic

    sc = synthetic.config(scene_name)
    ctx = binding.Context(sc.cam_origins, normals_mode=normals_mode, device=dev.index, profile=0)
    if svm is not None:
        ctx.load_svm(*svm)
    xyz_t, cam_t = torch.from_numpy(sc.xyz).to(dev), torch.from_numpy(sc.cam).to(dev)
    s_t = torch.from_numpy(sc.samples).to(dev)
    S = sc.samples.size
    out_t = torch.zeros(8 * S * 160, dtype=torch.uint8, device=dev)
    nout_t = torch.zeros(1, dtype=torch.int64, device=dev)
    keep_t = torch.zeros(8 * S, dtype=torch.uint8, device=dev)

    def step():
        ctx.set_cloud_torch(xyz_t, cam_t, stream=stream)
        ctx.find_hands_torch(s_t, out_t, nout_t, stream=stream)
        if svm is not None:
            ctx.classify_torch(keep_t, stream=stream)

    settle(ctx, step, torch.cuda.synchronize)
    spin(step, torch.cuda.synchronize, min(args.spin_seconds, 0.2))
    for _ in range(max(args.warmup, 5)):
        step()
    steps = steps or max(10, args.steps // 2)
    spread = timed_intervals(step, torch.cuda.synchronize, steps, 5)
    dt = spread["median_ms"] * 1e-3
    ctx.synchronize()
    n_hyp = int(nout_t.item())
    valid = int((ctx.frames()["valid"] != 0).sum())
    kept = int(keep_t[:n_hyp].sum().item()) if svm is not None else None
    k_ms, roof = {}, None
    if not args.no_events:  # per-kernel HIP events of a second, untimed pass
        ctx.set_profile(1)
        ctx.timing()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        k_ms = {k: v / steps for k, v in ctx.timing().items()}
        nt, nh = ctx.neighbor_counts()
        sweep_bytes = 16.0 * float(nh.sum()) + 200.0 * S + (160.0 + 1000.0) * n_hyp
        sweep_s = k_ms.get("hand_sweep", 0.0) * 1e-3
        if sweep_s > 0:
            roof = {"kernel": "k_hand_sweep", "achieved": sweep_bytes / sweep_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": sweep_bytes / sweep_s / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": sweep_bytes,
                    "launch_ms": k_ms["hand_sweep"], "note": "HIP events of an untimed pass of the same steps"}
    ctx.close()
    res = {"workload": label, "points": sc.n, "samples": S, "frames": valid, "hypotheses": n_hyp, "steps": steps,
           "ms_per_step": dt * 1e3, "ms_per_step_spread": spread, "value": n_hyp / dt, "unit": "hypotheses/s",
           "kernel_ms_per_step": k_ms}
    if kept is not None:
        res["svm_kept"] = kept
    if roof is not None:
        res["roofline"] = roof
    return res


def side_by_side(contexts, launch, collect, make_ctx, tries: int = 6, close=lambda lane: lane.close()):
    """Contexts whose chains really run side by side.  HIP maps a process's streams onto a handful of hardware queues in turn, and two
    streams that land on the SAME queue run one after the other -- in a process that has created many streams (this one) that is a
    coin toss per pair.  So: time the last context's chain in flight TOGETHER with the first one's against the two one after the
    other, and while that shows no overlap replace the last context by a new one (whose stream is the next queue's).  `launch(ctx)`
    queues a chain without waiting, `collect(ctx)` waits for it."""
    def both(overlapped):
        t = []
        for _ in range(5):
            t0 = time.perf_counter()
            if overlapped:
                launch(contexts[0]); launch(contexts[-1]); collect(contexts[0]); collect(contexts[-1])
            else:
                launch(contexts[0]); collect(contexts[0]); launch(contexts[-1]); collect(contexts[-1])
            t.append(time.perf_counter() - t0)
        return statistics.median(t)

    for _ in range(tries):
        both(True)
        if both(True) < 0.85 * both(False):
            return True
        close(contexts[-1])
        contexts[-1] = make_ctx()
    return False


def two_streams_extra(args, dev, scene_name, normals_mode, n_lanes: int = 2):
    """The same C2 step on TWO (n_lanes) contexts and as many HIP streams, taking turns: cloud k's chain (grid build ... compaction) on
    one stream while cloud k + 1's runs on the other -- each chain is 2.6 rounds of work-groups per kernel, and the other stream's kernels fill
    the rounds that are not full.  The throughput of a stream of clouds WITHOUT batching them into one launch set (the `batched` key
    is the upper end of that); never the headline: the headline's steps run one after the other on one stream."""
    from agile_grasp_amd import binding, synthetic

    sc = synthetic.config(scene_name)
    xyz_t, cam_t = torch.from_numpy(sc.xyz).to(dev), torch.from_numpy(sc.cam).to(dev)
    s_t = torch.from_numpy(sc.samples).to(dev)
    S = sc.samples.size
    torch.cuda.synchronize()  # (the lanes launch on streams of their own: the tensors above must be there)

    # a lane: a context, a (non-blocking) HIP stream, its output buffers.  (On the contexts' OWN streams -- hipStreamCreate's
    # blocking kind -- two such chains did not overlap at all: 0.195 ms per step; the online chain's do, see pipeline_extra.)
    def make_lane():
        return (binding.Context(sc.cam_origins, normals_mode=normals_mode, device=dev.index, profile=0), torch.cuda.Stream(device=dev),
                torch.zeros(8 * S * 160, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int64, device=dev))

    def launch(lane):
        lane[0].set_cloud_torch(xyz_t, cam_t, stream=lane[1].cuda_stream)
        lane[0].find_hands_torch(s_t, lane[2], lane[3], stream=lane[1].cuda_stream)

    lanes, all_side_by_side = [], True
    for _ in range(n_lanes):
        lanes.append(make_lane())
        torch.cuda.synchronize()
        if len(lanes) > 1:
            all_side_by_side = side_by_side(lanes, launch, lambda lane: lane[1].synchronize(), make_lane,
                                            close=lambda lane: lane[0].close()) and all_side_by_side
    k = [0]

    def step():
        ctx, st, out_t, nout_t = lanes[k[0] % n_lanes]
        k[0] += 1
        ctx.set_cloud_torch(xyz_t, cam_t, stream=st.cuda_stream)
        ctx.find_hands_torch(s_t, out_t, nout_t, stream=st.cuda_stream)

    for ctx, st, _, _ in lanes:
        for _ in range(n_lanes):  # (settle runs two steps: every context gets its own)
            settle(ctx, step, torch.cuda.synchronize)
    spin(step, torch.cuda.synchronize, min(args.spin_seconds, 0.2))
    steps = max(10, args.steps)
    steps += (-steps) % n_lanes
    spread = timed_intervals(step, torch.cuda.synchronize, steps, 5)
    dt = spread["median_ms"] * 1e-3
    n = [int(l[3].item()) for l in lanes]
    for ctx, _, _, _ in lanes:
        ctx.synchronize()
        ctx.close()
    assert len(set(n)) == 1
    return {"workload": f"{scene_name}: the headline's step on {n_lanes} contexts and {n_lanes} streams, taking turns (cloud k + 1's chain runs "
                        "beside cloud k's)", "contexts": n_lanes, "side_by_side": all_side_by_side, "hypotheses": n[0], "steps": steps, "ms_per_step": dt * 1e3, "ms_per_step_spread": spread,
            "value": n[0] / dt, "unit": "hypotheses/s"}


def host_api_extra(args, dev, sc, normals_mode):
    """What a caller of the HOST-buffer entry points pays (agh_set_cloud + agh_find_hands: the C++ adapter's
    HandSearch::findHands, hand_search.h:101-104 takes a host cloud): upload of the cloud, grid build, search, the list written
    to pinned host memory by the concatenation kernel, one synchronisation -- through the Python binding (ctypes + numpy; the
    same calls from C++ are the key host_api_c).  SURVEY 8d asks for both figures; this one is never `value`."""
    from agile_grasp_amd import binding

    ctx = binding.Context(sc.cam_origins, normals_mode=normals_mode, device=dev.index, profile=0)
    for _ in range(3):
        ctx.set_cloud(sc.xyz, sc.cam)
        hyps = ctx.find_hands(sc.samples)
    calls = max(10, args.steps // 2)
    t_set = t_find = 0.0
    per = []
    for _ in range(calls):
        t0 = time.perf_counter()
        ctx.set_cloud(sc.xyz, sc.cam)
        t1 = time.perf_counter()
        hyps = ctx.find_hands(sc.samples)
        t2 = time.perf_counter()
        t_set += t1 - t0
        t_find += t2 - t1
        per.append(t2 - t0)
    ctx.close()
    dt = statistics.median(per)
    return {"what": "agh_set_cloud (H2D of 12 B/point + camera ids, grid build) + agh_find_hands (search; count, flags and 160-byte "
                    "records land in pinned host memory), host numpy buffers in and out, Python binding", "calls": calls,
            "ms_per_call": dt * 1e3, "ms_per_call_min": min(per) * 1e3, "ms_per_call_max": max(per) * 1e3,
            "ms_per_call_mean": sum(per) / calls * 1e3,
            "ms_set_cloud": t_set / calls * 1e3, "ms_find_hands": t_find / calls * 1e3, "value": len(hyps) / dt,
            "unit": "hypotheses/s", "hypotheses": int(len(hyps))}


def host_api_c_extra(sc, calls: int):
    """The same host-buffer entry points called from plain C++ (scripts/micro/host_api_c.cpp, built here with g++ against the
    in-tree library): what the adapter's HandSearch::findHands pays per cloud without the Python binding between the calls."""
    import shutil
    import struct
    import subprocess
    import tempfile

    gxx = shutil.which("g++")
    if not gxx:
        return {"error": "no g++ on this box"}
    libdir = os.path.join(ROOT, "agile_grasp_amd", "lib")
    with tempfile.TemporaryDirectory() as tmp:
        exe, cloud = os.path.join(tmp, "host_api_c"), os.path.join(tmp, "cloud.bin")
        cmd = [gxx, "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "scripts", "micro", "host_api_c.cpp"),
               "-o", exe, "-L" + libdir, "-lagile_grasp_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
        try:
            subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
            with open(cloud, "wb") as f:
                f.write(struct.pack("<qq", sc.n, sc.samples.size))
                f.write(np.asarray(sc.cam_origins, np.float64).tobytes())
                f.write(sc.xyz.astype(np.float32).tobytes())
                f.write(sc.cam.astype(np.int32).tobytes())
                f.write(sc.samples.astype(np.int32).tobytes())
            out = subprocess.run([exe, cloud, str(calls)], capture_output=True, text=True, timeout=120)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
            if out.returncode != 0 or not line:
                return {"error": (out.stdout + out.stderr)[-300:]}
            r = json.loads(line[-1])
            r["ms_per_call"] = r["us_per_call_median"] / 1e3
            r["value"] = r["hypotheses"] / (r["us_per_call_median"] * 1e-6)
            r["unit"] = "hypotheses/s"
            return r
        except Exception as e:  # noqa: BLE001
            return {"error": str(e)[-300:]}


def sq_issue_figures(kernel: str):
    """Issue-slot figures of one kernel from the committed SQ-counter pass of this workload (profiles/rNN_c2_sq_counters.txt:
    one `rocprofv3 --pmc SQ_*` run of `bench.py --config C2`, per-dispatch averages per shader engine, scripts/pmc_table.py).
    Static like roofline.traffic: counters cannot be collected inside the timed run."""
    import hashlib

    for name in ("r06_c2_sq_counters.txt", "r05_c2_sq_counters.txt", "r04_c2_sq_counters.txt"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        blob = open(path, "rb").read()
        cur, tab = None, {}
        for line in blob.decode().splitlines():
            if not line.startswith(" "):
                cur = line.strip()
            elif cur is not None and cur.startswith(kernel):
                k, v = line.split()
                tab[k] = float(v)
        need = ("SQ_INSTS_VALU", "SQ_WAVES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES")
        if not all(k in tab for k in need) or tab["SQ_WAVES"] <= 0 or tab["SQ_WAVE_CYCLES"] <= 0:
            continue
        # Per shader engine (the table holds per-SE averages; 32 SEs x 8 CUs x 4 SIMDs): SQ_ACTIVE_INST_VALU counts, in units of
        # four cycles, the time a wave's VALU instruction occupies its SIMD; a SIMD executes one at a time, so
        # 4 x SQ_ACTIVE_INST_VALU / (32 SIMDs x SQ_BUSY_CYCLES) is the fraction of the kernel's duration the average SIMD's vector
        # ALU is busy -- it cannot exceed 1 (the old `simd_issue_frac` = resident waves x a wave's issuing share did).
        simds_per_se = 32.0
        out = {"valu_insts_per_wave": tab["SQ_INSTS_VALU"] / tab["SQ_WAVES"], "waves": int(tab["SQ_WAVES"]) * 32,
               "wave_issuing_frac": tab["SQ_ACTIVE_INST_ANY"] / tab["SQ_WAVE_CYCLES"],
               "wave_valu_frac": tab["SQ_ACTIVE_INST_VALU"] / tab["SQ_WAVE_CYCLES"],
               "source": f"static: profiles/{name} sha256 {hashlib.sha256(blob).hexdigest()[:16]} (SQ counters of a separate "
                         "rocprofv3 --pmc pass over this workload, per-dispatch averages per shader engine)"}
        if tab.get("SQ_BUSY_CYCLES", 0) > 0:
            out["simd_valu_busy_frac"] = 4.0 * tab["SQ_ACTIVE_INST_VALU"] / (simds_per_se * tab["SQ_BUSY_CYCLES"])
            # SQ_WAVE_CYCLES is tallied in the same four-cycle units: resident wave-time over SIMD-time (<= 3 here: LDS bounds
            # the kernel to three work-groups per CU, one wave of each on every SIMD; what is missing from 3 is the tail)
            out["mean_resident_waves_per_simd"] = 4.0 * tab["SQ_WAVE_CYCLES"] / (simds_per_se * tab["SQ_BUSY_CYCLES"])
        return out
    return None


def pipeline_extra(steps: int):
    """grasp_localizer.cpp:95-103 per raw capture, host buffers in and out: the four entry points (preprocess, find_hands, classify,
    find_handles: four synchronisations) against agh_localize (one call, one synchronisation), same samples, same handles."""
    from agile_grasp_amd import binding, synthetic

    rc = synthetic.make_raw_cloud(700_000, 21)
    z = np.load(os.path.join(ROOT, "tests", "golden", "svm_weights.npz"))
    ctx = binding.Context(rc.cam_origins)
    ctx.load_svm(z["w"], float(z["rho"]))
    nv = ctx.preprocess(rc.xyz, rc.size_left, rc.workspace)
    samples = np.sort(np.random.default_rng(5).permutation(nv)[:2000]).astype(np.int32)

    def four():
        t0 = time.perf_counter()
        ctx.preprocess(rc.xyz, rc.size_left, rc.workspace)
        h = ctx.find_hands(samples)
        k = ctx.classify().astype(bool)
        hd, _ = ctx.find_handles(h[k], 3, 0.005)
        return time.perf_counter() - t0, len(h), int(k.sum()), len(hd)

    def one():
        t0 = time.perf_counter()
        r = ctx.localize(rc.xyz, rc.size_left, rc.workspace, samples=samples, classify=True, min_inliers=3, min_length=0.005)
        return time.perf_counter() - t0, r["n_hypotheses"], len(r["hands"]), len(r["handles"])

    for _ in range(3):
        four()
    t4 = [four() for _ in range(steps)]
    for _ in range(3):
        one()
    t1 = [one() for _ in range(steps)]
    xyz_dev = torch.from_numpy(rc.xyz).cuda()

    def one_dev():  # the raw capture already in device memory (agh_localize_device): no upload
        t0 = time.perf_counter()
        r = ctx.localize(xyz_dev, rc.size_left, rc.workspace, samples=samples, classify=True, min_inliers=3, min_length=0.005)
        return time.perf_counter() - t0, r["n_hypotheses"], len(r["hands"]), len(r["handles"])

    for _ in range(3):
        one_dev()
    td = [one_dev() for _ in range(steps)]
    assert td[-1][1:] == t1[-1][1:], (td[-1], t1[-1])
    assert t4[-1][1:] == t1[-1][1:], (t4[-1], t1[-1])
    # a STREAM of captures: the next capture staged (agh_localize_stage: second raw buffer, second stream) between
    # agh_localize_begin and agh_localize_end of this one, so its upload runs under this one's kernels
    caps = [np.ascontiguousarray(rc.xyz.copy()) for _ in range(3)]
    kw = dict(samples=samples, classify=True, min_inliers=3, min_length=0.005)

    def stream(n):
        t = []
        ctx.localize_begin(caps[0], rc.size_left, rc.workspace, **kw)
        t0 = time.perf_counter()
        for i in range(n):
            if i + 1 < n:
                ctx.localize_stage(caps[(i + 1) % 3])
            r = ctx.localize_end()
            if i + 1 < n:
                ctx.localize_begin(caps[(i + 1) % 3], rc.size_left, rc.workspace, **kw)
            t1_ = time.perf_counter()
            t.append((t1_ - t0, r["n_hypotheses"], len(r["hands"]), len(r["handles"])))
            t0 = t1_
        return t

    stream(4)
    ts = stream(steps + 2)[1:-1]  # (the first capture of a stream has nothing to hide behind, the last one stages nothing)
    assert ts[-1][1:] == t1[-1][1:], (ts[-1], t1[-1])
    # ... and with TWO contexts taking turns: capture k + 1 begins (upload and all) on the other context before capture k is
    # collected, so the two chains' kernels run side by side (the `two_streams` key, for the online chain)
    def make_ctx():
        c2 = binding.Context(rc.cam_origins)
        c2.load_svm(z["w"], float(z["rho"]))
        c2.preprocess(rc.xyz, rc.size_left, rc.workspace)
        return c2

    lanes = [ctx, make_ctx()]
    # (two streams of this process may share a hardware queue: side_by_side() replaces the second context until they do not)
    overlap = side_by_side(lanes, lambda c: c.localize_begin(caps[0] if c is lanes[0] else caps[1], rc.size_left, rc.workspace, **kw),
                           lambda c: c.localize_end(), make_ctx)
    ctx2 = lanes[1]

    def turns(n):
        t = []
        lanes[0].localize_begin(caps[0], rc.size_left, rc.workspace, **kw)
        t0 = time.perf_counter()
        for i in range(n):
            if i + 1 < n:
                lanes[(i + 1) & 1].localize_begin(caps[(i + 1) % 3], rc.size_left, rc.workspace, **kw)
            r = lanes[i & 1].localize_end()
            t1_ = time.perf_counter()
            t.append((t1_ - t0, r["n_hypotheses"], len(r["hands"]), len(r["handles"])))
            t0 = t1_
        return t

    turns(4)
    tt = turns(steps + 2)[1:-1]
    assert tt[-1][1:] == t1[-1][1:], (tt[-1], t1[-1])
    ctx2.close()
    return {"workload": "raw two-view capture, 699999 points -> 3 mm voxels -> 2000-sample search -> HOG + SVM -> handle search "
                        "(grasp_localizer.cpp:95-103), host buffers in and out",
            "voxels": int(nv), "hypotheses": int(t1[-1][1]), "svm_kept": int(t1[-1][2]), "handles": int(t1[-1][3]),
            "four_calls_ms": statistics.median(t[0] for t in t4) * 1e3, "agh_localize_ms": statistics.median(t[0] for t in t1) * 1e3,
            "agh_localize_min_ms": min(t[0] for t in t1) * 1e3, "agh_localize_max_ms": max(t[0] for t in t1) * 1e3,
            "four_calls_min_ms": min(t[0] for t in t4) * 1e3, "four_calls_max_ms": max(t[0] for t in t4) * 1e3,
            "agh_localize_device_ms": statistics.median(t[0] for t in td) * 1e3, "calls": steps,
            "begin_stage_end_ms": statistics.median(t[0] for t in ts) * 1e3, "begin_stage_end_min_ms": min(t[0] for t in ts) * 1e3,
            "begin_stage_end_max_ms": max(t[0] for t in ts) * 1e3,
            "two_contexts_ms": statistics.median(t[0] for t in tt) * 1e3, "two_contexts_min_ms": min(t[0] for t in tt) * 1e3,
            "two_contexts_max_ms": max(t[0] for t in tt) * 1e3, "two_contexts_side_by_side": overlap,
            "two_contexts_note": "per capture of a stream, two contexts taking turns: agh_localize_begin(k + 1) on the other context before "
                                 "agh_localize_end(k) -- the two chains' kernels side by side; same results",
            "begin_stage_end_note": "per capture of a stream, steady state: agh_localize_begin(k) / agh_localize_stage(k + 1) / "
                                    "agh_localize_end(k) -- capture k + 1 goes up under capture k's kernels; same results"}


def settle(ctx, step, fence):
    """Two untimed steps on every rank before anything is measured.  A context starts with the launches of the larger
    capacity classes switched off (they are empty for voxelised clouds); the first step of a cloud that needs them reports
    AGH_ERR_RETRY and switches them on for good, so the steps that follow -- warm-up and timed -- run the context's final
    configuration.  (Always two steps, so that every rank executes the same collectives.)"""
    from agile_grasp_amd import binding

    for _ in range(2):
        step()
        fence()
        try:
            ctx.synchronize()
        except binding.AghError as e:
            if e.code != binding.AGH_ERR_RETRY:
                raise


def cloud_per_gpu_secondary(args, dev, stream, rank, world, normals_mode):
    """N > 1, after the headline (sample-sharded) measurement: the same GPUs with one cloud of the C5 batch each (rank r:
    seed 10 + r), all of its 2000 samples, lists exchanged by the same library call -- weak scaling, reported as an extra
    key so that one multi-GPU run shows both ways of using the node."""
    import torch.distributed as dist

    from agile_grasp_amd import binding, synthetic

    sc = synthetic.config(f"C5_{rank}")
    ctx = binding.Context(sc.cam_origins, normals_mode=normals_mode, device=dev.index, profile=0)
    idt = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        idt.copy_(torch.frombuffer(bytearray(binding.comm_unique_id()), dtype=torch.uint8))
    dist.broadcast(idt, 0)
    ctx.comm_init(rank, world, bytes(idt.cpu().numpy().tobytes()))
    S = sc.samples.size
    xyz_t = torch.from_numpy(sc.xyz).to(dev)
    cam_t = torch.from_numpy(sc.cam).to(dev)
    s_all_t = torch.zeros(world * S, dtype=torch.int32, device=dev)
    s_all_t[rank * S:(rank + 1) * S] = torch.from_numpy(sc.samples).to(dev)
    out_t = torch.zeros(8 * S * 160 * world, dtype=torch.uint8, device=dev)
    nout_t = torch.zeros(1, dtype=torch.int64, device=dev)

    def step():
        ctx.set_cloud_torch(xyz_t, cam_t, stream=stream)
        ctx.find_hands_sharded_torch(s_all_t, out_t, nout_t, stream=stream)

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    settle(ctx, step, fence)  # (capacity classes / segment size: every rank learns both from the segment headers)
    for attempt in range(3):
        for _ in range(max(args.warmup, 3)):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        try:
            ctx.synchronize()
            break
        except binding.AghError as e:  # a segment overflowed: the context now exchanges full segments, measure again
            if e.code != binding.AGH_ERR_RETRY or attempt == 2:
                raise
    tv = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(tv, op=dist.ReduceOp.MAX)
    dt = float(tv[0].item())
    n_hyp = int(nout_t.item())
    ctx.comm_destroy()
    ctx.close()
    return {"workload": f"C5: {world} two-view 300000-point clouds (seeds 10..{9 + world}), one per GPU, 2000 samples each; lists "
                        "all-gathered by the library", "scaling": "weak", "n_gpus": world, "steps": args.steps,
            "ms_per_step": dt / args.steps * 1e3, "value": n_hyp * args.steps / dt, "unit": "hypotheses/s", "hypotheses": n_hyp}


def c5_batch_sharded_secondary(args, dev, stream, rank, world, normals_mode):
    """N ranks, extra key: BASELINE config C5 TO THE LETTER -- the FIXED batch of eight 300k-point clouds (seeds 10..17, 2000 samples
    each), its cloud-major sample list sharded contiguously over the ranks: with N dividing 8, rank r holds clouds
    [8 r / N, 8 (r + 1) / N) as one batch in its context (agh_set_cloud_batch_device) and searches all their samples in one
    launch set; the lists are exchanged by the library's all-gather.  Total work is fixed: "strong".  (N = 8 is one cloud per
    GPU, the headline's own configuration; N = 1 is the `batched` key of the single-GPU line through the sharded call.)"""
    import torch.distributed as dist

    from agile_grasp_amd import binding, synthetic

    C = 8
    if C % world:
        return {"error": f"{world} ranks do not divide the batch of {C} clouds"}
    per = C // world
    scs = [synthetic.config(f"C5_{k}") for k in range(rank * per, (rank + 1) * per)]
    ctx = binding.Context(scs[0].cam_origins, normals_mode=normals_mode, device=dev.index, profile=0)
    idt = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        idt.copy_(torch.frombuffer(bytearray(binding.comm_unique_id()), dtype=torch.uint8))
    dist.broadcast(idt, 0)
    ctx.comm_init(rank, world, bytes(idt.cpu().numpy().tobytes()))
    off = np.zeros(per + 1, np.int64)
    off[1:] = np.cumsum([s.n for s in scs])
    xyz_t = torch.from_numpy(np.concatenate([s.xyz for s in scs])).to(dev)
    cam_t = torch.from_numpy(np.concatenate([s.cam for s in scs])).to(dev)
    mine = np.concatenate([s.samples + off[k] for k, s in enumerate(scs)]).astype(np.int32)  # positions in MY point array
    S_all = C * 2000
    assert mine.size * world == S_all
    s_all_t = torch.zeros(S_all, dtype=torch.int32, device=dev)
    s_all_t[rank * mine.size:(rank + 1) * mine.size] = torch.from_numpy(mine).to(dev)
    out_t = torch.zeros(8 * S_all * 160, dtype=torch.uint8, device=dev)
    nout_t = torch.zeros(1, dtype=torch.int64, device=dev)

    def step():
        ctx.set_cloud_batch_torch(xyz_t, cam_t, off, stream=stream)
        ctx.find_hands_sharded_torch(s_all_t, out_t, nout_t, stream=stream)

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    settle(ctx, step, fence)
    steps = max(5, args.steps // 2)
    for attempt in range(3):
        for _ in range(max(args.warmup, 3)):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        try:
            ctx.synchronize()
            break
        except binding.AghError as e:
            if e.code != binding.AGH_ERR_RETRY or attempt == 2:
                raise
    tv = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(tv, op=dist.ReduceOp.MAX)
    dt = float(tv[0].item())
    n_hyp = int(nout_t.item())
    ctx.comm_destroy()
    ctx.close()
    return {"workload": f"C5 to the letter: the fixed batch of {C} two-view 300000-point clouds (seeds 10..17), {S_all} samples, the "
                        f"cloud-major sample list sharded over {world} GPUs ({per} cloud{'s' if per > 1 else ''} per GPU in one context), "
                        "lists all-gathered by the library", "scaling": "strong", "n_gpus": world, "steps": steps,
            "ms_per_step": dt / steps * 1e3, "ms_per_cloud": dt / steps * 1e3 / C, "value": n_hyp * steps / dt, "unit": "hypotheses/s",
            "hypotheses": n_hyp}


def sample_sharded_secondary(args, dev, stream, rank, world, normals_mode, cfg):
    """N > 1, extra keys: ONE cloud with its samples sharded over the GPUs (strong scaling) -- BASELINE config C4 (1M points,
    8000 samples), the smallest configuration whose sample count warrants sharding, and C2, whose 2000
    samples do not (every rank still builds the whole grid and runs the same latency chains)."""
    import torch.distributed as dist

    from agile_grasp_amd import binding, synthetic

    sc = synthetic.config(cfg)
    ctx = binding.Context(sc.cam_origins, normals_mode=normals_mode, device=dev.index, profile=0)
    idt = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        idt.copy_(torch.frombuffer(bytearray(binding.comm_unique_id()), dtype=torch.uint8))
    dist.broadcast(idt, 0)
    ctx.comm_init(rank, world, bytes(idt.cpu().numpy().tobytes()))
    S = sc.samples.size
    xyz_t, cam_t = torch.from_numpy(sc.xyz).to(dev), torch.from_numpy(sc.cam).to(dev)
    s_t = torch.from_numpy(sc.samples).to(dev)
    out_t = torch.zeros(8 * S * 160, dtype=torch.uint8, device=dev)
    nout_t = torch.zeros(1, dtype=torch.int64, device=dev)

    def step():
        ctx.set_cloud_torch(xyz_t, cam_t, stream=stream)
        ctx.find_hands_sharded_torch(s_t, out_t, nout_t, stream=stream)

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    settle(ctx, step, fence)
    steps = max(5, args.steps // 2) if cfg == "C4" else args.steps
    for attempt in range(3):
        for _ in range(max(args.warmup, 3)):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        try:
            ctx.synchronize()
            break
        except binding.AghError as e:
            if e.code != binding.AGH_ERR_RETRY or attempt == 2:
                raise
    tv = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(tv, op=dist.ReduceOp.MAX)
    dt = float(tv[0].item())
    n_hyp = int(nout_t.item())
    ctx.comm_destroy()
    ctx.close()
    return {"workload": f"{cfg}: two-view {sc.n}-point cloud, {S} samples sharded over {world} GPUs, one all-gather of the lists",
            "scaling": "strong", "n_gpus": world, "steps": steps, "ms_per_step": dt / steps * 1e3, "value": n_hyp * steps / dt,
            "unit": "hypotheses/s", "hypotheses": n_hyp}
