"""Row f1 measurement: the GPU voxeliser (agh_preprocess_device, cloud resident in HBM) against the oracle's
std::set restatement on the host, on raw two-view captures.  Prints one JSON line per size."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from agile_grasp_amd import binding, synthetic
from oracle import oracle_py as orc

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ts = torch.cuda.Stream()
for n_raw in (100_000, 700_000, 2_400_000):
    rc = synthetic.make_raw_cloud(n_raw, 21)
    ctx = binding.Context(rc.cam_origins, profile=1)
    x_t = torch.from_numpy(rc.xyz).to(dev)
    torch.cuda.synchronize()
    for _ in range(3):
        nv = ctx.preprocess_torch(x_t, rc.size_left, rc.workspace, stream=ts.cuda_stream)
    ctx.synchronize(); ctx.timing()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        nv = ctx.preprocess_torch(x_t, rc.size_left, rc.workspace, stream=ts.cuda_stream)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / K
    tm = ctx.timing()
    t0 = time.perf_counter()
    v, cam = orc.preprocess(rc.xyz, rc.size_left, rc.workspace)
    cpu = time.perf_counter() - t0
    gv, gcam = ctx.cloud()
    ok = bool(np.array_equal(gv.view(np.uint32), v.view(np.uint32)) and np.array_equal(gcam, cam))
    pre_ms = tm.get("preprocess", 0.0) / K
    alg_bytes = 12.0 * rc.xyz.shape[0] * 3 + 1.0 * rc.xyz.shape[0] * 2 + 16.0 * nv  # 3 reads of xyz, code w+r, voxels out
    print(json.dumps({"raw_points": int(rc.xyz.shape[0]), "voxels": int(nv), "bit_exact_vs_oracle": ok,
                      "gpu_kernels_ms": pre_ms, "gpu_grid_build_ms": tm.get("grid_build", 0.0) / K,
                      "gpu_wall_ms_incl_2_syncs_and_grid": wall * 1e3, "cpu_oracle_ms": cpu * 1e3,
                      "raw_points_per_s": rc.xyz.shape[0] / wall,
                      "algorithmic_GBps_kernels": alg_bytes / (pre_ms * 1e-3) / 1e9 if pre_ms else None}))
    ctx.close()
