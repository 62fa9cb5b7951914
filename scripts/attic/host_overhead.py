"""Host-side enqueue cost of one step (no GPU wait) and of the RCCL all-gather call, vs the GPU time per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.distributed as dist
from agile_grasp_amd import binding, sharding, synthetic

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
sc = synthetic.config("C2")
for profile in (False, True):
    ctx = binding.Context(sc.cam_origins, device=0, profile=profile)
    S = sc.samples.size
    xyz_t = torch.from_numpy(sc.xyz).to(dev); cam_t = torch.from_numpy(sc.cam).to(dev); s_t = torch.from_numpy(sc.samples).to(dev)
    buf_t = torch.zeros(sharding.buffer_bytes(S), dtype=torch.uint8, device=dev)
    nout_t = buf_t[:8].view(torch.int64); out_t = buf_t[160:]
    nb = sharding.buffer_bytes_records(S)
    g_t = torch.zeros(nb, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    def step(gather):
        ctx.set_cloud_torch(xyz_t, cam_t, stream=stream)
        ctx.find_hands_torch(s_t, out_t, nout_t, stream=stream)
        if gather:
            dist.all_gather_into_tensor(g_t, buf_t[:nb])
    for gather in (False, True):
        for _ in range(5): step(gather)
        torch.cuda.synchronize()
        K = 50
        t0 = time.perf_counter()
        for _ in range(K): step(gather)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"profile={profile} gather={gather}: host enqueue {1e6*(t1-t0)/K:.0f} us/step, total {1e6*(t2-t0)/K:.0f} us/step")
    # double-buffered, asynchronous exchange: step k's all-gather overlaps step k+1's kernels
    bufs = [torch.zeros(sharding.buffer_bytes(S), dtype=torch.uint8, device=dev) for _ in range(2)]
    gts = [torch.zeros(nb, dtype=torch.uint8, device=dev) for _ in range(2)]
    works = [None, None]
    def step2(k):
        i = k & 1
        if works[i] is not None: works[i].wait()
        ctx.set_cloud_torch(xyz_t, cam_t, stream=stream)
        ctx.find_hands_torch(s_t, bufs[i][160:], bufs[i][:8].view(torch.int64), stream=stream)
        works[i] = dist.all_gather_into_tensor(gts[i], bufs[i][:nb], async_op=True)
    for k in range(6): step2(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(50): step2(k)
    for w in works: w.wait()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"  async double-buffered: host {1e6*(t1-t0)/50:.0f} us, total {1e6*(t2-t0)/50:.0f} us/step")
    # the collective alone
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): dist.all_gather_into_tensor(g_t, buf_t[:nb])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"  all_gather alone: host {1e6*(t1-t0)/50:.0f} us, total {1e6*(t2-t0)/50:.0f} us")
    ctx.close()
dist.destroy_process_group()
