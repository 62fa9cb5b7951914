"""Why does one interleaved torch op cost ~200 us per step?  Variants: explicit stream, op count, op position."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from agile_grasp_amd import binding, sharding, synthetic

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
sc = synthetic.config("C2")
ctx = binding.Context(sc.cam_origins, device=0, profile=os.environ.get("XS_PROFILE") == "1")
S = sc.samples.size
xyz_t = torch.from_numpy(sc.xyz).to(dev); cam_t = torch.from_numpy(sc.cam).to(dev); s_t = torch.from_numpy(sc.samples).to(dev)
buf_t = torch.zeros(sharding.buffer_bytes(S), dtype=torch.uint8, device=dev)
nout_t = buf_t[:8].view(torch.int64); out_t = buf_t[160:]
nb = sharding.buffer_bytes_records(S)
g_t = torch.zeros(nb, dtype=torch.uint8, device=dev)
src = buf_t[:nb]
small_a = torch.zeros(64, device=dev); small_b = torch.zeros(64, device=dev)
expl = torch.cuda.Stream()

def run(mode, stream_obj):
    stream = stream_obj.cuda_stream
    def step():
        if mode == "op_first":
            torch.add(small_a, 1, out=small_b)
        ctx.set_cloud_torch(xyz_t, cam_t, stream=stream)
        if mode == "op_middle":
            torch.add(small_a, 1, out=small_b)
        ctx.find_hands_torch(s_t, out_t, nout_t, stream=stream)
        if mode == "op_last":
            torch.add(src, 0, out=g_t)
        if mode == "op_last_small":
            torch.add(small_a, 1, out=small_b)
        if mode == "op_last_x2":
            torch.add(small_a, 1, out=small_b); torch.add(small_b, 1, out=small_a)
    for _ in range(5): step()
    torch.cuda.synchronize()
    ctx.timing() if os.environ.get("XS_PROFILE") == "1" else None
    K = 50
    t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    extra = ""
    if os.environ.get("XS_PROFILE") == "1":
        extra = " " + " ".join(f"{k}={v / K * 1e3:.0f}" for k, v in ctx.timing().items())
    print(f"{mode:14s} stream={'null' if stream == 0 else 'explicit'}: host {1e6*(t1-t0)/K:.0f} total {1e6*(t2-t0)/K:.0f} us/step{extra}", flush=True)

for mode in ("none", "op_last", "op_last_small", "op_last_x2", "op_first", "op_middle", "none"):
    run(mode, torch.cuda.current_stream())
with torch.cuda.stream(expl):
    for mode in ("none", "op_last", "op_last_small", "none"):
        run(mode, expl)
