"""Where does the per-step cost of the RCCL hand-over come from?  (a) a side-stream copy with event waits,
(b) ncclAllGather called directly on the SAME stream as the kernels (1-rank communicator)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from agile_grasp_amd import binding, sharding, synthetic

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
sc = synthetic.config("C2")
ctx = binding.Context(sc.cam_origins, device=0, profile=False)
S = sc.samples.size
xyz_t = torch.from_numpy(sc.xyz).to(dev); cam_t = torch.from_numpy(sc.cam).to(dev); s_t = torch.from_numpy(sc.samples).to(dev)
buf_t = torch.zeros(sharding.buffer_bytes(S), dtype=torch.uint8, device=dev)
nout_t = buf_t[:8].view(torch.int64); out_t = buf_t[160:]
nb = sharding.buffer_bytes_records(S)
g_t = torch.zeros(nb, dtype=torch.uint8, device=dev)
main = torch.cuda.current_stream(); stream = main.cuda_stream
side = torch.cuda.Stream()

rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
class UID(C.Structure):
    _fields_ = [("b", C.c_char * 128)]
uid = UID(); assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
rccl.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]

def step(mode):
    ctx.set_cloud_torch(xyz_t, cam_t, stream=stream)
    ctx.find_hands_torch(s_t, out_t, nout_t, stream=stream)
    if mode == "side":
        side.wait_stream(main)
        with torch.cuda.stream(side):
            g_t.copy_(buf_t[:nb])
        main.wait_stream(side)
    elif mode == "same":
        g_t.copy_(buf_t[:nb])
    elif mode == "same_kernel":
        torch.add(buf_t[:nb], 0, out=g_t)
    elif mode == "side_kernel":
        side.wait_stream(main)
        with torch.cuda.stream(side):
            torch.add(buf_t[:nb], 0, out=g_t)
        main.wait_stream(side)
    elif mode == "side_kernel_async":
        side.wait_stream(main)
        with torch.cuda.stream(side):
            torch.add(buf_t[:nb], 0, out=g_t)
    elif mode == "rccl_same":
        rc = rccl.ncclAllGather(C.c_void_p(buf_t.data_ptr()), C.c_void_p(g_t.data_ptr()), nb, 0, comm, C.c_void_p(stream))
        assert rc == 0

MODES = os.environ.get("XS_MODES", "none,same,same_kernel,side_kernel,side_kernel_async,rccl_same,none").split(",")
for mode in MODES:
    for _ in range(5): step(mode)
    torch.cuda.synchronize()
    K = 50
    t0 = time.perf_counter()
    for _ in range(K): step(mode)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{mode:10s}: host {1e6*(t1-t0)/K:.0f} us/step, total {1e6*(t2-t0)/K:.0f} us/step", flush=True)
