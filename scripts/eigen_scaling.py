"""k_taubin_eigen time against the number of samples (waves = S / 4): is the 62 us a latency chain (flat in S until every
SIMD holds a wave) or does it grow with the waves per CU (a shared resource: instruction fetch of the 45 KB kernel)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from agile_grasp_amd import binding, synthetic

sc = synthetic.config("C4")
ctx = binding.Context(sc.cam_origins, profile=1)
dev = torch.device("cuda:0")
xyz_t, cam_t = torch.from_numpy(sc.xyz).to(dev), torch.from_numpy(sc.cam).to(dev)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
for S in (4, 16, 64, 256, 1024, 2048, 4096, 8000):
    s_t = torch.from_numpy(sc.samples[:S].copy()).to(dev)
    out_t = torch.zeros(8 * S * 160, dtype=torch.uint8, device=dev); n_t = torch.zeros(1, dtype=torch.int64, device=dev)
    ctx.set_cloud_torch(xyz_t, cam_t, stream=st.cuda_stream)
    for _ in range(3):
        ctx.find_hands_torch(s_t, out_t, n_t, stream=st.cuda_stream)
    torch.cuda.synchronize(); ctx.timing()
    for _ in range(20):
        ctx.find_hands_torch(s_t, out_t, n_t, stream=st.cuda_stream)
    torch.cuda.synchronize()
    t = {k: v / 20 * 1000 for k, v in ctx.timing().items()}
    print(S, "waves", (S + 3) // 4, {k: round(v, 1) for k, v in t.items() if k.startswith("taubin") or k == "hand_sweep"})
