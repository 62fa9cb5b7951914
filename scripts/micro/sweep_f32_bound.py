"""EXPERIMENT: upper bound of what a float32 pre-classification in k_hand_sweep's pass A could save (debug build:
AGH_DEBUG_STOP_SWEEP=20 runs pass A's arithmetic on float32 -- results are wrong at the margins, the timing is what is asked)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from agile_grasp_amd import binding, synthetic

sc = synthetic.config(sys.argv[1] if len(sys.argv) > 1 else "C2")
S = sc.samples.size
xyz_t = torch.from_numpy(sc.xyz).cuda(); cam_t = torch.from_numpy(sc.cam).cuda(); s_t = torch.from_numpy(sc.samples).cuda()
out_t = torch.zeros(8 * S * 160, dtype=torch.uint8, device="cuda"); n_t = torch.zeros(1, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream

def run(stop):
    os.environ.pop("AGH_DEBUG_STOP_SWEEP", None)
    if stop:
        os.environ["AGH_DEBUG_STOP_SWEEP"] = str(stop)
    ctx = binding.Context(sc.cam_origins, profile=True)
    for i in range(33):
        if i == 3:
            torch.cuda.synchronize(); ctx.timing()
        ctx.set_cloud_torch(xyz_t, cam_t, stream=st)
        ctx.find_hands_torch(s_t, out_t, n_t, stream=st)
    torch.cuda.synchronize()
    t = ctx.timing()
    return round(t["hand_sweep"] / 30 * 1e3, 1), int(n_t.item())

for rep in range(2):
    print("fp64 pass A:", run(0), " float32 pass A:", run(20))
