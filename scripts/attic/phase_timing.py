"""Cumulative phase timings of the moments and hand-sweep kernels via their debug_stop early exits (not a test)."""
# NOTE: needs a library built with the phase-timing hooks: AGH_DEBUG_BUILD=1 python -c "from agile_grasp_amd import build; build.build(force=True)"
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from agile_grasp_amd import binding, synthetic

sc = synthetic.config(sys.argv[1] if len(sys.argv) > 1 else "C2")
S = sc.samples.size
xyz_t = torch.from_numpy(sc.xyz).cuda(); cam_t = torch.from_numpy(sc.cam).cuda(); s_t = torch.from_numpy(sc.samples).cuda()
out_t = torch.zeros(8 * S * 160, dtype=torch.uint8, device="cuda"); n_t = torch.zeros(1, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream

def run(env):
    for k in ("AGH_DEBUG_STOP_SWEEP", "AGH_DEBUG_STOP_MOMENTS", "AGH_DEBUG_STOP_FRAME"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ctx = binding.Context(sc.cam_origins, profile=True,
                          normals_mode=binding.NORMALS_RAND50 if "rand50" in sys.argv else binding.NORMALS_DETERMINISTIC)
    for i in range(13):
        if i == 3:
            torch.cuda.synchronize(); ctx.timing()
        ctx.set_cloud_torch(xyz_t, cam_t, stream=st)
        ctx.find_hands_torch(s_t, out_t, n_t, stream=st)
    torch.cuda.synchronize()
    t = ctx.timing()
    return {k: round(v / 10 * 1e3, 1) for k, v in t.items()}

print("full", run({}))
for stop in (1, 2, 3, 4):
    print("sweep stop", stop, run({"AGH_DEBUG_STOP_SWEEP": str(stop)}).get("hand_sweep"))
for stop in (1, 2):
    print("moments stop", stop, run({"AGH_DEBUG_STOP_MOMENTS": str(stop)}).get("taubin_moments"))
for stop in (1, 2, 3, 4, 5, 6, 7):
    print("frame stop", stop, run({"AGH_DEBUG_STOP_FRAME": str(stop)}).get("taubin_frame"))
