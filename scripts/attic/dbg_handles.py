import sys, numpy as np
sys.path.insert(0, '.')
from agile_grasp_amd import binding, synthetic
from oracle import oracle_py as orc
from tests.test_handles import _cases
hands, mi, ml = _cases()["crafted_long"]
ctx = binding.Context(synthetic.camera_origins())
ghd, gidx = ctx.find_handles(hands, mi, ml)
hd, idx = orc.find_handles(hands, mi, ml)
print("n_inliers gpu", ghd["n_inliers"], "oracle", hd["n_inliers"], "first", ghd["first_inlier"], hd["first_inlier"])
for h in range(len(hd)):
    a = gidx[ghd["first_inlier"][h]:ghd["first_inlier"][h] + ghd["n_inliers"][h]]
    b = idx[hd["first_inlier"][h]:hd["first_inlier"][h] + hd["n_inliers"][h]]
    same = len(a) == len(b) and np.array_equal(a, b)
    print("handle", h, "same" if same else "DIFF", len(a), len(b))
    if not same:
        k = next(i for i in range(min(len(a), len(b))) if a[i] != b[i]) if len(a) and len(b) else 0
        print("  first diff at", k, a[max(0,k-3):k+5], b[max(0,k-3):k+5], "set equal", set(a.tolist()) == set(b.tolist()))
