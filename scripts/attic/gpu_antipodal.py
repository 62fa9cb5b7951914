import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from agile_grasp_amd import binding, synthetic
from oracle import oracle_py as O
sc = synthetic.config("tiny")
ctx = binding.Context(sc.cam_origins)
ctx.set_cloud(sc.xyz, sc.cam)
sub = sc.samples[:40]
hyps = ctx.find_hands(sub, calculates_antipodal=True)
p = O.default_params(sc.cam_origins)
ref = O.find_hands(p, sc.xyz, sc.cam, sub, calculates_antipodal=True)
print(len(hyps), len(ref["hyps"]))
nr = ctx.normals()
allp = np.arange(sc.n, dtype=np.int32)
fr = O.fit_frames(p, sc.xyz, sc.cam, allp, 0.01)
exp = np.where(fr["valid"][:, None] != 0, fr["normal"], 0.0)
f2 = O.fit_frames(p, sc.xyz, sc.cam, sub, 0.03)
exp[sub] = np.where(f2["valid"][:, None] != 0, f2["normal"], exp[sub])
bad = np.nonzero((nr != exp).any(1))[0]
print("normals mismatching points:", bad.size, "of", sc.n, bad[:10])
if bad.size:
    i = bad[0]; print(i, nr[i], exp[i], fr["n_nb"][i], fr["valid"][i], fr["max_index"][i])
    print("n_nb of bad", np.bincount(fr["n_nb"][bad])[:20], "valid frac", fr["valid"][bad].mean())
    print("max abs diff", np.abs(nr - exp).max())
if len(hyps) == len(ref["hyps"]):
    for f in hyps.dtype.names:
        if f != "epoch" and not np.array_equal(hyps[f], ref["hyps"][f]): print("DIFF", f, np.nonzero(hyps[f] != ref["hyps"][f])[0][:10])
