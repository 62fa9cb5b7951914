"""First-contact GPU diagnostics: prints where the HIP path and the oracle differ (not a test)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from agile_grasp_amd import binding, synthetic
from oracle import oracle_py as O

name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
sc = synthetic.config(name)
ctx = binding.Context(sc.cam_origins, profile=True)
print("selftest mismatches:", ctx.selftest_math(1 << 20))
t = time.time(); ctx.set_cloud(sc.xyz, sc.cam); print("set_cloud", time.time() - t)
t = time.time(); hyps = ctx.find_hands(sc.samples); print("find_hands", time.time() - t, len(hyps))
print("timing", ctx.timing())
fr = ctx.frames(); nt, nh = ctx.neighbor_counts()
p = O.default_params(sc.cam_origins)
t = time.time(); ref = O.find_hands(p, sc.xyz, sc.cam, sc.samples, want_images=True); print("oracle", time.time() - t, len(ref["hyps"]))
rf = ref["frames"]
print("n_nb equal:", np.array_equal(fr["n_nb"], rf["n_nb"]), "nh equal:", np.array_equal(nh, ref["nh"]))
for f in ("valid", "majority_cam", "max_index"):
    print(f, "equal:", np.array_equal(fr[f], rf[f]))
for f in ("params", "eigenvalue", "normal", "axis", "binormal"):
    d = np.abs(fr[f] - rf[f]); print(f, "bit-equal:", np.array_equal(fr[f], rf[f]), "max abs diff", d.max())
rh = ref["hyps"]
print("hyp count", len(hyps), len(rh))
if len(hyps) == len(rh):
    for f in hyps.dtype.names:
        if f == "epoch": continue
        eq = np.array_equal(hyps[f], rh[f])
        print(" ", f, "equal" if eq else ("DIFF max %.3e" % np.abs(hyps[f].astype(np.float64) - rh[f].astype(np.float64)).max()))
    im = ctx.images()
    print("images equal:", np.array_equal(im, ref["images"]), (im != ref["images"]).sum())
    w, rho = O.load_svm(os.path.join(ROOT, "tests", "golden", "svm_032015_linear_20_20_same"))
    ctx.load_svm(w, rho)
    keep = ctx.classify()
    desc, sums = ctx.hog()
    okeep, osums = O.classify(ref["images"], w, rho)
    odesc = np.stack([O.hog(i) for i in ref["images"]])
    print("desc equal:", np.array_equal(desc, odesc), np.abs(desc - odesc).max(), "sums equal:", np.array_equal(sums, osums), "keep equal:", np.array_equal(keep, okeep), keep.sum())
else:
    a = set(zip(hyps["sample"], hyps["orientation"])); b = set(zip(rh["sample"], rh["orientation"]))
    print("only gpu", sorted(a - b)[:20]); print("only oracle", sorted(b - a)[:20])
