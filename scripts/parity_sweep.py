"""Bit-exact parity of the HIP path with the oracle over MANY scenes (not a pytest: a soak for rare events -- a pivot on the
deflation threshold, a tie in the argmax, a point exactly on a hand threshold).  Scenes: 300k points / 2000 samples, seeds
from the command line, tilted and axis-aligned, deterministic and production normals, every fourth one with the all-points
antipodal pass.  Usage: python scripts/parity_sweep.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from agile_grasp_amd import binding, synthetic
from oracle import oracle_py as O

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
FR = ("normal", "axis", "binormal", "params", "eigenvalue", "n_nb", "max_index", "majority_cam", "valid")
HY = ("sample", "orientation", "cam_source", "n_in_box", "half_antipodal", "full_antipodal", "finger_index", "depth_index", "axis",
      "approach", "binormal", "bottom", "surface", "width")
w, rho = O.load_svm(os.path.join(ROOT, "tests", "golden", "svm_032015_linear_20_20_same"))
bad = 0
total_h = 0
t0 = time.time()
for k in range(count):
    seed = first + k
    tilt = (k % 2) == 0
    sc = synthetic.make_scene(300_000, 2000, seed=seed, two_view=True, tilt=tilt)
    for mode in ("det", "rand50"):
        anti = (k % 4) == 3 and mode == "det"
        nm_b = binding.NORMALS_RAND50 if mode == "rand50" else binding.NORMALS_DETERMINISTIC
        nm_o = O.NORMALS_RAND50 if mode == "rand50" else O.NORMALS_DETERMINISTIC
        ctx = binding.Context(sc.cam_origins, normals_mode=nm_b, rand_seed=seed)
        ctx.set_cloud(sc.xyz, sc.cam)
        hyps = ctx.find_hands(sc.samples, calculates_antipodal=anti)
        ctx.load_svm(w, rho)
        keep = ctx.classify()
        fr = ctx.frames()
        ref = O.find_hands(O.default_params(sc.cam_origins, normals_mode=nm_o, rand_seed=seed), sc.xyz, sc.cam, sc.samples,
                           calculates_antipodal=anti, want_images=True)
        okeep, _ = O.classify(ref["images"], w, rho)
        msg = []
        if len(hyps) != len(ref["hyps"]):
            msg.append(f"count {len(hyps)} vs {len(ref['hyps'])}")
        else:
            msg += [f for f in HY if not np.array_equal(hyps[f], ref["hyps"][f], equal_nan=True)]
            if not np.array_equal(keep, okeep):
                msg.append("svm_keep")
        msg += ["frame." + f for f in FR if not np.array_equal(fr[f], ref["frames"][f], equal_nan=True)]
        total_h += len(hyps)
        bad += bool(msg)
        print(f"seed {seed} {'tilted' if tilt else 'axis-aligned'} {mode}{' antipodal' if anti else ''}: {len(hyps)} hypotheses, "
              f"{int(keep.sum())} kept, {'OK' if not msg else 'MISMATCH ' + ','.join(msg)}  [{time.time() - t0:.0f} s]", flush=True)
        ctx.close()
print(f"DONE: {2 * count} runs, {total_h} hypotheses, {bad} mismatching runs")
sys.exit(1 if bad else 0)
