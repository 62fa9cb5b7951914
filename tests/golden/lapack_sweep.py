"""How often does the oracle differ from the LAPACK path?  The committed fixture (make_e2e_goldens.py) pins twelve cases; this
script runs the same comparison over MANY C2-sized scenes (300 000 points, the first N samples of each, fresh seeds, tilted)
and adds the counts up, so that the rates behind the contract of DESIGN.md section 2.1 -- hypotheses only in one list, SVM label
flips, index / flag flips -- rest on tens of thousands of hypotheses instead of 1 478; and beside every count it puts the
YARDSTICK: the LAPACK path against ITSELF with one unit in the last place added to or subtracted from every entry of M.  Needs
scipy; CPU only; prints one JSON line per scene and a total.  Usage: python tests/golden/lapack_sweep.py [first_seed] [scenes] [samples_per_scene] [antipodal | rand50]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_e2e_goldens as G  # noqa: E402

from oracle import oracle_py as O  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 500
scenes = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
antipodal = len(sys.argv) > 4 and sys.argv[4] == "antipodal"  # calculates_antipodal: the r = 0.01 all-points fits through LAPACK too
rand50 = len(sys.argv) > 4 and sys.argv[4] == "rand50"        # the reference's production normals mode (50 x rand() % n)
w, rho = O.load_svm(os.path.join(G.ROOT, "tests", "golden", "svm_032015_linear_20_20_same"))
keys = ("n_a", "only_a", "only_b", "svm_label_flips", "flips_finger_index", "flips_depth_index", "flips_cam_source", "flips_n_in_box",
        "max_index_mismatch", "degenerate_samples", "flips_half_antipodal", "flips_full_antipodal")
tot = {k: 0 for k in keys}
tot_self, worst_self = {}, {}
worst = {"max_abs_axis": 0.0, "max_abs_bottom": 0.0, "max_abs_surface": 0.0, "max_abs_width": 0.0, "max_abs_svm_sum": 0.0,
         "max_angle_params_rad": 0.0, "max_abs_n_in_box": 0}
for k in range(scenes):
    _sc, _samples, _gold, _hyps, _keep, _sums, rep = G.run_case(None, f"seed{first + k}", ns, w, rho, rand50=rand50, antipodal=antipodal,
                                                               self_all=not (antipodal or rand50))
    rep.setdefault("self", {})
    if antipodal:
        rep["antipodal_set"] = [int(_hyps["half_antipodal"].sum()), int(_hyps["full_antipodal"].sum())]
    for key in keys:
        tot[key] += rep[key]
        if key in rep["self"]:
            tot_self[key] = tot_self.get(key, 0) + rep["self"][key]
    for key in worst:
        worst[key] = max(worst[key], rep[key])
        if key in rep["self"]:
            worst_self[key] = max(worst_self.get(key, 0), rep["self"][key])
    line = {"oracle_vs_lapack": {key: rep[key] for key in ("case",) + keys + tuple(worst)}, "lapack_vs_lapack_one_ulp": rep["self"]}
    if antipodal:
        line["antipodal_flags_set_half_full"] = rep["antipodal_set"]
        line["normals"] = {key: rep[key] for key in rep if key.startswith("normals_")}
    print(json.dumps(line), flush=True)
print(json.dumps({"scenes": scenes, "samples_per_scene": ns, "oracle_vs_lapack": {"total": tot, "worst": worst},
                  "lapack_vs_lapack_one_ulp": {"total": tot_self, "worst": worst_self}}))
