"""Builds libagile_grasp_hip.so (hipcc, gfx950 only) in-tree.  No CPU fallback exists, by design."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = ["voxelize.hip", "grid.hip", "taubin.hip", "hand_sweep.hip", "hog_svm.hip", "handles.hip", "train.hip", "points.hip", "shard.hip", "localize.hip", "api.hip"]
LIB = os.path.join(HERE, "lib", "libagile_grasp_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value",
         "-fno-gpu-rdc"]
FLAGS += os.environ.get("AGH_EXTRA_FLAGS", "").split()  # experiments (e.g. -DAGH_FRAME_PAD=2048: one work-group less per CU)
if os.environ.get("AGH_DEBUG_BUILD") == "1":  # phase-timing hooks (AGH_DEBUG_STOP_*, AGH_DEBUG_CLOCKS) for scripts/
    FLAGS.append("-DAGH_DEBUG_HOOKS")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, "csrc", f) for f in SRC + ["agh_internal.h", "taubin_eigen.h"]] + [os.path.join(ROOT, "include", "agh.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.join(HERE, "lib"), exist_ok=True)
    objs = []
    procs = []
    for f in SRC:
        o = os.path.join(HERE, "lib", f.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(HERE, "csrc", f), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
