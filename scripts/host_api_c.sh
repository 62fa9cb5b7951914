#!/bin/bash
# gpurun -- 'bash scripts/host_api_c.sh [scene] [calls]': builds and runs scripts/micro/host_api_c.cpp on a scene dump
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
scene=${1:-C2}; calls=${2:-50}
cd $R
python - <<PY
import struct, numpy as np
from agile_grasp_amd import synthetic
sc = synthetic.config("$scene")
with open("/tmp/cloud_$scene.bin", "wb") as f:
    f.write(struct.pack("<qq", sc.n, sc.samples.size)); f.write(np.asarray(sc.cam_origins, np.float64).tobytes())
    f.write(sc.xyz.astype(np.float32).tobytes()); f.write(sc.cam.astype(np.int32).tobytes()); f.write(sc.samples.astype(np.int32).tobytes())
PY
g++ -O2 -std=c++17 -Iinclude scripts/micro/host_api_c.cpp -o /tmp/host_api_c -Lagile_grasp_amd/lib -lagile_grasp_hip \
  -Wl,-rpath,$R/agile_grasp_amd/lib -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib || exit 1
/tmp/host_api_c /tmp/cloud_$scene.bin $calls
/tmp/host_api_c /tmp/cloud_$scene.bin $calls tests/golden/svm_032015_linear_20_20_same
