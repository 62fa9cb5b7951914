set -u
cd $GRAFT_REPO_ROOT
for cfg in "3 1408" "2 3800" "2 2176"; do
  set -- $cfg
  AGH_EXTRA_FLAGS="-DAGH_SWEEP_WGS=$1 -DAGH_SWEEP_TILE=$2" python -c "from agile_grasp_amd import build; build.build(force=True)"
  echo "=== WGS=$1 TILE=$2"
  timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_path" 2>&1 | tail -2
  timeout 200 bash scripts/quick_trace.sh C2 | grep -E "hand_sweep"
  timeout 200 bash scripts/quick_trace.sh C4 | grep -E "hand_sweep"
done
