#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-.}
for f in "-DAGH_CTX_STREAM_NONBLOCKING" "" ; do
  AGH_EXTRA_FLAGS="$f" python -c "from agile_grasp_amd import build; build.build(force=True)" > /dev/null 2>&1
  echo "=== flags: [$f]"
  python scripts/micro/two_threads_host_api.py 100 2>&1 | grep threads
  [ -n "$f" ] && timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error|^FAILED" | tail -3
done
python -c "from agile_grasp_amd import build; build.build(force=True)" > /dev/null 2>&1
