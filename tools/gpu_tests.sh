#!/bin/bash
# GPU parity tests (+ optionally extra test files given as arguments, e.g. tests/wip_no_alloc.py), then smoke.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for f in "$@"; do   # files pytest does not collect by pattern (tests/wip_*.py): each named on its own
  timeout -k 10 300 python -u -m pytest "$f" -m gpu -x -v --timeout 200 --timeout-method thread > $O/gpu_$(basename $f .py).log 2>&1 || { tail -40 $O/gpu_$(basename $f .py).log; exit 1; }
  tail -5 $O/gpu_$(basename $f .py).log
done
[ -n "$ONLY_EXTRA" ] && exit 0
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread --durations=15 > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -22 $O/gpu_tests.log
timeout -k 10 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
