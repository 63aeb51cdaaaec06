#!/bin/bash
# A/B of the two-stream overlap (default) against RTX_FLAG_SERIAL on the BASELINE workloads, after the GPU tests.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $O/gpu_tests_ov.log 2>&1 || { tail -30 $O/gpu_tests_ov.log; exit 1; }
tail -n 2 $O/gpu_tests_ov.log
for w in ${@:-sponza bunny sibenik}; do
  for m in "" "--serial"; do
    timeout -k 10 200 python bench.py --workload $w --cpu-blocks 0 --steps 3 --warmup 1 $m > $O/ov_$w$m.json 2> $O/ov_$w$m.err || { echo "$w $m failed"; tail -5 $O/ov_$w$m.err; exit 1; }
    python - "$w$m" "$O/ov_$w$m.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-18s frame_ms %9.2f  Mrays/s %9.1f  dev %s" % (sys.argv[1], d["frame_ms"], d["value"], {k: round(v, 1) for k, v in d["device_ms_rank0"].items()}))
PY
  done
done
