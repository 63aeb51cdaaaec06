#!/bin/bash
# GPU tests on librtx.so, then librtx.so against librtx_old.so on the given workloads (default overlap and --serial)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $O/gpu_tests_ab2.log 2>&1 || { tail -30 $O/gpu_tests_ab2.log; exit 1; }
tail -n 2 $O/gpu_tests_ab2.log
for w in "$@"; do
  for lib in librtx.so librtx_old.so; do
    for m in "" "--serial"; do
      RTX_LIB=$lib timeout -k 10 200 python bench.py --workload $w --cpu-blocks 0 --steps 3 --warmup 1 $m > $O/ab2.json 2> $O/ab2.err || { echo "$w $lib $m failed"; tail -5 $O/ab2.err; exit 1; }
      python - "$w $lib $m" "$O/ab2.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-32s frame_ms %9.2f  Mrays/s %9.1f  dev %s" % (sys.argv[1], d["frame_ms"], d["value"], {k: round(v, 1) for k, v in d["device_ms_rank0"].items()}))
PY
    done
  done
done
