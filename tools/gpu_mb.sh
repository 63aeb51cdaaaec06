#!/bin/bash
# mailbox A/B: GPU tests, then heptoroid / bunny / sponza with librtx.so (8 slots), librtx_mb16.so and librtx_old.so
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $O/gpu_tests_mb.log 2>&1 || { tail -30 $O/gpu_tests_mb.log; exit 1; }
tail -n 2 $O/gpu_tests_mb.log
for lib in librtx.so librtx_mb16.so librtx_old.so; do
  echo "== $lib"
  RTX_LIB=$lib timeout -k 10 100 python tools/gpu_hept.py 240 136 10 2>&1 | tail -n 1 || exit 1
  for w in bunny sponza; do
    RTX_LIB=$lib timeout -k 10 200 python bench.py --workload $w --cpu-blocks 0 --steps 3 --warmup 1 --serial > $O/mb.json 2> $O/mb.err || { echo "$w $lib failed"; tail -5 $O/mb.err; exit 1; }
    python - "$w $lib serial" "$O/mb.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-32s frame_ms %9.2f  Mrays/s %9.1f  dev %s" % (sys.argv[1], d["frame_ms"], d["value"], {k: round(v, 1) for k, v in d["device_ms_rank0"].items()}))
PY
  done
done
