#!/bin/bash
# A/B timing of library variants on the bench workload: gpu_ab.sh [-t variant_to_test] variant...
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "$1" = "-t" ]; then
  RTX_LIB=librtx_$2.so timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $O/gpu_tests_$2.log 2>&1 || { tail -30 $O/gpu_tests_$2.log; exit 1; }
  tail -2 $O/gpu_tests_$2.log
  shift; shift
fi
for v in "$@"; do
  lib=librtx_$v.so; [ "$v" = base ] && lib=librtx.so
  RTX_LIB=$lib timeout -k 10 100 python bench.py --cpu-blocks 0 > $O/bench_$v.json 2> $O/bench_$v.err || { tail -20 $O/bench_$v.err; exit 1; }
  python - "$v" "$O/bench_$v.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "frame_ms %.1f" % d["frame_ms"], {k: round(v, 1) for k, v in d["device_ms_rank0"].items()}, d.get("gpu_work_per_ray"))
PY
done
