#!/bin/bash
# Round-end style check on the GPU box: parity tests, smoke, bench, A/B variant benches, kernel trace.
# Every GPU step has its own time limit and the chain stops at the first failure.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
timeout -k 10 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 200 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/bench.json
for v in "$@"; do
  if [ -f rust-raytracer_amd/librtx_$v.so ]; then
    RTX_LIB=librtx_$v.so timeout -k 10 100 python bench.py --cpu-blocks 0 > $O/bench_$v.json 2> $O/bench_$v.err || { tail -20 $O/bench_$v.err; exit 1; }
    python - "$v" "$O/bench_$v.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "frame_ms %.1f" % d["frame_ms"], d["device_ms_rank0"], d.get("gpu_work_per_ray"))
PY
  fi
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof -o sponza -- python3 $R/bench.py --cpu-blocks 0 > $O/prof.log 2>&1 || { tail -20 $O/prof.log; exit 1; }
ls $O/prof
