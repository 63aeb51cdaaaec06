#!/bin/bash
# frame time against the batch size (rays per launch): gpu_batch_sweep.sh workload [bench args...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
W=$1; shift
for b in 4194304 8388608 16777216 33554432; do
  timeout -k 10 150 python bench.py --workload $W --cpu-blocks 0 --steps 3 --warmup 1 --batch-rays $b "$@" > $O/bs_$b.json 2> $O/bs_$b.err || { echo "batch $b failed"; tail -5 $O/bs_$b.err; exit 1; }
  python - $b $O/bs_$b.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
m = d["device_ms_rank0"]
print("batch %s frame %.2f ms %.0f Mrays/s launches %s" % (sys.argv[1], d["frame_ms"], d["value"], m["launches_closest"]))
PY
done
