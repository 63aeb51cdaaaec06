#!/bin/bash
# heptoroid frame time against the batch size (rays per launch): gpu_hept_batch.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
for b in 0 4194304 8388608 16777216; do
  timeout -k 10 150 python bench.py --workload heptoroid --width 480 --height 272 --reflect 10 --refract 10 --cpu-blocks 0 --steps 2 --warmup 1 --batch-rays $b > $O/hb_$b.json 2> $O/hb_$b.err || { echo "batch $b failed"; tail -5 $O/hb_$b.err; exit 1; }
  python - $b $O/hb_$b.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
m = d["device_ms_rank0"]
print("batch %s frame %.1f ms %.0f Mrays/s launches %s" % (sys.argv[1], d["frame_ms"], d["value"], m["launches_closest"]))
PY
done
