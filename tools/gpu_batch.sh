#!/bin/bash
# frame time against rays per wavefront step: gpu_batch.sh workload batch...
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
w=$1; shift
for b in "$@"; do
  timeout -k 10 200 python bench.py --workload $w --cpu-blocks 0 --steps 3 --warmup 1 --batch-rays $b > $O/bt_$w$b.json 2> $O/bt_$w$b.err || { echo "$w $b failed"; tail -5 $O/bt_$w$b.err; exit 1; }
  python - "$w $b" "$O/bt_$w$b.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-18s frame_ms %9.2f  Mrays/s %9.1f  dev %s" % (sys.argv[1], d["frame_ms"], d["value"], {k: round(v, 1) for k, v in d["device_ms_rank0"].items()}))
PY
done
