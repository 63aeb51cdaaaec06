#!/bin/bash
# Times every BASELINE.json config once (GPU only): gpu_workloads.sh [workload ...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for w in ${@:-sphere bunny sponza heptoroid sibenik}; do
  timeout -k 10 280 python bench.py --workload $w --cpu-blocks 0 --steps 2 --warmup 1 > $O/wl_$w.json 2> $O/wl_$w.err || { echo "$w failed"; tail -5 $O/wl_$w.err; exit 1; }
  python - "$w" "$O/wl_$w.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d["rays_per_frame"]
print("%-10s frame_ms %9.2f  Mrays/s %9.1f  rays %s  dev %s" % (sys.argv[1], d["frame_ms"], d["value"],
      {k: int(v) for k, v in r.items()}, {k: round(v, 1) for k, v in d["device_ms_rank0"].items()}))
PY
done
