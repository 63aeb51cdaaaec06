#!/bin/bash
# The default bench line with its live counter passes for one library build: gpu_pmc_one.sh TAG [workload]  (RTX_LIB selects the build)
set -o pipefail
TAG=$1; W=${2:-sponza}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 400 python bench.py --workload $W --cpu-blocks 0 --steps 2 --warmup 1 --profile-dir $O/pmcdir_$TAG > $O/pmc_${TAG}_$W.json 2> $O/pmc_${TAG}_$W.err || { tail -5 $O/pmc_${TAG}_$W.err; exit 1; }
python - "$O/pmc_${TAG}_$W.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k in ("kernel", "avg_launch_ms_alone", "valu_busy", "lane_utilisation", "valu_insts_per_wave", "waves_waiting_frac", "frac", "l2_hit_rate", "traffic", "traffic_write", "useful_flop_frac")})
PY
