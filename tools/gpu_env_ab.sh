#!/bin/bash
# A/B of environment switches on the bench workload: gpu_env_ab.sh "NAME=VAL ..." "NAME2=VAL2" ...  ("-" = no switch); RTX_BENCH_ARGS as in gpu_var.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
i=0
for e in "$@"; do
  i=$((i+1))
  [ "$e" = "-" ] && e=""
  env $e timeout -k 10 150 python bench.py --cpu-blocks 0 --steps 3 --warmup 1 ${RTX_BENCH_ARGS:---serial} > $O/env_$i.json 2> $O/env_$i.err || { echo "$e failed"; tail -5 $O/env_$i.err; exit 1; }
  python - "$e" "$O/env_$i.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-28s frame_ms %9.2f  dev %s build_warm %s" % (sys.argv[1] or "-", d["frame_ms"], {k: round(v, 1) for k, v in d["device_ms_rank0"].items()}, d["config"].get("scene_build_upload_warm_s")))
PY
done
