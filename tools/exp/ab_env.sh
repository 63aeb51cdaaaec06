#!/bin/bash
# frame ms of a workload under environment / argument variants: ab_env.sh workload "ENV=.. ENV=.." "bench args" ...   (pairs)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
w=$1; shift
while [ $# -gt 0 ]; do
  envs=$1; args=$2; shift 2
  for rep in 1 2; do
    env $envs timeout -k 10 200 python bench.py --workload $w $args --cpu-blocks 0 --pmc off --steps 5 --warmup 2 --child 2>>gpurun_out/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w [$envs] [$args]', round(d['frame_ms'],3))" || exit 1
  done
done
