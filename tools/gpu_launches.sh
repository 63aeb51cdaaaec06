#!/bin/bash
# per-launch kernel times of one serial frame: gpu_launches.sh TAG bench-args...
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/kl_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace -d $O -o kt -- python3 $R/bench.py --cpu-blocks 0 --pmc off --child --steps 1 --warmup 0 "$@" > $O/kt.log 2>&1 || { tail -20 $O/kt.log; exit 1; }
DB=$(find $O -name "*.db" | head -1)
python3 $R/tools/rocpd_launches.py $DB > $O/../kl_$TAG.csv || exit 1
wc -l $O/../kl_$TAG.csv
