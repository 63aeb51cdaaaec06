#!/bin/bash
# rocprofv3 kernel stats of one bench workload: gpu_kstats.sh TAG bench-args...
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ks_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $O -o kt -- python3 $R/bench.py --cpu-blocks 0 "$@" > $O/kt.log 2>&1 || { tail -20 $O/kt.log; exit 1; }
DB=$(find $O -name "*.db" | head -1)
python3 $R/tools/rocpd_summary.py $DB > $O/../ks_$TAG.txt || exit 1
head -14 $O/../ks_$TAG.txt
tail -1 $O/kt.log | cut -c1-400
