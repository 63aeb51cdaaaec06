#!/bin/bash
# average duration of one kernel (name prefix) per library variant, serial mode: gpu_kstats_lib.sh KERNEL "bench args" variant...
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
K=$1; ARGS=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=librtx_$v.so; [ "$v" = base ] && lib=librtx.so
  O=$R/gpurun_out/kl_$v
  rm -rf $O; mkdir -p $O
  export RTX_LIB=$lib
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O -o kt -- python3 $R/bench.py --cpu-blocks 0 --serial --steps 3 --warmup 1 $ARGS > $O/kt.log 2>&1 || { tail -20 $O/kt.log; exit 1; }
  DB=$(find $O -name "*.db" | head -1)
  echo "$v: $(python3 $R/tools/rocpd_summary.py $DB | grep "$K" | head -3 | awk '{print $(NF-3), $(NF-2), $(NF-1)}' | tr '\n' ';')"
done
