#!/bin/bash
# kernel stats (serial mode) of library variants on one workload: gpu_kstats_vars.sh WORKLOAD variant...
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
W=$1; shift
for v in "$@"; do
  lib=librtx_$v.so; [ "$v" = base ] && lib=librtx.so
  RTX_LIB=$lib bash $R/tools/gpu_kstats.sh ${W}_$v --workload $W --serial --steps 3 --warmup 1 > /dev/null || exit 1
  echo "== $v"; sed -n 2,5p $R/gpurun_out/ks_${W}_$v.txt | cut -c1-50,105-175
done
