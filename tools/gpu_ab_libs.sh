#!/bin/bash
# frame ms of library variants on workloads (overlapped, as the bench line): gpu_ab_libs.sh "w1 w2" lib1 lib2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
WL=$1; shift
for w in $WL; do
  for v in "$@"; do
    lib=librtx_$v.so; [ "$v" = base ] && lib=librtx.so
    for rep in 1 2; do
      RTX_LIB=$lib timeout -k 10 200 python bench.py --workload $w --cpu-blocks 0 --pmc off --steps 5 --warmup 2 --child 2>>gpurun_out/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $v', round(d['frame_ms'],3), {k: round(x,2) for k,x in d['device_ms_rank0'].items()})" || exit 1
    done
  done
done
