#!/bin/bash
# every kernel alone (serial frame) for ablation builds of k_shade (timing only, wrong images): shade_abl.sh "sponza bunny" base nosky notri ...
# (the -DRTX_ABL_SPAWN_* / -DRTX_ABL_SHADE_NO{SKY,TRI,TEX} switches these variants were built with were removed from rtx_kernels.hip after the
#  experiment: git show 5536f8d..1781f2d; the results are in DESIGN.md section 11)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
WL=$1; shift
for w in $WL; do
  for v in "$@"; do
    lib=librtx_$v.so; [ "$v" = base ] && lib=librtx.so
    RTX_LIB=$lib timeout -k 10 200 python bench.py --workload $w --cpu-blocks 0 --pmc off --steps 3 --warmup 1 2>>gpurun_out/abl.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_alone_live',{}); r=d['rays_per_frame']
print('%-8s %-8s frame %8.3f reflect %.2fM ' % ('$w', '$v', d['frame_ms'], r['reflect']/1e6), '  '.join('%s %.3f' % (n, x['ms_per_frame']) for n, x in k.items()))" || exit 1
  done
done
