#!/bin/bash
# k_shade's cost per ray by generation: the frame at reflect depth 1, 2, 3 and the config's own: shade_depth.sh sponza
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for w in "$@"; do
  for dep in 1 2 3 0; do
    extra=""; [ $dep != 0 ] && extra="--reflect $dep"
    timeout -k 10 200 python bench.py --workload $w $extra --cpu-blocks 0 --pmc off --steps 3 --warmup 1 2>>gpurun_out/depth.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_alone_live',{})
print('$w reflect $dep frame', round(d['frame_ms'],3), 'rays', d.get('rays_per_frame'), d.get('rays_by_type'), '  '.join('%s %.3f (%d)' % (n, x['ms_per_frame'], x.get('launches', 0)) for n, x in k.items()))" || exit 1
  done
done
