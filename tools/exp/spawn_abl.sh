#!/bin/bash
# per-launch k_shade times of a serial frame for spawn-path ablation builds: spawn_abl.sh workload variant...
# (the -DRTX_ABL_SPAWN_* / -DRTX_ABL_SHADE_NO{SKY,TRI,TEX} switches these variants were built with were removed from rtx_kernels.hip after the
#  experiment: git show 5536f8d..1781f2d; the results are in DESIGN.md section 11)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
w=$1; shift
for v in "$@"; do
  lib=librtx_$v.so; [ "$v" = base ] && lib=librtx.so
  RTX_LIB=$lib tools/gpu_launches.sh ${w}_$v --workload $w --serial > /dev/null || exit 1
  python3 - gpurun_out/kl_${w}_$v.csv $v <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = []
for i, r in enumerate(rows):
    if r['name'].startswith('void k_shade'):
        out.append("%.0f/%.2fM" % (int(r['duration']) / 1e3, int(r['grid_x']) / 1e6))
print("%-14s k_shade launches (us / rays): %s" % (sys.argv[2], "  ".join(out)))
PY
done
