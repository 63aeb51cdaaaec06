#!/bin/bash
# serial-mode timing of library variants on several workloads: gpu_var_wl.sh "workloads" variant...   ("base" = librtx.so)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
WL=$1; shift
for v in "$@"; do
  lib=librtx_$v.so; [ "$v" = base ] && lib=librtx.so
  line="$v:"
  for w in $WL; do
    RTX_LIB=$lib timeout -k 10 100 python bench.py --workload $w --cpu-blocks 0 --steps 3 --warmup 1 ${RTX_WL_ARGS---serial} > $O/vw_${v}_$w.json 2> $O/vw_${v}_$w.err || { echo "$v $w failed"; tail -5 $O/vw_${v}_$w.err; exit 1; }
    line="$line $(python - "$w" "$O/vw_${v}_$w.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
m = d["device_ms_rank0"]
print("%s frame %.2f (closest+shade %.2f shadow %.2f)" % (sys.argv[1], d["frame_ms"], m["closest_shade"], m["shadow"]))
PY
)"
  done
  echo "$line"
done
