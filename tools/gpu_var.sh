#!/bin/bash
# serial-mode sponza timing of library variants: gpu_var.sh name...   (name = librtx_<name>.so, "base" = librtx.so)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
for v in "$@"; do
  lib=librtx_$v.so; [ "$v" = base ] && lib=librtx.so
  RTX_LIB=$lib timeout -k 10 100 python bench.py --cpu-blocks 0 --steps 3 --warmup 1 ${RTX_BENCH_ARGS:---serial} > $O/var_$v.json 2> $O/var_$v.err || { echo "$v failed"; tail -5 $O/var_$v.err; exit 1; }
  python - "$v" "$O/var_$v.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-12s frame_ms %9.2f  dev %s" % (sys.argv[1], d["frame_ms"], {k: round(v, 1) for k, v in d["device_ms_rank0"].items()}))
PY
done
