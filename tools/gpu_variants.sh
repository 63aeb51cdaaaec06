#!/bin/bash
# A/B of library variants on the bench workload: gpu_variants.sh [--args "bench args"] name1 name2 ...  (rust-raytracer_amd/librtx_<name>.so; "base" = librtx.so)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
ARGS="--cpu-blocks 0"
if [ "$1" = "--args" ]; then ARGS="$2"; shift 2; fi
for v in "$@"; do
  lib=librtx_$v.so; [ "$v" = base ] && lib=librtx.so
  RTX_LIB=$lib timeout -k 10 150 python bench.py $ARGS > $O/var_$v.json 2> $O/var_$v.err || { tail -20 $O/var_$v.err; exit 1; }
  python - "$v" "$O/var_$v.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "frame_ms %.2f" % d.get("frame_ms", d.get("ms_per_frame_node", 0)), d.get("device_ms_rank0"), "Mrays/s %.0f" % d["value"])
PY
done
