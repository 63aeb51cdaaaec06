#!/bin/bash
# Quick A/B of a build: frame time and every kernel's time alone for the four mesh workloads, no counters, no CPU baseline.
# gpu_quick.sh TAG [workload ...]   (RTX_LIB selects a variant library)
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for w in ${@:-sponza bunny sibenik heptoroid}; do
  extra=""
  [ "$w" = heptoroid ] && extra="--width 480 --height 272 --reflect 10 --refract 10"
  timeout -k 10 200 python bench.py --workload $w $extra --cpu-blocks 0 --pmc off --steps 3 --warmup 1 > $O/q_${TAG}_$w.json 2> $O/q_${TAG}_$w.err || { echo "$w failed"; tail -5 $O/q_${TAG}_$w.err; exit 1; }
  python - "$w" "$O/q_${TAG}_$w.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = d.get("kernel_ms_alone_live", {})
print("%-10s frame_ms %8.2f  Mrays/s %8.0f  alone: %s" % (sys.argv[1], d["frame_ms"], d["value"],
      "  ".join("%s %.2f" % (n, v["ms_per_frame"]) for n, v in k.items())))
PY
done
