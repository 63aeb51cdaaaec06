#!/bin/bash
# Whole-frame parity of a BASELINE config at its stated settings: every 8-row block of the frame traced by the oracle on the box's
# host cores and compared with the GPU frame (bench.py --cpu-blocks <all>).  Minutes of CPU work: a ticker keeps the call alive.
# gpu_fullframe.sh TAG workload [bench args...]
set -o pipefail
TAG=$1; W=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
H=$(python - "$W" "$@" <<'PY'
import sys, importlib.util
spec = importlib.util.spec_from_file_location("bench", "bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
h = b.WORKLOADS[sys.argv[1]][2]
a = sys.argv[2:]
if "--height" in a: h = int(a[a.index("--height") + 1])
print(h // 8)
PY
)
timeout -k 10 1100 python bench.py --workload $W --cpu-blocks $H --pmc off --steps 1 --warmup 1 "$@" > $O/full_${TAG}_$W.json 2> $O/full_${TAG}_$W.err &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 30; echo "  ... $W: oracle at work ($(date +%T))"; done
wait $pid || { tail -5 $O/full_${TAG}_$W.err; exit 1; }
python - "$O/full_${TAG}_$W.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"][:70], "| parity", d["parity_sample"], "| cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores")}, d["cpu_baseline"]["sample"][-40:])
PY
