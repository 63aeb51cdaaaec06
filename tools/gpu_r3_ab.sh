#!/bin/bash
# Round-3 A/B: parity tests, then every workload (frame ms), then serial per-kernel stats of sponza, bunny, heptoroid(480x272 d10).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
TAG=${1:-x}
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/gpu_tests_$TAG.log 2>&1 || { tail -30 $O/gpu_tests_$TAG.log; exit 1; }
tail -2 $O/gpu_tests_$TAG.log
bash tools/gpu_workloads.sh sphere bunny sponza sibenik || exit 1
timeout -k 10 200 python bench.py --cpu-blocks 0 --workload heptoroid --width 480 --height 272 --reflect 10 --refract 10 --steps 2 > $O/wl_hept_$TAG.json 2> $O/wl_hept_$TAG.err || { tail -5 $O/wl_hept_$TAG.err; exit 1; }
python - $O/wl_hept_$TAG.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("heptoroid480 frame_ms %.2f Mrays/s %.1f dev %s" % (d["frame_ms"], d["value"], {k: round(v, 1) for k, v in d["device_ms_rank0"].items()}))
PY
bash tools/gpu_kstats.sh ${TAG}_sponza_serial --serial --steps 2 || exit 1
bash tools/gpu_kstats.sh ${TAG}_bunny_serial --serial --workload bunny --steps 3 || exit 1
