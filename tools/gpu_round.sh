#!/bin/bash
# One GPU call: bench (default, serial), animation bench, then the rocprofv3 passes of tools/gpu_profile.sh.  Usage: gpu_round.sh TAG
set -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 240 python bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err || { tail -20 $O/bench_$TAG.err; exit 1; }
cat $O/bench_$TAG.json
timeout -k 10 120 python bench.py --serial --cpu-blocks 0 > $O/bench_${TAG}_serial.json 2> $O/bench_${TAG}_serial.err || { tail -20 $O/bench_${TAG}_serial.err; exit 1; }
cat $O/bench_${TAG}_serial.json
timeout -k 10 120 python bench.py --workload sibenik --animate --frames 24 > $O/bench_${TAG}_anim.json 2> $O/bench_${TAG}_anim.err || { tail -20 $O/bench_${TAG}_anim.err; exit 1; }
cat $O/bench_${TAG}_anim.json
bash tools/gpu_profile.sh $TAG
# the other BASELINE configs (frame time, Mrays/s) and a kernel trace of the heptoroid config
cd $R
for w in "sphere" "bunny" "sibenik" "heptoroid --width 480 --height 272 --reflect 10 --refract 10 --steps 2"; do
  n=$(echo $w | cut -d' ' -f1)
  timeout -k 10 200 python bench.py --cpu-blocks 0 --workload $w > $O/bench_${TAG}_$n.json 2> $O/bench_${TAG}_$n.err || { tail -20 $O/bench_${TAG}_$n.err; exit 1; }
  python - "$n" "$O/bench_${TAG}_$n.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "frame_ms %.3f" % d["frame_ms"], "Mrays/s %.0f" % d["value"], d["device_ms_rank0"])
PY
done
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG/hept -o kt -- python3 $R/bench.py --cpu-blocks 0 --workload heptoroid --width 480 --height 272 --reflect 10 --refract 10 --steps 2 > $O/prof_$TAG/hept.log 2>&1 || { tail -20 $O/prof_$TAG/hept.log; exit 1; }
python3 $R/tools/rocpd_summary.py $O/prof_$TAG/hept/kt_results.db | cut -c1-70,110-170 | head -8
