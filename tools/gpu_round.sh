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
