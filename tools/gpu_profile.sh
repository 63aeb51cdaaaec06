#!/bin/bash
# rocprofv3 evidence for the bench workload: kernel trace + stats, then three separate --pmc passes
# (HBM fetch, HBM write + L2 requests, SQ VALU/wait counters).  Usage: gpu_profile.sh TAG
set -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 1 --warmup 0 --cpu-blocks 0"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --cpu-blocks 0 > $O/kt.log 2>&1 || { tail -20 $O/kt.log; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d $O/f -o f -- $CMD > $O/f.log 2>&1 || { tail -20 $O/f.log; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d $O/w -o w -- $CMD > $O/w.log 2>&1 || { tail -20 $O/w.log; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/s -o s -- $CMD > $O/s.log 2>&1 || { tail -20 $O/s.log; exit 1; }
find $O -name "*.csv" -o -name "*.db" | head -20
