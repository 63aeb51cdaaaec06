#!/bin/bash
# Round-3 evidence of the final tree, into gpurun_out/profiles/ (copy to profiles/): gpu_r3_final.sh TAG
#  1. the bench line of every BASELINE config with live counter passes (tools/gpu_r3_profiles.sh)
#  2. rocprofv3 --kernel-trace --stats of the default bench command (overlap), of the same frame with RTX_FLAG_SERIAL (every kernel
#     alone) and of the heptoroid config
#  3. one SQ counter pass over the default (overlapped) frame's kernels (the profiler serialises the dispatches)
set -o pipefail
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
P=$O/profiles
mkdir -p $P
bash $R/tools/gpu_r3_profiles.sh $TAG || exit 1
cd /tmp && export TMPDIR=/tmp
kt() {   # name, bench args...
  local n=$1; shift
  timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $O/kt_${TAG}_$n -o kt -- python3 $R/bench.py --cpu-blocks 0 --pmc off "$@" > $O/kt_${TAG}_$n.log 2>&1 || { tail -20 $O/kt_${TAG}_$n.log; return 1; }
  local db=$(find $O/kt_${TAG}_$n -name "*.db" | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --cpu-blocks 0 --pmc off $*"; python3 $R/tools/rocpd_summary.py $db; } > $P/${TAG}_${n}_kernel_stats.txt || return 1
  head -8 $P/${TAG}_${n}_kernel_stats.txt | cut -c1-60,110-170
}
kt sponza || exit 1
kt sponza_serial --serial || exit 1
kt heptoroid --workload heptoroid --width 480 --height 272 --reflect 10 --refract 10 --steps 2 || exit 1
kt bunny --workload bunny || exit 1
kt sibenik --workload sibenik || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_${TAG}_overlap -o s -- python3 $R/bench.py --cpu-blocks 0 --pmc off --steps 1 --warmup 0 --child > $O/pmc_${TAG}_overlap.log 2>&1 || { tail -20 $O/pmc_${TAG}_overlap.log; exit 1; }
db=$(find $O/pmc_${TAG}_overlap -name "*.db" | head -1)
{ echo "# rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -- python3 bench.py --cpu-blocks 0 --pmc off --steps 1 --warmup 0 --child   (the default frame)"; python3 $R/tools/pmc_summary.py $db; } > $P/${TAG}_pmc_overlap_summary.txt || exit 1
grep -A12 "k_shade" $P/${TAG}_pmc_overlap_summary.txt | head -30
