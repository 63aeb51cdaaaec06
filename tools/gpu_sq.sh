#!/bin/bash
# one SQ counter pass (VALU instructions, busy, waits) of a bench command per library variant: gpu_sq.sh "bench args" variant...  ("base" = librtx.so)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=librtx_$v.so; [ "$v" = base ] && lib=librtx.so
  O=$R/gpurun_out/sq_$v
  rm -rf $O; mkdir -p $O
  export RTX_LIB=$lib
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O -o s -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-blocks 0 --serial $ARGS > $O/s.log 2>&1 || { tail -20 $O/s.log; exit 1; }
  DB=$(find $O -name "*.db" | head -1)
  echo "== $v"
  python3 $R/tools/pmc_summary.py $DB | python3 -c "
import sys
cur=None; d={}
def flush():
    if cur and 'SQ_WAVES' in d and d['SQ_WAVES']>0:
        w=d['SQ_WAVES']; print('%-42s valu/wave %7.0f salu/wave %6.0f lanes %.2f busy %.2f wait_inst %.2f' % (cur, d['SQ_INSTS_VALU']/w, d['SQ_INSTS_SALU']/w, d['SQ_THREAD_CYCLES_VALU']/max(1,d['SQ_ACTIVE_INST_VALU'])/64, d['SQ_ACTIVE_INST_VALU']*4/max(1,d['GRBM_GUI_ACTIVE'])/1024 , d['SQ_WAIT_INST_ANY']/max(1,d['SQ_WAVE_CYCLES'])))
for l in sys.stdin:
    if l.startswith('#'): continue
    if not l.startswith(' '):
        flush(); cur=l.split('dispatches')[0].strip()+' '+l.split('total')[1].strip(); d={}
    else:
        k,v=l.split(); d[k]=float(v)
flush()
"
done
