#!/bin/bash
# sort_coherence.sh [levels] [scene]: per-launch times of k_shadow_products for the three feeler orders of tools/exp/sort_coherence.py
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sortco
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $O -o kt -- python3 $R/tools/exp/sort_coherence.py "$@" > $O/run.log 2>&1 || { tail -20 $O/run.log; exit 1; }
grep -v "^W\|amdgpu.ids" $O/run.log | tail -15
DB=$(find $O -name "*.db" | head -1)
python3 $R/tools/rocpd_launches.py $DB 2>/dev/null | grep "k_shadow_products\|k_nearest" | cut -d, -f1,4,5 | cut -c1-200
