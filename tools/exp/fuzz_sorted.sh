#!/bin/bash
# The options campaign (fuzz_campaign2.py) on the tree with the sorted launches: 100 seeds as they come, 100 with both sorts forced on
# every step of every scene, 100 with the default rule applied to steps of any size and the deferred accumulation forced.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
L=$O/fuzz_sorted.log
: > $L
echo "== default" >> $L
timeout -k 10 900 python -u tools/exp/fuzz_campaign2.py 9000 100 >> $L 2>&1 || { tail -20 $L; exit 1; }
echo "== RTX_SORT_ALL=1 RTX_SORT_OPAQUE=1 RTX_SORT_MIN=0" >> $L
RTX_SORT_ALL=1 RTX_SORT_OPAQUE=1 RTX_SORT_MIN=0 timeout -k 10 900 python -u tools/exp/fuzz_campaign2.py 9100 100 >> $L 2>&1 || { tail -20 $L; exit 1; }
echo "== RTX_SORT_MIN=0 RTX_DEFER_ACC=1" >> $L
RTX_SORT_MIN=0 RTX_DEFER_ACC=1 timeout -k 10 900 python -u tools/exp/fuzz_campaign2.py 9200 100 >> $L 2>&1 || { tail -20 $L; exit 1; }
grep "^==\|findings\|FINDING" $L | grep -v "seed .* done"
