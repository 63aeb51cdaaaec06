#!/bin/bash
# The fuzz campaigns 1, 3, 4 and 5 of tools/exp on new seeds: fuzz_all.sh FIRST   (log: gpurun_out/fuzz_all.log)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
F=${1:-30000}
L=$O/fuzz_all.log
: > $L
timeout -k 10 600 python -u tools/exp/fuzz_campaign.py $F 150 >> $L 2>&1 || { tail -20 $L; exit 1; }
timeout -k 10 600 python -u tools/exp/fuzz_campaign3.py $F 200 >> $L 2>&1 || { tail -20 $L; exit 1; }
timeout -k 10 600 python -u tools/exp/fuzz_campaign4.py $F 150 >> $L 2>&1 || { tail -20 $L; exit 1; }
timeout -k 10 600 python -u tools/exp/fuzz_campaign5.py $F 100 >> $L 2>&1 || { tail -20 $L; exit 1; }
grep "^seeds" $L
