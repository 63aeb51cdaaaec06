#!/bin/bash
# The N = 2 flow of bench.py on a one-GPU box: both ranks on cuda:0, collectives on host tensors over gloo (everything but RCCL itself).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export RTX_BENCH_BACKEND=gloo
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 > $O/rehearsal_still.json 2> $O/rehearsal_still.err || { tail -20 $O/rehearsal_still.err; exit 1; }
tail -1 $O/rehearsal_still.json | cut -c1-900
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --workload sibenik --animate --frames 24 > $O/rehearsal_anim.json 2> $O/rehearsal_anim.err || { tail -20 $O/rehearsal_anim.err; exit 1; }
tail -1 $O/rehearsal_anim.json | cut -c1-1500
