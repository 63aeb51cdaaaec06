#!/bin/bash
# Loop statistics of k_shadow_coop (librtx_loops.so = -DRTX_COUNT_LOOPS): per wave steps, drain rounds, exact-test rounds and their fill.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
W=${1:-sponza}
RTX_LIB=${RTX_LIB:-librtx_loops.so} timeout -k 10 200 python bench.py --workload $W --cpu-blocks 0 --pmc off --steps 1 --warmup 0 --child > $O/loops_$W.json 2> $O/loops_$W.err || { tail -5 $O/loops_$W.err; exit 1; }
python - "$O/loops_$W.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
w = d["gpu_work_per_ray"]
sh = d["rays_per_frame"]["shadow"]; cl = sum(d["rays_per_frame"][k] for k in ("primary", "reflect", "gloss", "refract"))
waves = sh / 64.0
steps = w["shadow"]["nodes"] * sh; drain = w["shadow"]["leaf_entries"] * sh; pairs = w["shadow"]["prim_tests"] * sh
pl = w["closest"]["nodes"] * cl; gsteps = w["closest"]["leaf_entries"] * cl; dev = w["closest"]["prim_tests"] * cl
print("per wave: steps %.2f  group-steps %.2f (/4 = %.2f per group)  drain events %.2f  drain rounds %.2f  exact rounds %.2f  pairs %.1f (fill %.1f of 64)" %
      (steps / waves, gsteps / waves, gsteps / waves / 4, dev / waves, drain / waves, pairs / waves, pl / waves, pl / max(pairs, 1)))
PY
