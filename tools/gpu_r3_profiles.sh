#!/bin/bash
# The bench line of every BASELINE config with LIVE counters (bench.py --pmc auto), the counter passes kept under gpurun_out/profiles/
# (copy to profiles/ afterwards).  Usage: gpu_r3_profiles.sh TAG [workload ...]
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O/profiles
cd $R
[ -f profiles/pmc_traffic.json ] && [ ! -f $O/profiles/pmc_traffic.json ] && cp profiles/pmc_traffic.json $O/profiles/
for w in ${@:-sponza bunny sibenik heptoroid sphere}; do
  extra=""
  [ "$w" = heptoroid ] && extra="--width 480 --height 272 --reflect 10 --refract 10 --steps 2"
  timeout -k 10 900 python bench.py --workload $w $extra --save-profile $TAG --profile-dir $O/profiles > $O/profiles/${TAG}_bench_$w.json 2> $O/profiles/${TAG}_bench_$w.err || { echo "$w failed"; tail -5 $O/profiles/${TAG}_bench_$w.err; exit 1; }
  python - "$w" "$O/profiles/${TAG}_bench_$w.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("%-10s frame_ms %9.2f Mrays/s %9.1f | %s: bound %s frac %s valu %s hbm %s useful_flop %s alone_ms %s (profiled %s) src %s | cpu %s | errs %s %s" % (
    sys.argv[1], d["frame_ms"], d["value"], r.get("kernel"), r.get("bound"), r.get("frac"), r.get("valu_frac"), r.get("hbm_frac"),
    r.get("useful_flop_frac"), r.get("avg_launch_ms_alone"), r.get("avg_launch_ms_alone_profiled"), (r.get("source") or "")[:9],
    {k: d.get("cpu_baseline", {}).get(k) for k in ("value", "cores", "host_cores", "reference_num_cpus")}, d.get("pmc_live_error"), d.get("count_work_error")))
print("   alone:", {k: round(v["ms_per_frame"], 2) for k, v in d.get("kernel_ms_alone_live", {}).items()})
PY
done
