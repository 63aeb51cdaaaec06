#!/usr/bin/env python3
"""Derives profiles/pmc_traffic.json (what bench.py quotes) from the three rocprofv3 --pmc passes of
tools/gpu_profile.sh.  usage: pmc_to_json.py gpurun_out/prof_TAG workload launches_shadow launches_closest tag"""
import collections
import json
import os
import sqlite3
import sys

d, workload, l_shadow, l_closest, tag = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4]), sys.argv[5]


def load(db):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    c = sqlite3.connect(db)
    seen = set()
    for name, did, cname, val, start, end in c.execute(
            "select kernel_name, dispatch_id, counter_name, value, start, end from counters_collection"):
        k = name.split("(")[0].split()[-1].split("<")[0]
        agg[k][cname] += float(val)
        if did not in seen:   # the profiler serialises the dispatches of a --pmc pass: this is the kernel alone
            seen.add(did)
            agg[k]["_seconds"] += (int(end) - int(start)) * 1e-9
            agg[k]["_dispatches"] += 1
    return agg


f, w, s = load(os.path.join(d, "f/f_results.db")), load(os.path.join(d, "w/w_results.db")), load(os.path.join(d, "s/s_results.db"))


def traffic(kernels, launches):   # gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2; KB -> bytes
    return sum(f[k]["FETCH_SIZE"] * 2 + w[k]["WRITE_SIZE"] for k in kernels) * 1024 / launches


def valu(k):
    c = s[k]
    simd_cycles = c["GRBM_GUI_ACTIVE"] / 8 * 1024          # GRBM_GUI_ACTIVE sums the 8 XCDs; 1024 SIMDs
    return {"busy": c["SQ_ACTIVE_INST_VALU"] * 4 / simd_cycles,           # SQ_ACTIVE_INST_* count quad-cycles
            "lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64),
            "valu_insts_per_wave": c["SQ_INSTS_VALU"] / c["SQ_WAVES"],
            "waves_waiting_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
            "l2_hit_rate": 1.0 - w[k]["TCC_MISS_sum"] / w[k]["TCC_REQ_sum"],
            # L2 -> CU: 128-byte requests served by the L2 per second of the kernel running alone
            "l2_request_gbs": w[k]["TCC_REQ_sum"] * 128 / w[k]["_seconds"] / 1e9,
            "hbm_gbs_alone": (f[k]["FETCH_SIZE"] * 2 + w[k]["WRITE_SIZE"]) * 1024 / w[k]["_seconds"] / 1e9,
            "alone_ms_per_launch": s[k]["_seconds"] * 1e3 / max(1.0, s[k]["_dispatches"])}


path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
out = json.load(open(path)) if os.path.exists(path) else {}
out["_source"] = ("profiles/%s_pmc_summary.txt (tools/gpu_profile.sh: separate rocprofv3 --pmc passes on `bench.py --steps 1 --warmup 0 "
                  "--cpu-blocks 0`): bytes per launch = (FETCH_SIZE x2 + WRITE_SIZE) KB x 1024 / launches; valu.busy = "
                  "SQ_ACTIVE_INST_VALU x4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)" % tag)
out[workload] = {"k_shadow": traffic(["k_shadow_coop"], l_shadow), "k_trace": traffic(["k_closest", "k_shade"], l_closest),
                 "k_shadow_valu": valu("k_shadow_coop"), "k_trace_valu": valu("k_closest")}
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out[workload], indent=1))
