#!/usr/bin/env python3
"""The two tables of DESIGN.md that quote profiles/<TAG>_*: design_tables.py TAG
  1. section 5: every frame kernel of every workload (alone): ms per frame, VALU busy, lanes, busy x lanes, instructions per wave,
     scalar issue, HBM, L2 (from <TAG>_bench_<workload>.json: kernel_ms_alone_live, and <TAG>_pmc_<workload>.json: derived)
  2. section 11: frame time, Mrays/s, CPU baseline and the dominant kernel's fractions per workload
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
P = os.path.join(ROOT, "profiles")


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


print("| kernel (alone, `%s`) | ms per frame | VALU busy | lanes per VALU instr. | **busy × lanes** | VALU / SALU instr. per wave | SALU issue per CU-cycle (peak 1) | HBM | L2 → CU, hit rate |" % tag)
print("|---|---|---|---|---|---|---|---|---|")
for w in ("sponza", "bunny", "sibenik", "heptoroid"):
    b = last_json(os.path.join(P, "%s_bench_%s.json" % (tag, w)))
    pm = json.load(open(os.path.join(P, "%s_pmc_%s.json" % (tag, w))))["derived"]
    dom = (b.get("roofline") or {}).get("kernel", "")
    for name, v in b["kernel_ms_alone_live"].items():
        d = pm.get(name)
        if not d:
            print("| %s `%s` | %.2f | (no counters) |" % (w, name, v["ms_per_frame"]))
            continue
        print("| %s `%s`%s | %.2f | %.3f | %.3f | **%.3f** | %s / %s | %.2f | %.2f TB/s = %.0f %% | %.2f TB/s, %.2f; %.0f %% of wave-cycles waiting |" % (
            w, name, " (dominant)" if name == dom else "", v["ms_per_frame"], d["busy"], d["lane_utilisation"], d["busy"] * d["lane_utilisation"],
            "{:,.0f}".format(d["valu_insts_per_wave"]).replace(",", " "), "{:,.0f}".format(d["salu_insts_per_wave"]).replace(",", " "),
            d["salu_issue_per_cu_cycle"], d["hbm_gbs_alone"] / 1e3, d["hbm_gbs_alone"] / 80.0, d["l2_request_gbs"] / 1e3, d["l2_hit_rate"],
            100.0 * d["waves_waiting_frac"]))
print()
print("| workload | frame ms | Mrays/s | CPU baseline Mrays/s (kind, cores) | dominant kernel | bound | frac | valu_frac | hbm_frac | useful_flop_frac |")
print("|---|---|---|---|---|---|---|---|---|---|")
for w in ("sphere", "bunny", "sponza", "heptoroid", "sibenik"):
    path = os.path.join(P, "%s_bench_%s.json" % (tag, w))
    if not os.path.exists(path):
        continue
    b = last_json(path)
    r = b.get("roofline") or {}
    c = b.get("cpu_baseline") or {}
    print("| %s | %.2f | %.0f | %s | `%s` | %s | %s | %s | %s | %s |" % (
        w, b["ms_per_step"], b["value"], ("%.2f (%s, %s)" % (c.get("value", 0.0), c.get("kind"), c.get("cores"))) if c else "-",
        r.get("kernel"), r.get("bound"), *("%.3f" % r[k] if isinstance(r.get(k), float) else "-" for k in ("frac", "valu_frac", "hbm_frac", "useful_flop_frac"))))
