#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSV per kernel.  usage: pmc_summary.py x_counter_collection.csv [...]"""
import collections
import csv
import sys

for f in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print("# " + f)
    for k in sorted(agg, key=lambda k: -dur[k]):
        if dur[k] < 0.05:
            continue
        print("%-40s dispatches %4d  total %.2f ms  VGPR/SGPR see csv" % (k, len(cnt[k]), dur[k]))
        for c, v in sorted(agg[k].items()):
            print("    %-28s %.6g" % (c, v))
