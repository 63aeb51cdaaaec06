#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel.  Reads either a counter_collection CSV or a rocpd
results.db (view `counters_collection`).  usage: pmc_summary.py file [...]"""
import collections
import csv
import sqlite3
import sys


def rows(f):
    if f.endswith(".db"):
        c = sqlite3.connect(f)
        for name, did, cname, val, start, end in c.execute(
                "select kernel_name, dispatch_id, counter_name, value, start, end from counters_collection"):
            yield name, did, cname, float(val), int(start), int(end)
    else:
        for r in csv.DictReader(open(f)):
            yield (r["Kernel_Name"], r["Dispatch_Id"], r["Counter_Name"], float(r["Counter_Value"]),
                   int(r["Start_Timestamp"]), int(r["End_Timestamp"]))


for f in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    seen = set()
    for name, did, cname, val, start, end in rows(f):
        k = name.split("(")[0][-40:]
        agg[k][cname] += val
        cnt[k].add(did)
        if did not in seen:
            seen.add(did)
            dur[k] += (end - start) / 1e6
    print("# " + f)
    for k in sorted(agg, key=lambda k: -dur[k]):
        if dur[k] < 0.05:
            continue
        print("%-40s dispatches %4d  total %.2f ms" % (k, len(cnt[k]), dur[k]))
        for c, v in sorted(agg[k].items()):
            print("    %-28s %.6g" % (c, v))
