#!/usr/bin/env python3
"""Per-frame kernel times from a rocprofv3 --stats kernel_stats.csv: kstats.py file.csv frames"""
import csv, sys
frames = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['Percentage']) > 0.5:
        print("%-42s calls %5s  ms/frame %8.2f  avg ms %7.3f  %5s%%" % (r['Name'][:42], r['Calls'], float(r['TotalDurationNs'])/1e6/frames, float(r['AverageNs'])/1e6, r['Percentage']))
