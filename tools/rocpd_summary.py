#!/usr/bin/env python3
"""Dump the per-kernel summary (`top_kernels` view = what `rocprofv3 --stats` tabulates) of a
rocprofv3 rocpd SQLite database as text.  usage: rocpd_summary.py results.db > profiles/x.txt"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cur = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
print("%-110s %8s %16s %14s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, total, avg, pct in cur.fetchall():
    print("%-110s %8d %16.3f %14.3f %8.3f" % (name[:110], calls, total, avg, pct))
