#!/usr/bin/env python3
"""Every kernel launch of a rocprofv3 rocpd database in start order (name, start, duration, grid): rocpd_launches.py results.db > x.csv"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
sys.stderr.write("kernels view columns: %s\n" % cols)
want = [x for x in ("name", "start", "end", "duration", "grid_x", "grid_size_x", "workgroup_x", "workgroup_size_x", "stream_id", "queue_id") if x in cols]
print(",".join(want))
for row in c.execute("select %s from kernels order by start" % ",".join(want)):
    print(",".join(str(x).replace(",", ";")[:60] for x in row))
