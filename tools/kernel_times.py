#!/usr/bin/env python3
"""Per-kernel times out of a rocprofv3 --kernel-trace database (rocprofv3 7.x writes <name>_results.db): kernels in launch
order, grouped by (name, grid), with count, average and minimum duration in microseconds. usage: kernel_times.py <results.db>"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
agg = collections.OrderedDict()
for name, s, e, g in db.execute("select name, start, end, grid_x from kernels order by start"):
    agg.setdefault((name[:100], g), []).append((e - s) / 1e3)
for (name, g), v in agg.items():
    print("%-100s grid %9d n=%3d avg %9.1f us min %9.1f" % (name, g, len(v), sum(v) / len(v), min(v)))
