#!/usr/bin/env python
"""Kernel timeline of the LAST match in a rocprofv3 kernel-trace CSV (kernels separated from the previous ones by > 40 us of idle
device start a new match): start (us from the first kernel of the match), duration, name."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dvo_hip::", "")) for r in rows)
groups, cur, last_end = [], [], None
for s, e, n in ev:
    if last_end is not None and s - last_end > 40000 and cur:
        groups.append(cur)
        cur = []
    cur.append((s, e, n))
    last_end = e
if cur:
    groups.append(cur)
g = groups[-1]
t0 = g[0][0]
for s, e, n in g:
    print("%9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n[:60]))
print("# %d kernels, span %.1f us, kernels busy %.1f us" % (len(g), (g[-1][1] - t0) / 1e3, sum(e - s for s, e, n in g) / 1e3))
