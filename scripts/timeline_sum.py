#!/usr/bin/env python
"""Per-kernel totals of the LAST match in a rocprofv3 kernel-trace CSV (see timeline_last.py for how matches are told apart)."""
import csv
import sys
from collections import OrderedDict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("dvo_hip::", "").split("(")[0]) for r in rows)
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
tot = OrderedDict()
for s, e, n in g:
    c, t = tot.get(n, (0, 0))
    tot[n] = (c + 1, t + e - s)
span = (g[-1][1] - g[0][0]) / 1e3
for n, (c, t) in tot.items():
    print("%4d x %-50s %9.1f us  (%5.1f us each, %4.1f %% of the span)" % (c, n[:50], t / 1e3, t / 1e3 / c, 100 * t / 1e3 / span))
print("# %d kernels, span %.1f us, kernels busy %.1f us" % (len(g), span, sum(e - s for s, e, n in g) / 1e3))
