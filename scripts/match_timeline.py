#!/usr/bin/env python
"""Timeline of the LAST launch-path match in a rocprofv3 kernel-trace CSV: every kernel between the last two k_finish launches with
its start offset, duration and the idle gap in front of it (us), then totals.  usage: match_timeline.py <kernel_trace.csv> [skip]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ev = []
for r in rows:
    name = r["Kernel_Name"].replace("void ", "").replace("dvo_hip::", "").split("(")[0]
    grid, wg = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0), int(r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or 1)
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, grid // max(wg, 1)))
ev.sort()
fin = [k for k, e in enumerate(ev) if e[2].startswith("k_finish")]
a, b = fin[-2 - skip] + 1, fin[-1 - skip]
t0 = ev[a][0]
busy = gaps = 0.0
prev_end = None
for s, e, n, g in ev[a:b + 1]:
    gap = 0.0 if prev_end is None else max(0.0, (s - prev_end) / 1e3)
    print("%9.1f %8.1f  gap %6.1f  %-46s %7d" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n[:46], g))
    busy += (e - s) / 1e3
    gaps += gap
    prev_end = e if prev_end is None else max(prev_end, e)
print("# %d kernels, span %.1f us, kernel time %.1f us, idle gaps %.1f us" % (b + 1 - a, (ev[b][1] - t0) / 1e3, busy, gaps))
