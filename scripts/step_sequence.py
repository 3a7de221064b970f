#!/usr/bin/env python
"""The kernels of the LAST bench step in a rocprofv3 kernel-trace CSV, in start order: start (us from the step's begin), duration, name, grid.
usage: step_sequence.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    name = r["Kernel_Name"].replace("void ", "").replace("dvo_hip::", "").split("(")[0]
    grid, wg = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0), int(r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or 1)
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, grid // max(wg, 1)))
ev.sort()
marks = [s for s, e, n, g in ev if n.startswith("k_build_from_raw<0") or n.startswith("k_ingest_strips<0")]
t0, t1 = marks[-2], marks[-1]
for s, e, n, g in ev:
    if t0 <= s < t1:
        print("%9.1f %9.1f  %-44s %8d" % ((s - t0) / 1e3, (e - s) / 1e3, n[:44], g))
