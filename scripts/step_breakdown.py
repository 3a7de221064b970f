#!/usr/bin/env python
"""Where a bench step goes: per (kernel, grid) totals of the LAST `--steps` steps in a rocprofv3 kernel-trace CSV of bench.py, the
steps delimited by the launches of k_build_from_raw<0 (one per step: the ingest of the next batch's current frames).
usage: step_breakdown.py <kernel_trace.csv> [steps]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = []
for r in rows:
    name = r["Kernel_Name"].replace("void ", "").replace("dvo_hip::", "").split("(")[0]
    grid, wg = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0), int(r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or 1)
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, grid // max(wg, 1), r.get("Queue_Id") or r.get("Stream_Id") or ""))
ev.sort()
marks = [s for s, e, n, g, q in ev if n.startswith("k_build_from_raw<0") or n.startswith("k_ingest_strips<0")]
# the bench builds once per step; the measurement legs after the timed loop build nothing in that role
t0, t1 = marks[-steps - 1], marks[-1]
win = [x for x in ev if t0 <= x[0] < t1]
tot = defaultdict(lambda: [0, 0.0, 0.0])
for s, e, n, g, q in win:
    k = (n, g)
    tot[k][0] += 1
    tot[k][1] += (e - s) / 1e3
    tot[k][2] = max(tot[k][2], (e - s) / 1e3)
span = (t1 - t0) / 1e3
# union of busy intervals
busy, cur_s, cur_e = 0.0, None, None
for s, e, n, g, q in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += (cur_e - cur_s) if cur_e else 0
print("# %d steps, %.2f ms per step; some kernel running %.1f %% of the time; kernel time summed %.2f ms per step" % (steps, span / steps / 1e3, 100 * busy / 1e3 / span, sum(v[1] for v in tot.values()) / steps / 1e3))
print("%-52s %9s %8s %10s %10s %10s" % ("kernel", "wg", "n/step", "ms/step", "avg us", "max us"))
for (n, g), (c, t, m) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-52s %9d %8.1f %10.3f %10.1f %10.1f" % (n[:52], g, c / steps, t / steps / 1e3, t / c, m))
