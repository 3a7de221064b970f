#!/usr/bin/env python
"""Kernel timeline of the LAST full step in a rocprofv3 kernel-trace CSV of `bench.py --loop-only` (steps delimited by the batched ingest
of the next batch's current frames): per kernel its start (us from the step's first match kernel), duration, the idle time of ITS stream
in front of it, the stream, name and workgroups; then the sums per stream.
usage: r5_step_timeline.py <kernel_trace.csv> [step from the end, default 2 = the one before the last]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = []
for r in rows:
    name = r["Kernel_Name"].replace("void ", "").replace("dvo_hip::", "").split("(")[0]
    grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y") or 1) * int(r.get("Grid_Size_Z") or 1)
    wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y") or 1) * int(r.get("Workgroup_Size_Z") or 1)
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, grid // max(wg, 1), r.get("Stream_Id") or r.get("Queue_Id") or "?"))
ev.sort()
# a step's match ends with the copy of its results to the host (batches up to 256 pairs have no k_finish launch any more)
begins = [i for i, e in enumerate(ev) if e[2] == "__amd_rocclr_copyBuffer"]
lo = begins[-back - 1] + 1 if len(begins) > back else 0
hi = begins[-back] + 1
win = ev[lo:hi]
match_stream = win[-1][4]
t0 = min(s for s, e, n, g, q in win if q == match_stream)
last_end = {}
busy = defaultdict(float)
gap_sum = 0.0
for s, e, n, g, q in win:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = max(e, last_end.get(q, 0))
    busy[q] += (e - s) / 1e3
    if q == match_stream:
        gap_sum += max(gap, 0.0)
    print("%9.1f %8.1f  gap %6.1f  s%-3s %-44s %7d" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, n[:44], g))
end = max(e for s, e, n, g, q in win if q == match_stream)
print("# match stream %s: span %.1f us, kernels busy %.1f us, idle between its kernels %.1f us; other streams busy: %s"
      % (match_stream, (end - t0) / 1e3, busy[match_stream], gap_sum, {q: round(v, 1) for q, v in busy.items() if q != match_stream}))
