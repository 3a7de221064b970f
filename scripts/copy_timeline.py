#!/usr/bin/env python
"""Large host-to-device copies of a rocprofv3 --memory-copy-trace run next to the big kernels (rocprofv3 --kernel-trace): start, duration, rate."""
import csv, sys
copies = list(csv.DictReader(open(sys.argv[1])))
kernels = list(csv.DictReader(open(sys.argv[2]))) if len(sys.argv) > 2 else []
ev = []
for r in copies:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s > 2_000_000:
        ev.append((s, e, "COPY " + r.get("Direction", r.get("Name", ""))))
for r in kernels:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s > 1_000_000:
        ev.append((s, e, r["Kernel_Name"].split("(")[0].replace("void dvo_hip::", "")[:40]))
ev.sort()
t0 = ev[0][0] if ev else 0
for s, e, n in ev[-60:]:
    print("%10.2f ms  %8.2f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, n))
