#!/usr/bin/env python
"""Prints the kernel timeline of one batched match from a rocprofv3 kernel-trace CSV."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void dvo_hip::', '').replace('dvo_hip::', '')) for r in rows)
regions, cur = [], None
for s, e, n in ev:
    if n.startswith('k_init_pairs'):
        cur = [(s, e, n)]
    elif cur is not None:
        cur.append((s, e, n))
        if n.startswith('k_finish'):
            regions.append(cur)
            cur = None
reg = regions[int(sys.argv[2]) if len(sys.argv) > 2 else 3]
t0 = reg[0][0]
out = ["%7.0f+%5.1f %s" % ((s - t0) / 1e3, (e - s) / 1e3, n.replace('k_residual_reduce_mfma', 'K1').replace('k_solver_step', 'K3').replace('k_loglik', 'K2').replace('__amd_rocclr_copyBuffer', 'cp')[:14]) for s, e, n in reg]
for i in range(0, len(out), 5):
    print(" | ".join(out[i:i + 5]))
print("span us", (reg[-1][1] - t0) / 1e3, "busy us", sum(e - s for s, e, n in reg) / 1e3)
