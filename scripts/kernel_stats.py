#!/usr/bin/env python
"""Per-kernel in-situ statistics from a rocprofv3 --kernel-trace CSV, with the no-op launches separated.

The batched Gauss-Newton loop enqueues one iteration AHEAD of the host's poll (so that the GPU never waits for the host); at
the end of a level that iteration is a launch whose workgroups all leave on `!active` after a few microseconds.  Averaging those
with the real sweeps makes `--stats` meaningless for the sweep kernel.  A launch counts as a no-op when it is shorter than
`--noop-fraction` (default 0.25) of the median of the longest half of that kernel's launches.
usage: kernel_stats.py <kernel_trace.csv> [--noop-fraction 0.25]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    frac = float(sys.argv[sys.argv.index("--noop-fraction") + 1]) if "--noop-fraction" in sys.argv else 0.25
    dur = defaultdict(list)
    for row in csv.DictReader(open(path)):
        name = row.get("Kernel_Name") or row.get("kernel_name")
        # one kernel serves several pyramid levels (same name, another grid): told apart by the grid size, in workgroups
        grid = row.get("Grid_Size") or row.get("Grid_Size_X") or ""
        wg = row.get("Workgroup_Size") or row.get("Workgroup_Size_X") or ""
        if grid and wg and int(wg) > 0:
            name = "[%7d wg] %s" % (int(grid) // int(wg), name.replace("void ", "").replace("dvo_hip::", ""))
        dur[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)   # us
    total = sum(sum(v) for v in dur.values())
    print("%-100s %6s %6s %10s %10s %10s %7s" % ("kernel", "real", "no-op", "avg us", "min us", "max us", "% time"))
    for name, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        v.sort()
        top = v[len(v) // 2:]
        ref = top[len(top) // 2]
        real = [x for x in v if x >= frac * ref]
        noop = len(v) - len(real)
        print("%-100s %6d %6d %10.2f %10.2f %10.2f %6.1f%%" % (name[:100], len(real), noop, sum(real) / len(real), real[0], real[-1], 100.0 * sum(v) / total))


if __name__ == "__main__":
    main()
