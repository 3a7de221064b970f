#!/usr/bin/env python
"""Per-kernel rooflines of the bench's streaming loop from rocprofv3 passes over ONE command (scripts/r5_rooflines.sh):

  <dir>/trace   --kernel-trace only: in-situ durations (the ingest of batch k + 1 runs beside the match of batch k)
  <dir>/fetch   --kernel-trace --pmc FETCH_SIZE          (kB; gfx950 reports half of a wide coalesced stream: x 2, MI355X_MICROARCH.md)
  <dir>/write   --kernel-trace --pmc WRITE_SIZE          (kB)
  <dir>/sq      --kernel-trace --pmc SQ_* / GRBM_GUI_ACTIVE   (vector-ALU and wait shares)

Kernels are told apart by name AND grid (one kernel serves several pyramid levels).  For every class only the FULL launches count
(at least half of the class's longest duration / largest traffic: a step enqueued ahead of the host's poll, or the tail of a level
with a few pairs left, moves next to nothing).  Algorithmic bytes per launch are the DESIGN.md section 4 figures for the bench's
workload (1024 pairs of 640 x 480, levels 3 -> 0).

usage: kernel_rooflines.py <dir> <pairs> [out.json] > table.md"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

HBM_PEAK = 8000.0  # GB/s


def klass(row):
    """(kernel name without arguments, workgroups of the launch): the kernel-trace CSV has the grid per dimension, the counter CSV in total"""
    name = (row.get("Kernel_Name") or "").replace("void ", "").replace("dvo_hip::", "")
    name = name.split("(")[0]
    if row.get("Grid_Size_X"):
        grid = int(row["Grid_Size_X"]) * int(row.get("Grid_Size_Y") or 1) * int(row.get("Grid_Size_Z") or 1)
        wg = int(row["Workgroup_Size_X"]) * int(row.get("Workgroup_Size_Y") or 1) * int(row.get("Workgroup_Size_Z") or 1)
    else:
        grid, wg = int(row.get("Grid_Size") or 0), int(row.get("Workgroup_Size") or 1)
    return name, grid // max(wg, 1)


def loop_rows(path):
    """rows of the streaming loop only: everything before the first batched ingest (frame creation, table uploads of the set-up) is dropped"""
    rows = list(csv.DictReader(open(path)))
    starts = [int(r["Start_Timestamp"]) for r in rows if "k_ingest_strips<0" in r["Kernel_Name"] or "k_ingest_strips<1" in r["Kernel_Name"]]
    t0 = min(starts) if starts else 0
    return [r for r in rows if int(r["Start_Timestamp"]) >= t0]


def load_trace(d):
    out = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in loop_rows(f):
            out[klass(row)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)   # us
    return out


def load_counters(d):
    """counter values per class, and under "_us" the durations of the same dispatches: a counter pass runs one kernel at a time, so
    these are the kernels ALONE (no ingest beside the match)"""
    out = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for row in loop_rows(f):
            out[klass(row)][row["Counter_Name"]].append(float(row["Counter_Value"]))
            if row["Dispatch_Id"] not in seen:
                seen.add(row["Dispatch_Id"])
                out[klass(row)]["_us"].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
    return out


def full(values):
    if not values:
        return []
    top = max(values)
    return [v for v in values if v >= 0.5 * top]


def mean(v):
    return sum(v) / len(v) if v else None


def main():
    d, pairs = sys.argv[1], int(sys.argv[2])
    out_json = sys.argv[3] if len(sys.argv) > 3 else None
    W, H = 640, 480
    px = [W * H >> (2 * l) for l in range(4)]
    trace = load_trace(os.path.join(d, "trace"))
    fetch = load_counters(os.path.join(d, "fetch"))
    write = load_counters(os.path.join(d, "write"))
    sq = load_counters(os.path.join(d, "sq"))
    tiles = {0: 300, 1: 75, 2: 24}                          # 64 x 16 tiles of the window sweep per pair (level 2: 3 x 8, last column half empty)

    def algorithmic(name, grid):
        """(bytes per full launch, what they are), DESIGN.md section 4"""
        if name.startswith("k_sweep_fast") or name.startswith("k_sweep_window"):
            for l, t in tiles.items():
                if grid == t * pairs or grid == ((t * pairs + 7) // 8) * 8:
                    return 40.0 * px[l] * pairs, "40 B x %d level-%d pixels x %d pairs (SURVEY 8d)" % (px[l], l, pairs)
        if name.startswith("k_residual_reduce_mfma"):
            return 40.0 * px[3] * pairs, "40 B x %d level-3 pixels x %d pairs" % (px[3], pairs)
        if name.startswith("k_loglik"):
            return 8.0 * px[0] * pairs, "8 B x %d level-0 pixels x %d pairs (one residual pair per pixel; the packed layout holds the constraints only)" % (px[0], pairs)
        if name.startswith("k_ingest_strips<1"):
            # (33 B per level-0 pixel of a reference frame with the 3-B copy of its raw planes; the streaming loop's
            #  reference frames keep none, option keep_raw_copy)
            return 30.0 * px[0] * pairs, "30 B per level-0 pixel of a reference frame that keeps no copy of its raw planes x %d frames" % pairs
        if name.startswith("k_ingest_strips<0"):
            return None, "current frames: the planes written follow the batch (plane C only for a batch this size): moved bytes are the measure"
        return None, ""

    rows = []
    for key in sorted(trace, key=lambda k: -sum(trace[k])):
        name, grid = key
        if sum(trace[key]) < 0.005 * sum(sum(v) for v in trace.values()):
            continue
        dur = full(trace[key])
        f = full(fetch.get(key, {}).get("FETCH_SIZE", []))
        w = full(write.get(key, {}).get("WRITE_SIZE", []))
        moved = (2.0 * mean(f) + mean(w)) * 1024.0 if f and w else None
        algo, what = algorithmic(name, grid)
        ms = mean(dur) * 1e-3
        alone = full(fetch.get(key, {}).get("_us", []))
        ms_alone = mean(alone) * 1e-3 if alone else None
        s = sq.get(key, {})
        def share(num, den, scale=1.0):
            a, b = s.get(num), s.get(den)
            if not a or not b:
                return None
            a, b = full(a), full(b)
            return scale * mean(a) / mean(b) if mean(b) else None
        # (the SQ counters of this collection are one XCD's -- 32 compute units, 128 SIMDs: SQ_ACTIVE_INST_VALU x 4 SIMD-cycles per
        # count against GRBM_GUI_ACTIVE x 128; profiles/r04_pmc3_v8_summary.txt reads 0.83 for the finest sweep the same way)
        valu = share("SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", 4.0 / 128.0)
        wait = share("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")
        rec = dict(kernel=name, workgroups=grid, launches_full=len(dur), launches=len(trace[key]), ms=round(ms, 4),
                   ms_alone=None if ms_alone is None else round(ms_alone, 4),
                   moved_GBps_alone=None if (moved is None or ms_alone is None) else round(moved / ms_alone / 1e6, 1),
                   step_share=round(sum(trace[key]) / sum(sum(v) for v in trace.values()), 4),
                   algorithmic_bytes=algo, algorithmic_note=what,
                   achieved_GBps=None if algo is None else round(algo / ms / 1e6, 1),
                   frac=None if algo is None else round(algo / ms / 1e6 / HBM_PEAK, 4),
                   moved_bytes=moved, moved_GBps=None if moved is None else round(moved / ms / 1e6, 1),
                   moved_frac=None if moved is None else round(moved / ms / 1e6 / HBM_PEAK, 4),
                   valu_active_share=None if valu is None else round(valu, 3), wave_wait_share=None if wait is None else round(wait, 3))
        if moved is not None and rec["moved_frac"] >= 0.45:
            rec["limited_by"] = "HBM"
        elif valu is not None and valu >= 0.6:
            rec["limited_by"] = "vector-instruction issue"
        elif ms < 0.1:
            rec["limited_by"] = "latency (dependent round trips of a short launch)"
        else:
            rec["limited_by"] = "latency / issue mix"
        rows.append(rec)
    print("| kernel | workgroups | full launches | ms in situ | ms alone | share of kernel time | achieved GB/s @ algorithmic | frac | moved GB/s (PMC) | moved frac | moved GB/s alone | VALU active | waves waiting | limited by |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    fmt = lambda v, f="%.3g": "-" if v is None else f % v
    for r in rows:
        print("| `%s` | %d | %d of %d | %.4f | %s | %.1f %% | %s | %s | %s | %s | %s | %s | %s | %s |" % (
            r["kernel"][:70], r["workgroups"], r["launches_full"], r["launches"], r["ms"], fmt(r["ms_alone"], "%.4f"), 100 * r["step_share"], fmt(r["achieved_GBps"], "%.0f"),
            fmt(r["frac"]), fmt(r["moved_GBps"], "%.0f"), fmt(r["moved_frac"]), fmt(r["moved_GBps_alone"], "%.0f"), fmt(r["valu_active_share"]), fmt(r["wave_wait_share"]),
            r["limited_by"]))
    worst = [r for r in rows if r["frac"] is not None]
    if worst:
        w = min(worst, key=lambda r: r["frac"])
        print("\nworst kernel at its algorithmic bytes: `%s` (%d workgroups): %.3f of the HBM roofline" % (w["kernel"], w["workgroups"], w["frac"]))
    if out_json:
        json.dump(dict(pairs=pairs, hbm_peak_GBps=HBM_PEAK, kernels=rows,
                       note="rocprofv3 passes over `bench.py --loop-only` (scripts/r5_rooflines.sh): durations in situ, FETCH_SIZE x 2 + WRITE_SIZE per full launch"),
                  open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
