#!/bin/bash
# kernel timeline of the last frames of the reference's benchmark_slam on the engine -> gpurun_out/r03/dropin_timeline.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
D=$(python - <<'PY'
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
from dvo_slam_amd import datagen, tum
root = tempfile.mkdtemp()
seq = datagen.synth_sequence(31, 61, 640, 480, depth_noise=2.0, grey_noise=4.0, exposure=0.02)
tum.write_dataset(root, seq["grey"], seq["depth"], seq["poses"])
print(root)
PY
)
( cd $D && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/dtrace -o d -- $R/tests/dropin/_build/benchmark_slam _rgbdpair_file:=$D/assoc.txt _groundtruth_file:=$D/groundtruth.txt _estimate_trajectory:=true _trajectory_file:=$D/t.txt > $O/dropin_trace.log 2>&1 )
f=$(find /tmp/dtrace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' > $O/dropin_timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    name = r["Kernel_Name"].replace("void ", "").replace("dvo_hip::", "").split("(")[0]
    grid, wg = int(r.get("Grid_Size") or 0), int(r.get("Workgroup_Size") or 1)
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, grid // max(wg, 1)))
ev.sort()
res = [k for k, e in enumerate(ev) if e[2].startswith("k_match_resident")]
a = res[-3]
t0 = ev[a - 12][0]
for s, e, n, g in ev[a - 12:]:
    print("%10.1f %8.1f  %-40s %6d" % ((s - t0) / 1e3, (e - s) / 1e3, n[:40], g))
PY
tail -45 $O/dropin_timeline.txt
