#!/bin/bash
# HIP API + kernel + copy trace of the batched validator (first process on the box): what a slow batch preparation waits for
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
seq = datagen.synth_sequence(77, 33, 640, 480)
tum.write_dataset(root, seq["grey"], seq["depth"], seq["poses"])
print(root)
PY
)
export DVO_VALIDATOR_TRACE=1 DVO_HIP_TRACE_SLOW=2
( cd /tmp && timeout 300 rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d /tmp/vtrace -o v -- $R/tests/cpp/validator_check gpu $D/assoc.txt $D/groundtruth.txt > $O/validation_trace.log 2>&1 )
grep "run\|slow" $O/validation_trace.log | head -12
ls /tmp/vtrace/*/ | head
for f in /tmp/vtrace/*/*hip_api_trace.csv /tmp/vtrace/*/*kernel_trace.csv /tmp/vtrace/*/*memory_copy_trace.csv; do cp $f $O/validation_$(basename $f); done
python - <<'PY'
import csv, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03")
api = list(csv.DictReader(open(os.path.join(O, "validation_v_hip_api_trace.csv"))))
slow = [r for r in api if (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) > 5e6]
print("API calls longer than 5 ms:")
for r in slow[:20]:
    print(" ", r["Function"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "ms at", int(r["Start_Timestamp"]))
PY
