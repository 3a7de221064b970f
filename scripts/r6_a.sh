#!/bin/bash
# Round 6, visit a: baseline of HEAD on this round's boxes (GPU tests, bench line as the driver runs it, 128-pair loop)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06a
mkdir -p $O
cd $R
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | grep -m2 -iE "instinct|MI3|gfx9" > $O/device.txt; lscpu | grep -m1 "Model name" >> $O/device.txt
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-600 $O/bench.json; tail -2 $O/bench.err
for n in 128 256 512; do timeout 300 python bench.py --pairs $n --steps 20 --warmup 3 --loop-only > $O/loop$n.log 2>&1; tail -1 $O/loop$n.log | cut -c1-300; done
