#!/bin/bash
# Round 6, visit i: phase clocks of the sweep's tail (DVO_TAIL_CLOCKS build)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06i
mkdir -p $O
cd $R
export DVO_HIP_LIBRARY=$R/scripts/ubench/_build/tailclk/libdvo_hip.so
for n in 128 1024; do timeout 120 python scripts/r6_tailclk.py $n > $O/tailclk_$n.txt 2>&1 < /dev/null; cat $O/tailclk_$n.txt | tail -8; done
