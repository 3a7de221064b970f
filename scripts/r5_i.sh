#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05i
mkdir -p $O
cd /tmp
for n in 128; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o mid$n -- python $R/bench.py --pairs $n --steps 8 --warmup 3 --loop-only > $O/run$n.log 2>&1; tail -1 $O/run$n.log | cut -c1-200
f=$(find $O/prof -name "mid${n}_kernel_trace.csv" | head -1)
python $R/scripts/r5_step_timeline.py "$f" 2 > $O/timeline_$n.txt; tail -1 $O/timeline_$n.txt
python $R/scripts/step_breakdown.py "$f" 4 > $O/breakdown_$n.txt; head -24 $O/breakdown_$n.txt
done
rm -rf $O/prof
