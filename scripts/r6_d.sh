#!/bin/bash
# Round 6, visit d: why the sweep with the solver step in its tail is slower -- kernel traces of the streaming loop with the tail on / off
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06d
mkdir -p $O
cd /tmp
for n in 1024 128; do
  for t in 0 1; do
    timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_${n}_tail$t -o t -- python $R/bench.py --pairs $n --steps 6 --warmup 2 --loop-only --option sweep_tail=$t > $O/trace_${n}_tail$t.log 2>&1 < /dev/null
    f=$(ls $O/trace_${n}_tail$t/*/*kernel_trace.csv 2>/dev/null | head -1)
    echo "== $n pairs tail $t: $(grep -o '"ms_per_step": [0-9.]*' $O/trace_${n}_tail$t.log | tail -1)"
    if [ -n "$f" ]; then timeout 60 python $R/scripts/step_breakdown.py "$f" 3 > $O/breakdown_${n}_tail$t.txt 2>&1 < /dev/null; head -16 $O/breakdown_${n}_tail$t.txt | cut -c1-140; rm -rf $O/trace_${n}_tail$t; fi
  done
done
