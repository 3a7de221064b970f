#!/bin/bash
# kernel trace of the bench loop -> per-step breakdown (gpurun_out/r04/step_breakdown.txt) and per-(kernel, grid) statistics
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-from-host --no-scaling-model --no-ref-compat $R4_BENCH_ARGS > $O/prof_bench.log 2>&1 ); echo "rocprof rc=$?"
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python $R/scripts/step_breakdown.py "$f" 3 > $O/step_breakdown.txt; cat $O/step_breakdown.txt
python $R/scripts/kernel_stats.py "$f" > $O/kernel_stats_insitu.txt
python $R/scripts/step_sequence.py "$f" > $O/step_sequence.txt
rm -rf $O/prof
