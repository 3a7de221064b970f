#!/bin/bash
# Round 4 evidence run: PMC traffic of the finest-level sweep (1024 pairs per launch) -> profiles/pmc_finest_kernel.json, the bench
# line, the rocprofv3 kernel trace of the same command.  Everything under gpurun_out/r04/.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
PMC_PAIRS=1024 bash scripts/pmc.sh > $O/pmc.log 2>&1; tail -3 $O/pmc.log | cut -c1-400
cp gpurun_out/pmc/pmc_finest_kernel.json profiles/pmc_finest_kernel.json 2>/dev/null; cp gpurun_out/pmc/pmc_finest_kernel.json $O/ 2>/dev/null
timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-600 $O/bench.json; tail -3 $O/bench.err
if [ "${DO_TRACE:-1}" = "1" ]; then
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-from-host --no-scaling-model --no-ref-compat > $O/prof_bench.log 2>&1 ); echo "rocprof rc=$?"
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/kernel_stats.py "$f" > $O/kernel_stats_insitu.txt && head -24 $O/kernel_stats_insitu.txt
s=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $O/bench_kernel_stats.csv
m=$(find $O/prof -name "*marker*stats*.csv" | head -1); [ -n "$m" ] && cp "$m" $O/bench_marker_stats.csv
rm -rf $O/prof
fi
