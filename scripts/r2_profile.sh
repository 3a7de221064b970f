#!/bin/bash
# Round-2 evidence run on the GPU box: bench line, rocprofv3 kernel trace + markers of the same command, PMC traffic and
# utilisation passes of the finest-level sweep.  Everything under gpurun_out/r02/.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-from-host > $O/prof_bench.log 2>&1 ); echo "rocprof rc=$?"
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/kernel_stats.py "$f" > $O/kernel_stats_insitu.txt && head -20 $O/kernel_stats_insitu.txt
s=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $O/bench_kernel_stats.csv
m=$(find $O/prof -name "*marker*stats*.csv" | head -1); [ -n "$m" ] && cp "$m" $O/bench_marker_stats.csv && head -12 "$m"
PMC_PAIRS=1024 bash scripts/pmc.sh > $O/pmc.log 2>&1; tail -4 $O/pmc.log; cp gpurun_out/pmc/pmc_finest_kernel.json $O/ 2>/dev/null
PMC_PAIRS=128 bash scripts/pmc2.sh > $O/pmc2.log 2>&1; grep -E "per-dispatch" $O/pmc2.log | head -60
rm -rf $O/prof
