#!/bin/bash
# Round 6 evidence run (everything under gpurun_out/r06/; the summaries to keep are copied into profiles/ by hand afterwards):
#   1. PMC traffic of the finest-level sweep at 1024 pairs per launch      -> pmc_finest_kernel.json (bench.py's roofline.traffic)
#   2. per-kernel rooflines of the bench loop (trace + PMC passes)          -> roof/kernel_rooflines.{md,json}, kernel_stats, step breakdown
#   3. the bench line as the driver runs it                                 -> bench.json
#   4. rocprofv3 kernel trace of the bench command (incl. the ref_compat leg) -> bench_kernel_stats_insitu.txt, bench_kernel_stats.csv
#   5. step breakdown and timeline of a 128-pair step                       -> step_breakdown_128_pairs.txt, timeline_128_pairs.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | grep -m2 -iE "instinct|MI3|gfx9" > $O/device.txt; lscpu | grep -m1 "Model name" >> $O/device.txt
PMC_PAIRS=1024 bash scripts/pmc.sh > $O/pmc.log 2>&1; tail -2 $O/pmc.log | cut -c1-300
cp gpurun_out/pmc/pmc_finest_kernel.json profiles/pmc_finest_kernel.json 2>/dev/null; cp gpurun_out/pmc/pmc_finest_kernel.json $O/ 2>/dev/null
bash scripts/r6_rooflines.sh > $O/rooflines.log 2>&1; tail -16 $O/rooflines.log | cut -c1-260
cp $O/roof/kernel_rooflines.json profiles/r06_kernel_rooflines.json 2>/dev/null; cp $O/roof/kernel_rooflines.md profiles/r06_kernel_rooflines.md 2>/dev/null
timeout 900 python bench.py --steps ${BENCH_STEPS:-20} --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json; tail -2 $O/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-from-host --no-scaling-model --no-guard-stress > $O/prof_bench.log 2>&1 ); echo "rocprof rc=$?"
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/kernel_stats.py "$f" > $O/bench_kernel_stats_insitu.txt && head -30 $O/bench_kernel_stats_insitu.txt | cut -c1-200
s=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $O/bench_kernel_stats.csv
m=$(find $O/prof -name "*marker*stats*.csv" | head -1); [ -n "$m" ] && cp "$m" $O/bench_marker_stats.csv
rm -rf $O/prof
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof128 -o mid -- python $R/bench.py --pairs 128 --steps 8 --warmup 3 --loop-only --lanes 1 > $O/run128.log 2>&1 )
f=$(find $O/prof128 -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/step_breakdown.py "$f" 4 > $O/step_breakdown_128_pairs.txt && python scripts/r5_step_timeline.py "$f" 2 > $O/timeline_128_pairs.txt; head -22 $O/step_breakdown_128_pairs.txt
rm -rf $O/prof128
