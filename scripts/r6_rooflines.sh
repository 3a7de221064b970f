#!/bin/bash
# Round 6: per-kernel rooflines of the bench loop -- kernel trace (in-situ durations) + PMC passes (separate runs, --kernel-trace only).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06/roof
PAIRS=${ROOF_PAIRS:-1024}
mkdir -p $O
cd /tmp
CMD="python $R/bench.py --pairs $PAIRS --steps 4 --warmup 2 --loop-only --lanes 1"
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- $CMD > $O/trace.log 2>&1; echo "trace rc=$?"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o t -- $CMD > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o t -- $CMD > $O/write.log 2>&1; echo "write rc=$?"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o t -- $CMD > $O/sq.log 2>&1; echo "sq rc=$?"
cd $R
python scripts/kernel_rooflines.py $O $PAIRS $O/kernel_rooflines.json > $O/kernel_rooflines.md 2> $O/kernel_rooflines.err; echo "table rc=$?"; cat $O/kernel_rooflines.md | cut -c1-330; tail -3 $O/kernel_rooflines.err
f=$(find $O/trace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/kernel_stats.py "$f" > $O/kernel_stats_insitu.txt
[ -n "$f" ] && python scripts/step_breakdown.py "$f" 3 > $O/step_breakdown.txt 2>/dev/null
# the raw per-dispatch CSVs are large (gpurun_out/ merges back at most 64 MiB): only the summaries stay
find $O -name "*.csv" -size +2M -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete 2>/dev/null
