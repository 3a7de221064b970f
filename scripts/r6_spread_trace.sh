#!/bin/bash
# Round 6: kernel totals of the wide-spread batch (scripts/r6_spread.py) on the synchronous chain and with the slow lane
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06sp
mkdir -p $O
cd $R
for mode in sync lane; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$mode -o t -- python $R/scripts/r6_spread.py 1024 8 --only $mode > $O/run_$mode.log 2>&1 < /dev/null )
  tail -1 $O/run_$mode.log
  t=$(find $O/prof_$mode -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python scripts/kernel_stats.py "$t" > $O/kernel_stats_$mode.txt && head -28 $O/kernel_stats_$mode.txt | cut -c1-190
  rm -rf $O/prof_$mode
done
