#!/bin/bash
# Round 6: kernel timeline of a 1024-pair streaming step with the slow lane (option "overlap_tails")
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06ovt
mkdir -p $O
cd $R
for f in ${FRACTIONS:-8 3}; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof$f -o t -- python $R/bench.py --pairs ${PAIRS:-1024} --steps 6 --warmup 3 --loop-only --lanes 1 --option overlap_tails=1 --option overlap_fraction=$f > $O/run$f.log 2>&1 < /dev/null )
  t=$(find $O/prof$f -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python scripts/step_breakdown.py "$t" 3 > $O/step_breakdown_$f.txt && python scripts/r5_step_timeline.py "$t" 2 > $O/timeline_$f.txt
  head -24 $O/step_breakdown_$f.txt
  rm -rf $O/prof$f
done
