#!/bin/bash
# kernel trace of mid-size batches (launch path): the last 64-pair and 16-pair full matches -> gpurun_out/r03/midsize_*.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
for n in ${MID_SIZES:-64 16}; do
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_mid -o mid$n -- python $R/scripts/batch_sizes.py $n > $O/midsize_run$n.log 2>&1 )
f=$(find $O/prof_mid -name "mid${n}_kernel_trace.csv" | head -1)
python $R/scripts/match_timeline.py "$f" > $O/midsize_$n.txt; tail -1 $O/midsize_$n.txt; tail -4 $O/midsize_run$n.log | cut -c1-200
done
rm -rf $O/prof_mid
