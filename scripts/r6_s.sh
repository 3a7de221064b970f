#!/bin/bash
# Round 6, visit s: the deferred ingest for large batches again, now that the ingest's host time is halved
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06s
mkdir -p $O
cd $R
run() { # pairs, tag, extra args
  timeout 300 python bench.py --pairs $1 --steps 20 --warmup 3 --loop-only "${@:3}" > $O/loop_$1_$2.log 2>&1 < /dev/null
  echo "$1 $2: $(grep -o '"ms_per_step": [0-9.]*' $O/loop_$1_$2.log | tail -1) $(grep -o '"ms_per_step_one_lane": [0-9.a-z]*' $O/loop_$1_$2.log | tail -1)"
}
for rep in 1 2 3; do
  for n in 1024 512 384; do
    run $n early_$rep --lanes 1
    DVO_STREAM_DEFER_MAX=4096 run $n deferred_$rep --lanes 1
  done
  run 1024 lanes2_$rep --lanes 2
  DVO_STREAM_DEFER_MAX=4096 run 1024 lanes2_deferred_$rep --lanes 2
done
