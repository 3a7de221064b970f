#!/bin/bash
# Round 6, visit m: the streaming loop over lanes (dvo_stream_lanes_*): tests, then bench.py's loop with 1 / 2 / 3 lanes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06m
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_stream_lanes.py tests/test_gpu_groups.py -m gpu -q -x --timeout 600 > $O/pytest_lanes.log 2>&1 < /dev/null; echo "pytest lanes rc=$?"; tail -6 $O/pytest_lanes.log
run() { # pairs, tag, extra args
  timeout 300 python bench.py --pairs $1 --steps 20 --warmup 3 --loop-only "${@:3}" > $O/loop_$1_$2.log 2>&1 < /dev/null
  echo "$1 $2: $(grep -o '"ms_per_step": [0-9.]*' $O/loop_$1_$2.log | tail -1) $(grep -o '"ms_per_step_one_lane": [0-9.a-z]*' $O/loop_$1_$2.log | tail -1)"
}
for rep in 1 2; do
  run 1024 auto_$rep --option batch_groups=1
  run 1024 l2_$rep --lanes 2 --option batch_groups=1
  run 1024 l3_$rep --lanes 3 --option batch_groups=1
  run 1024 l4_$rep --lanes 4 --option batch_groups=1
  run 512 auto_$rep --option batch_groups=1
  run 768 auto_$rep --option batch_groups=1
  run 256 l2_$rep --lanes 2 --option batch_groups=1
  run 1024 autogroups_$rep
done
