#!/bin/bash
# Round 6, visit n: lanes -- depth and background-build cap
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06n
mkdir -p $O
cd $R
run() { # pairs, tag, extra args
  timeout 300 python bench.py --pairs $1 --steps 20 --warmup 3 --loop-only "${@:3}" > $O/loop_$1_$2.log 2>&1 < /dev/null
  echo "$1 $2: $(grep -o '"ms_per_step": [0-9.]*' $O/loop_$1_$2.log | tail -1) $(grep -o '"ms_per_step_one_lane": [0-9.a-z]*' $O/loop_$1_$2.log | tail -1)"
}
for rep in 1 2; do
  run 1024 d1_$rep --lane-depth 1
  run 1024 d2_$rep --lane-depth 2
  run 1024 d3_$rep --lane-depth 3
  run 1024 d4_$rep --lane-depth 4
  run 1024 cap128_$rep --build-workgroups 128
  run 1024 cap256_$rep --build-workgroups 256
  run 1024 cap0_$rep --build-workgroups 0
done
