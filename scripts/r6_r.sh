#!/bin/bash
# Round 6, visit r: the quadratic duplicate check out of the host's way (ensure_roles): host time of a step, streaming loops
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06r
mkdir -p $O
cd $R
python scripts/r6_hosttime.py 1024 2>&1 | grep -v amdgpu | tail -2
python scripts/r6_hosttime.py 128 2>&1 | grep -v amdgpu | tail -2
run() { # pairs, tag, extra args
  timeout 300 python bench.py --pairs $1 --steps 20 --warmup 3 --loop-only "${@:3}" > $O/loop_$1_$2.log 2>&1 < /dev/null
  echo "$1 $2: $(grep -o '"ms_per_step": [0-9.]*' $O/loop_$1_$2.log | tail -1) $(grep -o '"ms_per_step_one_lane": [0-9.a-z]*' $O/loop_$1_$2.log | tail -1)"
}
for rep in 1 2 3; do
  run 1024 auto_$rep
  run 512 auto_$rep
  run 256 auto_$rep
  run 128 auto_$rep
  run 64 auto_$rep
done
