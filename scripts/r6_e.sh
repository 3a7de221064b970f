#!/bin/bash
# Round 6, visit e: tests of the fixed coarse kernel and the tail; streaming loop against the background build's cap, the solver step's
# occupancy, the coarse kernel after its fix
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06e
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_tail.py tests/test_gpu_coarse.py -m gpu -q -x --timeout 600 > $O/pytest_tail.log 2>&1 < /dev/null; echo "pytest tail+coarse rc=$?"; tail -4 $O/pytest_tail.log
run() { # pairs, tag, extra args
  timeout 300 python bench.py --pairs $1 --steps 20 --warmup 3 --loop-only "${@:3}" > $O/loop_$1_$2.log 2>&1 < /dev/null
  echo "$1 $2: $(grep -o '"ms_per_step": [0-9.]*' $O/loop_$1_$2.log | tail -1)"
}
for rep in 1 2; do
  run 1024 base_$rep
  run 1024 bw384_$rep --build-workgroups 384
  run 1024 bw512_$rep --build-workgroups 512
  run 1024 bw1024_$rep --build-workgroups 1024
  run 1024 occ3_$rep --option solver_occupancy=3
  run 1024 occ4_$rep --option solver_occupancy=4
  run 1024 coarse_$rep --option coarse=1
  run 128 base_$rep
  run 128 occ3_$rep --option solver_occupancy=3
  run 128 occ4_$rep --option solver_occupancy=4
  run 128 coarse_$rep --option coarse=1
  run 512 base_$rep
  run 512 bw512_$rep --build-workgroups 512
  run 512 occ4_$rep --option solver_occupancy=4
done
