#!/bin/bash
# Round 6, visit c: the Gauss-Newton step in the sweep's launch (solver_step.h, option sweep_tail) -- bit identity with the two-launch form,
# then the streaming loop with and without it at 128 / 256 / 512 / 1024 pairs per step (alternated on this box)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06c
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_tail.py tests/test_gpu_coarse.py -m gpu -q -x --timeout 600 > $O/pytest_tail.log 2>&1; echo "pytest tail rc=$?"; tail -15 $O/pytest_tail.log
for rep in 1 2; do
for n in 1024 512 256 128 64; do
  for opt in "sweep_tail=0" "sweep_tail=1"; do
    timeout 300 python bench.py --pairs $n --steps 20 --warmup 3 --loop-only --option $opt > $O/loop_${n}_${opt}_$rep.log 2>&1
    echo "$n $opt: $(tail -1 $O/loop_${n}_${opt}_$rep.log | grep -o '"ms_per_step": [0-9.]*')"
  done
done
done
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
