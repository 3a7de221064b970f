#!/bin/bash
# Round 6, visit k: concurrent pair groups inside dvo_hip_match_batch (option batch_groups) -- tests, then the streaming loop
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06k
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_groups.py -m gpu -q -x --timeout 600 > $O/pytest_groups.log 2>&1 < /dev/null; echo "pytest groups rc=$?"; tail -6 $O/pytest_groups.log
run() { # pairs, tag, extra args
  timeout 300 python bench.py --pairs $1 --steps 20 --warmup 3 --loop-only "${@:3}" > $O/loop_$1_$2.log 2>&1 < /dev/null
  echo "$1 $2: $(grep -o '"ms_per_step": [0-9.]*' $O/loop_$1_$2.log | tail -1)"
}
for rep in 1 2; do
  run 1024 g1_$rep --option batch_groups=1
  run 1024 g2_$rep --option batch_groups=2
  run 1024 g3_$rep --option batch_groups=3
  run 1024 g4_$rep --option batch_groups=4
  run 1024 auto_$rep
  run 768 g1_$rep --option batch_groups=1
  run 768 g2_$rep --option batch_groups=2
  run 768 g3_$rep --option batch_groups=3
  run 512 g1_$rep --option batch_groups=1
  run 512 g2_$rep --option batch_groups=2
  run 384 g1_$rep --option batch_groups=1
  run 384 g2_$rep --option batch_groups=2
  run 256 g1_$rep --option batch_groups=1
  run 256 g2_$rep --option batch_groups=2
done
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
