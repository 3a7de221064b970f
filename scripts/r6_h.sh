#!/bin/bash
# Round 6, visit h: the wide half of the step in the sweep's tail + one-wavefront serial step (sweep_tail 2) against the two-launch form
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06h
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_tail.py -m gpu -q -x --timeout 600 > $O/pytest_tail.log 2>&1 < /dev/null; echo "pytest tail rc=$?"; tail -6 $O/pytest_tail.log
run() { # pairs, tag, extra args
  timeout 300 python bench.py --pairs $1 --steps 20 --warmup 3 --loop-only "${@:3}" > $O/loop_$1_$2.log 2>&1 < /dev/null
  echo "$1 $2: $(grep -o '"ms_per_step": [0-9.]*' $O/loop_$1_$2.log | tail -1)"
}
for rep in 1 2; do
  for n in 1024 512 256 128; do
    run $n tail0_$rep --option sweep_tail=0
    run $n tail2_$rep --option sweep_tail=2
  done
done
cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_1024_tail2 -o t -- python $R/bench.py --pairs 1024 --steps 6 --warmup 2 --loop-only --option sweep_tail=2 > $O/trace_1024_tail2.log 2>&1 < /dev/null
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_128_tail2 -o t -- python $R/bench.py --pairs 128 --steps 6 --warmup 2 --loop-only --option sweep_tail=2 > $O/trace_128_tail2.log 2>&1 < /dev/null
ls $O/trace_1024_tail2 | head
