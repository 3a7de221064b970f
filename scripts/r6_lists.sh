#!/bin/bash
# Round 6: active-pair lists (option "tail_lists") -- the tests, then the streaming loop with and without, alternated
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06ov
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_overlap.py -x -q > $O/pytest.log 2>&1 < /dev/null; tail -5 $O/pytest.log | cut -c1-300
for pairs in ${PAIRS:-1024 512}; do
  for rep in 1 2 3 4; do
    for opt in 0 1; do
      timeout 200 python bench.py --pairs $pairs --steps 12 --warmup 3 --loop-only --lanes 1 --option tail_lists=$opt 2> /dev/null < /dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('pairs $pairs tail_lists $opt  ms_per_step', j['ms_per_step'])
"
    done
  done
done
