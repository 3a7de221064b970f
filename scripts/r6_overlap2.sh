#!/bin/bash
# Round 6: overlap_tails at several split thresholds (overlap_fraction), alternated with the synchronous chain
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
run() {  # pairs, extra options...
  local pairs=$1; shift
  timeout 200 python bench.py --pairs $pairs --steps 12 --warmup 3 --loop-only --lanes 1 "$@" 2> /dev/null < /dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('pairs $pairs', '$*', ' ms_per_step', j['ms_per_step'], j.get('counters'))
"
}
for pairs in ${PAIRS:-1024}; do
  for rep in 1 2; do
    run $pairs --option overlap_tails=0
    for f in ${FRACTIONS:-2 3 4 6}; do
      run $pairs --option overlap_tails=1 --option overlap_fraction=$f
    done
  done
done
