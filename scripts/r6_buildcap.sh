#!/bin/bash
# Round 6: the cap on the background build's workgroups below one per compute unit, alternated
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for cap in ${CAPS:-256 192 128 224}; do
    timeout 200 python bench.py --pairs ${PAIRS:-1024} --steps 12 --warmup 3 --loop-only --lanes 1 --build-workgroups $cap 2> /dev/null < /dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('build cap $cap  ms_per_step', j['ms_per_step'])
"
  done
done
