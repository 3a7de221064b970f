#!/bin/bash
# Round 6: the next batch's ingest carried out when the launch chain reaches level 1 (option "defer_ingest_pixels"), alternated with today's order
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
run() {  # label, pairs, env..., -- options
  local label=$1 pairs=$2; shift 2
  env "$@" timeout 200 python bench.py --pairs $pairs --steps 12 --warmup 3 --loop-only --lanes 1 $OPTS 2> /dev/null < /dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('pairs $pairs', '$label', ' ms_per_step', j['ms_per_step'])
"
}
for pairs in ${PAIRS:-1024 512}; do
  for rep in 1 2 3; do
    OPTS="" run "today" $pairs DVO_X=1
    OPTS="--option defer_ingest_pixels=76800" run "ingest at 320x240" $pairs DVO_STREAM_DEFER_MAX=100000
    OPTS="--option defer_ingest_pixels=19200" run "ingest at 160x120" $pairs DVO_STREAM_DEFER_MAX=100000
    OPTS="--option defer_ingest_pixels=307200" run "ingest at 640x480" $pairs DVO_STREAM_DEFER_MAX=100000
  done
done
