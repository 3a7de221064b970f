#!/bin/bash
# Round 6: sweeps of existing schedule options on the streaming loop (usage under gpurun: bash scripts/r6_options_sweep.sh "<pairs>" "<opt=val> ..." reps)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06opts
mkdir -p $O
cd $R
SIZES=${1:-1024}; OPTS=${2:-"tail_speculation=0"}; REPS=${3:-2}
for rep in $(seq 1 $REPS); do
  for n in $SIZES; do
    for opt in $OPTS; do
      timeout 300 python bench.py --pairs $n --steps 20 --warmup 3 --loop-only --option $opt > $O/loop_${n}_${opt}_$rep.log 2>&1 < /dev/null
      echo "$n $opt r$rep: $(grep -o '"ms_per_step": [0-9.]*' $O/loop_${n}_${opt}_$rep.log | tail -1)"
    done
  done
done
