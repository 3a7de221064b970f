#!/bin/bash
# A/B of two builds of libdvo_hip.so on ONE box through bench.py's streaming loop (round 6's way of measuring a change; run under gpurun):
#   scripts/build_base.sh <commit>                      # here: the base build into scripts/ubench/_build/base/
#   gpurun -- 'bash scripts/r6_ab_loop.sh "1024 512 128" 3 [bench.py arguments]'
# prints ms per step, base and new alternated `reps` times per batch size.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ab_loop
mkdir -p $O
cd $R
BASE=$R/scripts/ubench/_build/base/libdvo_hip.so
SIZES=${1:-"1024 128"}; REPS=${2:-3}; shift 2
for rep in $(seq 1 $REPS); do
  for lib in base new; do
    for n in $SIZES; do
      if [ "$lib" = base ]; then export DVO_HIP_LIBRARY=$BASE; else unset DVO_HIP_LIBRARY; fi
      timeout 300 python bench.py --pairs $n --steps 20 --warmup 3 --loop-only "$@" > $O/loop_${lib}_${n}_$rep.log 2>&1 < /dev/null
      echo "$lib $n r$rep: $(grep -o '"ms_per_step": [0-9.]*' $O/loop_${lib}_${n}_$rep.log | tail -1)"
    done
  done
done
