#!/bin/bash
# Round 6, visit g: the pivoted 6 x 6 solve out of line (solver step 184 -> 168 registers, three workgroups per compute unit) against the base build
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06g
mkdir -p $O
cd $R
BASE=$R/scripts/ubench/_build/base/libdvo_hip.so
run() { # lib tag, pairs, tag, extra args
  if [ "$1" = base ]; then export DVO_HIP_LIBRARY=$BASE; else unset DVO_HIP_LIBRARY; fi
  timeout 300 python bench.py --pairs $2 --steps 20 --warmup 3 --loop-only "${@:4}" > $O/loop_$1_$2_$3.log 2>&1 < /dev/null
  echo "$1 $2 $3: $(grep -o '"ms_per_step": [0-9.]*' $O/loop_$1_$2_$3.log | tail -1)"
}
for rep in 1 2 3; do
  for lib in base new; do
    for n in 1024 512 256 128 64 16; do run $lib $n r$rep; done
  done
done
unset DVO_HIP_LIBRARY
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
