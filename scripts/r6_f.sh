#!/bin/bash
# Round 6, visit f: the ingest with two strips in flight per wavefront against the base build (scripts/build_base.sh HEAD), alternated
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06f
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "plane or ingest or pyramid or role or select or strip" > $O/pytest_planes.log 2>&1 < /dev/null; echo "pytest planes rc=$?"; tail -3 $O/pytest_planes.log
BASE=$R/scripts/ubench/_build/base/libdvo_hip.so
run() { # lib tag, pairs, tag, extra args
  if [ "$1" = base ]; then export DVO_HIP_LIBRARY=$BASE; else unset DVO_HIP_LIBRARY; fi
  timeout 300 python bench.py --pairs $2 --steps 20 --warmup 3 --loop-only "${@:4}" > $O/loop_$1_$2_$3.log 2>&1 < /dev/null
  echo "$1 $2 $3: $(grep -o '"ms_per_step": [0-9.]*' $O/loop_$1_$2_$3.log | tail -1)"
}
for rep in 1 2; do
  for lib in base new; do
    run $lib 1024 bw256_$rep
    run $lib 1024 bw384_$rep --build-workgroups 384
    run $lib 1024 bw512_$rep --build-workgroups 512
    run $lib 512 bw256_$rep
    run $lib 128 bw256_$rep
  done
done
for lib in base new; do
  if [ "$lib" = base ]; then export DVO_HIP_LIBRARY=$BASE; else unset DVO_HIP_LIBRARY; fi
  timeout 200 python scripts/build_rate.py > $O/build_rate_$lib.txt 2>&1 < /dev/null; echo "== build rate $lib"; tail -4 $O/build_rate_$lib.txt
done
