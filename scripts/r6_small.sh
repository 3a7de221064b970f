#!/bin/bash
# Round 6: the small-level sweep (align_small.hip) -- tests, then the streaming loop with and without it
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06small
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_small.py -m gpu -q -x --timeout 600 > $O/pytest_small.log 2>&1 < /dev/null; echo "pytest small rc=$?"; tail -12 $O/pytest_small.log
bash scripts/r6_options_sweep.sh "1024 512 256 128" "small_sweep=0 small_sweep=1" 3
bash scripts/r6_options_sweep.sh "1024" "small_tiles=2 small_tiles=3 small_tiles=4 small_tiles=6" 2
bash scripts/r6_options_sweep.sh "128" "small_tiles=3 small_tiles=4 small_tiles=6 small_tiles=8" 2
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
