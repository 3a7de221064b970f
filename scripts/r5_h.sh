#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05h
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log
timeout 900 python scripts/r5_midsize.py 128 40 "" "table_cache=0" > $O/midsize_128.txt 2>&1; grep -v amdgpu.ids $O/midsize_128.txt
timeout 900 python scripts/r5_midsize.py 1024 8 "" > $O/midsize_1024.txt 2>&1; grep -v amdgpu.ids $O/midsize_1024.txt
timeout 900 python scripts/r5_midsize.py 16 60 "" > $O/midsize_16.txt 2>&1; grep -v amdgpu.ids $O/midsize_16.txt
