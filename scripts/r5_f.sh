#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05f
mkdir -p $O
cd $R
timeout 900 python scripts/r5_midsize.py 128 40 "" "table_cache=0" > $O/midsize_128.txt 2>&1; grep -v amdgpu.ids $O/midsize_128.txt
timeout 900 python scripts/r5_midsize.py 1024 8 "" "table_cache=0" > $O/midsize_1024.txt 2>&1; grep -v amdgpu.ids $O/midsize_1024.txt
