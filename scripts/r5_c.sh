#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05c
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "finest_level_records" > $O/t_parity.log 2>&1; echo "parity subset rc=$?"; grep -E "passed|failed|error|worst over|finest level:" $O/t_parity.log | tail -4
timeout 600 python scripts/r5_compat.py 256 > $O/compat.txt 2>&1; cat $O/compat.txt | grep -v amdgpu.ids
