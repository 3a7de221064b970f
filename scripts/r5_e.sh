#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05e
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "two_wavefront or packed_residuals or batch_equals or large_ragged or deterministic_mode or device_ingest" > $O/t_parity.log 2>&1; echo "parity subset rc=$?"; tail -3 $O/t_parity.log
timeout 900 python scripts/r5_midsize.py 128 40 "" "table_cache=0" "fused_ll_pixels=76800" > $O/midsize_128.txt 2>&1; grep -v amdgpu.ids $O/midsize_128.txt
timeout 900 python scripts/r5_midsize.py 1024 8 "" "table_cache=0" > $O/midsize_1024.txt 2>&1; grep -v amdgpu.ids $O/midsize_1024.txt
timeout 900 python scripts/r5_midsize.py 16 60 "" "table_cache=0" > $O/midsize_16.txt 2>&1; grep -v amdgpu.ids $O/midsize_16.txt
