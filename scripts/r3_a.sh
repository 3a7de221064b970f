#!/bin/bash
# Round 3, GPU visit A: what host is this (the reference's _mm_rcp_ps is CPU-specific), baseline of the round's first commit.
export TMPDIR=/tmp
mkdir -p gpurun_out/r3a
O=gpurun_out/r3a
lscpu > $O/lscpu.txt 2>&1
gcc -O2 -msse2 scripts/ubench/rcp_probe.c -o /tmp/rcp_probe && /tmp/rcp_probe > $O/rcp_probe.txt 2>&1
cat $O/rcp_probe.txt; grep -m1 "Model name" $O/lscpu.txt
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
echo "== pytest -m gpu"; SECONDS=0; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? in $SECONDS s"; tail -5 $O/pytest_gpu.log
echo "== bench"; SECONDS=0; timeout 900 python bench.py > $O/bench.log 2>$O/bench.err; echo "bench rc=$? in $SECONDS s"; tail -1 $O/bench.log | cut -c1-1500; tail -3 $O/bench.err
