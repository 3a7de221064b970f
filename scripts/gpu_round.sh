#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, tuning sweep, rocprofv3 kernel trace.  Everything is wrapped
# in `timeout`; every log lands under gpurun_out/.
export TMPDIR=/tmp
mkdir -p gpurun_out
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > gpurun_out/device.txt
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 > gpurun_out/bench.log 2>gpurun_out/bench.err; echo "bench rc=$?"; tail -2 gpurun_out/bench.log; tail -3 gpurun_out/bench.err
if [ "${DO_SWEEP:-1}" = "1" ]; then echo "== sweep"; timeout 900 python scripts/sweep.py ${SWEEP_N:-128} > gpurun_out/sweep.log 2>&1; echo "sweep rc=$?"; tail -40 gpurun_out/sweep.log; fi
if [ "${DO_PROF:-1}" = "1" ]; then
  echo "== rocprofv3"
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1 ); echo "rocprof rc=$?"
  find gpurun_out/prof -name "*stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"; true
fi
