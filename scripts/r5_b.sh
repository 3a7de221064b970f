#!/bin/bash
# Round 5, second visit: ref_compat with the table in LDS, the mid-size step under options.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05b
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "default_schedule or range_guard" > $O/t_parity.log 2>&1; echo "parity subset rc=$?"; grep -E "passed|failed|error" $O/t_parity.log | tail -3
timeout 900 python -m pytest tests/test_gpu_replay.py -x -q -s -k "single_matches" > $O/t_replay.log 2>&1; echo "replay subset rc=$?"; grep -E "passed|failed|error" $O/t_replay.log | tail -3
timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-from-host --no-guard-stress > $O/bench_b.json 2> $O/bench_b.err; echo "bench rc=$?"; tail -3 $O/bench_b.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05b/bench_b.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms/step", j["ms_per_step"], "kernel_ms", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"])
print("ref_compat", j["ref_compat"] and (j["ref_compat"]["value"], j["ref_compat"]["ms_per_step"]))
print("scaling", j["scaling_model"] and (j["scaling_model"]["ms_per_step_at_pairs_per_gpu"], j["scaling_model"]["predicted_efficiency"]))
PY
timeout 600 python bench.py --steps 8 --warmup 2 --loop-only --option ref_compat=2 > $O/bench_compat_mem.json 2>&1; tail -1 $O/bench_compat_mem.json | cut -c1-300
timeout 900 python scripts/r5_midsize.py 128 40 "" "build_workgroups=64" "build_workgroups=128" "build_workgroups=512" "build_workgroups=0" "fused_ll_pixels=19200" "resident_group=1" "ll_blocks=16" "solver_waves=2" > $O/midsize_128.txt 2>&1; cat $O/midsize_128.txt
timeout 600 python scripts/r5_midsize.py 256 20 "" "build_workgroups=128" "build_workgroups=512" > $O/midsize_256.txt 2>&1; cat $O/midsize_256.txt
