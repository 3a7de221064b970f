#!/bin/bash
# Round 5, first GPU visit: the new / changed tests, a bench line (ref_compat leg, guard stress, scaling model), per-kernel rooflines.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > $O/device.txt; lscpu | grep -m1 "Model name" >> $O/device.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "default_schedule or range_guard or contracted_sweep_against or single_linearisation or contracted_sweep_random or golden" > $O/t_parity.log 2>&1; echo "parity subset rc=$?"; grep -E "passed|failed|error" $O/t_parity.log | tail -3
timeout 900 python -m pytest tests/test_gpu_replay.py -x -q -s -k "single_matches" > $O/t_replay.log 2>&1; echo "replay subset rc=$?"; grep -E "passed|failed|error" $O/t_replay.log | tail -3
timeout 600 python -m pytest tests/test_parallel.py -x -q -m gpu -k "launches_its_own" > $O/t_parallel.log 2>&1; echo "self-launch rc=$?"; tail -3 $O/t_parallel.log
timeout 600 python -m pytest tests/test_dropin.py -x -q -s -m gpu -k "remaining_public" > $O/t_dropin.log 2>&1; echo "dropin subset rc=$?"; grep -E "passed|failed|error|ref_compat" $O/t_dropin.log | tail -5
timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-from-host > $O/bench_a.json 2> $O/bench_a.err; echo "bench rc=$?"; tail -3 $O/bench_a.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05/bench_a.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms/step", j["ms_per_step"], "kernel_ms", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"], "per level", j["roofline"]["per_level_kernel_ms"])
print("ref_compat", j["ref_compat"] and (j["ref_compat"]["value"], j["ref_compat"]["ms_per_step"]))
print("guard", j["f16_guard_stress"])
print("scaling", j["scaling_model"] and (j["scaling_model"]["ms_per_step_at_pairs_per_gpu"], j["scaling_model"]["predicted_efficiency"]))
print("latency", j["latency_ms"])
PY
if [ "${DO_ROOF:-1}" = "1" ]; then bash scripts/r5_rooflines.sh; fi
