#!/bin/bash
# Utilisation counters of the finest-level sweep kernel (separate rocprofv3 --pmc passes, --kernel-trace only).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc2
cd /tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc2/$name -o $name -- python $R/scripts/kernel_driver.py ${PMC_PAIRS:-128} 0 3 > $R/gpurun_out/pmc2/$name.log 2>&1; echo "$name rc=$?"; tail -2 $R/gpurun_out/pmc2/$name.log | cut -c1-300; }
run a GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS
run c TA_TA_BUSY_sum TA_BUSY_avr TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum
# (a pass with the TCP_*_STALL_CYCLES counters never returned on this pool: left out)
run e SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
run f TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc2/*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:60]
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[(k,row['Counter_Name'])] += 1
    for k in agg:
        if 'residual_reduce' in k:
            for c, v in agg[k].items(): print("%-6s %-44s per-dispatch %.5g  (n=%d)" % (f.split('/')[-2], c, v/cnt[(k,c)], cnt[(k,c)]))
PY
