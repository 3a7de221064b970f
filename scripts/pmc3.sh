#!/bin/bash
# Utilisation counters of the finest-level sweep for one schedule variant (separate rocprofv3 --pmc passes, --kernel-trace only).
#   usage: pmc3.sh <variant> [pairs]      output: gpurun_out/pmc3_v<variant>/summary.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${1:-6}
N=${2:-128}
O=$R/gpurun_out/pmc3_v$V
mkdir -p $O
cd /tmp
run() { name=$1; shift; DVO_VARIANT=$V timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -o $name -- python $R/scripts/kernel_driver.py $N 0 3 > $O/$name.log 2>&1; echo "$name rc=$?"; }
run a GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS
run c TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_WAIT_ANY SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run e SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
run f TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
run g FETCH_SIZE
run h WRITE_SIZE
cd $R
python - $O <<'PY' > $O/summary.txt
import csv, glob, collections, sys
O = sys.argv[1]
for f in sorted(glob.glob(O + '/*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        agg[row['Kernel_Name'][:70]][row['Counter_Name']].append(float(row['Counter_Value']))
    for k in agg:
        if 'residual_reduce' in k or 'sweep_window' in k or 'sweep_fast' in k:
            for c, v in agg[k].items():
                big = [x for x in v if x > 0.5 * max(v)] if max(v) > 0 else v          # (the speculative no-op launches of the warm-up steps excluded)
                print("%-4s %-40s %-34s per-dispatch %.5g  (n=%d of %d)" % (f.split('/')[-2], k[9:49], c, sum(big) / len(big), len(big), len(v)))
PY
cat $O/summary.txt
# the raw per-dispatch CSVs are large (gpurun_out/ merges back at most 64 MiB): only the summaries stay
find $R/gpurun_out/pmc* -name "*kernel_trace.csv" -delete 2>/dev/null; find $R/gpurun_out/pmc* -name "*agent_info.csv" -delete 2>/dev/null
