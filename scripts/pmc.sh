#!/bin/bash
# PMC passes for the finest-level reduce kernel (separate runs per counter group; --kernel-trace only, as required).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc/$name -o $name -- python $R/scripts/kernel_driver.py 128 0 5 > $R/gpurun_out/pmc/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc/*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:60]
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[(k,row['Counter_Name'])] += 1
    for k in agg:
        if 'residual_reduce' in k:
            print(f.split('/')[-2], k)
            for c, v in agg[k].items(): print("    %-32s total %.4g  per-dispatch %.4g  (n=%d)" % (c, v, v/cnt[(k,c)], cnt[(k,c)]))
PY
