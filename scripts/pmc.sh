#!/bin/bash
# PMC passes for the finest-level reduce kernel (separate runs per counter group; --kernel-trace only, as required).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc/$name -o $name -- python $R/scripts/kernel_driver.py ${PMC_PAIRS:-1024} 0 3 > $R/gpurun_out/pmc/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
# LDS array cycles and bank conflicts, scalar and matrix pipes (SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT are cycles summed over the CUs)
run lds1 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE
run lds2 SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CU_CYCLES
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc/*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:60]
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[(k,row['Counter_Name'])] += 1
    for k in agg:
        if 'residual_reduce' in k or 'sweep_window' in k or 'sweep_fast' in k:
            print(f.split('/')[-2], k)
            for c, v in agg[k].items(): print("    %-32s total %.4g  per-dispatch %.4g  (n=%d)" % (c, v, v/cnt[(k,c)], cnt[(k,c)]))
# summary consumed by bench.py (roofline.traffic): FETCH_SIZE is in KB and reports half of a wide coalesced stream on gfx950
import json
vals = {}
for f in glob.glob('gpurun_out/pmc/*/*counter_collection.csv'):
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if ('residual_reduce' in row['Kernel_Name'] or 'sweep_window' in row['Kernel_Name'] or 'sweep_fast' in row['Kernel_Name']) and row['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            per[row['Counter_Name']].append(float(row['Counter_Value']))
    for c, v in per.items():
        big = [x for x in v if x > 0.5 * max(v)]
        vals[c] = sum(big) / len(big)
import hashlib, os
h = hashlib.sha256()
for f in ("align_fast.hip", "gram_f16.h", "sweep_parts.h", "pixel_math.h"):
    h.update(open(os.path.join("dvo_slam_amd", "csrc", f), "rb").read())
pairs = int(os.environ.get("PMC_PAIRS", "1024"))
if 'FETCH_SIZE' in vals and 'WRITE_SIZE' in vals:
    rec = dict(pairs_per_launch=pairs, level=0, kernel_source_sha256=h.hexdigest(), fetch_size_kb=vals['FETCH_SIZE'], write_size_kb=vals['WRITE_SIZE'],
               fetch_correction=2.0, traffic_bytes_per_launch=(2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0,
               algorithmic_bytes_per_launch=40.0 * 640 * 480 * pairs,
               note="rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (scripts/pmc.sh); FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B)")
    json.dump(rec, open('gpurun_out/pmc/pmc_finest_kernel.json', 'w'), indent=1)
    print(rec)
PY
# the raw per-dispatch CSVs are large (gpurun_out/ merges back at most 64 MiB): only the summaries stay
find $R/gpurun_out/pmc* -name "*kernel_trace.csv" -delete 2>/dev/null; find $R/gpurun_out/pmc* -name "*agent_info.csv" -delete 2>/dev/null
