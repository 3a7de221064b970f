#!/bin/bash
# LDS bank-conflict and array-busy cycles of the finest-level sweep for one build of the library (rocprofv3 --pmc, --kernel-trace only):
#   usage: lds_conflicts.sh <library> [variant] [pairs]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LIB=$1; V=${2:-8}; N=${3:-256}
O=$R/gpurun_out/ldsc
rm -rf $O; mkdir -p $O
cd /tmp
DVO_HIP_LIBRARY=$R/$LIB DVO_VARIANT=$V timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/e -o e -- python $R/scripts/kernel_driver.py $N 0 3 > $O/e.log 2>&1
python - $O $LIB <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/e/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        if 'sweep_fast' in row['Kernel_Name'] or 'sweep_window' in row['Kernel_Name']:
            agg[row['Counter_Name']].append(float(row['Counter_Value']))
out = {c: (lambda v: sum(x for x in v if x > 0.5 * max(v)) / max(1, sum(1 for x in v if x > 0.5 * max(v))))(v) for c, v in agg.items()}
print(sys.argv[2], {k: "%.4g" % v for k, v in out.items()}, "conflict share of LDS cycles %.3f" % (out.get('SQ_LDS_BANK_CONFLICT', 0) / max(out.get('SQ_LDS_IDX_ACTIVE', 1), 1)))
PY
rm -rf $O
