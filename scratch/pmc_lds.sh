#!/bin/bash
# LDS bank-conflict share of the leaf kernels for a given library: bash scratch/pmc_lds.sh <tag> <lib.so>
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_lds_$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p -o t -- python3 $R/scratch/run_with_lib.py $R/$2 $R/scratch/fit_loop.py c2 3 timing=0 > $OUT/log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in f:
    for row in csv.DictReader(open(fn)):
        k = row["Kernel_Name"].split("(")[0][:40]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in agg.items():
    if v.get("SQ_LDS_IDX_ACTIVE", 0) > 0 and ("potf2" in k or "trsm" in k):
        print("%-42s conflict cycles / LDS active cycles = %.3f" % (k, v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]))
PY
rm -rf $OUT/p
