#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_gemm
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -o t -- python3 $R/scratch/gemm_one.py 7168 384 > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/p2 -o t -- python3 $R/scratch/gemm_one.py 7168 384 > $OUT/p2.log 2>&1
python3 - <<'P'
import csv, collections, glob, os
R=os.environ['GRAFT_REPO_ROOT']
for p in ('p1','p2'):
    f = glob.glob(R+'/gpurun_out/pmc_gemm/%s/**/*counter_collection.csv' % p, recursive=True)
    if not f: print(p, 'no csv'); continue
    rows = list(csv.DictReader(open(f[0])))
    per = collections.defaultdict(dict)
    for r in rows:
        if 'gemm_nt_kernel' in r['Kernel_Name']:
            d = per[r['Dispatch_Id']]; d[r['Counter_Name']] = d.get(r['Counter_Name'], 0) + float(r['Counter_Value']); d['dur'] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    last = list(per.values())[-1]
    print(p, {k: ('%.4g' % v) for k, v in last.items()})
P
