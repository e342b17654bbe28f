#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one trailing-update launch (m=7168, k=384) for several tile orders
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_order
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for o in 8,8,0 8,8,1 32,8,1 16,16,1 64,8,1 16,12,1; do
  export GPT_TILE_ORDER=$o
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f_$o -o t -- python3 $R/scratch/gemm_one.py 7168 384 > $OUT/f.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w_$o -o t -- python3 $R/scratch/gemm_one.py 7168 384 > $OUT/w.log 2>&1
done
python3 - <<'P'
import csv, collections, glob, os
R=os.environ['GRAFT_REPO_ROOT']
for d in sorted(glob.glob(R+'/gpurun_out/pmc_order/*_*')):
    f = glob.glob(d+'/**/*counter_collection.csv', recursive=True)
    if not f: print(d, 'no csv'); continue
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        if 'gemm_nt_kernel' in r['Kernel_Name']: per[r['Dispatch_Id']] += float(r['Counter_Value'])
    v = list(per.values())[-1]
    print(os.path.basename(d), "counter %.5g KB -> %.1f MB (FETCH x2 on gfx950: %.1f MB)" % (v, v * 1024 / 1e6, 2 * v * 1024 / 1e6))
P
rm -rf $OUT/*/
