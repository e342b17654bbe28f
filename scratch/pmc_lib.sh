#!/bin/bash
# FETCH_SIZE of one 7168-row trailing update for several library builds: pmc_lib.sh lib1.so lib2.so ...
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_lib
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for l in "$@"; do
  b=$(basename $l .so)
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f_$b -o t -- python3 $R/scratch/run_with_lib.py $R/$l $R/scratch/gemm_one.py 7168 384 > $OUT/f.log 2>&1
done
python3 - <<'P'
import csv, collections, glob, os
R=os.environ['GRAFT_REPO_ROOT']
for d in sorted(glob.glob(R+'/gpurun_out/pmc_lib/f_*')):
    f = glob.glob(d+'/**/*counter_collection.csv', recursive=True)
    if not f: print(d, 'no csv'); continue
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        if 'gemm_nt_kernel' in r['Kernel_Name']: per[r['Dispatch_Id']] += float(r['Counter_Value'])
    v = list(per.values())[-1]
    print(os.path.basename(d), "FETCH x2: %.1f MB" % (2 * v * 1024 / 1e6))
P
rm -rf $OUT/*/
