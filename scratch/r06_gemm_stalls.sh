#!/bin/bash
# Runs on the GPU box: stall attribution of the dominant kernel (gemm_nt_kernel<64,64>).
# Counter passes (8 SQ slots each) over (a) the trailing-update shape alone (scratch/gemm_one.py: m x m x 384 lower trapezoid on the
# CU-masked main stream, six launches) and (b) the bench workload itself (C3), where the kernel shares the chip with the panel stream.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_stalls
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_avail.txt 2>&1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES"
P3="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM"
P4="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_WAIT_INST_ANY SQ_IFETCH SQ_BUSY_CU_CYCLES"
P5="TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  for m in 7168 4096; do
    timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/one_m${m}_p$i -o t -- python3 $R/scratch/gemm_one.py $m 384 > $O/one_m${m}_p$i.log 2>&1
  done
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/bench_p$i -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-batched --no-predict --no-gp-api > $O/bench_p$i.log 2>&1
done
python3 $R/scratch/r06_gemm_stalls.py $O > $O/summary.txt 2>&1
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
for f in $(find $O -name "t_counter_collection.csv"); do python3 - "$f" <<'PY'
import csv, sys, collections, re
p = sys.argv[1]
acc = collections.defaultdict(float)
for r in csv.DictReader(open(p)):
    k = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
    if 'gemm_nt_kernel' in k and int(r['Grid_Size']) >= 54528:
        k += ' (>= 1 GFLOP launches)'
    acc[(k, r['Counter_Name'])] += float(r['Counter_Value'])
open(p.replace('t_counter_collection.csv', 'sums.txt'), 'w').write("\n".join("%s\t%s\t%.6g" % (k[0], k[1], v) for k, v in sorted(acc.items())) + "\n")
PY
rm -f $f; done
cat $O/summary.txt
