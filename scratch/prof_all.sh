#!/bin/bash
# Runs on the GPU box: kernel trace + three PMC passes of the bench workload (C3, N=8192).
set -x
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${ROUND_TAG:-r02}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (counter collection runs one kernel at a time: the library switches its cross-kernel flag edges off when it sees
# ROCPROF_COUNTER_COLLECTION; every pass is under a timeout all the same)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-batched --no-predict --no-gp-api > $OUT/trace.log 2>&1
rm -f $OUT/pmc_fetch_gemm_shapes.txt; export GPT_GEMM_LOG=$OUT/pmc_fetch_gemm_shapes.txt
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-batched --no-predict --no-gp-api > $OUT/pmc_fetch.log 2>&1
unset GPT_GEMM_LOG
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-batched --no-predict --no-gp-api > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-batched --no-predict --no-gp-api > $OUT/pmc_sq.log 2>&1
ls -R $OUT | head -40
tail -2 $OUT/trace.log
