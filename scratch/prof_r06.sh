#!/bin/bash
# Runs on the GPU box: kernel trace of the bench workload (C3, N=8192) + PMC passes over a replay of the SAME launch shapes.
R=$GRAFT_REPO_ROOT
TAG=${ROUND_TAG:-r06/prof}
mkdir -p "$R/gpurun_out/$TAG"
OUT="$R/gpurun_out/$TAG"
cd /tmp && export TMPDIR=/tmp
# tail_wait=0: a kernel trace cannot tell a launch's wait from its work -- with the default (the update's last workgroup awaits the next
# panel's flag) the launches of the transition report 20-50 us of waiting as their own duration; the traced run keeps the wait in a kernel
# of its own, which is also what the bench's own instrumented steps do (the library drops the tail wait while profile_gemm times launches)
B="--steps 5 --warmup 2 --no-cpu --no-batched --no-predict --no-gp-api --ctx-opt tail_wait=0"
# (1) kernel trace + stats of the bench command itself: flag schedule, merged launches -- the timed population
GPT_BENCH_MIN_TIMED_S=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 $R/bench.py $B > "$OUT/trace.log" 2>&1
# (2) the shapes of that population (unprofiled evaluation, flag schedule)
: > "$OUT/gemm_shapes.txt"
GPT_GEMM_LOG="$OUT/gemm_shapes.txt" GPT_BENCH_MIN_TIMED_S=0 timeout 300 python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-batched --no-predict --no-gp-api > "$OUT/shapes_run.log" 2>&1
# (3) counter passes over the replay of those shapes (one kernel at a time is what a counter pass does to any launch)
for P in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/pmc_$P" -o t -- python3 $R/scratch/gemm_replay.py "$OUT/gemm_shapes.txt" 3 > "$OUT/pmc_$P.log" 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc_sq" -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-batched --no-predict --no-gp-api > "$OUT/pmc_sq.log" 2>&1
python3 $R/scratch/pmc_summary_r06.py "$OUT" "$OUT/rocprof_summary.txt" "$OUT/gemm_traffic.json" "$OUT/gemm_trace.json" ${ROUND_NO:-6} > "$OUT/pmc_summary.log" 2>&1
# the raw per-dispatch CSVs are tens of MB: only the summaries travel back
find "$R/gpurun_out/$TAG" -name "*_kernel_trace.csv" -delete
find "$R/gpurun_out/$TAG" -name "*counter_collection.csv" -delete
tail -30 "$OUT/pmc_summary.log"
