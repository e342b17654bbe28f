#!/bin/bash
# Runs on the GPU box: everything profiles/r06_* is made from, at ONE HEAD in ONE gpurun.  ~12 GPU-minutes.
R=$GRAFT_REPO_ROOT
mkdir -p "$R/gpurun_out/r06"
O="$R/gpurun_out/r06"
cd $R
cp scratch/HEAD_for_collect.txt $O/HEAD.txt 2>/dev/null
# (1) trace + counter passes first: bench.py reads the two json files they produce (frac_trace, traffic)
ROUND_TAG=r06/prof ROUND_NO=6 timeout 2400 bash scratch/prof_r06.sh > $O/prof.log 2>&1
cp $O/prof/rocprof_summary.txt $O/bench_c3_N8192_rocprof_summary.txt
cp $O/prof/gemm_traffic.json $O/gemm_traffic.json
cp $O/prof/gemm_trace.json $O/gemm_trace.json
cp $O/prof/gemm_traffic.json profiles/r06_gemm_traffic.json
cp $O/prof/gemm_trace.json profiles/r06_gemm_trace.json
# (2) the bench lines
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c3_N8192_line.json 2> $O/bench_c3.err
timeout 300 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err
timeout 300 python bench.py --steps 10 --warmup 3 --workload c2 --no-batched > $O/bench_c2_line.json 2> $O/bench_c2.err
timeout 400 python bench.py --steps 10 --warmup 3 --workload c5 --no-batched > $O/bench_c5_line.json 2> $O/bench_c5.err
timeout 600 python bench.py --steps 5 --warmup 2 --workload c4 --no-batched > $O/bench_c4_line.json 2> $O/bench_c4.err
timeout 900 python bench.py --gpus 1 --dist --workload c4 --steps 5 --warmup 2 > $O/bench_dist_c4_line.json 2> $O/bench_dist_c4.err
# (3) tests
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu.txt
# (4) stall attribution of the dominant kernel
timeout 900 bash scratch/r06_gemm_stalls.sh > $O/gemm_stalls.log 2>&1
cp gpurun_out/r06_stalls/summary.txt $O/gemm_stalls_summary.txt
# (5) compiled schedules: host enqueue, world-1 engine against gpt_fit
timeout 600 python scratch/plan_host.py 32768 8 3 2>&1 | grep -v amdgpu.ids > $O/plan_host.txt
timeout 600 python scratch/plan_host.py 16384 8 3 2>&1 | grep -v amdgpu.ids >> $O/plan_host.txt
timeout 600 python scratch/plan_host_grid.py 32768 2 4 5 2>&1 | grep -v amdgpu.ids >> $O/plan_host.txt
timeout 600 python scratch/plan_host_grid.py 32768 4 2 2 2>&1 | grep -v amdgpu.ids | grep "rank\|matrix\|grid" >> $O/plan_host.txt
# (6) chain under contention
(timeout 600 python scratch/chain_contention.py 16384; timeout 300 python scratch/chain_contention.py 8192 alone default pad=24576 pad=65536;
 for r in 64 96; do GPT_RESERVE_CUS=$r timeout 300 python scratch/chain_contention.py 16384 default pad=24576; done) 2>&1 | grep -v amdgpu.ids > $O/chain_contention.txt
for pad in 0 24576; do echo "== GPT_DIST_MAIN_PAD=$pad (the engines' own trailing updates with the cap)"; GPT_DIST_MAIN_PAD=$pad timeout 600 python scratch/plan_host.py 32768 8 3 2>&1 | grep "world 1";
  GPT_DIST_MAIN_PAD=$pad timeout 900 python scratch/sim_model.py c4 8 bcast scatter_gather 20 2,3,8,32 1 30 2>&1 | tail -2; done >> $O/chain_contention.txt 2>&1
# (7) the rest of the per-round table
for N in 256 1024 2048; do timeout 200 python scratch/batch_grid_bench.py $N 64; done > $O/batch_grid.txt 2>&1
timeout 200 python scratch/kb_alone.py 8192 > $O/kbuilder.txt 2>&1
timeout 200 python scratch/predict_bench.py c3 64 256 1024 > $O/predict.txt 2>&1
timeout 300 python scratch/c5_map_grad.py 16384 > $O/c5_map_gradient.txt 2>&1
timeout 600 python scratch/fuzz_fit.py 240 > $O/fuzz.txt 2>&1
timeout 300 python scratch/repeat.py 100 > $O/repeat.txt 2>&1
timeout 400 python scratch/pad_ab.py c3 c2 2>&1 | grep -v amdgpu.ids > $O/pad_ab.txt
bash scratch/trace_fit.sh c3 6 eager_alpha=1 > /dev/null 2>&1; cp gpurun_out/tl/timeline.txt $O/timeline_c3_eager.txt
tail -3 $O/pytest_gpu.txt
