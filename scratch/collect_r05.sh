#!/bin/bash
# Runs on the GPU box: everything profiles/r05_* is made from, at ONE HEAD in ONE gpurun.  ~15 GPU-minutes.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
rm -rf $O; mkdir -p $O
cd $R
cp scratch/HEAD_for_collect.txt $O/HEAD.txt 2>/dev/null
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_c3_line.json 2> $O/bench_c3.err
timeout 200 python bench.py --steps 10 --warmup 3 --workload c2 --no-batched > $O/bench_c2_line.json 2> $O/bench_c2.err
timeout 300 python bench.py --steps 10 --warmup 3 --workload c5 --no-batched > $O/bench_c5_line.json 2> $O/bench_c5.err
timeout 400 python bench.py --steps 5 --warmup 2 --workload c4 --no-batched > $O/bench_c4_line.json 2> $O/bench_c4.err
timeout 600 python bench.py --gpus 1 --dist --workload c4 --steps 5 --warmup 2 > $O/bench_dist_c4_line.json 2> $O/bench_dist_c4.err
for N in 256 1024 2048; do timeout 200 python scratch/batch_grid_bench.py $N 64; done > $O/batch_grid.txt 2>&1
for N in 512 1024 2048; do timeout 300 python scratch/batch_terms_bench.py $N 64; done >> $O/batch_grid.txt 2>&1
timeout 200 python scratch/kb_alone.py 8192 > $O/kb_alone.txt 2>&1
timeout 200 python scratch/kb_alone.py 16384 >> $O/kb_alone.txt 2>&1
timeout 120 python scratch/chain_terms.py > $O/chain_terms.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_a_dist_processes.py -x -q -s -k "xcds" 2>&1 | tail -6 > $O/edge_stress.txt
timeout 200 python scratch/predict_bench.py c3 64 256 1024 > $O/predict.txt 2>&1
timeout 300 python scratch/c5_map_grad.py 16384 > $O/c5_map_gradient.txt 2>&1
ROUND_TAG=r05/prof timeout 2400 bash scratch/prof_all.sh > $O/prof_all.log 2>&1
python scratch/pmc_summary.py $O/prof $O/rocprof_summary.txt $O/gemm_traffic.json 5 > $O/pmc_summary.log 2>&1
rm -rf $O/prof/*/t_*trace.csv $O/prof/*/t_counter_collection.csv
if [ -x scratch/r05_upd_stamps ]; then (timeout 60 scratch/r05_upd_stamps 4096 128; timeout 60 scratch/r05_upd_stamps 4096 256) > $O/upd_stamps.txt 2>&1; fi
timeout 300 python scratch/r05_upd_ab.py time > $O/upd_ab.txt 2>&1
timeout 300 python scratch/r05_pair_ab.py 0 6144 3072 > $O/pair_ab.txt 2>&1
timeout 300 python scratch/grid_diag.py c4 2 4 0 > $O/grid_diag.txt 2>&1
timeout 300 python scratch/r05_alpha_ab.py c3 > $O/alpha_ab.txt 2>&1
timeout 300 python scratch/r05_alpha_ab.py c2 >> $O/alpha_ab.txt 2>&1
timeout 300 python scratch/r05_binv_ab.py c3 >> $O/alpha_ab.txt 2>&1
timeout 300 python scratch/r05_splitk_ab.py c3 16 64 128 256 > $O/splitk_ab.txt 2>&1
timeout 300 python scratch/r05_splitk_ab.py c2 16 64 >> $O/splitk_ab.txt 2>&1
timeout 300 python scratch/r05_first_predict.py c2 c3 > $O/first_predict.txt 2>&1
timeout 900 python scratch/r05_fuzz_alpha.py 300 11 > $O/fuzz_alpha.txt 2>&1
bash scratch/trace_any.sh scratch/alpha_one.py c3 > /dev/null 2>&1; awk '/logdet_dot/{f=1} f' gpurun_out/tla/timeline.txt > $O/timeline_alpha_c3.txt
bash scratch/trace_any.sh scratch/predict_one.py c3 64 1 > /dev/null 2>&1; cp gpurun_out/tla/timeline.txt $O/timeline_predict_m64.txt
bash scratch/trace_fit.sh c3 6 > /dev/null 2>&1; cp gpurun_out/tl/timeline.txt $O/timeline_c3.txt
bash scratch/trace_fit.sh c2 6 > /dev/null 2>&1; cp gpurun_out/tl/timeline.txt $O/timeline_c2.txt
tail -3 $O/pytest.txt
