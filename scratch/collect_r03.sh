#!/bin/bash
# Runs on the GPU box: everything profiles/r03_* is made from (about 12 GPU-minutes).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03d
rm -rf $O; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest.txt
python bench.py --steps 20 --warmup 5 > $O/bench_c3_line.json 2> $O/bench_c3.err
python bench.py --steps 10 --warmup 3 --workload c2 --no-batched > $O/bench_c2_line.json 2> $O/bench_c2.err
python bench.py --steps 10 --warmup 3 --workload c5 --no-batched > $O/bench_c5_line.json 2> $O/bench_c5.err
python bench.py --steps 5 --warmup 2 --workload c4 --no-batched > $O/bench_c4_line.json 2> $O/bench_c4.err
python bench.py --gpus 1 --dist --workload c4 --steps 5 --warmup 2 > $O/bench_dist_c4_line.json 2> $O/bench_dist_c4.err
for N in 256 1024 2048; do python scratch/batch_grid_bench.py $N 64; done > $O/batch_grid.txt 2>&1
python scratch/small_n.py > $O/small_n.txt 2>&1
python scratch/instr_ab.py c3 20 > $O/instr_ab.txt 2>&1
python scratch/instr_ab.py c2 40 >> $O/instr_ab.txt 2>&1
python scratch/kb_alone.py 8192 > $O/kb_alone.txt 2>&1
python scratch/kb_alone.py 16384 >> $O/kb_alone.txt 2>&1
for v in NOCOMPUTE NOSTORE; do echo "== measurement build $v" >> $O/kb_alone.txt; python scratch/run_with_lib.py scratch/lib_kb_$v.so scratch/kb_alone.py 8192 >> $O/kb_alone.txt 2>&1; done
bash scratch/pmc_lds.sh head gptools_amd/libgpt_hip.so > $O/pmc_lds.txt 2>&1
python scratch/predict_cov_time.py c3 4096 > $O/predict.txt 2>&1
python scratch/predict_bench.py c3 64 256 1024 >> $O/predict.txt 2>&1
GPT_GRAD_TIMING=1 python scratch/grad_bench.py > $O/grad_bench.txt 2>&1
python scratch/c5_map_grad.py 16384 > $O/c5_map_gradient.txt 2>&1
python scratch/c5_map_grad.py 8192 >> $O/c5_map_gradient.txt 2>&1
ROUND_TAG=r03d/prof bash scratch/prof_all.sh > $O/prof_all.log 2>&1
python scratch/pmc_summary.py $O/prof $O/rocprof_summary.txt $O/gemm_traffic.json 3 > $O/pmc_summary.log 2>&1
bash scratch/trace_two.sh r03final c3 6 > $O/trace.log 2>&1
cp $R/gpurun_out/tl_r03final/timeline.txt $O/timeline_c3.txt
rm -rf $O/prof/*/t_*trace.csv $O/prof/*/t_counter_collection.csv
tail -3 $O/pytest.txt
