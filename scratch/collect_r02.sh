#!/bin/bash
# Runs on the GPU box: everything profiles/r02_* is made from.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c
rm -rf $O; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest.txt
python bench.py --steps 20 --warmup 5 > $O/bench_c3_line.json 2> $O/bench_c3.err
python bench.py --steps 10 --warmup 3 --workload c2 --no-batched > $O/bench_c2_line.json 2> $O/bench_c2.err
python bench.py --steps 10 --warmup 3 --workload c5 --no-batched > $O/bench_c5_line.json 2> $O/bench_c5.err
python bench.py --steps 5 --warmup 2 --workload c4 --no-batched > $O/bench_c4_line.json 2> $O/bench_c4.err
python bench.py --gpus 1 --dist --workload c4 --steps 5 --warmup 2 > $O/bench_dist_c4_line.json 2> $O/bench_dist_c4.err
# (the stamp programs compile the PRODUCT sources with -DGPT_*_STAMPS: rebuilt here so that they match the library)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -DGPT_PD_STAMPS -Iinclude -o scratch/potf2_stamps scratch/potf2_stamps.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -DGPT_GEMM_STAMPS -Iinclude -o scratch/gemm_stamps scratch/gemm_stamps.hip
./scratch/potf2_stamps > $O/potf2_stamps.txt 2>&1
./scratch/gemm_stamps 7168 384 32 > $O/gemm_stamps.txt 2>&1
ROUND_TAG=r02c/prof bash scratch/prof_all.sh > $O/prof_all.log 2>&1
python scratch/pmc_summary.py $O/prof $O/rocprof_summary.txt $O/gemm_traffic.json 2 > $O/pmc_summary.log 2>&1
bash scratch/trace_two.sh r02final c3 6 > $O/trace.log 2>&1
cp $R/gpurun_out/tl_r02final/timeline.txt $O/timeline_c3.txt
rm -rf $O/prof/*/t_*trace.csv $O/prof/*/t_counter_collection.csv
tail -3 $O/pytest.txt
