R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b
rm -rf $O; mkdir -p $O
cd $R
cp scratch/HEAD_for_collect.txt $O/HEAD.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/pytest.txt
JITTER_REPS=8 timeout 600 python tests/dist_jitter_worker.py 3 2>&1 | grep -v Gloo > $O/jitter3.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_c3_line.json 2> $O/bench_c3.err
ROUND_TAG=r04b/prof timeout 2400 bash scratch/prof_all.sh > $O/prof_all.log 2>&1
python scratch/pmc_summary.py $O/prof $O/rocprof_summary.txt $O/gemm_traffic.json 4 > $O/pmc_summary.log 2>&1
rm -rf $O/prof/*/t_*trace.csv $O/prof/*/t_counter_collection.csv
tail -4 $O/pytest.txt; cat $O/jitter3.txt; tail -22 $O/pmc_summary.log
