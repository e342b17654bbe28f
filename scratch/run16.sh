R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04c
rm -rf $O; mkdir -p $O
cd $R
cp scratch/HEAD_for_collect.txt $O/HEAD.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_c3_line.json 2> $O/bench_c3.err
timeout 300 python bench.py --workload c2 --steps 50 --warmup 10 --no-cpu > $O/bench_c2_line.json 2> $O/bench_c2.err
ROUND_TAG=r04c/prof timeout 2400 bash scratch/prof_all.sh > $O/prof_all.log 2>&1
python scratch/pmc_summary.py $O/prof $O/rocprof_summary.txt $O/gemm_traffic.json 4 > $O/pmc_summary.log 2>&1
rm -rf $O/prof/*/t_*trace.csv $O/prof/*/t_counter_collection.csv
python - <<PY
import json
for f in ("bench_c3_line.json","bench_c2_line.json"):
    d=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"))
PY
head -16 $O/rocprof_summary.txt; tail -12 $O/pmc_summary.log
