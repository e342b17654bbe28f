#!/bin/bash
# Runs on the GPU box: the trace / counter passes and the bench lines only (after a change of bench.py alone) -> gpurun_out/r06b
R=$GRAFT_REPO_ROOT
mkdir -p "$R/gpurun_out/r06b"
O="$R/gpurun_out/r06b"
cd $R
cp scratch/HEAD_for_collect.txt $O/HEAD.txt 2>/dev/null
ROUND_TAG=r06b/prof ROUND_NO=6 timeout 2400 bash scratch/prof_r06.sh > $O/prof.log 2>&1
cp $O/prof/rocprof_summary.txt $O/bench_c3_N8192_rocprof_summary.txt
cp $O/prof/gemm_traffic.json $O/gemm_traffic.json
cp $O/prof/gemm_trace.json $O/gemm_trace.json
cp $O/prof/gemm_traffic.json profiles/r06_gemm_traffic.json
cp $O/prof/gemm_trace.json profiles/r06_gemm_trace.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c3_N8192_line.json 2> $O/bench_c3.err
timeout 300 python bench.py --steps 10 --warmup 3 --workload c2 --no-batched > $O/bench_c2_line.json 2> $O/bench_c2.err
timeout 400 python bench.py --steps 10 --warmup 3 --workload c5 --no-batched > $O/bench_c5_line.json 2> $O/bench_c5.err
timeout 600 python bench.py --steps 5 --warmup 2 --workload c4 --no-batched > $O/bench_c4_line.json 2> $O/bench_c4.err
timeout 300 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err
python - <<PY
import json
for f in ("bench_c3_N8192_line.json", "bench_default_line.json", "bench_c2_line.json", "bench_c5_line.json", "bench_c4_line.json"):
    d = json.load(open("$O/" + f)); r = d.get("roofline") or {}
    print(f, "ms %.3f pct %.2f lazy %.3f (%.2f) frac_ev %.3f frac_tr %s" % (d["ms_per_step"], d["pct_fp64_mfma_peak"], d["lazy_alpha"]["ms_per_step"], d["lazy_alpha"]["pct_fp64_mfma_peak"], r.get("frac", 0), r.get("frac_trace")))
PY
