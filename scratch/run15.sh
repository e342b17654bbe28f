cd $GRAFT_REPO_ROOT
O=gpurun_out/multi; mkdir -p $O
export GPT_BENCH_BACKEND=gloo GPT_BENCH_ONE_GPU=1 GPT_BENCH_WATCHDOG_S=400
for n in 2 4; do
  timeout 900 python bench.py --gpus $n --workload c2 --steps 3 --warmup 1 --no-probe --no-ref > $O/b$n.json 2> $O/b$n.err; echo "rc $?" >> $O/b$n.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/b$n.json") if l.startswith("{")][-1])
    print("gpus", d["n_gpus"], "ms", d["ms_per_step"], "parity", d["parity"]["ok"], "plan", d["config"].get("parallelism"))
    print(json.dumps(d.get("schedules_ms"), indent=0)[:1500])
except Exception as e:
    print("FAILED", e); print(open("$O/b$n.err").read()[-1500:])
PY
done
