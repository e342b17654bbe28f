#!/bin/bash
# run-to-run spread of the bench line on ONE box: N processes back to back -> gpurun_out/spread.txt
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
: > gpurun_out/spread.txt
for i in $(seq 1 ${1:-8}); do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-batched --no-predict --no-gp-api 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['timed_region']
print('run $i: %.3f ms (%.2f %%) min %.3f max %.3f lazy %.3f extra %.3f frac_ev %.3f' % (d['ms_per_step'], d['pct_fp64_mfma_peak'], t['ms_per_step_min'], t['ms_per_step_max'], d['lazy_alpha']['ms_per_step'], d['lazy_alpha']['alpha_extra_ms'], d['roofline']['frac_events']))" | tee -a gpurun_out/spread.txt
done
