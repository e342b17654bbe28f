cd $GRAFT_REPO_ROOT
O=gpurun_out/la; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $O/pytest_full.txt
(for wl in c2 c3 c5; do echo "== $wl"; timeout 600 python scratch/env_ab.py $wl 30 3 GPT_POTF2_LA 0 1; done) > $O/ab_lib.txt 2>&1
GPT_POTF2_LA=1 timeout 30 ./scratch/potf2_la_stamps > $O/la.txt 2>&1
GPT_POTF2_LA=1 timeout 30 ./scratch/potf2_la_events > $O/la_ev.txt 2>&1
GPT_POTF2_LA=0 timeout 30 ./scratch/potf2_la_events > $O/lock.txt 2>&1
timeout 30 ./scratch/dpp_rate > $O/dpp_rate.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err
cat $O/pytest_full.txt $O/ab_lib.txt; tail -c 600 $O/bench_c3.json | head -c 300; python -c "
import json; d=json.loads(open('$O/bench_c3.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('methodology'))"
