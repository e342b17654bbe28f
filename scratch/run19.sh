cd $GRAFT_REPO_ROOT
O=gpurun_out/la; mkdir -p $O
(for wl in c2 c3 c5; do echo "== $wl"; timeout 600 python scratch/env_ab.py $wl 30 3 GPT_POTF2_LA 0 1; done) > $O/ab_lib.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > $O/pytest_full.txt
GPT_POTF2_LA=1 timeout 30 ./scratch/potf2_la_stamps > $O/la.txt 2>&1
GPT_POTF2_LA=1 timeout 30 ./scratch/potf2_la_events > $O/la_ev.txt 2>&1
GPT_POTF2_LA=0 timeout 30 ./scratch/potf2_la_events > $O/lock.txt 2>&1
timeout 30 ./scratch/chain_probe > $O/chain_probe.txt 2>&1
cat $O/ab_lib.txt $O/pytest_full.txt
