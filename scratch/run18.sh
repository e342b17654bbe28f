cd $GRAFT_REPO_ROOT
O=gpurun_out/la; mkdir -p $O
(for wl in c2 c3; do echo "== $wl"; timeout 600 python scratch/env_ab.py $wl 30 3 GPT_POTF2_LA 0 1; done) > $O/ab_lib2.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2 > $O/pytest_la2.txt
GPT_POTF2_LA=1 timeout 30 ./scratch/potf2_la_events | grep -v "^  " > $O/la_ev2.txt 2>&1
cat $O/ab_lib2.txt $O/pytest_la2.txt $O/la_ev2.txt
