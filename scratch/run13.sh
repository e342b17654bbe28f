cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/la
(for wl in c2 c3 c5; do echo "== $wl"; timeout 600 python scratch/env_ab.py $wl 30 3 GPT_POTF2_LA 0 1; done) > gpurun_out/la/ab_lib.txt 2>&1
GPT_POTF2_LA=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/la/pytest_la.txt
cat gpurun_out/la/ab_lib.txt gpurun_out/la/pytest_la.txt
