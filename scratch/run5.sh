mkdir -p gpurun_out
timeout 60 python scratch/xcc_dbg.py > gpurun_out/xcc_dbg.log 2>&1
timeout 300 python scratch/env_ab.py c3 40 3 GPT_GEMM_LOOP 0 4 5 > gpurun_out/loop_ab_c3.log 2>&1
timeout 300 python scratch/env_ab.py c5 10 2 GPT_GEMM_LOOP 0 4 > gpurun_out/loop_ab_c5.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -6 > gpurun_out/t4.log
timeout 200 python bench.py --steps 20 --warmup 3 > gpurun_out/b2.json 2> gpurun_out/b2.err
timeout 500 python scratch/sim_model_grid.py c4 4 2 20 40 > gpurun_out/simgrid_4x2_20us.log 2>&1
SIM_BW=1e15 timeout 400 python scratch/sim_model_grid.py c4 4 2 0 40 > gpurun_out/simgrid_4x2_free.log 2>&1
cat gpurun_out/xcc_dbg.log gpurun_out/loop_ab_c3.log gpurun_out/loop_ab_c5.log gpurun_out/t4.log
tail -n 3 gpurun_out/simgrid_4x2_*.log
python -c "import json; d=json.load(open('gpurun_out/b2.json')); print(d['ms_per_step'], d['roofline']['frac'], d['kbuild_standalone'], d['kbuild_ms'])"
