mkdir -p gpurun_out
timeout 300 python scratch/env_ab.py c3 40 3 GPT_GEMM_LOOP 0 4 > gpurun_out/loop_ab_c3.log 2>&1
timeout 300 python scratch/env_ab.py c5 10 2 GPT_GEMM_LOOP 0 4 > gpurun_out/loop_ab_c5.log 2>&1
timeout 600 python -m pytest tests/test_gpu_a_dist_processes.py -x -q -k "xcds" 2>&1 | tail -12 > gpurun_out/t5.log
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "alpha or extents or predict_result" 2>&1 | tail -4 >> gpurun_out/t5.log
GPT_GEMM_LOOP=0 timeout 200 python bench.py --steps 20 --warmup 3 > gpurun_out/b3.json 2> gpurun_out/b3.err
timeout 500 python scratch/sim_model_grid.py c4 4 2 20 40 > gpurun_out/simgrid_4x2_20us.log 2>&1
cat gpurun_out/loop_ab_c3.log gpurun_out/loop_ab_c5.log gpurun_out/t5.log
head -3 gpurun_out/simgrid_4x2_20us.log; tail -n 2 gpurun_out/simgrid_4x2_20us.log
python -c "import json; d=json.load(open('gpurun_out/b3.json')); print(d['ms_per_step'], d['roofline']['frac'], d['with_alpha'])"
