mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_a_dist_processes.py -x -q 2>&1 | tail -12 > gpurun_out/t6.log
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "alpha or extents or distributed_plan or c4" 2>&1 | tail -4 >> gpurun_out/t6.log
timeout 200 python bench.py --steps 20 --warmup 3 --no-batched --no-predict > gpurun_out/b4.json 2> gpurun_out/b4.err
GPT_ALPHA_TWO_LAUNCH=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-batched --no-predict --no-cpu > gpurun_out/b4_two.json 2> gpurun_out/b4_two.err
timeout 500 python scratch/sim_model_grid.py c4 4 2 20 40 > gpurun_out/simgrid_4x2_20us.log 2>&1
timeout 500 python scratch/sim_model_grid.py c4 2 4 20 40 > gpurun_out/simgrid_2x4_20us.log 2>&1
cat gpurun_out/t6.log
head -3 gpurun_out/simgrid_4x2_20us.log; tail -n 2 gpurun_out/simgrid_4x2_20us.log gpurun_out/simgrid_2x4_20us.log
python -c "
import json
for f in ('b4','b4_two'):
    d=json.load(open('gpurun_out/%s.json'%f)); print(f, d['ms_per_step'], d['roofline']['frac'], d['with_alpha'])"
