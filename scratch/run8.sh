mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_a_dist_processes.py -x -q -s -k "xcds" 2>&1 | tail -8 > gpurun_out/t7.log
timeout 500 python scratch/sim_model_grid.py c4 4 2 20 30 > gpurun_out/simgrid_4x2_20us.log 2>&1
timeout 500 python scratch/sim_model_grid.py c4 2 4 20 30 > gpurun_out/simgrid_2x4_20us.log 2>&1
SIM_BW=1e15 timeout 500 python scratch/sim_model_grid.py c4 2 4 0 30 > gpurun_out/simgrid_2x4_free.log 2>&1
cat gpurun_out/t7.log
for f in gpurun_out/simgrid_4x2_20us.log gpurun_out/simgrid_2x4_20us.log gpurun_out/simgrid_2x4_free.log; do head -4 $f | tail -3; tail -n 2 $f; done
