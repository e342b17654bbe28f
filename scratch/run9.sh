mkdir -p gpurun_out
timeout 500 python scratch/sim_model_grid.py c4 2 4 20 30 > gpurun_out/simgrid_2x4_20us.log 2>&1
timeout 500 python scratch/sim_model_grid.py c4 4 2 20 30 > gpurun_out/simgrid_4x2_20us.log 2>&1
timeout 200 python scratch/grid_diag.py c4 2 4 0 > gpurun_out/grid_diag_2x4_r0.log 2>&1
timeout 200 python scratch/grid_diag.py c4 2 4 5 > gpurun_out/grid_diag_2x4_r5.log 2>&1
for f in gpurun_out/simgrid_2x4_20us.log gpurun_out/simgrid_4x2_20us.log; do head -4 $f | tail -3; tail -n 2 $f; done
grep "no communication" gpurun_out/grid_diag_2x4_r*.log
