#!/bin/bash
# the modelled 8-rank runs of profiles/r03_sim_ranks_modelled.txt (one MI355X; ~6 GPU-minutes)
for lat in 5 20 50; do
  python scratch/sim_model.py c4 8 bcast bcast $lat 2,3,8,32 1 40 2>&1 | tail -2
  python scratch/sim_model.py c4 8 bcast scatter_gather $lat 2,3,8,32 1 40 2>&1 | tail -2
done
python scratch/sim_model.py c4 8 pipelined bcast 20 2,3,8,32 1 40 2>&1 | tail -2
python scratch/sim_model.py c4 8 pipelined scatter_gather 20 2,3,8,32 head 40 2>&1 | tail -2
SIM_BW=1e15 python scratch/sim_model.py c4 8 bcast bcast 0 2,3,8,32 1 40 2>&1 | tail -2
