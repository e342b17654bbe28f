#!/bin/bash
# kernel timeline of ONE rank's schedule of a W-rank job (scratch/sim_ranks.py): trace_sim.sh <schedule> <rank> [tag]
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/sim_${1}_${3:-x}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $R/scratch/sim_ranks.py c4 8 $1 $2 > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log
python3 $R/scratch/timeline.py $OUT/trace 0 1e9 > $OUT/timeline.txt
tail -1 $OUT/timeline.txt
rm -rf $OUT/trace
