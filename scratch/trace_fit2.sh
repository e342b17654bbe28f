#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tl2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPT_RESERVE_CUS=64 GPT_PANEL_CUS=64
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $R/scratch/fit_loop.py c3 6 > $OUT/trace.log 2>&1
python3 $R/scratch/timeline.py $OUT/trace 0 1e9 > $OUT/timeline.txt
tail -1 $OUT/timeline.txt
