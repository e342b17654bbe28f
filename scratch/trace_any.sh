#!/bin/bash
# usage: trace_any.sh <python script> [args]  -> gpurun_out/tla/{timeline.txt,stats.txt}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tla
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/"$@" > $OUT/trace.log 2>&1
python3 $R/scratch/timeline.py $OUT/trace 0 1e9 > $OUT/timeline.txt
tail -1 $OUT/timeline.txt
