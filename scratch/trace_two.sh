#!/bin/bash
# usage: trace_two.sh <tag> <workload> <reps> [opt=val ...]  -> gpurun_out/tl_<tag>/timeline.txt
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/tl_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $R/scratch/fit_loop.py "$@" > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log
python3 $R/scratch/timeline.py $OUT/trace 0 1e9 > $OUT/timeline.txt
tail -1 $OUT/timeline.txt
rm -rf $OUT/trace
