#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tld
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py --dist --workload ${1:-c5} --steps 2 --warmup 1 --no-cpu --no-ref > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log | cut -c1-200
python3 $R/scratch/timeline.py $OUT/trace 0 1e9 > $OUT/timeline.txt
tail -1 $OUT/timeline.txt
