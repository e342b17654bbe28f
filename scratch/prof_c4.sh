#!/bin/bash
# kernel-trace statistics of the C4-sized evaluation on one GPU -> gpurun_out/r01c4/stats.txt
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r01c4
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py --workload c4 --steps 3 --warmup 1 --no-cpu --no-batched > $OUT/trace.log 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/trace/t_kernel_stats.csv")))
lines = ["# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --workload c4 --steps 3 --warmup 1 --no-cpu --no-batched   (C4: SE, N=32768, d=4, one GPU)",
         "%-44s %7s %14s %12s %10s %10s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "%")]
import re
for r in rows:
    nm = re.sub(r'\(.*', '', r['Name']).replace('void ', '')[:44]
    lines.append("%-44s %7s %14s %12.0f %10s %10s %7s" % (nm, r['Calls'], r['TotalDurationNs'], float(r['AverageNs']), r['MinNs'], r['MaxNs'], r['Percentage']))
b = [l for l in open("$OUT/trace.log") if l.startswith('{')]
if b: lines.append("\n# bench.py line of the traced run:\n" + b[-1].strip())
open("$OUT/stats.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:8]))
PY
rm -rf $OUT/trace
