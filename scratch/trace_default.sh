#!/bin/bash
# kernel trace + stats of the bench command at the DEFAULT schedule (tail_wait on: the launches of the transition include their wait
# for the panel stream) -> gpurun_out/trace_default/summary.txt; the committed roofline trace is scratch/prof_r06.sh's (tail_wait=0)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace_default
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
GPT_BENCH_MIN_TIMED_S=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-batched --no-predict --no-gp-api > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
kt = list(csv.DictReader(open(glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0])))
mq = [r['Queue_Id'] for r in kt if 'kbuild_kernel' in r['Kernel_Name']][0]
big = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in kt if 'gemm_nt_kernel' in r['Kernel_Name'] and '64, 64' in r['Kernel_Name'] and r['Queue_Id'] == mq and int(r['Grid_Size_X']) >= 81408]
with open("$OUT/summary.txt", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-batched --no-predict --no-gp-api   (DEFAULT schedule: tail_wait on)\n")
    o.write("# >= 1 GFLOP main-stream launches of gemm_nt_kernel<64,64>: %d dispatches, average %.1f us (with tail_wait=0, the committed roofline trace: see r06_gemm_trace.json)\n" % (len(big), sum(big) / len(big) * 1e-3))
    o.write("# the wait kernels of the main stream are gone (wait_flag_kernel count below), their time is inside the launches of the transition\n")
    for r in rows[:14]:
        o.write("%-70s calls %6s  total %10.1f us  avg %8.1f us  %5s %%\n" % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs']) * 1e-3, float(r['AverageNs']) * 1e-3, r['Percentage']))
print(open("$OUT/summary.txt").read())
PY
find $OUT -name "*_kernel_trace.csv" -delete
