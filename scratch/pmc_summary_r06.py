"""Summarise scratch/prof_r06.sh: kernel stats of the traced bench run, the dominant kernel's launches in that trace
(-> gemm_trace.json: what bench.py's roofline.frac_trace reads), HBM traffic of the replayed launch shapes (-> gemm_traffic.json)."""
import collections
import csv
import glob
import json
import os
import re
import sys

src, out, out_traffic, out_trace, rnd = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])


def short(n):
    return re.sub(r'\(.*', '', n).replace('void ', '')


def find(d, pat):
    f = glob.glob(os.path.join(d, '**', pat), recursive=True)
    return f[0] if f else None


lines = []
rows = list(csv.DictReader(open(find(src + '/trace', '*kernel_stats.csv'))))
lines.append("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-batched --no-predict "
             "--no-gp-api   (C3: Matern52, N=8192, d=3; every step computes and returns alpha)")
lines.append("%-44s %7s %14s %12s %10s %10s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "%"))
for r in rows:
    lines.append("%-44s %7s %14s %12.0f %10s %10s %7s" % (short(r['Name'])[:44], r['Calls'], r['TotalDurationNs'], float(r['AverageNs']),
                                                           r['MinNs'], r['MaxNs'], r['Percentage']))
bench = [l for l in open(src + '/trace.log') if l.startswith('{')]
bl = json.loads(bench[-1]) if bench else None
if bench:
    lines.append("\n# bench.py line of the traced run:\n" + bench[-1].strip())
K_OUTER = 384
MIN_GRID = -(-10**9 // (2 * K_OUTER * 64 * 64)) * 256
kt = list(csv.DictReader(open(find(src + '/trace', '*kernel_trace.csv'))))
mq = [r['Queue_Id'] for r in kt if 'kbuild_kernel' in r['Kernel_Name']][0]
bigd = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in kt
        if 'gemm_nt_kernel' in r['Kernel_Name'] and '64, 64' in r['Kernel_Name'] and r['Queue_Id'] == mq and int(r['Grid_Size_X']) >= MIN_GRID]
avg_us = sum(bigd) / len(bigd) * 1e-3
tr = {"round": rnd, "kernel": "gemm_nt_kernel<64,64>",
      "launch_filter": "main-stream queue, Grid_Size_X >= %d threads (the >= 1 GFLOP trailing updates of the flag schedule)" % MIN_GRID,
      "dispatches": len(bigd), "avg_launch_us": avg_us, "trace_file": "profiles/r%02d_bench_c3_N8192_rocprof_summary.txt" % rnd,
      "traced_command": "python3 bench.py --steps 5 --warmup 2 --no-cpu --no-batched --no-predict --no-gp-api --ctx-opt tail_wait=0 "
                        "(scratch/prof_r06.sh): the timed schedule with the main stream's wait for the next panel in a kernel of its own -- by default "
                        "the update's last workgroup waits (tail_wait), and a trace would count that wait, 20-50 us in the launches of the transition "
                        "to the chain-bound end, as the GEMM's duration; bench.py's own instrumented steps do the same (the library drops the tail "
                        "wait while profile_gemm times the launches)"}
if bl and bl.get("roofline"):
    tr["flops_per_launch"] = bl["roofline"]["flops_per_launch"]
    tr["events_avg_launch_us_same_run"] = bl["roofline"]["avg_launch_us"]
    tr["launches_per_step"] = bl["roofline"]["launches_per_step"]
    tr["frac_trace"] = tr["flops_per_launch"] / (avg_us * 1e-6) * 1e-12 / 78.6
    tr["frac_events_same_run"] = bl["roofline"]["frac_events"]
    lines.append("\n# dominant kernel, the launches bench.py's roofline times (gemm_nt_kernel<64,64> on the main queue, Grid_Size_X >= %d): %d "
                 "dispatches in the trace, average %.1f us -> frac_trace = %.3g GFLOP / %.1f us / 78.6 TFLOP/s = %.3f;  HIP events of the same run: "
                 "%.1f us -> frac_events %.3f" % (MIN_GRID, len(bigd), avg_us, tr["flops_per_launch"] * 1e-9, avg_us, tr["frac_trace"],
                                                  tr["events_avg_launch_us_same_run"], tr["frac_events_same_run"]))
json.dump(tr, open(out_trace, 'w'), indent=1)


def agg(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        k = short(r['Kernel_Name'])
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if (r['Dispatch_Id'], k) not in seen:
            seen.add((r['Dispatch_Id'], k))
            cnt[k] += 1
            acc[k]['_dur_ns'] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    return acc, cnt


lines.append("\n# SQ counter pass over the bench command (event schedule under counter collection): rocprofv3 --kernel-trace --pmc "
             "SQ_VALU_MFMA_BUSY_CYCLES ... -- python3 bench.py --steps 2 --warmup 1 ...")
f = find(src + '/pmc_sq', '*counter_collection.csv')
if f:
    acc, cnt = agg(f)
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES', 0)):
        s = "%-28s disp %4d" % (k[:28], cnt[k])
        if v.get('_dur_ns') and 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
            s += "  MFMA-busy %.1f%% (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x 2.4 GHz x kernel time))" % (
                100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * 2.4 * v['_dur_ns']))
        if v.get('SQ_LDS_IDX_ACTIVE'):
            s += "  LDS conflict cycles %.1f%%" % (100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / v['SQ_LDS_IDX_ACTIVE'])
        if 'SQ_INSTS_VALU_MFMA_MOPS_F64' in v:
            s += "  MFMA f64 flops %.4g" % (v['SQ_INSTS_VALU_MFMA_MOPS_F64'] * 512)
        lines.append(s)


# ---- traffic of the replayed launches (the timed schedule's shapes) ----
def replay_bytes(d, counter):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(find(d, '*counter_collection.csv'))):
        if r['Counter_Name'] == counter and 'gemm_nt_kernel' in r['Kernel_Name'] and '64, 64' in r['Kernel_Name']:
            per[r['Dispatch_Id']] += float(r['Counter_Value'])
    return len(per), sum(per.values()) * 1024.0


rep = [l for l in open(src + '/pmc_FETCH_SIZE.log') if l.startswith('REPLAY ')]
rj = json.loads(rep[-1][7:])
nf, fb = replay_bytes(src + '/pmc_FETCH_SIZE', 'FETCH_SIZE')
nw, wb = replay_bytes(src + '/pmc_WRITE_SIZE', 'WRITE_SIZE')
tj = {"round": rnd, "kernel": "gemm_nt_kernel<64,64>", "population": "replay of the flag schedule's launches",
      "note": "the >= 1 GFLOP main-stream launches of ONE flag-schedule evaluation (shapes logged by the library, GPT_GEMM_LOG, in an unprofiled run "
              "of bench.py), replayed one at a time by scratch/gemm_replay.py under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; "
              "FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md): the same launch shapes as roofline.flops_per_launch; a counter pass "
              "serialises kernels, so the bench command itself would run the event schedule there",
      "launches": nf, "launches_per_evaluation": rj["launches_per_evaluation"], "replay_repeats": rj["repeats"],
      "flops_per_launch": rj["flops_per_launch"], "algorithmic_bytes_per_launch": rj["algorithmic_bytes_per_launch"],
      "fetch_bytes_per_launch_x2_corrected": 2 * fb / nf, "write_bytes_per_launch": wb / nw,
      "hbm_bytes_per_launch": 2 * fb / nf + wb / nw, "ratio": (2 * fb / nf + wb / nw) / rj["algorithmic_bytes_per_launch"],
      "launch_count_matches_log": nf == rj["launches_per_evaluation"] * rj["repeats"], "shapes_m_n_k_tri": rj["shapes"]}
json.dump(tj, open(out_traffic, 'w'), indent=1)
lines.append("\n# HBM traffic of the timed schedule's launch shapes (replay, %d launches): %.1f MB per launch (fetch x2-corrected %.1f + write %.1f) "
             "against %.1f MB algorithmic: ratio %.2f" % (nf, tj["hbm_bytes_per_launch"] * 1e-6, 2 * fb / nf * 1e-6, wb / nw * 1e-6,
                                                          tj["algorithmic_bytes_per_launch"] * 1e-6, tj["ratio"]))
open(out, 'w').write("\n".join(lines) + "\n")
print("\n".join(lines[-12:]))
