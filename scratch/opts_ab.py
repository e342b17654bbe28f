"""Same-process A/B of option sets: python scratch/r05_opts_ab.py <workload> "k=v,k=v" "k=v" ...  (first set = baseline "")"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
wl = sys.argv[1]
sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[2:]]
keys = sorted(set(k for s in sets for k in s))
ctx = _lib.Context(0)
ctx.set_option("timing", 1)
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx.set_data(X, n)
DEF = {"nb_outer": 0, "nb_early": 0, "nb_switch_rows": 4608, "fuse_rows64": 2048, "fuse_rows32": 2048, "fuse_rows16": 0, "inner": 0, "inner_rows": 4608, "purg_rows_flags": 0,
       "merge_urgent": 1, "tail_wait": 0, "fuse_upd": 0, "pair_rows": 0, "ramp": 0, "fuse_trsm": 8192, "purg_rows": 6144, "late_pad": 0, "late_pad_rows": 4608, "gemm_pad": 1024, "head_wait_wgs": 33, "panel_prio": 2, "urgent_prio": 0, "urgent_split": 0, "merge_min_tiles": 512, "early_urgent": 0}
best = [1e9] * len(sets); wall = [1e9] * len(sets); res = [None] * len(sets)
for rnd in range(5):
    for i, s in enumerate(sets):
        for k in keys:
            ctx.set_option(k, s.get(k, DEF[k]))
        res[i] = ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
        t0 = time.perf_counter()
        reps = 8 if N <= 8192 else 3
        for it in range(reps):
            ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
            best[i] = min(best[i], ctx.last_timings()['total'])
        wall[i] = min(wall[i], (time.perf_counter() - t0) / reps * 1e3)
for i, s in enumerate(sets):
    print("%s N=%d %-60s best GPU %.3f ms  wall %.3f ms  ll %.12g" % (wl, N, s or "(default)", best[i], wall[i], res[i][0]), flush=True)
