"""A/B of run-time options inside ONE process / context (box-to-box variance on this pool is ~5 %: only compare inside
one gpurun call).  python scratch/sweep.py <workload> <reps> <rounds> name:opt=v,opt=v name2:opt=v ...
Round-robins the configurations; prints best and median GPU-timeline time of one LML evaluation per configuration."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
wl, reps, rounds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfgs = []
for a in sys.argv[4:]:
    name, _, opts = a.partition(":")
    cfgs.append((name, [(o.split("=")[0], int(o.split("=")[1])) for o in opts.split(",") if o]))
DEFAULTS = {"inner": 0, "inner_rows": 4608, "nb_outer": 0, "fuse_trsm": 8192, "gemm_pad": 1024, "helper_tf": 35, "ramp": 0,
            "early_rows": 0, "leaf256": 0, "defer_rows": 0, "late_rows": 0, "nb_early": 0, "nb_switch_rows": 4608, "purg_rows": 6144, "late_pad": 0, "late_pad_rows": 4608, "panel_prio": 2, "helper_min_n": 12288, "helper_tf": 45, "edge_flags": 1, "merge_urgent": 1, "merge_min_tiles": 512, "purg_rows_flags": 0, "tail_wait": 0, "tile": 0}
ctx = _lib.Context(0)
ctx.set_option("timing", 1)
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx.set_data(X, n)
times = {name: [] for name, _ in cfgs}
lls = {}
for rnd in range(rounds):
    for name, opts in cfgs:
        for k, v in DEFAULTS.items():
            try:
                ctx.set_option(k, v)
            except ValueError:
                pass
        for k, v in opts:
            ctx.set_option(k, v)
        for _ in range(reps):
            ll, ld = ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
            times[name].append(ctx.last_timings()["total"])
        lls[name] = (ll, ld)
ref = lls[cfgs[0][0]]
for name, _ in cfgs:
    t = np.array(times[name])
    print("%-28s best %.3f  median %.3f ms  (%.1f TF/s at best)  ll diff %.1e" % (
        name, t.min(), np.median(t), bench.flops_fit(N) / t.min() * 1e-9, abs(lls[name][0] - ref[0]) / abs(ref[0])))
