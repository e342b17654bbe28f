"""Cost of the bench's own instrumentation (timing events + per-launch GEMM events), same context, alternating:
python scratch/instr_ab.py <workload> <steps>"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
wl = sys.argv[1]; steps = int(sys.argv[2])
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx = _lib.Context(0)
ctx.set_data(X, n)
for _ in range(5): ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
modes = {"none": (0, 0), "timing": (1, 0), "gemm": (0, 1), "both": (1, 1)}
res = {k: [] for k in modes}
for rnd in range(5):
    for k, (tm, pg) in modes.items():
        ctx.set_option("timing", tm); ctx.set_option("profile_gemm", pg)
        ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
            if tm: ctx.last_timings()
        res[k].append((time.perf_counter() - t0) / steps * 1e3)
        if pg: ctx.gemm_profile_read()
for k in modes: print("%-8s min %.3f  median %.3f ms/step" % (k, min(res[k]), np.median(res[k])))
