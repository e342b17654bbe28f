"""K-build time inside a fit (lower triangle, fused diagonal): python scratch/kb_time.py [workload] [reps]"""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx = _lib.Context(0)
ctx.set_option("timing", 1)
ctx.set_option("lookahead", 0)       # K build alone on the chip (with look-ahead the panel stream starts under it)
ctx.set_data(X, n)
kb = []
for _ in range(reps):
    ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
    kb.append(ctx.last_timings()["kbuild"])
b = min(kb)
print("%s N=%d: K build (lower triangle) best %.4f ms -> %.2f TB/s written" % (wl, N, b, 0.5 * N * (N + 1) * 8 / b * 1e-9))
