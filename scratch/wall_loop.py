"""Wall time per LML evaluation without per-phase timing: python scratch/wall_loop.py <workload> <reps>"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
wl = sys.argv[1]; reps = int(sys.argv[2])
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx = _lib.Context(0)
ctx.set_data(X, n)
for _ in range(5): ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
w = []
for _ in range(reps):
    t0 = time.perf_counter()
    ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
    w.append((time.perf_counter() - t0) * 1e3)
print("%s N=%d: best %.3f ms wall (median %.3f)" % (wl, N, min(w), np.median(w)))
