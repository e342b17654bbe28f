"""K-builder time / write bandwidth at the BASELINE shapes (lower triangle + diagonal epilogue, as gpt_fit runs it)."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
ctx = _lib.Context(0)
ctx.set_option("timing", 1)
for wl in (sys.argv[1:] or ["c2", "c3", "c5", "c4"]):
    kernel, N, d, deriv = bench.WORKLOADS[wl]
    X, n, y, err, params = bench.synth(kernel, N, d, deriv)
    ctx.set_data(X, n)
    ts = []
    for _ in range(6):
        ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
        ts.append(ctx.last_timings()["kbuild"])
    t = min(ts[1:])
    byts = 8.0 * N * (N + 1) / 2
    print("%s %s N=%d d=%d: kbuild %.3f ms  %.2f TB/s written (lower)  %.1f Gpairs/s" % (wl, kernel, N, d, t, byts / t * 1e-9, N * (N + 1) / 2 / t * 1e-6))
