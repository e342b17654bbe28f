"""K-builder time of the rational-quadratic kernel against Matern-5/2 / SE on the C3 data set (derivative rows in the last quarter)."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
ctx = _lib.Context(0)
ctx.set_option("timing", 1)
kernel, N, d, deriv = bench.WORKLOADS["c3"]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx.set_data(X, n)
for name, kid, p in (("m52", 1, params), ("se", 0, params), ("rq", 4, np.concatenate(([1.0, 1.5], 0.3 * np.ones(d))))):
    ts, tt = [], []
    for _ in range(6):
        ll, ld = ctx.fit(kid, p, 0.0, y, err, 2.2e-14)
        tm = ctx.last_timings()
        ts.append(tm["kbuild"]); tt.append(tm["total"])
    t = min(ts[1:])
    print("%s N=%d d=%d: kbuild %.3f ms  %.2f TB/s written (lower)   total %.3f ms  ll %.10g" % (
        name, N, d, t, 8.0 * N * (N + 1) / 2 / t * 1e-9, min(tt[1:]), ll))
