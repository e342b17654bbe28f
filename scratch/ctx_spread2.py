"""Which part of a context fixes the level of the evaluation time: its buffers or its streams?  One context; between measurements the
factor storage is re-allocated (a fit at another N in between) -- the streams stay.   python scratch/ctx_spread2.py [K]"""
import sys, time, gc, numpy as np
sys.path.insert(0, "/root/repo")
from gptools_amd import _lib
import bench
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
kernel, N, d, deriv = bench.WORKLOADS["c3"]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
Xs, ns, ys, errs = X[:3000], n[:3000], y[:3000], err[:3000]
ctx = _lib.Context(0); ctx.set_option("eager_alpha", 1)
def level():
    ctx.set_data(X, n)
    for _ in range(8): ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14); ctx.get_alpha(N)
    ts = []
    for r in range(6):
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14); ctx.get_alpha(N)
        ts.append((time.perf_counter() - t0) / 20 * 1e3)
    return np.median(ts), min(ts), max(ts)
for k in range(K):
    print("same streams, allocation %d: median %.3f ms  min %.3f  max %.3f" % ((k,) + level()), flush=True)
    if k % 2 == 1:
        print("   (again, nothing re-allocated: median %.3f)" % level()[0], flush=True)
    ctx.set_data(Xs, ns); ctx.fit(bench.KID[kernel], params, 0.0, ys, errs, 2.2e-14)      # another N: the factor storage is re-allocated
