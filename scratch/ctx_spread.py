"""Is the run-to-run level of the evaluation time fixed per PROCESS or per CONTEXT?  One process, the context (its streams, its
buffers) destroyed and re-created K times; eager C3 step timed each time.   python scratch/ctx_spread.py [K]"""
import sys, time, gc, numpy as np
sys.path.insert(0, "/root/repo")
from gptools_amd import _lib
import bench
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
kernel, N, d, deriv = bench.WORKLOADS["c3"]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
for k in range(K):
    ctx = _lib.Context(0); ctx.set_data(X, n); ctx.set_option("eager_alpha", 1)
    for _ in range(8): ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14); ctx.get_alpha(N)
    ts = []
    for r in range(6):
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14); ctx.get_alpha(N)
        ts.append((time.perf_counter() - t0) / 20 * 1e3)
    print("context %d: median %.3f ms  min %.3f  max %.3f" % (k, np.median(ts), min(ts), max(ts)), flush=True)
    ctx.close(); del ctx; gc.collect()
