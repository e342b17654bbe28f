import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
kernel, N, d, deriv = bench.WORKLOADS["c3"]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx = _lib.Context(0)
ctx.set_option("timing", 1)
ctx.set_data(X, n)
for _ in range(3): ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
for rnd in range(3):
    for prof in (0, 1):
        ctx.set_option("profile_gemm", prof)
        ctx.gemm_profile_read()
        w, g = [], []
        for _ in range(10):
            t0 = time.perf_counter()
            ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
            w.append((time.perf_counter() - t0) * 1e3)
            g.append(ctx.last_timings()["total"])
        print("profile_gemm=%d: wall median %.3f ms; GPU timeline median %.3f ms" % (prof, np.median(w), np.median(g)))
