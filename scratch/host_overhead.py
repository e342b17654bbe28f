"""Wall time per LML evaluation against the GPU timeline of the same evaluations, with the bench's instrumentation on/off."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
ctx = _lib.Context(0)
kernel, N, d, deriv = bench.WORKLOADS["c3"]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx.set_data(X, n)
for timing, prof in ((1, 1), (1, 0), (0, 0)):
    ctx.set_option("timing", timing); ctx.set_option("profile_gemm", prof)
    for _ in range(3): ctx.fit(1, params, 0.0, y, err, 2.2e-14)
    ws, gs = [], []
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(10): ctx.fit(1, params, 0.0, y, err, 2.2e-14)
        ws.append((time.perf_counter() - t0) / 10 * 1e3)
        if timing: gs.append(ctx.last_timings()["total"])
    ctx.gemm_profile_read()
    print("timing %d profile_gemm %d: wall %.3f ms per evaluation (best of 5 x 10)%s" % (
        timing, prof, min(ws), ("; GPU timeline of the last one %.3f ms" % min(gs)) if gs else ""))
