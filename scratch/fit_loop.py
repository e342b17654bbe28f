import sys, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = _lib.Context(0)
import os
idle = [_lib.Context(0) for _ in range(int(os.environ.get("IDLE_CTX", "0")))]
ctx.set_option("timing", 1)
for a in sys.argv[3:]:
    k, v = a.split("=")
    ctx.set_option(k, int(v))
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx.set_data(X, n)
best = 1e9
for _ in range(reps):
    ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
    best = min(best, ctx.last_timings()["total"])
print("%s N=%d: best %.3f ms -> %.1f TF/s" % (wl, N, best, bench.flops_fit(N) / best * 1e-9))
