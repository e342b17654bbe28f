import sys, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
ctx = _lib.Context(0)
ctx.set_option("timing", 1)
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx.set_data(X, n)
for _ in range(8):
    ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
    print({k: round(v * 1e3, 1) for k, v in ctx.last_timings().items()})
