"""fit + get_alpha at the end of the process (for scratch/trace_any.sh): python scratch/alpha_one.py <workload>"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
wl = sys.argv[1]
ctx = _lib.Context(0)
for a in sys.argv[2:]:
    k, v = a.split('='); ctx.set_option(k, int(v))
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx.set_data(X, n)
for _ in range(3):
    ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
    t0 = time.perf_counter(); a = ctx.get_alpha(N); t = time.perf_counter() - t0
print("get_alpha: %.3f ms wall" % (t * 1e3))
