"""eager vs lazy evaluation time, same process (C-ABI)"""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from gptools_amd import _lib
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx = _lib.Context(0); ctx.set_data(X, n)
def run(eager, reps=40):
    ctx.set_option("eager_alpha", eager)
    for _ in range(5): ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14); ctx.get_alpha(N)
    best = 1e9
    for r in range(6):
        t0 = time.perf_counter()
        for _ in range(reps):
            res = ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
            if eager: a = ctx.get_alpha(N)
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e3, res, (ctx.get_alpha(N) if True else None)
for rep in range(3):
    l, rl, al = run(0); e, re_, ae = run(1)
    print("%s lazy %.4f eager %.4f extra %.4f ms  ll equal %s  alpha max diff %.2e" % (wl, l, e, e - l, rl == re_, np.abs(al - ae).max()), flush=True)
