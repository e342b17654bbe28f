# old (15 launches) vs new (one kernel) 512-wide block inverses: alpha and a many-points predict through them, and the time of get_alpha
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
ctx = _lib.Context(0)
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx.set_data(X, n)
rs = np.random.RandomState(3)
M = 700
Xs = rs.rand(M, d); ns = np.zeros((M, d), int)
out = {}
for mode in (1, 0, 1, 0):
    ts = []
    for _ in range(4):
        ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
        ctx.set_option('binv_launches', mode)
        t0 = time.perf_counter(); a = ctx.get_alpha(N); ts.append(time.perf_counter() - t0)
    m, sd, _ = ctx.predict(Xs, ns, 1)
    out.setdefault(mode, []).append((a.copy(), m, sd))
    print("binv_launches=%d: get_alpha after a fit %.3f ms" % (mode, 1e3 * min(ts)), flush=True)
a1, m1, s1 = out[1][0]; a0, m0, s0 = out[0][0]
print("alpha: max |new - old| = %.3e (max |alpha| %.3e)" % (np.abs(a0 - a1).max(), np.abs(a1).max()))
print("std:   max |new - old| = %.3e" % np.abs(s0 - s1).max())
print("repeat new: %s" % all(np.array_equal(u, v) for u, v in zip(out[0][0], out[0][1])))
