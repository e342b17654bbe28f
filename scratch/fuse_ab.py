"""A/B of the fuse_trsm option (potf2_trsm_kernel) inside one process: python scratch/fuse_ab.py v1 v2 ..."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
ctx = _lib.Context(0)
ctx.set_option("timing", 1)
rs = np.random.RandomState(1234)
def flops(N): return N**3/3 + N**2/2 + N/6 + 2*N**2
vals = [int(v) for v in sys.argv[1:]] or [0, 4096]
for (kid, N, d, deriv) in ((0, 4096, 2, False), (1, 8192, 3, True), (0, 16384, 2, False)):
    X = rs.rand(N, d); n = np.zeros((N, d), int)
    if deriv:
        for i in range(3*N//4, N): n[i, i % d] = 1
    y = np.sin(X.sum(1)) + 0.05*rs.randn(N)
    p = np.concatenate(([1.0], 0.3*np.ones(d))); err = 0.05*np.ones(N)
    ctx.set_data(X, n)
    best = {v: 1e9 for v in vals}; res = {}
    for rnd in range(4):
        for v in vals:
            ctx.set_option("fuse_trsm", v)
            for it in range(4):
                ll, ld = ctx.fit(kid, p, 0.0, y, err, 2.2e-14)
                best[v] = min(best[v], ctx.last_timings()['total'])
            res[v] = (ll, ld)
    for v in vals:
        print("N %5d fuse_trsm %5d: best %.3f ms -> %.1f TF/s  ll %.15g logdet/2 %.15g" % (N, v, best[v], flops(N)/best[v]*1e-9, res[v][0], res[v][1]), flush=True)
