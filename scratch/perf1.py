import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib as L
ctx = L.Context(0)
ctx.set_option("timing", 1)
rs = np.random.RandomState(1234)
def flops(N): return N**3/3 + N**2/2 + N/6 + 2*N**2
for (kid, N, d, deriv) in ((0, 4096, 2, False), (1, 8192, 3, True), (0, 16384, 2, False)):
    X = rs.rand(N, d); n = np.zeros((N, d), int)
    if deriv:
        for i in range(3*N//4, N): n[i, i % d] = 1
    y = np.sin(X.sum(1)) + 0.05*rs.randn(N)
    p = np.concatenate(([1.0], 0.3*np.ones(d))); err = 0.05*np.ones(N)
    ctx.set_data(X, n)
    for nb in (256, 512):
        for la in (0, 1):
            for gr in (0, 1):
                ctx.set_option("nb_outer", nb); ctx.set_option("lookahead", la); ctx.set_option("graph", gr)
                try:
                    for it in range(3):
                        t0 = time.perf_counter()
                        ll, ld = ctx.fit(kid, p, 0.0, y, err, 2.2e-14)
                        t1 = time.perf_counter()
                    tm = ctx.last_timings()
                    print("N %5d nb %3d la %d graph %d: wall %.2f ms kbuild %.3f potrf %.3f tail %.3f total %.3f ms -> %.1f TF/s (potrf-only %.1f) ll=%.6f" % (
                        N, nb, la, gr, (t1-t0)*1e3, tm['kbuild'], tm['potrf'], tm['tail'], tm['total'], flops(N)/tm['total']*1e-9, (N**3/3)/tm['potrf']*1e-9, ll))
                except Exception as e:
                    print("N", N, nb, la, gr, "FAILED", repr(e))
