import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
ctx = _lib.Context(0)
ctx.set_option("timing", 1)
rs = np.random.RandomState(1234)
def flops(N): return N**3/3 + N**2/2 + N/6 + 2*N**2
for (kid, N, d, deriv) in ((1, 8192, 3, True), (0, 16384, 2, False)):
    X = rs.rand(N, d); n = np.zeros((N, d), int)
    if deriv:
        for i in range(3*N//4, N): n[i, i % d] = 1
    y = np.sin(X.sum(1)) + 0.05*rs.randn(N)
    p = np.concatenate(([1.0], 0.3*np.ones(d))); err = 0.05*np.ones(N)
    ctx.set_data(X, n)
    for pad in [int(v) for v in sys.argv[1:]]:
        ctx.set_option("gemm_pad", pad)
        best = 1e9
        for it in range(6):
            ll, ld = ctx.fit(kid, p, 0.0, y, err, 2.2e-14)
            best = min(best, ctx.last_timings()['total'])
        print("N %5d pad %6d: best total %.3f ms -> %.1f TF/s" % (N, pad, best, flops(N)/best*1e-9))
