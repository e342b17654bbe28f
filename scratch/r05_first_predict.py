# the FIRST predict after a fit (alpha + every block inverse it needs are built inside it) and the second one
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
ctx = _lib.Context(0)
for wl in sys.argv[1:] or ["c2", "c3"]:
    kernel, N, d, deriv = bench.WORKLOADS[wl]
    X, n, y, err, params = bench.synth(kernel, N, d, deriv)
    ctx.set_data(X, n)
    rs = np.random.RandomState(3)
    for M in (64, 1024):
        Xs = rs.rand(M, d); ns = np.zeros((M, d), int)
        for mode in (1, 0):
            ctx.set_option('binv_launches', mode)
            t1, t2 = [], []
            for _ in range(4):
                ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
                t0 = time.perf_counter(); ctx.predict(Xs, ns, 1); t1.append(time.perf_counter() - t0)
                t0 = time.perf_counter(); ctx.predict(Xs, ns, 1); t2.append(time.perf_counter() - t0)
            print("%s N=%d M=%d std, binv_launches=%d: first predict after a fit %.3f ms, second %.3f ms" % (wl, N, M, mode, 1e3 * min(t1), 1e3 * min(t2)), flush=True)
