"""Wall time per LML evaluation at small N (the sizes most gptools users have), GPU path against scipy on the host."""
import sys, time, numpy as np, scipy.linalg
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
for N in (256, 512, 1024, 2048, 4096):
    X, n, y, err, params = bench.synth("se", N, 2, False)
    ctx = _lib.Context(0)
    ctx.set_data(X, n)
    for _ in range(5): ctx.fit(bench.KID["se"], params, 0.0, y, err, 2.2e-14)
    w = []
    for _ in range(50):
        t0 = time.perf_counter(); ctx.fit(bench.KID["se"], params, 0.0, y, err, 2.2e-14); w.append(time.perf_counter() - t0)
    K = np.exp(-0.5 * ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1) / 0.09) + np.diag(err ** 2)
    t0 = time.perf_counter()
    for _ in range(5): scipy.linalg.cho_factor(K, lower=True)
    tc = (time.perf_counter() - t0) / 5
    print("N=%5d: GPU fit %.3f ms wall (median %.3f), edges %d | host scipy cho_factor alone %.2f ms" % (N, min(w) * 1e3, np.median(w) * 1e3, ctx.edge_count, tc * 1e3))
    del ctx
