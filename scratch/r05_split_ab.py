"""A/B of the split leaf updates (option split_rows) inside one process: bit-identity, then timings."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
ctx = _lib.Context(0)
ctx.set_option("timing", 1)
def flops(N): return N**3 / 3 + N**2 / 2 + N / 6
def data(N, d, deriv, seed=1234):
    rs = np.random.RandomState(seed)
    X = rs.rand(N, d); n = np.zeros((N, d), int)
    if deriv:
        for i in range(3 * N // 4, N): n[i, i % d] = 1
    y = np.sin(X.sum(1)) + 0.05 * rs.randn(N)
    return X, n, y, np.concatenate(([1.0], 0.3 * np.ones(d))), 0.05 * np.ones(N)
vals = [int(v) for v in sys.argv[1:]] or [0, 2048, 4096, 8192]
for (kid, N, d, deriv) in ((0, 2048, 2, False), (0, 4096, 2, False), (1, 8192, 3, True), (0, 12000, 2, False)):
    X, n, y, p, err = data(N, d, deriv)
    ctx.set_data(X, n)
    res = {}; best = {v: 1e9 for v in vals}; wall = {v: 1e9 for v in vals}
    for rnd in range(4):
        for v in vals:
            ctx.set_option("split_rows", v)
            ll, ld = ctx.fit(kid, p, 0.0, y, err, 2.2e-14)
            t0 = time.perf_counter()
            for it in range(8):
                ll, ld = ctx.fit(kid, p, 0.0, y, err, 2.2e-14)
                best[v] = min(best[v], ctx.last_timings()['total'])
            wall[v] = min(wall[v], (time.perf_counter() - t0) / 8 * 1e3)
            if rnd == 0:
                res[v] = (ll, ld, np.tril(ctx.get_L(N)) if N <= 8192 else None)
    for v in vals:
        same = res[v][0] == res[vals[0]][0] and res[v][1] == res[vals[0]][1] and (res[v][2] is None or np.array_equal(res[v][2], res[vals[0]][2]))
        print("N %5d split_rows %5d: best GPU %.3f ms  wall %.3f ms -> %.1f TF/s   ll %.15g  %s" % (
            N, v, best[v], wall[v], flops(N) / wall[v] * 1e-9, res[v][0], "bit-identical" if same else "DIFFERENT"), flush=True)
