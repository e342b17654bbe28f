"""A/B of the in-launch leaf update (option fuse_upd, potf2_trsm_upd_kernel) inside one process:
bit-identity of ll / log-det / L against the separate-update schedule, then timings.
    python scratch/r05_upd_ab.py [check] [time]
GPT_EDGE_FLAGS=0 in the environment runs the same on event edges."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib

what = sys.argv[1:] or ["check", "time"]
ctx = _lib.Context(0)
ctx.set_option("timing", 1)


def flops(N):
    return N**3 / 3 + N**2 / 2 + N / 6


def data(N, d, deriv, seed=1234):
    rs = np.random.RandomState(seed)
    X = rs.rand(N, d)
    n = np.zeros((N, d), int)
    if deriv:
        for i in range(3 * N // 4, N):
            n[i, i % d] = 1
    y = np.sin(X.sum(1)) + 0.05 * rs.randn(N)
    p = np.concatenate(([1.0], 0.3 * np.ones(d)))
    return X, n, y, p, 0.05 * np.ones(N)


if "check" in what:
    bad = 0
    for (kid, N, d, deriv) in ((0, 300, 2, False), (0, 640, 2, False), (1, 1000, 3, True), (0, 1408, 2, False), (1, 2047, 3, True),
                               (0, 2048, 2, False), (0, 3000, 2, False), (0, 4096, 2, False), (1, 5000, 3, True), (1, 8192, 3, True),
                               (0, 12000, 2, False)):
        X, n, y, p, err = data(N, d, deriv)
        ctx.set_data(X, n)
        out = {}
        for v in (0, 1, 2):
            ctx.set_option("fuse_upd", 1 if v == 2 else 0)
            ctx.set_option("fuse_rows64", 0 if v == 0 else 4096)
            for rep in range(2):
                ll, ld = ctx.fit(kid, p, 0.0, y, err, 2.2e-14)
            L = ctx.get_L(N) if N <= 8192 else None
            out[v] = (ll, ld, L)
        same = all(out[0][0] == out[v][0] and out[0][1] == out[v][1] and (out[0][2] is None or np.array_equal(np.tril(out[0][2]), np.tril(out[v][2]))) for v in (1, 2))
        dl = 0.0 if out[0][2] is None else float(np.abs(np.tril(out[0][2]) - np.tril(out[1][2])).max())
        print("N %5d kid %d: ll %.15g / %.15g  logdet %.15g / %.15g  max|dL| %.3g  %s" % (
            N, kid, out[0][0], out[1][0], out[0][1], out[1][1], dl, "BIT-IDENTICAL" if same else "DIFFERENT"), flush=True)
        bad += 0 if same else 1
    # not positive definite: same info either way
    X, n, y, p, err = data(2048, 2, False)
    X[1500] = X[100]
    ctx.set_data(X, n)
    for v in (0, 1):
        ctx.set_option("fuse_upd", v)
        try:
            ctx.fit(0, p, 0.0, y, 0.0 * err, 0.0)
            print("fuse_upd %d: duplicate point with no noise factored?!" % v)
        except Exception as e:
            print("fuse_upd %d: %s" % (v, str(e)[:100]))
    print("check:", "OK" if bad == 0 else "%d MISMATCHES" % bad, flush=True)

CONFIGS = {0: dict(fuse_upd=0, fuse_rows64=0), 1: dict(fuse_upd=0, fuse_rows64=2048), 2: dict(fuse_upd=0, fuse_rows64=4096), 3: dict(fuse_upd=1, fuse_rows64=2048)}
if "time" in what:
    for (kid, N, d, deriv) in ((0, 1024, 2, False), (0, 2048, 2, False), (0, 4096, 2, False), (1, 8192, 3, True), (0, 16384, 2, False)):
        X, n, y, p, err = data(N, d, deriv)
        ctx.set_data(X, n)
        best = {v: 1e9 for v in CONFIGS}
        wall = {v: 1e9 for v in CONFIGS}
        for rnd in range(4):
            for v in CONFIGS:
                for k_, v_ in CONFIGS[v].items():
                    ctx.set_option(k_, v_)
                ctx.fit(kid, p, 0.0, y, err, 2.2e-14)
                t0 = time.perf_counter()
                reps = 10 if N <= 8192 else 4
                for it in range(reps):
                    ctx.fit(kid, p, 0.0, y, err, 2.2e-14)
                    best[v] = min(best[v], ctx.last_timings()['total'])
                wall[v] = min(wall[v], (time.perf_counter() - t0) / reps * 1e3)
        for v in CONFIGS:
            print("N %5d %s: best GPU %.3f ms  wall %.3f ms -> %.1f TF/s" % (N, CONFIGS[v], best[v], wall[v], flops(N) / wall[v] * 1e-9), flush=True)
