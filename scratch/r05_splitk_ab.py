# A/B of option `splitk` on predict at few points (same process, same fit): time and the difference of the results
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
Ms = [int(v) for v in sys.argv[2:]] or [16, 64, 128, 256]
ctx = _lib.Context(0)
kernel, N, d, deriv = bench.WORKLOADS[wl]
X, n, y, err, params = bench.synth(kernel, N, d, deriv)
ctx.set_data(X, n)
ctx.fit(bench.KID[kernel], params, 0.0, y, err, 2.2e-14)
rs = np.random.RandomState(3)
for M in Ms:
    Xs = rs.rand(M, d); ns = np.zeros((M, d), int)
    for want in (1, 2):
        res = {}
        for sk in (0, 512, 1024, 2048, 0, 1024):
            ctx.set_option('splitk', sk)
            ts = []
            for _ in range(6):
                t0 = time.perf_counter(); r = ctx.predict(Xs, ns, want); ts.append(time.perf_counter() - t0)
            res.setdefault(sk, []).append((min(ts[1:]), r))
        base = res[0][0][1]
        line = "%s N=%d M=%4d want=%d:" % (wl, N, M, want)
        for sk in (0, 512, 1024, 2048):
            t = min(v[0] for v in res[sk]); r = res[sk][0][1]
            dv = max(float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / (np.abs(np.asarray(b)) + 1e-300))) for a, b in zip(r[1:], base[1:]) if a is not None and b is not None) if sk else 0.0
            rep = all(np.array_equal(np.asarray(a), np.asarray(b)) for v in res[sk][1:] for a, b in zip(v[1], r) if a is not None)
            line += "  sk=%d %.3f ms (rel %.1e%s)" % (sk, t * 1e3, dv, "" if rep else " NOT-REPEATABLE")
        print(line, flush=True)
