# random orders / kernels: alpha through the one-launch block inverses against rounds 2-4's launches, the residual K alpha - y, the
# eager path, and repeatability.  python scratch/r05_fuzz_alpha.py [cases] [seed]
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
import bench
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
ctx = _lib.Context(0)
worst = [0.0, 0.0, 0.0]
t0 = time.time()
for it in range(cases):
    N = int(rs.choice([rs.randint(1024, 2600), rs.randint(2600, 6000), rs.randint(6000, 9300)]))
    d = int(rs.randint(1, 4))
    kern = ["se", "m52"][rs.randint(2)]
    X = rs.rand(N, d)
    n = np.zeros((N, d), dtype=np.int32)
    if kern == "m52" and rs.rand() < 0.5:
        n[rs.rand(N) < 0.2, rs.randint(d)] = 1
    y = np.sin(3 * X.sum(1)) + 0.05 * rs.randn(N)
    err = np.full(N, 0.02 + 0.1 * rs.rand())
    p = np.concatenate(([0.5 + rs.rand()], 0.2 + 0.6 * rs.rand(d)))
    ctx.set_data(X, n)
    kid = bench.KID[kern]
    out = {}
    for name, opts in (("old", (1, 0)), ("new", (0, 0)), ("eager", (0, 1)), ("new2", (0, 0))):
        ctx.set_option("binv_launches", opts[0]); ctx.set_option("eager_alpha", opts[1])
        ctx.fit(kid, p, 0.0, y, err, 2.2e-14)
        out[name] = ctx.get_alpha(N)
    sc = np.abs(out["old"]).max()
    d1 = np.abs(out["new"] - out["old"]).max() / sc
    assert np.array_equal(out["new"], out["eager"]) and np.array_equal(out["new"], out["new2"]), (N, d, kern)
    # residual through the device's own K (compute_Kij) for the smaller cases
    r = 0.0
    if N <= 3000:
        K = ctx.kbuild(kid, p, X, n)
        K[np.diag_indices(N)] += err ** 2 + 2.2e-14
        r = np.abs(K.dot(out["new"]) - y).max() / max(1.0, np.abs(y).max())
    worst = [max(worst[0], d1), max(worst[1], r), max(worst[2], float(N))]
    if it % 10 == 9: print("%d cases, worst |new-old|/max|alpha| %.2e, worst residual %.2e (%.0f s)" % (it + 1, worst[0], worst[1], time.time() - t0), flush=True)
assert worst[0] < 1e-9 and worst[1] < 1e-6, worst
print("ok: %d cases, worst relative difference %.2e, worst residual %.2e" % (cases, worst[0], worst[1]))
