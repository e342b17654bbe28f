"""Randomised sweep over what round 3 added, against the CPU oracle / the single-evaluation path (test infrastructure only, uses
oracle/): gpt_fit_batch bit-identity at random N and batch sizes, gpt_predict (mean / std / cov, every block-inverse route by the
choice of M) against the oracle's predict on the oracle's factor, the device-resident covariance + gpt_cov_sample against the host
factor of the returned covariance.   timeout 900 python scratch/fuzz_r03.py [seconds] [seed]"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
from oracle import oracle as O
import scipy.linalg
O.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
KID = {"se": _lib.KERNEL_SE, "m52": _lib.KERNEL_M52, "rq": _lib.KERNEL_RQ}
EPS = np.finfo(float).eps
t_end = time.time() + budget
cases = 0
worst = {"batch": 0, "mean": 0.0, "std": 0.0, "cov": 0.0, "sample": 0.0}
while time.time() < t_end:
    kern = ["se", "m52", "rq"][rs.randint(3)]
    d = int(rs.randint(1, 4))
    N = int(rs.choice([rs.randint(130, 700), rs.randint(700, 2600), rs.randint(2600, 4400)]))
    X = rs.rand(N, d)
    n = np.zeros((N, d), dtype=int)
    frac = rs.choice([0.0, 0.1, 0.3])
    for i in np.where(rs.rand(N) < frac)[0]:
        n[i, rs.randint(d)] = 1
    s = X.sum(1)
    y = np.where(n.sum(1) > 0, np.cos(s), np.sin(s)) + 0.05 * rs.randn(N)
    err = 0.03 + 0.05 * rs.rand(N)
    ell = list(0.2 + 0.5 * rs.rand(d))
    p = np.array([0.5 + rs.rand()] + ([0.7 + 2 * rs.rand()] if kern == "rq" else []) + ell)
    try:
        ref = O.fit(kern, p, X, n, y, err, chol="scipy")
    except (np.linalg.LinAlgError, ValueError):
        continue
    ctx = _lib.Context(0)
    ctx.set_data(X, n)
    # --- batch against single evaluations
    B = int(rs.randint(1, 7))
    P = p[None, :] * (1.0 + 0.05 * rs.rand(B, len(p)))
    P[0] = p
    nv = 0.01 * rs.rand(B); nv[0] = 0.0
    Y = np.tile(y, (B, 1)) + 0.01 * rs.randn(B, N); Y[0] = y
    llb, ldb, info = ctx.fit_batch(KID[kern], P, nv, Y, err, 1e2 * EPS)
    for b in range(B - 1, -1, -1):                       # (ends on element 0: its factor stays resident for predict)
        l1, d1 = ctx.fit(KID[kern], P[b], nv[b], Y[b], err, 1e2 * EPS)
        assert info[b] == 0 and (l1, d1) == (llb[b], ldb[b]), ("batch", kern, N, d, B, b)
    assert abs(llb[0] - ref["ll_data"]) <= 1e-7 * abs(ref["ll_data"]), ("ll", kern, N, d)
    # --- predict routes
    M = int(rs.choice([1, rs.randint(2, 129), rs.randint(129, 700)]))
    Xs = rs.rand(M, d)
    ns = np.zeros((M, d), dtype=int)
    if frac > 0 and M > 3:
        ns[::3, rs.randint(d)] = 1
    om, osd, ocov = O.predict(kern, p, X, n, ref["L"], ref["alpha"], Xs, ns, want_cov=True)
    gm, gs, gc = ctx.predict(Xs, ns, 2)
    gm1, gs1, _ = ctx.predict(Xs, ns, 1)
    scale = p[0] ** 2 * (1.0 + 10.0 * ns.sum(1).max())
    e_mean = np.abs(gm - om).max() / max(1.0, np.abs(om).max())
    e_cov = np.abs(np.array(gc) - ocov).max() / scale
    e_std = np.abs(gs1 ** 2 - np.diag(ocov)).max() / scale
    assert e_mean < 1e-7 and e_cov < 1e-7 and e_std < 1e-7, ("predict", kern, N, d, M, e_mean, e_cov, e_std)
    assert np.array_equal(gm, gm1)
    # --- device-resident covariance -> factor -> samples, against the host factor of the returned covariance
    if M >= 2:
        u = rs.randn(M, 3)
        jit = 1e3 * EPS * max(1.0, float(np.diag(ocov).max())) * 1e6      # (well conditioned: the comparison is of arithmetic, not of jitter)
        gm2, _, _ = ctx.predict(Xs, ns, 2, device_cov=True)
        try:
            smp = ctx.cov_sample(jit, u)
            Lh = scipy.linalg.cholesky(np.array(gc) + jit * np.eye(M), lower=True)
            e_s = np.abs(smp - Lh.dot(u)).max() / max(1.0, np.abs(Lh).max())
            worst["sample"] = max(worst["sample"], e_s)
            assert e_s < 1e-6, ("sample", kern, N, d, M, e_s)
        except np.linalg.LinAlgError:
            pass
    worst["mean"] = max(worst["mean"], e_mean); worst["cov"] = max(worst["cov"], e_cov); worst["std"] = max(worst["std"], e_std)
    ctx.close()
    cases += 1
print("fuzz_r03: %d cases in %.0f s; batch bit-identical throughout; worst scaled differences: %s" % (cases, budget, worst))
