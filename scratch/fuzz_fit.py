"""Randomised parity sweep of gpt_fit against the CPU oracle: random N (also ragged, not multiples of 128), kernels, dimensions,
derivative patterns, hyperparameters; one context alive (flag edges in use where the schedule allows).  Test infrastructure
only (uses oracle/).   timeout 600 python scratch/fuzz_fit.py [seconds] [seed]"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
from oracle import oracle as O
O.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
KID = {"se": _lib.KERNEL_SE, "m52": _lib.KERNEL_M52, "rq": _lib.KERNEL_RQ, "matern": _lib.KERNEL_MATERN}
t_end = time.time() + budget
cases = worst = 0
worst_case = None
while time.time() < t_end:
    kern = ["se", "m52", "rq", "matern"][rs.randint(4)]
    d = int(rs.randint(1, 4))
    N = int(rs.choice([rs.randint(130, 700), rs.randint(700, 2600), rs.randint(2600, 5200)]))
    X = rs.rand(N, d)
    n = np.zeros((N, d), dtype=int)
    frac = 0.0 if kern == "matern" else rs.choice([0.0, 0.1, 0.3])      # (the reference's general Matern is not finite at tau = 0 with derivatives)
    idx = np.where(rs.rand(N) < frac)[0]
    for i in idx:
        n[i, rs.randint(d)] = 1
    s = X.sum(1)
    y = np.where(n.sum(1) > 0, np.cos(s), np.sin(s)) + 0.05 * rs.randn(N)
    err = 0.03 + 0.05 * rs.rand(N)
    ell = list(0.2 + 0.5 * rs.rand(d))
    if kern in ("se", "m52"): p = np.array([0.5 + rs.rand()] + ell)
    elif kern == "rq": p = np.array([0.5 + rs.rand(), 0.7 + 2 * rs.rand()] + ell)
    else: p = np.array([0.5 + rs.rand(), 0.6 + 2.3 * rs.rand()] + ell)
    try:
        ref = O.fit(kern, p, X, n, y, err, chol="scipy")
    except (np.linalg.LinAlgError, ValueError):
        continue
    ctx = _lib.Context(0)
    ctx.set_data(X, n)
    for opt in ({}, {"edge_flags": 0}):
        for k_, v_ in opt.items(): ctx.set_option(k_, v_)
        ll, ld = ctx.fit(KID[kern], p, 0.0, y, err, 1e2 * np.finfo(float).eps)
        e = max(abs(ll - ref["ll_data"]) / abs(ref["ll_data"]), abs(ld - ref["logdet_half"]) / max(1.0, abs(ref["logdet_half"])))
        if e > worst: worst, worst_case = e, (kern, N, d, frac, opt)
        assert e < 1e-7, (kern, N, d, frac, opt, ll, ref["ll_data"], ld, ref["logdet_half"])
    del ctx
    cases += 1
print("fuzz: %d cases in %.0f s, worst relative difference of ll / log-det %.2e at %r" % (cases, budget, worst, worst_case))
