#!/usr/bin/env python
"""Generate the golden input/output fixtures (``tests/golden/*.npz``).

Container-only: imports the *reference* gptools from /root/reference through
``ref_harness`` and records its outputs on seeded inputs.  The fixtures are data
(inputs + expected outputs); neither this script nor the harness copies
reference source.  Re-run with ``python tests/golden/gen_golden.py``.

Fixture groups follow SURVEY.md section 8(c):
  G1 pair-level ``Kernel.__call__``            (kernel/squared_exponential.py:82-174, kernel/matern.py:512-555)
  G2 Gram matrices ``compute_Kij``             (gaussian_process.py:1535-1605)
  G3 fit ``compute_K_L_alpha_ll``              (gaussian_process.py:1418-1522)
  G4 ``predict``                               (gaussian_process.py:913-1034)
  G5 ``update_hyperparameters`` sweep          (gaussian_process.py:1332-1416)
  G6 demo known-answer (config 1)              (demo/demo.py:133-253)
  G7 tests/test_matern.py scenario, seeded     (tests/test_matern.py:4-31)
  G8 RationalQuadraticKernel: pairs, Gram, fit, predict (kernel/rational_quadratic.py:30-164, kernel/core.py:691-816)
  G10 MaternKernel (general nu): pairs, Gram, fit, predict (kernel/matern.py:251-465, utils.py:1369-1527)
  G9 ProductKernel (k1 * k2) with derivative orders: pairs, fit, predict (kernel/core.py:587-671)
  G11 RQ / general Matern / product pairs at combined derivative orders 9 .. 12 (not in the default list: ~15 minutes)
  G12 independent evaluations and samples: ``compute_ll_matrix`` grids (gaussian_process.py:1607-1692), random starts
      of ``optimize_hyperparameters`` (:689-735, :2443-2486), ``draw_sample`` / ``predict(return_samples=True)`` (:1155-1330, :990-1005)
"""
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

gptools = ref_harness.import_reference()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-28s %8.1f KiB" % (name + ".npz", os.path.getsize(path) / 1024.0))


def se_kernel(d, params):
    return gptools.SquaredExponentialKernel(
        num_dim=d, initial_params=list(params), param_bounds=[(0.0, 1e3)] * (d + 1))


def m52_kernel(d, params):
    return gptools.Matern52Kernel(
        num_dim=d, initial_params=list(params), param_bounds=[(0.0, 1e3)] * (d + 1))


# ----------------------------------------------------------------------------
# G1: pair-level
# ----------------------------------------------------------------------------
def gen_g1():
    rs = np.random.RandomState(101)
    out = {}
    for d in (1, 2, 3, 4):
        M = 240
        Xi = rs.rand(M, d)
        Xj = rs.rand(M, d)
        # tau == 0 rows and duplicate points
        Xj[:20] = Xi[:20]
        Xj[20:30, 0] = Xi[20:30, 0]
        ni = rs.randint(0, 3, size=(M, d))
        nj = rs.randint(0, 3, size=(M, d))
        ni[40:60] = 0                       # one-sided
        nj[60:80] = 0
        ni[80:100] = 0
        nj[80:100] = 0                      # plain rows inside a derivative call
        params = np.concatenate(([1.7], 0.25 + 0.5 * rs.rand(d)))
        k = se_kernel(d, params)
        out["se_d%d_Xi" % d] = Xi
        out["se_d%d_Xj" % d] = Xj
        out["se_d%d_ni" % d] = ni.astype(np.int32)
        out["se_d%d_nj" % d] = nj.astype(np.int32)
        out["se_d%d_params" % d] = params
        out["se_d%d_k" % d] = k(Xi, Xj, ni, nj)
        for hd in range(0, d + 1):
            with np.errstate(all="ignore"):
                out["se_d%d_hd%d" % (d, hd)] = k(Xi, Xj, ni, nj, hyper_deriv=hd)
        # the "no derivatives anywhere" fast path (squared_exponential.py:110-113)
        z = np.zeros_like(ni)
        out["se_d%d_k0" % d] = k(Xi, Xj, z, z)
        for hd in range(0, d + 1):
            out["se_d%d_k0_hd%d" % (d, hd)] = k(Xi, Xj, z, z, hyper_deriv=hd)
        # high orders (up to 4 per side => combined order 8)
        nih = rs.randint(0, 5, size=(M, d))
        njh = rs.randint(0, 5, size=(M, d))
        out["se_d%d_nih" % d] = nih.astype(np.int32)
        out["se_d%d_njh" % d] = njh.astype(np.int32)
        out["se_d%d_kh" % d] = k(Xi, Xj, nih, njh)

        # Matern 5/2: all four branch classes x (r == 0 | r > 0) x (a == b | a != b)
        mi = np.zeros((M, d), dtype=np.int32)
        mj = np.zeros((M, d), dtype=np.int32)
        cls = rs.randint(0, 4, size=M)
        cls[:20] = np.arange(20) % 4        # r == 0 rows cover every class
        for m in range(M):
            if cls[m] in (1, 3):
                mi[m, rs.randint(0, d)] = 1
            if cls[m] in (2, 3):
                mj[m, rs.randint(0, d)] = 1
        if d > 1:
            mi[100:110] = 0
            mj[100:110] = 0
            mi[100:110, 0] = 1
            mj[100:110, 0] = 1             # a == b
            mi[110:120] = 0
            mj[110:120] = 0
            mi[110:120, 0] = 1
            mj[110:120, 1] = 1             # a != b
            mi[4:8] = 0
            mj[4:8] = 0
            mi[4:6, 0] = 1
            mj[4:6, 0] = 1                 # r == 0, a == b
            mi[6:8, 0] = 1
            mj[6:8, 1] = 1                 # r == 0, a != b
        km = m52_kernel(d, params)
        out["m52_d%d_ni" % d] = mi
        out["m52_d%d_nj" % d] = mj
        out["m52_d%d_k" % d] = km(Xi, Xj, mi, mj)
    save("g1_pairs", **out)


# ----------------------------------------------------------------------------
# G2: Gram matrices
# ----------------------------------------------------------------------------
def deriv_pattern(rs, N, d, frac=0.25, max_order=1):
    """Rows [0, N-frac*N) value observations; the rest derivative observations."""
    n = np.zeros((N, d), dtype=int)
    n0 = int(N * (1.0 - frac))
    for i in range(n0, N):
        n[i, i % d] = 1 if max_order == 1 else rs.randint(1, max_order + 1)
    return n


def gen_g2():
    rs = np.random.RandomState(202)
    out = {}
    for kname, mk, maxo in (("se", se_kernel, 2), ("m52", m52_kernel, 1)):
        for d in (1, 2, 3):
            N, P = 64, 40
            X = rs.rand(N, d)
            Xs = rs.rand(P, d)
            X[5] = X[4]                      # duplicate training point
            Xs[3] = X[7]                     # test point on a training point
            n = deriv_pattern(rs, N, d, 0.25, maxo)
            ns = deriv_pattern(rs, P, d, 0.5, maxo)
            params = np.concatenate(([1.3], 0.2 + 0.4 * rs.rand(d)))
            gp = gptools.GaussianProcess(mk(d, params))
            key = "%s_d%d_" % (kname, d)
            out[key + "X"] = X
            out[key + "Xs"] = Xs
            out[key + "n"] = n.astype(np.int32)
            out[key + "ns"] = ns.astype(np.int32)
            out[key + "params"] = params
            out[key + "K"] = gp.compute_Kij(X, None, n, None)
            out[key + "Ks"] = gp.compute_Kij(X, Xs, n, ns)
            out[key + "Kss"] = gp.compute_Kij(Xs, None, ns, None)
            z = np.zeros_like(n)
            out[key + "K0"] = gp.compute_Kij(X, None, z, None)
            if kname == "se":
                for hd in range(0, d + 1):
                    out[key + "K0_hd%d" % hd] = gp.compute_Kij(X, None, z, None, hyper_deriv=hd)
                    with np.errstate(all="ignore"):
                        out[key + "K_hd%d" % hd] = gp.compute_Kij(X, None, n, None, hyper_deriv=hd)
    save("g2_gram", **out)


# ----------------------------------------------------------------------------
# G3 / G4: fit and predict
# ----------------------------------------------------------------------------
def synth(rs, N, d, n):
    X = rs.rand(N, d)
    s = X.sum(axis=1)
    y = np.sin(s)
    der = n.sum(axis=1) > 0
    y[der] = np.cos(s[der])
    y = y + 0.05 * rs.randn(N)
    return X, y


def record_fit(out, key, gp, with_L=True):
    gp.compute_K_L_alpha_ll()
    out[key + "ll"] = np.float64(gp.ll)
    out[key + "prior"] = np.float64(gp.hyperprior(gp.params))
    out[key + "logdet_half"] = np.float64(np.log(np.diag(gp.L)).sum())
    out[key + "alpha"] = np.asarray(gp.alpha).ravel()
    if with_L:
        out[key + "L"] = np.asarray(gp.L)


def gen_g3_g4():
    rs = np.random.RandomState(303)
    out3, out4 = {}, {}
    cases = []
    for kname, mk in (("se", se_kernel), ("m52", m52_kernel)):
        for N, d in ((16, 1), (64, 2), (256, 3), (512, 2)):
            cases.append((kname, mk, N, d))
    for kname, mk, N, d in cases:
        key = "%s_N%d_d%d_" % (kname, N, d)
        n = deriv_pattern(rs, N, d, 0.25, 1)
        X, y = synth(rs, N, d, n)
        params = np.concatenate(([1.0], 0.3 * np.ones(d)))
        err_y = 0.05 if N != 64 else 0.02 + 0.06 * rs.rand(N)   # scalar / heteroscedastic
        gp = gptools.GaussianProcess(mk(d, params), X=X, y=y, err_y=err_y, n=n)
        out3[key + "X"] = X
        out3[key + "y"] = y
        out3[key + "n"] = n.astype(np.int32)
        out3[key + "err_y"] = np.asarray(gp.err_y)
        out3[key + "params"] = params
        record_fit(out3, key, gp, with_L=(N <= 256))

        # predict at M = 32 incl. derivative rows
        M = 32
        Xs = rs.rand(M, d)
        ns = deriv_pattern(rs, M, d, 0.5, 1)
        mean, cov = gp.predict(Xs, n=ns, return_cov=True)
        mean2, std = gp.predict(Xs, n=ns)
        out4[key + "Xs"] = Xs
        out4[key + "ns"] = ns.astype(np.int32)
        out4[key + "mean"] = mean
        out4[key + "cov"] = cov
        out4[key + "std"] = std
        # scalar n
        m0, s0 = gp.predict(Xs, n=0)
        out4[key + "mean_n0"] = m0
        out4[key + "std_n0"] = s0
        # output_transform
        OT = rs.rand(5, M)
        mo, co = gp.predict(Xs, n=ns, return_cov=True, output_transform=OT)
        out4[key + "OT"] = OT
        out4[key + "mean_ot"] = mo
        out4[key + "cov_ot"] = co

    # DiagonalNoiseKernel (+ predict noise=True)
    for kname, mk in (("se", se_kernel), ("m52", m52_kernel)):
        N, d = 64, 2
        key = "%s_noise_" % kname
        n = deriv_pattern(rs, N, d, 0.25, 1)
        X, y = synth(rs, N, d, n)
        params = np.array([1.1, 0.35, 0.25])
        nk = gptools.DiagonalNoiseKernel(num_dim=d, initial_noise=0.07, noise_bound=(0.0, 5.0))
        gp = gptools.GaussianProcess(mk(d, params), noise_k=nk, X=X, y=y, err_y=0.01, n=n)
        out3[key + "X"] = X
        out3[key + "y"] = y
        out3[key + "n"] = n.astype(np.int32)
        out3[key + "err_y"] = np.asarray(gp.err_y)
        out3[key + "params"] = params
        out3[key + "noise"] = np.float64(0.07)
        record_fit(out3, key, gp)
        M = 24
        Xs = rs.rand(M, d)
        Xs[2] = Xs[1]                        # duplicate test points (noise fires off-diagonal)
        ns = np.zeros((M, d), dtype=int)
        ns[M // 2:, 0] = 1
        mean, cov = gp.predict(Xs, n=ns, noise=True, return_cov=True)
        out4[key + "Xs"] = Xs
        out4[key + "ns"] = ns.astype(np.int32)
        out4[key + "mean"] = mean
        out4[key + "cov"] = cov
        mean_nn, cov_nn = gp.predict(Xs, n=ns, noise=False, return_cov=True)
        out4[key + "cov_nonoise"] = cov_nn

    # ConstantMeanFunction and T (linear transform) branch, SE
    N, d = 48, 1
    n = np.zeros((N, d), dtype=int)
    X, y = synth(rs, N, d, n)
    y = y + 2.5
    params = np.array([0.9, 0.2])
    mu = gptools.ConstantMeanFunction(initial_params=[2.4])
    gp = gptools.GaussianProcess(se_kernel(d, params), mu=mu, X=X, y=y, err_y=0.05)
    key = "se_mu_"
    out3[key + "X"] = X
    out3[key + "y"] = y
    out3[key + "n"] = n.astype(np.int32)
    out3[key + "err_y"] = np.asarray(gp.err_y)
    out3[key + "params"] = params
    out3[key + "mu"] = np.float64(2.4)
    record_fit(out3, key, gp)
    Xs = rs.rand(16, d)
    ns = np.zeros((16, d), dtype=int)
    ns[8:] = 1
    mean, std = gp.predict(Xs, n=ns)
    out4[key + "Xs"] = Xs
    out4[key + "ns"] = ns.astype(np.int32)
    out4[key + "mean"] = mean
    out4[key + "std"] = std

    # T: 12 transformed observations of 36 latent points + 10 plain points
    d = 1
    Xq = rs.rand(36, d)
    T = rs.rand(12, 36) / 36.0
    yT = T.dot(np.sin(3.0 * Xq[:, 0])) + 0.01 * rs.randn(12)
    Xp = rs.rand(10, d)
    yp = np.sin(3.0 * Xp[:, 0]) + 0.01 * rs.randn(10)
    gp = gptools.GaussianProcess(se_kernel(d, np.array([1.2, 0.3])))
    gp.add_data(Xp, yp, err_y=0.01)
    gp.add_data(Xq, yT, err_y=0.01, T=T)
    key = "se_T_"
    out3[key + "Xp"] = Xp
    out3[key + "yp"] = yp
    out3[key + "Xq"] = Xq
    out3[key + "yT"] = yT
    out3[key + "T"] = T
    out3[key + "params"] = np.array([1.2, 0.3])
    out3[key + "Tfull"] = np.asarray(gp.T)
    record_fit(out3, key, gp)
    Xs = rs.rand(16, d)
    mean, std = gp.predict(Xs)
    out4[key + "Xs"] = Xs
    out4[key + "mean"] = mean
    out4[key + "std"] = std

    # analytic hyper-derivative of ll (use_hyper_deriv=True), SE with noise kernel
    N, d = 40, 2
    n = np.zeros((N, d), dtype=int)
    X, y = synth(rs, N, d, n)
    nk = gptools.DiagonalNoiseKernel(num_dim=d, initial_noise=0.1, noise_bound=(0.0, 5.0))
    gp = gptools.GaussianProcess(se_kernel(d, [1.0, 0.3, 0.4]), noise_k=nk, X=X, y=y, err_y=0.02,
                                 use_hyper_deriv=True)
    key = "se_hd_"
    gp.compute_K_L_alpha_ll()
    out3[key + "X"] = X
    out3[key + "y"] = y
    out3[key + "params"] = np.array([1.0, 0.3, 0.4])
    out3[key + "noise"] = np.float64(0.1)
    out3[key + "ll"] = np.float64(gp.ll)
    out3[key + "ll_deriv"] = np.asarray(gp.ll_deriv)

    save("g3_fit", **out3)
    save("g4_predict", **out4)


# ----------------------------------------------------------------------------
# G5: update_hyperparameters sweep
# ----------------------------------------------------------------------------
def gen_g5():
    rs = np.random.RandomState(505)
    out = {}
    N, d = 96, 2
    n = deriv_pattern(rs, N, d, 0.25, 1)
    X, y = synth(rs, N, d, n)
    for kname, cls in (("se", gptools.SquaredExponentialKernel), ("m52", gptools.Matern52Kernel)):
        k = cls(num_dim=d, initial_params=[1.0, 0.3, 0.3], param_bounds=[(1e-3, 10.0)] * 3)
        gp = gptools.GaussianProcess(k, X=X, y=y, err_y=0.05, n=n)
        thetas = [
            [1.0, 0.3, 0.3], [0.5, 0.2, 0.6], [2.0, 0.8, 0.1], [0.1, 0.05, 0.05],
            [5.0, 3.0, 3.0], [1.5, 0.31, 0.29], [0.9, 1.0, 1.0], [3.0, 0.15, 0.45],
            [11.0, 0.3, 0.3],    # out of bounds -> +inf (gaussian_process.py:1387-1389)
            [1.0, -0.1, 0.3],    # out of bounds -> +inf
            [1.0, 0.3, 1e-4],    # below the lower bound -> +inf
        ]
        vals = [gp.update_hyperparameters(np.asarray(t, dtype=float)) for t in thetas]
        out[kname + "_thetas"] = np.asarray(thetas)
        out[kname + "_negll"] = np.asarray(vals, dtype=float)
    out["X"] = X
    out["y"] = y
    out["n"] = n.astype(np.int32)
    out["err_y"] = np.float64(0.05)
    # non positive-definite: duplicated points, noiseless, diag_factor = 0 -> LinAlgError -> +inf
    Xd = np.vstack((X[:8], X[:8]))
    yd = np.concatenate((y[:8], y[:8]))
    nd = np.zeros((16, d), dtype=int)
    k = gptools.SquaredExponentialKernel(num_dim=d, initial_params=[1.0, 0.3, 0.3],
                                         param_bounds=[(1e-3, 10.0)] * 3)
    gp = gptools.GaussianProcess(k, X=Xd, y=yd, err_y=0.0, n=nd, diag_factor=0.0)
    out["nonpd_X"] = Xd
    out["nonpd_y"] = yd
    out["nonpd_negll"] = np.float64(gp.update_hyperparameters(np.array([1.0, 0.3, 0.3])))
    save("g5_update", **out)


# ----------------------------------------------------------------------------
# G6: demo (config 1)
# ----------------------------------------------------------------------------
def gen_g6():
    out = {}
    with open(os.path.join(ref_harness.REF_ROOT, "demo", "sample_data_core.pkl"), "rb") as f:
        core = pickle.load(f, encoding="latin1")
    with open(os.path.join(ref_harness.REF_ROOT, "demo", "sample_data_edge.pkl"), "rb") as f:
        edge = pickle.load(f, encoding="latin1")
    for nm, dat in (("core", core), ("edge", edge)):
        for kk in ("X", "y", "err_y"):
            out["%s_%s" % (nm, kk)] = np.asarray(dat[kk], dtype=float)

    def make_gp():
        hp = gptools.UniformJointPrior(0, 20) * gptools.GammaJointPriorAlt(1, 0.7)
        gp = gptools.GaussianProcess(gptools.SquaredExponentialKernel(hyperprior=hp))
        gp.add_data(core["X"], core["y"], err_y=core["err_y"])
        gp.add_data(0, 0, n=1)
        return gp

    gp = make_gp()
    demo_params = np.array([1.8849006111246833, 0.97760159723344708])   # demo/demo.py:191
    out["demo_params"] = demo_params
    out["negll_demo"] = np.float64(gp.update_hyperparameters(demo_params))
    out["ll_demo"] = np.float64(gp.ll)
    out["prior_demo"] = np.float64(gp.hyperprior(gp.params))
    out["alpha_demo"] = np.asarray(gp.alpha).ravel()
    out["L_demo"] = np.asarray(gp.L)
    Xs = np.linspace(0, 1.1, 400)
    out["X_star"] = Xs
    y_star, err_y_star = gp.predict(Xs)
    out["y_star"] = y_star
    out["err_y_star"] = err_y_star
    g_star, err_g_star = gp.predict(Xs, n=1)
    out["grad_y_star"] = g_star
    out["err_grad_y_star"] = err_g_star
    full = gp.predict(np.concatenate((Xs, Xs)),
                      n=np.concatenate((np.zeros_like(Xs), np.ones_like(Xs))), full_output=True)
    out["full_mean"] = full["mean"]
    out["full_std"] = full["std"]
    out["full_cov_diag"] = np.diag(full["cov"]).copy()
    out["full_cov_rows"] = full["cov"][[0, 199, 400, 799], :].copy()

    # MAP estimate, deterministic (no random starts): L-BFGS-B and SLSQP from a fixed start
    for method in ("SLSQP", "L-BFGS-B"):
        gp = make_gp()
        gp.update_hyperparameters(np.array([1.0, 1.0]))
        res, nres = gp.optimize_hyperparameters(method=method, random_starts=0, num_proc=0)
        out["map_%s_x" % method.replace("-", "")] = np.asarray(res.x)
        out["map_%s_fun" % method.replace("-", "")] = np.float64(res.fun)

    # with a fitted DiagonalNoiseKernel (demo/demo.py:126-128,158-159,192)
    hp = gptools.UniformJointPrior(0, 20) * gptools.GammaJointPriorAlt(1, 0.7)
    k_noise = gptools.DiagonalNoiseKernel(noise_bound=[0, 5])
    gpn = gptools.GaussianProcess(gptools.SquaredExponentialKernel(hyperprior=hp), noise_k=k_noise)
    gpn.add_data(core["X"], core["y"])
    gpn.add_data(0, 0, n=1)
    pn = np.array([1.7095365754195335, 1.222639837707701, 0.12181881916114756])   # demo/demo.py:192
    out["demo_noise_params"] = pn
    out["negll_demo_noise"] = np.float64(gpn.update_hyperparameters(pn))
    yn, en = gpn.predict(Xs)
    out["y_star_noise"] = yn
    out["err_y_star_noise"] = en
    save("g6_demo", **out)


# ----------------------------------------------------------------------------
# G7: tests/test_matern.py scenario with a seeded length-scale draw
# ----------------------------------------------------------------------------
def gen_g7():
    f_X = np.random.RandomState(0).randn(5, 2)
    f_y = f_X[:, 0] ** 2 + f_X[:, 1] ** 2
    g_y_0 = 2 * f_X[:, 0]
    g_y_1 = 2 * f_X[:, 1]
    length_scales = np.random.RandomState(7).lognormal(size=2).tolist()
    K1 = gptools.MaternKernelArb(num_dim=2, initial_params=[10, 5.0 / 2.0] + length_scales)
    K2 = gptools.Matern52Kernel(num_dim=2, initial_params=[10] + length_scales)
    gp1 = gptools.GaussianProcess(K1)
    gp2 = gptools.GaussianProcess(K2)
    gp1.add_data(f_X, f_y)
    gp1.add_data(f_X, g_y_0, n=np.vstack((np.ones(len(f_X)), np.zeros(len(f_X)))).T)
    gp1.add_data(f_X, g_y_1, n=np.vstack((np.zeros(len(f_X)), np.ones(len(f_X)))).T)
    k1 = gp1.compute_Kij(gp1.X, None, gp1.n, None)
    k2 = gp2.compute_Kij(gp1.X, None, gp1.n, None)
    np.testing.assert_array_almost_equal(k1, k2, decimal=8)
    save("g7_test_matern", X=np.asarray(gp1.X), n=np.asarray(gp1.n, dtype=np.int32),
         y=np.asarray(gp1.y), length_scales=np.asarray(length_scales),
         K_arb=np.asarray(k1, dtype=float), K_m52=np.asarray(k2, dtype=float))


# ----------------------------------------------------------------------------
# G8: rational-quadratic kernel, params [sigma_f, alpha, l_1 .. l_D]
# ----------------------------------------------------------------------------
def rq_kernel(d, params):
    return gptools.RationalQuadraticKernel(
        num_dim=d, initial_params=list(params), param_bounds=[(0.0, 1e3)] * (d + 2))


def gen_g8():
    rs = np.random.RandomState(808)
    out = {}
    # pair level: derivative orders 0..2 per point and dimension, combined order of a pair capped at 6
    for d in (1, 2, 3, 4):
        for tag, alpha in (("a", 1.7), ("b", 2.0), ("c", 0.35)):
            M = 160
            Xi = rs.rand(M, d)
            Xj = rs.rand(M, d)
            Xj[:16] = Xi[:16]                      # tau == 0
            Xj[16:24, 0] = Xi[16:24, 0]            # one component zero
            ni = rs.randint(0, 3, size=(M, d))
            nj = rs.randint(0, 3, size=(M, d))
            ni[rs.rand(M) < 0.3] = 0
            nj[rs.rand(M) < 0.3] = 0
            for m in range(M):
                while ni[m].sum() + nj[m].sum() > 6:
                    w = ni if rs.rand() < 0.5 else nj
                    c = rs.randint(d)
                    if w[m, c] > 0:
                        w[m, c] -= 1
            params = np.concatenate(([1.3, alpha], 0.2 + 0.5 * rs.rand(d)))
            k = rq_kernel(d, params)
            key = "pairs_d%d%s_" % (d, tag)
            out[key + "Xi"], out[key + "Xj"] = Xi, Xj
            out[key + "ni"], out[key + "nj"] = ni.astype(np.int32), nj.astype(np.int32)
            out[key + "params"] = params
            out[key + "k"] = np.asarray(k(Xi, Xj, ni, nj), dtype=float)
    # Gram matrices with value / first / second derivative rows
    for d in (1, 2, 3):
        N, P = 48, 24
        X = rs.rand(N, d)
        Xs = rs.rand(P, d)
        X[5] = X[4]
        Xs[3] = X[7]
        n = deriv_pattern(rs, N, d, 0.3, 2)
        ns = deriv_pattern(rs, P, d, 0.5, 1)
        params = np.concatenate(([0.9, 1.2], 0.2 + 0.4 * rs.rand(d)))
        gp = gptools.GaussianProcess(rq_kernel(d, params))
        key = "gram_d%d_" % d
        out[key + "X"], out[key + "Xs"] = X, Xs
        out[key + "n"], out[key + "ns"] = n.astype(np.int32), ns.astype(np.int32)
        out[key + "params"] = params
        out[key + "K"] = gp.compute_Kij(X, None, n, None)
        out[key + "Ks"] = gp.compute_Kij(X, Xs, n, ns)
    # fit + predict (value and first-derivative observations / predictions)
    for N, d in ((64, 2), (200, 3)):
        key = "fit_N%d_d%d_" % (N, d)
        n = deriv_pattern(rs, N, d, 0.25, 1)
        X, y = synth(rs, N, d, n)
        params = np.concatenate(([1.0, 1.5], 0.3 * np.ones(d)))
        gp = gptools.GaussianProcess(rq_kernel(d, params))
        gp.add_data(X, y, err_y=0.05, n=n)
        record_fit(out, key, gp, with_L=False)
        M = 24
        Xs = rs.rand(M, d)
        ns = np.zeros((M, d), dtype=int)
        ns[M // 2:, 0] = 1
        mean, std = gp.predict(Xs, n=ns)
        out[key + "X"], out[key + "y"], out[key + "n"] = X, y, n.astype(np.int32)
        out[key + "params"] = params
        out[key + "Xs"], out[key + "ns"] = Xs, ns.astype(np.int32)
        out[key + "mean"], out[key + "std"] = np.asarray(mean), np.asarray(std)
    save("g8_rq", **out)


# ----------------------------------------------------------------------------
# G9: ProductKernel
# ----------------------------------------------------------------------------
def gen_g9():
    rs = np.random.RandomState(909)
    out = {}
    d = 2
    M = 120
    Xi, Xj = rs.rand(M, d), rs.rand(M, d)
    Xj[:10] = Xi[:10]
    # SE * SE: orders 0..2 per point and dimension
    ni, nj = rs.randint(0, 3, size=(M, d)), rs.randint(0, 3, size=(M, d))
    ni[rs.rand(M) < 0.3] = 0
    nj[rs.rand(M) < 0.3] = 0
    p1, p2 = np.array([1.2, 0.3, 0.5]), np.array([0.8, 0.7, 0.25])
    k = se_kernel(d, p1) * se_kernel(d, p2)
    out["sese_Xi"], out["sese_Xj"], out["sese_ni"], out["sese_nj"] = Xi, Xj, ni.astype(np.int32), nj.astype(np.int32)
    out["sese_p1"], out["sese_p2"] = p1, p2
    out["sese_k"] = np.asarray(k(Xi, Xj, ni, nj), dtype=float)
    # SE * Matern52: at most one first derivative per point
    n1 = np.zeros((M, d), dtype=int)
    n2 = np.zeros((M, d), dtype=int)
    for m in range(M):
        if rs.rand() < 0.5:
            n1[m, rs.randint(d)] = 1
        if rs.rand() < 0.5:
            n2[m, rs.randint(d)] = 1
    k = se_kernel(d, p1) * m52_kernel(d, p2)
    out["sem_ni"], out["sem_nj"] = n1.astype(np.int32), n2.astype(np.int32)
    out["sem_k"] = np.asarray(k(Xi, Xj, n1, n2), dtype=float)
    # fit + predict with SE * Matern52, first-derivative rows
    N = 60
    n = deriv_pattern(rs, N, d, 0.25, 1)
    X, y = synth(rs, N, d, n)
    gp = gptools.GaussianProcess(se_kernel(d, p1) * m52_kernel(d, p2))
    gp.add_data(X, y, err_y=0.05, n=n)
    record_fit(out, "fit_", gp, with_L=False)
    Xs = rs.rand(16, d)
    ns = np.zeros((16, d), dtype=int)
    ns[8:, 1] = 1
    mean, std = gp.predict(Xs, n=ns)
    out["fit_X"], out["fit_y"], out["fit_n"] = X, y, n.astype(np.int32)
    out["fit_Xs"], out["fit_ns"] = Xs, ns.astype(np.int32)
    out["fit_mean"], out["fit_std"] = np.asarray(mean), np.asarray(std)
    save("g9_product", **out)


# ----------------------------------------------------------------------------
# G10: general-order MaternKernel, params [sigma_f, nu, l_1 .. l_D]
# ----------------------------------------------------------------------------
def matern_kernel(d, params):
    return gptools.MaternKernel(num_dim=d, initial_params=list(params), param_bounds=[(0.0, 1e3)] * (d + 2))


def gen_g10():
    rs = np.random.RandomState(1010)
    out = {}
    # pair level: half-integer, integer and general orders; derivative orders 0..1 per point and dimension (what
    # regression with gradient observations uses), tau == 0 rows, rows inside the reference's series region y <= 5e-4
    for d in (1, 2, 3):
        for tag, nu in (("a", 2.5), ("b", 1.5), ("c", 3.2), ("d", 2.0), ("e", 0.5), ("f", 4.7), ("g", 1.0)):
            M = 120
            Xi = rs.rand(M, d)
            Xj = rs.rand(M, d)
            Xj[:12] = Xi[:12]                                          # tau == 0: finite limits, zeros, NaN (nu <= 1)
            Xj[12:24] = Xi[12:24] + 2e-3 * (rs.rand(12, d) - 0.5)      # 0 < y <= 5e-4: one-term series / nu -+ 0.001
            ni = rs.randint(0, 2, size=(M, d))
            nj = rs.randint(0, 2, size=(M, d))
            ni[rs.rand(M) < 0.3] = 0
            nj[rs.rand(M) < 0.3] = 0
            params = np.concatenate(([1.3, nu], 0.2 + 0.5 * rs.rand(d)))
            k = matern_kernel(d, params)
            key = "pairs_d%d%s_" % (d, tag)
            out[key + "Xi"], out[key + "Xj"] = Xi, Xj
            out[key + "ni"], out[key + "nj"] = ni.astype(np.int32), nj.astype(np.int32)
            out[key + "params"] = params
            out[key + "k"] = np.asarray(k(Xi, Xj, ni, nj), dtype=float)
    # second derivatives on one side (combined order up to 4): the reference's own sums are accurate to ~1e-9 here
    for d in (1, 2):
        M = 80
        Xi, Xj = rs.rand(M, d), rs.rand(M, d)
        ni = rs.randint(0, 3, size=(M, d))
        nj = rs.randint(0, 3, size=(M, d))
        for m in range(M):
            while ni[m].sum() + nj[m].sum() > 4:
                w = ni if rs.rand() < 0.5 else nj
                c = rs.randint(d)
                if w[m, c] > 0:
                    w[m, c] -= 1
        params = np.concatenate(([0.8, 3.7], 0.3 + 0.4 * rs.rand(d)))
        key = "pairs2_d%d_" % d
        out[key + "Xi"], out[key + "Xj"] = Xi, Xj
        out[key + "ni"], out[key + "nj"] = ni.astype(np.int32), nj.astype(np.int32)
        out[key + "params"] = params
        out[key + "k"] = np.asarray(matern_kernel(d, params)(Xi, Xj, ni, nj), dtype=float)
    # Gram matrices with value / first-derivative rows
    for d in (1, 2, 3):
        N, P = 40, 20
        X = rs.rand(N, d)
        Xs = rs.rand(P, d)
        Xs[3] = X[7]
        n = deriv_pattern(rs, N, d, 0.3, 1)
        ns = deriv_pattern(rs, P, d, 0.5, 1)
        params = np.concatenate(([0.9, 2.2], 0.2 + 0.4 * rs.rand(d)))
        gp = gptools.GaussianProcess(matern_kernel(d, params))
        key = "gram_d%d_" % d
        out[key + "X"], out[key + "Xs"] = X, Xs
        out[key + "n"], out[key + "ns"] = n.astype(np.int32), ns.astype(np.int32)
        out[key + "params"] = params
        out[key + "K"] = gp.compute_Kij(X, None, n, None)
        out[key + "Ks"] = gp.compute_Kij(X, Xs, n, ns)
    # fit + predict (value and first-derivative observations / predictions); nu = 5/2 must agree with Matern52Kernel
    for N, d, nu in ((64, 2, 2.5), (160, 3, 1.8)):
        key = "fit_N%d_d%d_" % (N, d)
        n = deriv_pattern(rs, N, d, 0.25, 1)
        X, y = synth(rs, N, d, n)
        params = np.concatenate(([1.0, nu], 0.3 * np.ones(d)))
        gp = gptools.GaussianProcess(matern_kernel(d, params))
        gp.add_data(X, y, err_y=0.05, n=n)
        record_fit(out, key, gp, with_L=False)
        M = 24
        Xs = rs.rand(M, d)
        ns = np.zeros((M, d), dtype=int)
        ns[M // 2:, 0] = 1
        mean, std = gp.predict(Xs, n=ns)
        out[key + "X"], out[key + "y"], out[key + "n"] = X, y, n.astype(np.int32)
        out[key + "params"] = params
        out[key + "Xs"], out[key + "ns"] = Xs, ns.astype(np.int32)
        out[key + "mean"], out[key + "std"] = np.asarray(mean), np.asarray(std)
    save("g10_matern", **out)


# ----------------------------------------------------------------------------
# G11 (round 5): combined derivative orders 9 .. 12 of a pair -- the device builder's limit went from 8 to 16 (GPT_RQ_MAXORD);
# the reference walks every set partition of the derivative multiset (core.py:752-816: Bell(12) = 4.2 million per pair), so a
# handful of pairs per order is what a generation run affords (RQ order 12: ~3 minutes per pair)
# ----------------------------------------------------------------------------
def orders_with_sum(rs, d, tot):
    """ni, nj (d,) with ni.sum() + nj.sum() == tot, spread over both points and all dimensions."""
    v = np.zeros(2 * d, dtype=int)
    for _ in range(tot):
        v[rs.randint(2 * d)] += 1
    return v[:d], v[d:]


def gen_g11():
    rs = np.random.RandomState(1111)
    out = {}
    plans = (("rq", 2, ((9, 5), (10, 3), (11, 2), (12, 1))), ("rq", 3, ((9, 4), (10, 2))),
             ("matern", 2, ((9, 3), (10, 2))), ("prod", 2, ((9, 3), (10, 1))))
    for name, d, plan in plans:
        Xi, Xj, ni, nj = [], [], [], []
        for tot, cnt in plan:
            for _ in range(cnt):
                a, b = orders_with_sum(rs, d, tot)
                Xi.append(rs.rand(d)); Xj.append(rs.rand(d)); ni.append(a); nj.append(b)
        Xi, Xj, ni, nj = np.array(Xi), np.array(Xj), np.array(ni), np.array(nj)
        if name == "rq":
            params = np.concatenate(([1.3, 1.7], 0.3 + 0.5 * rs.rand(d)))
            k = rq_kernel(d, params)
        elif name == "matern":
            params = np.concatenate(([1.1, 3.2], 0.4 + 0.4 * rs.rand(d)))
            k = gptools.MaternKernel(num_dim=d, initial_params=list(params), param_bounds=[(0.0, 1e3)] * (d + 2))
        else:
            p1 = np.concatenate(([1.1], 0.4 + 0.4 * rs.rand(d)))
            p2 = np.concatenate(([0.9, 1.7], 0.3 + 0.5 * rs.rand(d)))
            params = np.concatenate((p1, p2))
            k = se_kernel(d, p1) * rq_kernel(d, p2)
            out["prod_d%d_p1" % d], out["prod_d%d_p2" % d] = p1, p2
        key = "%s_d%d_" % (name, d)
        out[key + "Xi"], out[key + "Xj"] = Xi, Xj
        out[key + "ni"], out[key + "nj"] = ni.astype(np.int32), nj.astype(np.int32)
        out[key + "params"] = params
        vals = []
        for m in range(len(Xi)):                       # one pair per call: the progress shows, a run can be interrupted
            vals.append(float(np.asarray(k(Xi[m:m + 1], Xj[m:m + 1], ni[m:m + 1], nj[m:m + 1]))[0]))
            print("  %s order %d: %.15g" % (key, ni[m].sum() + nj[m].sum(), vals[-1]), flush=True)
        out[key + "k"] = np.array(vals)
    save("g11_high_orders", **out)


# ----------------------------------------------------------------------------
# G12 (round 6): the reference's independent-evaluation consumers and its posterior samples (8f-4)
# ----------------------------------------------------------------------------
def gen_g12():
    out = {}
    rs = np.random.RandomState(1212)
    # --- (a) compute_ll_matrix: 2 free parameters (l_2 fixed), grid reaching beyond the bounds of sigma_f -> -inf row;
    #         a Gamma hyperprior so that the prior term is not a constant over the grid
    N, d = 72, 2
    n = deriv_pattern(rs, N, d, 0.25, 1)
    X, y = synth(rs, N, d, n)
    out["grid_X"], out["grid_y"], out["grid_n"] = X, y, n.astype(np.int32)
    for kname, cls in (("se", gptools.SquaredExponentialKernel), ("m52", gptools.Matern52Kernel)):
        k = cls(num_dim=d, initial_params=[1.2, 0.35, 0.4], param_bounds=[(1e-3, 10.0)] * 3,
                fixed_params=[False, False, True])
        gp = gptools.GaussianProcess(k, X=X, y=y, err_y=0.05, n=n)
        before = np.array(gp.free_params[:], dtype=float)
        ll, pv = gp.compute_ll_matrix([(0.4, 12.0), (0.1, 0.9)], [5, 4])       # sigma_f = 12 is out of bounds
        out["grid_%s_ll" % kname] = np.asarray(ll, dtype=float)
        out["grid_%s_p0" % kname], out["grid_%s_p1" % kname] = np.asarray(pv[0]), np.asarray(pv[1])
        out["grid_%s_free_after" % kname] = np.array(gp.free_params[:], dtype=float)
        out["grid_%s_ll_after" % kname] = np.float64(gp.ll)
        assert np.array_equal(before, out["grid_%s_free_after" % kname])
    # three free parameters incl. the noise kernel's, one (bounds, num_pts) pair for all of them; Gamma hyperpriors
    k = gptools.SquaredExponentialKernel(
        num_dim=d, initial_params=[1.0, 0.3, 0.5], fixed_params=[False, False, True],
        hyperprior=gptools.GammaJointPriorAlt([1.0, 0.4, 0.5], [0.5, 0.2, 0.3]))
    nk = gptools.DiagonalNoiseKernel(num_dim=d, initial_noise=0.1, fixed_noise=False, noise_bound=(1e-3, 1.0))
    gp = gptools.GaussianProcess(k, noise_k=nk, X=X, y=y, err_y=0.02, n=n)
    ll, pv = gp.compute_ll_matrix((0.05, 0.8), 3)
    out["grid3_ll"] = np.asarray(ll, dtype=float)
    out["grid3_p"] = np.asarray(pv, dtype=float)
    # --- (b) random starts: the draws of the hyperprior (global numpy state), one minimize per draw, the best one kept
    N = 40
    nb = deriv_pattern(rs, N, 1, 0.2, 1)
    Xb = rs.rand(N, 1)
    yb = np.sin(4.0 * Xb[:, 0]) + 0.05 * rs.randn(N)
    yb[nb[:, 0] == 1] = 4.0 * np.cos(4.0 * Xb[nb[:, 0] == 1, 0])
    out["rs_X"], out["rs_y"], out["rs_n"] = Xb, yb, nb.astype(np.int32)
    for tag, deriv in (("fd", False), ("hd", True)):
        k = gptools.SquaredExponentialKernel(num_dim=1, initial_params=[1.0, 0.3], param_bounds=[(0.05, 20.0), (0.02, 3.0)])
        gp = gptools.GaussianProcess(k, X=Xb, y=yb, err_y=0.05, n=nb, use_hyper_deriv=deriv)
        np.random.seed(4242)
        draws = gp.hyperprior.random_draw(size=5).T[:, ~np.asarray(gp.fixed_params, dtype=bool)]   # (:702-703)
        out["rs_%s_draws" % tag] = np.asarray(draws, dtype=float)
        per = []
        for s in draws:       # what _OptimizeHyperparametersEval does per start (gaussian_process.py:2459-2466)
            import scipy.optimize
            r = scipy.optimize.minimize(gp.update_hyperparameters, s, method="SLSQP", jac=deriv,
                                        bounds=np.asarray(gp.free_param_bounds, dtype=float))
            per.append(np.concatenate((r.x, [r.fun, float(r.success)])))
        out["rs_%s_per_start" % tag] = np.asarray(per)
        np.random.seed(4242)
        res, count = gp.optimize_hyperparameters(method="SLSQP", random_starts=5, num_proc=0)
        out["rs_%s_x" % tag], out["rs_%s_fun" % tag] = np.asarray(res.x, dtype=float), np.float64(res.fun)
        out["rs_%s_count" % tag] = np.int64(count)
        out["rs_%s_params_after" % tag] = np.array(gp.params[:], dtype=float)
        out["rs_%s_ll_after" % tag] = np.float64(gp.ll)
    # --- (c) draw_sample / predict(return_samples=True)
    N, d, M = 48, 2, 12
    n = deriv_pattern(rs, N, d, 0.25, 1)
    X, y = synth(rs, N, d, n)
    Xs = rs.rand(M, d)
    ns = np.zeros((M, d), dtype=int)
    ns[M - 4:, 0] = 1
    out["samp_X"], out["samp_y"], out["samp_n"] = X, y, n.astype(np.int32)
    out["samp_Xs"], out["samp_ns"] = Xs, ns.astype(np.int32)
    u = rs.randn(M, 3)
    uu = rs.rand(M, 2)
    u6 = rs.randn(M, 6)
    out["samp_u"], out["samp_uu"], out["samp_u6"] = u, uu, u6
    for kname, cls in (("se", gptools.SquaredExponentialKernel), ("m52", gptools.Matern52Kernel)):
        k = cls(num_dim=d, initial_params=[1.1, 0.4, 0.3], param_bounds=[(1e-3, 10.0)] * 3)
        nk = gptools.DiagonalNoiseKernel(num_dim=d, initial_noise=0.07, fixed_noise=True)
        gp = gptools.GaussianProcess(k, noise_k=nk, X=X, y=y, err_y=0.05, n=n)
        key = "samp_%s_" % kname
        out[key + "chol"] = np.asarray(gp.draw_sample(Xs, n=ns, rand_vars=u))
        out[key + "chol_noise"] = np.asarray(gp.draw_sample(Xs, n=ns, rand_vars=u, noise=True))
        out[key + "chol_df"] = np.asarray(gp.draw_sample(Xs, n=ns, rand_vars=u, diag_factor=1e6))
        out[key + "chol_uniform"] = np.asarray(gp.draw_sample(Xs, n=ns, rand_vars=uu, rand_type="uniform"))
        out[key + "chol_scalar_n"] = np.asarray(gp.draw_sample(Xs, n=0, rand_vars=u))
        # eig: ascending eigenvalues; noise=True keeps them well separated from zero so that Q is reproducible
        for ms in (None, "left value", "right value", "left slope", "right slope", "left concavity", "right concavity"):
            tagm = "none" if ms is None else ms.replace(" ", "_")
            out[key + "eig_" + tagm] = np.asarray(
                gp.draw_sample(Xs, n=0, rand_vars=u, method="eig", modify_sign=ms, noise=True))
        out[key + "eig_num5"] = np.asarray(
            gp.draw_sample(Xs, n=0, rand_vars=u, method="eig", num_eig=5, modify_sign="left value", noise=True))
        # predict(return_samples=True): samples through samp_kwargs (deterministic), and full_output's other entries
        o = gp.predict(Xs, n=ns, full_output=True, return_samples=True, samp_kwargs={"rand_vars": u})
        out[key + "pred_mean"], out[key + "pred_std"] = np.asarray(o["mean"]), np.asarray(o["std"])
        out[key + "pred_cov"], out[key + "pred_samp"] = np.asarray(o["cov"]), np.asarray(o["samp"])
        # the seeded multivariate_normal route (no rand_vars): noise=True keeps the SVD's vectors well defined
        np.random.seed(99)
        o = gp.predict(Xs, n=0, noise=True, full_output=True, return_samples=True, num_samples=4)
        out[key + "pred_mvn_samp"] = np.asarray(o["samp"])
        out[key + "pred_mvn_cov"] = np.asarray(o["cov"])
        # full_MC: mean / covariance estimated from the samples (gaussian_process.py:1003-1005)
        o = gp.predict(Xs, n=0, noise=True, full_output=True, full_MC=True, num_samples=6,
                       samp_kwargs={"rand_vars": u6}, ddof=1)
        out[key + "mc_mean"], out[key + "mc_cov"], out[key + "mc_samp"] = (
            np.asarray(o["mean"]), np.asarray(o["cov"]), np.asarray(o["samp"]))
    save("g12_batch_samples", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g34", "g5", "g6", "g7", "g8", "g9", "g10", "g12"]
    if "g11" in which:
        gen_g11()
    if "g12" in which:
        gen_g12()
    if "g10" in which:
        gen_g10()
    if "g1" in which:
        gen_g1()
    if "g2" in which:
        gen_g2()
    if "g34" in which:
        gen_g3_g4()
    if "g5" in which:
        gen_g5()
    if "g6" in which:
        gen_g6()
    if "g7" in which:
        gen_g7()
    if "g8" in which:
        gen_g8()
    if "g9" in which:
        gen_g9()
