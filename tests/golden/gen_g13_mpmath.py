#!/usr/bin/env python
"""g13_high_orders_mpmath.npz: kernel values at COMBINED DERIVATIVE ORDERS 13 .. 16 of a pair from an independent
high-precision evaluation (mpmath, 90 digits) -- the orders the device builder accepts (GPT_RQ_MAXORD = 16) beyond what the
reference itself can be run at (g11: orders 9 .. 12; it walks every set partition of the derivative multiset, Bell(13) = 27.6
million per pair).  ADVICE r5: orders 13 .. 16 were checked only against the oracle, which shares the device's regrouped sums.

Method (shares nothing with the oracle's / the device's Faa di Bruno grouping or their closed forms of f^(m)): multivariate
Taylor arithmetic.  k(tau) = sigma^2 f(y(tau)), y quadratic in tau, so y(tau0 + h) = y0 + p(h) exactly; the Taylor coefficients
f^(m)(y0) / m! of the OUTER function come from mpmath's numerical differentiation of f itself (mp.taylor: (1 + y)^-alpha;
2^(1-nu) / Gamma(nu) y^(nu/2) K_nu(sqrt y) through mp.besselk; exp(-y)), the composition sum_m c_m p(h)^m is carried out on
truncated polynomials in h, a product kernel is the product of two such polynomials, and the mixed partial derivative is
prod_d n_d! times the coefficient of h^n, times (-1)^(sum nj) for the derivatives with respect to the second argument
(ref: kernel/core.py:746).  No reference code is involved; mpmath only.  ~15 minutes (the Bessel functions at nu = 25).
"""
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
mp.mp.dps = 90


def poly_mul(a, b, maxdeg):
    out = {}
    for ia, ca in a.items():
        for ib, cb in b.items():
            idx = tuple(x + y for x, y in zip(ia, ib))
            if all(i <= m for i, m in zip(idx, maxdeg)):
                out[idx] = out.get(idx, 0) + ca * cb
    return out


def compose(outer, y0, p, maxdeg):
    M = sum(maxdeg)
    c = mp.taylor(outer, y0, M)
    res, pw = {}, {tuple(0 for _ in maxdeg): mp.mpf(1)}
    for m in range(M + 1):
        for idx, v in pw.items():
            res[idx] = res.get(idx, 0) + c[m] * v
        pw = poly_mul(pw, p, maxdeg)
    return res


def quad(tau, scale, maxdeg):
    d = len(tau)
    y0 = sum(s * t * t for s, t in zip(scale, tau))
    p = {}
    for k in range(d):
        p[tuple(1 if i == k else 0 for i in range(d))] = 2 * scale[k] * tau[k]
        if maxdeg[k] >= 2:
            p[tuple(2 if i == k else 0 for i in range(d))] = scale[k]
    return y0, p


def kernel_jet(name, params, tau, maxdeg):
    P = [mp.mpf(float(v)) for v in params]
    if name == "se":
        y0, p = quad(tau, [1 / (2 * l * l) for l in P[1:]], maxdeg)
        outer = lambda y: mp.exp(-y)
    elif name == "rq":
        al = P[1]
        y0, p = quad(tau, [1 / (2 * al * l * l) for l in P[2:]], maxdeg)
        outer = lambda y: (1 + y) ** (-al)
    else:
        nu = P[1]
        y0, p = quad(tau, [2 * nu / (l * l) for l in P[2:]], maxdeg)
        outer = lambda y: 2 ** (1 - nu) / mp.gamma(nu) * mp.sqrt(y) ** nu * mp.besselk(nu, mp.sqrt(y))
    return {k: P[0] ** 2 * v for k, v in compose(outer, y0, p, maxdeg).items()}


def orders_with_sum(rs, d, tot):
    v = np.zeros(2 * d, dtype=int)
    for _ in range(tot):
        v[rs.randint(2 * d)] += 1
    return v[:d], v[d:]


def main():
    rs = np.random.RandomState(1313)
    out = {}
    cases = (("rq", 2, [1.3, 1.7, 0.45, 0.62], 1.0, 3), ("rq", 3, [0.8, 0.35, 0.5, 0.7, 0.4], 1.0, 2),
             ("rq", 2, [1.3, 2.0, 0.45, 0.62], 0.05, 2),            # close points: y ~ 1e-2
             ("matern", 2, [1.1, 3.2, 0.5, 0.7], 1.0, 2), ("matern", 2, [0.9, 0.6, 0.5, 0.7], 1.0, 1),
             ("matern", 2, [0.9, 25.0, 0.8, 0.9], 1.0, 1), ("matern", 2, [1.0, 7.5, 0.6, 0.5], 0.1, 1),
             ("prod", 2, [1.1, 0.55, 0.65, 0.9, 1.7, 0.5, 0.8], 1.0, 2))
    for ci, (name, d, params, spread, per_order) in enumerate(cases):
        Xi, Xj, ni, nj, vals = [], [], [], [], []
        for tot in (13, 14, 15, 16):
            for _ in range(per_order):
                a, b = orders_with_sum(rs, d, tot)
                xi = rs.rand(d)
                xj = xi + spread * (rs.rand(d) - 0.5)
                tau = [mp.mpf(float(u)) - mp.mpf(float(v)) for u, v in zip(xi, xj)]
                n = tuple(int(p + q) for p, q in zip(a, b))
                if name == "prod":
                    j1 = kernel_jet("se", params[:1 + d], tau, n)
                    j2 = kernel_jet("rq", params[1 + d:], tau, n)
                    jet = poly_mul(j1, j2, n)
                else:
                    jet = kernel_jet(name, params, tau, n)
                val = jet[n] * mp.fprod(mp.factorial(q) for q in n) * (-1) ** int(b.sum())
                Xi.append(xi); Xj.append(xj); ni.append(a); nj.append(b); vals.append(float(val))
                print("case %d %s order %d: %.17g" % (ci, name, tot, float(val)), flush=True)
        key = "c%d_" % ci
        out[key + "kernel"] = np.array(name)
        out[key + "params"] = np.array(params, dtype=float)
        out[key + "Xi"], out[key + "Xj"] = np.array(Xi), np.array(Xj)
        out[key + "ni"], out[key + "nj"] = np.array(ni, dtype=np.int32), np.array(nj, dtype=np.int32)
        out[key + "k"] = np.array(vals)
    out["ncases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, "g13_high_orders_mpmath.npz"), **out)
    print("wrote g13_high_orders_mpmath.npz")


if __name__ == "__main__":
    main()
