// kpair.hpp -- per-pair covariance device functions k(xi, xj, ni, nj) for gfx950.
//
// Mathematical content follows the reference; the formulation is this repo's own:
//   * SquaredExponentialKernel  (ref: gptools/kernel/squared_exponential.py:110-174).  The
//     reference multiplies (-1)^{sum nj} * prod_d (-1/(sqrt2 l_d))^{n_d} H_{n_d}(tau_d/(sqrt2 l_d))
//     with physicists' Hermite H.  Using H_n(x) = 2^{n/2} He_n(sqrt2 x) this is
//         (-1)^{sum ni} * prod_d l_d^{-n_d} He_{n_d}(tau_d / l_d),     n_d = ni_d + nj_d,
//     which needs no sqrt/pow and evaluates He by the upward three-term recurrence.
//     Hyper-derivative wrt l_j (ref :134-158) is written division-free:
//         f_j * t = l^-n [ He_n(u) (tau^2/l^3 - n/l) - n tau/l^2 He_{n-1}(u) ],  u = tau/l.
//   * Matern52Kernel  (ref: gptools/kernel/matern.py:545-555, gptools/kernel/src/matern.c:61-186),
//     all four derivative classes and their r^2 == 0 limits.
//   * DiagonalNoiseKernel / ZeroKernel  (ref: gptools/kernel/noise.py:103-110, :152).
//   * RationalQuadraticKernel  (ref: gptools/kernel/rational_quadratic.py:30-164 through ChainRuleKernel.__call__,
//     gptools/kernel/core.py:691-816): k = sigma^2 (-1)^{sum nj} d^n/dtau^n y^-alpha, y = 1 + sum tau_d^2/(2 alpha l_d^2).
//     The reference walks every set partition of the derivative multiset (Faa di Bruno).  y is quadratic in tau, so only
//     partitions into singletons (factor y1_d = tau_d/(alpha l_d^2)) and equal-index pairs (y2_d = 1/(alpha l_d^2))
//     contribute; grouped by the number of pairs j_d per dimension the sum is
//         sum_j prod_d [ n_d!/(j_d!(n_d-2j_d)! 2^j_d) y2_d^j_d y1_d^(n_d-2j_d) ] f^(sum_d (n_d-j_d))(y),
//         f^(m)(y) = (-alpha)(-alpha-1)...(-alpha-m+1) y^(-alpha-m),
//     accumulated here as a product of per-dimension polynomials in a marker for the number of blocks.
#pragma once
#include "common.hpp"

#define GPT_SQRT5 2.2360679774997898
#define GPT_FIVE_THIRDS 1.6666666666666667

template <int D>
__device__ __forceinline__ double se_pair(const KParams &kp, const double *xi, const double *xj,
                                          const int *ni, const int *nj)
{
    double tau[D];
    double r2 = 0.0;
    int nsum = 0, nisum = 0;
#pragma unroll
    for (int d = 0; d < D; d++) {
        tau[d] = xi[d] - xj[d];
        // core.py:416: a 0/0 term counts as 0 (tau * inf would be NaN)
        double t = (tau[d] == 0.0) ? 0.0 : tau[d] * kp.inv_l[d];
        r2 = fma(t, t, r2);
        nsum += ni[d] + nj[d];
        nisum += ni[d];
    }
    double k = kp.sigma * kp.sigma * exp(-0.5 * r2);
    const int hd = kp.hyper_deriv;
    if (nsum != 0 || hd > 0) {
        double prod = 1.0;
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int n = ni[d] + nj[d];
            const bool is_hd = (hd == d + 1);
            if (n > 0 || is_hd) {
                const double il = kp.inv_l[d];
                const double u = tau[d] * il;
                double hm1 = 0.0, h = 1.0, lpow = 1.0;       // He_{n-1}, He_n, l^-n
                for (int q = 0; q < n; q++) {
                    const double hn = u * h - (double)q * hm1;
                    hm1 = h;
                    h = hn;
                    lpow *= il;
                }
                double f = lpow * h;
                if (is_hd) {
                    const double dn = (double)n;
                    f = lpow * (h * (tau[d] * tau[d] * il * il * il - dn * il) - dn * tau[d] * il * il * hm1);
                }
                prod *= f;
            }
        }
        k = ((nisum & 1) ? -prod : prod) * k;
    }
    if (hd == 0) return (kp.sigma != 0.0) ? 2.0 * k / kp.sigma : 0.0;
    return k;
}

template <int D>
__device__ __forceinline__ double m52_pair(const KParams &kp, const double *xi, const double *xj,
                                           const int *ni, const int *nj)
{
    double r2 = 0.0;
    int ii = -1, ij = -1;
    double di = 0.0, dj = 0.0, ivi = 0.0, ivj = 0.0;   // displacement / inverse variance at the derivative dims
#pragma unroll
    for (int d = D - 1; d >= 0; d--) {                  // descending so the FIRST index with n == 1 wins (matern.c:32-39)
        const double disp = xi[d] - xj[d];
        r2 = fma(disp * disp, kp.inv_var[d], r2);
        if (ni[d] == 1) { ii = d; di = disp; ivi = kp.inv_var[d]; }
        if (nj[d] == 1) { ij = d; dj = disp; ivj = kp.inv_var[d]; }
    }
    double v;
    if (r2 == 0.0) {                                    // matern.c:83-84, :100-101, :123-127
        if (ii < 0 && ij < 0) v = 1.0;
        else if (ii >= 0 && ij >= 0 && ii == ij) v = GPT_FIVE_THIRDS * ivi;
        else v = 0.0;
    } else {
        const double r = sqrt(r2);
        const double s5r = GPT_SQRT5 * r;
        const double e = exp(-s5r);
        if (ii < 0 && ij < 0) {
            v = (1.0 + s5r + GPT_FIVE_THIRDS * r2) * e;
        } else if (ij < 0) {
            v = -GPT_FIVE_THIRDS * (1.0 + s5r) * e * (di * ivi);
        } else if (ii < 0) {
            v = GPT_FIVE_THIRDS * (1.0 + s5r) * e * (dj * ivj);       // arguments swapped => sign flips (matern.c:182-184)
        } else {
            double d2r = (di * ivi) * (dj * ivj);
            const double cross = d2r;
            if (ii == ij) d2r -= r2 * ivi;
            const double dk_over_r = -GPT_FIVE_THIRDS * (1.0 + s5r) * e;
            const double d2k = GPT_FIVE_THIRDS * (5.0 * r2 - s5r - 1.0) * e;
            v = (dk_over_r * d2r - d2k * cross) / r2;                  // term1 + term2, matern.c:143-146
        }
    }
    return kp.sigma * kp.sigma * v;
}

template <int D>
__device__ __forceinline__ double noise_pair(const KParams &kp, const double *xi, const double *xj,
                                             const int *ni, const int *nj)
{
    if (!kp.symmetric) return 0.0;
    bool hit = true;
#pragma unroll
    for (int d = 0; d < D; d++) hit = hit && (xi[d] == xj[d]) && (ni[d] == kp.noise_n[d]) && (nj[d] == kp.noise_n[d]);
    const double val = hit ? kp.sigma * kp.sigma : 0.0;
    return (kp.hyper_deriv < 0) ? val : 2.0 * val / kp.sigma;
}

template <int D>
__device__ __forceinline__ double rq_pair(const KParams &kp, const double *xi, const double *xj,
                                          const int *ni, const int *nj)
{
    double r2 = 0.0;
    int ntot = 0, njtot = 0;
#pragma unroll
    for (int d = 0; d < D; d++) {
        const double tau = xi[d] - xj[d];
        const double t = (tau == 0.0) ? 0.0 : tau * kp.inv_l[d];
        r2 = fma(t, t, r2);
        ntot += ni[d] + nj[d];
        njtot += nj[d];
    }
    const double alpha = kp.alpha;
    const double y = 1.0 + r2 / (2.0 * alpha);
    const double s2 = kp.sigma * kp.sigma;
    // y^-alpha as exp(-alpha log y): y >= 1, so |alpha log y| stays small and the few-ulp error of the product is far
    // inside the parity tolerance (1e-11 relative against the reference); the correctly rounded pow() costs 3x the
    // whole pair otherwise
    const double p0 = exp(-alpha * log(y));
    if (ntot == 0) return s2 * p0;
    if (ntot <= 2) {
        // the common cases in closed form (first-derivative observations on one or both sides, one second derivative):
        //   d/dtau_a f(y) = f'(y) y1_a,   d2/dtau_a dtau_b f(y) = f''(y) y1_a y1_b + [a == b] f'(y) y2_a
        int da = -1, db = -1;
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int n = ni[d] + nj[d];
            if (n >= 1) { if (da < 0) da = d; else db = d; }
            if (n == 2) db = d;
        }
        const double iy = 1.0 / y;
        const double f1 = -alpha * p0 * iy;                       // f'(y)
        const double y2a = kp.inv_var[da] / alpha;
        const double y1a = (xi[da] - xj[da]) * y2a;
        double v;
        if (ntot == 1) {
            v = f1 * y1a;
        } else {
            const double y1b = (xi[db] - xj[db]) * (kp.inv_var[db] / alpha);
            const double f2 = -(alpha + 1.0) * f1 * iy;           // f''(y)
            v = f2 * y1a * y1b + ((da == db) ? f1 * y2a : 0.0);
        }
        return s2 * ((njtot & 1) ? -v : v);
    }
    // c[m]: coefficient of f^(m)(y) so far (host code rejects ntot > GPT_RQ_MAXORD)
    double c[GPT_RQ_MAXORD + 1];
#pragma unroll
    for (int m = 0; m <= GPT_RQ_MAXORD; m++) c[m] = (m == 0) ? 1.0 : 0.0;
    int deg = 0;
#pragma unroll
    for (int d = 0; d < D; d++) {
        const int n = ni[d] + nj[d];
        if (n == 0) continue;
        const double y2 = kp.inv_var[d] / alpha;
        const double y1 = (xi[d] - xj[d]) * y2;
        // per-dimension polynomial: pd[n - j] = n!/(j!(n-2j)! 2^j) y2^j y1^(n-2j), j = 0 .. n/2
        double pd[GPT_RQ_MAXORD + 1];
#pragma unroll
        for (int m = 0; m <= GPT_RQ_MAXORD; m++) pd[m] = 0.0;
        double y1pow[GPT_RQ_MAXORD + 1];
        y1pow[0] = 1.0;
#pragma unroll
        for (int m = 1; m <= GPT_RQ_MAXORD; m++) y1pow[m] = y1pow[m - 1] * y1;
        double coef = 1.0, y2pow = 1.0;
        for (int j = 0; 2 * j <= n; j++) {
            if (j > 0) {
                coef = coef * (double)((n - 2 * j + 2) * (n - 2 * j + 1)) / (2.0 * (double)j);
                y2pow *= y2;
            }
            pd[n - j] = coef * y2pow * y1pow[n - 2 * j];
        }
        double cn[GPT_RQ_MAXORD + 1];
#pragma unroll
        for (int m = 0; m <= GPT_RQ_MAXORD; m++) cn[m] = 0.0;
        for (int a = 0; a <= deg; a++)
            for (int b = (n + 1) / 2; b <= n; b++)
                if (a + b <= GPT_RQ_MAXORD) cn[a + b] = fma(c[a], pd[b], cn[a + b]);
        deg += n;
#pragma unroll
        for (int m = 0; m <= GPT_RQ_MAXORD; m++) c[m] = cn[m];
    }
    // sum_m c[m] poch(1 - alpha - m, m) y^(-alpha - m)
    const double iy = 1.0 / y;
    double v = 0.0, poch = 1.0, ypow = p0;
    for (int m = 1; m <= deg && m <= GPT_RQ_MAXORD; m++) {
        poch *= -(alpha + (double)(m - 1));
        ypow *= iy;
        v = fma(c[m] * poch, ypow, v);
    }
    return s2 * ((njtot & 1) ? -v : v);
}

template <int KID, int D>
__device__ __forceinline__ double any_pair(const KParams &kp, const double *xi, const double *xj,
                                           const int *ni, const int *nj)
{
    if (KID == GPT_KERNEL_SE) return se_pair<D>(kp, xi, xj, ni, nj);
    if (KID == GPT_KERNEL_M52) return m52_pair<D>(kp, xi, xj, ni, nj);
    if (KID == GPT_KERNEL_DIAGNOISE) return noise_pair<D>(kp, xi, xj, ni, nj);
    if (KID == GPT_KERNEL_RQ) return rq_pair<D>(kp, xi, xj, ni, nj);
    return 0.0;
}
