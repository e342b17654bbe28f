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

// exp(-x) for x >= 0: k = rint(-x log2 e), r = -x - k ln 2 in two fused steps (|r| <= ln2 / 2), then
// exp(r) = 1 + r (1 + r q(r)) with q of degree 9 (Chebyshev interpolant of (e^r - 1 - r) / r^2, error 1.7e-17 relative;
// scratch note in DESIGN.md section 4), scaled by 2^k with one v_ldexp.  x is clamped at 800 (exp(-800) = 0 in double
// precision) so that k fits v_cvt_i32; the clamp (v_max) drops a NaN, which every caller multiplies back in through the
// factors that carry r.  18 instructions against the ~35 of the library exp with its overflow / denormal handling,
// within 1 ulp of it (the Matern-5/2 builder is bound by its arithmetic).
__device__ __forceinline__ double exp_neg(double x)
{
    const double t = fmax(-x, -800.0);
    const double k = __builtin_rint(t * 1.4426950408889634);
    double r = fma(k, -6.93147180369123816490e-01, t);
    r = fma(k, -1.90821492927058770002e-10, r);
    double q = 2.5100274028176466e-08;
    q = fma(q, r, 2.7620076086460433e-07);
    q = fma(q, r, 2.7557268773616192e-06);
    q = fma(q, r, 2.48015213222418e-05);
    q = fma(q, r, 1.9841269862757975e-04);
    q = fma(q, r, 1.3888888917196432e-03);
    q = fma(q, r, 8.3333333333301673e-03);
    q = fma(q, r, 4.1666666666624164e-02);
    q = fma(q, r, 1.6666666666666669e-01);
    q = fma(q, r, 5.0000000000000011e-01);
    q = fma(q, r, 1.0);
    q = fma(q, r, 1.0);
    return __builtin_amdgcn_ldexp(q, (int)k);
}

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
        // core.py:416: a 0/0 term counts as 0 (tau * inf would be NaN) -- only where a length scale IS zero (wave-uniform
        // flag from the host, api.hip make_kparams): the compare and two selects per dimension were ~10 % of the plain pair
        double t = tau[d] * kp.inv_l[d];
        if (kp.zero_l) t = (tau[d] == 0.0) ? 0.0 : t;
        r2 = fma(t, t, r2);
        nsum += ni[d] + nj[d];
        nisum += ni[d];
    }
    // exp(-r2 / 2) by the builder's own exp(-x) (18 instructions, within 1 ulp of the library's ~35: exp_neg above).  Its clamp
    // drops a NaN of r2 (non-finite inputs), which is put back through sigma_f^2; r2 = +inf (a zero length scale between
    // distinct points) must stay exp(-inf) = 0 as in the reference, so the NaN is selected, not multiplied in
    double k = ((r2 != r2) ? r2 : kp.sigma * kp.sigma) * exp_neg(0.5 * r2);
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

// (keeps what depends on v inside the branch it is written in: hipcc otherwise speculates the cheap arithmetic of every
// class above the class test and all pairs pay for it)
__device__ __forceinline__ double pin_here(double v)
{
    asm volatile("" : "+v"(v));
    return v;
}

template <int D>
__device__ __forceinline__ double m52_pair(const KParams &kp, const double *xi, const double *xj,
                                           const int *ni, const int *nj)
{
    double r2 = 0.0, disp[D];
    int ii = -1, ij = -1;
#pragma unroll
    for (int d = D - 1; d >= 0; d--) {                  // descending so the FIRST index with n == 1 wins (matern.c:32-39)
        disp[d] = xi[d] - xj[d];
        r2 = fma(disp[d] * disp[d], kp.inv_var[d], r2);
        if (ni[d] == 1) ii = d;
        if (nj[d] == 1) ij = d;
    }
    // r = sqrt(r2) and 1/r2 from ONE v_rsq_f64 + a Newton step (1/sqrt(r2) to 1.5 * 2^-52): the correctly rounded
    // sqrt and the IEEE division of the (e_a, e_b) class were ~50 of this kernel's instructions, and the Matern-5/2
    // builder is bound by its arithmetic, not by the stores.
    // Coincident points (r2 == 0: matern.c:83-84, :100-101, :123-127).  Round 2 ran the arithmetic on r2 = 1 and selected
    // the reference's constants at the end (a compare and four 32-bit selects per pair); round 4 clamps r2 at 2^-600
    // instead -- ONE v_max -- because every class then reaches its limit by itself: r = sqrt5 2^-300 vanishes against 1 and
    // exp(-r) == 1, so the value class gives exactly 1; a first-derivative class multiplies by the zero displacement; the
    // (e_a, e_a) class forms g (2^-600 / var_a) (1 / 2^-600) = (5/3) / var_a (2^600 is exact and finite), the (e_a, e_b)
    // class 0.  v_max drops a NaN, so a non-finite r2 is multiplied back in through sigma_f^2 (one FMA).
    const double r2s = fmax(r2, 0x1p-600);
    const double y0 = __builtin_amdgcn_rsq(r2s);
    const double yr = fma(0.5 * y0, fma(-(r2s * y0), y0, 1.0), y0);
    const double s5r = GPT_SQRT5 * (r2s * yr);
    const double e = exp_neg(s5r);
    double v;
    // The class of a pair -- which side carries a derivative -- is the same for all 64 columns of a wave nearly
    // everywhere (the rows are wave-uniform, the columns of one class are contiguous in any sensible ordering), so each
    // class has its own branch and a wave executes one of them; the select-everything form of round 1 computed all four
    // formulas for every pair (100 vector instructions per pair, 61 of them double precision).
    if (__builtin_amdgcn_ballot_w64(ij >= 0) == 0) {
        if (ii < 0) {
            v = fma(GPT_FIVE_THIRDS, r2s, 1.0 + s5r) * e;
        } else {
            const double ep = pin_here(e);
            double di = 0.0, ivi = 0.0;
#pragma unroll
            for (int d = 0; d < D; d++) {
                const double dd = pin_here(disp[d]);
                if (d == ii) { di = dd; ivi = kp.inv_var[d]; }
            }
            v = -(GPT_FIVE_THIRDS * (1.0 + s5r) * ep) * (di * ivi);
        }
    } else {
        const double ep = pin_here(e);
        double di = 0.0, dj = 0.0, ivi = 0.0, ivj = 0.0;   // displacement / inverse variance at the derivative dims
#pragma unroll
        for (int d = 0; d < D; d++) {
            const double dd = pin_here(disp[d]);
            if (d == ii) { di = dd; ivi = kp.inv_var[d]; }
            if (d == ij) { dj = dd; ivj = kp.inv_var[d]; }
        }
        const double g = GPT_FIVE_THIRDS * (1.0 + s5r) * ep;              // -k'(r) / r
        if (ii < 0) {
            v = (ij >= 0) ? g * (dj * ivj)                                // arguments swapped => sign flips (matern.c:182-184)
                          : fma(GPT_FIVE_THIRDS, r2s, 1.0 + s5r) * ep;
        } else {
            double d2r = (di * ivi) * (dj * ivj);
            const double cross = d2r;
            if (ii == ij) d2r -= r2s * ivi;
            const double d2k = GPT_FIVE_THIRDS * (5.0 * r2s - s5r - 1.0) * ep;
            const double both = (-g * d2r - d2k * cross) * (yr * yr);     // term1 + term2 over r^2, matern.c:143-146
            v = (ij >= 0) ? both : -g * (di * ivi);
        }
    }
    return fma(r2, 0.0, kp.sigma * kp.sigma) * v;            // (sigma_f^2, with a NaN of r2 multiplied back in)
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

// ---- MaternKernel, general order nu (ref: gptools/kernel/matern.py:251-465 on ChainRuleKernel.__call__, core.py:691-816) ----
//   k = sigma^2 f(y),  f(y) = 2^(1-nu)/Gamma(nu) y^(nu/2) K_nu(sqrt y),  y = 2 nu sum_d tau_d^2 / l_d^2.
// Derivatives: Faa di Bruno over the partitions of the derivative multiset as for the rational-quadratic kernel (y is
// quadratic in tau: singletons y1_d = 4 nu tau_d / l_d^2, equal-index pairs y2_d = 4 nu / l_d^2, matern.py:392-398), with
//   f^(m)(y) = 2^(1-nu)/Gamma(nu) (-1/2)^m y^((nu-m)/2) K_(nu-m)(sqrt y)
// -- the closed form of what the reference sums through the general Leibniz rule and Bell polynomials of kvp
// (utils.py:1397-1476).  K of real order: Temme's method (J. Comput. Phys. 19 (1975) 324) for K_mu, K_mu+1 with
// |mu| <= 1/2 -- power series for x <= 2, Steed's continued fraction above -- then the upward recurrence in the order; the
// Gamma-function constants of the series come from the host (make_kparams).  Near the origin the reference's own choices
// are reproduced: for 0 < y <= 5e-4 a one-term power series (utils.py:1493-1516), for integer nu the mean of nu -+ 0.001
// there (utils.py:1480-1484, :1498-1502), at y == 0 the finite limit / +-inf (utils.py:1486-1492) and the term-by-term
// masking of matern.py:437-446 (0 if 2 (nu - |pi|) + n1 > 0, else NaN).
__device__ __forceinline__ void bessk_temme(const KParams &kp, double mu, double x, double &kmu, double &kmu1, bool negmu)
{
    const double PI = 3.14159265358979323846, EPS = 1.0e-16;
    const double mu2 = mu * mu;
    // NaN in, NaN out at once (scipy.special.kv does the same): with a NaN the convergence tests below are never true and
    // every affected pair would walk the full 500 / 10000 iterations -- a K build with one NaN coordinate went from
    // microseconds to minutes.  x = +inf: K_mu(inf) = 0.
    if (!(x == x)) {
        kmu = kmu1 = x;
        return;
    }
    if (x > 1.0e300) {
        kmu = kmu1 = 0.0;
        return;
    }
    if (x <= 2.0) {
        // negmu: the same constants serve -mu (1/Gamma(1 +- mu) swap, gam1 and gam2 are even in mu)
        const double gampl = negmu ? kp.m_gammi : kp.m_gampl, gammi = negmu ? kp.m_gampl : kp.m_gammi;
        const double lg = log(2.0 / x), sg = mu * lg;
        const double fact2 = (fabs(sg) < 1.0e-8) ? 1.0 : sinh(sg) / sg;
        const double pimu = PI * mu;
        const double fact = (fabs(pimu) < 1.0e-8) ? 1.0 : pimu / sin(pimu);
        double ff = fact * (kp.m_gam1 * cosh(sg) + kp.m_gam2 * fact2 * lg);
        const double e = exp(sg);
        double p = 0.5 * e / gampl, q = 0.5 / (e * gammi);
        double c = 1.0, sum = ff, sum1 = p;
        const double d = 0.25 * x * x;
        for (int i = 1; i <= 500; i++) {
            const double fi = (double)i;
            ff = (fi * ff + p + q) / (fi * fi - mu2);
            c *= d / fi;
            p /= fi - mu;
            q /= fi + mu;
            const double del = c * ff;
            sum += del;
            sum1 += c * (p - fi * ff);
            if (fabs(del) < fabs(sum) * EPS) break;
        }
        kmu = sum;
        kmu1 = sum1 * 2.0 / x;
    } else {
        double a = mu2 - 0.25, b = 2.0 * (x + 1.0), Dd = 1.0 / b, f = Dd, delta = Dd;
        double prev = 0.0, cur = 1.0, C = -a, Q = -a, S = 1.0 + Q * delta;
        for (int i = 2; i <= 10000; i++) {
            a -= 2.0 * (double)(i - 1);
            b += 2.0;
            Dd = 1.0 / (b + a * Dd);
            delta *= b * Dd - 1.0;
            f += delta;
            const double qn = (prev - (b - 2.0) * cur) / a;
            prev = cur;
            cur = qn;
            C *= -a / (double)i;
            Q += C * qn;
            S += Q * delta;
            if (fabs(Q * delta) < fabs(S) * EPS) break;
        }
        kmu = sqrt(PI / (2.0 * x)) * exp(-x) / S;
        kmu1 = kmu * (0.5 + mu + x + (mu2 - 0.25) * f) / x;
    }
}

__device__ __forceinline__ double matern_poch(double a, int n)
{
    double p = 1.0;
    for (int k = 0; k < n; k++) p *= a + (double)k;
    return p;
}

// utils.py:1486-1516 for one non-integer order nus (with its Gamma(nus), Gamma(-nus)): y == 0 limit or the one-term series
__device__ __forceinline__ double matern_small(double nus, double g, double gm, double y, int n, bool at_zero)
{
    const double t1 = g / (exp2(1.0 - nus + 2.0 * (double)n) * matern_poch(1.0 - nus, n));
    if (at_zero) {
        if ((double)n > nus) return gm * matern_poch(1.0 + nus - (double)n, n) * (double)INFINITY;
        return t1;
    }
    return t1 + gm * matern_poch(1.0 + nus - (double)n, n) * exp((nus - (double)n) * log(y)) / exp2(1.0 + nus);
}

// F[m] = f^(m)(y) for m = 1 .. deg (y >= 0)
__device__ __forceinline__ void matern_F(const KParams &kp, double y, int deg, double (&F)[GPT_RQ_MAXORD + 1])
{
    const double nu = kp.alpha;
    if (y > 5.0e-4) {
        const double z = sqrt(y), lz = log(z);
        // orders nu - m = mu + (nint - m): non-negative offsets by the upward recurrence from (K_mu, K_mu+1); negative
        // offsets are orders (-mu) + j of the mirrored base
        double ka, kb;
        bessk_temme(kp, kp.m_mu, z, ka, kb, false);
        const int jtop = kp.m_nint - 1;                       // largest offset needed (m = 1)
        double kup[GPT_RQ_MAXORD + 1];                        // kup[m] = K_(nu-m) where nint - m >= 0
        {
            // walk up to offset jtop, remembering the offsets nint - m >= 0
            double k0 = ka, k1 = kb;                          // offsets j, j+1 with j = 0
            for (int m = 1; m <= deg; m++) kup[m] = 0.0;
            for (int j = 0; j <= jtop || j == 0; j++) {
                const int m = kp.m_nint - j;
                if (m >= 1 && m <= deg) kup[m] = k0;
                const double kn = k0 + 2.0 * (kp.m_mu + (double)(j + 1)) / z * k1;
                k0 = k1;
                k1 = kn;
            }
        }
        bool need_neg = false;
        for (int m = 1; m <= deg; m++) need_neg = need_neg || (kp.m_nint - m < 0);
        double kdn[GPT_RQ_MAXORD + 1];
        if (need_neg) {
            double k0, k1;
            bessk_temme(kp, -kp.m_mu, z, k0, k1, true);       // K_(-mu) (= K_mu), K_(1-mu)
            for (int m = 1; m <= deg; m++) kdn[m] = 0.0;
            for (int j = 0; j <= GPT_RQ_MAXORD; j++) {        // offset -j: order j - mu
                const int m = kp.m_nint + j;
                if (j >= 1 && m >= 1 && m <= deg) kdn[m] = k0;
                const double kn = k0 + 2.0 * (-kp.m_mu + (double)(j + 1)) / z * k1;
                k0 = k1;
                k1 = kn;
            }
        }
        double sgn = 1.0;
        for (int m = 1; m <= deg; m++) {
            sgn *= -0.5;
            const double kv = (kp.m_nint - m >= 0) ? kup[m] : kdn[m];
            F[m] = kp.m_cnu * sgn * exp((nu - (double)m) * lz) * kv;
        }
        return;
    }
    const bool z0 = (y == 0.0);
    for (int m = 1; m <= deg; m++) {
        double v = matern_small(kp.m_nus[0], kp.m_g[0], kp.m_gm[0], y, m, z0);
        if (kp.m_isint) v = 0.5 * (v + matern_small(kp.m_nus[1], kp.m_g[1], kp.m_gm[1], y, m, z0));
        F[m] = kp.m_cnu * v;
    }
}

template <int D>
__device__ __forceinline__ double matern_pair(const KParams &kp, const double *xi, const double *xj,
                                              const int *ni, const int *nj)
{
    const double nu = kp.alpha, s2 = kp.sigma * kp.sigma;
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
    const double y = 2.0 * nu * r2;
    if (ntot == 0) {                                          // matern.py:322-325
        if (r2 == 0.0) return s2;
        const double z = sqrt(y);
        double ka, kb;
        bessk_temme(kp, kp.m_mu, z, ka, kb, false);
        for (int j = 0; j < kp.m_nint; j++) {
            const double kn = ka + 2.0 * (kp.m_mu + (double)(j + 1)) / z * kb;
            ka = kb;
            kb = kn;
        }
        return s2 * kp.m_cnu * exp(nu * log(z)) * ka;
    }
    double F[GPT_RQ_MAXORD + 1];
    if (y == 0.0) {
        // every tau_d is zero: partitions with singletons count 0 or NaN (matern.py:437-446), the all-pairs one survives
        int jmax = 0;
        bool all_even = true;
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int n = ni[d] + nj[d];
            jmax += n / 2;
            all_even = all_even && ((n & 1) == 0);
        }
        for (int J = 0; J <= jmax; J++) {
            const int n1 = ntot - 2 * J, mblk = ntot - J;
            if (n1 > 0 && 2.0 * (nu - (double)mblk) + (double)n1 <= 0.0) return (double)NAN;
        }
        if (!all_even) return s2 * ((njtot & 1) ? -0.0 : 0.0);
        double fac = 1.0;
        int mm = 0;
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int n = ni[d] + nj[d], j = n / 2;
            double coef = 1.0, y2pow = 1.0;
            const double y2 = 4.0 * nu * kp.inv_var[d];
            for (int q = 1; q <= j; q++) {
                coef = coef * (double)((n - 2 * q + 2) * (n - 2 * q + 1)) / (2.0 * (double)q);
                y2pow *= y2;
            }
            fac *= coef * y2pow;
            mm += j;
        }
        matern_F(kp, 0.0, mm, F);
        const double v = F[mm] * fac;
        return s2 * ((njtot & 1) ? -v : v);
    }
    // c[m]: coefficient of f^(m)(y) (host code rejects ntot > GPT_RQ_MAXORD)
    double c[GPT_RQ_MAXORD + 1];
#pragma unroll
    for (int m = 0; m <= GPT_RQ_MAXORD; m++) c[m] = (m == 0) ? 1.0 : 0.0;
    int deg = 0;
#pragma unroll
    for (int d = 0; d < D; d++) {
        const int n = ni[d] + nj[d];
        if (n == 0) continue;
        const double y2 = 4.0 * nu * kp.inv_var[d];
        const double y1 = (xi[d] - xj[d]) * y2;
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
    if (deg > GPT_RQ_MAXORD) deg = GPT_RQ_MAXORD;
    matern_F(kp, y, deg, F);
    double v = 0.0;
    for (int m = 1; m <= deg; m++)
        if (c[m] != 0.0) v = fma(c[m], F[m], v);
    return s2 * ((njtot & 1) ? -v : v);
}

// The value class alone -- no derivative order on either side, no hyper-parameter derivative: what the builder's row loop
// runs for a tile whose 32 rows and whose wave's 64 columns are all plain (kbuild_kernel.hpp), with none of the class logic.
// Same arithmetic, in the same order, as the value branch of se_pair / m52_pair: the same bits.
template <int KID, int D>
__device__ __forceinline__ double plain_pair(const KParams &kp, const double *xi, const double *xj)
{
    double r2 = 0.0;
    if constexpr (KID == GPT_KERNEL_SE) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const double tau = xi[d] - xj[d];
            double t = tau * kp.inv_l[d];
            if (kp.zero_l) t = (tau == 0.0) ? 0.0 : t;
            r2 = fma(t, t, r2);
        }
        return ((r2 != r2) ? r2 : kp.sigma * kp.sigma) * exp_neg(0.5 * r2);
    } else {
#pragma unroll
        for (int d = D - 1; d >= 0; d--) {
            const double disp = xi[d] - xj[d];
            r2 = fma(disp * disp, kp.inv_var[d], r2);
        }
        const double r2s = fmax(r2, 0x1p-600);
        const double y0 = __builtin_amdgcn_rsq(r2s);
        const double yr = fma(0.5 * y0, fma(-(r2s * y0), y0, 1.0), y0);
        const double s5r = GPT_SQRT5 * (r2s * yr);
        const double e = exp_neg(s5r);
        return fma(r2, 0.0, kp.sigma * kp.sigma) * (fma(GPT_FIVE_THIRDS, r2s, 1.0 + s5r) * e);
    }
}

template <int KID, int D>
__device__ __forceinline__ double any_pair(const KParams &kp, const double *xi, const double *xj,
                                           const int *ni, const int *nj)
{
    if (KID == GPT_KERNEL_SE) return se_pair<D>(kp, xi, xj, ni, nj);
    if (KID == GPT_KERNEL_M52) return m52_pair<D>(kp, xi, xj, ni, nj);
    if (KID == GPT_KERNEL_DIAGNOISE) return noise_pair<D>(kp, xi, xj, ni, nj);
    if (KID == GPT_KERNEL_RQ) return rq_pair<D>(kp, xi, xj, ni, nj);
    if (KID == GPT_KERNEL_MATERN) return matern_pair<D>(kp, xi, xj, ni, nj);
    return 0.0;
}

// ---- product of two native kernels (ref: gptools/kernel/core.py:587-671) --------------------------------------------------
// The reference walks the power set of the derivative multiset of a pair and multiplies k1 with the subset's orders by k2
// with the complement's; equal subsets recur, so grouped by how many of the r_s derivatives of slot s (the D orders of ni, then
// the D of nj) go to k1 that is the general Leibniz rule  sum_a prod_s C(r_s, a_s) k1^(a) k2^(r - a).  Factors by run-time id.
template <int D>
__device__ __forceinline__ double factor_pair(const KParams &kp, const double *xi, const double *xj, const int *ni, const int *nj)
{
    switch (kp.kernel_id) {
    case GPT_KERNEL_SE: return se_pair<D>(kp, xi, xj, ni, nj);
    case GPT_KERNEL_M52: return m52_pair<D>(kp, xi, xj, ni, nj);
    case GPT_KERNEL_RQ: return rq_pair<D>(kp, xi, xj, ni, nj);
    case GPT_KERNEL_MATERN: return matern_pair<D>(kp, xi, xj, ni, nj);
    default: return 0.0;
    }
}

template <int D>
__device__ double prod_pair(const KParams &k1, const KParams &k2, const double *xi, const double *xj, const int *ni, const int *nj)
{
    int r[2 * D], a[2 * D];
#pragma unroll
    for (int s = 0; s < 2 * D; s++) {
        r[s] = (s < D) ? ni[s] : nj[s - D];
        a[s] = 0;
    }
    double sum = 0.0;
    for (;;) {
        int n1i[D], n1j[D], n2i[D], n2j[D];
        double w = 1.0;
#pragma unroll
        for (int s = 0; s < 2 * D; s++) {
            // C(r, a), r <= GPT_RQ_MAXORD
            double c = 1.0;
            for (int q = 0; q < a[s]; q++) c = c * (double)(r[s] - q) / (double)(q + 1);
            w *= c;
            if (s < D) {
                n1i[s] = a[s];
                n2i[s] = r[s] - a[s];
            } else {
                n1j[s - D] = a[s];
                n2j[s - D] = r[s] - a[s];
            }
        }
        sum += w * (factor_pair<D>(k1, xi, xj, n1i, n1j) * factor_pair<D>(k2, xi, xj, n2i, n2j));
        int s = 0;
        while (s < 2 * D) {
            if (a[s] < r[s]) {
                a[s]++;
                break;
            }
            a[s] = 0;
            s++;
        }
        if (s == 2 * D) break;
    }
    return sum;
}
