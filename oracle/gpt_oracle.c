/*
 * gpt_oracle.c -- CPU restatement of the gptools covariance-build + Cholesky LML hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: it is linked/loaded only by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product
 * path (gptools_amd/), which must fail loudly when the HIP library is missing.
 *
 * Plain C restatement, written from the reference's algorithm (citations are into
 * /root/reference):
 *   - kernel/core.py:384-421            Kernel._compute_r2l2 (0/0 -> 0 rule)
 *   - kernel/squared_exponential.py:82-174   SquaredExponentialKernel.__call__
 *   - kernel/matern.py:512-555, kernel/_matern.pyx:14-32, kernel/src/matern.c:61-186
 *                                       Matern52Kernel.__call__ / matern52()
 *   - kernel/noise.py:76-110, :123-152  DiagonalNoiseKernel / ZeroKernel
 *   - kernel/matern.py:251-465, utils.py:1369-1527   MaternKernel (general nu): ChainRuleKernel with
 *                                       f(y) = 2^(1-nu)/Gamma(nu) y^(nu/2) K_nu(sqrt y), y = 2 nu sum tau^2/l^2
 *   - gaussian_process.py:1535-1605     compute_Kij (row-major cartesian product)
 *   - gaussian_process.py:1418-1469     compute_K_L_alpha_ll
 *   - gaussian_process.py:965-1006      predict (non-MCMC branch)
 * Third-party arithmetic that is not under /root/reference and is restated here:
 *   - scipy.special.eval_hermite (scipy 1.15.3, scipy/special/orthogonal_eval.pxd:
 *     eval_hermite(n,x) = eval_hermitenorm(n, sqrt(2) x) * 2**(n/2), eval_hermitenorm by the
 *     downward three-term recurrence).  tests/test_oracle.py checks this restatement
 *     bit-for-bit against scipy on the GPU box and here.
 *   - scipy.linalg.cholesky / cho_solve / solve_triangular (LAPACK dpotrf/dpotrs/dtrtrs via
 *     scipy-openblas 0.3.28): restated as the textbook column-Crout Cholesky and
 *     forward/backward substitution; tests check it against scipy.linalg at rounding level.
 *   - scipy.special.kv / kvp / gamma (scipy 1.15.3 -> AMOS zbesk, cephes Gamma), reached by MaternKernel through
 *     utils.yn2Kn2Der / Kn2Der: restated as Temme's method for K_mu, K_mu+1 with |mu| <= 1/2 (N. M. Temme, J.
 *     Comput. Phys. 19 (1975) 324: power series for x <= 2, Steed's continued fraction CF2 above) + the upward
 *     recurrence in the order, and libm tgamma; checked against scipy.special.kv in tests/test_oracle_golden.py.
 * Pinning: the golden fixtures tests/golden/g1..g7 (generated from the imported reference by
 * tests/golden/gen_golden.py), the demo known answer (demo/demo.py:190-192) and, for
 * Matern-5/2, the reference's own matern.c compiled into oracle/_ref/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_KERNEL_SE 0
#define ORC_KERNEL_M52 1
#define ORC_KERNEL_DIAGNOISE 2
#define ORC_KERNEL_ZERO 3
#define ORC_KERNEL_RQ 4
#define ORC_KERNEL_MATERN 5     /* general-nu MaternKernel (kernel/matern.py:251-465) */
#define ORC_RQ_MAXORD 16        /* combined derivative order of a pair this restatement accepts */

#define ORC_OK 0
#define ORC_EVALUE 1          /* Python side raises ValueError      (matern.py:545-546) */
#define ORC_ENOTIMPL 2        /* Python side raises NotImplementedError (matern.py:543-544) */
#define ORC_EARG 3

/* ---- scipy.special.eval_hermitenorm / eval_hermite (integer order) -------------------- */
double orc_eval_hermitenorm(long n, double x)
{
    long k;
    double y1, y2, y3;
    if (isnan(x)) return x;
    if (n < 0) return NAN;
    if (n == 0) return 1.0;
    if (n == 1) return x;
    y3 = 0.0;
    y2 = 1.0;
    for (k = n; k > 1; k--) {
        y1 = x * y2 - k * y3;
        y3 = y2;
        y2 = y1;
    }
    return x * y2 - y3;
}

double orc_eval_hermite(long n, double x)
{
    if (n < 0) return NAN;
    return orc_eval_hermitenorm(n, sqrt(2.0) * x) * pow(2.0, n / 2.0);
}

/* ---- SquaredExponentialKernel.__call__ for one pair ----------------------------------- */
/* only_first_order is the call-wide flag of squared_exponential.py:110-113. */
static double se_pair(const double *params, int D, const double *xi, const double *xj,
                      const int32_t *ni, const int32_t *nj, int hyper_deriv, int only_first_order)
{
    const double *l = params + 1;        /* last num_dim params (core.py:414) */
    const double sigma = params[0];
    double r2l2 = 0.0, k;
    int d;
    for (d = 0; d < D; d++) {
        double tau = xi[d] - xj[d];
        double tol = tau / l[d];
        if (tau == 0.0 && l[d] == 0.0) tol = 0.0;      /* core.py:416 */
        r2l2 += tol * tol;
    }
    k = sigma * sigma * exp(-r2l2 / 2.0);                /* squared_exponential.py:116 */
    if (!only_first_order) {
        long n_tot_j = 0;
        double prod = 1.0;
        for (d = 0; d < D; d++) n_tot_j += nj[d];
        for (d = 0; d < D; d++) {
            long nc = (long)ni[d] + (long)nj[d];
            double tau = xi[d] - xj[d];
            double s2l = sqrt(2.0) * l[d];
            double hf = pow(-1.0 / s2l, (double)nc) * orc_eval_hermite(nc, tau / s2l);   /* :127-132 */
            if (hyper_deriv > 0 && d == hyper_deriv - 1) {                                 /* :134-158 */
                double lh = params[hyper_deriv];
                double t = tau * tau / pow(lh, 3.0);
                if (nc > 0) {
                    t -= (double)nc / lh;
                    if (tau != 0.0) {
                        double xa = tau / (sqrt(2.0) * lh);
                        t -= sqrt(2.0) * (double)nc * tau / pow(lh, 2.0) *
                             orc_eval_hermite(nc - 1, xa) / orc_eval_hermite(nc, xa);
                    }
                }
                hf *= t;
            }
            prod *= hf;
        }
        k = pow(-1.0, (double)n_tot_j) * prod * k;       /* :125, :160 */
    }
    if (hyper_deriv < 0) return k;
    if (hyper_deriv == 0) return (sigma != 0.0) ? 2.0 * k / sigma : 0.0;   /* :164-166 */
    if (only_first_order) {                                                 /* :169-171 */
        double tau = xi[hyper_deriv - 1] - xj[hyper_deriv - 1];
        return tau * tau / pow(params[hyper_deriv], 3.0) * k;
    }
    return k;
}

/* ---- Matern-5/2, restating kernel/src/matern.c ---------------------------------------- */
static const double M_SQRT_5 = 2.2360679774997898;
static const double M_FIVE_THIRDS = 1.6666666666666667;

static double m52_r2(const double *X, const double *Y, const double *var, int d)
{   /* matern.c:61-71 */
    int i;
    double r2 = 0.0;
    for (i = 0; i < d; i++) {
        double disp = X[i] - Y[i];
        r2 = r2 + disp * disp / var[i];
    }
    return r2;
}

static double m52_value(const double *X, const double *Y, int d, const double *var)
{   /* matern.c:77-88 */
    double r2 = m52_r2(X, Y, var, d), s5r;
    if (r2 == 0) return 1;
    s5r = M_SQRT_5 * sqrt(r2);
    return (1.0 + s5r + M_FIVE_THIRDS * r2) * exp(-s5r);
}

static double m52_dX(const double *X, const double *Y, int n, int d, const double *var)
{   /* matern.c:95-106 */
    double r2 = m52_r2(X, Y, var, d), s5r;
    if (r2 == 0) return 0;
    s5r = M_SQRT_5 * sqrt(r2);
    return (-M_FIVE_THIRDS * (1 + s5r) * exp(-s5r)) * ((X[n] - Y[n]) / var[n]);
}

static double m52_dXdY(const double *X, const double *Y, int n, int m, int d, const double *var)
{   /* matern.c:112-147 */
    double r2 = m52_r2(X, Y, var, d), r, dr_dXn, dr_dYm, d2r, s5r, e, dk_over_r, d2k;
    if (r2 == 0) {
        if (n == m) return M_FIVE_THIRDS / var[n];
        return 0;
    }
    r = sqrt(r2);
    dr_dXn = (X[n] - Y[n]) / (r * var[n]);
    dr_dYm = -(X[m] - Y[m]) / (r * var[m]);
    d2r = (X[n] - Y[n]) * (X[m] - Y[m]) / (var[n] * var[m]);
    if (n == m) d2r -= r * r / var[n];
    s5r = M_SQRT_5 * r;
    e = exp(-s5r);
    dk_over_r = -M_FIVE_THIRDS * (1 + s5r) * e;
    d2k = M_FIVE_THIRDS * (5 * r2 - s5r - 1) * e;
    return dk_over_r * d2r / r2 + d2k * dr_dXn * dr_dYm;
}

static int first_one(const int32_t *n, int d)
{   /* matern.c:32-39 */
    int i;
    for (i = 0; i < d; i++)
        if (n[i] == 1) return i;
    return -1;
}

static double m52_pair(const double *xi, const double *xj, const int32_t *ni, const int32_t *nj,
                       int d, const double *var)
{   /* matern.c:165-186 (orders > 1 are rejected by the caller, matern.py:545-546) */
    int ii = first_one(ni, d), ij = first_one(nj, d);
    if (ii == -1 && ij == -1) return m52_value(xi, xj, d, var);
    if (ii > -1 && ij == -1) return m52_dX(xi, xj, ii, d, var);
    if (ii == -1 && ij > -1) return m52_dX(xj, xi, ij, d, var);
    return m52_dXdY(xi, xj, ii, ij, d, var);
}

/* ---- helpers --------------------------------------------------------------------------- */
/* ---- RationalQuadraticKernel (kernel/rational_quadratic.py:30-164) through ChainRuleKernel.__call__
 * (kernel/core.py:691-816).  params = [sigma_f, alpha, l_1 .. l_D];
 *     k = sigma_f^2 (-1)^{sum nj} d^n/dtau^n f(y),  f(y) = y^-alpha,  y = 1 + sum_d tau_d^2 / (2 alpha l_d^2).
 * The reference sums Faa di Bruno's formula over every set partition of the derivative multiset
 * (core.py:794-816): a partition with p blocks contributes f^(p)(y) = poch(1 - alpha - p, p) y^(-alpha-p)
 * (rational_quadratic.py:127-128, utils.py:1369-1395: the plain product) times, per block, dy/dtau_a =
 * tau_a / (alpha l_a^2) for a single index, 1 / (alpha l_a^2) for a pair of equal indices and 0 otherwise
 * (rational_quadratic.py:155-162).  Only partitions into singletons and equal-index pairs survive; those with j_d pairs
 * in dimension d number n_d! / (j_d! (n_d - 2 j_d)! 2^j_d) per dimension, so the same sum is, grouped by (j_1..j_D),
 *     sum_j  prod_d [ n_d!/(j_d!(n_d-2j_d)! 2^j_d) y2_d^j_d y1_d^(n_d-2j_d) ]  f^(sum_d (n_d - j_d))(y)
 * evaluated below as a product of per-dimension polynomials in a marker z (power = number of blocks). */
static double rq_pair(const double *params, int D, const double *xi, const double *xj, const int32_t *ni,
                      const int32_t *nj)
{
    const double sigma = params[0], alpha = params[1];
    const double *l = params + 2;
    double c[ORC_RQ_MAXORD + 1], pd[ORC_RQ_MAXORD + 1], cn[ORC_RQ_MAXORD + 1];
    double r2l2 = 0.0, y, v = 0.0;
    int d, deg = 0, ntot = 0, njtot = 0, m, a, b;
    for (d = 0; d < D; d++) {
        double tau = xi[d] - xj[d];
        double t = (tau == 0.0) ? 0.0 : tau / l[d];                 /* core.py:416: 0/0 counts as 0 */
        r2l2 += t * t;
        ntot += ni[d] + nj[d];
        njtot += nj[d];
    }
    y = 1.0 + 1.0 / (2.0 * alpha) * r2l2;                            /* rational_quadratic.py:104-105 */
    if (ntot == 0) return sigma * sigma * pow(y, -alpha);            /* rational_quadratic.py:82-83 */
    if (ntot > ORC_RQ_MAXORD) return NAN;
    c[0] = 1.0;
    for (d = 0; d < D; d++) {
        const int n = ni[d] + nj[d];
        double tau, y1, y2, coef;
        int j;
        if (n == 0) continue;
        tau = xi[d] - xj[d];
        y1 = 1.0 / alpha * tau / (l[d] * l[d]);                      /* rational_quadratic.py:158 */
        y2 = 1.0 / (alpha * (l[d] * l[d]));                          /* rational_quadratic.py:160 */
        for (m = 0; m <= n; m++) pd[m] = 0.0;
        for (j = 0; 2 * j <= n; j++) {
            /* n! / (j! (n-2j)! 2^j) by the recurrence c_{j+1} = c_j (n-2j)(n-2j-1) / (2 (j+1)) */
            if (j == 0) coef = 1.0;
            else coef = coef * (double)((n - 2 * j + 2) * (n - 2 * j + 1)) / (2.0 * (double)j);
            pd[n - j] = coef * pow(y2, (double)j) * pow(y1, (double)(n - 2 * j));
        }
        for (m = 0; m <= deg + n; m++) cn[m] = 0.0;
        for (a = 0; a <= deg; a++)
            for (b = 0; b <= n; b++) cn[a + b] += c[a] * pd[b];
        deg += n;
        for (m = 0; m <= deg; m++) c[m] = cn[m];
    }
    for (m = 1; m <= deg; m++) {
        double poch = 1.0;
        int q;
        if (c[m] == 0.0) continue;
        for (q = 0; q < m; q++) poch *= (1.0 - alpha - (double)m) + (double)q;      /* utils.py:1394-1395 */
        v += c[m] * poch * pow(y, -alpha - (double)m);
    }
    return sigma * sigma * ((njtot & 1) ? -v : v);                   /* core.py:746-749 */
}

/* ---- modified Bessel function of the second kind, real order (scipy.special.kv) ------------------------ */
/* K_mu(x) and K_{mu+1}(x) for |mu| <= 1/2, x > 0.  Temme's method. */
static void bessk_temme(double mu, double x, double *kmu, double *kmu1)
{
    const double PI = 3.14159265358979323846, EPS = 1.0e-16;
    const double mu2 = mu * mu;
    int i;
    if (x <= 2.0) {
        /* K_mu = sum c_k f_k, K_mu+1 = (2/x) sum c_k (p_k - k f_k), c_k = (x^2/4)^k / k! */
        const double gampl = 1.0 / tgamma(1.0 + mu), gammi = 1.0 / tgamma(1.0 - mu);
        const double gam2 = 0.5 * (gammi + gampl);
        /* (1/Gamma(1-mu) - 1/Gamma(1+mu)) / (2 mu); near mu = 0 from 1/Gamma(1+t) = 1 + g t + a2 t^2 + a3 t^3 + ... */
        const double gam1 = (fabs(mu) < 1.0e-3)
            ? -(0.57721566490153286 + mu2 * (-0.042002635034095236 + mu2 * (-0.0072189432466630995)))
            : (gammi - gampl) / (2.0 * mu);
        const double lg = log(2.0 / x), sg = mu * lg;
        const double fact2 = (fabs(sg) < 1.0e-8) ? 1.0 : sinh(sg) / sg;
        const double pimu = PI * mu;
        const double fact = (fabs(pimu) < 1.0e-8) ? 1.0 : pimu / sin(pimu);
        double ff = fact * (gam1 * cosh(sg) + gam2 * fact2 * lg);
        const double e = exp(sg);
        double p = 0.5 * e / gampl, q = 0.5 / (e * gammi);
        double c = 1.0, sum = ff, sum1 = p;
        const double d = 0.25 * x * x;
        for (i = 1; i <= 500; i++) {
            double del;
            ff = ((double)i * ff + p + q) / ((double)i * (double)i - mu2);
            c *= d / (double)i;
            p /= (double)i - mu;
            q /= (double)i + mu;
            del = c * ff;
            sum += del;
            sum1 += c * (p - (double)i * ff);
            if (fabs(del) < fabs(sum) * EPS) break;
        }
        *kmu = sum;
        *kmu1 = sum1 * 2.0 / x;
    } else {
        /* Steed's algorithm for the continued fraction CF2 (Thompson & Barnett 1986; Temme 1975) */
        double a = mu2 - 0.25, b = 2.0 * (x + 1.0), D = 1.0 / b, f = D, delta = D;
        double prev = 0.0, cur = 1.0, C = -a, Q = -a, S = 1.0 + Q * delta;
        for (i = 2; i <= 10000; i++) {
            double qn;
            a -= 2.0 * (double)(i - 1);
            b += 2.0;
            D = 1.0 / (b + a * D);
            delta *= b * D - 1.0;
            f += delta;
            qn = (prev - (b - 2.0) * cur) / a;
            prev = cur;
            cur = qn;
            C *= -a / (double)i;
            Q += C * qn;
            S += Q * delta;
            if (fabs(Q * delta) < fabs(S) * EPS) break;
        }
        *kmu = sqrt(PI / (2.0 * x)) * exp(-x) / S;
        *kmu1 = *kmu * (0.5 + mu + x + (mu2 - 0.25) * f) / x;
    }
}

/* K_v(x), any real order v (K_-v = K_v), x > 0 */
double orc_bessel_k(double v, double x)
{
    double mu, k0, k1, kn;
    int n, i;
    v = fabs(v);
    n = (int)floor(v + 0.5);
    mu = v - (double)n;                         /* |mu| <= 1/2 */
    bessk_temme(mu, x, &k0, &k1);
    for (i = 1; i <= n; i++) {                  /* K_{w+1} = K_{w-1} + (2 w / x) K_w, upward: stable */
        kn = k0 + 2.0 * (mu + (double)i) / x * k1;
        k0 = k1;
        k1 = kn;
    }
    return k0;
}

static double poch_prod(double a, int n)        /* utils.py:1369-1395 fixed_poch: (a)_n as a plain product */
{
    double p = 1.0;
    int k;
    for (k = 0; k < n; k++) p *= a + (double)k;
    return p;
}

/* utils.py:1429-1518 yn2Kn2Der(nu, y, n) for n >= 1: the n-th derivative of y^(nu/2) K_nu(sqrt y) -- including the
 * reference's choices near the origin: a ONE-term power series for 0 < y <= 5e-4 (:1493-1516), the mean of nu -+ 0.001
 * for integer nu there and at y == 0 (:1480-1484, :1498-1502), +-inf / the finite limit at y == 0 (:1486-1492).
 * For y > 5e-4 the reference sums the general Leibniz rule over derivatives of K_nu(sqrt y) (Bell polynomials of kvp,
 * :1397-1427, :1472-1476); that sum IS (d/dy)^n [y^(nu/2) K_nu(sqrt y)] = (-1/2)^n y^((nu-n)/2) K_(nu-n)(sqrt y), which
 * is what is evaluated here (checked against fixtures generated by the reference itself). */
static double yn2kn2der_nonint_small(double nu, double y, int n, int at_zero)
{
    if (at_zero) {
        if ((double)n > nu) return tgamma(-nu) * poch_prod(1.0 + nu - (double)n, n) * INFINITY;          /* :1487 */
        return tgamma(nu) / (pow(2.0, 1.0 - nu + 2.0 * (double)n) * poch_prod(1.0 - nu, n));             /* :1489-1492 */
    }
    /* :1505-1516 with nterms = 1: k = n in the first sum, k = 0 in the second */
    return tgamma(nu) * poch_prod(1.0, n) / (pow(2.0, 1.0 - nu + 2.0 * (double)n) * poch_prod(1.0 - nu, n) * tgamma((double)n + 1.0))
         + tgamma(-nu) * poch_prod(1.0 + nu - (double)n, n) * pow(y, nu - (double)n) / pow(2.0, 1.0 + nu);
}

static double yn2kn2der(double nu, double y, int n)
{
    const double tol = 5.0e-4, nu_step = 0.001;
    if (y > tol) {
        const double z = sqrt(y);
        double v = pow(y, 0.5 * (nu - (double)n)) * orc_bessel_k(nu - (double)n, z);
        int k;
        for (k = 0; k < n; k++) v *= -0.5;
        return v;
    }
    if (nu == floor(nu))
        return 0.5 * (yn2kn2der_nonint_small(nu - nu_step, y, n, y == 0.0) + yn2kn2der_nonint_small(nu + nu_step, y, n, y == 0.0));
    return yn2kn2der_nonint_small(nu, y, n, y == 0.0);
}

/* MaternKernel.__call__ (kernel/matern.py:251-465 on ChainRuleKernel.__call__, core.py:691-816).  params =
 * [sigma_f, nu, l_1 .. l_D].  As for the rational-quadratic kernel only partitions into singletons (factor
 * y1_d = 4 nu tau_d / l_d^2, matern.py:394) and equal-index pairs (y2_d = 4 nu / l_d^2, :396) contribute; grouped by the
 * number of pairs per dimension.  At y == 0 the reference masks term by term (matern.py:437-446): a partition with n1 > 0
 * singletons and |pi| blocks counts 0 if 2 (nu - |pi|) + n1 > 0 and NaN otherwise. */
#define ORC_MATERN_MAXORD 16
static double matern_pair(const double *params, int D, const double *xi, const double *xj, const int32_t *ni,
                          const int32_t *nj)
{
    const double sigma = params[0], nu = params[1];
    const double *l = params + 2;
    const double cnu = pow(2.0, 1.0 - nu) / tgamma(nu);
    double c[ORC_MATERN_MAXORD + 1], pd[ORC_MATERN_MAXORD + 1], cn[ORC_MATERN_MAXORD + 1];
    double r2l2 = 0.0, y, v = 0.0;
    int d, deg = 0, ntot = 0, njtot = 0, m, a, b;
    for (d = 0; d < D; d++) {
        double tau = xi[d] - xj[d];
        double t = (tau == 0.0) ? 0.0 : tau / l[d];                 /* core.py:416 */
        r2l2 += t * t;
        ntot += ni[d] + nj[d];
        njtot += nj[d];
    }
    y = 2.0 * nu * r2l2;                                             /* matern.py:347-348 */
    if (ntot == 0) {                                                 /* matern.py:322-325 */
        if (r2l2 == 0.0) return sigma * sigma;
        return sigma * sigma * cnu * pow(y, 0.5 * nu) * orc_bessel_k(nu, sqrt(y));
    }
    if (ntot > ORC_MATERN_MAXORD) return NAN;
    if (y == 0.0) {
        /* every tau_d is zero: terms with singletons vanish or are NaN (see above); the all-pairs term survives */
        int all_even = 1, mm = 0, bad = 0;
        double fac = 1.0;
        /* does any (j_1..j_D) with n1 > 0 have 2 (nu - m) + n1 <= 0?  n1 = ntot - 2 J, m = ntot - J with J = sum j_d
         * pairs: the smallest value of 2 nu - 2 m + n1 = 2 nu - ntot over those is reached at every J alike */
        for (d = 0; d < D; d++) if ((ni[d] + nj[d]) & 1) all_even = 0;
        {
            int Jmax = 0, J;
            for (d = 0; d < D; d++) Jmax += (ni[d] + nj[d]) / 2;
            for (J = 0; J <= Jmax; J++) {
                const int n1 = ntot - 2 * J, mblk = ntot - J;
                if (n1 > 0 && 2.0 * (nu - (double)mblk) + (double)n1 <= 0.0) bad = 1;
            }
        }
        if (bad) return NAN;
        if (!all_even) return sigma * sigma * (((njtot & 1) ? -1.0 : 1.0) * 0.0);
        for (d = 0; d < D; d++) {
            const int n = ni[d] + nj[d], j = n / 2;
            double coef = 1.0;
            int q;
            for (q = 1; q <= j; q++) coef = coef * (double)((n - 2 * q + 2) * (n - 2 * q + 1)) / (2.0 * (double)q);
            fac *= coef * pow(4.0 * nu / (l[d] * l[d]), (double)j);
            mm += j;
        }
        v = cnu * yn2kn2der(nu, 0.0, mm) * fac;
        return sigma * sigma * ((njtot & 1) ? -v : v);
    }
    c[0] = 1.0;
    for (d = 0; d < D; d++) {
        const int n = ni[d] + nj[d];
        double tau, y1, y2, coef = 1.0;
        int j;
        if (n == 0) continue;
        tau = xi[d] - xj[d];
        y1 = 4.0 * nu * tau / (l[d] * l[d]);
        y2 = 4.0 * nu / (l[d] * l[d]);
        for (m = 0; m <= n; m++) pd[m] = 0.0;
        for (j = 0; 2 * j <= n; j++) {
            if (j > 0) coef = coef * (double)((n - 2 * j + 2) * (n - 2 * j + 1)) / (2.0 * (double)j);
            pd[n - j] = coef * pow(y2, (double)j) * pow(y1, (double)(n - 2 * j));
        }
        for (m = 0; m <= deg + n; m++) cn[m] = 0.0;
        for (a = 0; a <= deg; a++)
            for (b = 0; b <= n; b++) cn[a + b] += c[a] * pd[b];
        deg += n;
        for (m = 0; m <= deg; m++) c[m] = cn[m];
    }
    for (m = 1; m <= deg; m++) {
        if (c[m] == 0.0) continue;
        v += c[m] * cnu * yn2kn2der(nu, y, m);
    }
    return sigma * sigma * ((njtot & 1) ? -v : v);
}

static int all_zero(const int32_t *n, int64_t count)
{
    int64_t i;
    for (i = 0; i < count; i++)
        if (n[i] != 0) return 0;
    return 1;
}

static int any_rowsum_gt1(const int32_t *n, int64_t M, int D)
{
    int64_t i;
    int d;
    for (i = 0; i < M; i++) {
        long s = 0;
        for (d = 0; d < D; d++) s += n[i * D + d];
        if (s > 1) return 1;
    }
    return 0;
}

static int check_args(int kernel_id, const double *params, int nparams, int D, int hyper_deriv)
{
    (void)params;
    if (D < 1) return ORC_EARG;
    if (kernel_id == ORC_KERNEL_SE || kernel_id == ORC_KERNEL_M52) {
        if (nparams != D + 1) return ORC_EARG;
        if (hyper_deriv >= nparams) return ORC_EARG;
    } else if (kernel_id == ORC_KERNEL_RQ || kernel_id == ORC_KERNEL_MATERN) {
        if (nparams != D + 2) return ORC_EARG;
        if (hyper_deriv >= 0) return ORC_ENOTIMPL;                   /* core.py:723-726 */
    } else if (kernel_id == ORC_KERNEL_DIAGNOISE || kernel_id == ORC_KERNEL_ZERO) {
        if (nparams != 1) return ORC_EARG;
    } else {
        return ORC_EARG;
    }
    return ORC_OK;
}

/* One pair of any kernel.  noise_n: DiagonalNoiseKernel.n (D ints) or NULL (zeros). */
static double any_pair(int kernel_id, const double *params, const double *var, int D,
                       const double *xi, const double *xj, const int32_t *ni, const int32_t *nj,
                       int hyper_deriv, int symmetric, int only_first_order, const int32_t *noise_n)
{
    int d;
    switch (kernel_id) {
    case ORC_KERNEL_SE:
        return se_pair(params, D, xi, xj, ni, nj, hyper_deriv, only_first_order);
    case ORC_KERNEL_M52:
        return params[0] * params[0] * m52_pair(xi, xj, ni, nj, D, var);   /* matern.py:555 */
    case ORC_KERNEL_RQ:
        return rq_pair(params, D, xi, xj, ni, nj);
    case ORC_KERNEL_MATERN:
        return matern_pair(params, D, xi, xj, ni, nj);
    case ORC_KERNEL_DIAGNOISE: {                                            /* noise.py:103-110 */
        double val;
        if (!symmetric) return 0.0;
        for (d = 0; d < D; d++) {
            int32_t nn = noise_n ? noise_n[d] : 0;
            if (!(xi[d] == xj[d] && ni[d] == nn && nj[d] == nn)) return (hyper_deriv < 0) ? 0.0 : 2.0 * 0.0 / params[0];
        }
        val = params[0] * params[0];
        return (hyper_deriv < 0) ? val : 2.0 * val / params[hyper_deriv];
    }
    default:
        return 0.0;                                                          /* noise.py:152 */
    }
}

/* ---- Kernel.__call__: element-wise pair list (kernel/core.py:220-257) ------------------ */
int orc_kpairs(int kernel_id, const double *params, int nparams,
               const double *Xi, const double *Xj, const int32_t *ni, const int32_t *nj,
               int64_t M, int D, int hyper_deriv, int symmetric, const int32_t *noise_n,
               double *out)
{
    int64_t m;
    int d, rc, ofo = 1;
    double var[64];
    rc = check_args(kernel_id, params, nparams, D, hyper_deriv);
    if (rc) return rc;
    if (D > 64) return ORC_EARG;
    if (kernel_id == ORC_KERNEL_M52) {
        if (hyper_deriv >= 0) return ORC_ENOTIMPL;
        if (any_rowsum_gt1(ni, M, D) || any_rowsum_gt1(nj, M, D)) return ORC_EVALUE;
        for (d = 0; d < D; d++) var[d] = params[1 + d] * params[1 + d];   /* matern.py:552 */
    }
    if (kernel_id == ORC_KERNEL_SE) ofo = all_zero(ni, M * D) && all_zero(nj, M * D);
    for (m = 0; m < M; m++)
        out[m] = any_pair(kernel_id, params, var, D, Xi + m * D, Xj + m * D, ni + m * D, nj + m * D,
                          hyper_deriv, symmetric, ofo, noise_n);
    return ORC_OK;
}

/* ---- GaussianProcess.compute_Kij (gaussian_process.py:1535-1605) ----------------------- */
/* Xj == NULL  =>  Xj = Xi, nj = ni, symmetric = True.  out is (M, P) row-major: pair i*P + j. */
int orc_kbuild(int kernel_id, const double *params, int nparams,
               const double *Xi, const int32_t *ni, int64_t M,
               const double *Xj, const int32_t *nj, int64_t P, int D,
               int hyper_deriv, const int32_t *noise_n, double *out)
{
    int64_t i;
    int d, rc, ofo = 1, symmetric = 0;
    double var[64];
    rc = check_args(kernel_id, params, nparams, D, hyper_deriv);
    if (rc) return rc;
    if (D > 64) return ORC_EARG;
    if (Xj == NULL) {
        symmetric = 1;
        Xj = Xi;
        nj = ni;
        P = M;
    }
    if (kernel_id == ORC_KERNEL_M52) {
        if (hyper_deriv >= 0) return ORC_ENOTIMPL;
        if (any_rowsum_gt1(ni, M, D) || any_rowsum_gt1(nj, P, D)) return ORC_EVALUE;
        for (d = 0; d < D; d++) var[d] = params[1 + d] * params[1 + d];
    }
    if (kernel_id == ORC_KERNEL_SE) ofo = all_zero(ni, M * D) && all_zero(nj, P * D);
#pragma omp parallel for schedule(static)
    for (i = 0; i < M; i++) {
        int64_t j;
        for (j = 0; j < P; j++)
            out[i * P + j] = any_pair(kernel_id, params, var, D, Xi + i * D, Xj + j * D,
                                      ni + i * D, nj + j * D, hyper_deriv, symmetric, ofo, noise_n);
    }
    return ORC_OK;
}

/* ---- dense linear algebra (restating LAPACK dpotrf('L') / dtrtrs, row-major) ----------- */
/* Column-Crout Cholesky, lower, in place; the strict upper triangle is zeroed like
 * scipy.linalg.cholesky(lower=True) leaves it.  Returns 0, or j+1 when the leading minor of
 * order j+1 is not positive definite (LAPACK info convention). */
int64_t orc_potrf_lower(int64_t n, double *A, int64_t lda)
{
    int64_t j;
    for (j = 0; j < n; j++) {
        double *rj = A + j * lda;
        double s = rj[j];
        int64_t k, i;
        for (k = 0; k < j; k++) s -= rj[k] * rj[k];
        if (!(s > 0.0)) return j + 1;
        s = sqrt(s);
        rj[j] = s;
#pragma omp parallel for schedule(static) if (n - j > 256)
        for (i = j + 1; i < n; i++) {
            double *ri = A + i * lda;
            double t = ri[j];
            int64_t kk;
            for (kk = 0; kk < j; kk++) t -= ri[kk] * rj[kk];
            ri[j] = t / s;
        }
    }
    for (j = 0; j < n; j++) {
        int64_t k;
        for (k = j + 1; k < n; k++) A[j * lda + k] = 0.0;
    }
    return 0;
}

/* B (n x nrhs, row-major) <- L^{-1} B */
void orc_solve_lower(int64_t n, const double *L, int64_t ldl, int64_t nrhs, double *B, int64_t ldb)
{
    int64_t i, k, c;
    for (i = 0; i < n; i++) {
        double *bi = B + i * ldb;
        for (k = 0; k < i; k++) {
            double lik = L[i * ldl + k];
            const double *bk = B + k * ldb;
            for (c = 0; c < nrhs; c++) bi[c] -= lik * bk[c];
        }
        for (c = 0; c < nrhs; c++) bi[c] /= L[i * ldl + i];
    }
}

/* B (n x nrhs, row-major) <- L^{-T} B */
void orc_solve_lower_t(int64_t n, const double *L, int64_t ldl, int64_t nrhs, double *B, int64_t ldb)
{
    int64_t i, k, c;
    for (i = n - 1; i >= 0; i--) {
        double *bi = B + i * ldb;
        for (c = 0; c < nrhs; c++) bi[c] /= L[i * ldl + i];
        for (k = 0; k < i; k++) {
            double lik = L[i * ldl + k];
            double *bk = B + k * ldb;
            for (c = 0; c < nrhs; c++) bk[c] -= lik * bi[c];
        }
    }
}

/* ---- GaussianProcess.compute_K_L_alpha_ll without T (gaussian_process.py:1428-1467) ---- */
/* K_tot = K + noise_var*I + diag(err_y^2) + diag_add*I ; L = chol(K_tot) ; alpha = K_tot^-1 y ;
 * ll_data = -1/2 y.alpha - sum(log diag L) - N/2 log(2 pi).  y is already mean-subtracted.
 * K_out (N x N, may be NULL) receives the noise-free K; L_out (N x N) and alpha_out (N) are
 * required.  Returns 0, >0 = non-PD leading minor, <0 = -(orc error code). */
int64_t orc_fit(int kernel_id, const double *params, int nparams, double noise_var,
                const double *X, const int32_t *n, const double *y, const double *err_y,
                int64_t N, int D, double diag_add,
                double *K_out, double *L_out, double *alpha_out,
                double *ll_data_out, double *logdet_half_out)
{
    int64_t i, info;
    int rc;
    double quad = 0.0, ld = 0.0;
    rc = orc_kbuild(kernel_id, params, nparams, X, n, N, NULL, NULL, N, D, -1, NULL, L_out);
    if (rc) return -(int64_t)rc;
    if (K_out) memcpy(K_out, L_out, (size_t)N * (size_t)N * sizeof(double));
    for (i = 0; i < N; i++)
        L_out[i * N + i] = ((L_out[i * N + i] + noise_var) + err_y[i] * err_y[i]) + diag_add;   /* :1446-1451 */
    info = orc_potrf_lower(N, L_out, N);
    if (info) return info;
    for (i = 0; i < N; i++) alpha_out[i] = y[i];
    orc_solve_lower(N, L_out, N, 1, alpha_out, 1);
    orc_solve_lower_t(N, L_out, N, 1, alpha_out, 1);
    for (i = 0; i < N; i++) {
        quad += y[i] * alpha_out[i];
        ld += log(L_out[i * N + i]);
    }
    *logdet_half_out = ld;
    *ll_data_out = -0.5 * quad - ld - 0.5 * (double)N * log(2.0 * M_PI);
    return 0;
}

/* ---- GaussianProcess.predict, non-MCMC, no T / mu / output_transform (:965-1006) -------- */
/* mean = Kstar^T alpha ; v = L^-1 Kstar ; cov = Kss [+noise] - v^T v ; std = sqrt(diag cov).
 * noise_params != NULL adds the DiagonalNoiseKernel terms of :967-968 and :985-986.
 * work: N*M doubles.  cov_out may be NULL (then std_out must not be). */
int orc_predict(int kernel_id, const double *params, int nparams,
                const double *noise_params, const int32_t *noise_n,
                const double *X, const int32_t *n, int64_t N, int D,
                const double *L, const double *alpha,
                const double *Xs, const int32_t *ns, int64_t M,
                double *mean_out, double *std_out, double *cov_out, double *work)
{
    int64_t i, a, b;
    int rc;
    double *Kstar = work;
    double *cov = cov_out ? cov_out : (double *)malloc((size_t)M * (size_t)M * sizeof(double));
    if (!cov) return ORC_EARG;
    rc = orc_kbuild(kernel_id, params, nparams, X, n, N, Xs, ns, M, D, -1, NULL, Kstar);
    if (rc) goto done;
    /* noise kernel between X and Xstar is evaluated with symmetric=False => zeros (noise.py:109-110) */
    for (a = 0; a < M; a++) {
        double s = 0.0;
        for (i = 0; i < N; i++) s += Kstar[i * M + a] * alpha[i];
        mean_out[a] = s;
    }
    orc_solve_lower(N, L, N, M, Kstar, M);
    rc = orc_kbuild(kernel_id, params, nparams, Xs, ns, M, NULL, NULL, M, D, -1, NULL, cov);
    if (rc) goto done;
    if (noise_params) {
        double *nz = (double *)malloc((size_t)M * (size_t)M * sizeof(double));
        rc = orc_kbuild(ORC_KERNEL_DIAGNOISE, noise_params, 1, Xs, ns, M, NULL, NULL, M, D, -1, noise_n, nz);
        if (!rc)
            for (a = 0; a < M * M; a++) cov[a] = cov[a] + nz[a];
        free(nz);
        if (rc) goto done;
    }
    for (a = 0; a < M; a++)
        for (b = 0; b < M; b++) {
            double s = 0.0;
            for (i = 0; i < N; i++) s += Kstar[i * M + a] * Kstar[i * M + b];
            cov[a * M + b] -= s;
        }
    if (std_out)
        for (a = 0; a < M; a++) std_out[a] = sqrt(cov[a * M + a]);
done:
    if (!cov_out) free(cov);
    return rc;
}
