// kbuild.hip -- fused covariance-matrix builder K[i][j] = k(Xi[i], Xj[j], ni[i], nj[j]) for gfx950.
//
// Replaces GaussianProcess.compute_Kij (ref: gptools/gaussian_process.py:1535-1605): the reference
// materialises four (M*P, D) tiled arrays (ref :1591-1594) and calls the kernel on the pair list;
// here no tiling exists -- a workgroup owns a (32 x 256) tile of K, each lane keeps its column's point Xj[j], nj[j]
// in registers, the tile's rows arrive through wave-uniform (scalar) loads, and every store is a 512-byte
// contiguous wave store (8 bytes per pair).  Measured against the same launch with the ZeroKernel pair function
// (pure write pattern, 4.6-5.0 TB/s): SE runs at that ceiling (4.65-4.7 TB/s), Matern-5/2 is bound by its
// sqrt + exp + divide arithmetic (2.7-3.1 TB/s).  For the fit only the tiles of the lower triangle are launched.
// Fused epilogue: the diagonal loading of ref :1447-1451 ((K + noise_var) + err_y^2) + diag_add.
#include "kbuild_kernel.hpp"

// C[a][b] += noise_k(X[a], X[b], n[a], n[b]) for the symmetric predict(noise=True) term
// (ref: gptools/gaussian_process.py:985-986, gptools/kernel/noise.py:103-104).
template <int D>
__global__ __launch_bounds__(256) void add_noise_sym_kernel(KParams kp, const double *__restrict__ X,
                                                            const int32_t *__restrict__ n, int64_t M,
                                                            double *__restrict__ C, int64_t ldc)
{
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t a = blockIdx.y;
    if (b >= M) return;
    double xa[D], xb[D];
    int na[D], nb[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        xa[d] = X[a * D + d];
        xb[d] = X[b * D + d];
        na[d] = n[a * D + d];
        nb[d] = n[b * D + d];
    }
    C[a * ldc + b] += noise_pair<D>(kp, xa, xb, na, nb);
}

template <int KID>
static int kbuild_dispatch_d(hipStream_t st, const KParams &kp, const double *dXi, const int32_t *dni, int64_t M,
                             const double *dXj, const int32_t *dnj, int64_t P, int lower_only, int64_t i0,
                             int64_t j0, const double *d_err_y, double noise_var, double diag_add, double *dK,
                             int64_t ldk, int accumulate)
{
    dim3 grid((unsigned)((P + KB_COLS - 1) / KB_COLS), (unsigned)((M + KB_ROWS - 1) / KB_ROWS));
    dim3 block(KB_THREADS);
    if (lower_only && i0 == j0 && M == P) {        // square lower triangle: launch only the tiles that exist
        const int64_t nrt = (M + KB_ROWS - 1) / KB_ROWS;
        int64_t ntile = 0;
        for (int64_t rt = 0; rt < nrt; rt++) ntile += rt / KB_RATIO + 1;
        grid = dim3((unsigned)ntile, 1);
        lower_only = 2;
    }
#define KB_CASE(DD)                                                                                     \
    case DD:                                                                                            \
        hipLaunchKernelGGL((kbuild_kernel<KID, DD, false>), grid, block, 0, st, kp, dXi, dni, M, dXj, dnj, P,   \
                           lower_only, i0, j0, d_err_y, noise_var, diag_add, dK, ldk, accumulate,               \
                           (const KParams *)nullptr, (const double *)nullptr, (int64_t)0, KParams(), (const KParams *)nullptr);  \
        break;
    switch (kp.D) {
        KB_CASE(1) KB_CASE(2) KB_CASE(3) KB_CASE(4) KB_CASE(5) KB_CASE(6) KB_CASE(7) KB_CASE(8)
        KB_CASE(9) KB_CASE(10) KB_CASE(11) KB_CASE(12) KB_CASE(13) KB_CASE(14) KB_CASE(15) KB_CASE(16)
    default:
        gpt_set_error("kbuild: unsupported num_dim %d (max %d)", kp.D, GPT_MAX_DIM);
        return GPT_E_ARG;
    }
#undef KB_CASE
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

int launch_kbuild(hipStream_t st, const KParams &kp, const double *dXi, const int32_t *dni, int64_t M,
                  const double *dXj, const int32_t *dnj, int64_t P, int lower_only, int64_t i0, int64_t j0,
                  const double *d_err_y, double noise_var, double diag_add, double *dK, int64_t ldk, int accumulate,
                  const KParams *kp2)
{
    if (kp2 && kp2->kernel_id >= 0)
        return launch_kbuild_prod(st, kp, *kp2, dXi, dni, M, dXj, dnj, P, lower_only, i0, j0, d_err_y, noise_var, diag_add, dK, ldk,
                                  accumulate);
    gpt_jitter(st);
    if (M <= 0 || P <= 0) return GPT_OK;
    switch (kp.kernel_id) {
    case GPT_KERNEL_SE:
        return kbuild_dispatch_d<GPT_KERNEL_SE>(st, kp, dXi, dni, M, dXj, dnj, P, lower_only, i0, j0, d_err_y,
                                                noise_var, diag_add, dK, ldk, accumulate);
    case GPT_KERNEL_M52:
        return kbuild_dispatch_d<GPT_KERNEL_M52>(st, kp, dXi, dni, M, dXj, dnj, P, lower_only, i0, j0, d_err_y,
                                                 noise_var, diag_add, dK, ldk, accumulate);
    case GPT_KERNEL_DIAGNOISE:
        return kbuild_dispatch_d<GPT_KERNEL_DIAGNOISE>(st, kp, dXi, dni, M, dXj, dnj, P, lower_only, i0, j0,
                                                       d_err_y, noise_var, diag_add, dK, ldk, accumulate);
    case GPT_KERNEL_ZERO:
        return kbuild_dispatch_d<GPT_KERNEL_ZERO>(st, kp, dXi, dni, M, dXj, dnj, P, lower_only, i0, j0, d_err_y,
                                                  noise_var, diag_add, dK, ldk, accumulate);
    case GPT_KERNEL_RQ:
        return kbuild_dispatch_d<GPT_KERNEL_RQ>(st, kp, dXi, dni, M, dXj, dnj, P, lower_only, i0, j0, d_err_y,
                                                noise_var, diag_add, dK, ldk, accumulate);
    case GPT_KERNEL_MATERN:
        return kbuild_dispatch_d<GPT_KERNEL_MATERN>(st, kp, dXi, dni, M, dXj, dnj, P, lower_only, i0, j0, d_err_y,
                                                    noise_var, diag_add, dK, ldk, accumulate);
    default:
        gpt_set_error("kbuild: unknown kernel_id %d", kp.kernel_id);
        return GPT_E_ARG;
    }
}

template <int KID>
static int kpairs_dispatch_d(hipStream_t st, const KParams &kp, const double *dXi, const double *dXj,
                             const int32_t *dni, const int32_t *dnj, int64_t M, double *dout, int accumulate)
{
    dim3 grid((unsigned)((M + 255) / 256)), block(256);
#define KP_CASE(DD)                                                                                   \
    case DD:                                                                                          \
        hipLaunchKernelGGL((kpairs_kernel<KID, DD>), grid, block, 0, st, kp, dXi, dXj, dni, dnj, M, dout, accumulate, KParams()); \
        break;
    switch (kp.D) {
        KP_CASE(1) KP_CASE(2) KP_CASE(3) KP_CASE(4) KP_CASE(5) KP_CASE(6) KP_CASE(7) KP_CASE(8)
        KP_CASE(9) KP_CASE(10) KP_CASE(11) KP_CASE(12) KP_CASE(13) KP_CASE(14) KP_CASE(15) KP_CASE(16)
    default:
        gpt_set_error("kpairs: unsupported num_dim %d (max %d)", kp.D, GPT_MAX_DIM);
        return GPT_E_ARG;
    }
#undef KP_CASE
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

int launch_kpairs(hipStream_t st, const KParams &kp, const double *dXi, const double *dXj,
                  const int32_t *dni, const int32_t *dnj, int64_t M, double *dout, int accumulate, const KParams *kp2)
{
    if (M <= 0) return GPT_OK;
    if (kp2 && kp2->kernel_id >= 0) return launch_kpairs_prod(st, kp, *kp2, dXi, dXj, dni, dnj, M, dout, accumulate);
    switch (kp.kernel_id) {
    case GPT_KERNEL_SE: return kpairs_dispatch_d<GPT_KERNEL_SE>(st, kp, dXi, dXj, dni, dnj, M, dout, accumulate);
    case GPT_KERNEL_M52: return kpairs_dispatch_d<GPT_KERNEL_M52>(st, kp, dXi, dXj, dni, dnj, M, dout, accumulate);
    case GPT_KERNEL_DIAGNOISE: return kpairs_dispatch_d<GPT_KERNEL_DIAGNOISE>(st, kp, dXi, dXj, dni, dnj, M, dout, accumulate);
    case GPT_KERNEL_ZERO: return kpairs_dispatch_d<GPT_KERNEL_ZERO>(st, kp, dXi, dXj, dni, dnj, M, dout, accumulate);
    case GPT_KERNEL_RQ: return kpairs_dispatch_d<GPT_KERNEL_RQ>(st, kp, dXi, dXj, dni, dnj, M, dout, accumulate);
    case GPT_KERNEL_MATERN: return kpairs_dispatch_d<GPT_KERNEL_MATERN>(st, kp, dXi, dXj, dni, dnj, M, dout, accumulate);
    default:
        gpt_set_error("kpairs: unknown kernel_id %d", kp.kernel_id);
        return GPT_E_ARG;
    }
}

int launch_add_noise_sym(hipStream_t st, const KParams &kp, const double *dX, const int32_t *dn, int64_t M,
                         double *C, int64_t ldc)
{
    if (M <= 0) return GPT_OK;
    dim3 grid((unsigned)((M + 255) / 256), (unsigned)M), block(256);
#define AN_CASE(DD)                                                                                  \
    case DD:                                                                                         \
        hipLaunchKernelGGL((add_noise_sym_kernel<DD>), grid, block, 0, st, kp, dX, dn, M, C, ldc);   \
        break;
    switch (kp.D) {
        AN_CASE(1) AN_CASE(2) AN_CASE(3) AN_CASE(4) AN_CASE(5) AN_CASE(6) AN_CASE(7) AN_CASE(8)
        AN_CASE(9) AN_CASE(10) AN_CASE(11) AN_CASE(12) AN_CASE(13) AN_CASE(14) AN_CASE(15) AN_CASE(16)
    default:
        gpt_set_error("add_noise: unsupported num_dim %d", kp.D);
        return GPT_E_ARG;
    }
#undef AN_CASE
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

// ------------------------------------------------------------------------------------------------
// Gradient of the LML data term (ref: gptools/gaussian_process.py:1471-1520):
//     d ll / d theta_h = 1/2 ( alpha^T dK_h alpha - tr(K_tot^-1 dK_h) ) = 1/2 sum_ab (alpha_a alpha_b - W_ab) dK_h[a][b]
// with W = K_tot^-1 resident (lower triangle).  dK_h is never materialised: the tile loop of the builder evaluates
// the pair function with hyper_deriv = h (up to GR_MAXH parameters of one kernel term per launch) and folds it into
// the weight; symmetric, so the strictly lower part counts twice.  Deterministic: per-workgroup partial sums, added
// up by the host in index order.  Slot GR_MAXH of every partial row carries sum_i (alpha_i^2 - W_ii) (the noise term).
// ------------------------------------------------------------------------------------------------
#define GR_MAXH 8
template <int KID, int D>
__global__ __launch_bounds__(KB_THREADS) void grad_reduce_kernel(
    KParams kp, int nh, const int *__restrict__ hidx_unused, int h0, int h1, int h2, int h3, int h4, int h5, int h6, int h7,
    const double *__restrict__ X, const int32_t *__restrict__ nn, int64_t N, const double *__restrict__ alpha,
    const double *__restrict__ W, int64_t ldw, double *__restrict__ partial)
{
    (void)hidx_unused;
    const int hl[GR_MAXH] = {h0, h1, h2, h3, h4, h5, h6, h7};
    constexpr int64_t R = KB_RATIO;
    const int64_t b = blockIdx.x;
    int64_t g = (int64_t)((sqrt(1.0 + 8.0 * (double)b / (double)R) - 1.0) * 0.5);
    while (g > 0 && R * g * (g + 1) / 2 > b) g--;
    while (R * (g + 1) * (g + 2) / 2 <= b) g++;
    const int64_t rem = b - R * g * (g + 1) / 2;
    const int64_t rt = R * g + rem / (g + 1), ct = rem % (g + 1);
    double acc[GR_MAXH + 1];
#pragma unroll
    for (int h = 0; h <= GR_MAXH; h++) acc[h] = 0.0;
    const int64_t rbase = rt * KB_ROWS, j = ct * KB_COLS + threadIdx.x;
    if (rbase < N && j < N) {
        double xj[D];
        int njr[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            xj[d] = X[j * D + d];
            njr[d] = nn[j * D + d];
        }
        const double aj = alpha[j];
        const int64_t rend = (rbase + KB_ROWS < N) ? rbase + KB_ROWS : N;
        for (int64_t i = rbase; i < rend; i++) {
            if (j > i) continue;                                    // lower triangle (wave-uniform row, per-lane column)
            double xi[D];
            int nir[D];
#pragma unroll
            for (int d = 0; d < D; d++) {
                xi[d] = X[i * D + d];
                nir[d] = nn[i * D + d];
            }
            const double w = alpha[i] * aj - W[i * ldw + j];
            const double w2 = (i == j) ? w : 2.0 * w;
#pragma unroll
            for (int h = 0; h < GR_MAXH; h++)
                if (h < nh) {
                    KParams kh = kp;
                    kh.hyper_deriv = hl[h];
                    acc[h] = fma(w2, any_pair<KID, D>(kh, xi, xj, nir, njr), acc[h]);
                }
            if (i == j) acc[GR_MAXH] += w;
        }
    }
    __shared__ double red[KB_THREADS / 64][GR_MAXH + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int h = 0; h <= GR_MAXH; h++) {
        double v = acc[h];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) red[wave][h] = v;
    }
    __syncthreads();
    if (threadIdx.x <= GR_MAXH) {
        double v = 0.0;
        for (int wv = 0; wv < KB_THREADS / 64; wv++) v += red[wv][threadIdx.x];
        partial[(int64_t)blockIdx.x * (GR_MAXH + 1) + threadIdx.x] = v;
    }
}

int grad_reduce_blocks(int64_t N)
{
    const int64_t nrt = (N + KB_ROWS - 1) / KB_ROWS;
    int64_t ntile = 0;
    for (int64_t rt = 0; rt < nrt; rt++) ntile += rt / KB_RATIO + 1;
    return (int)ntile;
}

template <int KID>
static int grad_dispatch_d(hipStream_t st, const KParams &kp, int nh, const int *hl, const double *dX, const int32_t *dn,
                           int64_t N, const double *dalpha, const double *dW, int64_t ldw, double *dpartial)
{
    dim3 grid((unsigned)grad_reduce_blocks(N)), block(KB_THREADS);
    int h[GR_MAXH];
    for (int i = 0; i < GR_MAXH; i++) h[i] = i < nh ? hl[i] : -1;
#define GR_CASE(DD)                                                                                                  \
    case DD:                                                                                                         \
        hipLaunchKernelGGL((grad_reduce_kernel<KID, DD>), grid, block, 0, st, kp, nh, (const int *)nullptr, h[0], h[1], \
                           h[2], h[3], h[4], h[5], h[6], h[7], dX, dn, N, dalpha, dW, ldw, dpartial);                 \
        break;
    switch (kp.D) {
        GR_CASE(1) GR_CASE(2) GR_CASE(3) GR_CASE(4) GR_CASE(5) GR_CASE(6) GR_CASE(7) GR_CASE(8)
        GR_CASE(9) GR_CASE(10) GR_CASE(11) GR_CASE(12) GR_CASE(13) GR_CASE(14) GR_CASE(15) GR_CASE(16)
    default:
        gpt_set_error("grad_reduce: unsupported num_dim %d", kp.D);
        return GPT_E_ARG;
    }
#undef GR_CASE
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

// up to GR_MAXH hyper-derivative indices hl[] of ONE kernel term; partial: grad_reduce_blocks(N) x (GR_MAXH + 1)
int launch_grad_reduce(hipStream_t st, const KParams &kp, int nh, const int *hl, const double *dX, const int32_t *dn,
                       int64_t N, const double *dalpha, const double *dW, int64_t ldw, double *dpartial)
{
    if (nh < 0 || nh > GR_MAXH) return GPT_E_ARG;
    if (kp.kernel_id == GPT_KERNEL_SE)
        return grad_dispatch_d<GPT_KERNEL_SE>(st, kp, nh, hl, dX, dn, N, dalpha, dW, ldw, dpartial);
    gpt_set_error("hyper-parameter derivatives exist for the squared-exponential kernel only (ref: matern.py:543-544)");
    return GPT_E_NOTIMPL;
}
