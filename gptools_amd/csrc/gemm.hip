// gemm.hip -- fp64 MFMA "NT" GEMM / SYRK for gfx950:  C = beta*C + alpha * A * B^T,
// A (m x k), B (n x k), C (m x n), all row-major with the contraction index contiguous.
//
// This is the roofline kernel of the Cholesky (trailing SYRK/GEMM updates of LAPACK dpotrf, which
// the reference reaches through scipy.linalg.cholesky, ref: gptools/gaussian_process.py:1452) and
// of predict's cov = K** - v^T v (ref: gptools/gaussian_process.py:987).
//
// Design (CDNA4):
//   * v_mfma_f64_16x16x4_f64 (64 cycles/SIMD, 2048 flop): a wave owns a (BM/2 x BN/2) sub-tile as
//     RM x RN accumulator fragments held in AGPR/VGPRs for the whole k loop.
//   * LDS tile layout [k/4][row][4]: the fragment read "row = lane&15, k = lane>>4" of one
//     16-row block is 512 contiguous bytes -> conflict-free ds_read_b64; staging writes are
//     ds_write_b128.
//   * global -> register -> LDS staging in full 128-byte row segments, double-buffered so the
//     loads of k-tile t+1 are in flight during the MFMAs of k-tile t (one barrier per k-tile).
//   * XCD-aware bijective remap of the linear workgroup id so that each XCD's L2 sees a
//     contiguous run of tiles (shared A row-panels); `tri` enumerates only the lower tiles.
#include "common.hpp"

#define GM_BK 16

__device__ __forceinline__ int64_t xcd_remap(int64_t pid, int64_t nwg)
{
    const int64_t q = nwg >> 3, r = nwg & 7;
    const int64_t xcd = pid & 7, local = pid >> 3;
    const int64_t base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

template <int BM, int BN, int WPS>
__global__ __launch_bounds__(256, WPS) void gemm_nt_kernel(
    int64_t m, int64_t n, int64_t k, double alpha, const double *__restrict__ A, int64_t lda,
    const double *__restrict__ B, int64_t ldb, double beta, double *__restrict__ C, int64_t ldc,
    int tri, int64_t ntm, int64_t ntn, int64_t nwg)
{
    constexpr int WM = BM / 2, WN = BN / 2;          // wave sub-tile
    constexpr int RM = WM / 16, RN = WN / 16;        // MFMA fragment repeats
    constexpr int EPA = BM / 16, EPB = BN / 16;      // doubles staged per thread per k-tile
    constexpr int TPRA = GM_BK / EPA, TPRB = GM_BK / EPB;

    __shared__ __attribute__((aligned(16))) double sA[2][GM_BK / 4][BM][4];
    __shared__ __attribute__((aligned(16))) double sB[2][GM_BK / 4][BN][4];

    // ---- tile coordinates ----
    const int64_t id = xcd_remap(blockIdx.x, nwg);
    int64_t ti, tj;
    if (tri) {
        const int64_t ntri = ntn * (ntn + 1) / 2;
        if (id < ntri) {
            ti = (int64_t)((sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
            while (ti * (ti + 1) / 2 > id) ti--;
            while ((ti + 1) * (ti + 2) / 2 <= id) ti++;
            tj = id - ti * (ti + 1) / 2;
        } else {
            const int64_t r = id - ntri;
            ti = ntn + r / ntn;
            tj = r % ntn;
        }
    } else {
        ti = id / ntn;
        tj = id % ntn;
    }
    const int64_t row0 = ti * BM, col0 = tj * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- staging assignment ----
    const int a_row = tid / TPRA, a_seg = tid % TPRA;
    const int b_row = tid / TPRB, b_seg = tid % TPRB;
    int64_t ga_row = row0 + a_row;
    if (ga_row >= m) ga_row = m - 1;
    int64_t gb_row = col0 + b_row;
    if (gb_row >= n) gb_row = n - 1;
    const double *pa = A + ga_row * lda + a_seg * EPA;
    const double *pb = B + gb_row * ldb + b_seg * EPB;

    f64x2 ra[EPA / 2], rb[EPB / 2];
    f64x4 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; i++)
#pragma unroll
        for (int j = 0; j < RN; j++) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};

    const int64_t nk = k / GM_BK;

#define GM_LOAD(kt)                                                                        \
    do {                                                                                   \
        _Pragma("unroll") for (int q = 0; q < EPA / 2; q++)                                \
            ra[q] = *reinterpret_cast<const f64x2 *>(pa + (int64_t)(kt) * GM_BK + 2 * q);  \
        _Pragma("unroll") for (int q = 0; q < EPB / 2; q++)                                \
            rb[q] = *reinterpret_cast<const f64x2 *>(pb + (int64_t)(kt) * GM_BK + 2 * q);  \
    } while (0)

#define GM_STORE(buf)                                                                                   \
    do {                                                                                                \
        _Pragma("unroll") for (int q = 0; q < EPA / 2; q++) {                                           \
            const int kk = a_seg * EPA + 2 * q;                                                         \
            *reinterpret_cast<f64x2 *>(&sA[buf][kk >> 2][a_row][kk & 3]) = ra[q];                       \
        }                                                                                               \
        _Pragma("unroll") for (int q = 0; q < EPB / 2; q++) {                                           \
            const int kk = b_seg * EPB + 2 * q;                                                         \
            *reinterpret_cast<f64x2 *>(&sB[buf][kk >> 2][b_row][kk & 3]) = rb[q];                       \
        }                                                                                               \
    } while (0)

    GM_LOAD(0);
    GM_STORE(0);
    __syncthreads();

    const int fr = lane & 15, fk = lane >> 4;
    for (int64_t kt = 0; kt < nk; kt++) {
        const int cur = (int)(kt & 1);
        if (kt + 1 < nk) GM_LOAD(kt + 1);
#pragma unroll
        for (int g = 0; g < GM_BK / 4; g++) {
            double af[RM], bf[RN];
#pragma unroll
            for (int i = 0; i < RM; i++) af[i] = sA[cur][g][wm * WM + i * 16 + fr][fk];
#pragma unroll
            for (int j = 0; j < RN; j++) bf[j] = sB[cur][g][wn * WN + j * 16 + fr][fk];
#pragma unroll
            for (int i = 0; i < RM; i++)
#pragma unroll
                for (int j = 0; j < RN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            GM_STORE(cur ^ 1);
            __syncthreads();
        }
    }
#undef GM_LOAD
#undef GM_STORE

    // ---- epilogue: C = beta*C + alpha*acc  (fragment: col = lane&15, row = (lane>>4) + 4*r) ----
#pragma unroll
    for (int i = 0; i < RM; i++) {
#pragma unroll
        for (int j = 0; j < RN; j++) {
            const int64_t col = col0 + wn * WN + j * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int64_t row = row0 + wm * WM + i * 16 + fk + 4 * r;
                if (row < m && col < n) {
                    double *pc = C + row * ldc + col;
                    double v = alpha * acc[i][j][r];
                    if (beta != 0.0) v = fma(beta, *pc, v);
                    *pc = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WPS>
static int gemm_launch_t(hipStream_t st, int64_t m, int64_t n, int64_t k, double alpha, const double *A,
                         int64_t lda, const double *B, int64_t ldb, double beta, double *C, int64_t ldc, int tri)
{
    const int64_t ntm = (m + BM - 1) / BM, ntn = (n + BN - 1) / BN;
    int64_t nwg;
    if (tri) {
        if (ntm < ntn) {
            gpt_set_error("gemm_nt: tri requires m >= n");
            return GPT_E_ARG;
        }
        nwg = ntn * (ntn + 1) / 2 + (ntm - ntn) * ntn;
    } else {
        nwg = ntm * ntn;
    }
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WPS>), dim3((unsigned)nwg), dim3(256), 0, st, m, n, k, alpha, A,
                       lda, B, ldb, beta, C, ldc, tri, ntm, ntn, nwg);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

int launch_gemm_nt(hipStream_t st, int64_t m, int64_t n, int64_t k, double alpha, const double *A, int64_t lda,
                   const double *B, int64_t ldb, double beta, double *C, int64_t ldc, int tri, int force_tile)
{
    if (m <= 0 || n <= 0) return GPT_OK;
    if (k <= 0 || (k % GM_BK) != 0 || (lda & 1) || (ldb & 1)) {
        gpt_set_error("gemm_nt: k must be a positive multiple of %d and lda/ldb even (k=%lld)", GM_BK, (long long)k);
        return GPT_E_ARG;
    }
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) {
        gpt_set_error("gemm_nt: A and B must be 16-byte aligned");
        return GPT_E_ARG;
    }
    int tile = force_tile;
    if (tile == 0) {
        // 128x128 macro-tiles once they alone can fill the 256 CUs, else 64x64 to spread the work
        const int64_t t128 = ((m + 127) / 128) * ((n + 127) / 128) / (tri ? 2 : 1);
        tile = (t128 >= 192) ? 128 : 64;
    }
    if (tile == 128) return gemm_launch_t<128, 128, 2>(st, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tri);
    return gemm_launch_t<64, 64, 2>(st, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tri);
}
