// potrf.hip -- leaf kernels of the blocked right-looking Cholesky for gfx950 (fp64).
//
// Replaces LAPACK dpotrf as reached by scipy.linalg.cholesky(K_tot, lower=True)
// (ref: gptools/gaussian_process.py:1452).  The host-side blocking (outer block columns, recursive
// panel, look-ahead on a second stream) lives in api.hip; this file holds
//   potf2_diag_kernel  : one workgroup factors a 128x128 diagonal block entirely inside LDS,
//                        16-column inner blocks; inner TRSM/SYRK on v_mfma_f64_16x16x4_f64; the
//                        16x16 pivot blocks are factored by one wave with one matrix row per lane
//                        and v_readlane broadcasts (no barriers inside a pivot block); also emits
//                        the inverses of the 16x16 diagonal blocks of L ("invd").
//   trsm_panel_kernel  : X * L_kk^T = B for the rows below a diagonal block; one wave per 16
//                        rows, blocked forward substitution, every multiply on MFMA; the
//                        accumulator -> operand re-layout goes through a private LDS strip.
// MFMA f64 16x16x4 layouts (cdna_hip_programming.md section 3, verified by scratch/mfma_probe.hip):
//   A operand: lane l holds A[i = l & 15][k = l >> 4];  B operand: lane l holds B[k = l >> 4][j = l & 15];
//   C/D: lane l, register r holds D[row = (l >> 4) + 4 r][col = l & 15].
#include "common.hpp"

#define PD_NB 128
#define PD_PITCH 130          // LDS row pitch (doubles): 16 rows x (lane>>4) fragment reads are conflict-free
#define PD_TP 18              // pitch of the 16x16 inverse scratch

__device__ __forceinline__ double bcast_lane(double v, int srclane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

// Factor the 16x16 pivot block jb of S (one wave; row (lane & 15) of the block lives in a[0..15]),
// write L_jj back, and write inv(L_jj) to T (LDS) and invd_out (global, row-major 16x16).
__device__ __forceinline__ void pivot_block_16(double (*S)[PD_PITCH], double (*T)[PD_TP], int jb, int lane,
                                               double *invd_out, int32_t *info, int64_t info_col0)
{
    const int row = lane & 15;
    double a[16], rd[16];
#pragma unroll
    for (int c = 0; c < 16; c++) a[c] = S[jb * 16 + row][jb * 16 + c];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        double d = bcast_lane(a[j], j);
        if (!(d > 0.0)) {                                   // not positive definite (LAPACK info = j + 1)
            if (lane == 0) atomicCAS(info, 0, (int32_t)(info_col0 + jb * 16 + j + 1));
            d = 1.0;
        }
        const double s = sqrt(d);
        const double inv = 1.0 / s;
        rd[j] = inv;
        a[j] = (row == j) ? s : a[j] * inv;
#pragma unroll
        for (int c = j + 1; c < 16; c++) {
            const double l = bcast_lane(a[j], c);
            a[c] = fma(-a[j], l, a[c]);
        }
    }
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; c++)
            if (c <= row) S[jb * 16 + row][jb * 16 + c] = a[c];
    }
    // inverse, one column per lane: x = L^-1 e_col by forward substitution
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        double s = (i == row) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; k++) s = fma(-bcast_lane(a[k], i), x[k], s);
        x[i] = s * rd[i];
    }
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            T[i][row] = x[i];
            invd_out[jb * 256 + i * 16 + row] = x[i];
        }
    }
}

__global__ __launch_bounds__(256) void potf2_diag_kernel(double *__restrict__ A, int64_t lda,
                                                         double *__restrict__ invd, int32_t *info,
                                                         int64_t info_col0)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double (*S)[PD_PITCH] = reinterpret_cast<double (*)[PD_PITCH]>(smem);
    double (*T)[PD_TP] = reinterpret_cast<double (*)[PD_TP]>(smem + PD_NB * PD_PITCH);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;

    for (int idx = tid; idx < PD_NB * PD_NB / 2; idx += 256) {
        const int r = idx / (PD_NB / 2), c2 = (idx % (PD_NB / 2)) * 2;
        const f64x2 v = *reinterpret_cast<const f64x2 *>(A + (int64_t)r * lda + c2);
        S[r][c2] = v[0];
        S[r][c2 + 1] = v[1];
    }
    __syncthreads();
    if (wave == 0) pivot_block_16(S, T, 0, lane, invd, info, info_col0);
    __syncthreads();

    constexpr int NB16 = PD_NB / 16;
    for (int jb = 0; jb < NB16; jb++) {
        // (b) strip solve: X_ti = B_ti * inv(L_jj)^T for the 16-row tiles below the pivot block
        for (int ti = jb + 1 + wave; ti < NB16; ti += 4) {
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            double av[4], bv[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                av[kk] = S[ti * 16 + fr][jb * 16 + fk + 4 * kk];
                bv[kk] = T[fr][fk + 4 * kk];
            }
#pragma unroll
            for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) S[ti * 16 + fk + 4 * r][jb * 16 + fr] = acc[r];
        }
        __syncthreads();
        if (jb + 1 >= NB16) break;
        // (c) trailing update inside the block: S_titj -= X_ti X_tj^T, jb < tj <= ti.
        //     Wave 0 takes the next pivot tile first and factors it while waves 1..3 do the rest.
        const int rem = NB16 - 1 - jb;                 // tiles per side of the trailing part
        const int ntile = rem * (rem + 1) / 2;
        const int first = (wave == 0) ? 0 : wave;      // linear tile ids: 0 = next pivot tile
        const int step = (wave == 0) ? ntile : 3;      // wave 0 handles only tile 0
        for (int t = first; t < ntile; t += step) {
            int a_ = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while (a_ * (a_ + 1) / 2 > t) a_--;
            while ((a_ + 1) * (a_ + 2) / 2 <= t) a_++;
            const int b_ = t - a_ * (a_ + 1) / 2;
            const int ti = jb + 1 + a_, tj = jb + 1 + b_;
            f64x4 acc;
#pragma unroll
            for (int r = 0; r < 4; r++) acc[r] = S[ti * 16 + fk + 4 * r][tj * 16 + fr];
            double av[4], bv[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                av[kk] = -S[ti * 16 + fr][jb * 16 + fk + 4 * kk];
                bv[kk] = S[tj * 16 + fr][jb * 16 + fk + 4 * kk];
            }
#pragma unroll
            for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) S[ti * 16 + fk + 4 * r][tj * 16 + fr] = acc[r];
            if (wave == 0) break;
        }
        if (wave == 0) pivot_block_16(S, T, jb + 1, lane, invd, info, info_col0);
        __syncthreads();
    }

    for (int idx = tid; idx < PD_NB * PD_NB; idx += 256) {
        const int r = idx / PD_NB, c = idx % PD_NB;
        if (c <= r) A[(int64_t)r * lda + c] = S[r][c];
    }
}

int launch_potf2_diag(hipStream_t st, double *A, int64_t lda, double *invd, int32_t *info, int64_t info_base)
{
    static bool attr_set = false;
    const size_t shmem = (size_t)(PD_NB * PD_PITCH + 16 * PD_TP) * sizeof(double);
    if (!attr_set) {
        GPT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(potf2_diag_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        attr_set = true;
    }
    hipLaunchKernelGGL(potf2_diag_kernel, dim3(1), dim3(256), shmem, st, A, lda, invd, info, info_base);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

// ---- panel TRSM: B (m x 128) <- B * L^-T, L = 128x128 lower, invd = inverses of its 16x16 diagonal blocks ----
#define TP_WAVES 2
__global__ __launch_bounds__(64 * TP_WAVES) void trsm_panel_kernel(int64_t m, const double *__restrict__ L,
                                                                   int64_t ldl, const double *__restrict__ invd,
                                                                   double *__restrict__ B, int64_t ldb)
{
    __shared__ __attribute__((aligned(16))) double Xs[TP_WAVES][16][PD_PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int64_t row0 = ((int64_t)blockIdx.x * TP_WAVES + wave) * 16;
    if (row0 >= m) return;
    double (*X)[PD_PITCH] = Xs[wave];
    constexpr int NB16 = PD_NB / 16;
#pragma unroll 1
    for (int j = 0; j < NB16; j++) {
        f64x4 acc;
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = B[(row0 + fk + 4 * r) * ldb + j * 16 + fr];
        double dv[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) dv[kk] = invd[j * 256 + fr * 16 + fk + 4 * kk];
        for (int c = 0; c < j; c++) {
            double av[4], bv[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                bv[kk] = L[(int64_t)(j * 16 + fr) * ldl + c * 16 + fk + 4 * kk];
                av[kk] = -X[fr][c * 16 + fk + 4 * kk];
            }
#pragma unroll
            for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], acc, 0, 0, 0);
        }
        // re-layout the accumulator (C layout) into an A operand through the private LDS strip
#pragma unroll
        for (int r = 0; r < 4; r++) X[fk + 4 * r][j * 16 + fr] = acc[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        double av[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) av[kk] = X[fr][j * 16 + fk + 4 * kk];
        f64x4 res = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; kk++) res = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], dv[kk], res, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            X[fk + 4 * r][j * 16 + fr] = res[r];
            B[(row0 + fk + 4 * r) * ldb + j * 16 + fr] = res[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
}

int launch_trsm_panel(hipStream_t st, int64_t m, const double *L, int64_t ldl, const double *invd, double *B,
                      int64_t ldb)
{
    if (m <= 0) return GPT_OK;
    if (m % 16) {
        gpt_set_error("trsm_panel: m must be a multiple of 16 (m=%lld)", (long long)m);
        return GPT_E_ARG;
    }
    const int64_t nwave = m / 16;
    const unsigned grid = (unsigned)((nwave + TP_WAVES - 1) / TP_WAVES);
    hipLaunchKernelGGL(trsm_panel_kernel, dim3(grid), dim3(64 * TP_WAVES), 0, st, m, L, ldl, invd, B, ldb);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}
