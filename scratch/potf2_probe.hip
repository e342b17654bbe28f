#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../gptools_amd/csrc/common.hpp"
void gpt_set_error(const char*, ...) {}
#define PD_NB 128
#define PD_PITCH 130          // LDS row pitch (doubles): 16 rows x (lane>>4) fragment reads are conflict-free
#define PD_TP 18              // pitch of the 16x16 inverse scratch

__device__ __forceinline__ double bcast_lane(double v, int srclane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(d) from v_rsq_f64 (relative error ~2^-26) plus one Newton step: error 1.5 * 2^-52 -- an extra rounding of
// the same size as the Cholesky's own.  The pivot chain of the factorisation is instruction-issue bound on a single
// wave, so the IEEE sqrt + divide sequences (~60 instructions) are replaced by 5; the diagonal entry sqrt(d) is
// simply d * (1/sqrt(d)).  Checked against scipy/LAPACK factors in tests/test_gpu_parity.py.
__device__ __forceinline__ double rsqrt_nr(double d)
{
    const double y0 = __builtin_amdgcn_rsq(d);
    const double t = d * y0, h = 0.5 * y0;
    const double u = fma(-t, y0, 1.0);
    return fma(h, u, y0);
}

template <int J>
struct PivotUpd {
    template <int C>
    static __device__ __forceinline__ void run(double (&a)[16], double (&x)[16])
    {
        const double l = bcast_lane(a[J], C);            // L[c][j], wave-uniform (SGPR pair)
        a[C] = fma(-a[J], l, a[C]);
        x[C] = fma(-l, x[J], x[C]);
        // keep each {2 x v_readlane, 2 x v_fma} group together: left alone, hipcc hoists all 15 broadcasts of a column
        // ahead of the FMAs, runs out of SGPRs and spills every multiplier with v_writelane/s_nop, which doubles
        // the instruction count of this issue-bound chain.
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (C + 1 < 16) run<C + 1>(a, x);
    }
};

// One column of the pivot block.  d is the (wave-uniform) pivot; the NEXT pivot is formed in the uniform domain as
// a[j+1][j+1] - (a[j+1][j] * inv)^2 from two scalar broadcasts that do not wait for this column's updates, so the
// rsqrt of column j+1 overlaps the rank-1 update of column j.
template <int J>
__device__ __forceinline__ void pivot_col(double (&a)[16], double (&x)[16], double &d, int &bad)
{
    if (!(d > 0.0)) {                                     // not positive definite (LAPACK info = j + 1)
        if (bad == 0) bad = J + 1;
        d = 1.0;
    }
    const double inv = rsqrt_nr(d);
    if constexpr (J + 1 < 16) {
        const double p = bcast_lane(a[J], J + 1), q = bcast_lane(a[J + 1], J + 1);
        const double pl = p * inv;
        d = fma(-pl, pl, q);
    }
    a[J] *= inv;                                          // lane J: d * inv = sqrt(d)
    x[J] *= inv;
    if constexpr (J + 1 < 16) PivotUpd<J>::template run<J + 1>(a, x);
}

// Factor the 16x16 pivot block jb of S with one wave: lane (l & 15) keeps ROW l of the block in a[0..15] and, at
// the same time, COLUMN l of inv(L_jj) in x[0..15]; both recurrences consume the same broadcast L[c][j], so the
// inverse costs one extra FMA per broadcast.  Writes L_jj back to S, inv(L_jj) to T (LDS) and invd_out (global).
__device__ long long *g_dbg2;
__device__ __forceinline__ void pivot_block_16(double (*S)[PD_PITCH], double (*T)[PD_TP], int jb, int lane,
                                               double *invd_out, int32_t *info, int64_t info_col0)
{
    const int row = lane & 15;
    double a[16], x[16];
#pragma unroll
    for (int c = 0; c < 16; c++) {
        a[c] = S[jb * 16 + row][jb * 16 + c];
        x[c] = (c == row) ? 1.0 : 0.0;
    }
    long long ta = __builtin_readcyclecounter();
    int bad = 0;
    double d = bcast_lane(a[0], 0);
    pivot_col<0>(a, x, d, bad);
    pivot_col<1>(a, x, d, bad);
    pivot_col<2>(a, x, d, bad);
    pivot_col<3>(a, x, d, bad);
    pivot_col<4>(a, x, d, bad);
    pivot_col<5>(a, x, d, bad);
    pivot_col<6>(a, x, d, bad);
    pivot_col<7>(a, x, d, bad);
    pivot_col<8>(a, x, d, bad);
    pivot_col<9>(a, x, d, bad);
    pivot_col<10>(a, x, d, bad);
    pivot_col<11>(a, x, d, bad);
    pivot_col<12>(a, x, d, bad);
    pivot_col<13>(a, x, d, bad);
    pivot_col<14>(a, x, d, bad);
    pivot_col<15>(a, x, d, bad);
    asm volatile("" : "+v"(a[15]), "+v"(x[15]));
    long long tb = __builtin_readcyclecounter();
    if (bad != 0 && lane == 0) atomicCAS(info, 0, (int32_t)(info_col0 + jb * 16 + bad));
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; c++) {
            if (c <= row) S[jb * 16 + row][jb * 16 + c] = a[c];
            T[c][row] = x[c];
            invd_out[jb * 256 + c * 16 + row] = x[c];
        }
    }
    long long tc = __builtin_readcyclecounter();
    if (lane == 0) { g_dbg2[jb * 2] = tb - ta; g_dbg2[jb * 2 + 1] = tc - tb; }
}

#define PD_THREADS 512
#define PD_WAVES (PD_THREADS / 64)

__global__ __launch_bounds__(PD_THREADS) void potf2_diag_kernel(double *__restrict__ A, int64_t lda,
                                                                double *__restrict__ invd, int32_t *info,
                                                                int64_t info_col0, long long *dbg)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double (*S)[PD_PITCH] = reinterpret_cast<double (*)[PD_PITCH]>(smem);
    double (*T)[PD_TP] = reinterpret_cast<double (*)[PD_TP]>(smem + PD_NB * PD_PITCH);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;

    for (int pass = 0; pass < 3; pass++) {
    __syncthreads();
    {   // whole 128 x 128 block -> LDS, all 16 loads of a thread in flight together (one 1-KiB row per wave-instruction)
        constexpr int NLD = PD_NB * PD_NB / 2 / PD_THREADS;
        f64x2 v[NLD];
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int idx = tid + q * PD_THREADS;
            const int r = idx / (PD_NB / 2), c2 = (idx % (PD_NB / 2)) * 2;
            v[q] = *reinterpret_cast<const f64x2 *>(A + (int64_t)r * lda + c2);
        }
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int idx = tid + q * PD_THREADS;
            const int r = idx / (PD_NB / 2), c2 = (idx % (PD_NB / 2)) * 2;
            *reinterpret_cast<f64x2 *>(&S[r][c2]) = v[q];
        }
    }
    __syncthreads();
    if (wave == 0) pivot_block_16(S, T, 0, lane, invd, info, info_col0);
    __syncthreads();

    constexpr int NB16 = PD_NB / 16;
    for (int jb = 0; jb < NB16; jb++) {
        // (b) strip solve: X_ti = B_ti * inv(L_jj)^T for the 16-row tiles below the pivot block
        {
            const int ti = jb + 1 + wave;
            if (ti < NB16) {
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
        }
        __syncthreads();
        if (jb + 1 >= NB16) break;
        // (c) trailing update inside the block: S_titj -= X_ti X_tj^T, jb < tj <= ti.
        //     Wave 0 takes the next pivot tile first and factors it while the other waves do the rest.
        const int rem = NB16 - 1 - jb;
        const int ntile = rem * (rem + 1) / 2;
        const int first = (wave == 0) ? 0 : wave;
        const int step = (wave == 0) ? ntile : (PD_WAVES - 1);
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
        long long tq0 = __builtin_readcyclecounter();
        if (wave == 0) pivot_block_16(S, T, jb + 1, lane, invd, info, info_col0);
        long long tq1 = __builtin_readcyclecounter();
        if (tid == 0) dbg[pass * 8 + jb] = tq1 - tq0;
        __syncthreads();
    }
    }

    {   // lower triangle back to global; pairs that straddle the diagonal keep the caller's upper entry
        constexpr int NLD = PD_NB * PD_NB / 2 / PD_THREADS;
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int idx = tid + q * PD_THREADS;
            const int r = idx / (PD_NB / 2), c2 = (idx % (PD_NB / 2)) * 2;
            if (c2 + 1 <= r) {
                *reinterpret_cast<f64x2 *>(A + (int64_t)r * lda + c2) = *reinterpret_cast<const f64x2 *>(&S[r][c2]);
            } else if (c2 == r) {
                A[(int64_t)r * lda + c2] = S[r][c2];
            }
        }
    }
}


int main() {
    const int n = 128; std::vector<double> A(n * n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) A[i * n + j] = (i == j ? n : 0) + 0.5 * cos(i * 0.37 + j * 0.11) * cos(j * 0.37 + i * 0.11);
    double *dA, *dinv; int *dinfo; long long *ddbg;
    hipMalloc(&dA, n * n * 8); hipMalloc(&dinv, 8 * 256 * 8); hipMalloc(&dinfo, 4); hipMalloc(&ddbg, 64 * 8);
    hipMemset(dinfo, 0, 4);
    long long *d2; hipMalloc(&d2, 64 * 8); hipMemcpyToSymbol(HIP_SYMBOL(g_dbg2), &d2, sizeof(d2));
    const size_t shmem = (size_t)(PD_NB * PD_PITCH + 16 * PD_TP) * sizeof(double);
    hipFuncSetAttribute(reinterpret_cast<const void *>(potf2_diag_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    long long h[64];
    for (int rep = 0; rep < 2; rep++) {
        hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(potf2_diag_kernel, dim3(1), dim3(PD_THREADS), shmem, 0, dA, (int64_t)n, dinv, dinfo, (int64_t)0, ddbg);
        hipDeviceSynchronize();
        hipMemcpy(h, ddbg, sizeof(h), hipMemcpyDeviceToHost);
        long long h2[16]; hipMemcpy(h2, d2, sizeof(h2), hipMemcpyDeviceToHost); printf("inner (columns, stores):"); for (int j = 0; j < 8; j++) printf(" (%lld,%lld)", h2[2*j], h2[2*j+1]); printf("\n");
        for (int p = 0; p < 3; p++) { printf("launch %d pass %d pivot durations:", rep, p); for (int j = 0; j < 7; j++) printf(" %lld", h[p * 8 + j]); printf("\n"); }
    }
    return 0;
}
