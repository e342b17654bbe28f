#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../gptools_amd/csrc/common.hpp"
void gpt_set_error(const char*, ...) {}
#define PD_NB 128
#define PD_PITCH 130
#define PD_TP 18
#define TP_WAVES 4
#define TP_SP 18
__global__ __launch_bounds__(64 * TP_WAVES, 2) void trsm_panel_kernel(int64_t m, const double *__restrict__ L,
                                                                      int64_t ldl, const double *__restrict__ invd,
                                                                      double *__restrict__ B, int64_t ldb, long long *dbg)
{
    long long t0 = __builtin_readcyclecounter();
    __shared__ __attribute__((aligned(16))) double Lp[28][4][64];
    __shared__ __attribute__((aligned(16))) double Sc[TP_WAVES][16][TP_SP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    constexpr int NB16 = PD_NB / 16;

    // Everything a wave needs from memory is requested up front (L's 28 strictly-lower blocks for the shared pack,
    // its 8 B tiles, the 8 inv(L_jj) fragments) so the loads overlap instead of queueing behind one another.
    // pack: block (j, c), c < j, index j(j-1)/2 + c; thread (kk = tid / 64, l = tid % 64) owns the B-operand slot
    // [kk][l] = L[j*16 + (l & 15)][c*16 + (l >> 4) + 4 kk] -> contiguous, conflict-free LDS writes.
    double v[28];
    {
        const int pk = tid >> 6;
#pragma unroll
        for (int j = 1; j < NB16; j++)
#pragma unroll
            for (int c = 0; c < j; c++)
                v[j * (j - 1) / 2 + c] = L[(int64_t)(j * 16 + fr) * ldl + c * 16 + fk + 4 * pk];
    }
    const int64_t row0 = ((int64_t)blockIdx.x * TP_WAVES + wave) * 16;
    const bool active = row0 < m;
    f64x4 bt[NB16];
    double dv[NB16][4], xa[NB16][4];
    if (active) {
#pragma unroll
        for (int j = 0; j < NB16; j++) {
#pragma unroll
            for (int r = 0; r < 4; r++) bt[j][r] = B[(row0 + fk + 4 * r) * ldb + j * 16 + fr];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) dv[j][kk] = invd[j * 256 + fr * 16 + fk + 4 * kk];
        }
    }
    {
        const int pk = tid >> 6;
#pragma unroll
        for (int b = 0; b < 28; b++) Lp[b][pk][lane] = v[b];
    }
    __syncthreads();
    long long t1 = __builtin_readcyclecounter();
    if (!active) return;
    double (*X)[TP_SP] = Sc[wave];
#pragma unroll
    for (int j = 0; j < NB16; j++) {
        f64x4 acc = bt[j];
#pragma unroll
        for (int c = 0; c < j; c++) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[c][kk], Lp[j * (j - 1) / 2 + c][kk][lane], acc, 0, 0, 0);
        }
        // accumulator (C layout) -> A operand through the per-wave scratch
#pragma unroll
        for (int r = 0; r < 4; r++) X[fk + 4 * r][fr] = acc[r];
        double av[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) av[kk] = X[fr][fk + 4 * kk];
        f64x4 res = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; kk++) res = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], dv[j][kk], res, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            B[(row0 + fk + 4 * r) * ldb + j * 16 + fr] = res[r];
            X[fk + 4 * r][fr] = res[r];
        }
        if (j + 1 < NB16) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++) xa[j][kk] = -X[fr][fk + 4 * kk];
        }
        if (tid == 0 && blockIdx.x == 0) dbg[2 + j] = __builtin_readcyclecounter() - t0;
    }
    if (tid == 0 && blockIdx.x == 0) { dbg[0] = t1 - t0; dbg[1] = __builtin_readcyclecounter() - t0; }
}


int main() {
    const int m = 7680, ld = 8320;
    double *dL, *dB, *dinv; long long *ddbg;
    hipMalloc(&dL, 128 * ld * 8); hipMalloc(&dB, (size_t)m * ld * 8); hipMalloc(&dinv, 8 * 256 * 8); hipMalloc(&ddbg, 64 * 8);
    std::vector<double> L(128 * ld, 0.0), I(2048, 0.0);
    for (int i = 0; i < 128; i++) { for (int j = 0; j < i; j++) L[i * ld + j] = 0.01 * cos(i + 2 * j); L[i * ld + i] = 1.0; }
    for (int b = 0; b < 8; b++) for (int i = 0; i < 16; i++) I[b * 256 + i * 17] = 1.0;
    hipMemcpy(dL, L.data(), L.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dinv, I.data(), 2048 * 8, hipMemcpyHostToDevice);
    hipMemset(dB, 0, (size_t)m * ld * 8);
    long long h[16];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(trsm_panel_kernel, dim3(m / 64), dim3(256), 0, 0, (int64_t)m, dL, (int64_t)ld, dinv, dB, (int64_t)ld, ddbg);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, ddbg, sizeof(h), hipMemcpyDeviceToHost);
        printf("rep %d: %.1f us; prologue %lld cyc, total %lld cyc; j-steps:", rep, ms * 1e3, h[0], h[1]);
        for (int j = 0; j < 8; j++) printf(" %lld", h[2 + j]);
        printf("\n");
    }
    return 0;
}
