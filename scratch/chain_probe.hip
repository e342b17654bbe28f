// Stand-alone cost of the look-ahead body's chain recurrence (chain_block) and ride-along recurrence (ride_block):
// alone on a CU, with seven waves polling an LDS word beside them, with seven waves doing MFMA tile updates beside them.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o scratch/chain_probe scratch/chain_probe.hip
#include <cstdio>
#include <cstdarg>
#include <cmath>
#include <vector>
#include "../gptools_amd/csrc/potrf.hip"
void gpt_set_error(const char *, ...) {}
void gpt_jitter(hipStream_t) {}
// mode 0: other waves exit; 1: other waves poll an LDS word (no sleep); 2: poll with s_sleep(1); 3: MFMA + LDS tile updates
__global__ __launch_bounds__(512) void probe(const double *A, long long *out, double *sink, int mode)
{
    __shared__ double S[64][PD_PITCH];
    __shared__ double colbuf[512], invbuf[32];
    __shared__ int flags[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    if (tid < 16) flags[tid] = 0;
    for (int i = tid; i < 64 * PD_PITCH; i += 512) (&S[0][0])[i] = 0.001 * (i % 97);
    __syncthreads();
    if (wave == 0) {
        long long t[8];
        double a[16], v[16];
        for (int rep = 0; rep < 3; rep++) {
            for (int c = 0; c < 16; c++) { a[c] = A[fr * 16 + c]; v[c] = 0.01 * (c + lane); }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            t[rep * 2] = __builtin_amdgcn_s_memtime();
            chain_block(a, colbuf, invbuf, &flags[0], rep * 16, fr);
            t[rep * 2 + 1] = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            long long r0 = __builtin_amdgcn_s_memtime();
#ifndef CHAIN_NOPUB
            ride_block(v, &flags[0], colbuf, invbuf, rep * 16, fr, [] {});
#endif
            long long r1 = __builtin_amdgcn_s_memtime();
            if (rep == 2) { t[6] = r0; t[7] = r1; }
            sink[lane] = a[3] + v[5];
        }
        if (lane == 0) for (int i = 0; i < 8; i++) out[i] = t[i];
        __hip_atomic_store(&flags[8], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (mode == 1 || mode == 2) {
        while (lds_peek(&flags[8]) == 0) { if (mode == 2) __builtin_amdgcn_s_sleep(1); }
    } else if (mode == 3) {
        double (*Sp)[PD_PITCH] = S;
        while (lds_peek(&flags[8]) == 0) {
            TileUpd u;
            u.load(Sp, 1 + (wave & 1), 1, 0, fr, fk);
            u.mma();
            u.store(Sp, fr, fk);
        }
    }
}
int main()
{
    std::vector<double> A(256);
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) A[i * 16 + j] = (i == j ? 16 : 0) + 0.5 * cos(i * 0.37 + j * 0.11) * cos(j * 0.37 + i * 0.11);
    double *dA, *dsink; long long *dout;
    hipMalloc(&dA, 256 * 8); hipMalloc(&dsink, 64 * 8); hipMalloc(&dout, 8 * 8);
    hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
    const char *names[] = {"alone", "7 waves polling LDS", "7 waves polling with s_sleep(1)", "7 waves doing tile updates (LDS + MFMA)"};
    for (int mode = 0; mode < 4; mode++) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, dA, dout, dsink, mode);
        if (hipDeviceSynchronize() != hipSuccess) { printf("failed\n"); return 1; }
        long long h[8];
        hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-42s chain_block %lld %lld %lld cycles; ride_block (all columns there) %lld\n", names[mode], h[1] - h[0], h[3] - h[2], h[5] - h[4], h[7] - h[6]);
    }
    return 0;
}
