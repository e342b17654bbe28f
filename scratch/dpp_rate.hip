// Issue / latency numbers the diagonal-block chain is built on (one wave, s_memtime around unrolled sequences):
//   hipcc --offload-arch=gfx950 -O3 -o scratch/dpp_rate scratch/dpp_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define REP 8
template <int C> __device__ __forceinline__ void fm(double &acc, double src, double mult)
{
    asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mult), "i"(C));
}
__device__ __forceinline__ void fplain(double &acc, double a, double b) { asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b)); }
__global__ void k(long long *out, double *sink, double seed)
{
    double a[16], x[16];
    for (int i = 0; i < 16; i++) { a[i] = seed + i + threadIdx.x; x[i] = seed * i; }
    double src = seed * 3, m = seed * 5;
    long long t[12];
    int n = 0;
    // (0) 15 independent DPP fmacs, same source, different lanes
    t[n++] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < REP; r++) {
        fm<1>(a[1], src, m); fm<2>(a[2], src, m); fm<3>(a[3], src, m); fm<4>(a[4], src, m); fm<5>(a[5], src, m);
        fm<6>(a[6], src, m); fm<7>(a[7], src, m); fm<8>(a[8], src, m); fm<9>(a[9], src, m); fm<10>(a[10], src, m);
        fm<11>(a[11], src, m); fm<12>(a[12], src, m); fm<13>(a[13], src, m); fm<14>(a[14], src, m); fm<15>(a[15], src, m);
    }
    // (1) pairs sharing the lane (the lock-step body: a and x)
    t[n++] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < REP; r++) {
        fm<1>(a[1], src, m); fm<1>(x[1], src, m); fm<2>(a[2], src, m); fm<2>(x[2], src, m); fm<3>(a[3], src, m); fm<3>(x[3], src, m);
        fm<4>(a[4], src, m); fm<4>(x[4], src, m); fm<5>(a[5], src, m); fm<5>(x[5], src, m); fm<6>(a[6], src, m); fm<6>(x[6], src, m);
        fm<7>(a[7], src, m); fm<7>(x[7], src, m); fm<8>(a[8], src, m);
    }
    // (2) 15 plain (non-DPP) fmacs
    t[n++] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < REP; r++)
#pragma unroll
        for (int c = 1; c < 16; c++) fplain(a[c], src, m);
    // (3) dependent chain of 15 plain fmacs on one register
    t[n++] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < REP; r++)
#pragma unroll
        for (int c = 1; c < 16; c++) fplain(a[0], src, m);
    // (4) dependent chain: rsq -> mul -> fma -> fma (the Newton step), 15 times
    t[n++] = __builtin_amdgcn_s_memtime();
    double d = a[0] * a[0] + 2.0;
#pragma unroll
    for (int r = 0; r < 15; r++) {
        double y0 = __builtin_amdgcn_rsq(d);
        double tt = d * y0;
        double u = fma(-tt, y0, 1.0);
        d = fma(0.5 * y0, u, y0) + 1.5;
        asm volatile("" : "+v"(d));
    }
    // (5) dependent chain: rcp -> fma -> fma, 15 times
    t[n++] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < 15; r++) {
        double y = __builtin_amdgcn_rcp(d);
        double e = fma(-d, y, 1.0);
        d = fma(y, e, y) + 1.5;
        asm volatile("" : "+v"(d));
    }
    // (6) dependent chain of 15 DPP movs (with the two wait states)
    t[n++] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < 15; r++) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(d));
    // (7) 15 dependent DPP fmacs on one accumulator
    t[n++] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < 15; r++) fm<3>(d, src, m);
    // (8) 15 dependent rsq
    t[n++] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < 15; r++) { d = __builtin_amdgcn_rsq(d); asm volatile("" : "+v"(d)); }
    // (9) 15 dependent rcp
    t[n++] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < 15; r++) { d = __builtin_amdgcn_rcp(d); asm volatile("" : "+v"(d)); }
    t[n++] = __builtin_amdgcn_s_memtime();
    double s = d;
    for (int i = 0; i < 16; i++) s += a[i] + x[i];
    sink[threadIdx.x] = s;
    if (threadIdx.x == 0) for (int i = 0; i < n; i++) out[i] = t[i];
}
__global__ void prec(double *out)
{
    // worst relative error of one Newton step on v_rcp_f64 / v_rsq_f64 over a sweep
    double wr = 0, ws = 0, wr0 = 0, ws0 = 0;
    for (int i = 0; i < 4096; i++) {
        double d = 0.37 + 1e-3 * (threadIdx.x * 4096 + i) * 1.0000001;
        double y = __builtin_amdgcn_rcp(d), e = fma(-d, y, 1.0), r = fma(y, e, y);
        double y0 = __builtin_amdgcn_rsq(d), tt = d * y0, u = fma(-tt, y0, 1.0), iv = fma(0.5 * y0, u, y0);
        wr0 = fmax(wr0, fabs(y * d - 1.0));
        ws0 = fmax(ws0, fabs(y0 * y0 * d - 1.0));
        wr = fmax(wr, fabs(fma(r, d, -1.0)));
        ws = fmax(ws, fabs(iv * sqrt(d) - 1.0));
    }
    out[threadIdx.x * 4 + 0] = wr0; out[threadIdx.x * 4 + 1] = wr; out[threadIdx.x * 4 + 2] = ws0; out[threadIdx.x * 4 + 3] = ws;
}
int main()
{
    long long *d_out; double *d_sink, *d_p;
    hipMalloc(&d_out, 16 * 8); hipMalloc(&d_sink, 64 * 8); hipMalloc(&d_p, 64 * 4 * 8);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, d_sink, 1.25);
        hipDeviceSynchronize();
    }
    long long h[16];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[] = {"15 independent DPP fmacs (x8)", "15 DPP fmacs in lane pairs (x8)", "15 plain fmacs (x8)", "15 dependent plain fmacs (x8)",
                           "15 x (rsq, mul, fma, fma+add)", "15 x (rcp, fma, fma+add)", "15 dependent DPP movs (s_nop 1)", "15 dependent DPP fmacs", "15 dependent rsq", "15 dependent rcp"};
    const int cnt[] = {15 * REP, 15 * REP, 15 * REP, 15 * REP, 15, 15, 15, 15, 15, 15};
    for (int i = 0; i < 10; i++) printf("%-36s %6lld cycles = %.1f per item\n", names[i], h[i + 1] - h[i], (double)(h[i + 1] - h[i]) / cnt[i]);
    hipLaunchKernelGGL(prec, dim3(1), dim3(64), 0, 0, d_p);
    double p[256];
    hipMemcpy(p, d_p, sizeof(p), hipMemcpyDeviceToHost);
    double w[4] = {0, 0, 0, 0};
    for (int i = 0; i < 64; i++) for (int j = 0; j < 4; j++) w[j] = fmax(w[j], p[i * 4 + j]);
    printf("relative error: v_rcp_f64 %.3e, after one Newton step %.3e; v_rsq_f64 (y^2 d - 1) %.3e, after the Newton step (vs sqrt) %.3e\n", w[0], w[1], w[2], w[3]);
    return 0;
}
