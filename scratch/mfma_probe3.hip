// Probe 3: fp64 MFMA issue rate by (independent accumulators per wave) x (waves per SIMD), random operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_ __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int NACC>
__global__ __launch_bounds__(256) void rate_kernel(const double* __restrict__ in, double* out, int iters) {
    double4_ acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (double4_){0, 0, 0, 0};
    double a[8], b[8];
    for (int i = 0; i < 8; i++) { a[i] = in[(threadIdx.x * 16 + i) & 4095]; b[i] = in[(threadIdx.x * 16 + 8 + i + blockIdx.x) & 4095]; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[(i + j) % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[(i + j) % NACC], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(const double *in, double *out, int wgs_per_cu) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int iters = 2000, blocks = 256 * wgs_per_cu;
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        rate_kernel<NACC><<<blocks, 256>>>(in, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double flops = 2.0 * 16 * 16 * 4 * 64.0 * iters * blocks * 4;
    printf("accumulators %d, %d waves/SIMD: %.2f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", NACC, wgs_per_cu, flops / best * 1e-9,
           best * 1e-3 * 2.4e9 / (64.0 * iters * wgs_per_cu));
}
int main() {
    double* in; double* out; CK(hipMalloc(&in, 4096 * 8)); CK(hipMalloc(&out, 1 << 24));
    double h[4096];
    for (int i = 0; i < 4096; i++) h[i] = 2.0 * (rand() / (double)RAND_MAX) - 1.0;
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    for (int w = 1; w <= 4; w *= 2) { run<1>(in, out, w); run<2>(in, out, w); run<4>(in, out, w); run<8>(in, out, w); }
    return 0;
}
