// Probe 2: fp64 MFMA rate with random (high toggle) operands held in registers -- the realistic (DVFS) ceiling.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_ __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ __launch_bounds__(256) void rate_kernel(const double* __restrict__ in, double* out, int iters) {
    double4_ acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (double4_){0, 0, 0, 0};
    double a[8], b[8];
    for (int i = 0; i < 8; i++) { a[i] = in[(threadIdx.x * 16 + i) & 4095]; b[i] = in[(threadIdx.x * 16 + 8 + i + blockIdx.x) & 4095]; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[(i + j) & 7] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[(i + j) & 7], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* in; double* out; CK(hipMalloc(&in, 4096 * 8)); CK(hipMalloc(&out, 1 << 24));
    double h[4096];
    for (int mode = 0; mode < 3; mode++) {
        for (int i = 0; i < 4096; i++) h[i] = mode == 0 ? 1.0 : (mode == 1 ? (rand() / (double)RAND_MAX) : 2.0 * (rand() / (double)RAND_MAX) - 1.0);
        CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        int iters = 4000, blocks = 256 * 2;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            rate_kernel<<<blocks, 256>>>(in, out, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double flops = 2.0 * 16 * 16 * 4 * 64.0 * iters * blocks * 4;
            printf("mode %d (0=ones,1=[0,1),2=[-1,1)) rep %d: %.3f ms %.2f TFLOP/s\n", mode, rep, ms, flops / ms * 1e-9);
        }
    }
    return 0;
}
