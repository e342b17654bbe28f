// Probe: v_mfma_f64_16x16x4_f64 issue rate and operand/result layout on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double double4_ __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NACC>
__global__ void rate_kernel(double* out, int iters, double a0, double b0) {
    double4_ acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (double4_){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void layout_kernel(const double* A, const double* B, double* D) {
    // A: 16x4 row-major, B: 4x16 row-major, D: 16x16 row-major
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];
    double b = B[(l >> 4) * 16 + (l & 15)];
    double4_ c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    // layout check
    std::vector<double> A(64), B(64), D(256), R(256, 0.0);
    for (int i = 0; i < 64; i++) { A[i] = 1 + i * 0.37; B[i] = 2 - i * 0.11 + (i % 5); }
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 4; k++) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 2048));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    layout_kernel<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
    double maxerr = 0; for (int i = 0; i < 256; i++) maxerr = fmax(maxerr, fabs(D[i] - R[i]));
    printf("layout check max err %g\n", maxerr);
    // rate
    double* out; CK(hipMalloc(&out, 1 << 26));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int iters = 20000;
    for (int wpb = 1; wpb <= 8; wpb *= 2) {
      for (int variant = 0; variant < 3; variant++) {
        int nacc = variant == 0 ? 1 : (variant == 1 ? 4 : 16);
        int blocks = p.multiProcessorCount * 4 / (wpb >= 4 ? 1 : 1);
        dim3 g(blocks), b(64 * wpb);
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            if (nacc == 1) rate_kernel<1><<<g, b>>>(out, iters, 1.0, 1.0);
            else if (nacc == 4) rate_kernel<4><<<g, b>>>(out, iters, 1.0, 1.0);
            else rate_kernel<16><<<g, b>>>(out, iters, 1.0, 1.0);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double flops = 2.0 * 16 * 16 * 4 * (double)iters * nacc * blocks * wpb;
        printf("waves/block %d blocks %d nacc %2d: %.3f ms  %.2f TFLOP/s\n", wpb, blocks, nacc, ms, flops / ms * 1e-9);
      }
    }
    return 0;
}
