#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void lat(double* out, long long* cyc, double seed) {
    double v = seed + threadIdx.x * 1e-3;
    long long t0, t1;
    // dependent fma chain
    __builtin_amdgcn_sched_barrier(0); t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 256; i++) v = fma(v, 0.999999, 1e-7);
    asm volatile("" : "+v"(v)); __builtin_amdgcn_sched_barrier(0); t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0);
    if (threadIdx.x == 0) cyc[0] = (t1 - t0);
    double w = v;
    __builtin_amdgcn_sched_barrier(0); t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 64; i++) w = __builtin_amdgcn_rsq(w) + 1.0;   // rsq + add
    asm volatile("" : "+v"(w)); __builtin_amdgcn_sched_barrier(0); t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0);
    if (threadIdx.x == 0) cyc[1] = (t1 - t0);
    float f = (float)w;
    __builtin_amdgcn_sched_barrier(0); t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 64; i++) f = __builtin_amdgcn_rsqf(f) + 1.0f;
    asm volatile("" : "+v"(f)); __builtin_amdgcn_sched_barrier(0); t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0);
    if (threadIdx.x == 0) cyc[2] = (t1 - t0);
    double z = w + f;
    __builtin_amdgcn_sched_barrier(0); t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 64; i++) {
        int lo = __builtin_amdgcn_readlane(__double2loint(z), 3), hi = __builtin_amdgcn_readlane(__double2hiint(z), 3);
        z = z * 0.5 + __hiloint2double(hi, lo) * 0.25;
    }
    asm volatile("" : "+v"(z)); __builtin_amdgcn_sched_barrier(0); t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0);
    if (threadIdx.x == 0) cyc[3] = (t1 - t0);
    double m = z;
    __builtin_amdgcn_sched_barrier(0); t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 64; i++) m = m * 1.0000001;
    asm volatile("" : "+v"(m)); __builtin_amdgcn_sched_barrier(0); t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0);
    if (threadIdx.x == 0) cyc[4] = (t1 - t0);
    float g = f + (float)m;
    __builtin_amdgcn_sched_barrier(0); t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 256; i++) g = fmaf(g, 0.99999f, 1e-6f);
    asm volatile("" : "+v"(g)); __builtin_amdgcn_sched_barrier(0); t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0);
    if (threadIdx.x == 0) cyc[5] = (t1 - t0);
    out[threadIdx.x] = v + w + f + z + m + g;
}
int main() {
    double* o; long long* c; hipMalloc(&o, 512); hipMalloc(&c, 64);
    for (int r = 0; r < 2; r++) { lat<<<1, 64>>>(o, c, 1.0); hipDeviceSynchronize(); }
    long long h[8]; hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
    printf("dependent v_fma_f64: %.1f cyc\nrsq_f64+add_f64: %.1f cyc\nrsq_f32+add_f32: %.1f\nreadlane x2 + 2 dp ops: %.1f\ndependent v_mul_f64: %.1f\ndependent v_fma_f32: %.1f\n",
           h[0] / 256.0, h[1] / 64.0, h[2] / 64.0, h[3] / 64.0, h[4] / 64.0, h[5] / 256.0);
    return 0;
}
