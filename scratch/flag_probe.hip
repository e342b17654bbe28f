// flag_probe.hip -- latency of workgroup-to-workgroup signalling through global memory on MI355X (across XCDs).
// WG 0 publishes a payload + flag per step; every other WG polls the flag, then reads and checks the payload.
//   mode 0: payload by plain stores, __threadfence() (agent-scope release) before the flag; consumers fence (acquire) after it
//   mode 1: payload by agent-scope relaxed atomic stores / loads, flag by release / acquire atomics, no full fences
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define PAYLOAD 1152            // doubles per step (one pivot block's share of the packed workspace)
#define NSTEP 64

__device__ __forceinline__ double pay(int s, int i) { return (double)(s * 4096 + i) + 0.5; }

template <int MODE>
__global__ __launch_bounds__(256) void probe(double *buf, unsigned *flag, long long *t_pub, long long *t_seen, long long *t_read,
                                            unsigned *errs, int work_iters)
{
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        for (int s = 0; s < NSTEP; s++) {
            // simulated pivot work
            double x = 1.0 + tid;
            for (int it = 0; it < work_iters; it++) x = fma(x, 1.0000001, 1e-9);
            if (x == 12345.678) buf[0] = x;
            double *dst = buf + (size_t)(s & 1) * PAYLOAD;
            for (int i = tid; i < PAYLOAD; i += 256) {
                if (MODE == 0) dst[i] = pay(s, i);
                else __hip_atomic_store(dst + i, pay(s, i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (MODE == 0) __threadfence();
            __syncthreads();
            if (tid == 0) {
                t_pub[s] = wall_clock64();
                __hip_atomic_store(flag, (unsigned)(s + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            // do not overwrite buffer (s & 1) again before everybody has read step s - 1: wait for the consumers' counter
            if (s >= 1) {
                if (tid == 0) while (__hip_atomic_load(flag + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(s * (gridDim.x - 1))) __builtin_amdgcn_s_sleep(1);
                __syncthreads();
            }
        }
        return;
    }
    unsigned bad = 0;
    for (int s = 0; s < NSTEP; s++) {
        if (tid == 0) {
            while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(s + 1)) __builtin_amdgcn_s_sleep(1);
            t_seen[(size_t)blockIdx.x * NSTEP + s] = wall_clock64();
        }
        __syncthreads();
        if (MODE == 0) __threadfence();
        const double *src = buf + (size_t)(s & 1) * PAYLOAD;
        for (int i = tid; i < PAYLOAD; i += 256) {
            const double v = (MODE == 0) ? src[i] : __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != pay(s, i)) bad++;
        }
        __syncthreads();
        if (tid == 0) {
            t_read[(size_t)blockIdx.x * NSTEP + s] = wall_clock64();
            __hip_atomic_fetch_add(flag + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (bad) atomicAdd(errs, bad);
}

int main(int argc, char **argv)
{
    const int G = argc > 1 ? atoi(argv[1]) : 128;
    const int work = argc > 2 ? atoi(argv[2]) : 2000;
    double *buf; unsigned *flag, *errs; long long *t_pub, *t_seen, *t_read;
    hipMalloc(&buf, 2 * PAYLOAD * sizeof(double));
    hipMalloc(&flag, 64); hipMalloc(&errs, 4);
    hipMalloc(&t_pub, NSTEP * 8); hipMalloc(&t_seen, (size_t)G * NSTEP * 8); hipMalloc(&t_read, (size_t)G * NSTEP * 8);
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipMemset(flag, 0, 64); hipMemset(errs, 0, 4); hipMemset(buf, 0, 2 * PAYLOAD * sizeof(double));
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(G), dim3(256), 0, 0, buf, flag, t_pub, t_seen, t_read, errs, work);
            else hipLaunchKernelGGL(probe<1>, dim3(G), dim3(256), 0, 0, buf, flag, t_pub, t_seen, t_read, errs, work);
            hipEventRecord(e1);
            if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> hp(NSTEP), hs((size_t)G * NSTEP), hr((size_t)G * NSTEP);
            unsigned herr;
            hipMemcpy(hp.data(), t_pub, NSTEP * 8, hipMemcpyDeviceToHost);
            hipMemcpy(hs.data(), t_seen, (size_t)G * NSTEP * 8, hipMemcpyDeviceToHost);
            hipMemcpy(hr.data(), t_read, (size_t)G * NSTEP * 8, hipMemcpyDeviceToHost);
            hipMemcpy(&herr, errs, 4, hipMemcpyDeviceToHost);
            double sum_seen = 0, max_seen = 0, sum_read = 0, max_read = 0; long cnt = 0;
            for (int b = 1; b < G; b++)
                for (int s = 4; s < NSTEP; s++) {
                    const double ds = (hs[(size_t)b * NSTEP + s] - hp[s]) * 0.01, dr = (hr[(size_t)b * NSTEP + s] - hs[(size_t)b * NSTEP + s]) * 0.01;
                    sum_seen += ds; sum_read += dr; cnt++;
                    if (ds > max_seen) max_seen = ds;
                    if (dr > max_read) max_read = dr;
                }
            printf("mode %d G %d: kernel %.1f us (%.2f us/step)  flag publish->seen avg %.2f max %.2f us   payload read avg %.2f max %.2f us   errors %u\n",
                   mode, G, ms * 1e3, ms * 1e3 / NSTEP, sum_seen / cnt, max_seen, sum_read / cnt, max_read, herr);
        }
    }
    return 0;
}
