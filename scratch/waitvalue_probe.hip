// Does hipStreamWaitValue32 work on plain device memory, and what does the edge cost?  Stream P: kernel A (spins ~20 us,
// then its last workgroup stores the flag), kernel C right behind it.  Stream S: waitValue(flag >= epoch), kernel B.
// Compared with the event edge: A launched with a stop event (hipExtLaunchKernelGGL), S waits for the event.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void work(long long *stamp, int slot, unsigned *flag, unsigned val, unsigned *count, long long spin)
{
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[slot * 2] = t0;
    while (wall_clock64() - t0 < spin) { }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (flag) {
            __threadfence();
            const unsigned done = atomicAdd(count, 1u) + 1u;
            if (done == gridDim.x) {
                atomicExch(count, 0u);
                __hip_atomic_store(flag, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        if (blockIdx.x == 0) stamp[slot * 2 + 1] = wall_clock64();
    }
}
int main()
{
    hipStream_t P, S;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&P, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
    long long *stamp, h[16];
    unsigned *flag, *count;
    CK(hipMalloc(&stamp, 16 * 8));
    CK(hipMalloc(&flag, 64));
    CK(hipMemset(flag, 0, 64));
    count = flag + 8;
    hipEvent_t ev;
    CK(hipEventCreate(&ev));
    const long long spin = 2000;     // wall_clock64 ticks at 100 MHz: 20 us
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 4; rep++) {
            const unsigned val = 10u * mode + rep + 1u;
            if (mode == 0) {            // event edge, stop event on the kernel
                hipExtLaunchKernelGGL(work, dim3(32), dim3(256), 0, P, nullptr, ev, 0, stamp, 0, (unsigned *)nullptr, 0u, count, spin);
                hipLaunchKernelGGL(work, dim3(32), dim3(256), 0, P, stamp, 2, (unsigned *)nullptr, 0u, count, 100LL);
                CK(hipStreamWaitEvent(S, ev, 0));
            } else if (mode == 1) {     // flag + hipStreamWaitValue32
                hipLaunchKernelGGL(work, dim3(32), dim3(256), 0, P, stamp, 0, flag, val, count, spin);
                hipLaunchKernelGGL(work, dim3(32), dim3(256), 0, P, stamp, 2, (unsigned *)nullptr, 0u, count, 100LL);
                hipError_t e = hipStreamWaitValue32(S, flag, val, hipStreamWaitValueGte, 0xffffffffu);
                if (e != hipSuccess) { printf("hipStreamWaitValue32 -> %s\n", hipGetErrorString(e)); return 2; }
            } else {                    // no edge on the kernel at all (reference for the P -> P gap)
                hipLaunchKernelGGL(work, dim3(32), dim3(256), 0, P, stamp, 0, (unsigned *)nullptr, 0u, count, spin);
                hipLaunchKernelGGL(work, dim3(32), dim3(256), 0, P, stamp, 2, (unsigned *)nullptr, 0u, count, 100LL);
            }
            hipLaunchKernelGGL(work, dim3(32), dim3(256), 0, S, stamp, 1, (unsigned *)nullptr, 0u, count, 100LL);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, stamp, sizeof(h), hipMemcpyDeviceToHost));
            if (rep) printf("mode %d (%s): A ran %.1f us; A end -> next kernel on P starts %.1f us; A end -> B on S starts %.1f us\n", mode,
                            mode == 0 ? "stop event + hipStreamWaitEvent" : mode == 1 ? "flag + hipStreamWaitValue32" : "no edge",
                            (h[1] - h[0]) * 0.01, (h[4] - h[1]) * 0.01, (h[2] - h[1]) * 0.01);
        }
    }
    return 0;
}
