// simgate.hip -- timeline gate for scratch/sim_model.py (not part of the product): a kernel that stamps the device's 100 MHz
// wall clock into memory, and a one-wave kernel that holds its stream until that clock has advanced by a given amount.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scratch/libsimgate.so scratch/simgate.hip
#include <hip/hip_runtime.h>
__global__ void stamp_kernel(long long *t0) { *t0 = wall_clock64(); }
__global__ void wait_kernel(const long long *t0, long long delta)
{
    if (threadIdx.x == 0) {
        const long long s = __hip_atomic_load(t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (wall_clock64() - s < delta) __builtin_amdgcn_s_sleep(8);
    }
}
extern "C" int gate_stamp(void *stream, long long *d_t0)
{
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d_t0);
    return (int)hipGetLastError();
}
extern "C" int gate_wait(void *stream, const long long *d_t0, long long delta_ticks)
{
    hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_t0, delta_ticks);
    return (int)hipGetLastError();
}
