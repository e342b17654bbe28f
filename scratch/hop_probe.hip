// hop_probe.hip -- what does a flag hand-over between two workgroups cost when they sit on the SAME XCD (one L2) and the accesses stop
// at that L2 (sc0: bypass the CU's vector L1 only) instead of going through to memory (sc1, agent scope: what the library uses, since
// its producers and consumers are spread over all eight XCDs)?   hop_probe <peer workgroup index> : workgroup 0 <-> workgroup peer
// (workgroup b of a launch runs on XCD b % 8).  Prints ns per one-way hop for each form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int FORM>
__device__ __forceinline__ unsigned ld(const unsigned *p)
{
    unsigned v;
    if (FORM == 0) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (FORM == 1) asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int FORM>
__device__ __forceinline__ void st(unsigned *p, unsigned v)
{
    if (FORM == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (FORM == 1) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int FORM>
__global__ void pingpong(unsigned *flag, int peer, int rounds, long long *ticks, unsigned *xcc)
{
    if (threadIdx.x != 0) return;
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (blockIdx.x == 0) {
        xcc[0] = x & 0xf;
        const long long t0 = wall_clock64();
        for (int i = 1; i <= rounds; i++) {
            st<FORM>(flag, (unsigned)i);
            int spins = 0;
            while (ld<FORM>(flag + 32) != (unsigned)i && ++spins < (1 << 16)) {}
            if (spins >= (1 << 16)) { __hip_atomic_store(flag + 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ticks[1] = i; break; }
        }
        ticks[0] = wall_clock64() - t0;
    } else if ((int)blockIdx.x == peer) {
        xcc[1] = x & 0xf;
        for (int i = 1; i <= rounds; i++) {
            int spins = 0;
            while (ld<FORM>(flag) != (unsigned)i && ++spins < (1 << 16)) {
                if ((spins & 255) == 0 && __hip_atomic_load(flag + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
            }
            if (spins >= (1 << 16)) return;
            st<FORM>(flag + 32, (unsigned)i);
        }
    }
}
int main(int argc, char **argv)
{
    const int rounds = 2000;
    unsigned *flag, *xcc; long long *ticks;
    hipMalloc(&flag, 4096); hipMalloc(&xcc, 64); hipMalloc(&ticks, 64);
    for (int pi = 1; pi < argc; pi++) {
        const int peer = atoi(argv[pi]);
        for (int form = 0; form < 3; form++) {
            hipMemset(flag, 0, 4096); hipMemset(ticks, 0, 64);
            if (form == 0) hipLaunchKernelGGL(pingpong<0>, dim3(peer + 1), dim3(64), 0, 0, flag, peer, rounds, ticks, xcc);
            if (form == 1) hipLaunchKernelGGL(pingpong<1>, dim3(peer + 1), dim3(64), 0, 0, flag, peer, rounds, ticks, xcc);
            if (form == 2) hipLaunchKernelGGL(pingpong<2>, dim3(peer + 1), dim3(64), 0, 0, flag, peer, rounds, ticks, xcc);
            hipDeviceSynchronize();
            long long t, tt[2]; unsigned hx[2];
            hipMemcpy(tt, ticks, 16, hipMemcpyDeviceToHost); t = tt[0];
            unsigned ab; hipMemcpy(&ab, flag + 64, 4, hipMemcpyDeviceToHost);
            if (ab) { printf("workgroup 0 <-> %d form %d: STALE (never saw round %lld)\n", peer, form, tt[1]); fflush(stdout); continue; } hipMemcpy(hx, xcc, 8, hipMemcpyDeviceToHost);
            printf("workgroup 0 (XCC %u) <-> workgroup %d (XCC %u), %s: %.0f ns per one-way hop\n", hx[0], peer, hx[1],
                   form == 0 ? "agent-scope atomics (sc1)" : form == 1 ? "sc0 loads and stores          " : "sc0 sc1 (system scope)        ",
                   (double)t * 10.0 / (2.0 * rounds));
            fflush(stdout);
        }
    }
    return 0;
}
