// edge_stress.hip -- cross-XCD stress test of the flag-edge protocol (EdgeSig, ../common.hpp), VERDICT r3 item 6.
// TEST AID: built into gptools_amd/csrc/build/libedge_stress.so by `make edge_stress`; not linked into libgpt_hip.so.
//
// The protocol under test is the product's own (edge_signal / edge_poll of common.hpp, the same store / load forms as gemm.hip,
// potrf.hip): a PRODUCER kernel writes a payload and raises a flag word from its last workgroup; a CONSUMER kernel that is
// already running polls the word and then reads the payload.  On MI355X the eight XCDs have private, mutually non-coherent L2s,
// so what must be shown is that a consumer on ANOTHER XCD, whose L2 holds STALE lines of the payload, still reads the new
// values.  Per iteration:
//   consumer (stream masked to the CUs of XCD B):  1. reads the whole payload with plain loads (its L2 now caches the OLD
//       values), 2. raises `ready`, 3. polls the flag (edge_poll, bounded), 4. reads the payload in the form under test and
//       counts the words that are not the new value;
//   producer (stream masked to XCD A): waits for `ready`, writes the new payload in the form under test, edge_signal.
// Forms (mode):
//   0  producer: agent-scope (sc1, write-through) stores + vmcnt(0) + flag    consumer: agent-scope loads              [product: GEMM waiter, TRSM consumers]
//   1  producer: as 0                                                          consumer: acquire fence, plain loads    [product: first diagonal-block kernel]
//   2  producer: as 0                                                          consumer: plain loads, NO acquire       [negative control: must show stale reads]
//   3  producer: PLAIN stores + flag (no write-through)                        consumer: agent-scope loads              [negative control]
//   4  producer: as 0         consumer = one-wave wait kernel (step 3) and a SEPARATE reader kernel behind it on the same stream,
//      plain loads: the kernel boundary's own acquire                                                                     [product: stream-side wait]
#include "../common.hpp"
#include <stdlib.h>
#include <vector>

__global__ void xcc_probe_kernel(unsigned *out)
{
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[blockIdx.x] = xcc & 0xf;
}

// The first `nwork` workgroups of the launch that find themselves on XCD `want_xcc` take part (slot 0 .. nwork - 1); everybody
// else leaves at once.  `claim` is zeroed by the host in front of every launch.
__device__ __forceinline__ int64_t claim_slot(unsigned *claim, int want_xcc, int nwork, unsigned *xcc_out)
{
    __shared__ int slot_s;
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xf;
        int sl = -1;
        if ((int)xcc == want_xcc) {
            const unsigned c = atomicAdd(claim, 1u);
            if ((int)c < nwork) {
                sl = (int)c;
                atomicOr(xcc_out, 1u << xcc);
            }
        }
        slot_s = sl;
    }
    __syncthreads();
    return slot_s;
}

__device__ __forceinline__ double payload_value(unsigned it, int64_t i) { return (double)it * 65536.0 + (double)(i & 0xffff); }

__global__ __launch_bounds__(256) void producer_kernel(double *__restrict__ P, int64_t n, unsigned it, int mode, unsigned *flag,
                                                       const unsigned *ready, unsigned *xcc_out, unsigned *err, int want_xcc,
                                                       unsigned *claim, int nwork)
{
    const int64_t wb = claim_slot(claim, want_xcc, nwork, xcc_out), nw = nwork;      // (placement: see edge_stress_run)
    if (wb < 0) return;
    if (threadIdx.x == 0) edge_poll<2, false>(ready, it, err);   // the consumer has cached the old payload and is spinning
    __syncthreads();
    for (int64_t i = wb * 256 + threadIdx.x; i < n; i += nw * 256) {
        const double v = payload_value(it, i);
        if (mode == 3) P[i] = v;
        else __hip_atomic_store(&P[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    edge_signal(flag, it, (unsigned)nw);                         // vmcnt(0), barrier, last workgroup raises the word
}

__global__ __launch_bounds__(256) void consumer_kernel(const double *__restrict__ P, int64_t n, unsigned it, int mode,
                                                       const unsigned *flag, unsigned *ready, unsigned *ready_count,
                                                       unsigned long long *bad, double *sink, unsigned *xcc_out, int stage, unsigned *err, int want_xcc,
                                                       unsigned *claim, int nwork)
{
    // stage 0: whole consumer; 1: pre-warm + ready only (mode 4, kernel 1); 2: verify only (mode 4, kernel 3)
    const int64_t wb = claim_slot(claim, want_xcc, nwork, xcc_out), nw = nwork;
    if (wb < 0) return;
    if (stage != 2) {
        double s = 0.0;
        for (int64_t i = wb * 256 + threadIdx.x; i < n; i += nw * 256) s += P[i];   // plain: into this XCD's L2
        if (s == 1.2345e300) sink[0] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned done = __hip_atomic_fetch_add(ready_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            if (done == (unsigned)nw) {
                __hip_atomic_store(ready_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ready, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        if (stage == 1) return;
        if (threadIdx.x == 0) {
            if (mode == 1) edge_poll<2, true>(flag, it, err);        // + agent-scope acquire (buffer_inv sc1)
            else edge_poll<2, false>(flag, it, err);
        }
        __syncthreads();
    }
    unsigned long long nb = 0;
    for (int64_t i = wb * 256 + threadIdx.x; i < n; i += nw * 256) {
        const double v = (mode == 0 || mode == 3) ? __hip_atomic_load(&P[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : P[i];
        nb += (v != payload_value(it, i));
    }
    if (nb) atomicAdd(bad, nb);
}

__global__ void stress_wait_kernel(const unsigned *word, unsigned value, unsigned *err)
{
    if (threadIdx.x == 0) edge_poll<4, false>(word, value, err);
}

// Placement.  CU masks cannot pin a stream to one XCD on this GPU (measured: every 32-CU mask, contiguous or strided, sees all
// eight XCC ids -- the mask is applied inside every XCD), and a small launch is not dealt round robin either (64 workgroups
// all reported XCC 7).  So both kernels are launched LARGE (2048 workgroups, spread over all XCDs), every workgroup reads its
// own XCC id, and only the first `wgs` that find themselves on the wanted XCD work (producer: XCD 0, consumer: XCD 1; a claim
// counter zeroed in front of every launch); the XCC ids of the working workgroups are reported back.
extern "C" int edge_stress_run(int mode, int iters, long long n_doubles, int wgs, long long *out)
{
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    hipStream_t sp = nullptr, sc = nullptr;
    if (hipStreamCreateWithFlags(&sp, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&sc, hipStreamNonBlocking) != hipSuccess)
        return -3;
    double *P = nullptr, *sink = nullptr;
    unsigned *words = nullptr;                                  // [0,1] flag + counter, [16] ready, [17] ready counter, [32] xcc p, [33] xcc c
    unsigned long long *bad = nullptr;
    if (hipMalloc(&P, (size_t)n_doubles * 8) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess || hipMalloc(&words, 256) != hipSuccess ||
        hipMalloc(&bad, 8) != hipSuccess)
        return -6;
    hipMemsetAsync(P, 0, (size_t)n_doubles * 8, sp);
    hipMemsetAsync(words, 0, 256, sp);
    hipStreamSynchronize(sp);
    long long bad_iters = 0, bad_words = 0;
    for (int it = 1; it <= iters; it++) {
        hipMemsetAsync(bad, 0, 8, sc);
        hipMemsetAsync(words + 56, 0, 16, sc);                     // claim counters: [56] consumer kernel 1, [57] consumer kernel 2, [58] producer
        hipStreamSynchronize(sc);
        if (mode == 4) {
            hipLaunchKernelGGL(consumer_kernel, dim3(2048), dim3(256), 0, sc, P, (int64_t)n_doubles, (unsigned)it, mode, words, words + 16, words + 17, bad,
                               sink, words + 33, 1, words + 48, 1, words + 56, wgs);
            hipLaunchKernelGGL(stress_wait_kernel, dim3(1), dim3(64), 0, sc, words, (unsigned)it, words + 48);
            hipLaunchKernelGGL(consumer_kernel, dim3(2048), dim3(256), 0, sc, P, (int64_t)n_doubles, (unsigned)it, mode, words, words + 16, words + 17, bad,
                               sink, words + 33, 2, words + 48, 1, words + 57, wgs);
        } else {
            hipLaunchKernelGGL(consumer_kernel, dim3(2048), dim3(256), 0, sc, P, (int64_t)n_doubles, (unsigned)it, mode, words, words + 16, words + 17, bad,
                               sink, words + 33, 0, words + 48, 1, words + 56, wgs);
        }
        hipLaunchKernelGGL(producer_kernel, dim3(2048), dim3(256), 0, sp, P, (int64_t)n_doubles, (unsigned)it, mode, words, words + 16, words + 32, words + 48, 0, words + 58, wgs);
        unsigned long long hb = 0;
        if (hipStreamSynchronize(sp) != hipSuccess || hipMemcpyAsync(&hb, bad, 8, hipMemcpyDeviceToHost, sc) != hipSuccess ||
            hipStreamSynchronize(sc) != hipSuccess)
            return -7;
        bad_iters += hb != 0;
        bad_words += (long long)hb;
    }
    unsigned hx[2] = {0, 0}, herr = 0;
    hipMemcpy(hx, words + 32, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&herr, words + 48, 4, hipMemcpyDeviceToHost);
    out[4] = herr;                                              // a bounded wait timed out (the two kernels were not co-resident)
    out[0] = hx[0];
    out[1] = hx[1];
    out[2] = bad_iters;
    out[3] = bad_words;
    hipFree(P);
    hipFree(sink);
    hipFree(words);
    hipFree(bad);
    hipStreamDestroy(sp);
    hipStreamDestroy(sc);
    return 0;
}
