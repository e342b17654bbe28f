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

__device__ __forceinline__ double payload_value(unsigned it, int64_t i) { return (double)it * 65536.0 + (double)(i & 0xffff); }

__global__ __launch_bounds__(256) void producer_kernel(double *__restrict__ P, int64_t n, unsigned it, int mode, unsigned *flag,
                                                       const unsigned *ready, unsigned *xcc_out, unsigned *err)
{
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (blockIdx.x == 0) *xcc_out = xcc & 0xf;
        edge_poll<2, false>(ready, it, err);                     // the consumer has cached the old payload and is spinning
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = payload_value(it, i);
        if (mode == 3) P[i] = v;
        else __hip_atomic_store(&P[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    edge_signal(flag, it, gridDim.x);                            // vmcnt(0), barrier, last workgroup raises the word
}

__global__ __launch_bounds__(256) void consumer_kernel(const double *__restrict__ P, int64_t n, unsigned it, int mode,
                                                       const unsigned *flag, unsigned *ready, unsigned *ready_count,
                                                       unsigned long long *bad, double *sink, unsigned *xcc_out, int stage, unsigned *err)
{
    // stage 0: whole consumer; 1: pre-warm + ready only (mode 4, kernel 1); 2: verify only (mode 4, kernel 3)
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        *xcc_out = xcc & 0xf;
    }
    if (stage != 2) {
        double s = 0.0;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += P[i];   // plain: into this XCD's L2
        if (s == 1.2345e300) sink[0] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned done = __hip_atomic_fetch_add(ready_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            if (done == gridDim.x) {
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
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = (mode == 0 || mode == 3) ? __hip_atomic_load(&P[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : P[i];
        nb += (v != payload_value(it, i));
    }
    if (nb) atomicAdd(bad, nb);
}

__global__ void stress_wait_kernel(const unsigned *word, unsigned value, unsigned *err)
{
    if (threadIdx.x == 0) edge_poll<4, false>(word, value, err);
}

static hipStream_t masked_stream(int ncu, const std::vector<int> &cus)
{
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (int c : cus) mask[c / 32] |= 1u << (c % 32);
    hipStream_t st = nullptr;
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return st;
}

// Returns 0 on success.  out[0] = XCD of the producer, out[1] = XCD of the consumer, out[2] = iterations with at least one stale
// word, out[3] = stale words in total, out[4] = 1 if a bounded wait timed out (250 ms: the two kernels never ran side by side).
extern "C" int edge_stress_run(int mode, int iters, long long n_doubles, int wgs, long long *out)
{
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    const int ncu = prop.multiProcessorCount;
    if (ncu < 64) return -2;
    // two CU masks that each lie inside ONE XCD, on different XCDs: tried as contiguous runs of ncu / 8 bits and as every 8th bit
    unsigned *d_x = nullptr;
    if (hipMalloc(&d_x, 4096 * sizeof(unsigned)) != hipSuccess) return -3;
    hipStream_t sp = nullptr, sc = nullptr;
    int xp = -1, xc = -1;
    for (int layout = 0; layout < 2 && !sc; layout++)
        for (int g = 0; g < 8 && !sc; g++) {
            std::vector<int> cus;
            for (int i = 0; i < ncu; i++)
                if (layout == 0 ? (i / (ncu / 8) == g) : (i % 8 == g)) cus.push_back(i);
            hipStream_t st = masked_stream(ncu, cus);
            if (!st) continue;
            hipLaunchKernelGGL(xcc_probe_kernel, dim3(1024), dim3(64), 0, st, d_x);
            std::vector<unsigned> h(1024);
            if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(h.data(), d_x, 1024 * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return -4;
            bool one = true;
            for (unsigned v : h) one = one && v == h[0];
            if (getenv("GPT_EDGE_STRESS_DEBUG")) {
                unsigned seen = 0;
                for (unsigned v : h) seen |= 1u << v;
                fprintf(stderr, "edge_stress: CU mask layout %d group %d (%zu CUs) -> XCC ids seen 0x%x\n", layout, g, cus.size(), seen);
            }
            if (one && !sp) { sp = st; xp = (int)h[0]; }
            else if (one && (int)h[0] != xp) { sc = st; xc = (int)h[0]; }
            else hipStreamDestroy(st);
        }
    hipFree(d_x);
    if (!sp || !sc) return -5;                                  // no single-XCD masks found: the caller skips
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
        if (mode == 4) {
            hipLaunchKernelGGL(consumer_kernel, dim3(wgs), dim3(256), 0, sc, P, (int64_t)n_doubles, (unsigned)it, mode, words, words + 16, words + 17, bad,
                               sink, words + 33, 1, words + 48);
            hipLaunchKernelGGL(stress_wait_kernel, dim3(1), dim3(64), 0, sc, words, (unsigned)it, words + 48);
            hipLaunchKernelGGL(consumer_kernel, dim3(wgs), dim3(256), 0, sc, P, (int64_t)n_doubles, (unsigned)it, mode, words, words + 16, words + 17, bad,
                               sink, words + 33, 2, words + 48);
        } else {
            hipLaunchKernelGGL(consumer_kernel, dim3(wgs), dim3(256), 0, sc, P, (int64_t)n_doubles, (unsigned)it, mode, words, words + 16, words + 17, bad,
                               sink, words + 33, 0, words + 48);
        }
        hipLaunchKernelGGL(producer_kernel, dim3(wgs), dim3(256), 0, sp, P, (int64_t)n_doubles, (unsigned)it, mode, words, words + 16, words + 32, words + 48);
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
    (void)xp;
    (void)xc;
    hipFree(P);
    hipFree(sink);
    hipFree(words);
    hipFree(bad);
    hipStreamDestroy(sp);
    hipStreamDestroy(sc);
    return 0;
}
