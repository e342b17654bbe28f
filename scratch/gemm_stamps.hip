// Activity profile of ONE trailing-update launch (gemm.hip compiled with -DGPT_GEMM_STAMPS): wall-clock stamps of every
// workgroup (start, first k-tile ready, main loop done, stores issued) -> number of workgroups in each phase over time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGPT_GEMM_STAMPS -o scratch/gemm_stamps scratch/gemm_stamps.hip
#include <cstdio>
#include <cstdarg>
#include <cmath>
#include <vector>
#include <algorithm>
#include "../gptools_amd/csrc/gemm.hip"
void gpt_set_error(const char *, ...) {}
void gpt_jitter(hipStream_t) {}
int main(int argc, char **argv)
{
    const int64_t m = argc > 1 ? atoll(argv[1]) : 7168, k = argc > 2 ? atoll(argv[2]) : 384;
    const int reserve = argc > 3 ? atoi(argv[3]) : 32;
    const int64_t ncols = argc > 5 ? atoll(argv[5]) : m;          // narrow update: C is m x ncols (lower trapezoid)
    hipStream_t st;
    {
        hipDeviceProp_t prop;
        hipGetDeviceProperties(&prop, 0);
        const int ncu = prop.multiProcessorCount;
        std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
        for (int i = reserve; i < ncu; i++) mask[i / 32] |= (1u << (i % 32));
        hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
    }
    double *A, *C;
    hipMalloc(&A, m * k * 8);
    hipMalloc(&C, m * m * 8);
    std::vector<double> h((size_t)m * k);
    for (size_t i = 0; i < h.size(); i++) h[i] = sin(0.001 * (double)i) * 0.01;
    hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemset(C, 0, m * m * 8);
    const int64_t ntn = m / 64, nwg_max = ntn * (ntn + 1) / 2 + 4096;
    long long *dst;
    hipMalloc(&dst, nwg_max * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps), &dst, sizeof(dst));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipMemsetAsync(dst, 0, nwg_max * 8 * 8, st);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        launch_gemm_nt(st, m, ncols, k, -1.0, A, k, A, k, (argc > 4 ? atof(argv[4]) : 1.0), C, m, 1, 0, (argc > 6 ? atoi(argv[6]) : 1024), nullptr, nullptr);
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> s((size_t)nwg_max * 8);
        hipMemcpy(s.data(), dst, s.size() * 8, hipMemcpyDeviceToHost);
        long long t0 = -1, t1 = 0;
        int nw = 0;
        for (int64_t b = 0; b < nwg_max; b++)
            if (s[b * 8 + 3]) {
                nw++;
                if (t0 < 0 || s[b * 8] < t0) t0 = s[b * 8];
                t1 = std::max(t1, s[b * 8 + 3]);
            }
        // wall_clock64 ticks at 100 MHz: 10 ns
        const double span = (double)(t1 - t0) * 0.01;
        double sum_issue = 0, sum_wait = 0, sum_bar = 0;
        for (int64_t b = 0; b < nwg_max; b++)
            if (s[b * 8 + 3]) {
                sum_issue += (double)(s[b * 8 + 4] - s[b * 8]) * 0.01;
                sum_wait += (double)(s[b * 8 + 5] - s[b * 8 + 4]) * 0.01;
                sum_bar += (double)(s[b * 8 + 1] - s[b * 8 + 5]) * 0.01;
            }
        double sum_tab = 0, sum_dma = 0, sum_c = 0;
        for (int64_t b = 0; b < nwg_max; b++)
            if (s[b * 8 + 3]) {
                sum_tab += (double)(s[b * 8 + 6] - s[b * 8]) * 0.01;
                sum_dma += (double)(s[b * 8 + 7] - s[b * 8 + 6]) * 0.01;
                sum_c += (double)(s[b * 8 + 4] - s[b * 8 + 7]) * 0.01;
            }
        printf("        issue part: tile-table entry %.2f, addresses + first DMA issue %.2f, C loads + scaling %.2f us\n", sum_tab / nw, sum_dma / nw, sum_c / nw);
        double sum_pro = 0, sum_loop = 0, sum_epi = 0;
        for (int64_t b = 0; b < nwg_max; b++)
            if (s[b * 8 + 3]) {
                sum_pro += (double)(s[b * 8 + 1] - s[b * 8]) * 0.01;
                sum_loop += (double)(s[b * 8 + 2] - s[b * 8 + 1]) * 0.01;
                sum_epi += (double)(s[b * 8 + 3] - s[b * 8 + 2]) * 0.01;
            }
        printf("rep %d: m=%lld k=%lld, %d workgroups, event %.1f us, first start -> last end %.1f us; per workgroup: prologue %.1f, main loop %.1f, "
               "stores %.1f us\n", rep, (long long)m, (long long)k, nw, ms * 1e3, span, sum_pro / nw, sum_loop / nw, sum_epi / nw);
        printf("        prologue of wave 0 = issue (table, addresses, DMA + C loads out) %.1f + wait for them %.1f + barrier (other waves) %.1f us\n",
               sum_issue / nw, sum_wait / nw, sum_bar / nw);
        if (rep == 2) {
            const int NB = 24;
            printf("  time slice (us): workgroups resident (started, not finished) | in main loop\n");
            for (int q = 0; q < NB; q++) {
                const double ta = span * q / NB, tb = span * (q + 1) / NB, tm = 0.5 * (ta + tb);
                int res = 0, loop = 0;
                for (int64_t b = 0; b < nwg_max; b++)
                    if (s[b * 8 + 3]) {
                        const double a0 = (double)(s[b * 8] - t0) * 0.01, a1 = (double)(s[b * 8 + 1] - t0) * 0.01,
                                     a2 = (double)(s[b * 8 + 2] - t0) * 0.01, a3 = (double)(s[b * 8 + 3] - t0) * 0.01;
                        if (a0 <= tm && tm < a3) res++;
                        if (a1 <= tm && tm < a2) loop++;
                    }
                printf("  %6.1f: %4d | %4d\n", tm, res, loop);
            }
        }
    }
    return 0;
}
