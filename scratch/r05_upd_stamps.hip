// Wall-clock stamps (100 MHz) of the consumer waves of potf2_trsm_upd_kernel on one leaf: (128 + m) x (128 + m) matrix, the
// leaf at (0, 0), against the separate launches (potf2_trsm + rank-128 GEMM is not linked here: events only for the fused one).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGPT_PU_STAMPS -o scratch/r05_upd_stamps scratch/r05_upd_stamps.hip
#include <cstdio>
#include <cstdarg>
#include <cmath>
#include <vector>
#include "../gptools_amd/csrc/potrf.hip"
void gpt_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }
void gpt_jitter(hipStream_t) {}
int main(int argc, char **argv)
{
    const int m = argc > 1 ? atoi(argv[1]) : 4096;
    const int upd = argc > 2 ? atoi(argv[2]) : 256;
    const int n = 128 + m;
    std::vector<double> A((size_t)n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) A[(size_t)i * n + j] = (i == j ? 2.0 * n : 0) + 0.5 * cos(i * 0.37 + j * 0.11) * cos(j * 0.37 + i * 0.11);
    double *dA, *dws;
    int *dinfo;
    unsigned *dflag;
    long long *dst;
    const int nwg = 1 + m / 64;
    hipMalloc(&dA, A.size() * 8);
    hipMalloc(&dws, GPT_WS_BLOCK * 8);
    hipMalloc(&dinfo, 4);
    hipMalloc(&dflag, 256);
    hipMalloc(&dst, (size_t)nwg * 8 * 32 * 8);
    hipMemset(dinfo, 0, 4);
    hipMemset(dflag, 0, 256);
    hipMemcpyToSymbol(HIP_SYMBOL(g_pu_stamps), &dst, sizeof(dst));
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    unsigned epoch = 0, x1 = 0;
    std::vector<long long> h((size_t)nwg * 8 * 32);
    for (int rep = 0; rep < 4; rep++) {
        hipMemcpyAsync(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice, st);
        hipMemsetAsync(dst, 0, h.size() * 8, st);
        hipStreamSynchronize(st);
        epoch += 32;
        hipEventRecord(e0, st);
        int rc = launch_potf2_trsm_upd(st, dA, n, dws, dinfo, 0, m, dflag, epoch, x1, upd);
        hipEventRecord(e1, st);
        x1 += upd / 16;
        hipStreamSynchronize(st);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("rep %d rc %d: m %d upd %d: event time %.1f us\n", rep, rc, m, upd, ms * 1e3);
        if (rep == 3) break;                     // (the stamps of workgroup 0 kept below are the fused launch's)
        epoch += 32;
        hipMemcpyAsync(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice, st);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        launch_potf2_trsm(st, dA, n, dws, dinfo, 0, m, dflag, epoch);
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        hipEventElapsedTime(&ms, e0, e1);
        printf("        potf2_trsm alone (no update): %.1f us\n", ms * 1e3);
    }
    hipMemcpy(h.data(), dst, h.size() * 8, hipMemcpyDeviceToHost);
    // t0 = earliest stamp 0
    long long t0 = 0;
    for (int g = 1; g < nwg; g++)
        for (int w = 0; w < 8; w++) {
            long long v = h[((size_t)g * 8 + w) * 32];
            if (v && (!t0 || v < t0)) t0 = v;
        }
    auto us = [&](long long v) { return v ? (v - t0) * 0.01 : -1.0; };
    {
        const long long *q = &h[0], *q4 = &h[4 * 32];
        printf("wg 0 chain wave: step start / columns done:");
        for (int jb = 0; jb < 8; jb++) printf(" %d: %.2f %.2f |", jb, us(q[1 + 2 * jb]), us(q[2 + 2 * jb]));
        printf("\nwg 0 store wave: step's stores begin / previous step's flag raised:");
        for (int jb = 0; jb < 8; jb++) printf(" %d: %.2f %.2f |", jb, us(q4[1 + 2 * jb]), us(q4[2 + 2 * jb]));
        printf("\n");
    }
    const int show[] = {1, 2, 4, 5, nwg / 2, nwg - 1};
    for (int g : show) {
        if (g < 1 || g >= nwg) continue;
        for (int w : {0, 3, 4, 7}) {
            const long long *q = &h[((size_t)g * 8 + w) * 32];
            printf("wg %3d wave %d %s start %.2f |", g, w, w < 4 ? "strip " : "helper", us(q[0]));
            if (w < 4) for (int s = 0; s < 8; s++) printf(" %d: %.2f |", s, us(q[1 + s]));
            else for (int s = 0; s < 8; s++) printf(" %d: %.2f %.2f %.2f |", s, us(q[1 + 3 * s]), us(q[2 + 3 * s]), us(q[3 + 3 * s]));
            if (w >= 4) printf(" end %.2f", us(q[25]));
            printf("\n");
        }
    }
    printf("(strip: step done (x_j stored, folded); helper: x_s arrived / staged by all / update done; us after the first consumer wave's start)\n");
    // latest end over all helper waves
    double last = 0;
    for (int g = 1; g < nwg; g++)
        for (int w = 4; w < 8; w++) last = fmax(last, us(h[((size_t)g * 8 + w) * 32 + 25]));
    printf("last helper end: %.2f us\n", last);
    return 0;
}
