// Phase timing of the PRODUCT diagonal-block kernel (gptools_amd/csrc/potrf.hip compiled with -DGPT_PD_STAMPS):
// s_memtime stamps of wave 0 at every phase boundary of potf2_body, printed as durations in shader cycles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGPT_PD_STAMPS -o scratch/potf2_stamps scratch/potf2_stamps.hip
#include <cstdio>
#include <cstdarg>
#include <cmath>
#include <vector>
#include "../gptools_amd/csrc/potrf.hip"
void gpt_set_error(const char *, ...) {}
void gpt_jitter(hipStream_t) {}
int main()
{
    const int n = 128;
    std::vector<double> A(n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) A[i * n + j] = (i == j ? n : 0) + 0.5 * cos(i * 0.37 + j * 0.11) * cos(j * 0.37 + i * 0.11);
    double *dA, *dws;
    int *dinfo;
    long long *dst;
    hipMalloc(&dA, n * n * 8);
    hipMalloc(&dws, GPT_WS_BLOCK * 8);
    hipMalloc(&dinfo, 4);
    hipMalloc(&dst, 128 * 8);
    hipMemset(dinfo, 0, 4);
    hipMemcpyToSymbol(HIP_SYMBOL(g_pd_stamps), &dst, sizeof(dst));
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipMemcpyAsync(dA, A.data(), n * n * 8, hipMemcpyHostToDevice, st);
        hipMemsetAsync(dst, 0, 128 * 8, st);
        hipEventRecord(e0, st);
        launch_potf2_diag(st, dA, n, dws, dinfo, 0);
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        long long h[128];
        hipMemcpy(h, dst, sizeof(h), hipMemcpyDeviceToHost);
        printf("rep %d: event time %.1f us; stage+pivot0 -> barrier %lld, loop total %lld cycles (x / 2400 = us at 2.4 GHz)\n", rep,
               ms * 1e3, h[1] - h[0], h[2] - h[1]);
        printf("  jb: strip  barrier  tile-upd  pivot  barrier   (sum)\n");
        for (int jb = 0; jb < 8; jb++) {
            const long long *q = h + 8 + jb * 8;
            if (jb < 7) {
                const long long *qn = h + 8 + (jb + 1) * 8;       // the pivot of block jb+1 stamps into ITS slots 6, 7
                printf("  %d: %6lld %6lld %6lld %6lld %6lld   %6lld   pivot = load %lld + columns %lld + store %lld\n", jb, q[1] - q[0],
                       q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[5] - q[0], qn[6] - q[3], qn[7] - qn[6], q[4] - qn[7]);
            }
            else
                printf("  %d: %6lld %6lld (last: stores only) %6lld\n", jb, q[1] - q[0], q[2] - q[1], q[5] - q[2]);
        }
    }
    {
        long long h[128];
        hipMemcpy(h, dst, sizeof(h), hipMemcpyDeviceToHost);
        printf("wave 1, step jb = 0: start-of-trailing %lld after wave 0's barrier stamp; iterations (2 tiles each):", h[80] - h[8 + 2]);
        for (int i = 81; i < 86 && h[i]; i++) printf(" %lld", h[i] - h[i - 1]);
        printf("\n  step jb = 0, end of phase (c) per wave, cycles after barrier:");
        for (int w = 0; w < 8; w++) printf(" w%d %lld", w, h[86 + w] - h[8 + 2]);
        printf("\n");
    }
    {   // ---- the 256-column leaf kernel on a (256 + m) x 256 panel, m = 1024
        const int m = 1024, n2 = 256 + m, ld = 256;
        std::vector<double> P((size_t)n2 * ld);
        for (int i = 0; i < n2; i++)
            for (int j = 0; j < 256; j++) P[(size_t)i * ld + j] = (i == j ? 300.0 : 0) + 0.5 * cos(i * 0.37 + j * 0.11) * cos(j * 0.37 + i * 0.11);
        double *dP, *dws2, *dl10;
        unsigned *dflag;
        hipMalloc(&dP, P.size() * 8);
        hipMalloc(&dws2, 2 * GPT_WS_BLOCK * 8);
        hipMalloc(&dl10, 16384 * 8);
        hipMalloc(&dflag, 64);
        hipMemsetAsync(dflag, 0, 64, st);
        for (int rep = 0; rep < 3; rep++) {
            hipMemcpyAsync(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice, st);
            hipMemsetAsync(dst, 0, 128 * 8, st);
            hipEventRecord(e0, st);
            launch_potf2x2_trsm(st, dP, ld, dws2, dinfo, 0, m, dl10, dflag, 32u * (rep + 1), nullptr);
            hipEventRecord(e1, st);
            hipStreamSynchronize(st);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            long long h[128];
            hipMemcpy(h, dst, sizeof(h), hipMemcpyDeviceToHost);
            printf("potf2x2 rep %d: event %.1f us | WG0 cycles after first body: load A10 %lld, TRSM %lld, publish+S %lld, barrier %lld, SYRK %lld, "
                   "barrier+S write %lld, second body %lld\n", rep, ms * 1e3, h[73] - h[72], h[74] - h[73], h[75] - h[74], h[76] - h[75],
                   h[77] - h[76], h[78] - h[77], h[79] - h[78]);
        }
    }
    {   // ---- the fused diagonal-block + TRSM kernel (workgroup 0 publishes, PUBLISH = true) on a (128 + m) x 128 panel
        for (int m : {1024, 3072}) {
            const int n2 = 128 + m, ld = 128;
            std::vector<double> P((size_t)n2 * ld);
            for (int i = 0; i < n2; i++)
                for (int j = 0; j < 128; j++) P[(size_t)i * ld + j] = (i == j ? 300.0 : 0) + 0.5 * cos(i * 0.37 + j * 0.11) * cos(j * 0.37 + i * 0.11);
            double *dP, *dws2;
            unsigned *dflag;
            hipMalloc(&dP, P.size() * 8);
            hipMalloc(&dws2, GPT_WS_BLOCK * 8);
            hipMalloc(&dflag, 64);
            hipMemsetAsync(dflag, 0, 64, st);
            for (int rep = 0; rep < 3; rep++) {
                hipMemcpyAsync(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice, st);
                hipMemsetAsync(dst, 0, 128 * 8, st);
                hipEventRecord(e0, st);
                launch_potf2_trsm(st, dP, ld, dws2, dinfo, 0, m, dflag, 32u * (rep + 1));
                hipEventRecord(e1, st);
                hipStreamSynchronize(st);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                long long h[128];
                hipMemcpy(h, dst, sizeof(h), hipMemcpyDeviceToHost);
                if (rep < 2) continue;
                printf("potf2_trsm m=%d: event %.1f us; workgroup 0: stage+pivot0 -> barrier %lld, loop total %lld cycles\n", m, ms * 1e3,
                       h[1] - h[0], h[2] - h[1]);
                printf("  jb: strip  barrier  tile-upd  pivot  barrier   (sum)\n");
                for (int jb = 0; jb < 8; jb++) {
                    const long long *q = h + 8 + jb * 8;
                    if (jb < 7) printf("  %d: %6lld %6lld %6lld %6lld %6lld   %6lld\n", jb, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[5] - q[0]);
                    else printf("  %d: %6lld %6lld (last: stores only) %6lld\n", jb, q[1] - q[0], q[2] - q[1], q[5] - q[2]);
                }
            }
        }
    }
    int info = -1;
    hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost);
    std::vector<double> L(n * n);
    hipMemcpy(L.data(), dA, n * n * 8, hipMemcpyDeviceToHost);
    // residual check against the input (lower triangle)
    double worst = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) {
            double s = 0;
            for (int k = 0; k <= j; k++) s += L[i * n + k] * L[j * n + k];
            worst = fmax(worst, fabs(s - A[i * n + j]));
        }
    printf("info %d, max |L L^T - A| = %.3e\n", info, worst);
    return 0;
}
