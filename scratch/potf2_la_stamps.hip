// Phase timing of the look-ahead diagonal-block body (potf2_body_la, gptools_amd/csrc/potrf.hip, -DGPT_PD_STAMPS):
// stamps of the chain wave per 16-column step and of ride-along wave 1; residual and workspace checks; the lock-step body
// beside it (GPT_POTF2_LA=0 in the environment selects it: run the binary twice).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGPT_PD_STAMPS -o scratch/potf2_la_stamps scratch/potf2_la_stamps.hip
#include <cstdio>
#include <cstdarg>
#include <cmath>
#include <vector>
#include "../gptools_amd/csrc/potrf.hip"
void gpt_set_error(const char *, ...) {}
void gpt_jitter(hipStream_t) {}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main()
{
    const int n = 128;
    const bool la = potf2_lookahead();
    std::vector<double> A(n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) A[i * n + j] = (i == j ? n : 0) + 0.5 * cos(i * 0.37 + j * 0.11) * cos(j * 0.37 + i * 0.11);
    double *dA, *dws;
    int *dinfo;
    long long *dst;
    CK(hipMalloc(&dA, n * n * 8));
    CK(hipMalloc(&dws, GPT_WS_BLOCK * 8));
    CK(hipMalloc(&dinfo, 4));
    CK(hipMalloc(&dst, 128 * 8));
    CK(hipMemset(dinfo, 0, 4));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_pd_stamps), &dst, sizeof(dst)));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    long long *dtr;
    CK(hipMalloc(&dtr, 16 * 256 * 2 * 8));
    CK(hipMemset(dtr, 0, 16 * 256 * 2 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_pd_trace), &dtr, sizeof(dtr)));
    printf("body: %s\n", la ? "look-ahead (potf2_body_la)" : "lock-step (potf2_body)");
    for (int rep = 0; rep < 4; rep++) {
        CK(hipMemcpyAsync(dA, A.data(), n * n * 8, hipMemcpyHostToDevice, st));
        CK(hipMemsetAsync(dst, 0, 128 * 8, st));
        CK(hipMemsetAsync(dws, 0xff, GPT_WS_BLOCK * 8, st));
        CK(hipEventRecord(e0, st));
        if (launch_potf2_diag(st, dA, n, dws, dinfo, 0) != GPT_OK) { printf("launch failed\n"); return 1; }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        long long h[128];
        CK(hipMemcpy(h, dst, sizeof(h), hipMemcpyDeviceToHost));
        if (!la) { printf("rep %d: event %.1f us, loop %lld cycles\n", rep, ms * 1e3, h[2] - h[1]); continue; }
#ifdef GPT_PD_TRACE
        if (rep == 3) {
            std::vector<long long> tr(8 * 256 * 2);
            CK(hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost));
            long long t0 = tr[0];
            FILE *f = fopen("gpurun_out/la/trace.txt", "w");
            for (int w = 0; w < 8; w++)
                for (int k = 0; k < 250 && tr[(w * 256 + k) * 2 + 1] != 0; k++)
                    fprintf(f, "%d %lld %lld\n", w, tr[(w * 256 + k) * 2] - t0, tr[(w * 256 + k) * 2 + 1]);
            fclose(f);
        }
#endif
        CK(hipMemset(dtr, 0, 8 * 256 * 2 * 8));
        printf("rep %d: event time %.1f us; chain wave: first stamp -> copy-out done %lld cycles\n", rep, ms * 1e3, h[2] - h[0]);
        if (rep == 0) continue;
        printf("  jb: recurrence  L-store+wait-tiles  wait-X  read+MFMA+store  row-load   (step)   | ride-along waves relative to the chain's step start\n");
        for (int jb = 0; jb < 8; jb++) {
            const long long *q = h + 8 + jb * 8;
            if (jb < 7)
                printf("  %d: %8lld %8lld %8lld %8lld %8lld   %6lld   | w1: ride start %lld, done %lld, stored %lld | w2: %lld %lld %lld\n", jb, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3],
                       q[5] - q[4], q[8] - q[0], h[80 + jb] ? h[80 + jb] - q[0] : 0, h[88 + jb] ? h[88 + jb] - q[0] : 0, h[72 + jb] ? h[72 + jb] - q[0] : 0,
                       h[104 + jb] - q[0], h[112 + jb] - q[0], h[96 + jb] - q[0]);
            else printf("  %d: %8lld (last)\n", jb, q[1] - q[0]);
        }
    }
    int info = -1;
    CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
    std::vector<double> L(n * n), ws(GPT_WS_BLOCK);
    CK(hipMemcpy(L.data(), dA, n * n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ws.data(), dws, GPT_WS_BLOCK * 8, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j <= i; j++) {
            double s = 0;
            for (int k = 0; k <= j; k++) s += L[i * n + k] * L[j * n + k];
            worst = fmax(worst, fabs(s - A[i * n + j]));
        }
    // workspace: inverse block jb, element (r, c) at [jb * 256 + (c >> 2 ... )]: packed element (fr, fk + 4 kk) at [kk * 64 + lane], lane = fk * 16 + fr
    double worst_inv = 0, worst_pk = 0;
    for (int jb = 0; jb < 8; jb++) {
        double M[16][16];
        for (int kk = 0; kk < 4; kk++)
            for (int lane = 0; lane < 64; lane++) M[lane & 15][(lane >> 4) + 4 * kk] = ws[jb * 256 + kk * 64 + lane];
        for (int r = 0; r < 16; r++)
            for (int c = 0; c < 16; c++) {
                double s = 0;
                for (int k = 0; k < 16; k++) s += M[r][k] * ((k >= c) ? L[(jb * 16 + k) * n + jb * 16 + c] : 0.0);
                worst_inv = fmax(worst_inv, fabs(s - (r == c ? 1.0 : 0.0)));
            }
        for (int j = jb + 1; j < 8; j++) {
            const int b = j * (j - 1) / 2 + jb;
            for (int kk = 0; kk < 4; kk++)
                for (int lane = 0; lane < 64; lane++) {
                    const double v = ws[GPT_WS_LOFF + b * 256 + kk * 64 + lane];
                    worst_pk = fmax(worst_pk, fabs(v - L[(j * 16 + (lane & 15)) * n + jb * 16 + (lane >> 4) + 4 * kk]));
                }
        }
    }
    printf("info %d, max |L L^T - A| = %.3e, max |inv(L_jj) L_jj - I| = %.3e, packed blocks vs L: %.3e\n", info, worst, worst_inv, worst_pk);
    // not positive definite: column 37 (1-based 38)
    {
        std::vector<double> B = A;
        B[37 * n + 37] = -1.0;
        CK(hipMemcpyAsync(dA, B.data(), n * n * 8, hipMemcpyHostToDevice, st));
        CK(hipMemsetAsync(dinfo, 0, 4, st));
        if (launch_potf2_diag(st, dA, n, dws, dinfo, 1000) != GPT_OK) return 1;
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
        printf("not positive definite at column 37: info = %d (expected 1038)\n", info);
        CK(hipMemsetAsync(dinfo, 0, 4, st));
    }
    // ---- the fused diagonal-block + TRSM kernel on a (128 + m) x 128 panel
    for (int m : {256, 1024, 3072}) {
        const int n2 = 128 + m, ld = 128;
        std::vector<double> P((size_t)n2 * ld);
        for (int i = 0; i < n2; i++)
            for (int j = 0; j < 128; j++) P[(size_t)i * ld + j] = (i == j ? 300.0 : 0) + 0.5 * cos(i * 0.37 + j * 0.11) * cos(j * 0.37 + i * 0.11);
        double *dP, *dws2;
        unsigned *dflag;
        CK(hipMalloc(&dP, P.size() * 8));
        CK(hipMalloc(&dws2, GPT_WS_BLOCK * 8));
        CK(hipMalloc(&dflag, 64));
        CK(hipMemsetAsync(dflag, 0, 64, st));
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipMemcpyAsync(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice, st));
            CK(hipEventRecord(e0, st));
            if (launch_potf2_trsm(st, dP, ld, dws2, dinfo, 0, m, dflag, 32u * (rep + 1)) != GPT_OK) return 1;
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        std::vector<double> R(P.size());
        CK(hipMemcpy(R.data(), dP, P.size() * 8, hipMemcpyDeviceToHost));
        // X L^T = B for the rows below; L = lower triangle of the first 128 rows
        double w2 = 0;
        for (int i = 128; i < n2; i += 7)
            for (int j = 0; j < 128; j++) {
                double s = 0;
                for (int k = 0; k <= j; k++) s += R[(size_t)i * ld + k] * R[(size_t)j * ld + k];
                w2 = fmax(w2, fabs(s - P[(size_t)i * ld + j]));
            }
        printf("potf2_trsm m=%d: best event %.1f us; max |X L^T - B| over sampled rows %.3e\n", m, best * 1e3, w2);
#ifdef GPT_PD_TRACE
        if (m == 1024) {
            std::vector<long long> tr(16 * 256 * 2);
            CK(hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost));
            long long t0 = tr[0];
            FILE *f = fopen("gpurun_out/la/trace_fused.txt", "w");
            for (int w = 0; w < 16; w++)
                for (int k = 0; k < 250 && tr[(w * 256 + k) * 2 + 1] != 0; k++)
                    fprintf(f, "%d %lld %lld\n", w, tr[(w * 256 + k) * 2] - t0, tr[(w * 256 + k) * 2 + 1]);
            fclose(f);
        }
        CK(hipMemset(dtr, 0, 16 * 256 * 2 * 8));
#endif
    }
    return 0;
}
