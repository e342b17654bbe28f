// The leaf kernel with the fused update of the next 128 columns (potf2_trsm_kernel<.., true>) against leaf kernel + nothing:
// event times on a (128 + m) x 256 panel, and the update checked against the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o scratch/potf2_upd_events scratch/potf2_upd_events.hip
#include <cstdio>
#include <cstdarg>
#include <cmath>
#include <vector>
#include "../gptools_amd/csrc/potrf.hip"
void gpt_set_error(const char *, ...) {}
void gpt_jitter(hipStream_t) {}
#ifndef GPT_PD_STAMPS
__device__ long long *g_pd_stamps;
__device__ long long *g_pd_trace;
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int *dinfo;
    CK(hipMalloc(&dinfo, 4));
    CK(hipMemset(dinfo, 0, 4));
    for (int m : {256, 1024, 3072}) {
        const int n2 = 128 + m, ld = 256;
        std::vector<double> P((size_t)n2 * ld);
        for (int i = 0; i < n2; i++)
            for (int j = 0; j < 256; j++) P[(size_t)i * ld + j] = (i == j ? 300.0 : 0) + 0.5 * cos(i * 0.37 + j * 0.11) * cos(j * 0.37 + i * 0.11);
        double *dP, *dws;
        unsigned *dflag;
        CK(hipMalloc(&dP, P.size() * 8));
        CK(hipMalloc(&dws, GPT_WS_BLOCK * 8));
        CK(hipMalloc(&dflag, 64));
        CK(hipMemsetAsync(dflag, 0, 64, st));
        for (int upd = 0; upd < 2; upd++) {
            float best = 1e9;
            for (int rep = 0; rep < 6; rep++) {
                CK(hipMemcpyAsync(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice, st));
                CK(hipEventRecord(e0, st));
                if (launch_potf2_trsm(st, dP, ld, dws, dinfo, 0, m, dflag, 32u * (upd * 8 + rep + 1), nullptr, EdgeSig(), EdgeSig(),
                                      upd ? dflag + 8 : nullptr, 8u * (rep + 1)) != GPT_OK) return 1;
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            std::vector<double> R(P.size());
            CK(hipMemcpy(R.data(), dP, P.size() * 8, hipMemcpyDeviceToHost));
            double w2 = 0;
            if (upd)
                for (int i = 128; i < n2; i += 5)
                    for (int j = 0; j < 128; j++) {
                        double s = P[(size_t)i * ld + 128 + j];
                        for (int k = 0; k < 128; k++) s -= R[(size_t)i * ld + k] * R[(size_t)(128 + j) * ld + k];
                        w2 = fmax(w2, fabs(s - R[(size_t)i * ld + 128 + j]));
                    }
            printf("m=%d %s: best event %.1f us%s", m, upd ? "leaf + fused update" : "leaf alone", best * 1e3, upd ? "" : "\n");
            if (upd) printf("; max |C - X X1^T - result| over sampled rows %.3e\n", w2);
        }
    }
    return 0;
}
