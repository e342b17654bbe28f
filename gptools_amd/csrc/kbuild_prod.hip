// kbuild_prod.hip -- covariance builder / pair list for the PRODUCT of two native kernels (GPT_KERNEL_PRODUCT, kpair.hpp
// prod_pair; ref: gptools/kernel/core.py:587-671).  Same kernels as kbuild.hip (kbuild_kernel.hpp), instantiated once per
// num_dim with the factors chosen at run time; a translation unit of its own so that it compiles beside the others.
#include "kbuild_kernel.hpp"

int launch_kbuild_prod(hipStream_t st, const KParams &kp1, const KParams &kp2, const double *dXi, const int32_t *dni, int64_t M,
                       const double *dXj, const int32_t *dnj, int64_t P, int lower_only, int64_t i0, int64_t j0,
                       const double *d_err_y, double noise_var, double diag_add, double *dK, int64_t ldk, int accumulate)
{
    gpt_jitter(st);
    if (M <= 0 || P <= 0) return GPT_OK;
    dim3 grid((unsigned)((P + KB_COLS - 1) / KB_COLS), (unsigned)((M + KB_ROWS - 1) / KB_ROWS));
    dim3 block(KB_THREADS);
    if (lower_only && i0 == j0 && M == P) {
        const int64_t nrt = (M + KB_ROWS - 1) / KB_ROWS;
        int64_t ntile = 0;
        for (int64_t rt = 0; rt < nrt; rt++) ntile += rt / KB_RATIO + 1;
        grid = dim3((unsigned)ntile, 1);
        lower_only = 2;
    }
#define KBP_CASE(DD)                                                                                                  \
    case DD:                                                                                                          \
        hipLaunchKernelGGL((kbuild_kernel<GPT_KERNEL_PRODUCT, DD, false>), grid, block, 0, st, kp1, dXi, dni, M, dXj,  \
                           dnj, P, lower_only, i0, j0, d_err_y, noise_var, diag_add, dK, ldk, accumulate,              \
                           (const KParams *)nullptr, (const double *)nullptr, (int64_t)0, kp2, (const KParams *)nullptr);  \
        break;
    switch (kp1.D) {
        KBP_CASE(1) KBP_CASE(2) KBP_CASE(3) KBP_CASE(4) KBP_CASE(5) KBP_CASE(6) KBP_CASE(7) KBP_CASE(8)
        KBP_CASE(9) KBP_CASE(10) KBP_CASE(11) KBP_CASE(12) KBP_CASE(13) KBP_CASE(14) KBP_CASE(15) KBP_CASE(16)
    default:
        gpt_set_error("kbuild: unsupported num_dim %d (max %d)", kp1.D, GPT_MAX_DIM);
        return GPT_E_ARG;
    }
#undef KBP_CASE
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

int launch_kpairs_prod(hipStream_t st, const KParams &kp1, const KParams &kp2, const double *dXi, const double *dXj,
                       const int32_t *dni, const int32_t *dnj, int64_t M, double *dout, int accumulate)
{
    if (M <= 0) return GPT_OK;
    dim3 grid((unsigned)((M + 255) / 256)), block(256);
#define KPP_CASE(DD)                                                                                                  \
    case DD:                                                                                                          \
        hipLaunchKernelGGL((kpairs_kernel<GPT_KERNEL_PRODUCT, DD>), grid, block, 0, st, kp1, dXi, dXj, dni, dnj, M,    \
                           dout, accumulate, kp2);                                                                    \
        break;
    switch (kp1.D) {
        KPP_CASE(1) KPP_CASE(2) KPP_CASE(3) KPP_CASE(4) KPP_CASE(5) KPP_CASE(6) KPP_CASE(7) KPP_CASE(8)
        KPP_CASE(9) KPP_CASE(10) KPP_CASE(11) KPP_CASE(12) KPP_CASE(13) KPP_CASE(14) KPP_CASE(15) KPP_CASE(16)
    default:
        gpt_set_error("kpairs: unsupported num_dim %d (max %d)", kp1.D, GPT_MAX_DIM);
        return GPT_E_ARG;
    }
#undef KPP_CASE
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}
