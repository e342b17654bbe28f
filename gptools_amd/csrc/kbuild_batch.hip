// kbuild_batch.hip -- the covariance builder for a BATCH of independent hyperparameter vectors over the same resident
// points (gpt_fit_batch, include/gpt_hip.h): one launch builds the lower triangles (+ fused diagonal loading) of all the
// K_tot, element z of the batch from its own KParams in device memory.  Same kernel as the single-matrix path
// (kbuild_kernel.hpp), so an element of a batch holds the very numbers gpt_fit would build alone.
// ref: gptools/gaussian_process.py:1607-1692 (compute_ll_matrix) and :723-735 (random starts) evaluate the LML at many
// hyperparameter vectors one after another; this is their builder.
#include <string.h>
#include "kbuild_kernel.hpp"

// full != 0: the whole symmetric N x N matrix of every element (the transform path multiplies it by T from both sides), else
// the tiles of its lower triangle.  d_kps2 (product terms): element z's second factor.
template <int KID>
static int kbuild_batch_d(hipStream_t st, int D, const KParams *d_kps, const double *d_nv, int64_t nbatch, const double *dX,
                          const int32_t *dn, int64_t N, const double *d_err_y, double diag_add, double *dK, int64_t ldk,
                          int64_t bstride, int accumulate, int full, const KParams *d_kps2)
{
    const int64_t nrt = (N + KB_ROWS - 1) / KB_ROWS;
    int64_t ntile = 0;
    for (int64_t rt = 0; rt < nrt; rt++) ntile += rt / KB_RATIO + 1;
    dim3 grid((unsigned)ntile, 1, (unsigned)nbatch), block(KB_THREADS);
    if (full) grid = dim3((unsigned)((N + KB_COLS - 1) / KB_COLS), (unsigned)nrt, (unsigned)nbatch);
    const int lower = full ? 0 : 2;
    KParams dummy = KParams();
#define KBB_CASE(DD)                                                                                              \
    case DD:                                                                                                      \
        hipLaunchKernelGGL((kbuild_kernel<KID, DD, true>), grid, block, 0, st, dummy, dX, dn, N, dX, dn, N, lower, \
                           (int64_t)0, (int64_t)0, d_err_y, 0.0, diag_add, dK, ldk, accumulate, d_kps, d_nv, bstride, dummy, d_kps2); \
        break;
    switch (D) {
        KBB_CASE(1) KBB_CASE(2) KBB_CASE(3) KBB_CASE(4) KBB_CASE(5) KBB_CASE(6) KBB_CASE(7) KBB_CASE(8)
        KBB_CASE(9) KBB_CASE(10) KBB_CASE(11) KBB_CASE(12) KBB_CASE(13) KBB_CASE(14) KBB_CASE(15) KBB_CASE(16)
    default:
        gpt_set_error("kbuild_batch: unsupported num_dim %d (max %d)", D, GPT_MAX_DIM);
        return GPT_E_ARG;
    }
#undef KBB_CASE
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

int launch_kbuild_batch(hipStream_t st, int kernel_id, int D, const KParams *d_kps, const double *d_noise_var, int64_t nbatch,
                        const double *dX, const int32_t *dn, int64_t N, const double *d_err_y, double diag_add, double *dK,
                        int64_t ldk, int64_t bstride, int accumulate, int full, const KParams *d_kps2)
{
    if (N <= 0 || nbatch <= 0) return GPT_OK;
    if (d_kps2 != nullptr)           // a product term: the factors' kernel ids are read from the elements' KParams at run time
        return kbuild_batch_d<GPT_KERNEL_PRODUCT>(st, D, d_kps, d_noise_var, nbatch, dX, dn, N, d_err_y, diag_add, dK, ldk, bstride, accumulate,
                                                  full, d_kps2);
    switch (kernel_id) {
    case GPT_KERNEL_SE: return kbuild_batch_d<GPT_KERNEL_SE>(st, D, d_kps, d_noise_var, nbatch, dX, dn, N, d_err_y, diag_add, dK, ldk, bstride, accumulate, full, nullptr);
    case GPT_KERNEL_M52: return kbuild_batch_d<GPT_KERNEL_M52>(st, D, d_kps, d_noise_var, nbatch, dX, dn, N, d_err_y, diag_add, dK, ldk, bstride, accumulate, full, nullptr);
    case GPT_KERNEL_RQ: return kbuild_batch_d<GPT_KERNEL_RQ>(st, D, d_kps, d_noise_var, nbatch, dX, dn, N, d_err_y, diag_add, dK, ldk, bstride, accumulate, full, nullptr);
    case GPT_KERNEL_MATERN: return kbuild_batch_d<GPT_KERNEL_MATERN>(st, D, d_kps, d_noise_var, nbatch, dX, dn, N, d_err_y, diag_add, dK, ldk, bstride, accumulate, full, nullptr);
    default:
        gpt_set_error("kbuild_batch: kernel_id %d is not a fit kernel", kernel_id);
        return GPT_E_ARG;
    }
}
