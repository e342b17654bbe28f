/*
 * gpt_hip.h -- C ABI of libgpt_hip.so: the MI355X (gfx950) implementation of gptools'
 * covariance-build + Cholesky log-marginal-likelihood hot path.
 *
 * Plain C, no torch / numpy types: pointers, sizes and scalars only.  Matrices are row-major
 * (C order, like the numpy arrays the reference passes around).  Every entry point returns an
 * int status:
 *      0            success
 *     >0            LAPACK convention: the leading minor of that order is not positive
 *                   definite (the Python layer raises numpy.linalg.LinAlgError, which
 *                   GaussianProcess.update_hyperparameters turns into +inf exactly like
 *                   gaussian_process.py:1391-1406)
 *     <0            GPT_E_* below
 * gpt_last_error() returns a thread-local human-readable message for the last failure.
 *
 * Citations "ref: file:line" are into the reference tree (markchil/gptools).
 *
 * There are two layers:
 *   (1) context API, host pointers in / host scalars out, device state stays resident in HBM
 *       between calls -- this is what a ctypes/cffi binding inside gptools would call;
 *   (2) device API (gpt_dev_*), raw device pointers + the context's stream -- used by callers
 *       that own device memory themselves (the one-process-per-GPU block-cyclic Cholesky in
 *       gptools_amd/dist.py passes torch.Tensor.data_ptr() values).
 */
#ifndef GPT_HIP_H_
#define GPT_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPT_OK 0
#define GPT_E_ARG (-1)      /* bad argument / shape                       -> ValueError           */
#define GPT_E_VALUE (-2)    /* Matern52 derivative order > 1              -> ValueError  (ref: kernel/matern.py:545-546) */
#define GPT_E_NOTIMPL (-3)  /* unsupported hyper_deriv                     -> NotImplementedError (ref: kernel/matern.py:543-544) */
#define GPT_E_HIP (-4)      /* HIP runtime error                          -> RuntimeError         */
#define GPT_E_NOMEM (-5)    /* device allocation failed                   -> MemoryError          */
#define GPT_E_STATE (-6)    /* call out of order (e.g. predict before fit) -> RuntimeError         */

/* kernel_id values (ref: kernel/squared_exponential.py:31, kernel/matern.py:468,
 * kernel/noise.py:27, kernel/noise.py:112) */
#define GPT_KERNEL_SE 0
#define GPT_KERNEL_M52 1
#define GPT_KERNEL_DIAGNOISE 2
#define GPT_KERNEL_ZERO 3
#define GPT_KERNEL_RQ 4          /* RationalQuadraticKernel, params [sigma_f, alpha, l_1 .. l_D]
                                  * (ref: gptools/kernel/rational_quadratic.py:30-164 through ChainRuleKernel.__call__,
                                  * kernel/core.py:691-816); derivative orders of a pair may sum to GPT_RQ_MAXORD at most
                                  * (GPT_E_VALUE beyond; the reference has no limit, but walks every set partition of the
                                  * derivative multiset: Bell(12) = 4.2 million per pair); no hyper-parameter derivatives */
#define GPT_RQ_MAXORD 16
#define GPT_KERNEL_MATERN 5      /* MaternKernel (general order nu), params [sigma_f, nu, l_1 .. l_D] (ref: kernel/matern.py:251-465
                                  * through ChainRuleKernel.__call__); same limit on the derivative orders of a pair; nu must
                                  * lie in (0, 60) (GPT_E_VALUE otherwise: beyond it the closed form the device evaluates,
                                  * 2^(1-nu)/Gamma(nu) y^((nu-m)/2) K_(nu-m)(sqrt y), overflows in its factors before the
                                  * product is formed.  The reference accepts any nu and evaluates the same form through
                                  * scipy.special.kv, with the same fate: run here (round 5), MaternKernel(nu = 70) already returns nan for
                                  * a pair 1e-6 apart at l = 0.3 (50: still 1.0), nu = 130 nan / inf for every pair closer than
                                  * 0.02, nu = 160 nan throughout) */

#define GPT_KERNEL_PRODUCT 6     /* k1 * k2 of two of the kernels above (SE, Matern52, RQ, Matern), ref: kernel/core.py:587-671: the product
                                  * rule over the derivative orders of a pair, sum over sub-multi-indices a of prod_slots C(n, a)
                                  * k1^(a) k2^(n - a), evaluated per pair on the device (gpt_kpairs2 / gpt_kbuild2 / gpt_fit_terms);
                                  * combined derivative order of a pair <= GPT_RQ_MAXORD; no hyper-parameter derivatives
                                  * (NotImplementedError in the reference too) */

#define GPT_MAX_DIM 16      /* largest supported num_dim */
#define GPT_WS_BLOCK 9216    /* doubles of factorisation workspace per 128 columns (d_invd arguments) */

typedef struct gpt_ctx gpt_ctx;

int gpt_version(void);
const char *gpt_last_error(void);

/* ---- context ------------------------------------------------------------------------------ */
/* One context = one GPU + one main HIP stream (+ an internal high-priority panel stream used for
 * look-ahead).  stream == NULL: the library creates its own non-blocking stream; otherwise the
 * caller's hipStream_t is used for all work (pass torch.cuda.current_stream().cuda_stream). */
int gpt_ctx_create(int device_id, void *stream, gpt_ctx **out);
int gpt_ctx_destroy(gpt_ctx *ctx);
/* Options (gpt_ctx_set_option):
 *   "nb_outer"     outer block width of the factorisation, multiple of 128; 0 = by size (256 up to n = 5120, 384 up to
 *                  12288, 640 above)
 *   "lookahead"    0/1: factor panel k+1 on the high-priority panel stream while the main stream applies panel k
 *   "purg_rows"    while more rows than this remain the panel stream applies panel k to the columns of panel k+1 itself
 *                  (6144; 0 = the main stream always does)
 *   "panel_prio"   wave priority (0..3) of the panel stream's GEMM main loops (2); "gemm_prio" >= 0 forces one priority
 *                  for every GEMM of the context (the panel-side context of gptools_amd/dist.py)
 *   "fuse_trsm"    panels with at most this many rows under a leaf use the fused diagonal-block + TRSM kernel (8192)
 *   "fuse_rows64"  fused leaves with at most this many rows under them run 64-row consumer workgroups, one substitution
 *                  strip per SIMD (2048 = what fits the CUs reserved for the panel stream; 0 = always 128-row workgroups)
 *   "fuse_rows32", "fuse_rows16"  ... 32-row (2048) / 16-row (0 = never) consumer workgroups below that many rows: two / one strip
 *                  waves per CU -- every strip wave requests the same fragments of the diagonal block and a CU turns out ~32 bytes of
 *                  vector-load requests per cycle (round 5, bit-identical: N = 4096 1.158 -> 1.152 ms, N = 8192 4.346 -> 4.31 ms)
 *   "merge_urgent" 1 (default, with edge_flags): the two trailing updates per panel are one launch with a partial edge flag
 *   "edge_flags"   1 (default): the per-panel dependencies of the look-ahead are flag words in device memory (last workgroup
 *                  of the producer raises it; a bounded in-kernel wait or a one-wave wait kernel on the consumer side)
 *                  instead of events; 0: events.  Used by an evaluation only while it is the only one in flight in the
 *                  process (idle contexts do not count; see gpt_concurrency_hint) on a context that owns its streams; off by
 *                  itself under rocprofv3 counter collection, for n > 12288, and for the rest of the process once a wait has
 *                  timed out (250 ms: the evaluation is then repeated on events)
 *   "head_wait_wgs" the first leaf of a factorisation waits for the K build's head columns inside its own kernel while its launch has
 *                  at most this many workgroups (33 = what fits the reserved CUs), else a one-wave wait kernel in front of it
 *   "merge_min_tiles" smallest merged trailing update, in 64x64 tiles (512: the launch needs an order table)
 *   "purg_rows_flags" "purg_rows" while flag edges and merged launches are in use (0: never -- measured)
 *   "tail_wait"    1 (default since round 6): the main stream's last launch of a panel awaits the NEXT panel's flag at its end instead of
 *                  a wait kernel in front of the next launch (C3 -19 us, C2 -15 us); not while "profile_gemm" times the launches (a
 *                  launch that waits at its end reports the wait as its duration); 0: always the wait kernel
 *   "alpha_invalidate" (measurement aid) the next gpt_get_alpha recomputes alpha
 *   "eager_alpha"  1: every gpt_fit* also enqueues alpha = K_tot^-1 y behind its factorisation (the reference computes alpha in every
 *                  evaluation, gaussian_process.py:1462) and lands it in pinned memory under the call's own synchronisation;
 *                  gpt_get_alpha is then a host copy.  0 (default): alpha on first use (gpt_get_alpha, gpt_predict, gpt_ll_grad)
 *   "defer_pad"    1 (default): an eager evaluation whose N is a multiple of 512 (>= 1024) factors the last 128 columns of the padded
 *                  matrix -- the augmented row and the padding only -- on the main stream beside the substitution of alpha, which
 *                  follows the last real leaf at once (same bits); 0: everything on the panel stream, alpha behind it
 *   "binv_launches" 1 (measurement aid): the 512-wide block inverses by rounds 2-4's recursion over 15 launches instead of the one
 *                  launch of trinv512_kernel (solve.hip); results agree to rounding
 *   "splitk"       gpt_predict with std / cov at few points: the GEMMs of a triangular solve with at most 128 right-hand sides
 *                  and the V V^T of a covariance of at most 256 points, when k >= 1024 and they have fewer 32x32 tiles than this
 *                  (512), are split along k into up to 32 chunks of at least 256, summed in chunk order by a second kernel --
 *                  repeatable bit for bit, rounding differs from the unsplit sum; 0 = never split
 *   "helper_tf"    assumed rate of the helper stream (0.1 TFLOP/s per 24 CUs, 45) that takes a slice of the large trailing
 *                  updates on the reserved CUs when n > "helper_min_n" (12288); 0 = off
 *   "gemm_pad"     bytes of dummy dynamic LDS of the main-stream GEMM (residency cap)
 *   "dev_gemm_pad" the same for the device API's GEMM launches on this context (gpt_dev_gemm_nt, _stair, _gridstair), 0 = none
 *                  (default).  Measured in round 6 (profiles/r06_chain_contention.txt): with 24576 (two workgroups of a trailing
 *                  update per CU instead of five) an isolated m = 16384, k = 512 update keeps its rate (48.7 against 48.9 TFLOP/s)
 *                  and the serial chain beside it runs at 1.43 x its stand-alone time instead of 1.95 x -- but the partitioned
 *                  engines' staircase updates lose 10 % with it and the modelled 8-rank time does not improve: off
 *   "graph"        0/1: replay the factorisation from a captured hipGraph
 *   "timing"       0/1: record per-phase HIP events (gpt_last_timings)
 *   "profile_gemm" 0/1: HIP-event timing of each large GEMM launch (gpt_gemm_profile_read)
 *   "tile"         0 by launch size (default), 32, 64: force the GEMM macro-tile (bit-identical results either way)
 *   "debug_poison" 0/1 (test aid): gpt_ll_grad fills its scratch matrices with NaN before use
 *   "edge_test_stall" 1 (test aid): the next evaluation's first flag is withheld once, so that the bounded wait, the repeat on
 *                  event edges and the switch of the process to event edges can be tested
 *   measured and off by default: "ramp", "inner", "inner_rows", "defer_rows", "late_rows",
 *   "early_rows", "nb_early", "nb_switch_rows", "late_pad", "late_pad_rows".
 *   Removed in round 6 with the code behind them (each lost its A/B twice; NOTES_r04.md / NOTES_r05.md have the numbers, the
 *   history up to round 5's last commit the sources): "fuse_upd" / "fuse_upd_rows", "pair_rows", "leaf256", "tile" = 65 / 128 / 129,
 *   the environment switches GPT_GEMM_LOOP and GPT_GEMM_MIXED.  gpt_ctx_set_option refuses unknown keys.
 * Environment: GPT_RESERVE_CUS (CUs the main stream leaves to the panel stream, default 32), GPT_TILE_ORDER
 * ("rows,cols,mode": supertile shape and deal of the GEMM's XCD-aware tile order, default 64,8,1), GPT_GRAD_TIMING,
 * GPT_EDGE_FLAGS=0 (event edges only: set it for jobs that share one GPU between several processes), GPT_GEMM_SMALL (64x64-tile count under which a GEMM launch uses 32x32 tiles, 512),
 * GPT_JITTER (test aid: random delay kernels in front of every dense launch), GPT_ALPHA_NARROW (measurement aid: gpt_get_alpha
 * by 128-wide substitution steps instead of the 512-wide block inverses), GPT_POTF2_LA=0 (the 128-column diagonal-block kernels
 * with the lock-step body of rounds 1-3 instead of the look-ahead body; results agree to rounding, not bit for bit). */
int gpt_ctx_set_option(gpt_ctx *ctx, const char *key, int64_t value);
int gpt_ctx_synchronize(gpt_ctx *ctx);
void *gpt_ctx_stream(gpt_ctx *ctx);
/* Number of flag edges (common.hpp: EdgeSig -- cross-stream dependencies carried by a word in device memory instead of an
 * event) the context has raised so far; diagnostics / tests. */
int64_t gpt_ctx_edge_count(gpt_ctx *ctx);
/* Announce (delta = +1) / retract (delta = -1) that the caller is about to run evaluations on several contexts of this
 * process concurrently (GaussianProcess.ll_batch: one host thread per context; ref gaussian_process.py:723-735, 1607-1692
 * are the reference's pool.map sites this replaces).  While the count is positive every evaluation keeps its look-ahead on
 * event edges (a kernel spinning on a flag must not share the hardware queues with a second chain).  Optional: without it
 * the library notices the overlap itself (the first two evaluations are serialised, then the process stays on event edges
 * while evaluations keep overlapping and for 100 ms after the last overlap).  Returns the new count. */
int gpt_concurrency_hint(int delta);

/* ---- Kernel.__call__ ---------------------------------------------------------------------- */
/* Replaces  Kernel.__call__(Xi, Xj, ni, nj, hyper_deriv=None, symmetric=False) -> (M,) float64
 *   ref: kernel/core.py:220-257 (contract), kernel/squared_exponential.py:82-174,
 *        kernel/matern.py:512-555 -> kernel/_matern.pyx:14-32 -> kernel/src/matern.c:165-186
 *        (`double matern52(const double*, const double*, const int32_t*, const int32_t*, int32_t,
 *          const double*)`, kernel/include/matern.h:24-26), kernel/noise.py:76-110,123-152.
 * Element-wise pair list: row m of the four (M, D) inputs gives out[m].  params = the kernel's
 * full parameter vector ([sigma_f, l_1..l_D] for SE / Matern52, [sigma_n] for the noise kernels).
 * hyper_deriv = -1 for None.  noise_n: DiagonalNoiseKernel.n (D ints) or NULL (zeros).
 * All pointers are host pointers. */
int gpt_kpairs(gpt_ctx *ctx, int kernel_id, const double *params, int nparams,
               const double *Xi, const double *Xj, const int32_t *ni, const int32_t *nj,
               int64_t M, int D, int hyper_deriv, int symmetric, const int32_t *noise_n,
               double *out);

/* The same pair list for the PRODUCT of two native kernels (ProductKernel.__call__, ref: kernel/core.py:587-671). */
int gpt_kpairs2(gpt_ctx *ctx, int kernel_id1, const double *params1, int nparams1, int kernel_id2, const double *params2,
                int nparams2, const double *Xi, const double *Xj, const int32_t *ni, const int32_t *nj, int64_t M, int D,
                double *out);

/* ---- GaussianProcess.compute_Kij ---------------------------------------------------------- */
/* Replaces  GaussianProcess.compute_Kij(Xi, Xj, ni, nj, noise, hyper_deriv, k) -> (M, P)
 *   ref: gaussian_process.py:1535-1605.  Xj == NULL means Xj = Xi, nj = ni, symmetric = True
 *   (ref :1580-1585).  The (M*P, D) tiled temporaries of ref :1591-1594 are never formed: one
 *   fused kernel evaluates pair (i, j) from X rows held in registers/LDS and writes K[i*P + j].
 * K_out: host (M, P) row-major. */
int gpt_kbuild(gpt_ctx *ctx, int kernel_id, const double *params, int nparams,
               const double *Xi, const int32_t *ni, int64_t M,
               const double *Xj, const int32_t *nj, int64_t P, int D,
               int hyper_deriv, const int32_t *noise_n, double *K_out);

/* compute_Kij for the product of two native kernels (Xj == NULL: symmetric). */
int gpt_kbuild2(gpt_ctx *ctx, int kernel_id1, const double *params1, int nparams1, int kernel_id2, const double *params2,
                int nparams2, const double *Xi, const int32_t *ni, int64_t M, const double *Xj, const int32_t *nj, int64_t P,
                int D, double *K_out);

/* ---- GaussianProcess.add_data (device residency) ------------------------------------------ */
/* Uploads the training inputs the later calls use (ref: gaussian_process.py:376-503 defines the
 * host layout: X (N, D) float64, n (N, D) int).  Invalidates any factorisation. */
int gpt_set_data(gpt_ctx *ctx, const double *X, const int32_t *n, int64_t N, int D);

/* Linear transform of the latent values: observations y = T f(X) + noise (ref: gaussian_process.py:376-503 argument
 * `T` of add_data; :1443-1446, :966-970).  T is (Ny, N) row-major over the N resident points.  With a transform set,
 * gpt_fit expects y, err_y of length Ny and factors K_tot = T (K + noise_var I) T^T + diag(err_y^2) + diag_add I
 * (Ny x Ny; the products are fp64-MFMA GEMMs on the device), and gpt_predict applies T to the training side of
 * Kstar.  gpt_set_data drops the transform; T == NULL removes it. */
int gpt_set_T(gpt_ctx *ctx, const double *T, int64_t Ny);

/* ---- GaussianProcess.compute_K_L_alpha_ll ------------------------------------------------- */
/* Replaces the T-free body of GaussianProcess.compute_K_L_alpha_ll   ref: gaussian_process.py:1428-1467
 *   K_tot = K + noise_var*I + diag(err_y^2) + diag_add*I          (ref :1431-1451; diag_add = diag_factor*eps)
 *   L     = cholesky(K_tot, lower)                                (ref :1452, LAPACK dpotrf)
 *   z     = L^-1 y  ;  ll_data = -1/2 z.z - sum(log L_ii) - N/2 log(2 pi)    (ref :1462-1467)
 * y is the mean-subtracted target (ref :1455-1461); the log-prior (ref :1469) is added by the
 * caller.  K_tot / L stay resident in HBM; alpha (ref :1462) is computed on demand by
 * gpt_get_alpha.  err_y: (N,) host.  Only `params` need change between calls (MAP loop). */
int gpt_fit(gpt_ctx *ctx, int kernel_id, const double *params, int nparams, double noise_var,
            const double *y, const double *err_y, double diag_add,
            double *ll_data_out, double *logdet_half_out);

/* The same for a model kernel that is a SUM of native kernels (SumKernel, ref: kernel/core.py:549-584, k1 + k2 + ...):
 * nterms <= 8 kernel ids, their parameter vectors concatenated in `params` (nparams[t] entries each).  One builder pass
 * per term, accumulated on the device; everything else as gpt_fit (which is the nterms == 1 case).  gpt_predict
 * afterwards uses the same sum. */
int gpt_fit_sum(gpt_ctx *ctx, int nterms, const int *kernel_ids, const double *params, const int *nparams,
                double noise_var, const double *y, const double *err_y, double diag_add, double *ll_data_out,
                double *logdet_half_out);

/* gpt_fit_sum with PRODUCT terms: term t is kernel_ids[t] alone (kernel_ids2[t] < 0) or the product kernel_ids[t] *
 * kernel_ids2[t]; its parameters are nparams[t] consecutive entries of `params`, the first nparams1[t] of them the first
 * factor's (nparams1[t] == nparams[t] for a plain term).  A model like k1 * k2 + k3 therefore runs the fused builder, not a
 * host-assembled K_tot through gpt_fit_matrix.  gpt_predict afterwards uses the same terms. */
int gpt_fit_terms(gpt_ctx *ctx, int nterms, const int *kernel_ids, const int *kernel_ids2, const double *params,
                  const int *nparams, const int *nparams1, double noise_var, const double *y, const double *err_y,
                  double diag_add, double *ll_data_out, double *logdet_half_out);

/* nbatch INDEPENDENT evaluations of the resident data set in one launch sequence: element b uses params[b * nparams ..],
 * noise_var[b] and the target y[b * N ..] (the mean function may depend on the hyperparameters); err_y (N) is shared.
 * Replaces the loops over hyperparameter vectors of the reference's likelihood grid and random starts
 * (ref: gaussian_process.py:1607-1692, :723-735, gp_utils.py:98-115) at the sizes those run at: N <= 8192 resident
 * points (it pays up to N ~ 4096: 2.3x one gpt_fit per vector there, 9.5x at 1024; nbatch N^2 doubles of device memory), one native kernel
 * (sums, products and a linear transform: gpt_fit_batch_sum / gpt_fit_batch_terms below).
 * Every kernel of the factorisation carries the batch in a grid dimension; an element's results are bit-identical to
 * gpt_fit's for the same inputs.  info_out[b] = 0, or the LAPACK index of the leading minor that is not positive
 * definite (ll_data_out[b] is then meaningless); the call itself returns GPT_OK in both cases. */
int gpt_fit_batch(gpt_ctx *ctx, int nbatch, int kernel_id, const double *params, int nparams,
                  const double *noise_var, const double *y, const double *err_y, double diag_add,
                  double *ll_data_out, double *logdet_half_out, int32_t *info_out);

/* The same for a SumKernel of native kernels (ref: kernel/core.py:549-584; gpt_fit_sum for one matrix): element b's
 * parameters are the terms' parameters concatenated, params[b * ptot ..] with ptot = sum of nparams[t].  Bit-identical to
 * gpt_fit_sum per element. */
int gpt_fit_batch_sum(gpt_ctx *ctx, int nbatch, int nterms, const int *kernel_ids, const double *params,
                      const int *nparams, const double *noise_var, const double *y, const double *err_y,
                      double diag_add, double *ll_data_out, double *logdet_half_out, int32_t *info_out);

/* The batched evaluator for every model gpt_fit_terms takes: product terms (kernel_ids2 / nparams1 as in gpt_fit_terms) and, when
 * gpt_set_T has set one, the linear transform (y then holds nbatch x Ny targets, err_y Ny entries; K_tot = T (K + noise_var I) T^T +
 * ... per element by two batched GEMMs with the shared T).  Needs Nx, Ny <= 8192; bit-identical to gpt_fit_terms per element.
 * ref: gaussian_process.py:1607-1692, gp_utils.py:98-115 (the reference's grids work for any model). */
int gpt_fit_batch_terms(gpt_ctx *ctx, int nbatch, int nterms, const int *kernel_ids, const int *kernel_ids2, const double *params,
                        const int *nparams, const int *nparams1, const double *noise_var, const double *y, const double *err_y,
                        double diag_add, double *ll_data_out, double *logdet_half_out, int32_t *info_out);

/* Free / total bytes of the context's GPU, and the release of the batched evaluator's scratch (nbatch matrices; otherwise kept
 * until gpt_ctx_destroy): GaussianProcess.ll_batch sizes its chunks by the first and returns the memory with the second. */
int gpt_mem_info(gpt_ctx *ctx, int64_t *free_bytes, int64_t *total_bytes);
int gpt_release_batch_scratch(gpt_ctx *ctx);

/* Same as gpt_fit but for an explicit, caller-assembled symmetric K_tot (host, (N, N) row-major;
 * only the lower triangle is read): used for the `T` (linear transform) branch,
 * ref: gaussian_process.py:1443-1446, where K_tot = T (K + noise_K) T^T + ... is (N_y, N_y). */
int gpt_fit_matrix(gpt_ctx *ctx, const double *K_tot, int64_t N, const double *y,
                   double *ll_data_out, double *logdet_half_out);

/* State read-back (host outputs).  L: (N, N) lower, strict upper zero like scipy.linalg.cholesky
 * (ref: gaussian_process.py:1452).  alpha: (N,) = K_tot^-1 y (ref :1462). */
int gpt_get_L(gpt_ctx *ctx, double *L_out);
int gpt_get_alpha(gpt_ctx *ctx, double *alpha_out);

/* ---- GaussianProcess.predict (non-MCMC branch) -------------------------------------------- */
/* Analytic gradient of the LML data term (ref: gaussian_process.py:1471-1520), after gpt_fit / gpt_fit_sum (no transform):
 *   out[h]  = 1/2 * sum_ab (alpha_a alpha_b - (K_tot^-1)_ab) * dK[a][b] / d theta_h      for h < nh, where theta_h is
 *             parameter local_idx[h] (0 = sigma_f, j = l_j) of kernel term term_idx[h];
 *   out[nh] = 1/2 * (alpha^T alpha - tr K_tot^-1)   (times 2 sigma_n this is the derivative for a DiagonalNoiseKernel).
 * K_tot^-1 is formed on the device from the resident factor (triangular inverse + U U^T on the MFMA GEMM, ~N^3 flop
 * whatever nh is -- the reference spends 2 N^3 per parameter in cho_solve); dK is never materialised.
 * Squared-exponential terms only (GPT_E_NOTIMPL otherwise, like matern.py:543-544). */
int gpt_ll_grad(gpt_ctx *ctx, int nh, const int *term_idx, const int *local_idx, double *out);

/* Replaces  ref: gaussian_process.py:965-1006 for the T-free, untransformed case:
 *   Kstar = K(X, Xstar) ; mean = Kstar^T alpha ; v = L^-1 Kstar ; cov = K(Xstar,Xstar) - v^T v ;
 *   std = sqrt(diag(cov)).
 * want: 0 = mean only, 1 = mean + std, 2 = mean + cov (M, M) (+ std if std_out != NULL).
 * noise_params != NULL adds the DiagonalNoiseKernel term to K(Xstar, Xstar) (ref :985-986,
 * kernel/noise.py:103-104; the X-vs-Xstar term of ref :967-968 is identically zero because that
 * call is not `symmetric`, kernel/noise.py:109-110).
 * Uses the kernel / params of the last gpt_fit.
 * cov_out may be pageable or pinned host memory (gpt_host_alloc): into pinned memory the block rows of the covariance are
 * written by asynchronous DMA while the next block column of the SYRK is computed; pageable memory is reached through a
 * pinned staging ring.  want == 2 with cov_out == NULL leaves the covariance on the device (nothing of size M^2 crosses
 * PCIe; mean and, if std_out != NULL, std are still returned). */
int gpt_predict(gpt_ctx *ctx, const double *Xstar, const int32_t *nstar, int64_t M, int want,
                const double *noise_params, const int32_t *noise_n,
                double *mean_out, double *std_out, double *cov_out);

/* Posterior samples from the predictive covariance left on the device by gpt_predict(want = 2, cov_out = NULL)
 * (ref: gaussian_process.py:1295-1300, :1330 -- draw_sample with rand_vars, method = 'cholesky'):
 *   out (M x S, row-major) = cholesky(cov + diag_add I, lower) * rand (M x S, row-major);  the caller adds the mean.
 * M must be the M of that gpt_predict call (GPT_E_ARG otherwise).
 * The M x M covariance never leaves the device; the fit's resident factor is untouched; the covariance is consumed.
 * Status > 0: that leading minor is not positive definite (numpy.linalg.LinAlgError, as scipy.linalg.cholesky). */
int gpt_cov_sample(gpt_ctx *ctx, int64_t M, double diag_add, const double *rand, int64_t S, double *out);

/* Generic right-hand sides against the resident factor (host in/out, row-major):
 *   gpt_solve_L   : B (N, nrhs) <- L^-1 B          (ref: gaussian_process.py:983 solve_triangular)
 *   gpt_cho_solve : B (N, nrhs) <- K_tot^-1 B      (ref: gaussian_process.py:1462,1487,1503 cho_solve) */
/* Pinned (page-locked) host memory for large results: hipHostMalloc / hipHostFree behind a C ABI, so that a binding can hand
 * gpt_predict a destination the DMA engine writes directly. */
int gpt_host_alloc(int64_t bytes, void **out);
int gpt_host_free(void *p);

int gpt_solve_L(gpt_ctx *ctx, double *B, int64_t nrhs);
int gpt_cho_solve(gpt_ctx *ctx, double *B, int64_t nrhs);

/* Per-phase timings of the last gpt_fit in milliseconds (HIP events on the context's stream):
 * out[0]=upload, [1]=kbuild, [2]=potrf, [3]=ll tail, [4]=total; returns the count written. */
int gpt_last_timings(gpt_ctx *ctx, double *out_ms, int n);

/* With option "profile_gemm" = 1 every large (>= 1 GFLOP) trailing-update GEMM/SYRK launch is bracketed
 * by HIP events on the stream it runs on.  Reads and resets the accumulators:
 * out3[0] = algorithmic flops (2k per computed element of C, lower trapezoid for SYRK-style launches),
 * out3[1] = summed launch durations in ms, out3[2] = number of launches. */
int gpt_gemm_profile_read(gpt_ctx *ctx, double *out3);
/* The same with out4[3] = the launches' ALGORITHMIC bytes (16 B per computed element of C -- read and written once -- plus the
 * operand panel once, 8 k max(m, n)): what bench.py's roofline sets the measured HBM traffic of the same launches against. */
int gpt_gemm_profile_read4(gpt_ctx *ctx, double *out4);

/* Standalone dense kernels on host matrices (used by parity tests and the roofline bench). */
int gpt_potrf_host(gpt_ctx *ctx, double *A, int64_t N);                 /* in place, lower */
int gpt_gemm_nt_host(gpt_ctx *ctx, int64_t m, int64_t n, int64_t k, double alpha, const double *A,
                     const double *B, double beta, double *C);          /* C = beta C + alpha A B^T */

/* ---- device API --------------------------------------------------------------------------- */
/* All pointers below are DEVICE pointers unless marked host; work is enqueued on the context's
 * stream and NOT synchronised.  Leading dimensions are in elements.  Block sizes: every m, n, k
 * passed to the dense routines must be a multiple of 64 (callers pad). */

/* Fused covariance builder on device data.  Writes the (M, P) block K[i][j], i in [0,M), j in
 * [0,P) of k(Xi[i], Xj[j], ni[i], nj[j]) to dK (row stride ldk).  lower_only != 0: only tiles
 * that intersect {i + i0 >= j + j0} are written (i0, j0 = global offsets of the block).  If
 * d_err_y != NULL, entries with i + i0 == j + j0 get ((k + noise_var) + err_y[i+i0]^2) + diag_add. */
int gpt_dev_kbuild(gpt_ctx *ctx, int kernel_id, const double *params_host, int nparams,
                   const double *dXi, const int32_t *dni, int64_t M,
                   const double *dXj, const int32_t *dnj, int64_t P, int D,
                   int hyper_deriv, int symmetric, const int32_t *noise_n_host,
                   int lower_only, int64_t i0, int64_t j0,
                   const double *d_err_y, double noise_var, double diag_add,
                   double *dK, int64_t ldk);

/* C (m x n) = beta*C + alpha * A (m x k) * B (n x k)^T.  tri != 0: C's origin lies on the global
 * diagonal and only 64-aligned tiles with col_tile <= row_tile are computed (SYRK-style). */
int gpt_dev_gemm_nt(gpt_ctx *ctx, int64_t m, int64_t n, int64_t k, double alpha,
                    const double *dA, int64_t lda, const double *dB, int64_t ldb,
                    double beta, double *dC, int64_t ldc, int tri);
/* Block-cyclic trailing update in one launch: C (m x nseg*seg_cols, ldc) += alpha * A * B_q^T + beta-scaled C, where
 * column segment q (seg_cols columns) uses the B rows starting at row q * b_stride and only its rows >= q * row_step
 * are updated (lower trapezoid inside the segment; its diagonal block sits at row q * row_step).  With A = B = the
 * received panel this applies panel k to all block columns a rank owns to the right of it (they are adjacent in the
 * rank's storage, world * nb apart in the matrix): gptools_amd/dist.py, SURVEY.md section 8e.
 * seg_cols, row_step multiples of 64; k a multiple of 16; b_stride >= seg_cols. */
int gpt_dev_gemm_nt_stair(gpt_ctx *ctx, int64_t m, int64_t nseg, int64_t seg_cols, int64_t k, double alpha,
                          const double *dA, int64_t lda, const double *dB, int64_t ldb, int64_t b_stride,
                          int64_t row_step, double beta, double *dC, int64_t ldc);

/* Trailing update of one rank of the 2-D block-cyclic engine (gptools_amd/dist.py GridLML) in
 * one launch.  The rank holds the global block rows I = pr + li * den and block columns J = pc + lj * num of the matrix (block
 * size seg_cols both ways).  C = its local matrix from the update's first block row / column on, A (m x k) = the rank's rows of
 * the panel, B = its columns of the panel (column segment q: rows [q * seg_cols, (q + 1) * seg_cols)).  Segment q (global block
 * column J0 + q * num) is updated from its first block row with I >= J on: local block row ceil((off + q * num) / den) - base,
 * with off = J0 - pr and base = the local index of the update's first block row; where that first block is a diagonal block
 * of the matrix ((off + q * num) % den == 0) only its lower 64 x 64 tiles are computed.  The reference has nothing to mirror here
 * (gaussian_process.py:1452 is one LAPACK call). */
int gpt_dev_gemm_nt_gridstair(gpt_ctx *ctx, int64_t m, int64_t nseg, int64_t seg_cols, int64_t k, double alpha,
                              const double *dA, int64_t lda, const double *dB, int64_t ldb, int64_t off, int64_t num,
                              int64_t den, int64_t base, double beta, double *dC, int64_t ldc);
/* d_acc[0] += sum_{c < w} d_row[c]^2: the z.z part of ll (ref gaussian_process.py:1463) from the piece of the augmented row a
 * rank of that engine holds (one workgroup, fixed order). */
int gpt_dev_row_sumsq(gpt_ctx *ctx, const double *d_row, int64_t w, double *d_acc);

/* Factor one block column ("panel"): A is (m x nb), its top nb x nb block is the diagonal block.
 * On exit the top block holds L_kk (lower) and the rows below hold A21 * L_kk^-T.  d_invd:
 * workspace/output of GPT_WS_BLOCK (= 9216) doubles per 128 columns: the inverses of L's 16x16 diagonal
 * blocks and L's strictly-lower 16x16 blocks, packed in MFMA operand order for the panel TRSM.
 * d_info: int32 on device, set to info_base + j + 1 at the first non-positive pivot (never
 * cleared here). */
int gpt_dev_potrf_panel(gpt_ctx *ctx, int64_t m, int64_t nb, double *dA, int64_t lda,
                        double *d_invd, int32_t *d_info, int64_t info_base);

/* Whole lower Cholesky of the n x n matrix dA in place (blocked right-looking, look-ahead).
 * d_invd: (n/128)*GPT_WS_BLOCK doubles. */
int gpt_dev_potrf(gpt_ctx *ctx, int64_t n, double *dA, int64_t lda, double *d_invd, int32_t *d_info);

/* B (m x n) <- B * L^-T with L (n x n) lower, d_invd from the factorisation of L. */
int gpt_dev_trsm_rlt(gpt_ctx *ctx, int64_t m, int64_t n, const double *dL, int64_t ldl,
                     const double *d_invd, double *dB, int64_t ldb);

/* dW (n x n, row-major, ldw; zeros above the diagonal) <- L^-1, L (n x n) a factored diagonal block, d_invd from its
 * factorisation.  The TRSM of many rows then is one GEMM: gpt_dev_gemm_nt(m, n, n, 1, B, ldb, dW, ldw, 0, X, ldx, 0)
 * gives X = B L^-T (the reference reaches the same quantity through scipy.linalg.cho_factor / LAPACK dtrsm inside
 * dpotrf, gaussian_process.py:1452). */
int gpt_dev_trinv(gpt_ctx *ctx, int64_t n, const double *dL, int64_t ldl, const double *d_invd,
                  double *dW, int64_t ldw);

/* Small helpers of the one-process-per-GPU path (gptools_amd/dist.py), so that its queues carry no generic framework
 * kernels.  All asynchronous on the context's stream.
 *   gpt_dev_copy2d       : dst (rows x cols, ldd) <- src (rows x cols, lds) -- staging a block column of the local matrix
 *                          into a contiguous panel buffer.
 *   gpt_dev_pad_block    : rows [n_valid, n_pad) of a block column whose first column has global index c0 (nb columns,
 *                          row-major at dA = its element (row 0 of the matrix, column c0), row stride lda): the augmented row
 *                          n_valid gets y[c0 + c] (c0 + c < n_valid), the diagonal of the padding 1, the pivot under the
 *                          augmented row `big`, everything else 0 (layout of DESIGN.md section 3).
 *   gpt_dev_panel_scalars: d_acc[0] += sum_{i < w} log P[i][i]; if zrow >= 0 also d_acc[1] += sum_{c < w} P[zrow][c]^2 --
 *                          the two scalars of ll (ref gaussian_process.py:1463-1467) taken from a factored panel buffer
 *                          (P = its diagonal block, row stride ldp); accumulated in a fixed order (one workgroup). */
int gpt_dev_copy2d(gpt_ctx *ctx, int64_t rows, int64_t cols, const double *d_src, int64_t lds, double *d_dst, int64_t ldd);
/* The same copy on an explicit stream of the caller (a hipStream_t; NULL = the context's): the column exchange of the 2-D
 * engine gathers / scatters panel blocks on a queue that carries nothing else and has no context of its own. */
int gpt_dev_copy2d_on(gpt_ctx *ctx, void *stream, int64_t rows, int64_t cols, const double *d_src, int64_t lds, double *d_dst,
                      int64_t ldd);
int gpt_dev_pad_block(gpt_ctx *ctx, double *dA, int64_t lda, int64_t c0, int64_t nb, int64_t n_valid, int64_t n_pad,
                      const double *d_y, double big);
int gpt_dev_panel_scalars(gpt_ctx *ctx, const double *dP, int64_t ldp, int64_t w, int64_t zrow, double *d_acc);

/* ---- compiled schedules of the partitioned engines (round 6) ------------------------------------------------------------
 * The one-process-per-GPU block-cyclic Cholesky of gptools_amd/dist.py issues ~1000 operations per rank and evaluation; which
 * ones, on which buffers, in which order is static per (N, nb, world, rank).  The Python layer records its step loop once as an
 * op list -- GPT_PLAN_W = 20 int64 per op: [opcode, queue, a0 .. a17], doubles as their bit patterns, queues 0 = main, 1 = panel,
 * 2 = recv (the contexts given here, in this order), 3 + c = the stream of the plan's communication channel c -- and every
 * evaluation is then one gpt_plan_run: a C loop over the list, RCCL (dlopen'ed librccl) called directly for the panel exchanges.
 * Opcodes (csrc/api_plan.inc):
 *   0 record event a0 | 1 wait event a0 | 2 K block (r0, r1, c0, c1, out, ld) | 3 pad block (ptr, lda, c0, nb, N, NP, y, big) |
 *   4 copy2d (rows, cols, src, lds, dst, ldd) | 5 potrf_panel (m, nb, A, lda, invd, info, info_base) | 6 trinv (nb, L, ldl, invd, W, ldw) |
 *   7 gemm_nt (m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tri) | 8 gemm_nt_stair (m, nseg, seg_cols, k, alpha, A, lda, B, ldb,
 *   b_stride, row_step, beta, C, ldc) | 9 panel_scalars (buf, ld, w, zrow, red) | 10 broadcast (buf, count, root) |
 *   11 scatter (buf, count per rank, root) | 12 all-gather in place (buf, count per rank) | 13 K rectangle (Xi, ni, r0, r1, c0, c1,
 *   out, ld: rows from the caller's own row table, no diagonal terms) | 14 row_sumsq (row, count, out) | 15 gemm_nt_gridstair (m,
 *   nseg, seg_cols, k, alpha, A, lda, B, ldb, off, num, den, base, beta, C, ldc).  Collectives (10-12) sit on a channel queue,
 *   kernels on a context queue, events on either.
 * dX (N x D float64), dn (N x D int32), d_err (N float64) are the device arrays the K-block ops read; the per-evaluation inputs
 * are arguments of gpt_plan_run.  A CHANNEL is one communicator + one high-priority stream (up to GPT_PLAN_CHANNELS; the 1-D
 * engine has one over all ranks, the 2-D engine five: process row x 2, process column x 2, the whole grid): gpt_plan_set_channel
 * is collective over the channel's ranks (ncclCommInitRank; the 128-byte id comes from gpt_plan_unique_id on the channel's rank 0
 * and travels by whatever means the caller has; every rank sets its channels in the same order); root / rank are positions in
 * that communicator.  gpt_plan_set_comm = channel 0.  The ops of a channel without a communicator are skipped (single rank).
 * The caller keeps the buffers alive and synchronises the contexts' streams itself.  (Environment GPT_PLAN_PROFILE=1: gpt_plan_run prints
 * its host time per opcode to stderr.) */
#define GPT_PLAN_W 20
#define GPT_PLAN_CHANNELS 8
typedef struct gpt_plan gpt_plan;
int gpt_plan_unique_id(void *out128);
int gpt_plan_create(int nctx, gpt_ctx **ctxs, const int64_t *ops, int64_t nops, int nevents, const void *dX, const void *dn, int D,
                    const void *d_err, gpt_plan **out);
int gpt_plan_set_comm(gpt_plan *plan, int nranks, int rank, const void *unique_id128);
int gpt_plan_set_channel(gpt_plan *plan, int channel, int nranks, int rank, const void *unique_id128);
int gpt_plan_run(gpt_plan *plan, int kernel_id, const double *params, int nparams, double noise_var, double diag_add);
double gpt_plan_last_enqueue_ms(gpt_plan *plan);          /* host time gpt_plan_run spent enqueueing, last call */
int gpt_plan_destroy(gpt_plan *plan);

#ifdef __cplusplus
}
#endif
#endif /* GPT_HIP_H_ */
