// api.hip -- C ABI (include/gpt_hip.h) and host-side orchestration for libgpt_hip.so.
//
// Host logic restated here (not kernels): the blocked right-looking Cholesky with a recursive
// panel and one-panel look-ahead on a second, high-priority HIP stream; the padded / augmented
// matrix layout; GaussianProcess.compute_K_L_alpha_ll (ref: gptools/gaussian_process.py:1418-1469)
// and GaussianProcess.predict (ref: gptools/gaussian_process.py:965-1006) sequencing.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>
#include "common.hpp"

// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void gpt_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *gpt_last_error(void) { return g_err; }
extern "C" int gpt_version(void) { return 100; }

#define GPT_TRY(expr)          \
    do {                       \
        int rc_ = (expr);      \
        if (rc_ != GPT_OK) return rc_; \
    } while (0)

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

enum { SLOT_XI = 0, SLOT_XJ, SLOT_NI, SLOT_NJ, SLOT_OUT, SLOT_KST, SLOT_KSS, SLOT_XS, SLOT_NS, SLOT_VEC, SLOT_VEC2,
       SLOT_RHS, SLOT_LOW, SLOT_KFULL, SLOT_TK, SLOT_ZERO, SLOT_UINV, SLOT_WINV, SLOT_GPART, SLOT_BINV, SLOT_BTMP,
       SLOT_BINV2, SLOT_BINV3, SLOT_BINV3U, SLOT_BINVU, SLOT_BATCH_A, SLOT_BATCH_WS, SLOT_BATCH_MISC, SLOT_SPLITK, SLOT_COUNT };

struct gpt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t panel_stream = nullptr;
    hipStream_t helper_stream = nullptr;   // CU-masked to part of the reserved CUs (own streams only)
    hipStream_t late_panel_stream = nullptr;   // panel stream of the chain-bound end: masked to the reserved CUs only
    int64_t late_rows = 0;                 // panels with at most this many rows left run on it (0 = off)
    hipStream_t early_stream = nullptr;    // main stream of the update-bound head of a factorisation: fewer CUs reserved
    int64_t early_rows = 0;                // panels with more than this many rows left run their updates there (0 = off)
    int64_t nb_early = 0, nb_switch_rows = 4608;   // see potrf_enqueue (panel widths)
    int pad_now = 0;                       // (set per panel by potrf_enqueue: LDS pad of the main stream's updates right now)
    int late_pad = 0;                      // > 0: LDS pad of the main stream's updates once at most late_pad_rows rows remain -- fewer of its
    int64_t late_pad_rows = 4608;          //      workgroups per CU, so that the panel stream's chain kernels share the CUs with less contention
    unsigned *d_edge = nullptr;            // edge-flag words (EdgeSig, common.hpp): [0,1] "panel k final", [16,17] "urgent update k done"
    EdgeSig first_wait;                    // ... handed by potrf_enqueue to the first leaf launch (panel_ext)
    EdgeSig head_wait;                     // set by fit_terms: the first leaf of the next factorisation waits for this word (K-build head)
    unsigned edge_seq = 0;                 // value of the last edge raised (monotonic over the context's life)
    int64_t merge_min_tiles = 512;         // ... while the merged launch has at least this many 64x64 tiles (>= 512: it needs an order table)
    int64_t purg_rows_flags = 0;           // purg_rows while flag edges + merged launches are in use
    int64_t tail_wait = 0;                 // 1: the main stream's last launch of a panel awaits the NEXT panel's flag at its end (gemm.hip "tail wait";
                                           //    measured the same: 4.459 / 4.471 ms at N = 8192 -- the launches' own drain hides the wait kernel)
    int64_t merge_urgent = 1;              // 1: with flag edges, urgent + rest of a panel are ONE launch (urgent tiles first, partial flag)
    int64_t edge_flags = 1;                // 1: those two edges of the look-ahead may be flag words instead of events (see EvalScope)
    bool flags_now = false;                // ... and ARE, in the evaluation in progress (set by EvalScope)
    int reserve_cus = 0;                   // CUs the main stream's mask leaves to the panel stream (0: unmasked)
    int64_t head_wait_wgs = 33;            // first leaf: in-kernel wait for the K build's head while its launch has at most this many workgroups
    bool defer_join = false;               // potrf_enqueue leaves the final panel -> main join to its caller (factor_and_ll)
    hipStream_t tail_stream = nullptr;     // ... and reports the stream the factorisation ended on
    int64_t gemm_prio = -1;                // >= 0: wave priority of ALL GEMM main loops of this context
    int64_t panel_prio = 2;                // wave priority (0..3) of the panel stream's GEMM main loops
    int64_t purg_rows = 6144;              // > 0: while more rows than this remain, the panel stream does the "urgent" update itself
                                           // (N=8192: 5.36 against 5.44 ms, bit-identical; no effect below ~7k rows or with the helper stream)
    int64_t defer_rows = 0;             // chain-bound end: with at most this many rows left, the main stream's "rest" update of
                                           // panel k starts only after the panel stream's first update of panel k+1 (0 = off)
    int helper_cus = 0;
    std::vector<hipEvent_t> events;       // sync-only events (look-ahead fork/join)
    hipEvent_t tev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    // options
    int64_t nb_outer = 0;              // outer block width; 0 = by size (outer_width())
    int lookahead = 1;
    int use_graph = 0;
    long n_maxsum = 0;                 // largest row sum of the training derivative orders (gpt_set_data)
    int timing = 0;
    int tile = 0;
    int gemm_pad = 1024;
    int64_t fuse_trsm = 8192;          // panels with at most this many rows below the leaf use potf2_trsm_kernel (0 = never)
    int leaf256 = 0;                   // 256-column leaves (potf2x2_trsm_kernel) where two 128-column leaves of a short panel follow each other
    double *d_l10pk = nullptr;         // its scratch: the packed block between the two diagonal blocks (16384 doubles)
    unsigned *d_flag = nullptr;        // progress word of potf2_trsm_kernel (only ever raised)
    unsigned flag_epoch = 0;
    unsigned x1_count = 0;             // value of d_flag[32..39], the producers' per-step counters of potf2_trsm_upd_kernel (only ever counted up)
    int64_t fuse_upd = 0;              // 1: leaves with at most fuse_upd_rows rows below them apply their rank-128 update of the next
    int64_t fuse_upd_rows = 4096;      //    128 / 256 columns inside the leaf's launch (potf2_trsm_upd_kernel); measured slower than
                                       //    the separate update launch (profiles/r05_upd_ab.txt): off
    hipStream_t near_stream = nullptr;     // second main stream (same CU mask), created on first use: the rank-w "near" updates of paired panels
    int64_t binv_launches = 0;             // 1: the 512-wide block inverses by the recursion over 15 launches of rounds 2-4 instead of trinv512_kernel
    int64_t splitk = 512;                  // few-rows solves: GEMMs with k >= 1024 of fewer 32x32 tiles than this are split along k until they reach it (0: never)
    int64_t pair_rows = 0;                 // > 0: while more rows than this remain, panels are taken in PAIRS -- after the first one only the next
                                           // panel's columns are updated (rank w, near_stream), after the second everything to the right in ONE
                                           // rank-2w launch (potrf_enqueue "panel pairs")
    int64_t fuse_rows32 = 2048, fuse_rows16 = 0;   // ... 32 / 16 rows per consumer workgroup: two / one strip waves per CU (same-box A/B, bit-identical: N = 4096
                                                   // 1.158 -> 1.152 ms, N = 8192 4.346 -> 4.31 ms with 32 rows below 2048; 16 rows: no further gain)
    int64_t fuse_rows64 = 2048;        // fused leaves with at most this many rows below them: 64 rows per consumer workgroup (one strip
                                       // wave per SIMD, potf2_trsm_kernel<.., true>); 0 = always 128
    int64_t helper_min_n = 12288;      // the helper stream takes part only above this matrix size
    int helper_tf = 45;                // assumed rate of the helper stream, in 0.1 TFLOP/s per 24 CUs (0 = no helper);
                                       // measured: 0 / 25 / 35 / 50 -> 212 / 209 / 206 / 214 ms at N=32768, 30.8 / 30.6 / 30.2 / 32.0 at N=16384
    int ramp = 0;                      // first panels 128, 256, ... wide (see potrf_enqueue); measured slower, off
    int inner = 0;                     // look-ahead panel: 0 right-looking leaves, 1 left-looking (panel_ext_ll), 2 left-looking
    int64_t inner_rows = 4608;         //   once at most inner_rows rows remain (the chain-bound end of the factorisation)
    hipEvent_t head_event = nullptr;   // set by gpt_fit: the first nb_outer+128 columns of K_tot are built (panel 0 may start)
    // resident training inputs
    int64_t N = 0;             // order of the factorised matrix (= Nx without T, = Ny with T)
    int64_t Nx = 0;            // resident points
    int D = 0;
    double *dT = nullptr;      // linear transform T (Ny x Nx), zero-padded to (round_up(Ny,64) x round_up(Nx,16))
    int64_t Ny = 0, NxP = 0;
    double *dX = nullptr;
    int32_t *dn = nullptr;
    // factorisation state
    int64_t NP = 0;            // padded order (multiple of 128, > N)
    double *dA = nullptr;      // NP x NP, row-major, lower triangle meaningful
    double *d_invd = nullptr;  // (NP/128) x GPT_WS_BLOCK packed workspace (see common.hpp)
    int32_t *d_info = nullptr;
    double *d_y = nullptr, *d_erry = nullptr, *d_scal = nullptr, *d_alpha = nullptr;
    double *h_scal = nullptr;  // pinned
    double *h_yerr = nullptr;  // pinned staging for y | err_y (a pageable source would make the upload synchronous)
    double *h_alpha = nullptr; // pinned landing place of alpha (a pageable destination costs a staged, synchronous copy: ~70 us for 64 KB)
    bool h_alpha_valid = false;
    int32_t *h_info = nullptr; // pinned
    bool factored = false, alpha_valid = false, have_kernel = false;
    bool binv_valid = false;           // SLOT_BINV holds the inverses of the 512x512 diagonal blocks of the resident factor
    bool binv2_valid = false;          // SLOT_BINV2 those of its 1024x1024 diagonal blocks (solves with very few rows)
    bool binv3_valid = false;          // SLOT_BINV3 those of its 2048x2048 diagonal blocks (the same, large factors)
    int64_t eager_alpha = 0;           // 1: every evaluation also enqueues alpha = L^-T z behind the factorisation (no second host round trip)
    // (all three are always built for the whole padded order, floor(NP / width) blocks, whatever extent the caller needs:
    // gpt_ll_grad and the solves ask for different extents at N = 512 k - 128, and a valid flag says nothing about how far)
    unsigned alpha_counter = 0;        // value of the step counter of the wide back-substitution (d_edge[40], only ever raised)
    int64_t debug_poison = 0;          // option "debug_poison": gpt_ll_grad fills its scratch matrices with NaN first (test aid)
    int64_t edge_test_stall = 0;       // option "edge_test_stall": the next head flag is withheld once (test aid, see fit_terms_once)
    double *h_stage = nullptr;         // pinned staging ring for results that go to pageable host memory (2 x GPT_STAGE_BYTES)
    hipStream_t copy_stream = nullptr; // device-to-host copies that overlap the next block's compute (created on first use)
    hipEvent_t cev[4] = {nullptr, nullptr, nullptr, nullptr};
    double *h_batch = nullptr;         // pinned staging of gpt_fit_batch (y, KParams, noise variances, err_y in; results out)
    size_t h_batch_cap = 0;
    int64_t cov_M = 0;                 // > 0: SLOT_KSS holds the lower triangle of the predictive covariance of the last gpt_predict(want = 2, cov_out = NULL)
    KParams kp;                      // first term (single-kernel paths)
    std::vector<KParams> terms;      // the model kernel as a sum of native kernels (gpt_fit_sum)
    std::vector<KParams> terms2;     // ... term t is the PRODUCT terms[t] * terms2[t] where terms2[t].kernel_id >= 0 (gpt_fit_terms)
    double timings[5] = {0, 0, 0, 0, 0};
    // per-launch HIP-event timing of the dominant (large) GEMM/SYRK launches, for the roofline line
    int prof_gemm = 0;
    struct GemmProf { hipEvent_t e0, e1, stop; double flops; };
    std::vector<GemmProf> gprof;
    size_t gprof_used = 0;
    double prof_flops = 0, prof_ms = 0, prof_count = 0;
    // graph cache for the factorisation
    hipGraphExec_t gexec = nullptr;
    int64_t g_n = 0, g_nb = 0;
    int g_la = 0;
    double *g_A = nullptr;
    DevBuf slots[SLOT_COUNT];
};

// rows per consumer workgroup of a fused leaf with m rows below it, as launch_potf2_trsm's code: 0 = 128, 1 = 64, 2 = 32, 3 = 16
static int strip_rows_code(const gpt_ctx *c, int64_t m) { return m <= c->fuse_rows16 ? 3 : m <= c->fuse_rows32 ? 2 : m <= c->fuse_rows64 ? 1 : 0; }

static int ensure(gpt_ctx *c, int slot, size_t bytes, void **out)
{
    DevBuf &b = c->slots[slot];
    if (b.cap < bytes) {
        if (b.p) GPT_HIP_CHECK(hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
        const size_t want = bytes + bytes / 8 + 256;
        GPT_HIP_CHECK(hipMalloc(&b.p, want));
        b.cap = want;
    }
    *out = b.p;
    return GPT_OK;
}

static hipEvent_t get_event(gpt_ctx *c, size_t idx)
{
    while (c->events.size() <= idx) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;      // (timing-capable: an edge can double as a GEMM stop event)
        c->events.push_back(e);
    }
    return c->events[idx];
}

// ------------------------------------------------------------------------------------------------
static int make_kparams(int kernel_id, const double *params, int nparams, int D, int hyper_deriv, int symmetric,
                        const int32_t *noise_n, KParams *kp)
{
    if (D < 1 || D > GPT_MAX_DIM) {
        gpt_set_error("num_dim %d out of range [1, %d]", D, GPT_MAX_DIM);
        return GPT_E_ARG;
    }
    memset(kp, 0, sizeof(*kp));
    kp->kernel_id = kernel_id;
    kp->D = D;
    kp->hyper_deriv = hyper_deriv < 0 ? -1 : hyper_deriv;
    kp->symmetric = symmetric ? 1 : 0;
    if (kernel_id == GPT_KERNEL_SE || kernel_id == GPT_KERNEL_M52) {
        if (nparams != D + 1) {
            gpt_set_error("kernel %d expects %d params, got %d", kernel_id, D + 1, nparams);
            return GPT_E_ARG;
        }
        if (kernel_id == GPT_KERNEL_M52 && hyper_deriv >= 0) {
            gpt_set_error("Hyperparameter derivatives have not been implemented!");
            return GPT_E_NOTIMPL;
        }
        if (hyper_deriv >= nparams) {
            gpt_set_error("hyper_deriv %d out of range", hyper_deriv);
            return GPT_E_ARG;
        }
        kp->sigma = params[0];
        for (int d = 0; d < D; d++) {
            const double l = params[1 + d];
            kp->l[d] = l;
            kp->inv_l[d] = 1.0 / l;
            kp->inv_var[d] = 1.0 / (l * l);
            // (from the value the pair functions multiply by: a SUBNORMAL length scale has 1 / l = inf as well, and 0 * inf on the
            // diagonal / for coincident points would be NaN where the reference divides 0 / l = 0, core.py:416 -- ADVICE r4)
            if (l == 0.0 || std::isinf(kp->inv_l[d])) kp->zero_l = 1;
        }
    } else if (kernel_id == GPT_KERNEL_RQ) {
        // RationalQuadraticKernel: [sigma_f, alpha, l_1 .. l_D] (ref: rational_quadratic.py:30-45)
        if (nparams != D + 2) {
            gpt_set_error("kernel %d expects %d params, got %d", kernel_id, D + 2, nparams);
            return GPT_E_ARG;
        }
        if (hyper_deriv >= 0) {
            gpt_set_error("Hyperparameter derivatives have not been implemented!");      // ref: core.py:723-726
            return GPT_E_NOTIMPL;
        }
        kp->sigma = params[0];
        kp->alpha = params[1];
        for (int d = 0; d < D; d++) {
            const double l = params[2 + d];
            kp->l[d] = l;
            kp->inv_l[d] = 1.0 / l;
            kp->inv_var[d] = 1.0 / (l * l);
        }
    } else if (kernel_id == GPT_KERNEL_MATERN) {
        // MaternKernel: [sigma_f, nu, l_1 .. l_D] (ref: matern.py:299-306); the Gamma-function constants the device code
        // needs (Temme's series, the reference's small-y series at nu or nu -+ 0.001) are formed here, in libm
        if (nparams != D + 2) {
            gpt_set_error("kernel %d expects %d params, got %d", kernel_id, D + 2, nparams);
            return GPT_E_ARG;
        }
        if (hyper_deriv >= 0) {
            gpt_set_error("Hyperparameter derivatives have not been implemented!");      // ref: core.py:723-726
            return GPT_E_NOTIMPL;
        }
        const double nu = params[1];
        if (!(nu > 0.0) || !(nu < 60.0)) {
            gpt_set_error("MaternKernel: the order nu must lie in (0, 60), got %g", nu);
            return GPT_E_VALUE;
        }
        kp->sigma = params[0];
        kp->alpha = nu;
        for (int d = 0; d < D; d++) {
            const double l = params[2 + d];
            kp->l[d] = l;
            kp->inv_l[d] = 1.0 / l;
            kp->inv_var[d] = 1.0 / (l * l);
        }
        kp->m_cnu = pow(2.0, 1.0 - nu) / tgamma(nu);
        kp->m_nint = (int)floor(nu + 0.5);
        kp->m_mu = nu - (double)kp->m_nint;
        kp->m_isint = (nu == floor(nu)) ? 1 : 0;
        {
            const double mu = kp->m_mu, mu2 = mu * mu;
            kp->m_gampl = 1.0 / tgamma(1.0 + mu);
            kp->m_gammi = 1.0 / tgamma(1.0 - mu);
            kp->m_gam2 = 0.5 * (kp->m_gammi + kp->m_gampl);
            // (1/Gamma(1-mu) - 1/Gamma(1+mu)) / (2 mu); near mu = 0 from the Taylor series of 1/Gamma(1+t)
            kp->m_gam1 = (fabs(mu) < 1.0e-3)
                             ? -(0.57721566490153286 + mu2 * (-0.042002635034095236 + mu2 * (-0.0072189432466630995)))
                             : (kp->m_gammi - kp->m_gampl) / (2.0 * mu);
        }
        for (int q = 0; q < 2; q++) {
            const double nus = kp->m_isint ? nu + (q == 0 ? -0.001 : 0.001) : nu;      // utils.py:1481-1483 (nu_step)
            kp->m_nus[q] = nus;
            kp->m_g[q] = tgamma(nus);
            kp->m_gm[q] = tgamma(-nus);
        }
    } else if (kernel_id == GPT_KERNEL_DIAGNOISE || kernel_id == GPT_KERNEL_ZERO) {
        if (nparams != 1) {
            gpt_set_error("noise kernels expect 1 param, got %d", nparams);
            return GPT_E_ARG;
        }
        if (hyper_deriv > 0) {
            gpt_set_error("hyper_deriv %d out of range", hyper_deriv);
            return GPT_E_ARG;
        }
        kp->sigma = params[0];
        for (int d = 0; d < D; d++) kp->noise_n[d] = noise_n ? noise_n[d] : 0;
    } else {
        gpt_set_error("unknown kernel_id %d", kernel_id);
        return GPT_E_ARG;
    }
    return GPT_OK;
}

// Matern52 accepts only derivative orders summing to <= 1 per point (ref: kernel/matern.py:545-546).
static int check_m52_orders(const int32_t *n, int64_t M, int D)
{
    for (int64_t i = 0; i < M; i++) {
        long s = 0;
        for (int d = 0; d < D; d++) s += n[i * D + d];
        if (s > 1) {
            gpt_set_error("Matern52Kernel only supports 0th and 1st order derivatives");
            return GPT_E_VALUE;
        }
    }
    return GPT_OK;
}

// The device builder of the rational-quadratic kernel carries GPT_RQ_MAXORD + 1 Faa di Bruno coefficients: the derivative
// orders of any pair (row of ni + row of nj) may sum to GPT_RQ_MAXORD at most.  `pairwise`: rows are matched one to
// one (gpt_kpairs); otherwise every row of ni meets every row of nj (Gram blocks).
static int check_rq_orders(const int32_t *ni, int64_t M, const int32_t *nj, int64_t P, int D, bool pairwise)
{
    long mi = 0, mj = 0, mp = 0;
    for (int64_t i = 0; i < M; i++) {
        long s = 0;
        for (int d = 0; d < D; d++) s += ni[i * D + d];
        if (pairwise && i < P) {
            long t = s;
            for (int d = 0; d < D; d++) t += nj[i * D + d];
            if (t > mp) mp = t;
        }
        if (s > mi) mi = s;
    }
    for (int64_t j = 0; j < P && !pairwise; j++) {
        long s = 0;
        for (int d = 0; d < D; d++) s += nj[j * D + d];
        if (s > mj) mj = s;
    }
    const long worst = pairwise ? mp : mi + mj;
    if (worst > GPT_RQ_MAXORD) {
        gpt_set_error("RationalQuadraticKernel: derivative orders of a pair sum to %ld, the device builder supports %d",
                      worst, GPT_RQ_MAXORD);
        return GPT_E_VALUE;
    }
    return GPT_OK;
}

// ------------------------------------------------------------------------------------------------
// Dense drivers on device data
// ------------------------------------------------------------------------------------------------
static int gemm_nt(gpt_ctx *c, hipStream_t st, int64_t m, int64_t n, int64_t k, double alpha, const double *A,
                   int64_t lda, const double *B, int64_t ldb, double beta, double *C, int64_t ldc, int tri,
                   hipEvent_t done = nullptr, EdgeSig edge = EdgeSig(), EdgeSig wait = EdgeSig(), int64_t edge_cols = 0,
                   EdgeSig tail = EdgeSig())
{
    // algorithmic flop count: 2k per computed element of C (lower trapezoid when tri)
    const double elems = tri ? 0.5 * (double)n * (double)(n + 1) + (double)(m - n) * (double)n : (double)m * (double)n;
    const double flops = 2.0 * (double)k * elems;
    // (the roofline line of bench.py: the trailing updates on the MAIN stream only -- at large N the panel and helper
    // streams also launch >= 1 GFLOP updates, on the few CUs reserved for them and concurrently with these; summing their
    // durations with the main stream's would count the same wall time twice)
    const bool on_main = (st == c->stream) || (c->early_stream && st == c->early_stream);
    const bool prof = c->prof_gemm && flops >= 1.0e9 && on_main;
    gpt_ctx::GemmProf *gp = nullptr;
    if (prof) {
        if (c->gprof_used == c->gprof.size()) {
            gpt_ctx::GemmProf g;
            GPT_HIP_CHECK(hipEventCreate(&g.e0));
            GPT_HIP_CHECK(hipEventCreate(&g.e1));
            g.flops = 0;
            g.stop = g.e1;
            c->gprof.push_back(g);
        }
        gp = &c->gprof[c->gprof_used++];
        gp->flops = flops;
    }
    // trailing updates on the main stream leave room on every CU for the panel stream (see gemm.hip)
    const bool on_near = c->near_stream && st == c->near_stream;      // (shares the main stream's CUs: same room left for the panel stream)
    const int lds_pad = ((on_main || on_near) && c->lookahead) ? (c->pad_now > 0 ? c->pad_now : c->gemm_pad) : 0;
    // `done` (a cross-stream edge) and the timing events ride on the dispatch packet itself where possible
    // (hipExtLaunchKernelGGL): a separate hipEventRecord is a barrier packet, ~6 us of command-processor time
    const bool ext = !c->use_graph && (c->tile == 0 || c->tile == 64);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ext) {
        e0 = prof ? gp->e0 : nullptr;
        e1 = done ? done : (prof ? gp->e1 : nullptr);
        if (prof) gp->stop = e1;
    } else if (prof) {
        gp->stop = gp->e1;
        GPT_HIP_CHECK(hipEventRecord(gp->e0, st));
    }
    // GPT_GEMM_LOG=<file> (evidence aid, scratch/collect_r04.sh): the shape of every >= 1 GFLOP main-stream launch, in launch order, and
    // whether it runs the 64x64 kernel (launch_gemm_nt cuts launches under gemm_small_threshold() tiles into 32x32 tiles unless a
    // start event or a partial edge rides on them), so that a profiler's per-dispatch counters of that kernel can be set against
    // the algorithmic flops / bytes of THE SAME launches
    {
        static FILE *glog = getenv("GPT_GEMM_LOG") ? fopen(getenv("GPT_GEMM_LOG"), "a") : nullptr;
        if (glog && flops >= 1.0e9 && on_main) {
            const int64_t nt64 = ((m + 63) / 64) * ((n + 63) / 64);
            const int k64 = (c->tile == 64) || !(c->tile == 0 && nt64 < gemm_small_threshold() && !e0 && edge_cols == 0);
            fprintf(glog, "%lld %lld %lld %d %.6e %d\n", (long long)m, (long long)n, (long long)k, tri, flops, k64);
            fflush(glog);
        }
    }
    // the panel stream's updates keep a raised wave priority in their main loop (option panel_prio, see gemm.hip)
    // (option gemm_prio >= 0: every GEMM of this context -- the panel-side context of the block-cyclic engine, whose
    // launches all sit on the chain)
    const int prio = (c->gemm_prio >= 0) ? (int)c->gemm_prio : (!on_main && !on_near && c->lookahead) ? (int)c->panel_prio : 0;
    int rc = launch_gemm_nt(st, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tri, c->tile, lds_pad, e0, e1, prio, edge, wait, edge_cols, 1, 0, tail);
    if (!ext) {
        if (prof) GPT_HIP_CHECK(hipEventRecord(gp->e1, st));
        if (done) GPT_HIP_CHECK(hipEventRecord(done, st));
    }
    return rc;
}

// Outer block width of the factorisation.  Measured on MI355X (scratch/nb_sweep.py, with the fused diagonal-block + TRSM
// kernel): 256 below N ~ 5k (N=4096: 1.82 ms against 1.86 at 384, 1.90 at 512), 384 while the panel chain dominates
// (N=8192: 5.61 ms against 5.84 at 256 and 5.82 at 512), 512 once the trailing updates do (N=16384: 30.2 ms against
// 30.9 at 384 and 31.2 at 640).
static inline int64_t outer_width(const gpt_ctx *c, int64_t n)
{
    if (c->nb_outer > 0) return c->nb_outer;
    // (round 2, with the helper stream at its measured rate: 640 against 512 gains 1 % at N = 16384, 0.5 % at 32768)
    return (n <= 5120) ? 256 : (n <= 12288) ? 384 : 640;
}

// Two 128-column leaves at once (potf2x2_trsm_kernel): columns [lc, lc + 256) of the n x n matrix, m = n - lc - 256 rows
// below; usable while the panel is short enough for the fused kernels (every workgroup takes a whole CU).
static inline bool leaf256_ok(const gpt_ctx *c, int64_t rows_below)
{
    return c->leaf256 && c->fuse_trsm > 0 && !c->use_graph && rows_below >= 0 && rows_below + 128 <= c->fuse_trsm;
}

static int leaf256_factor(gpt_ctx *c, hipStream_t st, double *Ad, int64_t lda, int64_t rows_below, double *ws,
                          int32_t *info, int64_t info_base, hipEvent_t done_ev)
{
    if (c->flag_epoch > 0x3fffff00u) {
        GPT_HIP_CHECK(hipMemsetAsync(c->d_flag, 0, 256, st)); c->x1_count = 0;
        c->flag_epoch = 0;
    }
    c->flag_epoch += 32;
    return launch_potf2x2_trsm(st, Ad, lda, ws, info, info_base, rows_below, c->d_l10pk, c->d_flag, c->flag_epoch, done_ev);
}

// ------------------------------------------------------------------------------------------------
// When may an evaluation run its look-ahead on flag edges (EdgeSig, common.hpp)?
// ------------------------------------------------------------------------------------------------
// A kernel that waits for a flag holds its hardware queue's slot.  With two evaluations in flight in one process (two
// contexts in two host threads: GaussianProcess.ll_batch, the `batched` leg of bench.py) the streams of both plus the
// runtime's own can exceed the hardware queues the firmware keeps resident; the queue of the kernel that would raise the
// flag is then scheduled out behind the waiter and every hand-over costs a scheduling quantum: measured 0.8 s per evaluation
// instead of 1.3 ms at N = 4096, intermittently (scratch/stress_flags.py).  Event edges do not spin and are immune.
// What decides is therefore not how many contexts EXIST (an idle context's streams cost nothing: GaussianProcess keeps a
// pooled second context for ll_batch, Kernel.__call__ a process-wide one) but how many evaluations are IN FLIGHT:
//   * an evaluation (EvalScope, below) takes the flag edges only if it is the only one in flight in the process when it
//     starts, nobody has announced concurrent evaluations (gpt_concurrency_hint: ll_batch and bench.py bracket their
//     threaded sections with it), the context owns its streams, and the process has not tripped a flag timeout;
//   * an evaluation that starts while a flag-mode evaluation is in flight first waits for that one to end, and the process
//     then stays on event edges while un-announced evaluations keep overlapping (every overlapping start re-arms a 100 ms
//     window; without it two un-hinted threads would take turns, each alone at its start and on flags, the other waiting:
//     one at a time).  Measured, two contexts in two threads at N = 4096: 990-1176 evaluations/s un-hinted, 1147-1168 hinted,
//     765 from one thread (scratch/two_thread_modes.py); a lone evaluation 100 ms later is back on flags;
//   * every wait is bounded (common.hpp); a timeout -- queues oversubscribed by ANOTHER process on the same GPU, a tool that
//     serialises kernels and is not recognised below -- ends the evaluation with an internal status, the process goes to
//     event edges for good (g_flags_tripped) and the evaluation is repeated (fit_terms / gpt_fit_matrix).
// Not usable at all: under rocprofv3 counter collection (ROCPROF_COUNTER_COLLECTION: one kernel at a time -- a --pmc pass
// once hung until the box's limit; with bounded waits it would crawl instead), with GPT_EDGE_FLAGS=0 (the documented switch
// for jobs that share a GPU between processes), under graph capture, on a caller-supplied stream (its other work is
// invisible to the accounting above).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
static std::mutex g_eval_mu;
static std::condition_variable g_eval_cv;
static int g_evals = 0;                 // evaluations in flight in this process
static int g_flag_evals = 0;            // ... of which on flag edges (0 or 1)
static int g_announced = 0;             // gpt_concurrency_hint depth
static std::atomic<bool> g_flags_tripped{false};
static std::chrono::steady_clock::time_point g_contention_until;   // (under g_eval_mu) overlapping evaluations seen recently: no flags before
#define GPT_I_EDGE_TIMEOUT (-100)       // internal status of an evaluation whose flag wait timed out (never leaves the library)

static bool edge_flags_env_ok()
{
    static int flags_ok = -1;
    if (flags_ok < 0) {
        const char *e = getenv("GPT_EDGE_FLAGS"), *r = getenv("ROCPROF_COUNTER_COLLECTION");
        flags_ok = !((e && atoi(e) == 0) || (r && r[0] && r[0] != '0' && r[0] != 'F' && r[0] != 'f'));
    }
    return flags_ok != 0;
}

// One synchronous evaluation (K build + factorisation + reduction, ends with the streams drained): decides c->flags_now.
// never_flags: work that is in flight like an evaluation but has no flag edges of its own (gpt_fit_batch, gpt_cov_sample): it
// is COUNTED (a flag-mode evaluation started meanwhile sees it and stays on events) but never takes the flag-mode slot itself
// -- holding it would make every other thread's evaluation wait out the whole batch in the constructor below.
struct EvalScope {
    gpt_ctx *c;
    bool flags;
    explicit EvalScope(gpt_ctx *c_, bool never_flags = false) : c(c_), flags(false)
    {
        std::unique_lock<std::mutex> lk(g_eval_mu);
        const auto now = std::chrono::steady_clock::now();
        if (never_flags) {
            // (a flag-mode evaluation must be alone: this work fills all 256 CUs from the unmasked panel stream, and kernels that
            // spin on flags beside it would run into their bounded waits, repeat the evaluation and put the process on event edges
            // for good -- so it WAITS for one in flight like any other evaluation; it only never takes the flag-mode slot.  ADVICE r4)
            if (g_flag_evals != 0) {
                g_contention_until = now + std::chrono::milliseconds(100);
                g_eval_cv.wait(lk, [] { return g_flag_evals == 0; });
            }
            if (g_evals != 0 && g_announced == 0) g_contention_until = now + std::chrono::milliseconds(100);
            g_evals++;
            c->flags_now = false;
            return;
        }
        if (g_flag_evals != 0) {
            // Somebody else's flag-mode evaluation is in flight: this process runs evaluations from several threads without
            // having said so (gpt_concurrency_hint).  Wait for that one -- a flag-mode evaluation must be alone -- and keep the
            // process on event edges for a while: otherwise two un-hinted threads take turns, each alone at its start, each on
            // flags, the other one waiting -- serialised (measured: 2 x 600 evaluations at N = 4096 in 1.6 s against 0.8 s for
            // 600).  On events they overlap (1.5 x one at a time).
            g_contention_until = now + std::chrono::milliseconds(100);
            g_eval_cv.wait(lk, [] { return g_flag_evals == 0; });
        }
        flags = edge_flags_env_ok() && !g_flags_tripped.load() && g_evals == 0 && g_announced == 0 && c->edge_flags &&
                c->own_stream && c->d_edge && !c->use_graph && now >= g_contention_until;
        // (un-announced overlap seen: stay on events for the next 100 ms; announced sections -- ll_batch, bench.py -- end with
        // their bracket and the next lone evaluation is back on flags at once)
        if (g_evals != 0 && g_announced == 0) g_contention_until = now + std::chrono::milliseconds(100);
        g_evals++;
        if (flags) g_flag_evals++;
        c->flags_now = flags;
    }
    ~EvalScope()
    {
        std::lock_guard<std::mutex> lk(g_eval_mu);
        g_evals--;
        if (flags) g_flag_evals--;
        c->flags_now = false;
        g_eval_cv.notify_all();
    }
};

extern "C" int gpt_concurrency_hint(int delta)
{
    std::lock_guard<std::mutex> lk(g_eval_mu);
    g_announced += delta;
    if (g_announced < 0) g_announced = 0;
    return g_announced;
}

// May a rank-k update of an m x n block wait for its flag INSIDE the kernel?  Its workgroups spin until the word is up, so
// the launch must not be able to fill the chip in front of the update it waits for: only launches that launch_gemm_nt cuts
// into 32x32 tiles (fewer than gemm_small_threshold() 64x64 tiles: 8 KB of LDS and 256 threads per workgroup, many fit
// beside a trailing update) and only up to 1024 of those workgroups (half the chip's wave slots).  A launch of 64x64 tiles
// never does: measured at N = 8192 with 750 such workgroups waiting in the kernel, 4.84 against 4.45 ms per evaluation.
// Everything else waits on the stream, in front of the launch (stream_wait_flag).
static bool gemm_may_wait_in_kernel(const gpt_ctx *c, int64_t m, int64_t n)
{
    if (!(c->tile == 0 || c->tile == 64)) return false;
    const int64_t nt64 = ((m + 63) / 64) * ((n + 63) / 64);
    return nt64 < gemm_small_threshold() && ((m + 31) / 32) * ((n + 31) / 32) <= 1024;
}
// the error word of the context's bounded flag waits
static inline EdgeSig with_err(gpt_ctx *c, EdgeSig e)
{
    static const bool unbounded = getenv("GPT_EDGE_UNBOUNDED") != nullptr;      // (measurement aid)
    e.err = unbounded ? nullptr : c->d_edge + 60;
    return e;
}
// stream-side wait: own bounded kernel, or (GPT_EDGE_WAITVALUE, measurement aid) the runtime's hipStreamWaitValue32
static int stream_wait_flag(hipStream_t st, EdgeSig w)
{
    static const bool rt = getenv("GPT_EDGE_WAITVALUE") != nullptr;
    if (rt) {
        GPT_HIP_CHECK(hipStreamWaitValue32(st, w.word, w.value, hipStreamWaitValueGte, 0xffffffffu));
        return GPT_OK;
    }
    return launch_wait_flag(st, w);
}

static int panel_rec(gpt_ctx *c, hipStream_t st, double *Ap, int64_t lda, int64_t m, int64_t w, double *invd,
                     int32_t *info, int64_t base)
{
    if (w == 256 && leaf256_ok(c, m - 256)) return leaf256_factor(c, st, Ap, lda, m - 256, invd, info, base, nullptr);
    if (w == 128) {
        const int64_t mb = m - 128;
        if (c->fuse_trsm > 0 && mb >= 128 && mb <= c->fuse_trsm && !c->use_graph) {
            // short panel (the head chunk of the row-chunked multi-GPU schedule, the last panels of a factorisation):
            // diagonal block and TRSM in one launch (potf2_trsm_kernel), as in panel_ext
            if (c->flag_epoch > 0x3fffff00u) {
                GPT_HIP_CHECK(hipMemsetAsync(c->d_flag, 0, 256, st)); c->x1_count = 0;
                c->flag_epoch = 0;
            }
            c->flag_epoch += 32;           // (the 256-column leaf kernel raises the word by up to 17 per launch)
            return launch_potf2_trsm(st, Ap, lda, invd, info, base, mb, c->d_flag, c->flag_epoch, nullptr, EdgeSig(), EdgeSig(),
                                     strip_rows_code(c, mb));
        }
        GPT_TRY(launch_potf2_diag(st, Ap, lda, invd, info, base));
        return launch_trsm_panel(st, m - 128, Ap, lda, invd, Ap + 128 * lda, lda);
    }
    const int64_t h = (w / 256) * 128 > 0 ? (w / 256) * 128 : 128;
    GPT_TRY(panel_rec(c, st, Ap, lda, m, h, invd, info, base));
    GPT_TRY(gemm_nt(c, st, m - h, w - h, h, -1.0, Ap + h * lda, lda, Ap + h * lda, lda, 1.0, Ap + h * lda + h, lda, 1));
    return panel_rec(c, st, Ap + h * lda + h, lda, m - h, w - h, invd + (h / 128) * GPT_WS_BLOCK, info, base + h);
}

// Look-ahead panel: block column [c0, c0+w) of the n x n matrix, right-looking in 128-column leaves.  Every leaf's
// rank-128 update also reaches the GPT_PANEL_EXT columns that follow the block column, so when the panel is done the
// first leaf of the NEXT block column is already up to date with respect to this one and its pivot chain can start
// without waiting for anybody.  `wait_ev` (the other stream's update of the columns this panel reads beyond its first
// leaf) is waited for before the first update; `done_ev` is recorded once L of the block column is final.
#define GPT_PANEL_EXT 128
static int leaf_factor(gpt_ctx *c, hipStream_t st, double *A, int64_t lda, int64_t n, int64_t lc, double *invd,
                       int32_t *info, hipEvent_t done_ev)
{
    double *Ad = A + lc * lda + lc;
    double *ws = invd + (lc / 128) * GPT_WS_BLOCK;
    const int64_t m = n - (lc + 128);
    if (c->fuse_trsm > 0 && m >= 128 && m <= c->fuse_trsm && !c->use_graph) {
        if (c->flag_epoch > 0x3fffff00u) {
            GPT_HIP_CHECK(hipMemsetAsync(c->d_flag, 0, 256, st)); c->x1_count = 0;
            c->flag_epoch = 0;
        }
        c->flag_epoch += 32;           // (the 256-column leaf kernel raises the word by up to 17 per launch)
        return launch_potf2_trsm(st, Ad, lda, ws, info, lc, m, c->d_flag, c->flag_epoch, done_ev, EdgeSig(), EdgeSig(), strip_rows_code(c, m));
    }
    GPT_TRY(launch_potf2_diag(st, Ad, lda, ws, info, lc));
    GPT_TRY(launch_trsm_panel(st, m, Ad, lda, ws, Ad + 128 * lda, lda, (done_ev && !c->use_graph) ? done_ev : nullptr));
    if (done_ev && c->use_graph) GPT_HIP_CHECK(hipEventRecord(done_ev, st));
    return GPT_OK;
}

// The same look-ahead panel, LEFT-looking inside the block column (option "inner" = 1): leaf j first receives the
// update of the leaves 0..j-1 of this block column in ONE launch (k = 128 j, 128 columns wide), then is factored; after
// the last leaf the 128 columns that follow the block column (the next panel's first leaf) get the whole block column's
// update (k = w).  Same flops as the right-looking form, but every update launch on the chain is 128 columns wide --
// the right-looking form's first leaf carries a (w + 128 - 128)-column update that shares the chip with the main stream's
// trailing update and measured 60 us against 13-16 us for the narrow ones (profiles/r01_timeline_c3_N8192.txt).
static int panel_ext_ll(gpt_ctx *c, hipStream_t st, double *A, int64_t lda, int64_t n, int64_t c0, int64_t w,
                        double *invd, int32_t *info, hipEvent_t wait_ev, hipEvent_t done_ev)
{
    bool waited = (wait_ev == nullptr);
    for (int64_t lc = c0; lc < c0 + w; lc += 128) {
        const int64_t kk = lc - c0;
        if (kk > 0) {
            if (!waited) {
                GPT_HIP_CHECK(hipStreamWaitEvent(st, wait_ev, 0));
                waited = true;
            }
            GPT_TRY(gemm_nt(c, st, n - lc, 128, kk, -1.0, A + lc * lda + c0, lda, A + lc * lda + c0, lda, 1.0,
                            A + lc * lda + lc, lda, 1));
        }
        const bool last = (lc + 128 == c0 + w);
        GPT_TRY(leaf_factor(c, st, A, lda, n, lc, invd, info, last ? done_ev : nullptr));
    }
    const int64_t e0 = c0 + w;
    if (e0 < n) {
        if (!waited) GPT_HIP_CHECK(hipStreamWaitEvent(st, wait_ev, 0));
        const int64_t ew = (n - e0 < GPT_PANEL_EXT) ? n - e0 : GPT_PANEL_EXT;
        GPT_TRY(gemm_nt(c, st, n - e0, ew, w, -1.0, A + e0 * lda + c0, lda, A + e0 * lda + c0, lda, 1.0,
                        A + e0 * lda + e0, lda, 1));
    }
    return GPT_OK;
}

// `ext`: how many columns past the block column every leaf update reaches (GPT_PANEL_EXT, or 256 when the NEXT panel starts
// with a 256-column leaf: that kernel reads both of its leaves' columns when it starts).
static int panel_ext(gpt_ctx *c, hipStream_t st, double *A, int64_t lda, int64_t n, int64_t c0, int64_t w,
                     double *invd, int32_t *info, hipEvent_t wait_ev, hipEvent_t done_ev, int64_t ext = GPT_PANEL_EXT,
                     hipEvent_t first_ev = nullptr, EdgeSig wait_edge = EdgeSig(), EdgeSig done_edge = EdgeSig())
{
    // wait_edge / done_edge: the same two dependencies as wait_ev / done_ev carried by flag words (EdgeSig, common.hpp)
    // first_ev: recorded by the panel's FIRST update launch (see "deferred rest" in potrf_enqueue)
    if (c->inner == 1 || (c->inner == 2 && n - c0 <= c->inner_rows))
        return panel_ext_ll(c, st, A, lda, n, c0, w, invd, info, wait_ev, done_ev);
    const int64_t cend = (c0 + w + ext < n) ? c0 + w + ext : n;
    for (int64_t lc = c0; lc < c0 + w; lc += 128) {
        double *Ad = A + lc * lda + lc;
        double *ws = invd + (lc / 128) * GPT_WS_BLOCK;
        if (lc + 256 <= c0 + w && c->inner == 0 && leaf256_ok(c, n - lc - 256)) {
            // two leaves in one launch, then ONE rank-256 update of what follows inside the panel + its extension
            const int64_t r2 = lc + 256;
            const bool last2 = (r2 == c0 + w) && done_ev;
            GPT_TRY(leaf256_factor(c, st, Ad, lda, n - r2, ws, info, lc, last2 ? done_ev : nullptr));
            if (lc == c0 && wait_ev) GPT_HIP_CHECK(hipStreamWaitEvent(st, wait_ev, 0));
            if (cend > r2) {
                GPT_TRY(gemm_nt(c, st, n - r2, cend - r2, 256, -1.0, A + r2 * lda + lc, lda, A + r2 * lda + lc, lda, 1.0,
                                A + r2 * lda + r2, lda, 1, first_ev));
                first_ev = nullptr;
            }
            lc += 128;
            continue;
        }
        const int64_t r1 = lc + 128;
        const bool last = (r1 == c0 + w) && done_ev;
        const int64_t m = n - r1;
        EdgeSig fw;                                   // the first leaf of a factorisation may have to wait for the K build's head
        if (lc == 0 && c->first_wait.word) {
            fw = c->first_wait;
            c->first_wait = EdgeSig();
            // Inside the leaf's kernel only while that launch fits the CUs reserved for the panel stream: every workgroup of
            // the fused kernel holds a whole CU (135 KB of LDS) while it spins, and beyond the reserved CUs they would be
            // taken from the K build the launch is waiting for.  Otherwise a one-wave wait kernel in front of it.
            const bool fused = c->fuse_trsm > 0 && m >= 128 && m <= c->fuse_trsm && !c->use_graph;
            const bool r64 = fused && ((c->fuse_upd && m <= c->fuse_upd_rows && !first_ev) || m <= c->fuse_rows64);      // (64 rows per workgroup)
            const int64_t wgs = (fused && strip_rows_code(c, m) > 1) ? 1 + m / (128 >> strip_rows_code(c, m)) : r64 ? 1 + m / 64 : fused ? 1 + (m + 127) / 128 : 1;
            if (wgs > c->head_wait_wgs) {
                GPT_TRY(stream_wait_flag(st, fw));
                fw = EdgeSig();
            }
        }
        int64_t upd_done = 0;                         // columns [r1, r1 + upd_done) updated inside the leaf's launch
        if (c->fuse_trsm > 0 && m >= 128 && m <= c->fuse_trsm && !c->use_graph) {
            // short panel: diagonal block and TRSM in one launch, the substitution trailing the pivots (potrf.hip)
            if (c->flag_epoch > 0x3fffff00u) {
                GPT_HIP_CHECK(hipMemsetAsync(c->d_flag, 0, 256, st)); c->x1_count = 0;
                c->flag_epoch = 0;
            }
            c->flag_epoch += 32;           // (the 256-column leaf kernel raises the word by up to 17 per launch)
            if (c->fuse_upd && m <= c->fuse_upd_rows && cend > r1 && !first_ev) {
                // ... and the leaf's update of the next 128 / 256 columns as well (potf2_trsm_upd_kernel): the chain-bound end of
                // the factorisation.  What those columns wait for -- the main stream's update of the columns this panel touches
                // -- is awaited by the leaf's launch: inside the kernel, right before the accumulators are loaded (the word is
                // long up there: the main stream is ahead of the chain), or in front of the launch when the edge is an event.
                EdgeSig cw;
                if (lc == c0 && wait_ev) GPT_HIP_CHECK(hipStreamWaitEvent(st, wait_ev, 0));
                if (lc == c0 && wait_edge.word) cw = wait_edge;
                upd_done = (cend - r1 >= 256) ? 256 : 128;
                GPT_TRY(launch_potf2_trsm_upd(st, Ad, lda, ws, info, lc, m, c->d_flag, c->flag_epoch, c->x1_count, upd_done,
                                              last ? done_ev : nullptr, (r1 == c0 + w) ? done_edge : EdgeSig(), fw, cw));
                c->x1_count += (unsigned)(upd_done / 16);
            } else {
                GPT_TRY(launch_potf2_trsm(st, Ad, lda, ws, info, lc, m, c->d_flag, c->flag_epoch, last ? done_ev : nullptr,
                                          (r1 == c0 + w) ? done_edge : EdgeSig(), fw, strip_rows_code(c, m)));
            }
        } else {
            GPT_TRY(launch_potf2_diag(st, Ad, lda, ws, info, lc, fw));
            // (a stop event on the launch is not recorded by stream capture: under a graph use a plain record)
            GPT_TRY(launch_trsm_panel(st, m, Ad, lda, ws, Ad + 128 * lda, lda, (last && !c->use_graph) ? done_ev : nullptr,
                                      (r1 == c0 + w) ? done_edge : EdgeSig()));
            if (last && c->use_graph) GPT_HIP_CHECK(hipEventRecord(done_ev, st));
        }
        if (upd_done > 0) {
            // the columns beyond the in-launch update (a 384-wide panel's first leaf): one narrower launch, rows from there on
            const int64_t rr = r1 + upd_done;
            if (cend > rr)
                GPT_TRY(gemm_nt(c, st, n - rr, cend - rr, 128, -1.0, A + rr * lda + lc, lda, A + rr * lda + lc, lda, 1.0,
                                A + rr * lda + rr, lda, 1));
            continue;
        }
        if (lc == c0 && wait_ev) GPT_HIP_CHECK(hipStreamWaitEvent(st, wait_ev, 0));
        if (cend > r1) {
            // (the first update of the panel is the first kernel of the chain that touches what the main stream's urgent
            // update wrote: it waits for that edge itself, see gemm.hip)
            // Its workgroups SPIN until the word is up, so the launch must not be able to fill the chip (the update they
            // wait for needs room to run): in the kernel only while the launch stays under 1024 workgroups = half the wave
            // slots, otherwise as a stream operation in front of it (a kernel of the runtime, ~5 us).
            EdgeSig inwait;
            if (lc == c0 && wait_edge.word) {
                if (gemm_may_wait_in_kernel(c, n - r1, cend - r1)) inwait = wait_edge;
                else GPT_TRY(stream_wait_flag(st, wait_edge));
            }
            GPT_TRY(gemm_nt(c, st, n - r1, cend - r1, 128, -1.0, A + r1 * lda + lc, lda, A + r1 * lda + lc, lda, 1.0,
                            A + r1 * lda + r1, lda, 1, first_ev, EdgeSig(), inwait));
            first_ev = nullptr;
        }
    }
    return GPT_OK;
}

static int potrf_enqueue(gpt_ctx *c, int64_t n, double *A, int64_t lda, double *invd, int32_t *info)
{
    if (n % 128) {
        gpt_set_error("potrf: n must be a multiple of 128 (n=%lld)", (long long)n);
        return GPT_E_ARG;
    }
    const int64_t nbo = outer_width(c, n);
    const int64_t nblk = (n + nbo - 1) / nbo;
    hipStream_t S = c->stream, P = c->panel_stream;
    const bool la = c->lookahead && nblk > 1;
    hipEvent_t head = c->head_event;
    c->head_event = nullptr;
    const EdgeSig head_wait = c->head_wait;                   // (the K build's head columns as a flag word, see fit_terms)
    c->head_wait = EdgeSig();
    c->first_wait = EdgeSig();
    if (!la) {
        for (int64_t k = 0; k < nblk; k++) {
            const int64_t c0 = k * nbo, w = (n - c0 < nbo) ? n - c0 : nbo, m = n - c0;
            GPT_TRY(panel_rec(c, S, A + c0 * lda + c0, lda, m, w, invd + (c0 / 128) * GPT_WS_BLOCK, info, c0));
            const int64_t r0 = c0 + w;
            if (r0 < n)
                GPT_TRY(gemm_nt(c, S, n - r0, n - r0, w, -1.0, A + r0 * lda + c0, lda, A + r0 * lda + c0, lda, 1.0,
                                A + r0 * lda + r0, lda, 1));
        }
        return GPT_OK;
    }
    // ---- look-ahead.  P (high priority, reserved CUs) runs the latency-bound chain: the panels, each extended by
    // GPT_PANEL_EXT columns (panel_ext).  S applies panel k (rank nbo) to everything right of column
    // u0 = c0 + nbo + EXT in two launches: the nbo columns panel k+1 is going to touch first ("urgent", event e_cu),
    // then the rest.  P never waits for a large update: its only dependency is e_cu(k) before the first update of
    // panel k+1, and S has the whole first pivot block + TRSM of that panel to get there.
    if (head_wait.word && c->inner == 0 && !c->leaf256) {
        // the first leaf's kernel waits for the word itself (potrf.hip: edge_wait): no event edge in front of the chain
        c->first_wait = head_wait;
    } else {
        if (!head) {
            head = get_event(c, 0);
            if (!head) return GPT_E_HIP;
            GPT_HIP_CHECK(hipEventRecord(head, S));
        }
        GPT_HIP_CHECK(hipStreamWaitEvent(P, head, 0));
    }
    // Panel widths: nbo; optionally ("ramp") 128, 256, ... at the start so that the main stream gets its first update
    // after one leaf instead of after a whole panel -- measured slightly slower (N=8192: 5.95 against 5.90 ms,
    // N=16384: 31.4 against 31.2: the rank-128/256 updates it adds are inefficient), so it is off by default.
    std::vector<int64_t> widths;
    {
        int64_t c0 = 0, w = c->ramp ? 128 : nbo;
        while (c0 < n) {
            // (option nb_early: wider panels while more than nb_switch_rows rows remain -- the update-bound head of the
            // factorisation -- and nbo in the chain-bound rest)
            const int64_t cap = (c->nb_early > 0 && n - c0 > c->nb_switch_rows) ? c->nb_early : nbo;
            if (w > cap || (!c->ramp)) w = cap;
            if (w > n - c0) w = n - c0;
            widths.push_back(w);
            c0 += w;
            w += 128;
        }
    }
    // Helper: while the trailing updates dominate, the panel stream leaves its reserved CUs idle most of the time.
    // The bottom-right triangle [s, n)^2 of the rank-w update of panel k therefore runs on H, a stream masked to
    // those CUs (minus 8 that stay free for the diagonal-block kernel), concurrently with S's share; the split is
    // sized by the two streams' rates so that they finish together.  The split column only moves right: a helper
    // region lies inside the previous one, so H needs nothing but "panel k is final"; S waits for helper k before
    // it next touches columns >= s.
    // (only where the updates dominate: at n = 8192 the helper costs 0.5-1 %, at 16384 / 32768 it gains 2 / 3 %)
    hipStream_t H = (c->helper_stream && !c->use_graph && c->helper_tf > 0 && n > c->helper_min_n) ? c->helper_stream : nullptr;
    const double rate_s = 46e12, rate_h = 1e11 * (double)c->helper_tf * (double)c->helper_cus / 24.0;
    hipEvent_t e_cu_prev = nullptr, e_help_prev = nullptr, e_rest_prev = nullptr;
    int64_t c0 = 0, s_prev = 0;
    // Update-bound head (option early_rows): while more than early_rows rows remain, the trailing updates run on a
    // stream that leaves only a few CUs to the panel stream (the diagonal-block kernel needs ONE free CU; the fused
    // diagonal-block + TRSM kernel of the chain-bound end needs up to 33).  One event hands over between the two.
    const bool use_early = c->early_stream && c->early_rows > 0 && !H && !c->use_graph;
    hipStream_t S0 = S, S_cur = S;
    if (use_early) {
        hipEvent_t e_k = get_event(c, 2 + 4 * widths.size());
        if (!e_k) return GPT_E_HIP;
        GPT_HIP_CHECK(hipEventRecord(e_k, S0));                     // the K build (and everything before) on the main stream
        GPT_HIP_CHECK(hipStreamWaitEvent(c->early_stream, e_k, 0));
    }
    // Deferred rest (chain-bound end of the factorisation, option defer_rows).  There the panel stream sets the pace and the
    // main stream has slack, yet its large "rest" update of panel k used to start right after the urgent one and shared
    // the chip with the panel stream's first leaf update of panel k+1 -- which is ON the chain and ran 60 us instead of
    // ~17 (profiles/r01_timeline_c3_N8192.txt).  The rest of panel k is therefore held back until that leaf update is
    // done (one event), and is enqueued one loop iteration late so that the event is recorded before it is waited for.
    const bool use_late = c->late_panel_stream && c->late_rows > 0 && !c->use_graph;
    // Flag edges (EdgeSig): "panel k is final" (panel stream -> main stream) and "the urgent update of panel k is done"
    // (main -> panel stream) are raised by the last workgroup of the kernel that completes them and waited for with
    // hipStreamWaitValue32 -- 1.5 us per edge against 8-9 for an event, and no stop event on the chain's kernels (4.5 us
    // each).  In the chain-bound end both edges are on the critical path of every panel.
    // (tile: the edge flags live in the 64x64 / 32x32 GEMM kernels only)
    // (not with the helper stream: flags + helper -- the helper waiting behind a wait kernel, the merged launch cut at the
    // split column, code below -- measured the same as events + helper at N = 16384 (27.65 against 27.49 ms), and the helper
    // in the head of an N = 8192 factorisation still loses, 4.64 against 4.45 ms: round 3)
    const bool use_flags = c->flags_now && !H && c->inner == 0 && !c->leaf256 && !use_early && !use_late
                           && c->defer_rows == 0 && (c->tile == 0 || c->tile == 64);
    if (use_flags && c->edge_seq > 0xf0000000u && !head_wait.word) {           // (the words are only ever raised: start over long before a wrap)
        GPT_HIP_CHECK(hipStreamSynchronize(S));
        GPT_HIP_CHECK(hipStreamSynchronize(P));
        GPT_HIP_CHECK(hipMemsetAsync(c->d_edge, 0, 256, S));
        GPT_HIP_CHECK(hipStreamSynchronize(S));
        c->edge_seq = 0;
        c->alpha_counter = 0;                  // (the step counter of the wide back-substitution lives in the same words)
    }
    EdgeSig cu_edge_prev, rest_edge_prev;
    EdgeSig next_panel_edge;      // the NEXT panel's edge, allocated early: the main stream's last launch of this panel awaits it at its end
    bool pair_pending = false, near_synced = false;      // panel pairs (option pair_rows, see below)
    int64_t pair_c0 = 0, pair_w = 0;
    struct PendingRest { bool on; int64_t c0, w, u1, split; hipStream_t S; } pend = {false, 0, 0, 0, 0, nullptr};
    auto launch_rest = [&](const PendingRest &r) -> int {
        return gemm_nt(c, r.S, n - r.u1, r.split - r.u1, r.w, -1.0, A + r.u1 * lda + r.c0, lda, A + r.u1 * lda + r.c0, lda,
                       1.0, A + r.u1 * lda + r.u1, lda, 1);
    };
    for (size_t k = 0; k < widths.size(); k++) {
        const int64_t w = widths[k];
        if (use_early) {
            hipStream_t want = (n - c0 > c->early_rows) ? c->early_stream : S0;
            if (want != S_cur) {
                hipEvent_t e_sw = get_event(c, 3 + 4 * widths.size());
                if (!e_sw) return GPT_E_HIP;
                GPT_HIP_CHECK(hipEventRecord(e_sw, S_cur));
                GPT_HIP_CHECK(hipStreamWaitEvent(want, e_sw, 0));
                S_cur = want;
            }
            S = S_cur;
        }
        const int64_t wn = (k + 1 < widths.size()) ? widths[k + 1] : 0;      // width of the next panel
        hipEvent_t e_panel = get_event(c, 2 + 4 * k), e_cu = get_event(c, 3 + 4 * k), e_help = get_event(c, 4 + 4 * k);
        hipEvent_t e_sdone = get_event(c, 5 + 4 * k);
        if (!e_panel || !e_cu || !e_help || !e_sdone) return GPT_E_HIP;
        // reach of the leaf updates past a panel: 256 columns if the panel that follows starts with a 256-column leaf
        auto ext_after = [&](int64_t cstart, int64_t wnext) -> int64_t {
            return (wnext >= 256 && c->inner == 0 && leaf256_ok(c, n - cstart - 256)) ? 256 : GPT_PANEL_EXT;
        };
        const int64_t ext_k = ext_after(c0 + w, wn);
        const int64_t wnn = (k + 2 < widths.size()) ? widths[k + 2] : 0;
        const int64_t ext_k1 = ext_after(c0 + w + wn, wnn);
        if (use_late && P == c->panel_stream && n - c0 <= c->late_rows) {
            hipEvent_t e_pl = get_event(c, 6 + 4 * widths.size());
            if (!e_pl) return GPT_E_HIP;
            GPT_HIP_CHECK(hipEventRecord(e_pl, P));
            GPT_HIP_CHECK(hipStreamWaitEvent(c->late_panel_stream, e_pl, 0));
            P = c->late_panel_stream;
        }
        c->pad_now = (c->late_pad > 0 && n - c0 <= c->late_pad_rows) ? c->late_pad : 0;
        hipEvent_t e_first = pend.on ? get_event(c, 8 + 4 * widths.size() + k) : nullptr;
        if (pend.on && !e_first) return GPT_E_HIP;
        EdgeSig panel_edge;
        bool panel_awaited = false;      // (by the tail wait of the previous panel's last main-stream launch)
        if (use_flags && c0 + w + ext_k < n) {
            if (next_panel_edge.word) {
                panel_edge = next_panel_edge;
                panel_awaited = true;
            } else {
                panel_edge.word = c->d_edge;
                panel_edge.value = ++c->edge_seq;
                panel_edge = with_err(c, panel_edge);
            }
        }
        next_panel_edge = EdgeSig();
        GPT_TRY(panel_ext(c, P, A, lda, n, c0, w, invd, info, e_cu_prev, use_flags ? nullptr : e_panel, ext_k, e_first, cu_edge_prev,
                          panel_edge));
        cu_edge_prev = EdgeSig();
        if (pend.on) {
            GPT_HIP_CHECK(hipStreamWaitEvent(pend.S, e_first, 0));
            GPT_TRY(launch_rest(pend));
            pend.on = false;
        }
        e_cu_prev = nullptr;
        const int64_t u0 = c0 + w + ext_k;
        if (u0 < n) {
            // urgent: the columns panel k+1 touches beyond what panel k's own leaf updates reached,
            // [c0' + ext_k, c0' + w' + ext_k+1)
            const int64_t u1 = (c0 + w + wn + ext_k1 < n) ? c0 + w + wn + ext_k1 : n;
            int64_t split = n;                                        // S takes columns [u1, split), H [split, n)
            if (H && u1 < n) {
                const double side0 = (double)(n - u1);
                int64_t side = (int64_t)(sqrt(rate_h / (rate_s + rate_h)) * side0) / 64 * 64;
                if (n - side < s_prev) side = n - s_prev;
                // worth a launch only while the update is large (>= ~10 GFLOP) and the slice a real triangle
                if (side >= 1024 && n - side > u1 && (double)w * side0 * side0 >= 1e10) split = n - side;
            }
            if (split < n && !e_help_prev) {
                // first helper (or first after a gap): its region was last written by S (K build / a whole update)
                GPT_HIP_CHECK(hipEventRecord(e_sdone, S));
                GPT_HIP_CHECK(hipStreamWaitEvent(H, e_sdone, 0));
            }
            if (use_flags) {
                if (!panel_awaited) GPT_TRY(stream_wait_flag(S, panel_edge));
            } else {
                GPT_HIP_CHECK(hipStreamWaitEvent(S, e_panel, 0));
            }
            // Tail wait: the last launch of this panel on the main stream ends only when the NEXT panel's flag is up, so the
            // next panel's launch follows it without a wait kernel in between (the flag's value is fixed here, one panel early).
            EdgeSig tail;
            if (use_flags && c->tail_wait && split == n && k + 1 < widths.size() && c0 + w + wn + ext_k1 < n) {
                tail.word = c->d_edge;
                tail.value = ++c->edge_seq;
                tail = with_err(c, tail);
            }
            if (split < n) {
                if (use_flags) GPT_TRY(stream_wait_flag(H, panel_edge));
                else GPT_HIP_CHECK(hipStreamWaitEvent(H, e_panel, 0));
                GPT_TRY(gemm_nt(c, H, n - split, n - split, w, -1.0, A + split * lda + c0, lda, A + split * lda + c0,
                                lda, 1.0, A + split * lda + split, lda, 1, e_help));
            }
            bool waited = false;
            if (e_help_prev && u1 > s_prev) {
                GPT_HIP_CHECK(hipStreamWaitEvent(S, e_help_prev, 0));
                waited = true;
            }
            // Update-bound head (option purg_rows): the PANEL stream applies panel k to the columns panel k+1 touches
            // ("urgent") itself, right behind the panel -- no event round trip panel -> main -> panel on the chain and
            // one launch less on the main stream, which sets the pace there.  Both streams read-modify-write those
            // columns (the main stream's rest of panel k-1 covers them too), so the panel stream waits for that rest.
            // (with flag edges and merged launches the main stream has one launch per panel anyway and the urgent tiles run at
            // the large launch's rate: the panel stream's own urgent update no longer pays -- 4.444 against 4.492 ms at N = 8192)
            const int64_t purg_eff = (use_flags && c->merge_urgent) ? c->purg_rows_flags : c->purg_rows;
            const bool p_urgent = !H && purg_eff > 0 && n - c0 > purg_eff && !c->use_graph && u1 < n;
            if (p_urgent) {
                // (edge_flags >= 2: "the rest of panel k-1 is done" as a flag edge as well -- no stop event on the main
                // stream's large launches, which set the pace here; the whole C of that launch is then written through)
                const bool rest_flag = use_flags && c->edge_flags >= 2;
                EdgeSig rwait;
                if (e_rest_prev) GPT_HIP_CHECK(hipStreamWaitEvent(P, e_rest_prev, 0));
                if (!e_rest_prev && !rest_edge_prev.word) {
                    // First update the PANEL stream applies beyond the head columns: whatever the main stream still has in
                    // flight on those columns -- the rest of the K build of this evaluation -- must be through.  (Found with the
                    // `ramp` option, whose 128-wide first panel is done before the K build is: "8064-th leading minor not
                    // positive definite"; with 384-wide panels the build happened to finish first.)
                    hipEvent_t e_k = get_event(c, 9 + 5 * widths.size());
                    if (!e_k) return GPT_E_HIP;
                    GPT_HIP_CHECK(hipEventRecord(e_k, S));
                    GPT_HIP_CHECK(hipStreamWaitEvent(P, e_k, 0));
                }
                if (rest_edge_prev.word) {
                    if (gemm_may_wait_in_kernel(c, n - u0, u1 - u0)) rwait = rest_edge_prev;
                    else GPT_TRY(stream_wait_flag(P, rest_edge_prev));
                }
                GPT_TRY(gemm_nt(c, P, n - u0, u1 - u0, w, -1.0, A + u0 * lda + c0, lda, A + u0 * lda + c0, lda, 1.0,
                                A + u0 * lda + u0, lda, 1, nullptr, EdgeSig(), rwait));
                e_cu_prev = nullptr;
                rest_edge_prev = EdgeSig();
                if (rest_flag) {
                    EdgeSig re;
                    re.word = c->d_edge + 32;
                    re.value = ++c->edge_seq;
                    re = with_err(c, re);
                    GPT_TRY(gemm_nt(c, S, n - u1, n - u1, w, -1.0, A + u1 * lda + c0, lda, A + u1 * lda + c0, lda, 1.0,
                                    A + u1 * lda + u1, lda, 1, nullptr, re));
                    rest_edge_prev = re;
                    e_rest_prev = nullptr;
                    c0 += w;
                    continue;
                }
                hipEvent_t e_rest = get_event(c, 10 + 5 * widths.size() + k);
                if (!e_rest) return GPT_E_HIP;
                GPT_TRY(gemm_nt(c, S, n - u1, n - u1, w, -1.0, A + u1 * lda + c0, lda, A + u1 * lda + c0, lda, 1.0,
                                A + u1 * lda + u1, lda, 1, e_rest));
                e_rest_prev = e_rest;
                c0 += w;
                continue;
            }
            e_rest_prev = nullptr;
            rest_edge_prev = EdgeSig();
            if (use_flags) {
                EdgeSig cu_edge;
                cu_edge.word = c->d_edge + 16;
                cu_edge.value = ++c->edge_seq;
                cu_edge = with_err(c, cu_edge);
                // ---- panel pairs (round 5, option pair_rows): the trailing update with k = 2 w.  A rank-384 update spends ~20 % of
                // a 64x64 tile's time in the C prologue / epilogue and the launch has a fixed cost of ~20 us; alone on the chip the
                // same kernel runs at 48.2 / 51.8 / 53.5 TFLOP/s for k = 384 / 768 / 1152 (NOTES_r03).  First panel of a pair: only
                // the columns the second panel touches get their rank-w update now -- on near_stream, a second main stream, because
                // the main stream is still busy with the previous pair's large launch and this small one is on the chain.  Second
                // panel: everything to the right of it gets both panels' updates in ONE launch with k = 2 w; its urgent columns --
                // first in the launch, partial flag -- are the next TWO panels', so that the next pair's near update (which
                // read-modify-writes the second one's columns on the other stream) comes after them by way of the panel stream.
                // Every element still sums its products in the same order (k ascending, the accumulator carried through one
                // store and load between the launches of the unpaired schedule): the factor is bit-identical.
                if (pair_pending) {
                    const int64_t K2 = pair_w + w;
                    const int64_t u1p = (c0 + w + wn + wnn + GPT_PANEL_EXT < n) ? c0 + w + wn + wnn + GPT_PANEL_EXT : n;
                    const int64_t mtn = (n - u0 + 63) / 64;
                    const bool one = c->merge_urgent && u1p < n && mtn * (mtn + 1) / 2 >= c->merge_min_tiles && (u1p - u0) % 64 == 0;
                    if (one) {
                        GPT_TRY(gemm_nt(c, S, n - u0, n - u0, K2, -1.0, A + u0 * lda + pair_c0, lda, A + u0 * lda + pair_c0, lda, 1.0,
                                        A + u0 * lda + u0, lda, 1, nullptr, cu_edge, EdgeSig(), u1p - u0));
                    } else {
                        GPT_TRY(gemm_nt(c, S, n - u0, u1p - u0, K2, -1.0, A + u0 * lda + pair_c0, lda, A + u0 * lda + pair_c0, lda, 1.0,
                                        A + u0 * lda + u0, lda, 1, nullptr, cu_edge));
                        if (u1p < n)
                            GPT_TRY(gemm_nt(c, S, n - u1p, n - u1p, K2, -1.0, A + u1p * lda + pair_c0, lda, A + u1p * lda + pair_c0, lda,
                                            1.0, A + u1p * lda + u1p, lda, 1));
                    }
                    cu_edge_prev = cu_edge;
                    pair_pending = false;
                    c0 += w;
                    continue;
                }
                if (c->pair_rows > 0 && n - c0 > c->pair_rows && k + 1 < widths.size() && u1 < n && !tail.word && !panel_awaited) {
                    if (!c->near_stream) {
                        // (created on first use: every stream of a context costs, see gpt_ctx_create)
                        std::vector<uint32_t> mm;
                        int ncu = 0;
                        if (c->reserve_cus > 0 && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess) {
                            mm.assign((ncu + 31) / 32, 0u);
                            for (int i = c->reserve_cus; i < ncu; i++) mm[i / 32] |= (1u << (i % 32));
                        }
                        if (mm.empty() || hipExtStreamCreateWithCUMask(&c->near_stream, (uint32_t)mm.size(), mm.data()) != hipSuccess) {
                            (void)hipGetLastError();
                            GPT_HIP_CHECK(hipStreamCreateWithFlags(&c->near_stream, hipStreamNonBlocking));
                        }
                    }
                    if (!near_synced) {
                        // the near stream's first launch of this factorisation: behind everything the main stream has enqueued so far
                        // (the rest of the K build)
                        hipEvent_t e_k = get_event(c, 12 + 6 * widths.size());
                        if (!e_k) return GPT_E_HIP;
                        GPT_HIP_CHECK(hipEventRecord(e_k, S));
                        GPT_HIP_CHECK(hipStreamWaitEvent(c->near_stream, e_k, 0));
                        near_synced = true;
                    }
                    // (the main stream was made to wait for this panel above -- harmless: its next launch needs the next panel anyway)
                    GPT_TRY(stream_wait_flag(c->near_stream, panel_edge));
                    GPT_TRY(gemm_nt(c, c->near_stream, n - u0, u1 - u0, w, -1.0, A + u0 * lda + c0, lda, A + u0 * lda + c0, lda, 1.0,
                                    A + u0 * lda + u0, lda, 1, nullptr, cu_edge));
                    cu_edge_prev = cu_edge;
                    pair_pending = true;
                    pair_c0 = c0;
                    pair_w = w;
                    c0 += w;
                    continue;
                }
                // Urgent + rest as ONE launch (option merge_urgent) while the update is large enough for an order table: the
                // tiles of the urgent columns come first on every XCD, write through and raise the flag when THEY are done;
                // the rest follows in the same launch -- one drain and one ramp-up less per panel on the main stream, and
                // the small urgent launch (30 TFLOP/s on its own) runs at the large launch's rate.
                // (with a helper slice the merged launch is the lower trapezoid of the columns [u0, split))
                const int64_t mtn = (split - u0 + 63) / 64, mtm = (n - u0 + 63) / 64;
                const int64_t nt64m = mtn * (mtn + 1) / 2 + (mtm - mtn) * mtn;
                if (c->merge_urgent && u1 < split && nt64m >= c->merge_min_tiles && (c->tile == 0 || c->tile == 64) && (u1 - u0) % 64 == 0) {
                    if (e_help_prev && !waited) GPT_HIP_CHECK(hipStreamWaitEvent(S, e_help_prev, 0));
                    e_help_prev = nullptr;
                    GPT_TRY(gemm_nt(c, S, n - u0, split - u0, w, -1.0, A + u0 * lda + c0, lda, A + u0 * lda + c0, lda, 1.0,
                                    A + u0 * lda + u0, lda, 1, nullptr, cu_edge, EdgeSig(), u1 - u0, tail));
                    next_panel_edge = tail;
                    cu_edge_prev = cu_edge;
                    if (split < n) {
                        s_prev = split;
                        e_help_prev = e_help;
                    }
                    c0 += w;
                    continue;
                }
                GPT_TRY(gemm_nt(c, S, n - u0, u1 - u0, w, -1.0, A + u0 * lda + c0, lda, A + u0 * lda + c0, lda, 1.0,
                                A + u0 * lda + u0, lda, 1, nullptr, cu_edge));
                cu_edge_prev = cu_edge;
            } else {
                GPT_TRY(gemm_nt(c, S, n - u0, u1 - u0, w, -1.0, A + u0 * lda + c0, lda, A + u0 * lda + c0, lda, 1.0,
                                A + u0 * lda + u0, lda, 1, e_cu));
                e_cu_prev = e_cu;
            }
            if (u1 < n) {
                if (e_help_prev && !waited) GPT_HIP_CHECK(hipStreamWaitEvent(S, e_help_prev, 0));
                e_help_prev = nullptr;
                if (!H && c->defer_rows > 0 && n - c0 <= c->defer_rows && k + 1 < widths.size() && !c->use_graph) {
                    pend = PendingRest{true, c0, w, u1, split, S};
                } else {
                    // (tail wait on the 64x64 / 32x32 kernels only; an event may not ride on the same launch)
                    const bool tl = tail.word && (c->tile == 0 || c->tile == 64);
                    GPT_TRY(gemm_nt(c, S, n - u1, split - u1, w, -1.0, A + u1 * lda + c0, lda, A + u1 * lda + c0, lda, 1.0,
                                    A + u1 * lda + u1, lda, 1, nullptr, EdgeSig(), EdgeSig(), 0, tl ? tail : EdgeSig()));
                    if (tl) next_panel_edge = tail;
                }
                if (split < n) {
                    s_prev = split;
                    e_help_prev = e_help;
                }
            }
        }
        c0 += w;
    }
    if (pend.on) GPT_TRY(launch_rest(pend));
    if (near_synced) {
        // (every near launch was awaited by the panel stream through its flag; the join keeps the stream's work inside the evaluation)
        hipEvent_t e_n = get_event(c, 13 + 6 * widths.size());
        if (!e_n) return GPT_E_HIP;
        GPT_HIP_CHECK(hipEventRecord(e_n, c->near_stream));
        GPT_HIP_CHECK(hipStreamWaitEvent(S, e_n, 0));
    }
    c->pad_now = 0;
    if (e_help_prev) GPT_HIP_CHECK(hipStreamWaitEvent(S, e_help_prev, 0));
    if (use_early && S_cur != S0) {
        hipEvent_t e_sw = get_event(c, 3 + 4 * widths.size());
        if (!e_sw) return GPT_E_HIP;
        GPT_HIP_CHECK(hipEventRecord(e_sw, S_cur));
        GPT_HIP_CHECK(hipStreamWaitEvent(S0, e_sw, 0));
    }
    S = S0;
    if (c->defer_join && !c->use_graph) {
        // (the caller continues on the panel stream -- the last leaf runs there -- and joins the streams itself)
        c->tail_stream = P;
        return GPT_OK;
    }
    hipEvent_t e_end = get_event(c, 1);
    if (!e_end) return GPT_E_HIP;
    GPT_HIP_CHECK(hipEventRecord(e_end, P));
    GPT_HIP_CHECK(hipStreamWaitEvent(S, e_end, 0));
    return GPT_OK;
}

static int potrf_run(gpt_ctx *c, int64_t n, double *A, int64_t lda, double *invd, int32_t *info)
{
    if (!c->use_graph) return potrf_enqueue(c, n, A, lda, invd, info);
    if (c->gexec && (c->g_n != n || c->g_nb != c->nb_outer || c->g_la != c->lookahead || c->g_A != A)) {
        hipGraphExecDestroy(c->gexec);
        c->gexec = nullptr;
    }
    if (!c->gexec) {
        hipGraph_t graph = nullptr;
        GPT_HIP_CHECK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
        int rc = potrf_enqueue(c, n, A, lda, invd, info);
        hipError_t e = hipStreamEndCapture(c->stream, &graph);
        if (rc != GPT_OK) {
            if (graph) hipGraphDestroy(graph);
            return rc;
        }
        GPT_HIP_CHECK(e);
        GPT_HIP_CHECK(hipGraphInstantiate(&c->gexec, graph, nullptr, nullptr, 0));
        hipGraphDestroy(graph);
        c->g_n = n;
        c->g_nb = c->nb_outer;
        c->g_la = c->lookahead;
        c->g_A = A;
    }
    GPT_HIP_CHECK(hipGraphLaunch(c->gexec, c->stream));
    return GPT_OK;
}

// B (m x n) <- B L^-T, n a multiple of 128
static int trsm_rlt(gpt_ctx *c, hipStream_t st, int64_t m, int64_t n, const double *L, int64_t ldl, const double *invd,
                    double *B, int64_t ldb)
{
    if (n == 128) return launch_trsm_panel(st, m, L, ldl, invd, B, ldb);
    const int64_t h = (n / 256) * 128 > 0 ? (n / 256) * 128 : 128;
    GPT_TRY(trsm_rlt(c, st, m, h, L, ldl, invd, B, ldb));
    GPT_TRY(gemm_nt(c, st, m, n - h, h, -1.0, B, ldb, L + h * ldl, ldl, 1.0, B + h, ldb, 0));
    return trsm_rlt(c, st, m, n - h, L + h * ldl + h, ldl, invd + (h / 128) * GPT_WS_BLOCK, B + h, ldb);
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
extern "C" int gpt_ctx_create(int device_id, void *stream, gpt_ctx **out)
{
    if (!out) return GPT_E_ARG;
    int ndev = 0;
    GPT_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (ndev <= 0 || device_id < 0 || device_id >= ndev) {
        gpt_set_error("no such HIP device %d (count %d)", device_id, ndev);
        return GPT_E_HIP;
    }
    GPT_HIP_CHECK(hipSetDevice(device_id));
    gpt_ctx *c = new (std::nothrow) gpt_ctx();
    if (!c) return GPT_E_NOMEM;
    c->device = device_id;
    int lo = 0, hi = 0;
    GPT_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        // The context's own main stream is created with a CU mask that leaves a few CUs to the (unmasked,
        // high-priority) panel stream: the 128x128 diagonal-block kernel needs ~133 KB of LDS and would otherwise
        // never find a CU while a trailing update occupies the chip, which defeats the look-ahead.
        int reserve = 32;
        if (const char *e = getenv("GPT_RESERVE_CUS")) reserve = atoi(e);
        hipDeviceProp_t prop;
        GPT_HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
        const int ncu = prop.multiProcessorCount;
        bool masked = false;
        if (reserve > 0 && reserve < ncu) {
            std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
            for (int i = reserve; i < ncu; i++) mask[i / 32] |= (1u << (i % 32));
            masked = hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)mask.size(), mask.data()) == hipSuccess;
            if (!masked) (void)hipGetLastError();
            else c->reserve_cus = reserve;
            // helper stream: the reserved CUs except the first 8 (those stay free for the diagonal-block kernel, which
            // needs a whole CU's LDS).  While the trailing updates dominate, the panel stream leaves the reserved CUs
            // idle most of the time; a slice of every update runs there (potrf_enqueue).
            // NOTE: every extra stream of a context costs: with a FIFTH stream (main, panel, helper + two more) the
            // runtime maps two of them to one hardware queue and the whole factorisation ran 2x slower (5.5 -> 10.7 ms at
            // N=8192) even with the extra streams unused.  The experimental streams below are therefore created only on
            // request (environment), never by default.
            if (getenv("GPT_LATE_STREAM")) {   // the reserved CUs as a stream of their own (see potrf_enqueue)
                std::vector<uint32_t> lm((ncu + 31) / 32, 0u);
                for (int i = 0; i < reserve; i++) lm[i / 32] |= (1u << (i % 32));
                if (masked && hipExtStreamCreateWithCUMask(&c->late_panel_stream, (uint32_t)lm.size(), lm.data()) != hipSuccess) {
                    (void)hipGetLastError();
                    c->late_panel_stream = nullptr;
                }
            }
            int reserve_early = 0;
            if (const char *e = getenv("GPT_RESERVE_EARLY")) reserve_early = atoi(e);
            if (masked && reserve_early > 0 && reserve_early < reserve) {
                std::vector<uint32_t> em((ncu + 31) / 32, 0u);
                for (int i = reserve_early; i < ncu; i++) em[i / 32] |= (1u << (i % 32));
                if (hipExtStreamCreateWithCUMask(&c->early_stream, (uint32_t)em.size(), em.data()) != hipSuccess) {
                    (void)hipGetLastError();
                    c->early_stream = nullptr;
                }
            }
            if (masked && reserve >= 16) {
                std::vector<uint32_t> hm((ncu + 31) / 32, 0u);
                for (int i = 8; i < reserve; i++) hm[i / 32] |= (1u << (i % 32));
                if (hipExtStreamCreateWithCUMask(&c->helper_stream, (uint32_t)hm.size(), hm.data()) != hipSuccess) {
                    (void)hipGetLastError();
                    c->helper_stream = nullptr;
                }
                c->helper_cus = reserve - 8;
            }
        }
        if (!masked) GPT_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    GPT_HIP_CHECK(hipStreamCreateWithPriority(&c->panel_stream, hipStreamNonBlocking, hi));
    for (int i = 0; i < 5; i++) GPT_HIP_CHECK(hipEventCreate(&c->tev[i]));
    GPT_HIP_CHECK(hipMalloc(&c->d_info, sizeof(int32_t)));
    GPT_HIP_CHECK(hipMalloc((void **)&c->d_flag, 256));
    GPT_HIP_CHECK(hipMalloc((void **)&c->d_l10pk, 16384 * sizeof(double)));
    GPT_HIP_CHECK(hipMalloc((void **)&c->d_edge, 256));
    // (hipMemsetAsync on the context's stream, never hipMemset: one call on the legacy null stream and from then on
    // every kernel of this process starts ~40 us late on every stream -- measured on the block-cyclic engine,
    // 31 -> 41 ms per rank at N=32768 over 8 ranks, potf2 26 -> 45..90 us in the trace)
    GPT_HIP_CHECK(hipMemsetAsync(c->d_flag, 0, 256, c->stream));
    GPT_HIP_CHECK(hipMemsetAsync(c->d_edge, 0, 256, c->stream));
    GPT_HIP_CHECK(hipMalloc(&c->d_scal, 80 * sizeof(double)));     // logdet_dot's partial sums (64) + its counter
    GPT_HIP_CHECK(hipMemsetAsync(c->d_scal, 0, 80 * sizeof(double), c->stream));
    GPT_HIP_CHECK(hipHostMalloc((void **)&c->h_scal, 4 * sizeof(double), hipHostMallocDefault));
    GPT_HIP_CHECK(hipHostMalloc((void **)&c->h_info, sizeof(int32_t), hipHostMallocDefault));
    *out = c;
    return GPT_OK;
}

static void free_factor(gpt_ctx *c)
{
    if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
    if (c->dA) hipFree(c->dA);
    if (c->d_invd) hipFree(c->d_invd);
    if (c->d_y) hipFree(c->d_y);
    if (c->h_yerr) hipHostFree(c->h_yerr);
    c->h_yerr = nullptr;
    if (c->h_alpha) hipHostFree(c->h_alpha);
    c->h_alpha = nullptr;
    c->h_alpha_valid = false;
    if (c->d_alpha) hipFree(c->d_alpha);
    c->dA = c->d_invd = c->d_y = c->d_erry = c->d_alpha = nullptr;
    c->NP = 0;
    c->factored = c->h_alpha_valid = c->alpha_valid = c->binv_valid = c->binv2_valid = c->binv3_valid = false;
}

extern "C" int gpt_ctx_destroy(gpt_ctx *c)
{
    if (!c) return GPT_OK;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    hipStreamSynchronize(c->panel_stream);
    free_factor(c);
    if (c->dX) hipFree(c->dX);
    if (c->dn) hipFree(c->dn);
    if (c->dT) hipFree(c->dT);
    for (auto &b : c->slots)
        if (b.p) hipFree(b.p);
    for (auto e : c->events) hipEventDestroy(e);
    for (auto &g : c->gprof) { hipEventDestroy(g.e0); hipEventDestroy(g.e1); }
    for (int i = 0; i < 5; i++)
        if (c->tev[i]) hipEventDestroy(c->tev[i]);
    if (c->d_info) hipFree(c->d_info);
    if (c->d_flag) hipFree(c->d_flag);
    if (c->d_l10pk) hipFree(c->d_l10pk);
    if (c->d_edge) hipFree(c->d_edge);
    if (c->d_scal) hipFree(c->d_scal);
    if (c->h_scal) hipHostFree(c->h_scal);
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->h_batch) hipHostFree(c->h_batch);
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    for (auto &e : c->cev)
        if (e) hipEventDestroy(e);
    if (c->h_info) hipHostFree(c->h_info);
    hipStreamDestroy(c->panel_stream);
    if (c->helper_stream) hipStreamDestroy(c->helper_stream);
    if (c->early_stream) hipStreamDestroy(c->early_stream);
    if (c->near_stream) hipStreamDestroy(c->near_stream);
    if (c->late_panel_stream) hipStreamDestroy(c->late_panel_stream);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    return GPT_OK;
}

extern "C" int gpt_ctx_set_option(gpt_ctx *c, const char *key, int64_t value)
{
    if (!c || !key) return GPT_E_ARG;
    if (!strcmp(key, "nb_outer")) {
        if (value < 0 || value % 128) {
            gpt_set_error("nb_outer must be 0 (chosen by size) or a positive multiple of 128");
            return GPT_E_ARG;
        }
        c->nb_outer = value;
    } else if (!strcmp(key, "lookahead")) c->lookahead = value ? 1 : 0;
    else if (!strcmp(key, "graph")) c->use_graph = value ? 1 : 0;
    else if (!strcmp(key, "timing")) c->timing = value ? 1 : 0;
    else if (!strcmp(key, "profile_gemm")) c->prof_gemm = value ? 1 : 0;
    else if (!strcmp(key, "gemm_pad")) c->gemm_pad = (int)value;
    else if (!strcmp(key, "ramp")) c->ramp = value ? 1 : 0;
    else if (!strcmp(key, "early_rows")) c->early_rows = value;
    else if (!strcmp(key, "defer_rows")) c->defer_rows = value;
    else if (!strcmp(key, "late_rows")) c->late_rows = value;
    else if (!strcmp(key, "purg_rows")) c->purg_rows = value;
    else if (!strcmp(key, "panel_prio")) c->panel_prio = value;
    else if (!strcmp(key, "edge_flags")) c->edge_flags = value;
    else if (!strcmp(key, "head_wait_wgs")) c->head_wait_wgs = value;
    else if (!strcmp(key, "merge_urgent")) c->merge_urgent = value;
    else if (!strcmp(key, "tail_wait")) c->tail_wait = value;
    else if (!strcmp(key, "purg_rows_flags")) c->purg_rows_flags = value;
    else if (!strcmp(key, "merge_min_tiles")) c->merge_min_tiles = value < 512 ? 512 : value;
    else if (!strcmp(key, "gemm_prio")) c->gemm_prio = value;
    else if (!strcmp(key, "late_pad")) c->late_pad = (int)value;
    else if (!strcmp(key, "late_pad_rows")) c->late_pad_rows = value;
    else if (!strcmp(key, "nb_early")) c->nb_early = value;
    else if (!strcmp(key, "nb_switch_rows")) c->nb_switch_rows = value;
    else if (!strcmp(key, "inner")) c->inner = (int)value;
    else if (!strcmp(key, "inner_rows")) c->inner_rows = value;
    else if (!strcmp(key, "helper_tf")) c->helper_tf = (int)value;
    else if (!strcmp(key, "helper_min_n")) c->helper_min_n = value;
    else if (!strcmp(key, "fuse_trsm")) c->fuse_trsm = value;
    else if (!strcmp(key, "fuse_rows64")) c->fuse_rows64 = value;
    else if (!strcmp(key, "fuse_rows32")) c->fuse_rows32 = value;
    else if (!strcmp(key, "fuse_rows16")) c->fuse_rows16 = value;
    else if (!strcmp(key, "pair_rows")) c->pair_rows = value;
    else if (!strcmp(key, "splitk")) c->splitk = value;
    else if (!strcmp(key, "eager_alpha")) c->eager_alpha = value;
    else if (!strcmp(key, "binv_launches")) { c->binv_launches = value; c->binv_valid = c->binv2_valid = c->binv3_valid = false; c->h_alpha_valid = c->alpha_valid = false; }
    else if (!strcmp(key, "fuse_upd")) c->fuse_upd = value;
    else if (!strcmp(key, "fuse_upd_rows")) c->fuse_upd_rows = value;
    else if (!strcmp(key, "leaf256")) c->leaf256 = value ? 1 : 0;
    else if (!strcmp(key, "debug_poison")) c->debug_poison = value;
    else if (!strcmp(key, "alpha_invalidate")) c->h_alpha_valid = c->alpha_valid = false;          // (measurement aid: the next gpt_get_alpha recomputes)
    else if (!strcmp(key, "edge_test_stall")) c->edge_test_stall = value;
    else if (!strcmp(key, "tile")) {
        if (value != 0 && value != 32 && value != 64 && value != 65 && value != 128 && value != 129) {
            gpt_set_error("tile must be 0, 32, 64 or 128");
            return GPT_E_ARG;
        }
        c->tile = (int)value;
    } else {
        gpt_set_error("unknown option '%s'", key);
        return GPT_E_ARG;
    }
    return GPT_OK;
}

extern "C" int gpt_ctx_synchronize(gpt_ctx *c)
{
    if (!c) return GPT_E_ARG;
    GPT_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GPT_OK;
}

extern "C" void *gpt_ctx_stream(gpt_ctx *c) { return c ? (void *)c->stream : nullptr; }
extern "C" int64_t gpt_ctx_edge_count(gpt_ctx *c) { return c ? (int64_t)c->edge_seq : -1; }

#define CTX_ENTER(c)                          \
    do {                                      \
        if (!(c)) {                           \
            gpt_set_error("null context");    \
            return GPT_E_ARG;                 \
        }                                     \
        GPT_HIP_CHECK(hipSetDevice((c)->device)); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// Kernel.__call__ / compute_Kij
// ------------------------------------------------------------------------------------------------
extern "C" int gpt_kpairs(gpt_ctx *c, int kernel_id, const double *params, int nparams, const double *Xi,
                          const double *Xj, const int32_t *ni, const int32_t *nj, int64_t M, int D,
                          int hyper_deriv, int symmetric, const int32_t *noise_n, double *out)
{
    CTX_ENTER(c);
    if (M < 0 || !params || (M > 0 && (!Xi || !Xj || !ni || !nj || !out))) return GPT_E_ARG;
    KParams kp;
    GPT_TRY(make_kparams(kernel_id, params, nparams, D, hyper_deriv, symmetric, noise_n, &kp));
    if (kernel_id == GPT_KERNEL_M52) {
        GPT_TRY(check_m52_orders(ni, M, D));
        GPT_TRY(check_m52_orders(nj, M, D));
    }
    if ((kernel_id == GPT_KERNEL_RQ || kernel_id == GPT_KERNEL_MATERN) && M > 0) GPT_TRY(check_rq_orders(ni, M, nj, M, D, true));
    if (M == 0) return GPT_OK;
    double *dXi, *dXj, *dout;
    int32_t *dni, *dnj;
    const size_t xb = (size_t)M * D * sizeof(double), nb = (size_t)M * D * sizeof(int32_t);
    GPT_TRY(ensure(c, SLOT_XI, xb, (void **)&dXi));
    GPT_TRY(ensure(c, SLOT_XJ, xb, (void **)&dXj));
    GPT_TRY(ensure(c, SLOT_NI, nb, (void **)&dni));
    GPT_TRY(ensure(c, SLOT_NJ, nb, (void **)&dnj));
    GPT_TRY(ensure(c, SLOT_OUT, (size_t)M * sizeof(double), (void **)&dout));
    hipStream_t st = c->stream;
    GPT_HIP_CHECK(hipMemcpyAsync(dXi, Xi, xb, hipMemcpyHostToDevice, st));
    GPT_HIP_CHECK(hipMemcpyAsync(dXj, Xj, xb, hipMemcpyHostToDevice, st));
    GPT_HIP_CHECK(hipMemcpyAsync(dni, ni, nb, hipMemcpyHostToDevice, st));
    GPT_HIP_CHECK(hipMemcpyAsync(dnj, nj, nb, hipMemcpyHostToDevice, st));
    GPT_TRY(launch_kpairs(st, kp, dXi, dXj, dni, dnj, M, dout));
    GPT_HIP_CHECK(hipMemcpyAsync(out, dout, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, st));
    GPT_HIP_CHECK(hipStreamSynchronize(st));
    return GPT_OK;
}

extern "C" int gpt_kbuild(gpt_ctx *c, int kernel_id, const double *params, int nparams, const double *Xi,
                          const int32_t *ni, int64_t M, const double *Xj, const int32_t *nj, int64_t P, int D,
                          int hyper_deriv, const int32_t *noise_n, double *K_out)
{
    CTX_ENTER(c);
    const int symmetric = (Xj == nullptr);
    if (symmetric) {
        Xj = Xi;
        nj = ni;
        P = M;
    }
    if (M < 0 || P < 0 || !params) return GPT_E_ARG;
    KParams kp;
    GPT_TRY(make_kparams(kernel_id, params, nparams, D, hyper_deriv, symmetric, noise_n, &kp));
    if (kernel_id == GPT_KERNEL_M52) {
        GPT_TRY(check_m52_orders(ni, M, D));
        GPT_TRY(check_m52_orders(nj, P, D));
    }
    if ((kernel_id == GPT_KERNEL_RQ || kernel_id == GPT_KERNEL_MATERN) && M > 0 && P > 0 && ni && nj)
        GPT_TRY(check_rq_orders(ni, M, nj, P, D, false));
    if (M == 0 || P == 0) return GPT_OK;
    if (!Xi || !ni || !Xj || !nj || !K_out) return GPT_E_ARG;
    double *dXi, *dXj, *dK;
    int32_t *dni, *dnj;
    GPT_TRY(ensure(c, SLOT_XI, (size_t)M * D * sizeof(double), (void **)&dXi));
    GPT_TRY(ensure(c, SLOT_NI, (size_t)M * D * sizeof(int32_t), (void **)&dni));
    GPT_TRY(ensure(c, SLOT_XJ, (size_t)P * D * sizeof(double), (void **)&dXj));
    GPT_TRY(ensure(c, SLOT_NJ, (size_t)P * D * sizeof(int32_t), (void **)&dnj));
    GPT_TRY(ensure(c, SLOT_OUT, (size_t)M * P * sizeof(double), (void **)&dK));
    hipStream_t st = c->stream;
    GPT_HIP_CHECK(hipMemcpyAsync(dXi, Xi, (size_t)M * D * sizeof(double), hipMemcpyHostToDevice, st));
    GPT_HIP_CHECK(hipMemcpyAsync(dni, ni, (size_t)M * D * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GPT_HIP_CHECK(hipMemcpyAsync(dXj, Xj, (size_t)P * D * sizeof(double), hipMemcpyHostToDevice, st));
    GPT_HIP_CHECK(hipMemcpyAsync(dnj, nj, (size_t)P * D * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GPT_TRY(launch_kbuild(st, kp, dXi, dni, M, dXj, dnj, P, 0, 0, 0, nullptr, 0.0, 0.0, dK, P));
    GPT_HIP_CHECK(hipMemcpyAsync(K_out, dK, (size_t)M * P * sizeof(double), hipMemcpyDeviceToHost, st));
    GPT_HIP_CHECK(hipStreamSynchronize(st));
    return GPT_OK;
}

// Kernel.__call__ / compute_Kij of the product of two native kernels (ProductKernel, ref: kernel/core.py:587-671)
static int make_product(int kid1, const double *p1, int n1, int kid2, const double *p2, int n2, int D, KParams *k1, KParams *k2)
{
    if (!p1 || !p2) return GPT_E_ARG;
    for (int kid : {kid1, kid2})
        if (kid != GPT_KERNEL_SE && kid != GPT_KERNEL_M52 && kid != GPT_KERNEL_RQ && kid != GPT_KERNEL_MATERN) {
            gpt_set_error("product factors must be SE, Matern52, RationalQuadratic or Matern kernels");
            return GPT_E_ARG;
        }
    GPT_TRY(make_kparams(kid1, p1, n1, D, -1, 0, nullptr, k1));
    GPT_TRY(make_kparams(kid2, p2, n2, D, -1, 0, nullptr, k2));
    return GPT_OK;
}

extern "C" int gpt_kpairs2(gpt_ctx *c, int kernel_id1, const double *params1, int nparams1, int kernel_id2,
                           const double *params2, int nparams2, const double *Xi, const double *Xj, const int32_t *ni,
                           const int32_t *nj, int64_t M, int D, double *out)
{
    CTX_ENTER(c);
    if (M < 0 || (M > 0 && (!Xi || !Xj || !ni || !nj || !out))) return GPT_E_ARG;
    KParams k1, k2;
    GPT_TRY(make_product(kernel_id1, params1, nparams1, kernel_id2, params2, nparams2, D, &k1, &k2));
    if (kernel_id1 == GPT_KERNEL_M52 || kernel_id2 == GPT_KERNEL_M52) {
        GPT_TRY(check_m52_orders(ni, M, D));
        GPT_TRY(check_m52_orders(nj, M, D));
    }
    if (M > 0) GPT_TRY(check_rq_orders(ni, M, nj, M, D, true));       // combined order of a pair <= GPT_RQ_MAXORD
    if (M == 0) return GPT_OK;
    double *dXi, *dXj, *dout;
    int32_t *dni, *dnj;
    const size_t xb = (size_t)M * D * sizeof(double), nb = (size_t)M * D * sizeof(int32_t);
    GPT_TRY(ensure(c, SLOT_XI, xb, (void **)&dXi));
    GPT_TRY(ensure(c, SLOT_XJ, xb, (void **)&dXj));
    GPT_TRY(ensure(c, SLOT_NI, nb, (void **)&dni));
    GPT_TRY(ensure(c, SLOT_NJ, nb, (void **)&dnj));
    GPT_TRY(ensure(c, SLOT_OUT, (size_t)M * sizeof(double), (void **)&dout));
    hipStream_t st = c->stream;
    GPT_HIP_CHECK(hipMemcpyAsync(dXi, Xi, xb, hipMemcpyHostToDevice, st));
    GPT_HIP_CHECK(hipMemcpyAsync(dXj, Xj, xb, hipMemcpyHostToDevice, st));
    GPT_HIP_CHECK(hipMemcpyAsync(dni, ni, nb, hipMemcpyHostToDevice, st));
    GPT_HIP_CHECK(hipMemcpyAsync(dnj, nj, nb, hipMemcpyHostToDevice, st));
    GPT_TRY(launch_kpairs(st, k1, dXi, dXj, dni, dnj, M, dout, 0, &k2));
    GPT_HIP_CHECK(hipMemcpyAsync(out, dout, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, st));
    GPT_HIP_CHECK(hipStreamSynchronize(st));
    return GPT_OK;
}

extern "C" int gpt_kbuild2(gpt_ctx *c, int kernel_id1, const double *params1, int nparams1, int kernel_id2,
                           const double *params2, int nparams2, const double *Xi, const int32_t *ni, int64_t M,
                           const double *Xj, const int32_t *nj, int64_t P, int D, double *K_out)
{
    CTX_ENTER(c);
    if (!Xj) {
        Xj = Xi;
        nj = ni;
        P = M;
    }
    if (M < 0 || P < 0) return GPT_E_ARG;
    KParams k1, k2;
    GPT_TRY(make_product(kernel_id1, params1, nparams1, kernel_id2, params2, nparams2, D, &k1, &k2));
    if (M == 0 || P == 0) return GPT_OK;
    if (!Xi || !ni || !Xj || !nj || !K_out) return GPT_E_ARG;
    if (kernel_id1 == GPT_KERNEL_M52 || kernel_id2 == GPT_KERNEL_M52) {
        GPT_TRY(check_m52_orders(ni, M, D));
        GPT_TRY(check_m52_orders(nj, P, D));
    }
    GPT_TRY(check_rq_orders(ni, M, nj, P, D, false));
    double *dXi, *dXj, *dK;
    int32_t *dni, *dnj;
    GPT_TRY(ensure(c, SLOT_XI, (size_t)M * D * sizeof(double), (void **)&dXi));
    GPT_TRY(ensure(c, SLOT_NI, (size_t)M * D * sizeof(int32_t), (void **)&dni));
    GPT_TRY(ensure(c, SLOT_XJ, (size_t)P * D * sizeof(double), (void **)&dXj));
    GPT_TRY(ensure(c, SLOT_NJ, (size_t)P * D * sizeof(int32_t), (void **)&dnj));
    GPT_TRY(ensure(c, SLOT_OUT, (size_t)M * P * sizeof(double), (void **)&dK));
    hipStream_t st = c->stream;
    GPT_HIP_CHECK(hipMemcpyAsync(dXi, Xi, (size_t)M * D * sizeof(double), hipMemcpyHostToDevice, st));
    GPT_HIP_CHECK(hipMemcpyAsync(dni, ni, (size_t)M * D * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GPT_HIP_CHECK(hipMemcpyAsync(dXj, Xj, (size_t)P * D * sizeof(double), hipMemcpyHostToDevice, st));
    GPT_HIP_CHECK(hipMemcpyAsync(dnj, nj, (size_t)P * D * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GPT_TRY(launch_kbuild(st, k1, dXi, dni, M, dXj, dnj, P, 0, 0, 0, nullptr, 0.0, 0.0, dK, P, 0, &k2));
    GPT_HIP_CHECK(hipMemcpyAsync(K_out, dK, (size_t)M * P * sizeof(double), hipMemcpyDeviceToHost, st));
    GPT_HIP_CHECK(hipStreamSynchronize(st));
    return GPT_OK;
}

// ------------------------------------------------------------------------------------------------
// data residency + fit
// ------------------------------------------------------------------------------------------------
extern "C" int gpt_set_data(gpt_ctx *c, const double *X, const int32_t *n, int64_t N, int D)
{
    CTX_ENTER(c);
    if (N <= 0 || !X || !n || D < 1 || D > GPT_MAX_DIM) {
        gpt_set_error("set_data: bad arguments (N=%lld, D=%d)", (long long)N, D);
        return GPT_E_ARG;
    }
    GPT_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->dX) hipFree(c->dX);
    if (c->dn) hipFree(c->dn);
    c->dX = nullptr;
    c->dn = nullptr;
    GPT_HIP_CHECK(hipMalloc(&c->dX, (size_t)N * D * sizeof(double)));
    GPT_HIP_CHECK(hipMalloc(&c->dn, (size_t)N * D * sizeof(int32_t)));
    GPT_HIP_CHECK(hipMemcpyAsync(c->dX, X, (size_t)N * D * sizeof(double), hipMemcpyHostToDevice, c->stream));
    GPT_HIP_CHECK(hipMemcpyAsync(c->dn, n, (size_t)N * D * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    GPT_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->N = N;
    c->Nx = N;
    c->D = D;
    c->n_maxsum = 0;
    for (int64_t i = 0; i < N; i++) {
        long sn = 0;
        for (int d = 0; d < D; d++) sn += n[i * D + d];
        if (sn > c->n_maxsum) c->n_maxsum = sn;
    }
    c->factored = false;
    c->h_alpha_valid = c->alpha_valid = c->binv_valid = c->binv2_valid = c->binv3_valid = false;
    c->have_kernel = false;
    if (c->dT) hipFree(c->dT);          // a transform belongs to one data set
    c->dT = nullptr;
    c->Ny = 0;
    return GPT_OK;
}

// Linear transform of the latent values (ref: gptools/gaussian_process.py:376-503 `T`, :1443-1446): the observations are
// y = T f(X) + noise, K_tot = T (K + noise_K) T^T + diag(err_y^2) + diag_add I  (Ny x Ny).  T stays resident until the
// data change; T == NULL removes it.
extern "C" int gpt_set_T(gpt_ctx *c, const double *T, int64_t Ny)
{
    CTX_ENTER(c);
    if (!c->dX) {
        gpt_set_error("gpt_set_T: call gpt_set_data first");
        return GPT_E_STATE;
    }
    GPT_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->dT) hipFree(c->dT);
    c->dT = nullptr;
    c->Ny = 0;
    c->factored = false;
    c->h_alpha_valid = c->alpha_valid = c->binv_valid = c->binv2_valid = c->binv3_valid = false;
    c->have_kernel = false;
    if (!T || Ny <= 0) return GPT_OK;
    const int64_t NyP = round_up(Ny, 64), NxP = round_up(c->Nx, 16);
    GPT_HIP_CHECK(hipMalloc(&c->dT, (size_t)NyP * NxP * sizeof(double)));
    GPT_HIP_CHECK(hipMemsetAsync(c->dT, 0, (size_t)NyP * NxP * sizeof(double), c->stream));
    GPT_HIP_CHECK(hipMemcpy2DAsync(c->dT, (size_t)NxP * sizeof(double), T, (size_t)c->Nx * sizeof(double),
                                   (size_t)c->Nx * sizeof(double), (size_t)Ny, hipMemcpyHostToDevice, c->stream));
    GPT_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->Ny = Ny;
    c->NxP = NxP;
    return GPT_OK;
}

static int ensure_factor_storage(gpt_ctx *c, int64_t N)
{
    const int64_t NP = round_up(N + 1, 128);
    if (c->NP == NP && c->dA) return GPT_OK;
    GPT_HIP_CHECK(hipStreamSynchronize(c->stream));
    free_factor(c);
    GPT_HIP_CHECK(hipMalloc(&c->dA, (size_t)NP * NP * sizeof(double)));
    GPT_HIP_CHECK(hipMalloc(&c->d_invd, (size_t)(NP / 128) * GPT_WS_BLOCK * sizeof(double)));
    GPT_HIP_CHECK(hipMalloc(&c->d_y, (size_t)2 * NP * sizeof(double)));          // y | err_y, one upload per evaluation
    c->d_erry = c->d_y + NP;
    GPT_HIP_CHECK(hipHostMalloc((void **)&c->h_yerr, (size_t)2 * NP * sizeof(double), hipHostMallocDefault));
    GPT_HIP_CHECK(hipHostMalloc((void **)&c->h_alpha, (size_t)NP * sizeof(double), hipHostMallocDefault));
    GPT_HIP_CHECK(hipMalloc(&c->d_alpha, (size_t)2 * NP * sizeof(double)));      // alpha | work vector of its substitution
    c->NP = NP;
    return GPT_OK;
}

// Factor the (already assembled, lower) N x N matrix in dA, with y in d_y; produce ll terms.
static int harvest_gemm_profile(gpt_ctx *c);
static int alpha_to_host(gpt_ctx *c);

static int factor_and_ll(gpt_ctx *c, int64_t N, double *ll_data_out, double *logdet_half_out, bool padded = false)
{
    hipStream_t st = c->stream;
    const int64_t NP = c->NP;
    if (!padded) {
        GPT_HIP_CHECK(hipMemsetAsync(c->d_info, 0, sizeof(int32_t), st));
        GPT_TRY(launch_fill_pad(st, c->dA, NP, N, NP, c->d_y, 1e300));
    }
    if (c->timing) GPT_HIP_CHECK(hipEventRecord(c->tev[2], st));
    // The factorisation ends on the panel stream (its last leaf); the reduction over the diagonal and the augmented row
    // and the 24-byte copy follow it THERE -- handing back to the main stream first cost an event edge (~20 us of a 5 ms
    // evaluation) -- and the main stream is joined behind them, off the host's critical path.
    c->defer_join = true;
    c->tail_stream = nullptr;
    int rc_f = potrf_run(c, NP, c->dA, NP, c->d_invd, c->d_info);
    c->defer_join = false;
    GPT_TRY(rc_f);
    hipStream_t tl = c->tail_stream ? c->tail_stream : st;
    c->tail_stream = nullptr;
    // (the three results go straight to pinned host memory, and the two timing events ride on the kernel's dispatch packet:
    // a copy kernel and two barrier packets less on the tail of every evaluation, ~15 us)
    // (only a STOP event: a start event on the packet holds the kernel back ~7 us like a barrier packet would)
    GPT_TRY(launch_logdet_dot(tl, c->dA, NP, N, c->d_info, c->d_scal, c->h_scal, nullptr, c->timing ? c->tev[4] : nullptr,
                              c->flags_now ? c->d_edge + 60 : nullptr));
    c->h_alpha_valid = c->alpha_valid = c->binv_valid = c->binv2_valid = c->binv3_valid = false;
    if (tl != st) {
        hipEvent_t e_end = get_event(c, 1);
        if (!e_end) return GPT_E_HIP;
        GPT_HIP_CHECK(hipEventRecord(e_end, tl));
        GPT_HIP_CHECK(hipStreamWaitEvent(st, e_end, 0));
    }
    // option eager_alpha (the reference computes alpha in every evaluation, gaussian_process.py:1462): the block inverses and the
    // substitution go behind the factorisation at once -- no host round trip between the two, and gpt_get_alpha is a copy
    if (c->eager_alpha) GPT_TRY(alpha_to_host(c));          // (on a failed factorisation: finite work on garbage, flags reset below)
    if (tl != st) GPT_HIP_CHECK(hipStreamSynchronize(tl));
    GPT_HIP_CHECK(hipStreamSynchronize(st));
    if (c->gprof_used) GPT_TRY(harvest_gemm_profile(c));
    if (c->timing) {
        float ms = 0;
        for (int i = 0; i < 2; i++) {
            hipEventElapsedTime(&ms, c->tev[i], c->tev[i + 1]);
            c->timings[i] = ms;
        }
        hipEventElapsedTime(&ms, c->tev[2], c->tev[4]);      // factorisation + the reduction kernel behind it (~5 us)
        c->timings[2] = ms;
        c->timings[3] = 0.0;
        hipEventElapsedTime(&ms, c->tev[0], c->tev[4]);
        c->timings[4] = ms;
    }
    if (c->flags_now && c->h_scal[3] != 0.0) {
        c->h_alpha_valid = c->alpha_valid = c->binv_valid = c->binv2_valid = c->binv3_valid = false;
        // a flag wait of this evaluation timed out (common.hpp): its numbers mean nothing; the caller repeats it on events
        c->factored = false;
        return GPT_I_EDGE_TIMEOUT;
    }
    const int32_t info = (int32_t)c->h_scal[2];
    if (info != 0) {
        c->factored = false;
        c->h_alpha_valid = c->alpha_valid = c->binv_valid = c->binv2_valid = c->binv3_valid = false;
        if (info > N) {          // only the augmented / padding pivots failed: z.z overflowed
            gpt_set_error("factorisation failed in the augmented row (non-finite data?)");
            return (int)N;
        }
        gpt_set_error("%d-th leading minor of the array is not positive definite", (int)info);
        return (int)info;
    }
    c->factored = true;
    const double logdet_half = c->h_scal[0], zz = c->h_scal[1];
    if (logdet_half_out) *logdet_half_out = logdet_half;
    if (ll_data_out) *ll_data_out = -0.5 * zz - logdet_half - 0.5 * (double)N * log(2.0 * M_PI);
    return GPT_OK;
}

// K block of the model kernel = sum of c->terms (SumKernel, ref: gptools/kernel/core.py:549-584): one builder pass per
// term, later passes accumulate; the diagonal epilogue (err != nullptr) rides on the last pass, after the sum.
static int kbuild_terms(gpt_ctx *c, hipStream_t st, const std::vector<KParams> &terms, int symmetric, const double *dXi,
                        const int32_t *dni, int64_t M, const double *dXj, const int32_t *dnj, int64_t P, int lower_only,
                        int64_t i0, int64_t j0, const double *d_err, double noise_var, double diag_add, double *dK,
                        int64_t ldk)
{
    for (size_t t = 0; t < terms.size(); t++) {
        KParams kp = terms[t];
        kp.symmetric = symmetric;
        kp.hyper_deriv = -1;
        const bool last = t + 1 == terms.size();
        // (a product term brings its second factor: c->terms2 runs parallel to c->terms whenever `terms` IS c->terms)
        const KParams *kp2 = (&terms == &c->terms && t < c->terms2.size() && c->terms2[t].kernel_id >= 0) ? &c->terms2[t] : nullptr;
        GPT_TRY(launch_kbuild(st, kp, dXi, dni, M, dXj, dnj, P, lower_only, i0, j0, last ? d_err : nullptr, noise_var,
                              diag_add, dK, ldk, t > 0 ? 1 : 0, kp2));
    }
    return GPT_OK;
}

static int fit_terms(gpt_ctx *c, const std::vector<KParams> &terms, double noise_var, const double *y,
                     const double *err_y, double diag_add, double *ll_data_out, double *logdet_half_out);

extern "C" int gpt_fit(gpt_ctx *c, int kernel_id, const double *params, int nparams, double noise_var,
                       const double *y, const double *err_y, double diag_add, double *ll_data_out,
                       double *logdet_half_out)
{
    return gpt_fit_sum(c, 1, &kernel_id, params, &nparams, noise_var, y, err_y, diag_add, ll_data_out, logdet_half_out);
}

extern "C" int gpt_fit_sum(gpt_ctx *c, int nterms, const int *kernel_ids, const double *params, const int *nparams,
                           double noise_var, const double *y, const double *err_y, double diag_add,
                           double *ll_data_out, double *logdet_half_out)
{
    CTX_ENTER(c);
    if (!c->dX) {
        gpt_set_error("gpt_fit: call gpt_set_data first");
        return GPT_E_STATE;
    }
    if (nterms < 1 || nterms > 8 || !kernel_ids || !params || !nparams || !y || !err_y) return GPT_E_ARG;
    std::vector<KParams> terms((size_t)nterms);
    const double *p = params;
    for (int t = 0; t < nterms; t++) {
        if (kernel_ids[t] != GPT_KERNEL_SE && kernel_ids[t] != GPT_KERNEL_M52 && kernel_ids[t] != GPT_KERNEL_RQ &&
            kernel_ids[t] != GPT_KERNEL_MATERN) {
            gpt_set_error("gpt_fit: kernel_id must be SE, Matern52, RationalQuadratic or Matern");
            return GPT_E_ARG;
        }
        if ((kernel_ids[t] == GPT_KERNEL_RQ || kernel_ids[t] == GPT_KERNEL_MATERN) && 2 * c->n_maxsum > GPT_RQ_MAXORD) {
            gpt_set_error("RationalQuadratic / Matern kernel: derivative orders of a pair sum to %ld, the device builder supports %d",
                          2 * c->n_maxsum, GPT_RQ_MAXORD);
            return GPT_E_VALUE;
        }
        GPT_TRY(make_kparams(kernel_ids[t], p, nparams[t], c->D, -1, 1, nullptr, &terms[(size_t)t]));
        p += nparams[t];
    }
    c->terms2.assign(terms.size(), KParams());
    for (auto &k2 : c->terms2) k2.kernel_id = -1;
    return fit_terms(c, terms, noise_var, y, err_y, diag_add, ll_data_out, logdet_half_out);
}

// The same with product terms (include/gpt_hip.h)
static bool native_fit_kernel(int kid)
{
    return kid == GPT_KERNEL_SE || kid == GPT_KERNEL_M52 || kid == GPT_KERNEL_RQ || kid == GPT_KERNEL_MATERN;
}

extern "C" int gpt_fit_terms(gpt_ctx *c, int nterms, const int *kernel_ids, const int *kernel_ids2, const double *params,
                             const int *nparams, const int *nparams1, double noise_var, const double *y, const double *err_y,
                             double diag_add, double *ll_data_out, double *logdet_half_out)
{
    CTX_ENTER(c);
    if (!c->dX) {
        gpt_set_error("gpt_fit_terms: call gpt_set_data first");
        return GPT_E_STATE;
    }
    if (nterms < 1 || nterms > 8 || !kernel_ids || !kernel_ids2 || !params || !nparams || !nparams1 || !y || !err_y) return GPT_E_ARG;
    std::vector<KParams> terms((size_t)nterms), terms2((size_t)nterms);
    const double *p = params;
    for (int t = 0; t < nterms; t++) {
        const bool prod = kernel_ids2[t] >= 0;
        if (!native_fit_kernel(kernel_ids[t]) || (prod && !native_fit_kernel(kernel_ids2[t]))) {
            gpt_set_error("gpt_fit_terms: kernel ids must be SE, Matern52, RationalQuadratic or Matern");
            return GPT_E_ARG;
        }
        const int n1 = prod ? nparams1[t] : nparams[t];
        if (n1 < 1 || n1 > nparams[t]) return GPT_E_ARG;
        // derivative orders: a product meets the SUM of both points' orders in either factor
        const bool any_chain = kernel_ids[t] == GPT_KERNEL_RQ || kernel_ids[t] == GPT_KERNEL_MATERN ||
                               (prod && (kernel_ids2[t] == GPT_KERNEL_RQ || kernel_ids2[t] == GPT_KERNEL_MATERN));
        if ((any_chain || prod) && 2 * c->n_maxsum > GPT_RQ_MAXORD) {
            gpt_set_error("derivative orders of a pair sum to %ld, the device builder supports %d for products and the "
                          "RationalQuadratic / Matern kernels", 2 * c->n_maxsum, GPT_RQ_MAXORD);
            return GPT_E_VALUE;
        }
        GPT_TRY(make_kparams(kernel_ids[t], p, n1, c->D, -1, 1, nullptr, &terms[(size_t)t]));
        terms2[(size_t)t] = KParams();
        terms2[(size_t)t].kernel_id = -1;
        if (prod) GPT_TRY(make_kparams(kernel_ids2[t], p + n1, nparams[t] - n1, c->D, -1, 1, nullptr, &terms2[(size_t)t]));
        p += nparams[t];
    }
    c->terms2 = terms2;
    return fit_terms(c, terms, noise_var, y, err_y, diag_add, ll_data_out, logdet_half_out);
}

static int fit_terms_once(gpt_ctx *c, const std::vector<KParams> &terms, double noise_var, const double *y,
                          const double *err_y, double diag_add, double *ll_data_out, double *logdet_half_out);

// An evaluation whose flag wait timed out: the process goes to event edges for good and the evaluation runs again.
static int edge_timeout_fallback(gpt_ctx *c)
{
    g_flags_tripped.store(true);
    hipStreamSynchronize(c->stream);
    hipStreamSynchronize(c->panel_stream);
    static bool told = false;
    if (!told) {
        told = true;
        fprintf(stderr, "libgpt_hip: a flag-edge wait timed out (GPU shared with other work, or kernels serialised by a tool); "
                        "this process uses event edges from now on (GPT_EDGE_FLAGS=0 avoids the attempt)\n");
    }
    return GPT_OK;
}

static int fit_terms(gpt_ctx *c, const std::vector<KParams> &terms, double noise_var, const double *y,
                     const double *err_y, double diag_add, double *ll_data_out, double *logdet_half_out)
{
    int rc;
    {
        EvalScope scope(c);
        rc = fit_terms_once(c, terms, noise_var, y, err_y, diag_add, ll_data_out, logdet_half_out);
    }
    if (rc == GPT_I_EDGE_TIMEOUT) {
        edge_timeout_fallback(c);
        EvalScope scope(c);
        rc = fit_terms_once(c, terms, noise_var, y, err_y, diag_add, ll_data_out, logdet_half_out);
    }
    return rc;
}

static int fit_terms_once(gpt_ctx *c, const std::vector<KParams> &terms, double noise_var, const double *y,
                          const double *err_y, double diag_add, double *ll_data_out, double *logdet_half_out)
{
    const int64_t Nx = c->Nx;
    const int64_t N = c->dT ? c->Ny : Nx;          // order of K_tot
    c->N = N;
    GPT_TRY(ensure_factor_storage(c, N));
    hipStream_t st = c->stream;
    c->factored = false;
    if (c->timing) GPT_HIP_CHECK(hipEventRecord(c->tev[0], st));
    memcpy(c->h_yerr, y, (size_t)N * sizeof(double));
    memcpy(c->h_yerr + c->NP, err_y, (size_t)N * sizeof(double));
    const int64_t NP = c->NP;
    // (kernel path without T: upload, info = 0 and the padding rows are one launch, see upload_pad_kernel)
    const bool one_launch = (c->dT == nullptr) && NP > N;
    if (!one_launch) {
        GPT_HIP_CHECK(hipMemcpyAsync(c->d_y, c->h_yerr, (size_t)(c->NP + N) * sizeof(double), hipMemcpyHostToDevice, st));
        if (c->timing) GPT_HIP_CHECK(hipEventRecord(c->tev[1], st));
        GPT_HIP_CHECK(hipMemsetAsync(c->d_info, 0, sizeof(int32_t), st));
    }
    c->terms = terms;
    c->kp = terms[0];
    if (c->dT) {
        // ---- T path: K_tot = T (K + noise_var I) T^T + diag(err_y^2) + diag_add I, assembled on the device by the
        // K-builder (full symmetric K over the Nx latent points) and two fp64-MFMA GEMMs (ref :1443-1451)
        const int64_t NxP = c->NxP, NyP = round_up(N, 64);
        double *dK, *dTK, *dzero;
        GPT_TRY(ensure(c, SLOT_KFULL, (size_t)NxP * NxP * sizeof(double), (void **)&dK));
        GPT_TRY(ensure(c, SLOT_TK, (size_t)NyP * NxP * sizeof(double), (void **)&dTK));
        GPT_TRY(ensure(c, SLOT_ZERO, (size_t)Nx * sizeof(double), (void **)&dzero));
        GPT_HIP_CHECK(hipMemsetAsync(dzero, 0, (size_t)Nx * sizeof(double), st));
        if (NxP > Nx) GPT_HIP_CHECK(hipMemsetAsync(dK, 0, (size_t)NxP * NxP * sizeof(double), st));   // zero padding of k
        // (K + noise_K): the builder's diagonal epilogue with err = 0, diag_add = 0 adds exactly noise_var
        GPT_TRY(kbuild_terms(c, st, c->terms, 1, c->dX, c->dn, Nx, c->dX, c->dn, Nx, 0, 0, 0, dzero, noise_var, 0.0, dK, NxP));
        GPT_TRY(gemm_nt(c, st, NyP, NxP, NxP, 1.0, c->dT, NxP, dK, NxP, 0.0, dTK, NxP, 0));          // T K  (K = K^T)
        GPT_TRY(gemm_nt(c, st, NyP, NyP, NxP, 1.0, dTK, NxP, c->dT, NxP, 0.0, c->dA, NP, 1));        // (T K) T^T, lower
        GPT_TRY(launch_add_diag(st, c->dA, NP, N, c->d_erry, diag_add));
        GPT_TRY(launch_fill_pad(st, c->dA, NP, N, NP, c->d_y, 1e300));
        c->have_kernel = true;
        c->head_event = nullptr;
        return factor_and_ll(c, N, ll_data_out, logdet_half_out, true);
    }
    // The columns the first panel touches are built first so that its pivot chain overlaps the rest of the build.
    if (one_launch) {
        GPT_TRY(launch_upload_pad(st, c->h_yerr, c->d_y, NP + N, c->d_info, c->dA, NP, N, NP, 1e300));
        if (c->timing) GPT_HIP_CHECK(hipEventRecord(c->tev[1], st));
    } else {
        GPT_TRY(launch_fill_pad(st, c->dA, NP, N, NP, c->d_y, 1e300));
    }
    const int64_t w0 = (c->nb_early > outer_width(c, NP)) ? c->nb_early : outer_width(c, NP);
    int64_t head = round_up((c->ramp ? 128 : w0) + (c->leaf256 ? 256 : GPT_PANEL_EXT), 256);   // what panel 0 touches
    hipEvent_t e_head = nullptr;
    c->head_wait = EdgeSig();
    const bool head_flag = c->flags_now && c->edge_seq < 0xf0000000u;
    if (c->lookahead && !c->use_graph && head < N && (head_flag || (e_head = get_event(c, 0)) != nullptr)) {
        GPT_TRY(kbuild_terms(c, st, c->terms, 1, c->dX, c->dn, N, c->dX, c->dn, head, 1, 0, 0, c->d_erry, noise_var, diag_add,
                             c->dA, NP));
        if (head_flag) {
            // "the head columns are built" as a flag word raised from this stream (a one-thread kernel behind the build);
            // the first diagonal-block kernel of the panel stream polls it itself: no event record here, no event wait there
            c->head_wait.word = c->d_edge + 48;
            c->head_wait.value = ++c->edge_seq;
            c->head_wait = with_err(c, c->head_wait);
            // (option "edge_test_stall", test aid: the flag is NOT raised, once -- the waiter must time out, the evaluation be
            // repeated on events and give the right numbers: tests/test_gpu_a_dist_processes.py.  An explicit option of the
            // context, not an environment variable: a stray variable must not be able to push a process off its flag edges)
            if (c->edge_test_stall) c->edge_test_stall = 0;
            else GPT_TRY(launch_set_flag(st, c->head_wait.word, c->head_wait.value));
        } else {
            GPT_HIP_CHECK(hipEventRecord(e_head, st));
        }
        GPT_TRY(kbuild_terms(c, st, c->terms, 1, c->dX + head * c->D, c->dn + head * c->D, N - head, c->dX + head * c->D,
                             c->dn + head * c->D, N - head, 1, head, head, c->d_erry, noise_var, diag_add,
                             c->dA + head * NP + head, NP));
    } else {
        GPT_TRY(kbuild_terms(c, st, c->terms, 1, c->dX, c->dn, N, c->dX, c->dn, N, 1, 0, 0, c->d_erry, noise_var, diag_add,
                             c->dA, NP));
    }
    c->have_kernel = true;
    c->head_event = e_head;
    return factor_and_ll(c, N, ll_data_out, logdet_half_out, true);
}

static int fit_matrix_once(gpt_ctx *c, const double *K_tot, int64_t N, const double *y, double *ll_data_out,
                           double *logdet_half_out);

extern "C" int gpt_fit_matrix(gpt_ctx *c, const double *K_tot, int64_t N, const double *y, double *ll_data_out,
                              double *logdet_half_out)
{
    CTX_ENTER(c);
    if (!K_tot || !y || N <= 0) return GPT_E_ARG;
    int rc;
    {
        EvalScope scope(c);
        rc = fit_matrix_once(c, K_tot, N, y, ll_data_out, logdet_half_out);
    }
    if (rc == GPT_I_EDGE_TIMEOUT) {
        edge_timeout_fallback(c);
        EvalScope scope(c);
        rc = fit_matrix_once(c, K_tot, N, y, ll_data_out, logdet_half_out);
    }
    return rc;
}

static int fit_matrix_once(gpt_ctx *c, const double *K_tot, int64_t N, const double *y, double *ll_data_out,
                           double *logdet_half_out)
{
    GPT_TRY(ensure_factor_storage(c, N));
    hipStream_t st = c->stream;
    c->factored = false;
    c->have_kernel = false;
    c->N = N;
    if (c->timing) {
        GPT_HIP_CHECK(hipEventRecord(c->tev[0], st));
        GPT_HIP_CHECK(hipEventRecord(c->tev[1], st));
    }
    GPT_HIP_CHECK(hipMemcpyAsync(c->d_y, y, (size_t)N * sizeof(double), hipMemcpyHostToDevice, st));
    GPT_HIP_CHECK(hipMemcpy2DAsync(c->dA, (size_t)c->NP * sizeof(double), K_tot, (size_t)N * sizeof(double),
                                   (size_t)N * sizeof(double), (size_t)N, hipMemcpyHostToDevice, st));
    return factor_and_ll(c, N, ll_data_out, logdet_half_out);
}

// ------------------------------------------------------------------------------------------------
// batched small fits
// ------------------------------------------------------------------------------------------------
// nbatch independent LML evaluations of the resident data set -- the reference's likelihood grids and random starts
// (ref: gaussian_process.py:1607-1692, :723-735; gp_utils.py:98-115) live at N of a few hundred to a few thousand, where ONE
// evaluation cannot fill the GPU: its chain of 128-column leaves (diagonal block 20 us on one CU, TRSM, rank-128 update) is
// pure latency and the host's launch rate bounds it (0.37 ms at N = 1024).  Here every kernel of that chain carries the
// whole batch in a grid dimension: element b works on its own matrix A + b NP^2 with its own hyperparameters, the
// diagonal-block kernel runs nbatch workgroups on nbatch CUs at once, and the launch sequence (3 launches per leaf) is paid
// once per batch.  No look-ahead, one stream, left-looking leaves: the parallelism is across the batch.  Same kernels, same tile choice and the
// same summation orders as gpt_fit, so an element's ll / log-determinant carry the very bits gpt_fit returns for it alone
// (tests/test_gpu_parity.py::test_fit_batch_*).  N <= GPT_BATCH_MAX_N; one native kernel, no transform.
#define GPT_BATCH_MAX_N 8192
static int fit_batch_impl(gpt_ctx *c, int nbatch, int nterms, const int *kernel_ids, const int *kernel_ids2, const double *params,
                          const int *nparams_t, const int *nparams1_t, const double *noise_var, const double *y, const double *err_y,
                          double diag_add, double *ll_data_out, double *logdet_half_out, int32_t *info_out);

extern "C" int gpt_fit_batch(gpt_ctx *c, int nbatch, int kernel_id, const double *params, int nparams,
                             const double *noise_var, const double *y, const double *err_y, double diag_add,
                             double *ll_data_out, double *logdet_half_out, int32_t *info_out)
{
    return fit_batch_impl(c, nbatch, 1, &kernel_id, nullptr, params, &nparams, nullptr, noise_var, y, err_y, diag_add, ll_data_out,
                          logdet_half_out, info_out);
}

// The same for a SumKernel of native kernels (ref: kernel/core.py:549-584): element b's parameters are the terms' parameters
// concatenated (ptot = sum of nparams_t doubles per element); one accumulating builder pass per term for the whole batch, as
// gpt_fit_sum does for one matrix -- same kernels, same order: bit-identical to gpt_fit_sum per element.
extern "C" int gpt_fit_batch_sum(gpt_ctx *c, int nbatch, int nterms, const int *kernel_ids, const double *params,
                                 const int *nparams_t, const double *noise_var, const double *y, const double *err_y,
                                 double diag_add, double *ll_data_out, double *logdet_half_out, int32_t *info_out)
{
    return fit_batch_impl(c, nbatch, nterms, kernel_ids, nullptr, params, nparams_t, nullptr, noise_var, y, err_y, diag_add, ll_data_out,
                          logdet_half_out, info_out);
}

// ... and for every model gpt_fit_terms takes (the reference's likelihood grid works for any model, ref
// gaussian_process.py:1607-1692, gp_utils.py:98-115): PRODUCT terms (kernel_ids2[t] >= 0: term t is kernel_ids[t] *
// kernel_ids2[t], the first nparams1[t] of its nparams[t] parameters the first factor's) and, when gpt_set_T has set one, the
// linear TRANSFORM -- every element's full K over the latent points, then T K T^T as two batched fp64-MFMA GEMMs with the shared,
// resident T (batch stride 0), the diagonal loading and the factorisation as above over the Ny observations.  Same kernels and
// summation orders as gpt_fit_terms: bit-identical per element.
extern "C" int gpt_fit_batch_terms(gpt_ctx *c, int nbatch, int nterms, const int *kernel_ids, const int *kernel_ids2,
                                   const double *params, const int *nparams_t, const int *nparams1_t, const double *noise_var,
                                   const double *y, const double *err_y, double diag_add, double *ll_data_out,
                                   double *logdet_half_out, int32_t *info_out)
{
    return fit_batch_impl(c, nbatch, nterms, kernel_ids, kernel_ids2, params, nparams_t, nparams1_t, noise_var, y, err_y, diag_add,
                          ll_data_out, logdet_half_out, info_out);
}

static int fit_batch_impl(gpt_ctx *c, int nbatch, int nterms, const int *kernel_ids, const int *kernel_ids2, const double *params,
                          const int *nparams_t, const int *nparams1_t, const double *noise_var, const double *y, const double *err_y,
                          double diag_add, double *ll_data_out, double *logdet_half_out, int32_t *info_out)
{
    CTX_ENTER(c);
    if (!c->dX) {
        gpt_set_error("gpt_fit_batch: call gpt_set_data first");
        return GPT_E_STATE;
    }
    if (nbatch < 1 || nbatch > 65535 || nterms < 1 || nterms > 8 || !kernel_ids || !nparams_t || !params || !noise_var || !y ||
        !err_y || !ll_data_out || !info_out || (kernel_ids2 && !nparams1_t))
        return GPT_E_ARG;
    const int64_t Nx = c->Nx, N = c->dT ? c->Ny : Nx;                   // latent points / order of K_tot
    if (Nx > GPT_BATCH_MAX_N || N > GPT_BATCH_MAX_N) {
        gpt_set_error("gpt_fit_batch: needs N <= %d (N = %lld)", GPT_BATCH_MAX_N, (long long)(Nx > N ? Nx : N));
        return GPT_E_ARG;
    }
    int ptot = 0;
    bool any_prod = false;
    for (int t = 0; t < nterms; t++) {
        const bool prod = kernel_ids2 && kernel_ids2[t] >= 0;
        any_prod = any_prod || prod;
        if (!native_fit_kernel(kernel_ids[t]) || (prod && !native_fit_kernel(kernel_ids2[t]))) {
            gpt_set_error("gpt_fit_batch: kernel ids must be SE, Matern52, RationalQuadratic or Matern");
            return GPT_E_ARG;
        }
        const bool any_chain = kernel_ids[t] == GPT_KERNEL_RQ || kernel_ids[t] == GPT_KERNEL_MATERN ||
                               (prod && (kernel_ids2[t] == GPT_KERNEL_RQ || kernel_ids2[t] == GPT_KERNEL_MATERN));
        if ((any_chain || prod) && 2 * c->n_maxsum > GPT_RQ_MAXORD) {
            gpt_set_error("derivative orders of a pair sum to %ld, the device builder supports %d for products and the "
                          "RationalQuadratic / Matern kernels", 2 * c->n_maxsum, GPT_RQ_MAXORD);
            return GPT_E_VALUE;
        }
        if (nparams_t[t] < 1 || (prod && (nparams1_t[t] < 1 || nparams1_t[t] >= nparams_t[t]))) return GPT_E_ARG;
        ptot += nparams_t[t];
    }
    const int64_t NP = round_up(N + 1, 128), nleaf = NP / 128, bs = NP * NP, bws = nleaf * GPT_WS_BLOCK;
    // pinned staging: [y: nbatch N | err: N | noise: nbatch | KParams: nterms x nbatch (| second factors: the same) | results: 4
    // nbatch]; err .. KParams go to the device in one copy
    static_assert(sizeof(KParams) % 8 == 0, "KParams is copied as an array of doubles");
    const size_t kp_doubles = sizeof(KParams) / 8;
    const size_t nkp = (size_t)nterms * nbatch * kp_doubles * (any_prod ? 2 : 1);
    const size_t off_err = (size_t)nbatch * N, off_nv = off_err + (size_t)N, off_kp = off_nv + (size_t)nbatch, off_res = off_kp + nkp;
    const size_t need = (off_res + 4 * (size_t)nbatch) * sizeof(double);
    if (c->h_batch_cap < need) {
        if (c->h_batch) GPT_HIP_CHECK(hipHostFree(c->h_batch));
        c->h_batch = nullptr;
        c->h_batch_cap = 0;
        GPT_HIP_CHECK(hipHostMalloc((void **)&c->h_batch, need + need / 4, hipHostMallocDefault));
        c->h_batch_cap = need + need / 4;
    }
    double *h = c->h_batch;
    memcpy(h, y, (size_t)nbatch * N * sizeof(double));
    memcpy(h + off_err, err_y, (size_t)N * sizeof(double));
    memcpy(h + off_nv, noise_var, (size_t)nbatch * sizeof(double));
    char *hkp = reinterpret_cast<char *>(h + off_kp), *hkp2 = hkp + (size_t)nterms * nbatch * kp_doubles * 8;
    for (int b = 0; b < nbatch; b++) {
        const double *pb = params + (size_t)b * ptot;
        for (int t = 0; t < nterms; t++) {                                  // term-major on the device: [t][b]
            const bool prod = kernel_ids2 && kernel_ids2[t] >= 0;
            const int n1 = prod ? nparams1_t[t] : nparams_t[t];
            KParams kp, kp2 = KParams();
            GPT_TRY(make_kparams(kernel_ids[t], pb, n1, c->D, -1, 1, nullptr, &kp));
            memcpy(hkp + ((size_t)t * nbatch + (size_t)b) * kp_doubles * 8, &kp, sizeof(KParams));
            if (any_prod) {
                kp2.kernel_id = -1;
                if (prod) GPT_TRY(make_kparams(kernel_ids2[t], pb + n1, nparams_t[t] - n1, c->D, -1, 1, nullptr, &kp2));
                memcpy(hkp2 + ((size_t)t * nbatch + (size_t)b) * kp_doubles * 8, &kp2, sizeof(KParams));
            }
            pb += nparams_t[t];
        }
    }
    // device scratch: the nbatch matrices to factor (+ with a transform every element's K over the latent points and T K)
    const int64_t NxP = c->dT ? c->NxP : 0, NyP = c->dT ? round_up(N, 64) : 0;
    const size_t kfull = (size_t)NxP * NxP, tk = (size_t)NyP * NxP;
    double *dA, *dws, *dmisc;
    GPT_TRY(ensure(c, SLOT_BATCH_A, ((size_t)nbatch * bs + (size_t)nbatch * (kfull + tk)) * sizeof(double), (void **)&dA));
    double *dKf = dA + (size_t)nbatch * bs, *dTK = dKf + (size_t)nbatch * kfull;
    GPT_TRY(ensure(c, SLOT_BATCH_WS, (size_t)nbatch * bws * sizeof(double), (void **)&dws));
    // device side of the small inputs: [err: N | noise: nbatch | KParams (| second factors) | info: nbatch | zeros: Nx]
    const size_t d_off_nv = (size_t)N, d_off_kp = d_off_nv + (size_t)nbatch, d_off_info = d_off_kp + nkp,
                 d_off_zero = d_off_info + (size_t)nbatch;
    GPT_TRY(ensure(c, SLOT_BATCH_MISC, (d_off_zero + (size_t)(c->dT ? Nx : 0)) * sizeof(double), (void **)&dmisc));
    int32_t *dinfo = reinterpret_cast<int32_t *>(dmisc + d_off_info);
    const KParams *dkp = reinterpret_cast<const KParams *>(dmisc + d_off_kp);
    const KParams *dkp2 = any_prod ? dkp + (size_t)nterms * nbatch : nullptr;
    EvalScope scope(c, true);                // (in flight like an evaluation for the flag-edge accounting; has no flag edges)
    // everything on the panel stream: unmasked (all 256 CUs), the main stream is idle here
    hipStream_t st = c->panel_stream;
    {
        hipEvent_t e = get_event(c, 0);
        if (!e) return GPT_E_HIP;
        GPT_HIP_CHECK(hipEventRecord(e, c->stream));
        GPT_HIP_CHECK(hipStreamWaitEvent(st, e, 0));
    }
    GPT_HIP_CHECK(hipMemcpyAsync(dmisc, h + off_err, ((size_t)N + (size_t)nbatch + nkp) * sizeof(double), hipMemcpyHostToDevice, st));
    auto term_kp2 = [&](int t) -> const KParams * {
        return (kernel_ids2 && kernel_ids2[t] >= 0) ? dkp2 + (size_t)t * nbatch : nullptr;
    };
    if (c->dT) {
        // K_tot = T (K + noise_var I) T^T + diag(err_y^2) + diag_add I per element (ref :1443-1451), as fit_terms_once does for one
        double *dzero = dmisc + d_off_zero;
        GPT_HIP_CHECK(hipMemsetAsync(dzero, 0, (size_t)Nx * sizeof(double), st));
        if (NxP > Nx) GPT_HIP_CHECK(hipMemsetAsync(dKf, 0, (size_t)nbatch * kfull * sizeof(double), st));
        for (int t = 0; t < nterms; t++)
            GPT_TRY(launch_kbuild_batch(st, kernel_ids[t], c->D, dkp + (size_t)t * nbatch, dmisc + d_off_nv, nbatch, c->dX, c->dn, Nx,
                                        t + 1 == nterms ? dzero : nullptr, 0.0, dKf, NxP, (int64_t)kfull, t > 0 ? 1 : 0, 1, term_kp2(t)));
        GPT_TRY(launch_gemm_nt(st, NyP, NxP, NxP, 1.0, c->dT, NxP, dKf, NxP, 0.0, dTK, NxP, 0, 0, 0, nullptr, nullptr, 0, EdgeSig(),
                               EdgeSig(), 0, nbatch, 0, EdgeSig(), (int64_t)kfull, (int64_t)tk));
        GPT_TRY(launch_gemm_nt(st, NyP, NyP, NxP, 1.0, dTK, NxP, c->dT, NxP, 0.0, dA, NP, 1, 0, 0, nullptr, nullptr, 0, EdgeSig(),
                               EdgeSig(), 0, nbatch, (int64_t)tk, EdgeSig(), 0, bs));
        GPT_TRY(launch_add_diag(st, dA, NP, N, dmisc, diag_add, nbatch, bs));
        GPT_TRY(launch_batch_pad(st, h, nbatch, dA, NP, bs, N, NP, 1e300, dinfo));
    } else {
        GPT_TRY(launch_batch_pad(st, h, nbatch, dA, NP, bs, N, NP, 1e300, dinfo));
        for (int t = 0; t < nterms; t++)                                    // (as kbuild_terms: later terms accumulate, the last
            GPT_TRY(launch_kbuild_batch(st, kernel_ids[t], c->D, dkp + (size_t)t * nbatch, dmisc + d_off_nv, nbatch,   //  one carries the
                                        c->dX, c->dn, N, t + 1 == nterms ? dmisc : nullptr, diag_add, dA, NP, bs,     //  diagonal epilogue)
                                        t > 0 ? 1 : 0, 0, term_kp2(t)));
    }
    // LEFT-looking over the 128-column leaves: leaf j first receives the update of ALL leaves before it in one launch
    // (k = 128 j; element by element the same sums in the same order as the right-looking rank-128 updates of gpt_fit, whose
    // accumulators also start from C and walk k upwards: bit-identical), then its diagonal block and TRSM.  A right-looking
    // batch re-reads and re-writes every element of every trailing matrix once per leaf (16 bytes per 256 flops at k = 128):
    // 0.77 of the 1.31 ms of a 64 x N = 1024 batch were those updates; here every element of the factor is written once.
    // (GPT_BATCH_RIGHT=1: the right-looking form, for comparison)
    // Measured, 64 elements: N = 1024 1.376 against 1.412 ms, N = 2048 5.79 against 6.22 ms, N = 256 0.247 against 0.237 ms.
    // Larger N (round 3, bit-identical throughout): N = 3000 x 32 elements 7.8 ms (4100 evaluations/s against 1109 one by one and
    // 1848 with two contexts in two threads), N = 4096 x 32 18.2 ms (1755 / 774 / 1168), N = 8192 x 8 34.3 ms (233 / 224 / 248: a
    // single evaluation fills the chip there -- GaussianProcess.ll_batch takes this path up to N = 4096).
    static const bool force_right = getenv("GPT_BATCH_RIGHT") != nullptr;
    const bool right_looking = force_right || NP <= 512;
    for (int64_t lc = 0; lc < NP; lc += 128) {
        if (!right_looking && lc > 0)
            GPT_TRY(launch_gemm_nt(st, NP - lc, 128, lc, -1.0, dA + lc * NP, NP, dA + lc * NP, NP, 1.0, dA + lc * NP + lc, NP, 1, 0, 0,
                                   nullptr, nullptr, 0, EdgeSig(), EdgeSig(), 0, nbatch, bs));
        GPT_TRY(launch_potf2_diag(st, dA + lc * NP + lc, NP, dws + (lc / 128) * GPT_WS_BLOCK, dinfo, lc, EdgeSig(), nbatch, bs, bws));
        const int64_t r1 = lc + 128, m = NP - r1;
        if (m <= 0) break;
        GPT_TRY(launch_trsm_panel(st, m, dA + lc * NP + lc, NP, dws + (lc / 128) * GPT_WS_BLOCK, dA + r1 * NP + lc, NP, nullptr,
                                  EdgeSig(), nbatch, bs, bws));
        if (right_looking)
            GPT_TRY(launch_gemm_nt(st, m, m, 128, -1.0, dA + r1 * NP + lc, NP, dA + r1 * NP + lc, NP, 1.0, dA + r1 * NP + r1, NP, 1, 0, 0,
                                   nullptr, nullptr, 0, EdgeSig(), EdgeSig(), 0, nbatch, bs));
    }
    GPT_TRY(launch_batch_logdet_dot(st, dA, NP, bs, N, nbatch, dinfo, h + off_res));
    GPT_HIP_CHECK(hipStreamSynchronize(st));
    for (int b = 0; b < nbatch; b++) {
        const double logdet_half = h[off_res + 4 * b], zz = h[off_res + 4 * b + 1];
        int32_t info = (int32_t)h[off_res + 4 * b + 2];
        if (info > N) info = (int32_t)N;                       // only the augmented pivot failed (non-finite data)
        const double ll = -0.5 * zz - logdet_half - 0.5 * (double)N * log(2.0 * M_PI);
        if (info == 0 && !(ll == ll)) info = (int32_t)N;
        info_out[b] = info;
        ll_data_out[b] = ll;
        if (logdet_half_out) logdet_half_out[b] = logdet_half;
    }
    return GPT_OK;
}

extern "C" int gpt_last_timings(gpt_ctx *c, double *out_ms, int n)
{
    if (!c || !out_ms) return GPT_E_ARG;
    const int cnt = n < 5 ? n : 5;
    for (int i = 0; i < cnt; i++) out_ms[i] = c->timings[i];
    return cnt;
}

// Fold the finished launches' event pairs into the running sums.  Must run after the streams have drained and before
// the next factorisation re-records the edge events that double as stop events.
static int harvest_gemm_profile(gpt_ctx *c)
{
    for (size_t i = 0; i < c->gprof_used; i++) {
        float t = 0;
        GPT_HIP_CHECK(hipEventElapsedTime(&t, c->gprof[i].e0, c->gprof[i].stop));
        c->prof_ms += t;
        c->prof_flops += c->gprof[i].flops;
    }
    c->prof_count += (double)c->gprof_used;
    c->gprof_used = 0;
    return GPT_OK;
}

extern "C" int gpt_gemm_profile_read(gpt_ctx *c, double *out3)
{
    CTX_ENTER(c);
    if (!out3) return GPT_E_ARG;
    GPT_HIP_CHECK(hipStreamSynchronize(c->stream));
    GPT_HIP_CHECK(hipStreamSynchronize(c->panel_stream));
    GPT_TRY(harvest_gemm_profile(c));
    out3[0] = c->prof_flops;
    out3[1] = c->prof_ms;
    out3[2] = c->prof_count;
    c->prof_flops = c->prof_ms = c->prof_count = 0.0;
    return GPT_OK;
}

#define NEED_FACTOR(c)                                                        \
    do {                                                                      \
        if (!(c)->factored) {                                                 \
            gpt_set_error("no valid factorisation resident (call gpt_fit)");  \
            return GPT_E_STATE;                                               \
        }                                                                     \
    } while (0)

extern "C" int gpt_get_L(gpt_ctx *c, double *L_out)
{
    CTX_ENTER(c);
    NEED_FACTOR(c);
    if (!L_out) return GPT_E_ARG;
    const int64_t N = c->N;
    double *dlow;
    GPT_TRY(ensure(c, SLOT_LOW, (size_t)N * N * sizeof(double), (void **)&dlow));
    GPT_TRY(launch_extract_lower(c->stream, c->dA, c->NP, N, dlow, N));
    GPT_HIP_CHECK(hipMemcpyAsync(L_out, dlow, (size_t)N * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    GPT_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GPT_OK;
}

// alpha = L^-T z with z = the augmented row (z = L^-1 y)   (ref: gaussian_process.py:1462, the second half of cho_solve)
// In 512-wide steps against the inverse transposes of the factor's 512 x 512 diagonal blocks (solve.hip launch_trsv_lt_wide;
// the blocks' inverses are the ones predict and the gradient use, built once per factorisation -- by ONE launch since round 5,
// trinv512_kernel): 2 N / 512 dependent launches instead of 2 N / 128.  N = 8192: 128-wide 1.0 ms; 512-wide with the inverses
// from 15 batched launches (rounds 2-4) 0.43 ms; now 0.34 ms, 0.24 of them the 31 launches of the substitution.  What is left of
// the order beyond the last whole 512-block falls in 128-wide steps first (last rows first).
// (Round 5 also built 1024-wide steps -- the off-diagonal quadrant of a 1024-block's inverse transpose, U12 = -(U11 L21^T) U22,
// from two batched GEMMs: 15 launches instead of 31 save 95 us, the two GEMMs are 2 x 2.1 GFLOP = 119 us.  Removed.)
// GPT_ALPHA_NARROW=1 (measurement aid): the 128-wide form throughout.  Option binv_launches = 1: the inverses by rounds 2-4's launches.
static int ensure_block_inverses(gpt_ctx *c, int64_t nb, int64_t nfull, double **out, double **out_u = nullptr);
static int ensure_alpha(gpt_ctx *c)
{
    if (c->alpha_valid) return GPT_OK;
    const int64_t N = c->N, NP = c->NP, n128 = round_up(N, 128);
    hipStream_t st = c->stream;
    static const bool narrow = getenv("GPT_ALPHA_NARROW") != nullptr;
    const int64_t nwide = (n128 / 512) * 512;
    GPT_HIP_CHECK(hipMemsetAsync(c->d_alpha, 0, (size_t)2 * NP * sizeof(double), st));
    if (narrow || nwide < 1024) {
        GPT_HIP_CHECK(hipMemcpyAsync(c->d_alpha, c->dA + N * NP, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, st));
        GPT_TRY(launch_trsv_lt(st, n128, c->dA, NP, c->d_invd, c->d_alpha));
        c->alpha_valid = true;
        return GPT_OK;
    }
    double *W = nullptr, *U = nullptr, *w = c->d_alpha + NP;
    GPT_TRY(ensure_block_inverses(c, 512, nwide, &W, &U));
    GPT_HIP_CHECK(hipMemcpyAsync(w, c->dA + N * NP, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (n128 > nwide) {
        GPT_TRY(launch_trsv_lt(st, n128, c->dA, NP, c->d_invd, w, nwide / 128));
        GPT_HIP_CHECK(hipMemcpyAsync(c->d_alpha + nwide, w + nwide, (size_t)(n128 - nwide) * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    GPT_TRY(launch_trsv_lt_wide(st, nwide, c->dA, NP, U, w, c->d_alpha));
    c->alpha_valid = true;
    return GPT_OK;
}

// alpha to the pinned landing buffer, behind whatever computes it on the main stream (valid for the host after the stream's next sync)
static int alpha_to_host(gpt_ctx *c)
{
    GPT_TRY(ensure_alpha(c));
    if (c->h_alpha_valid) return GPT_OK;
    GPT_HIP_CHECK(hipMemcpyAsync(c->h_alpha, c->d_alpha, (size_t)c->N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    c->h_alpha_valid = true;
    return GPT_OK;
}

extern "C" int gpt_get_alpha(gpt_ctx *c, double *alpha_out)
{
    CTX_ENTER(c);
    NEED_FACTOR(c);
    if (!alpha_out) return GPT_E_ARG;
    const bool landed = c->alpha_valid && c->h_alpha_valid;      // (an eager evaluation: the fit's own sync covered the copy)
    if (!landed) {
        GPT_TRY(alpha_to_host(c));
        GPT_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    memcpy(alpha_out, c->h_alpha, (size_t)c->N * sizeof(double));
    return GPT_OK;
}

// ------------------------------------------------------------------------------------------------
// analytic gradient of the LML data term (ref: gptools/gaussian_process.py:1471-1520)
// ------------------------------------------------------------------------------------------------
// U = L^-T (upper triangular, row-major) for the block range [lo, hi) of the resident factor.  U must hold the identity
// on its 128x128 diagonal blocks and zeros elsewhere on entry.  [[U11, U12], [0, U22]] with
// U12 = -(U11 L21^T) L22^-T: one NT GEMM and one right-TRSM per level, leaves by the packed-inverse panel kernel.
static int trtri_u(gpt_ctx *c, hipStream_t st, int64_t lo, int64_t hi, const double *L, int64_t ldl, const double *ws,
                   double *U, int64_t ldu)
{
    const int64_t n = hi - lo;
    if (n == 128)
        return launch_trsm_panel(st, 128, L + lo * ldl + lo, ldl, ws + (lo / 128) * GPT_WS_BLOCK, U + lo * ldu + lo, ldu);
    const int64_t h = (n / 256) * 128 > 0 ? (n / 256) * 128 : 128, mid = lo + h;
    GPT_TRY(trtri_u(c, st, lo, mid, L, ldl, ws, U, ldu));
    GPT_TRY(trtri_u(c, st, mid, hi, L, ldl, ws, U, ldu));
    double *U12 = U + lo * ldu + mid;
    // U11 is upper triangular: row chunk q only has non-zeros from its own first column on, so its k range starts there
    const int64_t nq = (h >= 2048) ? 8 : (h >= 512 ? 4 : 1), hc = h / nq;
    for (int64_t q = 0; q < nq; q++) {
        const int64_t o = q * hc;
        GPT_TRY(gemm_nt(c, st, hc, hi - mid, h - o, -1.0, U + (lo + o) * ldu + lo + o, ldu, L + mid * ldl + lo + o, ldl,
                        0.0, U12 + o * ldu, ldu, 0));
    }
    return trsm_rlt(c, st, h, hi - mid, L + mid * ldl + mid, ldl, ws + (mid / 128) * GPT_WS_BLOCK, U12, ldu);
}

__global__ void eye_blocks_kernel(double *__restrict__ U, int64_t ldu, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) U[i * ldu + i] = 1.0;
}

// W (n x n, row-major) = U^T through a 32x33 LDS tile
__global__ __launch_bounds__(256) void transpose_kernel(const double *__restrict__ U, int64_t ldu, double *__restrict__ W,
                                                        int64_t ldw, int64_t n)
{
    __shared__ double t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < n && c0 + tx < n) t[i][tx] = U[(r0 + i) * ldu + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < n && r0 + tx < n) W[(c0 + i) * ldw + r0 + tx] = t[tx][i];
}

// dW (n x n, row-major, zeros above the diagonal) <- L^-1 for a factored diagonal block and its workspace: the identity
// pushed through the panel TRSM (U = I L^-T) and one transpose.  With it the TRSM of a tall block of
// rows becomes ONE fp64-MFMA GEMM, X = B * (L^-1)^T = gemm_nt(B, dW) -- twice the flops of the substitution, but at the
// GEMM's rate instead of four latency-bound 128-column leaves and three narrow updates (31k x 512: 0.33 against
// 0.86 ms).  Used by gptools_amd/dist.py for the rows of a panel below its head.
extern "C" int gpt_dev_trinv(gpt_ctx *c, int64_t n, const double *dL, int64_t ldl, const double *d_invd, double *dW,
                             int64_t ldw)
{
    CTX_ENTER(c);
    if (n <= 0 || n % 128 || ldl < n || ldw < n || !dL || !d_invd || !dW) {
        gpt_set_error("trinv: n must be a positive multiple of 128");
        return GPT_E_ARG;
    }
    hipStream_t st = c->stream;
    double *U;
    GPT_TRY(ensure(c, SLOT_UINV, (size_t)n * n * sizeof(double), (void **)&U));
    // (the partitioned engines' block size: one launch instead of ten -- 69 -> ~43 us on the panel chain and nine launches less of
    // host enqueue per panel; solve.hip trinv512_kernel)
    if (n == 512 && ldw == 512 && !c->binv_launches) return launch_trinv512(st, 1, dL, ldl, d_invd, U, dW);
    GPT_HIP_CHECK(hipMemsetAsync(U, 0, (size_t)n * n * sizeof(double), st));
    hipLaunchKernelGGL(eye_blocks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, U, n, n);
    GPT_LAUNCH_CHECK();
    // U = I L^-T by the panel TRSM (7 launches at n = 512; the structured trtri_u of the gradient path needs 12 and the
    // zeros it would skip are not worth a launch at this size)
    GPT_TRY(trsm_rlt(c, st, n, n, dL, ldl, d_invd, U, n));
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32)), dim3(256), 0, st, U, n,
                       dW, ldw, n);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

#define GPT_BINV_NB 512
#define GPT_BINV_NB2 1024
static int trsm_rlt_binv(gpt_ctx *c, hipStream_t st, int64_t m, int64_t nb, int64_t lo, int64_t hi, const double *W, double *B,
                         int64_t ldb, double *V, int64_t ldv);

// dst (rows x cols, ldd) = src (cols x rows, lds)^T through a 32x33 LDS tile
__global__ __launch_bounds__(256) void transpose_rect_kernel(const double *__restrict__ src, int64_t lds, double *__restrict__ dst,
                                                             int64_t ldd, int64_t rows, int64_t cols, int64_t sstride = 0,
                                                             int64_t dstride = 0)
{
    __shared__ double t[32][33];
    src += (int64_t)blockIdx.z * sstride;                   // (blockIdx.z: a batch of equal blocks)
    dst += (int64_t)blockIdx.z * dstride;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;      // tile of dst
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) t[i][tx] = src[(c0 + i) * lds + r0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) dst[(r0 + i) * ldd + c0 + tx] = t[tx][i];
}

// U = L^-T over the block range [lo, hi) of the resident factor (multiples of GPT_BINV_NB = 512), GEMMs only:
//   leaf:   U_jj = W_j^T, W_j = L_jj^-1 the 512 x 512 block inverses that the many-right-hand-side solves keep anyway;
//   level:  U12 = -(U11 L21^T) L22^-T -- the product into the scratch matrix T (row chunks of U11 start their k range at
//           their own first column: U11 is upper triangular), then the right-TRSM against L22 as trsm_rlt_binv, whose
//           leaves are GEMMs against the block inverses and whose result lands in U.
// Round 2 did the TRSM by substitution in 128-column leaves (trtri_u above, kept for the ragged last block): 45 ms at
// N = 16384, 32 TFLOP/s on N^3/3.
static int trtri_u_gemm(gpt_ctx *c, hipStream_t st, int64_t lo, int64_t hi, const double *Wb, double *U, int64_t ldu, double *T,
                        int64_t ldt)
{
    const int64_t nb = GPT_BINV_NB, n = hi - lo;
    if (n == nb) {
        hipLaunchKernelGGL(transpose_rect_kernel, dim3((unsigned)(nb / 32), (unsigned)(nb / 32)), dim3(256), 0, st, Wb + lo * nb, nb,
                           U + lo * ldu + lo, ldu, nb, nb);
        GPT_LAUNCH_CHECK();
        return GPT_OK;
    }
    const int64_t h = (n / (2 * nb)) * nb > 0 ? (n / (2 * nb)) * nb : nb, mid = lo + h;
    GPT_TRY(trtri_u_gemm(c, st, lo, mid, Wb, U, ldu, T, ldt));
    GPT_TRY(trtri_u_gemm(c, st, mid, hi, Wb, U, ldu, T, ldt));
    const double *L = c->dA;
    const int64_t ldl = c->NP;
    const int64_t nq = (h >= 2048) ? 8 : (h >= 1024 ? 4 : 1), hc = h / nq;
    for (int64_t q = 0; q < nq; q++) {
        const int64_t o = q * hc;
        GPT_TRY(gemm_nt(c, st, hc, hi - mid, h - o, -1.0, U + (lo + o) * ldu + lo + o, ldu, L + mid * ldl + lo + o, ldl, 0.0,
                        T + (lo + o) * ldt + mid, ldt, 0));
    }
    return trsm_rlt_binv(c, st, h, nb, mid, hi, Wb, T + lo * ldt, ldt, U + lo * ldu, ldu);
}

// The same for `nbatch` equal diagonal blocks at once (the nb-wide inverses of few-rows solves: nb = 1024 / 2048, block J at
// factor rows [J nb, J nb + nb)): every GEMM and transpose of the recursion is ONE launch with the blocks as its batch dimension
// (round 5; block by block the 2048-wide inverses of N = 8192 were 64 dependent launches, 0.9 ms of the first predict after a fit).
// Indices are those of block 0; U and T are strips of nb-wide blocks (block J at rows [J nb, J nb + nb), row stride nb).
struct WideBatch { int64_t n, sU, sT, sL, sW; };
static int gemm_b(hipStream_t st, int64_t m, int64_t n, int64_t k, double alpha, const double *A, int64_t lda, int64_t sA, const double *B,
                  int64_t ldb, int64_t sB, double beta, double *C, int64_t ldc, int64_t sC, int64_t nbatch)
{
    return launch_gemm_nt(st, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, 0, 0, 0, nullptr, nullptr, 0, EdgeSig(), EdgeSig(), 0, nbatch, sA,
                          EdgeSig(), sB, sC);
}
static int trsm_rlt_binv_b(gpt_ctx *c, hipStream_t st, int64_t m, int64_t nb, int64_t lo, int64_t hi, const double *W, double *B, int64_t ldb,
                           double *V, int64_t ldv, const WideBatch &wb)
{
    const int64_t n = hi - lo;
    if (n == nb) return gemm_b(st, m, nb, nb, 1.0, B + lo, ldb, wb.sT, W + lo * nb, nb, wb.sW, 0.0, V + lo, ldv, wb.sU, wb.n);
    const int64_t h = (n / (2 * nb)) * nb > 0 ? (n / (2 * nb)) * nb : nb;
    const int64_t mid = lo + h;
    GPT_TRY(trsm_rlt_binv_b(c, st, m, nb, lo, mid, W, B, ldb, V, ldv, wb));
    GPT_TRY(gemm_b(st, m, hi - mid, h, -1.0, V + lo, ldv, wb.sU, c->dA + mid * c->NP + lo, c->NP, wb.sL, 1.0, B + mid, ldb, wb.sT, wb.n));
    return trsm_rlt_binv_b(c, st, m, nb, mid, hi, W, B, ldb, V, ldv, wb);
}
static int trtri_u_gemm_b(gpt_ctx *c, hipStream_t st, int64_t lo, int64_t hi, const double *Wb, double *U, int64_t ldu, double *T, int64_t ldt,
                          const WideBatch &wb)
{
    const int64_t nb = GPT_BINV_NB, n = hi - lo;
    if (n == nb) {
        hipLaunchKernelGGL(transpose_rect_kernel, dim3((unsigned)(nb / 32), (unsigned)(nb / 32), (unsigned)wb.n), dim3(256), 0, st,
                           Wb + lo * nb, nb, U + lo * ldu + lo, ldu, nb, nb, wb.sW, wb.sU);
        GPT_LAUNCH_CHECK();
        return GPT_OK;
    }
    const int64_t h = (n / (2 * nb)) * nb > 0 ? (n / (2 * nb)) * nb : nb, mid = lo + h;
    GPT_TRY(trtri_u_gemm_b(c, st, lo, mid, Wb, U, ldu, T, ldt, wb));
    GPT_TRY(trtri_u_gemm_b(c, st, mid, hi, Wb, U, ldu, T, ldt, wb));
    const int64_t nq = (h >= 1024) ? 2 : 1, hc = h / nq;          // (row chunks of U11 start their k range at their own first column)
    for (int64_t q = 0; q < nq; q++) {
        const int64_t o = q * hc;
        GPT_TRY(gemm_b(st, hc, hi - mid, h - o, -1.0, U + (lo + o) * ldu + lo + o, ldu, wb.sU, c->dA + mid * c->NP + lo + o, c->NP, wb.sL, 0.0,
                       T + (lo + o) * ldt + mid, ldt, wb.sT, wb.n));
    }
    return trsm_rlt_binv_b(c, st, h, nb, mid, hi, Wb, T + lo * ldt, ldt, U + lo * ldu, ldu, wb);
}

extern "C" int gpt_ll_grad(gpt_ctx *c, int nh, const int *term_idx, const int *local_idx, double *out)
{
    CTX_ENTER(c);
    NEED_FACTOR(c);
    if (!c->have_kernel) {
        gpt_set_error("gpt_ll_grad needs a factorisation produced by gpt_fit / gpt_fit_sum");
        return GPT_E_STATE;
    }
    if (nh < 0 || (nh > 0 && (!term_idx || !local_idx)) || !out) return GPT_E_ARG;
    for (int h = 0; h < nh; h++) {
        if (term_idx[h] < 0 || term_idx[h] >= (int)c->terms.size()) return GPT_E_ARG;
        const KParams &t = c->terms[(size_t)term_idx[h]];
        const bool is_prod = (size_t)term_idx[h] < c->terms2.size() && c->terms2[(size_t)term_idx[h]].kernel_id >= 0;
        if (t.kernel_id != GPT_KERNEL_SE || is_prod) {
            gpt_set_error("hyper-parameter derivatives exist for the squared-exponential kernel only "
                          "(ref: matern.py:543-544)");
            return GPT_E_NOTIMPL;
        }
        if (local_idx[h] < 0 || local_idx[h] > c->D) return GPT_E_ARG;
    }
    const int64_t N = c->N, NP = c->NP;
    // Everything here runs on the panel stream: it is not CU-masked (the main stream leaves 32 of the 256 CUs to it), and
    // nothing else is in flight.
    hipStream_t st = c->panel_stream;
    {
        hipEvent_t e = get_event(c, 0);
        if (!e) return GPT_E_HIP;
        GPT_HIP_CHECK(hipEventRecord(e, c->stream));
        GPT_HIP_CHECK(hipStreamWaitEvent(st, e, 0));
    }
    double *U, *W, *dpart;
    // K_tot^-1 = L^-T L^-1 = U U^T: triangular inverse (N^3/3 flop), then the lower half of U U^T block row by block row
    // with k starting at the diagonal (N^3/3) -- all on the fp64-MFMA GEMM
    const bool gt = getenv("GPT_GRAD_TIMING") != nullptr;
    hipEvent_t ge[4] = {nullptr, nullptr, nullptr, nullptr};
    if (gt) {
        for (auto &e : ge) GPT_HIP_CHECK(hipEventCreate(&e));
        GPT_HIP_CHECK(hipEventRecord(ge[0], st));
    }
    const int64_t nfull = (NP / GPT_BINV_NB) * GPT_BINV_NB;
    double *Wb = nullptr;
    GPT_TRY(ensure(c, SLOT_UINV, (size_t)NP * NP * sizeof(double), (void **)&U));
    GPT_TRY(ensure(c, SLOT_WINV, (size_t)NP * NP * sizeof(double), (void **)&W));
    if (nfull >= 2 * GPT_BINV_NB) {
        // (the block inverses are built on the main stream with the head of SLOT_UINV as their scratch: before U is touched)
        GPT_TRY(ensure_block_inverses(c, GPT_BINV_NB, nfull, &Wb));
        hipEvent_t e = get_event(c, 1);
        if (!e) return GPT_E_HIP;
        GPT_HIP_CHECK(hipEventRecord(e, c->stream));
        GPT_HIP_CHECK(hipStreamWaitEvent(st, e, 0));
    }
    GPT_TRY(ensure_alpha(c));                                  // (main stream; joined below before the pair pass)
    if (c->debug_poison)                                       // test aid: every byte 0xff = NaN in whatever is not written below
        GPT_HIP_CHECK(hipMemsetAsync(W, 0xff, (size_t)NP * NP * sizeof(double), st));
    GPT_HIP_CHECK(hipMemsetAsync(U, 0, (size_t)NP * NP * sizeof(double), st));
    if (Wb) {
        GPT_TRY(trtri_u_gemm(c, st, 0, nfull, Wb, U, NP, W, NP));           // (W doubles as the scratch matrix T)
        if (nfull < NP) {
            // ragged last block [nfull, NP): by substitution, then its column block of U as above with an in-place TRSM
            hipLaunchKernelGGL(eye_blocks_kernel, dim3((unsigned)((NP - nfull + 255) / 256)), dim3(256), 0, st,
                               U + nfull * NP + nfull, NP, NP - nfull);
            GPT_LAUNCH_CHECK();
            GPT_TRY(trtri_u(c, st, nfull, NP, c->dA, NP, c->d_invd, U, NP));
            const int64_t nq = 8, hc = nfull / nq / 64 * 64;
            for (int64_t o = 0; o < nfull; o += hc) {
                const int64_t rows = (nfull - o < hc || o + 2 * hc > nfull) ? nfull - o : hc;
                GPT_TRY(gemm_nt(c, st, rows, NP - nfull, nfull - o, -1.0, U + o * NP + o, NP, c->dA + nfull * NP + o, NP, 0.0,
                                U + o * NP + nfull, NP, 0));
                if (rows != hc) break;
            }
            GPT_TRY(trsm_rlt(c, st, nfull, NP - nfull, c->dA + nfull * NP + nfull, NP, c->d_invd + (nfull / 128) * GPT_WS_BLOCK,
                             U + nfull, NP));
        }
    } else {
        hipLaunchKernelGGL(eye_blocks_kernel, dim3((unsigned)((NP + 255) / 256)), dim3(256), 0, st, U, NP, NP);
        GPT_LAUNCH_CHECK();
        GPT_TRY(trtri_u(c, st, 0, NP, c->dA, NP, c->d_invd, U, NP));
    }
    // (block rows of about NP/8: skinnier launches exploit more of U's zeros but run the GEMM far below its rate)
    if (gt) GPT_HIP_CHECK(hipEventRecord(ge[1], st));
    int64_t nbdiv = 8;
    if (const char *e = getenv("GPT_GRAD_ROWS_DIV")) nbdiv = atoi(e) > 0 ? atoi(e) : 8;
    const int64_t nb = (NP / nbdiv >= 512) ? (NP / nbdiv) / 128 * 128 : 512;
    for (int64_t r0 = 0; r0 < NP; r0 += nb) {
        const int64_t rows = (NP - r0 < nb) ? NP - r0 : nb;
        // the block row left of the diagonal block as a rectangle, the diagonal block itself as a lower trapezoid (the pair
        // pass reads the lower triangle of W only)
        if (r0 > 0)
            GPT_TRY(gemm_nt(c, st, rows, r0, NP - r0, 1.0, U + r0 * NP + r0, NP, U + r0, NP, 0.0, W + r0 * NP, NP, 0));
        GPT_TRY(gemm_nt(c, st, rows, rows, NP - r0, 1.0, U + r0 * NP + r0, NP, U + r0 * NP + r0, NP, 0.0, W + r0 * NP + r0, NP, 1));
    }
    if (gt) GPT_HIP_CHECK(hipEventRecord(ge[2], st));
    {   // alpha (main stream) is needed from here on
        hipEvent_t e = get_event(c, 0);
        GPT_HIP_CHECK(hipEventRecord(e, c->stream));
        GPT_HIP_CHECK(hipStreamWaitEvent(st, e, 0));
    }
    // With a linear transform (ref :1499-1500, dK_tot = T dK T^T):  tr(K_tot^-1 T dK T^T) = tr((T^T K_tot^-1 T) dK) and
    // alpha^T T dK T^T alpha = (T^T alpha)^T dK (T^T alpha): the pair pass runs over the Nx LATENT points with
    // W' = T^T W T (two GEMMs with the resident T) and alpha' = T^T alpha.  The noise entry (out[nh]) is taken from the
    // untransformed pair afterwards (see below).
    const double *Wp = W, *ap = c->d_alpha;
    int64_t Np = N, ldwp = NP;
    double *Wt = nullptr, *at = nullptr;
    if (c->dT) {
        const int64_t Nx = c->Nx, NxP = c->NxP, NyP = round_up(N, 64), NxQ = round_up(Nx, 64);
        double *TT, *Yt;
        // T^T (NxQ x NyP, zero padded), Y^T = T^T W (NxQ x NyP; W made fully symmetric first), W' = T^T Y (lower)
        GPT_TRY(ensure(c, SLOT_TK, (size_t)NxQ * NyP * sizeof(double) * 2, (void **)&TT));
        Yt = TT + NxQ * NyP;
        GPT_HIP_CHECK(hipMemsetAsync(TT, 0, (size_t)NxQ * NyP * sizeof(double), st));
        hipLaunchKernelGGL(transpose_rect_kernel, dim3((unsigned)((N + 31) / 32), (unsigned)((Nx + 31) / 32)), dim3(256), 0, st,
                           c->dT, NxP, TT, NyP, Nx, N);
        GPT_LAUNCH_CHECK();
        // (the GEMM below runs its k loop to NyP = round_up(N, 64) and reads W[j][k] for every j, k < NyP: the whole NyP x NyP
        // block must hold finite numbers -- its lower triangle does (rows >= N: the inverse of the padded matrix' augmented /
        // padding part, multiplied by the zero padding of T^T), the strictly upper entries only once mirrored.  Mirroring just
        // the N x N part left the columns [N, NyP) to whatever the scratch slot held: ADVICE r3.)
        GPT_TRY(launch_mirror_rows(st, W, NP, 0, NyP, NyP));
        GPT_TRY(gemm_nt(c, st, NxQ, NyP, NyP, 1.0, TT, NyP, W, NP, 0.0, Yt, NyP, 0));
        GPT_TRY(ensure(c, SLOT_KFULL, ((size_t)NxQ * NxQ + (size_t)NxQ) * sizeof(double), (void **)&Wt));
        at = Wt + NxQ * NxQ;
        GPT_TRY(gemm_nt(c, st, NxQ, NxQ, NyP, 1.0, TT, NyP, Yt, NyP, 0.0, Wt, NxQ, 1));
        GPT_TRY(launch_gemv_n(st, Nx, N, TT, NyP, c->d_alpha, at));
        Wp = Wt;
        ap = at;
        Np = Nx;
        ldwp = NxQ;
    }
    // sum_ab (alpha_a alpha_b - W_ab) dK_h[a][b], per kernel term, GPT_GRAD_MAXH parameters per launch
    const int nblk = grad_reduce_blocks(Np);
    GPT_TRY(ensure(c, SLOT_GPART, (size_t)nblk * (GPT_GRAD_MAXH + 1) * sizeof(double), (void **)&dpart));
    std::vector<double> hpart((size_t)nblk * (GPT_GRAD_MAXH + 1));
    bool have_trace = false;
    for (size_t t = 0; t < c->terms.size() || !have_trace; t++) {
        std::vector<int> hs, where;
        if (t < c->terms.size())
            for (int h = 0; h < nh; h++)
                if ((size_t)term_idx[h] == t) {
                    hs.push_back(local_idx[h]);
                    where.push_back(h);
                }
        if (hs.empty() && have_trace) continue;
        const KParams &kp = c->terms[t < c->terms.size() ? t : 0];
        for (size_t b0 = 0; b0 < hs.size() || !have_trace; b0 += GPT_GRAD_MAXH) {
            const int cnt = (int)((hs.size() - b0 < (size_t)GPT_GRAD_MAXH) ? hs.size() - b0 : (size_t)GPT_GRAD_MAXH);
            KParams ks = kp;
            ks.symmetric = 1;
            GPT_TRY(launch_grad_reduce(st, ks, cnt > 0 ? cnt : 0, cnt > 0 ? hs.data() + b0 : nullptr, c->dX, c->dn, Np,
                                       ap, Wp, ldwp, dpart));
            GPT_HIP_CHECK(hipMemcpyAsync(hpart.data(), dpart, hpart.size() * sizeof(double), hipMemcpyDeviceToHost, st));
            GPT_HIP_CHECK(hipStreamSynchronize(st));
            for (int q = 0; q < cnt; q++) {
                double sum = 0.0;
                for (int b = 0; b < nblk; b++) sum += hpart[(size_t)b * (GPT_GRAD_MAXH + 1) + q];
                out[where[b0 + q]] = 0.5 * sum;
            }
            if (!have_trace) {
                double sum = 0.0;
                for (int b = 0; b < nblk; b++) sum += hpart[(size_t)b * (GPT_GRAD_MAXH + 1) + GPT_GRAD_MAXH];
                out[nh] = 0.5 * sum;
                have_trace = true;
            }
            if (cnt <= 0) break;
        }
    }
    if (c->dT) {
        // The noise entry with a transform: the reference differentiates the DiagonalNoiseKernel as 2 sigma_n I over the
        // OBSERVATIONS, not transformed by T (ref :1482-1488, dK = 2 sigma_n eye(len(y))): out[nh] = 1/2 sum_i (alpha_i^2 -
        // (K_tot^-1)_ii) from the untransformed pair, replacing what the pass over the latent points left there.
        GPT_TRY(launch_alpha_trace(st, c->d_alpha, W, NP, N, dpart));
        double tr = 0.0;
        GPT_HIP_CHECK(hipMemcpyAsync(&tr, dpart, sizeof(double), hipMemcpyDeviceToHost, st));
        GPT_HIP_CHECK(hipStreamSynchronize(st));
        out[nh] = 0.5 * tr;
    }
    if (gt) {
        GPT_HIP_CHECK(hipEventRecord(ge[3], st));
        GPT_HIP_CHECK(hipStreamSynchronize(st));
        float a = 0, b = 0, d = 0;
        hipEventElapsedTime(&a, ge[0], ge[1]);
        hipEventElapsedTime(&b, ge[1], ge[2]);
        hipEventElapsedTime(&d, ge[2], ge[3]);
        fprintf(stderr, "gpt_ll_grad N=%lld: triangular inverse %.2f ms, U U^T %.2f ms, pair pass %.2f ms\n", (long long)N, a, b, d);
        for (auto &e : ge) hipEventDestroy(e);
    }
    {   // the main stream continues behind this (the block inverses stay valid for predict)
        hipEvent_t e = get_event(c, 1);
        GPT_HIP_CHECK(hipEventRecord(e, st));
        GPT_HIP_CHECK(hipStreamWaitEvent(c->stream, e, 0));
    }
    return GPT_OK;
}

// ------------------------------------------------------------------------------------------------
// solves with many right-hand sides against the resident factor: 512-column leaves by explicit block inverses
// ------------------------------------------------------------------------------------------------
// B (m x n) <- B L^-T is a chain of n/128 (leaf TRSM + update) pairs, ~20 us each: 2.5 ms at N = 8192 however few rows B
// has (predict at a handful of points, gpt_solve_L).  With W_j = L_jj^-1 of the 512x512 diagonal blocks at hand (built
// once per factorisation, on first use: 10 launches per block) a leaf is one copy and one GEMM, B_j <- B_j W_j^T, and
// the chain has a quarter of the links: predict with std at M <= 256, N = 8192 goes from 2.55 to ~0.7 ms once the
// inverses exist.
// W (nfull x nb, block j at row j): the inverses of the nb x nb diagonal blocks of the resident factor, built on first
// use after a factorisation (identity pushed through the panel TRSM, transposed).  nb = 512 for every solve, and
// additionally nb = 1024 for solves with at most GPT_FEW_ROWS rows, whose time is the LENGTH of the chain of dependent
// GEMMs (two per block), not their flops.
#define GPT_FEW_ROWS 128
__global__ void eye_strip_kernel(double *__restrict__ U, int64_t n, int64_t w)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) U[i * w + i % w] = 1.0;
}

// W_b (w x w) = U_b^T for every block b of a strip (blockIdx.z), through a 32x33 LDS tile
__global__ __launch_bounds__(256) void transpose_strip_kernel(const double *__restrict__ U, double *__restrict__ W, int64_t w)
{
    __shared__ double t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32, o = (int64_t)blockIdx.z * w * w;
    for (int i = ty; i < 32; i += 8) t[i][tx] = U[o + (r0 + i) * w + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8) W[o + (c0 + i) * w + r0 + tx] = t[tx][i];
}

// The inverses of ALL 512 x 512 diagonal blocks of the resident factor at once: U (strip, block j at rows [512 j, 512 j + 512),
// row stride 512) = L_jj^-T, W = L_jj^-1 = its transpose.  The recursion of trtri_u -- leaves by the panel TRSM on the identity,
// U12 = -(U11 L21^T) L22^-T per level -- with every launch BATCHED over the blocks (blockIdx.y; the three operands of the
// GEMMs lie in three arrays with a stride each): 2 + 4 + 4 + 4 + 1 = 15 launches whatever the order, where the per-block
// loop of round 3 needed 10 per block (N = 8192: 160 launches, ~1.3 ms before the first predict / gradient / alpha).
// (Rounds 2-4's builder: since round 5 only behind option binv_launches = 1 -- the default is ONE launch, solve.hip trinv512_kernel,
// 205 -> 59 us at N = 8192 -- kept as the same-process A/B baseline and as a second implementation for the tests.)
static int build_block_inverses_512(gpt_ctx *c, hipStream_t st, int64_t nblk, double *U, double *W)
{
    const int64_t nb = 512, bs = nb * nb, NP = c->NP, bsl = nb * (NP + 1), WS = GPT_WS_BLOCK;
    GPT_HIP_CHECK(hipMemsetAsync(U, 0, (size_t)nblk * bs * sizeof(double), st));
    hipLaunchKernelGGL(eye_strip_kernel, dim3((unsigned)((nblk * nb + 255) / 256)), dim3(256), 0, st, U, nblk * nb, nb);
    GPT_LAUNCH_CHECK();
    for (int p = 0; p < 4; p++)                                        // leaves: the identity through the panel TRSM
        GPT_TRY(launch_trsm_panel(st, 128, nullptr, 0, c->d_invd + p * WS, U + (int64_t)p * 128 * nb + p * 128, nb, nullptr, EdgeSig(),
                                  nblk, bs, 4 * WS));
    for (int q = 0; q < 4; q += 2) {                                   // 128 -> 256: the pairs (0, 1) and (2, 3) of every block
        const int64_t lo = q * 128, mid = lo + 128;
        GPT_TRY(launch_gemm_nt(st, 128, 128, 128, -1.0, U + lo * nb + lo, nb, c->dA + mid * NP + lo, NP, 0.0, U + lo * nb + mid, nb, 0,
                               0, 0, nullptr, nullptr, 0, EdgeSig(), EdgeSig(), 0, nblk, bs, EdgeSig(), bsl, bs));
        GPT_TRY(launch_trsm_panel(st, 128, nullptr, 0, c->d_invd + (q + 1) * WS, U + lo * nb + mid, nb, nullptr, EdgeSig(), nblk, bs,
                                  4 * WS));
    }
    // 256 -> 512: U12 = -(U11 L21^T), then the right-TRSM against L22 in two 128-column leaves
    GPT_TRY(launch_gemm_nt(st, 256, 256, 256, -1.0, U, nb, c->dA + 256 * NP, NP, 0.0, U + 256, nb, 0, 0, 0, nullptr, nullptr, 0, EdgeSig(),
                           EdgeSig(), 0, nblk, bs, EdgeSig(), bsl, bs));
    GPT_TRY(launch_trsm_panel(st, 256, nullptr, 0, c->d_invd + 2 * WS, U + 256, nb, nullptr, EdgeSig(), nblk, bs, 4 * WS));
    GPT_TRY(launch_gemm_nt(st, 256, 128, 128, -1.0, U + 256, nb, c->dA + 384 * NP + 256, NP, 1.0, U + 384, nb, 0, 0, 0, nullptr, nullptr, 0,
                           EdgeSig(), EdgeSig(), 0, nblk, bs, EdgeSig(), bsl, bs));
    GPT_TRY(launch_trsm_panel(st, 256, nullptr, 0, c->d_invd + 3 * WS, U + 384, nb, nullptr, EdgeSig(), nblk, bs, 4 * WS));
    hipLaunchKernelGGL(transpose_strip_kernel, dim3((unsigned)(nb / 32), (unsigned)(nb / 32), (unsigned)nblk), dim3(256), 0, st, U, W, nb);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

// (the inverses are always built for the WHOLE padded order -- floor(NP / nb) blocks -- whatever extent `nfull` the
// caller is going to use: gpt_ll_grad asks for floor(NP / 512) blocks, the solves for floor(round_up(N, 128) / 512), one block
// less at N = 512 k - 128; a strip sized for the smaller request used to be reallocated, and not rebuilt, by the larger one)
static int ensure_block_inverses_wide(gpt_ctx *c, int64_t nb, int64_t nfull, double **out);
static int ensure_block_inverses(gpt_ctx *c, int64_t nb, int64_t nfull, double **out, double **out_u)
{
    if (nb == GPT_BINV_NB2) return ensure_block_inverses_wide(c, nb, nfull, out);
    const int64_t nall = (c->NP / nb) * nb;
    if (nfull > nall || nall <= 0) {
        gpt_set_error("block inverses: extent %lld exceeds the padded order %lld", (long long)nfull, (long long)c->NP);
        return GPT_E_ARG;
    }
    double *W, *Us = nullptr;
    GPT_TRY(ensure(c, SLOT_BINV, (size_t)nall * nb * sizeof(double), (void **)&W));
    GPT_TRY(ensure(c, SLOT_BINVU, (size_t)nall * nb * sizeof(double), (void **)&Us));
    *out = W;
    if (out_u) *out_u = Us;
    if (c->binv_valid) return GPT_OK;
    hipStream_t st = c->stream;
    if (c->binv_launches) GPT_TRY(build_block_inverses_512(c, st, nall / nb, Us, W));
    else GPT_TRY(launch_trinv512(st, nall / nb, c->dA, c->NP, c->d_invd, Us, W));      // one launch (solve.hip)
    c->binv_valid = true;
    return GPT_OK;
}

// The 2048 x 2048 diagonal blocks' inverses, for solves with at most GPT_FEW_ROWS rows against a large factor (the chain of
// dependent GEMMs -- two per block -- is what such a solve costs: 4 blocks at N = 8192 instead of 8 or 16).  Built from the
// 512-wide inverses by the GEMM-only triangular inverse of the gradient path (trtri_u_gemm: U = L^-T of the block, N^3/3 flop
// with the block's zeros skipped -- the identity pushed through the panel TRSM would cost 6 x that), then transposed.
#define GPT_BINV_NB3 2048
static int ensure_block_inverses_wide(gpt_ctx *c, int64_t nb, int64_t nfull, double **out)
{
    // (nb = 1024 -- solves with very few rows against factors of 4096 <= n < 8192 -- went through the substitution leaves block
    // by block until round 5: a dozen dependent launches per block)
    const bool w3 = (nb == GPT_BINV_NB3);
    double *W, *U, *T, *Wb;
    if (nfull > (c->NP / nb) * nb) {
        gpt_set_error("block inverses: extent %lld exceeds the padded order %lld", (long long)nfull, (long long)c->NP);
        return GPT_E_ARG;
    }
    nfull = (c->NP / nb) * nb;                                  // (always the whole padded order, see ensure_block_inverses)
    GPT_TRY(ensure(c, w3 ? SLOT_BINV3 : SLOT_BINV2, (size_t)nfull * nb * sizeof(double), (void **)&W));
    *out = W;
    bool &valid = w3 ? c->binv3_valid : c->binv2_valid;
    if (valid) return GPT_OK;
    GPT_TRY(ensure_block_inverses(c, GPT_BINV_NB, nfull, &Wb));
    // U strip (nfull x nb, block J at rows [J nb, J nb + nb)) followed by a strip of the same shape for the products
    GPT_TRY(ensure(c, SLOT_BINV3U, (size_t)2 * nfull * nb * sizeof(double), (void **)&U));
    T = U + nfull * nb;
    hipStream_t st = c->stream;
    GPT_HIP_CHECK(hipMemsetAsync(U, 0, (size_t)nfull * nb * sizeof(double), st));
    const WideBatch wb = {nfull / nb, nb * nb, nb * nb, nb * (c->NP + 1), nb * GPT_BINV_NB};
    GPT_TRY(trtri_u_gemm_b(c, st, 0, nb, Wb, U, nb, T, nb, wb));
    hipLaunchKernelGGL(transpose_rect_kernel, dim3((unsigned)(nb / 32), (unsigned)(nb / 32), (unsigned)wb.n), dim3(256), 0, st, U, nb, W, nb, nb,
                       nb, nb * nb, nb * nb);
    GPT_LAUNCH_CHECK();
    valid = true;
    return GPT_OK;
}

// Columns [lo, hi) of V <- the same columns of B L^-T (multiples of nb) by halving; a leaf is ONE out-of-place GEMM against
// the block inverse, V_j = B_j W_j^T, the update in between B[:, mid:hi] -= V[:, lo:mid] L[mid:hi, lo:mid]^T.  B is consumed.
static int trsm_rlt_binv(gpt_ctx *c, hipStream_t st, int64_t m, int64_t nb, int64_t lo, int64_t hi, const double *W, double *B,
                         int64_t ldb, double *V, int64_t ldv)
{
    const int64_t n = hi - lo;
    if (n == nb) return gemm_nt(c, st, m, nb, nb, 1.0, B + lo, ldb, W + lo * nb, nb, 0.0, V + lo, ldv, 0);
    const int64_t h = (n / (2 * nb)) * nb > 0 ? (n / (2 * nb)) * nb : nb;
    const int64_t mid = lo + h;
    GPT_TRY(trsm_rlt_binv(c, st, m, nb, lo, mid, W, B, ldb, V, ldv));
    GPT_TRY(gemm_nt(c, st, m, hi - mid, h, -1.0, V + lo, ldv, c->dA + mid * c->NP + lo, c->NP, 1.0, B + mid, ldb, 0));
    return trsm_rlt_binv(c, st, m, nb, mid, hi, W, B, ldb, V, ldv);
}

// V (m x n128, ldv) <- B L^-T for the resident factor (n128 = N rounded up to 128); B (m x n128, ldb) is consumed.
// C = beta * C + P_0 + P_1 + ... + P_{S-1} (each m x n, row stride n, `pstride` doubles apart), summed in that order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(int64_t m, int64_t n2, int S, const double2 *__restrict__ P, int64_t pstride2,
                                                            double beta, double *__restrict__ C, int64_t ldc)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m * n2) return;
    const int64_t r = i / n2, q = i - r * n2;
    double2 *cp = reinterpret_cast<double2 *>(C + r * ldc) + q;
    double2 acc = make_double2(0.0, 0.0);
    if (beta != 0.0) {
        acc = *cp;
        acc.x *= beta;
        acc.y *= beta;
    }
    // (S is a power of two >= 2: pairs, fours or eights of loads in flight, added in slab order)
    if (S % 8 == 0) {
        for (int s = 0; s < S; s += 8) {
            double2 p[8];
#pragma unroll
            for (int q = 0; q < 8; q++) p[q] = P[(int64_t)(s + q) * pstride2 + i];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                acc.x += p[q].x;
                acc.y += p[q].y;
            }
        }
    } else {
        for (int s = 0; s < S; s += 2) {
            const double2 p0 = P[(int64_t)s * pstride2 + i], p1 = P[(int64_t)(s + 1) * pstride2 + i];
            acc.x = (acc.x + p0.x) + p1.x;
            acc.y = (acc.y + p0.y) + p1.y;
        }
    }
    *cp = acc;
}

// A GEMM of a few-rows solve: m <= 256 rows against k in the thousands is a handful of 32x32 tiles (128 at m = 64, n = 2048: half the
// chip idle) each walking a long k loop at ~0.3 us per 16-wide k-tile -- 37 us for 0.5 GFLOP.  Split along k instead: S chunks as the
// batch dimension of ONE launch (chunk s reads columns [s k/S, (s+1) k/S) of A and B and writes alpha * A_s B_s^T to its own m x n slab),
// then one pass adds the slabs to C in chunk order -- a fixed summation order, so results repeat bit for bit run to run (they differ in
// rounding from the unsplit sum; option `splitk` 0 restores that).  Measured (scratch/r05_splitk_ab.py, N = 8192, predict with std):
// 16 / 64 / 128 points 0.352 / 0.352 / 0.467 -> 0.281 / 0.280 / 0.424 ms; the leaf GEMM 37 -> 22 us + 4.8 us for the sum.  A split
// launch is bound by the 32x32 kernel's throughput with four workgroups per CU (~25 TFLOP/s), no longer by one workgroup's k loop.
// Not for k = 512 (the blocks of solves with 129..256 rows): one launch is 11 us there against 7.6 + 4.8 split in two.
// `tri`: the caller needs the lower triangle of C only (the predictive covariance: k = N against at most 256 x 256 outputs -- a
// dozen tiles walking 512 k-tiles each, 150 us at N = 8192); a split launch computes all of C, an unsplit one the lower tiles.
static int gemm_nt_few(gpt_ctx *c, hipStream_t st, int64_t m, int64_t n, int64_t k, double alpha, const double *A, int64_t lda,
                       const double *B, int64_t ldb, double beta, double *C, int64_t ldc, int tri = 0)
{
    const int64_t tiles = ((m + 31) / 32) * ((n + 31) / 32);
    int64_t S = 1;
    if (c->splitk && !c->use_graph && c->tile == 0 && k >= 1024 && (n % 2) == 0 && (ldc % 2) == 0)
        while (S < 32 && tiles * S < c->splitk && (k / (2 * S)) >= 256 && (k % (2 * S * 16)) == 0) S *= 2;
    if (S == 1) return gemm_nt(c, st, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tri);
    double *P;
    const int64_t pstride = ((m + 31) / 32) * 32 * n;
    GPT_TRY(ensure(c, SLOT_SPLITK, (size_t)S * pstride * sizeof(double), (void **)&P));
    const int64_t kc = k / S;
    GPT_TRY(launch_gemm_nt(st, m, n, kc, alpha, A, lda, B, ldb, 0.0, P, n, 0, 0, 0, nullptr, nullptr, 0, EdgeSig(), EdgeSig(), 0, S, kc,
                           EdgeSig(), kc, pstride));
    const int64_t n2 = n / 2, tot = m * n2;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, m, n2, (int)S,
                       reinterpret_cast<const double2 *>(P), pstride / 2, beta, C, ldc);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

static int solve_rows_resident(gpt_ctx *c, hipStream_t st, int64_t m, int64_t n128, double *B, int64_t ldb, double *V, int64_t ldv)
{
    const int64_t nfull = (n128 / GPT_BINV_NB) * GPT_BINV_NB;
    if (nfull < 2 * GPT_BINV_NB) {
        GPT_TRY(launch_copy2d(st, m, n128, B, ldb, V, ldv));
        return trsm_rlt(c, st, m, n128, c->dA, c->NP, c->d_invd, V, ldv);
    }
    double *W;
    GPT_TRY(ensure_block_inverses(c, GPT_BINV_NB, nfull, &W));
    const int64_t rem = n128 - nfull;
    if (m <= 256) {
        // few rows: the halving recursion ends in updates with 64..256 rows and k of thousands -- a handful of workgroups
        // walking long k loops (1.6 ms at N = 8192).  Right-looking instead: after each leaf (one out-of-place GEMM against the
        // block inverse) one update of everything to its right.  What such a solve costs is the LENGTH of that chain of
        // dependent GEMMs, so up to GPT_FEW_ROWS rows the widest inverses that fit are used: 2048-wide blocks, then 1024-wide,
        // then 512-wide ones for what is left (predict with std at 64 points, N = 8192: 16 / 8 / 4 blocks -> 0.91 / 0.43 / ~0.35 ms).
        double *W3 = nullptr, *W2 = nullptr;
        int64_t n3 = 0, n2 = 0;
        if (m <= GPT_FEW_ROWS && n128 >= 4 * GPT_BINV_NB3) {
            n3 = (n128 / GPT_BINV_NB3) * GPT_BINV_NB3;
            GPT_TRY(ensure_block_inverses_wide(c, GPT_BINV_NB3, n3, &W3));
        } else if (m <= GPT_FEW_ROWS && n128 >= 4 * GPT_BINV_NB2) {
            n2 = (n128 / GPT_BINV_NB2) * GPT_BINV_NB2;
            GPT_TRY(ensure_block_inverses(c, GPT_BINV_NB2, n2, &W2));
        }
        int64_t j = 0;
        while (j < nfull) {
            const int64_t nb = (j + GPT_BINV_NB3 <= n3) ? GPT_BINV_NB3 : (j + GPT_BINV_NB2 <= n2) ? GPT_BINV_NB2 : GPT_BINV_NB;
            const double *Wj = (nb == GPT_BINV_NB3) ? W3 + j * nb : (nb == GPT_BINV_NB2) ? W2 + j * nb : W + j * nb;
            GPT_TRY(gemm_nt_few(c, st, m, nb, nb, 1.0, B + j, ldb, Wj, nb, 0.0, V + j, ldv));
            const int64_t r0 = j + nb;
            if (r0 < n128)
                GPT_TRY(gemm_nt_few(c, st, m, n128 - r0, nb, -1.0, V + j, ldv, c->dA + r0 * c->NP + j, c->NP, 1.0, B + r0, ldb));
            j = r0;
        }
    } else {
        GPT_TRY(trsm_rlt_binv(c, st, m, GPT_BINV_NB, 0, nfull, W, B, ldb, V, ldv));
        if (rem > 0) GPT_TRY(gemm_nt(c, st, m, rem, nfull, -1.0, V, ldv, c->dA + nfull * c->NP, c->NP, 1.0, B + nfull, ldb, 0));
    }
    if (rem > 0) {
        GPT_TRY(launch_copy2d(st, m, rem, B + nfull, ldb, V + nfull, ldv));
        GPT_TRY(trsm_rlt(c, st, m, rem, c->dA + nfull * c->NP + nfull, c->NP, c->d_invd + (nfull / 128) * GPT_WS_BLOCK, V + nfull, ldv));
    }
    return GPT_OK;
}

// ------------------------------------------------------------------------------------------------
// results to the host
// ------------------------------------------------------------------------------------------------
// Rows of a device matrix to host memory on the context's copy stream.  Pinned destinations (gpt_host_alloc, or anything
// hipHostRegister'ed / hipHostMalloc'ed by the caller) are written by the DMA engine directly and asynchronously; pageable
// ones go through a two-slot pinned ring with a host memcpy behind each slot (the runtime's own pageable path is a
// synchronous version of the same).  d2h_finish drains the ring.
#define GPT_STAGE_BYTES ((size_t)16 << 20)
struct StagePending { double *dst; int64_t ldd, rows, cols; bool on; };
static thread_local StagePending g_stage_pend[2] = {{nullptr, 0, 0, 0, false}, {nullptr, 0, 0, 0, false}};
static thread_local int g_stage_next = 0;

static bool host_ptr_is_pinned(const void *p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

static int stage_drain(gpt_ctx *c, int slot)
{
    StagePending &p = g_stage_pend[slot];
    if (!p.on) return GPT_OK;
    GPT_HIP_CHECK(hipEventSynchronize(c->cev[slot]));
    const double *src = c->h_stage + (size_t)slot * (GPT_STAGE_BYTES / sizeof(double));
    for (int64_t r = 0; r < p.rows; r++) memcpy(p.dst + r * p.ldd, src + r * p.cols, (size_t)p.cols * sizeof(double));
    p.on = false;
    return GPT_OK;
}

static int d2h_rows(gpt_ctx *c, double *dst, int64_t ldd, const double *dsrc, int64_t lds, int64_t rows, int64_t cols)
{
    if (host_ptr_is_pinned(dst)) {
        GPT_HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)ldd * sizeof(double), dsrc, (size_t)lds * sizeof(double),
                                       (size_t)cols * sizeof(double), (size_t)rows, hipMemcpyDeviceToHost, c->copy_stream));
        return GPT_OK;
    }
    if (!c->h_stage) GPT_HIP_CHECK(hipHostMalloc((void **)&c->h_stage, 2 * GPT_STAGE_BYTES, hipHostMallocDefault));
    const int64_t per = (int64_t)(GPT_STAGE_BYTES / sizeof(double)) / cols;      // rows per slot
    if (per < 1) {
        gpt_set_error("d2h_rows: a row of %lld doubles exceeds the staging slot", (long long)cols);
        return GPT_E_ARG;
    }
    for (int64_t r0 = 0; r0 < rows; r0 += per) {
        const int64_t nr = (rows - r0 < per) ? rows - r0 : per;
        const int slot = g_stage_next;
        g_stage_next ^= 1;
        GPT_TRY(stage_drain(c, slot));
        double *ring = c->h_stage + (size_t)slot * (GPT_STAGE_BYTES / sizeof(double));
        GPT_HIP_CHECK(hipMemcpy2DAsync(ring, (size_t)cols * sizeof(double), dsrc + r0 * lds, (size_t)lds * sizeof(double),
                                       (size_t)cols * sizeof(double), (size_t)nr, hipMemcpyDeviceToHost, c->copy_stream));
        GPT_HIP_CHECK(hipEventRecord(c->cev[slot], c->copy_stream));
        g_stage_pend[slot] = StagePending{dst + r0 * ldd, ldd, nr, cols, true};
    }
    return GPT_OK;
}

static int d2h_finish(gpt_ctx *c)
{
    GPT_TRY(stage_drain(c, g_stage_next));
    GPT_TRY(stage_drain(c, g_stage_next ^ 1));
    GPT_HIP_CHECK(hipStreamSynchronize(c->copy_stream));
    return GPT_OK;
}

// Pinned host memory for large results (the (M, M) predictive covariance): hipHostMalloc / hipHostFree.  gpt_predict writes
// into such a buffer with asynchronous DMA, overlapped with the computation; into pageable memory it stages.
extern "C" int gpt_host_alloc(int64_t bytes, void **out)
{
    if (!out || bytes <= 0) return GPT_E_ARG;
    *out = nullptr;
    GPT_HIP_CHECK(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
    return GPT_OK;
}

// Free / total device memory of the context's GPU (hipMemGetInfo): callers that size scratch by what is there (the batched
// evaluator of GaussianProcess.ll_batch) ask first instead of failing in hipMalloc.
extern "C" int gpt_mem_info(gpt_ctx *c, int64_t *free_bytes, int64_t *total_bytes)
{
    CTX_ENTER(c);
    size_t f = 0, t = 0;
    GPT_HIP_CHECK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return GPT_OK;
}

// Returns the scratch of the batched evaluator (gpt_fit_batch*: nbatch matrices) to the device: the slots are otherwise kept
// until the context is destroyed, which suits a grid walked in many chunks and nobody else.
extern "C" int gpt_release_batch_scratch(gpt_ctx *c)
{
    CTX_ENTER(c);
    GPT_HIP_CHECK(hipStreamSynchronize(c->stream));
    GPT_HIP_CHECK(hipStreamSynchronize(c->panel_stream));
    for (int slot : {SLOT_BATCH_A, SLOT_BATCH_WS, SLOT_BATCH_MISC}) {
        DevBuf &b = c->slots[slot];
        if (b.p) GPT_HIP_CHECK(hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    return GPT_OK;
}

extern "C" int gpt_host_free(void *p)
{
    if (p) GPT_HIP_CHECK(hipHostFree(p));
    return GPT_OK;
}

// ------------------------------------------------------------------------------------------------
// predict
// ------------------------------------------------------------------------------------------------
extern "C" int gpt_predict(gpt_ctx *c, const double *Xstar, const int32_t *nstar, int64_t M, int want,
                           const double *noise_params, const int32_t *noise_n, double *mean_out, double *std_out,
                           double *cov_out)
{
    CTX_ENTER(c);
    NEED_FACTOR(c);
    if (!c->have_kernel) {
        gpt_set_error("gpt_predict needs a factorisation produced by gpt_fit");
        return GPT_E_STATE;
    }
    if (M <= 0 || !Xstar || !nstar || !mean_out || want < 0 || want > 2 || (want == 1 && !std_out)) {
        gpt_set_error("gpt_predict: bad arguments");
        return GPT_E_ARG;
    }
    if (M > 65535 * 32) return GPT_E_ARG;
    const int D = c->D;
    const int64_t N = c->N, n128 = round_up(N, 128), MP = round_up(M, 64);
    const int64_t Nx = c->Nx;
    std::vector<KParams> all_factors(c->terms);
    bool any_product = false;
    for (const auto &t2 : c->terms2)
        if (t2.kernel_id >= 0) {
            all_factors.push_back(t2);
            any_product = true;
        }
    for (const auto &t : all_factors)
        if (t.kernel_id == GPT_KERNEL_M52) {
            GPT_TRY(check_m52_orders(nstar, M, D));
            break;
        }
    for (const auto &t : all_factors)
        if (t.kernel_id == GPT_KERNEL_RQ || t.kernel_id == GPT_KERNEL_MATERN || any_product) {
            long ms = 0;
            for (int64_t i = 0; i < M; i++) {
                long sn = 0;
                for (int d = 0; d < D; d++) sn += nstar[i * D + d];
                if (sn > ms) ms = sn;
            }
            if (ms + (ms > c->n_maxsum ? ms : c->n_maxsum) > GPT_RQ_MAXORD) {
                gpt_set_error("RationalQuadraticKernel: derivative orders of a pair sum to more than %d", GPT_RQ_MAXORD);
                return GPT_E_VALUE;
            }
            break;
        }
    hipStream_t st = c->stream;
    double *dXs, *dKst, *dmean;
    int32_t *dns;
    GPT_TRY(ensure(c, SLOT_XS, (size_t)M * D * sizeof(double), (void **)&dXs));
    GPT_TRY(ensure(c, SLOT_NS, (size_t)M * D * sizeof(int32_t), (void **)&dns));
    GPT_TRY(ensure(c, SLOT_KST, (size_t)MP * n128 * sizeof(double), (void **)&dKst));
    GPT_TRY(ensure(c, SLOT_VEC, (size_t)MP * 2 * sizeof(double), (void **)&dmean));
    double *dvar = dmean + MP;
    GPT_HIP_CHECK(hipMemcpyAsync(dXs, Xstar, (size_t)M * D * sizeof(double), hipMemcpyHostToDevice, st));
    GPT_HIP_CHECK(hipMemcpyAsync(dns, nstar, (size_t)M * D * sizeof(int32_t), hipMemcpyHostToDevice, st));
    // Kstar^T: row a = test point a, column i = training point i  (k is symmetric under swapping its
    // two (point, derivative-order) arguments, so this equals Kstar[i][a] of ref :966)
    GPT_TRY(launch_zero2d(st, MP, n128, dKst, n128));
    if (c->dT) {
        // with a transform the training side is T f(X): Kstar^T <- k(Xstar, X) T^T  (ref :966-970)
        double *dKx;
        GPT_TRY(ensure(c, SLOT_TK, (size_t)MP * c->NxP * sizeof(double), (void **)&dKx));
        GPT_TRY(launch_zero2d(st, MP, c->NxP, dKx, c->NxP));
        GPT_TRY(kbuild_terms(c, st, c->terms, 0, dXs, dns, M, c->dX, c->dn, Nx, 0, 0, 0, nullptr, 0.0, 0.0, dKx, c->NxP));
        GPT_TRY(gemm_nt(c, st, MP, round_up(N, 64), c->NxP, 1.0, dKx, c->NxP, c->dT, c->NxP, 0.0, dKst, n128, 0));
    } else
    GPT_TRY(kbuild_terms(c, st, c->terms, 0, dXs, dns, M, c->dX, c->dn, N, 0, 0, 0, nullptr, 0.0, 0.0, dKst, n128));
    GPT_TRY(ensure_alpha(c));
    GPT_TRY(launch_gemv_n(st, M, N, dKst, n128, c->d_alpha, dmean));
    GPT_HIP_CHECK(hipMemcpyAsync(mean_out, dmean, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, st));
    if (want >= 1) {
        double *dV;                                                               // V' = Kstar^T L^-T (Kstar^T is consumed)
        GPT_TRY(ensure(c, SLOT_BTMP, (size_t)MP * n128 * sizeof(double), (void **)&dV));
        GPT_TRY(solve_rows_resident(c, st, MP, n128, dKst, n128, dV, n128));
        KParams kn;
        if (noise_params) GPT_TRY(make_kparams(GPT_KERNEL_DIAGNOISE, noise_params, 1, D, -1, 1, noise_n, &kn));
        if (want == 1) {
            double *dkd;
            GPT_TRY(ensure(c, SLOT_VEC2, (size_t)M * sizeof(double), (void **)&dkd));
            for (size_t t = 0; t < c->terms.size(); t++) {
                KParams ks = c->terms[t];
                ks.symmetric = 1;
                ks.hyper_deriv = -1;
                GPT_TRY(launch_kpairs(st, ks, dXs, dXs, dns, dns, M, dkd, t > 0 ? 1 : 0,
                                      (t < c->terms2.size() && c->terms2[t].kernel_id >= 0) ? &c->terms2[t] : nullptr));
            }
            GPT_TRY(launch_rowsumsq_sub(st, M, n128, dV, n128, dkd, dvar));
            GPT_HIP_CHECK(hipMemcpyAsync(std_out, dvar, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, st));
            GPT_HIP_CHECK(hipStreamSynchronize(st));
            double nv = 0.0;
            for (int64_t a = 0; a < M; a++) {
                if (noise_params) {          // diagonal of the symmetric noise term (ref: noise.py:103-104)
                    bool hit = true;
                    for (int d = 0; d < D; d++) hit = hit && nstar[a * D + d] == kn.noise_n[d];
                    nv = hit ? noise_params[0] * noise_params[0] : 0.0;
                }
                std_out[a] = sqrt(std_out[a] + nv);
            }
            return GPT_OK;
        }
        // cov = K(Xstar, Xstar) - V' V'^T.  Only the lower triangle is computed (SYRK flops, not GEMM's), one 512-wide block
        // column at a time; behind each block column a mirror kernel completes ITS block row (the transposes of the rows
        // below it), and that block row starts for the host on a second stream while the next block column is computed: the
        // M^2 doubles of the result cross PCIe under the SYRK (M = 4096, N = 8192: 18.3 -> ~10 ms).
        // cov_out == NULL: the covariance stays on the device (lower triangle, SLOT_KSS) for gpt_cov_sample; nothing moves.
        double *dcov;
        // (row stride LDC = M rounded up to 128: gpt_cov_sample factors this matrix in place, padded to whole 128-blocks)
        const int64_t LDC = round_up(M, 128);
        GPT_TRY(ensure(c, SLOT_KSS, (size_t)LDC * LDC * sizeof(double), (void **)&dcov));
        GPT_TRY(launch_zero2d(st, LDC, LDC, dcov, LDC));
        GPT_TRY(kbuild_terms(c, st, c->terms, 1, dXs, dns, M, dXs, dns, M, 0, 0, 0, nullptr, 0.0, 0.0, dcov, LDC));
        if (noise_params) GPT_TRY(launch_add_noise_sym(st, kn, dXs, dns, M, dcov, LDC));
        c->cov_M = 0;
        const int64_t CB = 512;
        const int64_t nblk = (MP + CB - 1) / CB;
        if (cov_out && !c->copy_stream) {
            GPT_HIP_CHECK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
            for (auto &e : c->cev) GPT_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        for (int64_t q = 0; q < nblk; q++) {
            const int64_t c0 = q * CB, w = (MP - c0 < CB) ? MP - c0 : CB;
            if (MP <= 256)
                GPT_TRY(gemm_nt_few(c, st, MP, MP, n128, -1.0, dV, n128, dV, n128, 1.0, dcov, LDC, 1));
            else
                GPT_TRY(gemm_nt(c, st, MP - c0, w, n128, -1.0, dV + c0 * n128, n128, dV + c0 * n128, n128, 1.0, dcov + c0 * LDC + c0, LDC, 1));
            if (!cov_out) continue;
            GPT_TRY(launch_mirror_rows(st, dcov, LDC, c0, w, MP));
            hipEvent_t e = get_event(c, 100 + (size_t)q);
            if (!e) return GPT_E_HIP;
            GPT_HIP_CHECK(hipEventRecord(e, st));
        }
        if (!cov_out) {
            GPT_TRY(launch_diag_gather(st, dcov, LDC, M, dvar));
            if (std_out) GPT_HIP_CHECK(hipMemcpyAsync(std_out, dvar, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, st));
            GPT_HIP_CHECK(hipStreamSynchronize(st));
            if (std_out)
                for (int64_t a = 0; a < M; a++) std_out[a] = sqrt(std_out[a]);
            c->cov_M = M;
            return GPT_OK;
        }
        for (int64_t q = 0; q < nblk; q++) {
            const int64_t c0 = q * CB;
            if (c0 >= M) break;
            const int64_t rows = (M - c0 < CB) ? M - c0 : CB;
            GPT_HIP_CHECK(hipStreamWaitEvent(c->copy_stream, get_event(c, 100 + (size_t)q), 0));
            GPT_TRY(d2h_rows(c, cov_out + c0 * M, M, dcov + c0 * LDC, LDC, rows, M));
        }
        GPT_TRY(d2h_finish(c));
        GPT_HIP_CHECK(hipStreamSynchronize(st));
        if (std_out)
            for (int64_t a = 0; a < M; a++) std_out[a] = sqrt(cov_out[a * M + a]);
        return GPT_OK;
    }
    GPT_HIP_CHECK(hipStreamSynchronize(st));
    return GPT_OK;
}

// Posterior samples from the device-resident predictive covariance (ref: gaussian_process.py:1295-1300, :1330, draw_sample
// with rand_vars and method='cholesky'): after gpt_predict(want = 2, cov_out = NULL),
//   L = cholesky(cov + diag_add I) in place on the device,  out (M x S) = L rand (M x S)
// -- the M x M covariance never crosses PCIe (the caller adds the mean).  The resident factor of the fit is untouched (the
// factorisation here has its own workspace).  Status > 0: that leading minor of the loaded covariance is not positive definite.
extern "C" int gpt_cov_sample(gpt_ctx *c, int64_t M_rows, double diag_add, const double *rand, int64_t S, double *out)
{
    CTX_ENTER(c);
    if (c->cov_M <= 0) {
        gpt_set_error("gpt_cov_sample: call gpt_predict(want = 2, cov_out = NULL) first");
        return GPT_E_STATE;
    }
    if (!rand || !out || S <= 0) return GPT_E_ARG;
    if (M_rows != c->cov_M) {
        gpt_set_error("gpt_cov_sample: rand has %lld rows, the resident covariance %lld", (long long)M_rows, (long long)c->cov_M);
        return GPT_E_ARG;
    }
    const int64_t M = c->cov_M, LDC = round_up(M, 128), SP = round_up(S, 64);
    hipStream_t st = c->stream;
    double *dcov = (double *)c->slots[SLOT_KSS].p, *dzero, *ws, *dlow, *dRt, *dOut;
    GPT_TRY(ensure(c, SLOT_ZERO, (size_t)M * sizeof(double), (void **)&dzero));
    GPT_TRY(ensure(c, SLOT_BATCH_WS, ((size_t)(LDC / 128) * GPT_WS_BLOCK + 8) * sizeof(double), (void **)&ws));
    int32_t *dinfo = reinterpret_cast<int32_t *>(ws + (LDC / 128) * GPT_WS_BLOCK);
    GPT_TRY(ensure(c, SLOT_LOW, (size_t)LDC * LDC * sizeof(double), (void **)&dlow));
    GPT_TRY(ensure(c, SLOT_RHS, (size_t)SP * LDC * sizeof(double), (void **)&dRt));
    GPT_TRY(ensure(c, SLOT_OUT, (size_t)LDC * SP * sizeof(double), (void **)&dOut));
    c->cov_M = 0;                                                    // (the covariance is consumed)
    EvalScope scope(c, true);                                        // counted like an evaluation in flight, on event edges
    GPT_HIP_CHECK(hipMemsetAsync(dzero, 0, (size_t)M * sizeof(double), st));
    GPT_HIP_CHECK(hipMemsetAsync(dinfo, 0, sizeof(int32_t), st));
    GPT_TRY(launch_add_diag(st, dcov, LDC, M, dzero, diag_add));
    GPT_TRY(launch_fill_pad(st, dcov, LDC, M, LDC, nullptr, 0.0));      // unit diagonal on the padding rows
    // rand^T, zero padded: the GEMM wants the contraction index (rows of rand) contiguous
    std::vector<double> Rt((size_t)SP * LDC, 0.0);
    for (int64_t k = 0; k < M; k++)
        for (int64_t s_ = 0; s_ < S; s_++) Rt[(size_t)s_ * LDC + k] = rand[(size_t)k * S + s_];
    GPT_HIP_CHECK(hipMemcpyAsync(dRt, Rt.data(), Rt.size() * sizeof(double), hipMemcpyHostToDevice, st));
    GPT_TRY(potrf_run(c, LDC, dcov, LDC, ws, dinfo));
    GPT_TRY(launch_extract_lower(st, dcov, LDC, LDC, dlow, LDC));
    GPT_TRY(gemm_nt(c, st, LDC, SP, LDC, 1.0, dlow, LDC, dRt, LDC, 0.0, dOut, SP, 0));
    int32_t info = 0;
    GPT_HIP_CHECK(hipMemcpyAsync(&info, dinfo, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GPT_HIP_CHECK(hipMemcpy2DAsync(out, (size_t)S * sizeof(double), dOut, (size_t)SP * sizeof(double), (size_t)S * sizeof(double),
                                   (size_t)M, hipMemcpyDeviceToHost, st));
    GPT_HIP_CHECK(hipStreamSynchronize(st));
    if (info != 0) {
        gpt_set_error("%d-th leading minor of the array is not positive definite", (int)info);
        return (int)(info > M ? M : info);
    }
    return GPT_OK;
}

// ------------------------------------------------------------------------------------------------
// generic solves against the resident factor
// ------------------------------------------------------------------------------------------------
static int solve_common(gpt_ctx *c, double *B, int64_t nrhs, bool full)
{
    CTX_ENTER(c);
    NEED_FACTOR(c);
    if (!B || nrhs <= 0) return GPT_E_ARG;
    const int64_t N = c->N, n128 = round_up(N, 128), RP = round_up(nrhs, 64);
    hipStream_t st = c->stream;
    double *dBt;
    GPT_TRY(ensure(c, SLOT_RHS, (size_t)RP * n128 * sizeof(double), (void **)&dBt));
    std::vector<double> Bt((size_t)nrhs * N);
    for (int64_t i = 0; i < N; i++)
        for (int64_t r = 0; r < nrhs; r++) Bt[(size_t)r * N + i] = B[(size_t)i * nrhs + r];
    GPT_TRY(launch_zero2d(st, RP, n128, dBt, n128));
    GPT_HIP_CHECK(hipMemcpy2DAsync(dBt, (size_t)n128 * sizeof(double), Bt.data(), (size_t)N * sizeof(double),
                                   (size_t)N * sizeof(double), (size_t)nrhs, hipMemcpyHostToDevice, st));
    {
        double *dIn = dBt;                                                        // rows: (L^-1 b_r)^T, out of place
        GPT_TRY(ensure(c, SLOT_BTMP, (size_t)RP * n128 * sizeof(double), (void **)&dBt));
        GPT_TRY(solve_rows_resident(c, st, RP, n128, dIn, n128, dBt, n128));
    }
    if (full)
        for (int64_t r = 0; r < nrhs; r++) {
            // the padded tail of each row is (numerically) zero except a ~1e-150 entry in the augmented
            // column; clear it so the backward substitution sees an exact zero there
            if (n128 > N) GPT_HIP_CHECK(hipMemsetAsync(dBt + r * n128 + N, 0, (size_t)(n128 - N) * sizeof(double), st));
            GPT_TRY(launch_trsv_lt(st, n128, c->dA, c->NP, c->d_invd, dBt + r * n128));
        }
    GPT_HIP_CHECK(hipMemcpy2DAsync(Bt.data(), (size_t)N * sizeof(double), dBt, (size_t)n128 * sizeof(double),
                                   (size_t)N * sizeof(double), (size_t)nrhs, hipMemcpyDeviceToHost, st));
    GPT_HIP_CHECK(hipStreamSynchronize(st));
    for (int64_t i = 0; i < N; i++)
        for (int64_t r = 0; r < nrhs; r++) B[(size_t)i * nrhs + r] = Bt[(size_t)r * N + i];
    return GPT_OK;
}

extern "C" int gpt_solve_L(gpt_ctx *c, double *B, int64_t nrhs) { return solve_common(c, B, nrhs, false); }
extern "C" int gpt_cho_solve(gpt_ctx *c, double *B, int64_t nrhs) { return solve_common(c, B, nrhs, true); }

// ------------------------------------------------------------------------------------------------
// standalone dense entry points on host matrices
// ------------------------------------------------------------------------------------------------
extern "C" int gpt_potrf_host(gpt_ctx *c, double *A, int64_t N)
{
    CTX_ENTER(c);
    if (!A || N <= 0) return GPT_E_ARG;
    std::vector<double> y((size_t)N, 0.0);
    double ll, ld;
    int rc = gpt_fit_matrix(c, A, N, y.data(), &ll, &ld);
    if (rc != GPT_OK) return rc;
    return gpt_get_L(c, A);
}

extern "C" int gpt_gemm_nt_host(gpt_ctx *c, int64_t m, int64_t n, int64_t k, double alpha, const double *A,
                                const double *B, double beta, double *C)
{
    CTX_ENTER(c);
    if (m <= 0 || n <= 0 || k <= 0 || !A || !B || !C) return GPT_E_ARG;
    const int64_t mp = round_up(m, 64), np = round_up(n, 64), kp = round_up(k, 64);
    double *dA, *dB, *dC;
    GPT_TRY(ensure(c, SLOT_KST, (size_t)mp * kp * sizeof(double), (void **)&dA));
    GPT_TRY(ensure(c, SLOT_KSS, (size_t)np * kp * sizeof(double), (void **)&dB));
    GPT_TRY(ensure(c, SLOT_RHS, (size_t)mp * np * sizeof(double), (void **)&dC));
    hipStream_t st = c->stream;
    GPT_TRY(launch_zero2d(st, mp, kp, dA, kp));
    GPT_TRY(launch_zero2d(st, np, kp, dB, kp));
    GPT_TRY(launch_zero2d(st, mp, np, dC, np));
    GPT_HIP_CHECK(hipMemcpy2DAsync(dA, kp * sizeof(double), A, k * sizeof(double), k * sizeof(double), m, hipMemcpyHostToDevice, st));
    GPT_HIP_CHECK(hipMemcpy2DAsync(dB, kp * sizeof(double), B, k * sizeof(double), k * sizeof(double), n, hipMemcpyHostToDevice, st));
    GPT_HIP_CHECK(hipMemcpy2DAsync(dC, np * sizeof(double), C, n * sizeof(double), n * sizeof(double), m, hipMemcpyHostToDevice, st));
    GPT_TRY(gemm_nt(c, st, mp, np, kp, alpha, dA, kp, dB, kp, beta, dC, np, 0));
    GPT_HIP_CHECK(hipMemcpy2DAsync(C, n * sizeof(double), dC, np * sizeof(double), n * sizeof(double), m, hipMemcpyDeviceToHost, st));
    GPT_HIP_CHECK(hipStreamSynchronize(st));
    return GPT_OK;
}

// ------------------------------------------------------------------------------------------------
// device API
// ------------------------------------------------------------------------------------------------
extern "C" int gpt_dev_kbuild(gpt_ctx *c, int kernel_id, const double *params_host, int nparams, const double *dXi,
                              const int32_t *dni, int64_t M, const double *dXj, const int32_t *dnj, int64_t P, int D,
                              int hyper_deriv, int symmetric, const int32_t *noise_n_host, int lower_only, int64_t i0,
                              int64_t j0, const double *d_err_y, double noise_var, double diag_add, double *dK,
                              int64_t ldk)
{
    CTX_ENTER(c);
    KParams kp;
    GPT_TRY(make_kparams(kernel_id, params_host, nparams, D, hyper_deriv, symmetric, noise_n_host, &kp));
    return launch_kbuild(c->stream, kp, dXi, dni, M, dXj, dnj, P, lower_only, i0, j0, d_err_y, noise_var, diag_add,
                         dK, ldk);
}

extern "C" int gpt_dev_gemm_nt(gpt_ctx *c, int64_t m, int64_t n, int64_t k, double alpha, const double *dA,
                               int64_t lda, const double *dB, int64_t ldb, double beta, double *dC, int64_t ldc,
                               int tri)
{
    CTX_ENTER(c);
    return gemm_nt(c, c->stream, m, n, k, alpha, dA, lda, dB, ldb, beta, dC, ldc, tri);
}

extern "C" int gpt_dev_gemm_nt_stair(gpt_ctx *c, int64_t m, int64_t nseg, int64_t seg_cols, int64_t k, double alpha,
                                     const double *dA, int64_t lda, const double *dB, int64_t ldb, int64_t b_stride,
                                     int64_t row_step, double beta, double *dC, int64_t ldc)
{
    CTX_ENTER(c);
    // profiled like the single-GPU updates (bench.py's roofline object of the partitioned line): algorithmic flops =
    // 2k per element of the staircase (segment q: lower trapezoid of (m - q row_step) x seg_cols)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->prof_gemm && !c->use_graph) {
        double elems = 0.0;
        for (int64_t q = 0; q < nseg; q++) {
            const double mq = (double)(m - q * row_step), w = (double)seg_cols;
            if (mq > 0) elems += 0.5 * w * (w + 1.0) + (mq - w > 0 ? (mq - w) * w : 0.0);
        }
        const double flops = 2.0 * (double)k * elems;
        if (flops >= 1.0e9) {
            if (c->gprof_used == c->gprof.size()) {
                gpt_ctx::GemmProf g;
                GPT_HIP_CHECK(hipEventCreate(&g.e0));
                GPT_HIP_CHECK(hipEventCreate(&g.e1));
                g.flops = 0;
                g.stop = g.e1;
                c->gprof.push_back(g);
            }
            gpt_ctx::GemmProf *gp = &c->gprof[c->gprof_used++];
            gp->flops = flops;
            gp->stop = gp->e1;
            e0 = gp->e0;
            e1 = gp->e1;
        }
    }
    return launch_gemm_nt_stair(c->stream, m, nseg, seg_cols, k, alpha, dA, lda, dB, ldb, b_stride, row_step, beta, dC,
                                ldc, 0, e0, e1);
}

// Trailing update of one rank of the 2-D block-cyclic engine in one launch (gemm.hip launch_gemm_nt_gridstair; profiled like the
// staircase above).
extern "C" int gpt_dev_gemm_nt_gridstair(gpt_ctx *c, int64_t m, int64_t nseg, int64_t seg_cols, int64_t k, double alpha,
                                         const double *dA, int64_t lda, const double *dB, int64_t ldb, int64_t off, int64_t num,
                                         int64_t den, int64_t base, double beta, double *dC, int64_t ldc)
{
    CTX_ENTER(c);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->prof_gemm && !c->use_graph && den > 0 && seg_cols > 0) {
        double elems = 0.0;
        for (int64_t q = 0; q < nseg; q++) {
            const int64_t v = off + q * num, rs = (v + den - 1) / den - base;
            const double mq = (double)(m - rs * seg_cols), w = (double)seg_cols;
            if (mq <= 0) continue;
            elems += (v % den == 0) ? 0.5 * w * (w + 1.0) + (mq - w > 0 ? (mq - w) * w : 0.0) : mq * w;
        }
        const double flops = 2.0 * (double)k * elems;
        if (flops >= 1.0e9) {
            if (c->gprof_used == c->gprof.size()) {
                gpt_ctx::GemmProf g;
                GPT_HIP_CHECK(hipEventCreate(&g.e0));
                GPT_HIP_CHECK(hipEventCreate(&g.e1));
                g.flops = 0;
                g.stop = g.e1;
                c->gprof.push_back(g);
            }
            gpt_ctx::GemmProf *gp = &c->gprof[c->gprof_used++];
            gp->flops = flops;
            gp->stop = gp->e1;
            e0 = gp->e0;
            e1 = gp->e1;
        }
    }
    const int lds_pad = c->lookahead ? c->gemm_pad : 0;
    return launch_gemm_nt_gridstair(c->stream, m, nseg, seg_cols, k, alpha, dA, lda, dB, ldb, off, num, den, base, beta, dC, ldc,
                                    lds_pad, e0, e1);
}

extern "C" int gpt_dev_row_sumsq(gpt_ctx *c, const double *d_row, int64_t w, double *d_acc)
{
    CTX_ENTER(c);
    return launch_row_sumsq(c->stream, d_row, w, d_acc);
}

extern "C" int gpt_dev_potrf_panel(gpt_ctx *c, int64_t m, int64_t nb, double *dA, int64_t lda, double *d_invd,
                                   int32_t *d_info, int64_t info_base)
{
    CTX_ENTER(c);
    if (nb <= 0 || nb % 128 || m < nb || m % 16) {
        gpt_set_error("potrf_panel: need nb a multiple of 128 and m >= nb, m a multiple of 16");
        return GPT_E_ARG;
    }
    return panel_rec(c, c->stream, dA, lda, m, nb, d_invd, d_info, info_base);
}

extern "C" int gpt_dev_potrf(gpt_ctx *c, int64_t n, double *dA, int64_t lda, double *d_invd, int32_t *d_info)
{
    CTX_ENTER(c);
    return potrf_run(c, n, dA, lda, d_invd, d_info);
}

extern "C" int gpt_dev_trsm_rlt(gpt_ctx *c, int64_t m, int64_t n, const double *dL, int64_t ldl, const double *d_invd,
                                double *dB, int64_t ldb)
{
    CTX_ENTER(c);
    if (n <= 0 || n % 128 || m % 16) {
        gpt_set_error("trsm_rlt: n must be a multiple of 128 and m a multiple of 16");
        return GPT_E_ARG;
    }
    return trsm_rlt(c, c->stream, m, n, dL, ldl, d_invd, dB, ldb);
}

extern "C" int gpt_dev_copy2d(gpt_ctx *c, int64_t rows, int64_t cols, const double *d_src, int64_t lds, double *d_dst,
                              int64_t ldd)
{
    CTX_ENTER(c);
    return launch_copy2d(c->stream, rows, cols, d_src, lds, d_dst, ldd);
}

extern "C" int gpt_dev_copy2d_on(gpt_ctx *c, void *stream, int64_t rows, int64_t cols, const double *d_src, int64_t lds, double *d_dst,
                                 int64_t ldd)
{
    CTX_ENTER(c);
    return launch_copy2d(stream ? (hipStream_t)stream : c->stream, rows, cols, d_src, lds, d_dst, ldd);
}

extern "C" int gpt_dev_pad_block(gpt_ctx *c, double *dA, int64_t lda, int64_t c0, int64_t nb, int64_t n_valid,
                                 int64_t n_pad, const double *d_y, double big)
{
    CTX_ENTER(c);
    return launch_pad_block(c->stream, dA, lda, c0, nb, n_valid, n_pad, d_y, big);
}

extern "C" int gpt_dev_panel_scalars(gpt_ctx *c, const double *dP, int64_t ldp, int64_t w, int64_t zrow, double *d_acc)
{
    CTX_ENTER(c);
    return launch_panel_scalars(c->stream, dP, ldp, w, zrow, d_acc);
}
