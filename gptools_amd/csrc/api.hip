// api.hip -- C ABI (include/gpt_hip.h) and host-side orchestration for libgpt_hip.so.
// One translation unit, kept in parts by entry-point family (api_*.inc, included at the end of this file in dependency order):
// this file holds the error channel and struct gpt_ctx.
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
#include <chrono>
#include <mutex>
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
    int64_t tail_wait = 1;                 // 1: the main stream's last launch of a panel awaits the NEXT panel's flag at its end (gemm.hip "tail wait"): no
                                           //    wait kernel between two trailing updates.  Round 4 measured it the same as 0 (4.459 / 4.471 ms at N = 8192); at
                                           //    round 6's schedule: C3 -2 ... -47 us in six same-process A/Bs (mean 19), C2 -11 ... -18 us -- on.  The gain sits in
                                           //    the launches that DO wait (1-6 GFLOP: only-large or only-small launches gain nothing); not while launches are timed.
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
    unsigned *d_flag = nullptr;        // progress word of potf2_trsm_kernel (only ever raised)
    unsigned flag_epoch = 0;
    int64_t binv_launches = 0;             // 1: the 512-wide block inverses by the recursion over 15 launches of rounds 2-4 instead of trinv512_kernel
    int64_t splitk = 512;                  // few-rows solves: GEMMs with k >= 1024 of fewer 32x32 tiles than this are split along k until they reach it (0: never)
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
    int64_t dev_gemm_pad = 0;          // dummy LDS (bytes) per workgroup of the device API's GEMM launches on this context (gpt_dev_gemm_nt*):
                                       // a residency cap for the partitioned engines' trailing updates, see "dev_gemm_pad" in gpt_hip.h
    int64_t eager_alpha = 0;           // 1: every evaluation also enqueues alpha = L^-T z behind the factorisation (no second host round trip)
    int64_t binv_early = 0;            // this factorisation: the 512-wide inverses of the diagonal blocks [0, binv_early) were enqueued on the
                                       // main stream under the last panel (enqueue_early_block_inverses); e_binv_early follows them there
    hipEvent_t e_binv_early = nullptr;
    bool want_early_binv = false;      // set by factor_and_ll around potrf_run for an eager evaluation
    // Eager alpha with N a multiple of 512: the factor's last 128 columns hold only the augmented row and the padding.  alpha needs
    // nothing of that leaf, so its pivot block (and the rank-128 updates that reach it) leave the panel stream: the substitution
    // follows the last REAL leaf at once, the pad leaf and the reduction run beside it on the main stream (potrf_enqueue).
    int64_t defer_pad = 1;             // option "defer_pad": 0 keeps the pad leaf on the panel stream (A/B)
    bool defer_pad_leaf = false;       // in: set by factor_and_ll around potrf_run
    bool pad_leaf_deferred = false;    // out: the pad leaf went to the main stream (the reduction must follow it there)
    int pad_upd_n = 0;                 // leaves whose update of the pad block was held back, by first column
    int64_t pad_upd_lc[8] = {0};
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
    struct GemmProf { hipEvent_t e0, e1, stop; double flops, bytes; };
    std::vector<GemmProf> gprof;
    size_t gprof_used = 0;
    double prof_flops = 0, prof_ms = 0, prof_count = 0, prof_bytes = 0;
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

#include "api_kparams.inc"      // kernel-parameter marshalling and argument checks shared by the entry points (make_kparams, order limits)
#include "api_schedule.inc"      // the factorisation: GEMM driver, flag-edge policy (EvalScope), panels, the look-ahead schedule potrf_enqueue / potrf_run
#include "api_context.inc"      // gpt_ctx_create / destroy / set_option / synchronize, workspace slots
#include "api_kernels.inc"      // Kernel.__call__ and compute_Kij: gpt_kpairs, gpt_kbuild, gpt_kpairs2, gpt_kbuild2
#include "api_fit.inc"      // data residency and the fit: gpt_set_data, gpt_set_T, gpt_fit, gpt_fit_sum, gpt_fit_terms, gpt_fit_matrix
#include "api_batch.inc"      // batched small fits (gpt_fit_batch*), timings / GEMM profile read-out, gpt_get_L, alpha (gpt_get_alpha)
#include "api_grad.inc"      // triangular inverses (gpt_dev_trinv) and the analytic LML gradient gpt_ll_grad
#include "api_solve.inc"      // many-right-hand-side solves against the resident factor: block inverses, right-TRSM by GEMMs, split-k; pinned host buffers
#include "api_predict.inc"      // gpt_predict, gpt_cov_sample, gpt_solve_L / gpt_cho_solve, gpt_potrf_host / gpt_gemm_nt_host
#include "api_dev.inc"      // device API (gpt_dev_*): raw device pointers on the context's streams, for the one-process-per-GPU engines
#include "api_plan.inc"      // compiled schedules of the partitioned engines: gpt_plan_* (op list replay, RCCL called directly)
