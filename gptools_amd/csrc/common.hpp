// common.hpp -- shared declarations for libgpt_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/gpt_hip.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

#define GPT_WAVE 64
// Per 128-column diagonal block the factorisation leaves a packed workspace ("ws") of GPT_WS_BLOCK doubles:
//   [0, 2048)     inverses of the eight 16x16 diagonal blocks of L_kk,
//   [2048, 9216)  the 28 strictly-lower 16x16 blocks of L_kk (block (j, c), c < j, at index j(j-1)/2 + c),
// every 16x16 block stored in MFMA B-operand lane order: element (r, c) at [c / 4][r + 16 * (c % 4)], so that a
// fragment is 512 contiguous bytes for the panel TRSM.
#define GPT_WS_BLOCK 9216
#define GPT_WS_LOFF 2048

void gpt_set_error(const char *fmt, ...);

#define GPT_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            gpt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (e_ == hipErrorOutOfMemory) ? GPT_E_NOMEM : GPT_E_HIP;                \
        }                                                                                \
    } while (0)

#define GPT_LAUNCH_CHECK()                                                               \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess) {                                                          \
            gpt_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
            return GPT_E_HIP;                                                            \
        }                                                                                \
    } while (0)

// Kernel hyperparameters as passed by value to the device kernels.
struct KParams {
    int kernel_id;
    int D;
    int hyper_deriv;      // -1 = None
    int symmetric;        // DiagonalNoiseKernel only fires for symmetric calls
    double sigma;         // params[0]
    double alpha;         // RationalQuadraticKernel: params[1]
    double l[GPT_MAX_DIM];      // length scales (SE / M52)
    double inv_l[GPT_MAX_DIM];  // 1 / l
    double inv_var[GPT_MAX_DIM];// 1 / l^2
    int noise_n[GPT_MAX_DIM];   // DiagonalNoiseKernel.n
    // MaternKernel (general nu): nu = alpha; constants of make_kparams (api.hip)
    double m_cnu;             // 2^(1-nu) / Gamma(nu)
    double m_mu;              // nu - round(nu), |mu| <= 1/2
    int m_nint;               // round(nu)
    int m_isint;              // nu is an integer (the reference then averages nu -+ 0.001 near the origin)
    int zero_l;               // some length scale is exactly 0 (1 / l = inf): the squared-exponential pair function then applies the
    int pad_;                 //   reference's 0/0 -> 0 rule per dimension (core.py:416); otherwise it is one multiply
    double m_gampl, m_gammi, m_gam1, m_gam2;      // Temme's 1/Gamma(1 +- mu) and their combinations
    double m_g[2], m_gm[2], m_nus[2];             // Gamma(nu_s), Gamma(-nu_s), nu_s for the small-y series (nu_s = nu, or nu -+ 0.001)
};

// A cross-stream edge without an event: the kernel that completes a piece of work raises a 32-bit word in device
// memory when its LAST workgroup is through (its results written with write-through stores and drained first) and the
// consumer polls that word -- inside its first kernel, or in a one-wave wait kernel in front of it (launch_wait_flag).
// Measured (scratch/waitvalue_probe.hip): producer end -> consumer start 1.5 us and nothing added to the producer's own
// stream, against 7.6-9.1 us and 1.7-4.5 us for a stop event on the dispatch packet + hipStreamWaitEvent.
// word[0] = the flag (only ever raised), word[1] = workgroup counter.
//
// Memory ordering.  Producer: the payload is written with agent-scope (write-through, sc1) stores, every wave drains its
// own (s_waitcnt vmcnt(0)) before the barrier, then ONE thread raises the word -- no release fence: on gfx950 that is
// buffer_wbl2, a write-back of every dirty line of the XCD's L2 (1.7-6 us per hand-over, measured) for data that is
// already in memory.  Consumer: whatever it reads of the payload BEHIND the flag inside the same kernel must not come from
// a line cached before the producer wrote it.  Two forms: (a) the payload is read with agent-scope (sc1, cache-bypassing)
// loads -- the GEMM waiter's C tile, the TRSM consumers' packed blocks -- and nothing else is needed; (b) the payload is read
// with plain loads (the first diagonal-block kernel of a factorisation: its 128 x 128 block and rows) and the polling
// thread issues an agent-scope ACQUIRE fence (buffer_inv sc1) after it has seen the value and before the workgroup
// barrier (edge_poll<.., true>).  The fence is NOT free for the rest of the chip -- it drops the non-coherent lines of the
// whole XCD's L2, i.e. the operand panels of a trailing update running there: issued by every workgroup of the panel
// stream's waiting GEMM launches it cost 10 % of an evaluation at N = 8192 (4.96 against 4.50 ms, same-box A/B) -- so it is
// used only in form (b), once per evaluation and workgroup.  A stream-side wait (wait_flag_kernel) needs neither: the
// kernel behind it starts with the dispatch packet's own acquire.
//
// Every wait is BOUNDED: a kernel that waits for another kernel never ends if the other one cannot run -- a tool that runs
// one kernel at a time, queues oversubscribed by other processes on the GPU, a lost launch.  After GPT_EDGE_TIMEOUT_TICKS of
// the 100 MHz wall clock the waiter raises `*err`, stops waiting and carries on (on data that may be incomplete); the host
// finds the error word at the end of the evaluation, switches the process to event edges for good and runs the
// evaluation again (api.hip: EvalScope / fit_terms).
struct EdgeSig {
    unsigned *word = nullptr;
    unsigned value = 0;
    unsigned *err = nullptr;      // waits only: raised when the wait timed out
};
#define GPT_EDGE_TIMEOUT_TICKS 25000000ll      // 250 ms
// device side: called by every workgroup that takes part, after its results are stored (all threads of the workgroup)
#ifdef __HIPCC__
__device__ __forceinline__ void edge_signal(unsigned *word, unsigned value, unsigned total)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's (write-through) stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(word + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (done == total) {
            __hip_atomic_store(word + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// Consumer side, ONE thread of the workgroup (the caller follows with __syncthreads()): poll until *word >= value (signed
// difference: the words only ever go up), SLEEP = s_sleep argument between polls; bounded; ACQUIRE: see above.
template <int SLEEP, bool ACQUIRE>
__device__ __forceinline__ void edge_poll(const unsigned *word, unsigned value, unsigned *err)
{
    if ((int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) {
        const long long t0 = wall_clock64();
        while ((int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) {
            __builtin_amdgcn_s_sleep(SLEEP);
            if (err && wall_clock64() - t0 > GPT_EDGE_TIMEOUT_TICKS) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    if (ACQUIRE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
#endif

// ---- launchers implemented in the .hip files (all asynchronous on `st`) -------------------
int launch_kpairs(hipStream_t st, const KParams &kp, const double *dXi, const double *dXj,
                  const int32_t *dni, const int32_t *dnj, int64_t M, double *dout, int accumulate = 0,
                  const KParams *kp2 = nullptr);      // kp2 (kernel_id >= 0): the pair list of the product kp * kp2
int launch_kbuild(hipStream_t st, const KParams &kp, const double *dXi, const int32_t *dni, int64_t M,
                  const double *dXj, const int32_t *dnj, int64_t P, int lower_only, int64_t i0, int64_t j0,
                  const double *d_err_y, double noise_var, double diag_add, double *dK, int64_t ldk,
                  int accumulate = 0, const KParams *kp2 = nullptr);
int launch_kbuild_prod(hipStream_t st, const KParams &kp1, const KParams &kp2, const double *dXi, const int32_t *dni, int64_t M,
                       const double *dXj, const int32_t *dnj, int64_t P, int lower_only, int64_t i0, int64_t j0,
                       const double *d_err_y, double noise_var, double diag_add, double *dK, int64_t ldk, int accumulate);
int launch_kpairs_prod(hipStream_t st, const KParams &kp1, const KParams &kp2, const double *dXi, const double *dXj,
                       const int32_t *dni, const int32_t *dnj, int64_t M, double *dout, int accumulate);
int launch_check_orders(hipStream_t st, const int32_t *dn, int64_t M, int D, int32_t *d_flag);
int launch_gemm_nt(hipStream_t st, int64_t m, int64_t n, int64_t k, double alpha, const double *A, int64_t lda,
                   const double *B, int64_t ldb, double beta, double *C, int64_t ldc, int tri, int force_tile, int lds_pad,
                   hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, int prio = 0, EdgeSig edge = EdgeSig(),
                   EdgeSig wait = EdgeSig(), int64_t edge_cols = 0, int64_t nbatch = 1, int64_t bstride = 0,
                   EdgeSig tail = EdgeSig(), int64_t bstride_b = -1, int64_t bstride_c = -1);      // batch strides of B / C (< 0: bstride)
int launch_potf2_diag(hipStream_t st, double *A, int64_t lda, double *invd, int32_t *info, int64_t info_base,
                      EdgeSig wait = EdgeSig(), int64_t nbatch = 1, int64_t bstride_a = 0, int64_t bstride_ws = 0);
int launch_gemm_nt_stair(hipStream_t st, int64_t m, int64_t nseg, int64_t seg_cols, int64_t k, double alpha,
                         const double *A, int64_t lda, const double *B, int64_t ldb, int64_t b_stride, int64_t row_step,
                         double beta, double *C, int64_t ldc, int lds_pad, hipEvent_t ev0 = nullptr,
                         hipEvent_t ev1 = nullptr);
int launch_gemm_nt_gridstair(hipStream_t st, int64_t m, int64_t nseg, int64_t seg_cols, int64_t k, double alpha, const double *A,
                             int64_t lda, const double *B, int64_t ldb, int64_t off, int64_t num, int64_t den, int64_t base,
                             double beta, double *C, int64_t ldc, int lds_pad, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
int launch_potf2_trsm(hipStream_t st, double *A, int64_t lda, double *invd, int32_t *info, int64_t info_base, int64_t m,
                      unsigned *flag, unsigned flag_base, hipEvent_t done = nullptr, EdgeSig edge = EdgeSig(),
                      EdgeSig wait = EdgeSig(), int rows64 = 0);
int launch_trsm_panel(hipStream_t st, int64_t m, const double *L, int64_t ldl, const double *invd,
                      double *B, int64_t ldb, hipEvent_t done = nullptr, EdgeSig edge = EdgeSig(), int64_t nbatch = 1,
                      int64_t bstride_a = 0, int64_t bstride_ws = 0);
// batched small fits (gpt_fit_batch)
int launch_kbuild_batch(hipStream_t st, int kernel_id, int D, const KParams *d_kps, const double *d_noise_var, int64_t nbatch,
                        const double *dX, const int32_t *dn, int64_t N, const double *d_err_y, double diag_add, double *dK,
                        int64_t ldk, int64_t bstride, int accumulate = 0, int full = 0, const KParams *d_kps2 = nullptr);
int launch_batch_pad(hipStream_t st, const double *h_y, int64_t nbatch, double *A, int64_t lda, int64_t bstride, int64_t n_valid,
                     int64_t n_pad, double big, int32_t *info);
int launch_batch_logdet_dot(hipStream_t st, const double *A, int64_t lda, int64_t bstride, int64_t n, int64_t nbatch,
                            const int32_t *d_info, double *out3);
#define GPT_GRAD_MAXH 8
int grad_reduce_blocks(int64_t N);
int launch_grad_reduce(hipStream_t st, const KParams &kp, int nh, const int *hl, const double *dX, const int32_t *dn,
                       int64_t N, const double *dalpha, const double *dW, int64_t ldw, double *dpartial);
int launch_add_diag(hipStream_t st, double *A, int64_t lda, int64_t n, const double *err, double diag_add, int64_t nbatch = 1,
                    int64_t bstride = 0);
// Test aid (GPT_JITTER=<max microseconds> in the environment): a delay kernel of random length on `st`, called in front of
// every dense launch.  The schedules express every dependency as an event, so results must not move with the relative
// timing of the streams; a missing edge shows up as a wrong number (tests/test_gpu_parity.py).  Off: one branch.
void gpt_jitter(hipStream_t st);
int launch_set_flag(hipStream_t st, unsigned *word, unsigned value);
// One-wave kernel on `st` that waits (bounded, see EdgeSig) until *w.word >= w.value: the stream-side end of a flag edge.
int launch_wait_flag(hipStream_t st, EdgeSig w);
int gemm_small_threshold();      // 64x64-tile count under which launch_gemm_nt cuts a launch into 32x32 tiles (gemm.hip)
int launch_upload_pad(hipStream_t st, const double *h_src, double *d_dst, int64_t ncopy, int32_t *info, double *A,
                      int64_t lda, int64_t n_valid, int64_t n_pad, double big);
int launch_fill_pad(hipStream_t st, double *A, int64_t lda, int64_t n_valid, int64_t n_pad, const double *dy,
                    double big);
int launch_logdet_dot(hipStream_t st, const double *A, int64_t lda, int64_t n, const int32_t *d_info, double *d_part,
                      double *out4, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, unsigned *edge_err = nullptr);
int launch_extract_lower(hipStream_t st, const double *A, int64_t lda, int64_t n, double *out, int64_t ldo);
int launch_trsv_lt(hipStream_t st, int64_t n, const double *L, int64_t ldl, const double *invd, double *x, int64_t b_lo = 0);
int launch_trsv_lt_wide(hipStream_t st, int64_t nwide, const double *L, int64_t ldl, const double *U, double *w, double *x,
                        const double *w0 = nullptr);
int launch_alpha_init(hipStream_t st, int64_t n, int64_t np, const double *z, double *w, double *x);
int launch_gemv_n(hipStream_t st, int64_t m, int64_t n, const double *A, int64_t lda, const double *x, double *y);
int launch_trinv512(hipStream_t st, int64_t nblk, const double *L, int64_t ldl, const double *invd, double *U, double *W);
int launch_rowsumsq_sub(hipStream_t st, int64_t m, int64_t n, const double *V, int64_t ldv, const double *kdiag,
                        double *var_out);
int launch_add_noise_sym(hipStream_t st, const KParams &kp, const double *dX, const int32_t *dn, int64_t M,
                         double *C, int64_t ldc);
int launch_mirror_rows(hipStream_t st, double *A, int64_t lda, int64_t c0, int64_t w, int64_t n);
int launch_alpha_trace(hipStream_t st, const double *alpha, const double *W, int64_t ldw, int64_t n, double *out);
int launch_diag_gather(hipStream_t st, const double *A, int64_t lda, int64_t n, double *out);
int launch_copy2d(hipStream_t st, int64_t rows, int64_t cols, const double *src, int64_t lds, double *dst, int64_t ldd);
int launch_zero2d(hipStream_t st, int64_t rows, int64_t cols, double *dst, int64_t ldd);
int launch_pad_block(hipStream_t st, double *A, int64_t lda, int64_t c0, int64_t nb, int64_t n_valid, int64_t n_pad,
                     const double *dy, double big);
int launch_row_sumsq(hipStream_t st, const double *row, int64_t w, double *acc);
int launch_panel_scalars(hipStream_t st, const double *P, int64_t ldp, int64_t w, int64_t zrow, double *acc);
