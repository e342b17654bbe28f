// gemm.hip -- fp64 MFMA "NT" GEMM / SYRK for gfx950:  C = beta*C + alpha * A * B^T,
// A (m x k), B (n x k), C (m x n), all row-major with the contraction index contiguous.
//
// This is the roofline kernel of the Cholesky (trailing SYRK/GEMM updates of LAPACK dpotrf, which
// the reference reaches through scipy.linalg.cholesky, ref: gptools/gaussian_process.py:1452) and
// of predict's cov = K** - v^T v (ref: gptools/gaussian_process.py:987).
//
// Design (CDNA4):
//   * v_mfma_f64_16x16x4_f64 (64 cycles/SIMD, 2048 flop): a wave owns a (BM/2 x BN/2) sub-tile as
//     RM x RN accumulator fragments that stay in registers for the whole k loop.
//   * Operand k-tiles (BK = 16 doubles = one 128-byte line per row) go HBM/L2 -> LDS with
//     global_load_lds_dwordx4 (LDS-DMA, no VGPR staging, no ds_write); double-buffered, one barrier
//     per k-tile.  The LDS image is row-major [row][16] with the eight 16-byte chunks of a row XOR-
//     swizzled by ((row >> 1) & 7): the DMA destination stays lane-linear (the swizzle is applied to
//     the per-lane SOURCE address) and both the DMA writes and the ds_read_b128 fragment reads are
//     bank-conflict free.
//   * One ds_read_b128 gives a lane two consecutive k values; the sum over k is order independent, so
//     MFMA step 2t+u of a k-tile contracts k = 8t + 2*(lane>>4) + u for A and B alike.
//   * The accumulators start from (beta/alpha)*C: the C tile is fetched in the prologue next to the
//     first operand tiles, the epilogue only stores.
//   * One output tile per workgroup, four workgroups per CU: the other three cover a workgroup's prologue and its DMA waits.
//     (A tile loop inside the launch, a persistent 128x128 variant and mixed tile sizes were built, measured slower and removed.)
//   * `tri` enumerates only the lower tiles of a SYRK-style update.
#include "common.hpp"

#define GM_BK 16
#define GPT_TRY_RC(expr) do { int rc_ = (expr); if (rc_ != GPT_OK) return rc_; } while (0)

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

__device__ __forceinline__ int64_t xcd_remap(int64_t pid, int64_t nwg)
{
    const int64_t q = nwg >> 3, r = nwg & 7;
    const int64_t xcd = pid & 7, local = pid >> 3;
    const int64_t base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

__device__ __forceinline__ void tile_decode(int64_t id, int tri, int64_t ntn, int64_t *ti, int64_t *tj)
{
    if (tri) {
        const int64_t ntri = ntn * (ntn + 1) / 2;
        if (id < ntri) {
            int64_t t = (int64_t)((sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
            while (t * (t + 1) / 2 > id) t--;
            while ((t + 1) * (t + 2) / 2 <= id) t++;
            *ti = t;
            *tj = id - t * (t + 1) / 2;
        } else {
            const int64_t r = id - ntri;
            *ti = ntn + r / ntn;
            *tj = r % ntn;
        }
    } else {
        *ti = id / ntn;
        *tj = id % ntn;
    }
}

// Per-lane source pointers of the LDS-DMA that stages an R-row operand tile: wave-instruction q covers rows
// [8q, 8q+8) of the tile, lane -> (row = 8q + lane/8, physical 16-byte chunk p = lane%8) and fetches the
// logical chunk p ^ ((row >> 1) & 7) of that row.  Each wave issues R/32 of the R/8 instructions.
template <int R>
__device__ __forceinline__ void stage_ptrs(const double *M, int64_t ld, int64_t r0, int64_t lim, int wave, int lane,
                                           const double *(&src)[R / 32])
{
#pragma unroll
    for (int qq = 0; qq < R / 32; qq++) {
        const int q = wave * (R / 32) + qq;
        const int row = q * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int64_t gr = r0 + row;
        if (gr >= lim) gr = lim - 1;
        src[qq] = M + gr * ld + c * 2;
    }
}

// LDS-DMA issue through inline asm: hipcc drains every builtin LDS-DMA with s_waitcnt vmcnt(0) before the next
// ds_read, which would expose the full memory latency in every k-tile; an asm statement is invisible to that
// bookkeeping, so the DMA stays in flight under the MFMAs and is drained by the explicit dma_wait() in front of
// the barrier that publishes the buffer (cdna_hip_programming.md section 5.7: M0 is written and restored inside
// the same statement).
__device__ __forceinline__ void glds16(const double *gsrc, unsigned lds_byte_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte_addr)
                 : "memory");
}

__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

template <int R>
__device__ __forceinline__ void stage_issue(const double *const (&src)[R / 32], int64_t koff, unsigned tile_lds, int wave)
{
#pragma unroll
    for (int qq = 0; qq < R / 32; qq++) {
        const int q = wave * (R / 32) + qq;
        glds16(src[qq] + koff, tile_lds + (unsigned)(q * 8 * GM_BK * sizeof(double)));
    }
}

// One k-tile of MFMAs for a wave: RM x RN fragments, A rows start at arow0, B rows at brow0 (tile-local).
template <int RM, int RN>
__device__ __forceinline__ void mma_ktile(const double *tA, const double *tB, int arow0, int brow0, int fr, int fk,
                                          f64x4 (&acc)[RM][RN])
{
    const int sw = (fr >> 1) & 7;
#pragma unroll
    for (int t = 0; t < GM_BK / 8; t++) {
        const int p = ((fk + 4 * t) ^ sw) * 2;
        f64x2 af[RM], bf[RN];
#pragma unroll
        for (int i = 0; i < RM; i++) af[i] = *reinterpret_cast<const f64x2 *>(tA + (arow0 + i * 16 + fr) * GM_BK + p);
#pragma unroll
        for (int j = 0; j < RN; j++) bf[j] = *reinterpret_cast<const f64x2 *>(tB + (brow0 + j * 16 + fr) * GM_BK + p);
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int i = 0; i < RM; i++)
#pragma unroll
                for (int j = 0; j < RN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i][u], bf[j][u], acc[i][j], 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------------
// One output tile per workgroup (used with 64x64 tiles for small / latency-bound updates).
// ------------------------------------------------------------------------------------------------
// Per-workgroup wall-clock stamps (scratch/gemm_stamps.hip compiles this file with -DGPT_GEMM_STAMPS; absent from the library)
#ifdef GPT_GEMM_STAMPS
__device__ long long *g_gemm_stamps;
#define GM_STAMP(i) do { if (threadIdx.x == 0) g_gemm_stamps[(long long)blockIdx.x * 8 + (i)] = (long long)wall_clock64(); } while (0)
#else
#define GM_STAMP(i) do { } while (0)
#endif

// C accesses.  Streaming C past the L2 (nontemporal loads and stores, -DGPT_GEMM_C_NT) leaves the L2 to the operand panels
// and cuts the fabric traffic of a 7168-row update from 503 + 207 MB to 395 + 207 MB (algorithmic 229 + 207), but the next
// kernels of the factorisation read what this one wrote: measured 4.995 against 4.862 ms at N = 8192 (27.48 against 27.65 at
// N = 16384), nontemporal loads alone 4.878, stores alone 4.900 -- so C stays cached.  Re-measured at the end of round 2
// (-DGPT_GEMM_C_NTLOAD, loads only): fetch 501 -> 421 MB per 7168-row launch (1.62x -> 1.44x the algorithmic bytes), N = 8192
// 4.563 -> 4.611 ms, N = 16384 27.83 -> 27.65 ms: the traffic is not the bound, the evaluation time decides.
#ifdef GPT_GEMM_C_NT
#define GM_LOADC(p) __builtin_nontemporal_load(p)
#define GM_STOREC(v, p) __builtin_nontemporal_store((v), (p))
#elif defined(GPT_GEMM_C_NTLOAD)
#define GM_LOADC(p) __builtin_nontemporal_load(p)
#define GM_STOREC(v, p) (*(p) = (v))
#else
#define GM_LOADC(p) (*(p))
#define GM_STOREC(v, p) (*(p) = (v))
#endif
// The tile body (everything behind "this workgroup's tile starts at (row0, col0)"), force-inlined into the kernel below.
template <int BM, int BN, int NSTAGE>
__device__ __forceinline__ void gemm_tile_body(
    int64_t row0, int64_t col0, int64_t tj, double *sA, double *sB,
    int64_t m, int64_t n, int64_t k, double alpha, const double *__restrict__ A, int64_t lda,
    const double *__restrict__ B, int64_t ldb, double beta, double *__restrict__ C, int64_t ldc,
    int64_t seg_cols, int64_t bskip, int prio, unsigned *edge, unsigned edge_val,
    unsigned edge_total, const unsigned *wait_word, unsigned wait_val, unsigned *wait_err, int edge_cols,
    const unsigned *tail_word, unsigned tail_val, unsigned *tail_err)
{
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int RM = WM / 16, RN = WN / 16;
    GM_STAMP(6);
    const int tid = threadIdx.x, lane = tid & 63;
    // (edge_cols > 0: only the workgroups of the first edge_cols tile columns -- first in the order table -- write through
    // and count towards the edge flag: the "urgent" part of a merged trailing update, see launch_gemm_nt)
    if (edge_cols > 0 && tj >= edge_cols) edge = nullptr;
    // In-kernel side of a flag edge (EdgeSig): C of this launch is produced by a kernel of another stream that raises
    // *wait_word to wait_val when it is through.  hipStreamWaitValue32 in front of this launch would be a kernel of its own
    // (__amd_rocclr_streamOpsWait, ~5 us on the chain); here the common case -- the word is already up -- costs one load.
    // C is then read past the L2 (the producer wrote it through to memory while this launch may already have been running).
    if (wait_word != nullptr) {
        // (~1000 cycles between polls: the waiting waves must not eat issue slots; bounded: common.hpp edge_poll; no acquire
        // fence: what the producer wrote -- this launch's C -- is read with agent-scope loads below)
        if (tid == 0) edge_poll<16, false>(wait_word, wait_val, wait_err);
        __syncthreads();
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fk = lane >> 4;

    const double *srcA[BM / 32], *srcB[BN / 32];
    stage_ptrs<BM>(A, lda, row0, m, wave, lane, srcA);
    // "staircase" launches (block-cyclic column segments): the B rows of column segment q start bskip rows further
    // down per segment; plain launches have seg_cols == 0
    const int64_t brow0 = seg_cols ? col0 + (col0 / seg_cols) * bskip : col0;
    stage_ptrs<BN>(B, ldb, brow0, seg_cols ? (int64_t)1 << 62 : n, wave, lane, srcB);
    const unsigned ldsA = lds_addr(sA), ldsB = lds_addr(sB);
    constexpr unsigned ABYTES = BM * GM_BK * sizeof(double), BBYTES = BN * GM_BK * sizeof(double);
    const int64_t nk = k / GM_BK;
    constexpr int PRE = NSTAGE - 1;                         // k-tiles requested ahead of the one being multiplied
    constexpr int PERSTAGE = BM / 32 + BN / 32;             // DMA instructions per wave per k-tile
    f64x4 acc[RM][RN];
    const double cs = (beta != 0.0) ? beta / alpha : 0.0;
    // The C tile is requested first, as RAW values (all sixteen loads per lane go out before anything waits: with the
    // scaling folded into the load expression the prologue measured 12 us per workgroup, 8.5 us this way), the first
    // operand tiles behind it, the scaling last.
#pragma unroll
    for (int i = 0; i < RM; i++)
#pragma unroll
        for (int j = 0; j < RN; j++) {
            const int64_t col = col0 + wn * WN + j * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int64_t row = row0 + wm * WM + i * 16 + fk + 4 * r;
                acc[i][j][r] = (beta != 0.0 && row < m && col < n)
                                   ? (wait_word ? __hip_atomic_load(&C[row * ldc + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                : GM_LOADC(&C[row * ldc + col]))
                                   : 0.0;
            }
        }
    GM_STAMP(7);
#pragma unroll
    for (int t = 0; t < PRE; t++)
        if (t < nk) {
            stage_issue<BM>(srcA, (int64_t)t * GM_BK, ldsA + t * ABYTES, wave);
            stage_issue<BN>(srcB, (int64_t)t * GM_BK, ldsB + t * BBYTES, wave);
        }
#pragma unroll
    for (int i = 0; i < RM; i++)
#pragma unroll
        for (int j = 0; j < RN; j++) acc[i][j] = acc[i][j] * cs;
    GM_STAMP(4);
    // k-tile 0 must have landed (the C loads above are older than nothing newer than the DMAs -> full drain is
    // correct here; in the loop the wait is counted)
    if (NSTAGE == 2 || nk < PRE) dma_wait();
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PRE - 1) * PERSTAGE) : "memory");
    GM_STAMP(5);
    __syncthreads();
#ifndef GPT_GEMM_NOPRIO
    // (prio: launches of the latency-bound panel stream keep a raised wave priority in the main loop, so that on a CU they
    // share with three workgroups of the main stream's trailing update their MFMAs issue first)
    if (prio == 0) __builtin_amdgcn_s_setprio(0);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
#endif
    GM_STAMP(1);

    for (int64_t kt = 0; kt < nk; kt++) {
        const int cur = (int)(kt % NSTAGE);
        if (kt + PRE < nk) {
            const int nxt = (int)((kt + PRE) % NSTAGE);
            stage_issue<BM>(srcA, (kt + PRE) * GM_BK, ldsA + nxt * ABYTES, wave);
            stage_issue<BN>(srcB, (kt + PRE) * GM_BK, ldsB + nxt * BBYTES, wave);
        }
        mma_ktile<RM, RN>(sA + cur * (BM * GM_BK), sB + cur * (BN * GM_BK), wm * WM, wn * WN, fr, fk, acc);
        // k-tile kt+1 must be complete before the barrier; up to PRE-1 younger k-tiles may stay in flight
        if (NSTAGE == 2 || kt + PRE >= nk) dma_wait();
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PRE - 1) * PERSTAGE) : "memory");
        __syncthreads();
    }
    GM_STAMP(2);

#pragma unroll
    for (int i = 0; i < RM; i++)
#pragma unroll
        for (int j = 0; j < RN; j++) {
            const int64_t col = col0 + wn * WN + j * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int64_t row = row0 + wm * WM + i * 16 + fk + 4 * r;
                if (row < m && col < n) {
                    // (a launch that raises an edge flag writes C through to memory: the waiting kernel starts before this
                    // one's end-of-kernel cache write-back)
                    if (edge) __hip_atomic_store(&C[row * ldc + col], alpha * acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else GM_STOREC(alpha * acc[i][j][r], &C[row * ldc + col]);
                }
            }
        }
    GM_STAMP(3);
    if (edge) edge_signal(edge, edge_val, edge_total);
    // Tail wait (EdgeSig `tail`, main stream's trailing updates): the launch does not END before *tail_word is up -- the
    // workgroup dispatched last polls it when its own tile is stored.  What the NEXT launch on this stream waits for (the next
    // panel, factored by the panel stream meanwhile) is thereby awaited without a wait kernel between the two launches (~5 us
    // per panel while the updates set the pace; the word is then up long before this point and the poll is one load), and the
    // next launch still starts behind a kernel boundary, i.e. with the caches invalidated AFTER the word was seen.
    if (tail_word != nullptr && blockIdx.x == gridDim.x - 1 && blockIdx.y == 0 && tid == 0) edge_poll<16, false>(tail_word, tail_val, tail_err);
}

template <int BM, int BN, int WPS, int NSTAGE>
__global__ __launch_bounds__(256, WPS) void gemm_nt_kernel(
    int64_t m, int64_t n, int64_t k, double alpha, const double *__restrict__ A, int64_t lda,
    const double *__restrict__ B, int64_t ldb, double beta, double *__restrict__ C, int64_t ldc,
    int tri, int64_t ntm, int64_t ntn, int64_t nwg, const int2 *__restrict__ order, int64_t seg_cols, int64_t bskip, int prio, unsigned *edge, unsigned edge_val,
    unsigned edge_total, const unsigned *wait_word, unsigned wait_val, unsigned *wait_err, int edge_cols, int64_t bstride,
    const unsigned *tail_word, unsigned tail_val, unsigned *tail_err, int64_t bstride_b, int64_t bstride_c)
{
    // batched launches (blockIdx.y = batch element).  gpt_fit_batch: independent small matrices, A, B and C all inside the
    // element's own matrix, one stride; the batched block inverses (api.hip build_block_inverses): three operands in three
    // different arrays, a stride each
    A += (int64_t)blockIdx.y * bstride;
    B += (int64_t)blockIdx.y * bstride_b;
    C += (int64_t)blockIdx.y * bstride_c;
    // NSTAGE LDS buffers per operand: 2 for the large launches (four workgroups per CU hide the DMA latency for each
    // other), 4 for the small, latency-bound launches of the panel (one workgroup per CU: the DMA of k-tile t+3 is
    // in flight while k-tile t is multiplied, drained with a COUNTED s_waitcnt vmcnt).
    __shared__ __attribute__((aligned(16))) double sA[NSTAGE * BM * GM_BK];
    __shared__ __attribute__((aligned(16))) double sB[NSTAGE * BN * GM_BK];

    // tile of this workgroup: from the host-built, XCD-aware order table (see tile_order()) or, without one, the
    // closed-form enumeration
    GM_STAMP(0);
#ifndef GPT_GEMM_NOPRIO
    // A new workgroup's waves are the YOUNGEST on their SIMDs and lose the issue arbitration to the three older
    // workgroups' main loops: measured with per-workgroup stamps (scratch/gemm_stamps.hip) the prologue -- C tile and
    // first operand tiles requested, then waited for -- took 20 us of a workgroup's 52 us at steady state against < 9 us
    // when a launch starts on an empty chip, so only half of the resident workgroups were multiplying.  The prologue
    // therefore runs at raised priority (its few dozen instructions go out at once; then the wave sleeps on memory).
    __builtin_amdgcn_s_setprio(3);
#endif
    int64_t ti, tj;
    if (order != nullptr) {
        const int2 t = order[blockIdx.x];
        if (t.x < 0) {
            if (tail_word != nullptr && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) edge_poll<16, false>(tail_word, tail_val, tail_err);
            return;
        }
        ti = t.x;
        tj = t.y;
    } else {
        tile_decode(xcd_remap(blockIdx.x, nwg), tri, ntn, &ti, &tj);
    }
    gemm_tile_body<BM, BN, NSTAGE>(ti * BM, tj * BN, tj, sA, sB, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, seg_cols, bskip, prio, edge,
                                   edge_val, edge_total, wait_word, wait_val, wait_err, edge_cols, tail_word, tail_val, tail_err);
}

// ---- XCD-aware tile order ------------------------------------------------------------------------------------
// Workgroup b runs on XCD b % 8 and the workgroups of one XCD start in the order of b / 8.  The needed tiles are put
// in ONE sequence -- supertiles of 64 x 8 tiles (rows x columns) in row-major order, each walked row by row -- and the
// sequence is cut into eight contiguous pieces of equal length, one per XCD.  The ~110 workgroups resident on an XCD then
// share 8 B-panels (kept in its L2 for 64 tile rows) and each A-panel eight times, and every XCD gets the same number of
// tiles to within one.  Measured on a 7168-row rank-384 update (scratch/pmc_order.sh, scratch/gemm_time.py): 8 x 8
// supertiles dealt round robin (round 1) 637 MB fetched / 412-417 us, the same cut evenly 674 MB / 402 us (the XCDs'
// lists differed by up to one supertile = 8 % of a small launch: 4096 rows 157 -> 145 us), 64 x 8 cut evenly 503 MB /
// 404 us, 16 x 16 846 MB (the streamed C tiles leave the panels well under the L2's 4 MB).  GPT_TILE_ORDER="rows,cols,mode"
// overrides (mode 0 = round-robin deal).  Tables are built once per (ntm, ntn, tri) and cached on the device; slots past
// an XCD's list hold (-1, -1).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>
struct TileOrder {
    int64_t ntm, ntn;
    int tri, dev;
    int64_t seg_t, rss_t;          // staircase tables: tiles per column segment, row-start step per segment (tiles)
    int64_t g_off, g_num, g_den, g_base;      // grid staircase (tri == 3), see tile_needed
    int2 *d_tab;
    int64_t grid;
    int64_t ntiles;                // entries of the table that hold a tile (the rest are (-1, -1) padding)
    int64_t edge_cols, nedge;      // partial edge flag: the tiles of the first edge_cols columns come first; their number
    uint64_t last_use;             // tick of the last look-up (eviction takes the least recently used tables)
    bool pinned;                   // looked up during a stream capture: a captured graph holds d_tab, never evicted
};
static std::vector<TileOrder> g_orders;
static std::mutex g_orders_mu;
static uint64_t g_orders_tick = 0;

// tri == 2: staircase.  Column segment q = j / seg_t starts (its diagonal block) at tile row q * rss_t; tile (i, j) is
// needed iff i >= q * rss_t + (j - q * seg_t).
// tri == 3: grid staircase (2-D block-cyclic layout, gptools_amd/dist.py GridLML).  Column segment q is local block column q of
// the update, global block column J = J0 + q * num; the local block rows hold the global block rows I = pr + li * den.  Its
// first needed block row is the first I >= J:  rs(q) = ceil((off + q * num) / den) - base  with off = J0 - pr and base = the
// local index of the update's first block row.  Below that row the segment is a full rectangle; the first block itself is a
// DIAGONAL block of the matrix iff (off + q * num) is a multiple of den, and then only its lower tiles are needed.
struct GridStair { int64_t off = 0, num = 0, den = 1, base = 0; };
static inline bool tile_needed(int tri, int64_t i, int64_t j, int64_t seg_t, int64_t rss_t, const GridStair &g = GridStair())
{
    if (tri == 1) return j <= i;
    if (tri == 2) {
        const int64_t q = j / seg_t;
        return i >= q * rss_t + (j - q * seg_t);
    }
    if (tri == 3) {
        const int64_t q = j / seg_t, v = g.off + q * g.num;
        const int64_t i0 = ((v + g.den - 1) / g.den - g.base) * seg_t;
        if (i < i0) return false;
        if (v % g.den == 0 && i < i0 + seg_t) return (i - i0) >= (j - q * seg_t);
        return true;
    }
    return true;
}

static int tile_order(int64_t ntm, int64_t ntn, int tri, const int2 **tab, int64_t *grid, int64_t seg_t = 0,
                      int64_t rss_t = 0, int64_t *ntiles = nullptr, int64_t edge_cols = 0, int64_t *nedge = nullptr,
                      const GridStair &gs = GridStair(), hipStream_t st = nullptr)
{
    static int sgm = 0, sgn = 0, mode = 0;
    if (sgm == 0) {
        sgm = 64, sgn = 8, mode = 1;
        if (const char *e = getenv("GPT_TILE_ORDER")) {
            int a = 0, b = 0, c = 0;
            if (sscanf(e, "%d,%d,%d", &a, &b, &c) >= 2 && a > 0 && b > 0) sgm = a, sgn = b, mode = c;
        }
    }
    int dev = 0;
    GPT_HIP_CHECK(hipGetDevice(&dev));
    // (a look-up under stream capture: the captured launch keeps the table's address for every replay)
    bool capturing = false;
    if (st != nullptr) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess) capturing = (cs != hipStreamCaptureStatusNone);
        else (void)hipGetLastError();
    }
    std::lock_guard<std::mutex> lk(g_orders_mu);
    for (auto &o : g_orders)
        if (o.ntm == ntm && o.ntn == ntn && o.tri == tri && o.dev == dev && o.seg_t == seg_t && o.rss_t == rss_t && o.edge_cols == edge_cols &&
            o.g_off == gs.off && o.g_num == gs.num && o.g_den == gs.den && o.g_base == gs.base) {
            o.last_use = ++g_orders_tick;
            o.pinned = o.pinned || capturing;
            *tab = o.d_tab;
            *grid = o.grid;
            if (ntiles) *ntiles = o.ntiles;
            if (nedge) *nedge = o.nedge;
            return GPT_OK;
        }
    // A cap (the block-cyclic engines make a table per shape -- ADVICE r4): at 1024 tables the least recently used half of the
    // unpinned ones goes.  Tables may be in use by launches in flight, so only behind a device-wide synchronisation -- which is
    // not allowed while a stream of the process is capturing: then (and whenever the synchronisation fails) nothing is evicted
    // and the cache grows until the next miss outside a capture.  Done BEFORE the new table is allocated.
    if (g_orders.size() >= 1024 && !capturing) {
        if (hipDeviceSynchronize() == hipSuccess) {
            std::vector<uint64_t> ticks;
            for (const auto &o : g_orders)
                if (!o.pinned) ticks.push_back(o.last_use);
            if (ticks.size() >= 2) {
                std::nth_element(ticks.begin(), ticks.begin() + ticks.size() / 2, ticks.end());
                const uint64_t cut = ticks[ticks.size() / 2];
                std::vector<TileOrder> keep;
                for (const auto &o : g_orders) {
                    if (!o.pinned && o.last_use < cut) (void)hipFree(o.d_tab);
                    else keep.push_back(o);
                }
                g_orders.swap(keep);
            }
        } else {
            (void)hipGetLastError();
        }
    }
    std::vector<std::vector<int2>> per(8);
    std::vector<int2> seq, sequ;          // sequ: the tiles of the first edge_cols columns (they go first on every XCD)
    const int64_t sm = (ntm + sgm - 1) / sgm, sn = (ntn + sgn - 1) / sgn;
    int64_t sidx = 0;
    for (int64_t si = 0; si < sm; si++)
        for (int64_t sj = 0; sj < sn; sj++) {
            std::vector<int2> &dst = (mode || edge_cols > 0) ? seq : per[sidx % 8];
            bool any = false;
            for (int64_t i = si * sgm; i < (si + 1) * sgm && i < ntm; i++)
                for (int64_t j = sj * sgn; j < (sj + 1) * sgn && j < ntn; j++) {
                    if (!tile_needed(tri, i, j, seg_t, rss_t, gs)) continue;
                    (j < edge_cols ? sequ : dst).push_back(make_int2((int)i, (int)j));
                    any = true;
                }
            if (any) sidx++;
        }
    if (mode || edge_cols > 0) {
        for (const std::vector<int2> *sq : {&sequ, &seq}) {
            const size_t T = sq->size(), q = T / 8, r = T % 8;
            size_t at = 0;
            for (int x = 0; x < 8; x++) {
                const size_t len = q + ((size_t)x < r ? 1 : 0);
                per[x].insert(per[x].end(), sq->begin() + at, sq->begin() + at + len);
                at += len;
            }
        }
    }
    size_t mx = 0;
    for (auto &v : per) mx = v.size() > mx ? v.size() : mx;
    std::vector<int2> flat(mx * 8, make_int2(-1, -1));
    for (int x = 0; x < 8; x++)
        for (size_t l = 0; l < per[x].size(); l++) flat[l * 8 + x] = per[x][l];
    TileOrder o;
    o.ntm = ntm;
    o.ntn = ntn;
    o.tri = tri;
    o.dev = dev;
    o.seg_t = seg_t;
    o.rss_t = rss_t;
    o.g_off = gs.off;
    o.g_num = gs.num;
    o.g_den = gs.den;
    o.g_base = gs.base;
    o.edge_cols = edge_cols;
    o.nedge = (int64_t)sequ.size();
    o.grid = (int64_t)flat.size();
    o.ntiles = 0;
    for (auto &v : per) o.ntiles += (int64_t)v.size();
    o.d_tab = nullptr;
    GPT_HIP_CHECK(hipMalloc(&o.d_tab, flat.size() * sizeof(int2)));
    {   // upload on a private non-blocking stream: the caller may be inside a stream capture (hipGraph option), where
        // a copy on the legacy stream would be an illegal dependency on the capturing stream
        hipStream_t up = nullptr;
        GPT_HIP_CHECK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
        hipError_t e1 = hipMemcpyAsync(o.d_tab, flat.data(), flat.size() * sizeof(int2), hipMemcpyHostToDevice, up);
        hipError_t e2 = hipStreamSynchronize(up);
        hipStreamDestroy(up);
        if (e1 != hipSuccess || e2 != hipSuccess) (void)hipFree(o.d_tab);
        GPT_HIP_CHECK(e1);
        GPT_HIP_CHECK(e2);
    }
    o.last_use = ++g_orders_tick;
    o.pinned = capturing;
    g_orders.push_back(o);
    *tab = o.d_tab;
    *grid = o.grid;
    if (ntiles) *ntiles = o.ntiles;
    if (nedge) *nedge = o.nedge;
    return GPT_OK;
}

template <int BM, int BN, int WPS, int NSTAGE>
static int gemm_launch_t(hipStream_t st, int64_t m, int64_t n, int64_t k, double alpha, const double *A,
                         int64_t lda, const double *B, int64_t ldb, double beta, double *C, int64_t ldc, int tri,
                         int lds_pad, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, int64_t seg_cols = 0,
                         int64_t bskip = 0, int64_t row_step = 0, int prio = 0, EdgeSig edge = EdgeSig(),
                         EdgeSig wait = EdgeSig(), int64_t edge_cols_elems = 0, int64_t nbatch = 1, int64_t bstride = 0,
                         EdgeSig tail = EdgeSig(), int64_t bstride_b = -1, int64_t bstride_c = -1, const GridStair &gs = GridStair())
{
    if (bstride_b < 0) bstride_b = bstride;
    if (bstride_c < 0) bstride_c = bstride;
    const int64_t ntm = (m + BM - 1) / BM, ntn = (n + BN - 1) / BN;
    int64_t nwg = (tri == 1) ? ntn * (ntn + 1) / 2 + (ntm - ntn) * ntn : ntm * ntn;
    int64_t nreal = nwg;                   // workgroups that compute a tile (and count towards an edge flag)
    int64_t nedge = 0;                     // ... of which in the first edge_cols columns (partial edge flag)
    const int2 *order = nullptr;
    if (tri == 2) {                        // staircase: the tile list always comes from a table
        int64_t grid = 0;
        GPT_TRY_RC(tile_order(ntm, ntn, 2, &order, &grid, seg_cols / BN, row_step / BM, &nreal, 0, nullptr, GridStair(), st));
        nwg = grid;
    } else if (tri == 3) {                 // grid staircase: likewise
        int64_t grid = 0;
        GPT_TRY_RC(tile_order(ntm, ntn, 3, &order, &grid, seg_cols / BN, 0, &nreal, 0, nullptr, gs, st));
        nwg = grid;
    } else if (nwg >= 512) {               // large launches only: small ones live in L2 anyway
        int64_t grid = 0;
        GPT_TRY_RC(tile_order(ntm, ntn, tri, &order, &grid, 0, 0, &nreal, edge_cols_elems / BN, &nedge, GridStair(), st));
        nwg = grid;
    }
    if (edge_cols_elems > 0 && (order == nullptr || !edge.word || nedge <= 0)) {
        gpt_set_error("gemm_nt: a partial edge flag needs a launch large enough for an order table");
        return GPT_E_ARG;
    }
    if (nwg <= 0) return GPT_OK;
    // lds_pad bytes of unused dynamic LDS cap the residency of the 64x64 kernel (32 KiB static): the trailing
    // update on the main stream asks for 8 KiB -> three workgroups per CU, leaving 40 KiB of LDS and over 40 % of the
    // register file on every CU to the high-priority panel stream (whose own GEMMs and TRSMs need 32 / 9 KiB).
    const size_t dyn = (BM == 64) ? (size_t)lds_pad : 0;
    // timing events (roofline line of bench.py) ride on the dispatch packet itself: separate hipEventRecord calls
    // would add two barrier packets per launch to the stream being measured
    if (ev0 || ev1)
        hipExtLaunchKernelGGL((gemm_nt_kernel<BM, BN, WPS, NSTAGE>), dim3((unsigned)nwg, (unsigned)nbatch), dim3(256), dyn, st, ev0, ev1, 0,
                              m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tri, ntm, ntn, nwg, order, seg_cols, bskip, prio, edge.word, edge.value,
                              (unsigned)(edge_cols_elems > 0 ? nedge : nreal), wait.word, wait.value, wait.err, (int)(edge_cols_elems / BN), bstride,
                              tail.word, tail.value, tail.err, bstride_b, bstride_c);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WPS, NSTAGE>), dim3((unsigned)nwg, (unsigned)nbatch), dim3(256), dyn, st, m, n, k, alpha,
                           A, lda, B, ldb, beta, C, ldc, tri, ntm, ntn, nwg, order, seg_cols, bskip, prio, edge.word, edge.value,
                              (unsigned)(edge_cols_elems > 0 ? nedge : nreal), wait.word, wait.value, wait.err, (int)(edge_cols_elems / BN), bstride,
                              tail.word, tail.value, tail.err, bstride_b, bstride_c);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

// Launches of fewer than this many 64x64 tiles are cut into 32x32 tiles (see launch_gemm_nt); api.hip sizes the in-kernel
// wait budget of the panel stream's updates with the same number.
int gemm_small_threshold()
{
    static int small_below = -1;
    if (small_below < 0) {
        small_below = 512;
        if (const char *e = getenv("GPT_GEMM_SMALL")) small_below = atoi(e);
    }
    return small_below;
}

int launch_gemm_nt(hipStream_t st, int64_t m, int64_t n, int64_t k, double alpha, const double *A, int64_t lda,
                   const double *B, int64_t ldb, double beta, double *C, int64_t ldc, int tri, int force_tile, int lds_pad,
                   hipEvent_t ev0, hipEvent_t ev1, int prio, EdgeSig edge, EdgeSig wait, int64_t edge_cols, int64_t nbatch,
                   int64_t bstride, EdgeSig tail, int64_t bstride_b, int64_t bstride_c)
{
    gpt_jitter(st);
    if (m <= 0 || n <= 0) {
        if (edge.word) {
            gpt_set_error("gemm_nt: an empty launch cannot raise an edge flag");
            return GPT_E_ARG;
        }
        return GPT_OK;
    }
    if (alpha == 0.0) {
        gpt_set_error("gemm_nt: alpha must be non-zero");
        return GPT_E_ARG;
    }
    if (k <= 0 || (k % GM_BK) != 0 || (lda & 1) || (ldb & 1)) {
        gpt_set_error("gemm_nt: k must be a positive multiple of %d and lda/ldb even (k=%lld)", GM_BK, (long long)k);
        return GPT_E_ARG;
    }
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15)) {
        gpt_set_error("gemm_nt: A and B must be 16-byte aligned");
        return GPT_E_ARG;
    }
    if (tri && m < n) {
        gpt_set_error("gemm_nt: tri requires m >= n");
        return GPT_E_ARG;
    }
    if (nbatch > 1 && (force_tile != 0 || edge.word || wait.word || edge_cols || ev0 || ev1)) {
        gpt_set_error("gemm_nt: batched launches take the default tiles, no edges and no events");
        return GPT_E_ARG;
    }
    int tile = force_tile;
    if (tile == 0) {
        tile = 64;   // measured on MI355X: 64x64 tiles at 4-5 workgroups per CU beat the 128x128 variants
        // Launches of fewer than 512 64x64 tiles -- two per CU: the rank-128 updates of the chain (71 x 6 tiles at 4500 rows,
        // 16 x 6 at 1000) -- are cut into 32x32 tiles instead: four times the workgroups, a quarter of the MFMAs per wave
        // (128 -> 32 at k = 128), so a launch that is mostly latency gets off the chain sooner.  Same sums in the same
        // order, bit-identical results.  Measured (scratch/env_ab.py, GPT_GEMM_SMALL = threshold): N = 8192 4.757 ->
        // 4.625 ms at 512 (4.709 at 256, 4.648 at 1024, 4.692 at 2048), N = 4096 1.468 -> 1.386, N = 16384 27.28 -> 27.15.
        const int small_below = gemm_small_threshold();
        // (batched: the tile choice follows the SINGLE matrix, so that an element of a batch is computed exactly as alone)
        const int64_t nt64 = ((m + 63) / 64) * ((n + 63) / 64);
        if (nt64 < small_below && !ev0 && edge_cols == 0) tile = 32;
    }
    if (tile == 32) {
        // Small launches with a LONG k loop (the leaf / update GEMMs of a triangular solve with few right-hand sides: predict
        // at a handful of points, k = 512) run one workgroup per CU, so nobody hides the DMA latency of the 2-stage pipeline:
        // 0.65 us per 16-wide k-tile for 4 MFMAs per wave.  Four LDS stages (DMA three k-tiles ahead, counted vmcnt) bring it
        // to ~0.3 us: predict with std at 64 points, N = 8192: 0.91 -> 0.62 ms.  The rank-128 updates of the factorisation's
        // chain (8 k-tiles, mostly prologue) do not gain (N = 4096: 1.242 / 1.250 / 1.261 ms with 2 / 3 / 4 stages), hence
        // the k threshold.  GPT_GEMM_SMALL_STAGES=2|4 forces one variant.
        static int stages = -1;
        if (stages < 0) {
            stages = 0;
            if (const char *e = getenv("GPT_GEMM_SMALL_STAGES")) stages = atoi(e);
        }
        if ((stages == 4 && k >= 64) || (stages == 0 && k >= 256))
            return gemm_launch_t<32, 32, 2, 4>(st, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tri, 0, ev0, ev1, 0, 0, 0, prio, edge, wait, 0, nbatch, bstride, tail, bstride_b, bstride_c);
        return gemm_launch_t<32, 32, 2, 2>(st, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tri, 0, ev0, ev1, 0, 0, 0, prio, edge, wait, 0, nbatch, bstride, tail, bstride_b, bstride_c);
    }
    if ((edge.word || wait.word || edge_cols || tail.word) && tile != 64) {
        gpt_set_error("gemm_nt: edge flags exist for the 64x64 / 32x32 kernels only");
        return GPT_E_ARG;
    }
    // (measured and removed: 128x128 tiles, one workgroup per CU, persistent or not -- 47 against 58 TFLOP/s in the main loop, round 1; a
    // 4-stage 64x64 variant -- 41 against 48 TFLOP/s, round 3; a tile loop inside the launch and quarter tiles for a launch's last
    // partial round -- rounds 4 and 5, NOTES_r04.md / NOTES_r05.md; the sources are in the history up to round 5's last commit)
    if (tile != 64) {
        gpt_set_error("gemm_nt: option tile takes 0 (by size), 32 or 64");
        return GPT_E_ARG;
    }
    return gemm_launch_t<64, 64, 2, 2>(st, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tri, lds_pad, ev0, ev1, 0, 0, 0, prio, edge, wait, edge_cols,
                                       nbatch, bstride, tail, bstride_b, bstride_c);
}

// Staircase update: C (m x nseg*seg_cols) += alpha * A B_q^T per column segment q, where segment q (seg_cols
// columns of C) takes its B rows from B + q * b_stride rows and only its rows >= q * row_step are touched, lower
// trapezoid inside (its "diagonal block" sits at row q * row_step).  This is the shape of a block-cyclic rank's
// trailing update -- its block columns are adjacent in local storage but W * nb apart in the global matrix -- done in
// ONE launch (one tail, one XCD-aware tile order) instead of one launch per block column.
int launch_gemm_nt_stair(hipStream_t st, int64_t m, int64_t nseg, int64_t seg_cols, int64_t k, double alpha,
                         const double *A, int64_t lda, const double *B, int64_t ldb, int64_t b_stride, int64_t row_step,
                         double beta, double *C, int64_t ldc, int lds_pad, hipEvent_t ev0, hipEvent_t ev1)
{
    gpt_jitter(st);
    if (m <= 0 || nseg <= 0) return GPT_OK;
    if (alpha == 0.0 || k <= 0 || (k % GM_BK) != 0 || (lda & 1) || (ldb & 1) || ((uintptr_t)A & 15) ||
        ((uintptr_t)B & 15) || seg_cols <= 0 || (seg_cols % 64) || (row_step % 64) || b_stride < seg_cols) {
        gpt_set_error("gemm_nt_stair: bad arguments (k multiple of %d, seg_cols / row_step multiples of 64, 16-byte "
                      "aligned operands)", GM_BK);
        return GPT_E_ARG;
    }
    return gemm_launch_t<64, 64, 2, 2>(st, m, nseg * seg_cols, k, alpha, A, lda, B, ldb, beta, C, ldc, 2, lds_pad, ev0,
                                       ev1, seg_cols, b_stride - seg_cols, row_step);
}

// Grid staircase (2-D block-cyclic trailing update, see tile_needed): C (m x nseg*seg_cols) += alpha * A * B^T where column
// segment q takes its B rows from B + q * seg_cols rows and only the block rows from  ceil((off + q * num) / den) - base  on
// are touched.  One launch per step and rank, one XCD-aware tile order.
int launch_gemm_nt_gridstair(hipStream_t st, int64_t m, int64_t nseg, int64_t seg_cols, int64_t k, double alpha, const double *A,
                             int64_t lda, const double *B, int64_t ldb, int64_t off, int64_t num, int64_t den, int64_t base,
                             double beta, double *C, int64_t ldc, int lds_pad, hipEvent_t ev0, hipEvent_t ev1)
{
    gpt_jitter(st);
    if (m <= 0 || nseg <= 0) return GPT_OK;
    if (alpha == 0.0 || k <= 0 || (k % GM_BK) != 0 || (lda & 1) || (ldb & 1) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) ||
        seg_cols <= 0 || (seg_cols % 64) || num <= 0 || den <= 0 || off <= -den) {
        gpt_set_error("gemm_nt_gridstair: bad arguments (k multiple of %d, seg_cols a multiple of 64, num, den > 0, 16-byte "
                      "aligned operands)", GM_BK);
        return GPT_E_ARG;
    }
    GridStair gs;
    gs.off = off;
    gs.num = num;
    gs.den = den;
    gs.base = base;
    return gemm_launch_t<64, 64, 2, 2>(st, m, nseg * seg_cols, k, alpha, A, lda, B, ldb, beta, C, ldc, 3, lds_pad, ev0, ev1, seg_cols, 0,
                                       0, 0, EdgeSig(), EdgeSig(), 0, 1, 0, EdgeSig(), -1, -1, gs);
}
