// potrf.hip -- leaf kernels of the blocked right-looking Cholesky for gfx950 (fp64).
//
// Replaces LAPACK dpotrf as reached by scipy.linalg.cholesky(K_tot, lower=True)
// (ref: gptools/gaussian_process.py:1452).  The host-side blocking (outer block columns, recursive
// panel, look-ahead on a second stream) lives in api.hip; this file holds
//   potf2_diag_kernel  : one workgroup factors a 128x128 diagonal block entirely inside LDS,
//                        16-column inner blocks; inner TRSM/SYRK on v_mfma_f64_16x16x4_f64; the
//                        16x16 pivot blocks are factored by one wave with one matrix row per lane
//                        and v_readlane broadcasts (no barriers inside a pivot block); also emits
//                        the inverses of the 16x16 diagonal blocks of L ("invd").
//   trsm_panel_kernel  : X * L_kk^T = B for the rows below a diagonal block; one wave per 16
//                        rows, blocked forward substitution, every multiply on MFMA; the
//                        accumulator -> operand re-layout goes through a private LDS strip.
// MFMA f64 16x16x4 layouts (cdna_hip_programming.md section 3, verified by scratch/mfma_probe.hip):
//   A operand: lane l holds A[i = l & 15][k = l >> 4];  B operand: lane l holds B[k = l >> 4][j = l & 15];
//   C/D: lane l, register r holds D[row = (l >> 4) + 4 r][col = l & 15].
#include "common.hpp"

#define PD_NB 128
#define PD_PITCH 130          // LDS row pitch (doubles): 16 rows x (lane>>4) fragment reads are conflict-free
#define PD_TP 18              // pitch of the 16x16 inverse scratch

// Phase stamps of wave 0 (scratch/potf2_stamps.hip compiles this file with -DGPT_PD_STAMPS; absent from the library).
#ifdef GPT_PD_STAMPS
// cheap stamps: s_memtime into a small static LDS array (lane 0 of wave 0), dumped to global memory at the end of the kernel
__device__ long long *g_pd_stamps;
__shared__ long long pd_stamp_lds[128];
#define PD_STAMP(i)                                                             \
    do {                                                                        \
        if (wave == 0 && lane == 0) pd_stamp_lds[(i)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#define PD_STAMP1(i)                                                            \
    do {                                                                        \
        if (wave == 1 && lane == 0) pd_stamp_lds[(i)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#define PD_STAMPW(i)                                                            \
    do {                                                                        \
        if (lane == 0) pd_stamp_lds[(i)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#define PD_STAMP_DUMP()                                                         \
    do {                                                                        \
        __syncthreads();                                                        \
        if (threadIdx.x < 128) g_pd_stamps[threadIdx.x] = pd_stamp_lds[threadIdx.x]; \
    } while (0)
// event trace of every wave (look-ahead body; -DGPT_PD_TRACE on top of the stamps: every point costs several hundred cycles):
// (time, code) pairs appended to a global array, lane 0 of the wave
__device__ long long *g_pd_trace;
#ifdef GPT_PD_TRACE
#define PD_TRACE_DECL int pd_tr_n = 0
#define PD_TRACE(code)                                                          \
    do {                                                                        \
        if (lane == 0 && g_pd_trace != nullptr && pd_tr_n < 250) {              \
            long long *q_ = g_pd_trace + ((long long)wave * 256 + pd_tr_n) * 2; \
            q_[0] = (long long)__builtin_amdgcn_s_memtime();                    \
            q_[1] = (long long)(code);                                          \
        }                                                                       \
        pd_tr_n++;                                                              \
    } while (0)
#else
#define PD_TRACE_DECL do { } while (0)
#define PD_TRACE(code) do { } while (0)
#endif
#else
#define PD_TRACE_DECL do { } while (0)
#define PD_TRACE(code) do { } while (0)
#define PD_STAMP(i) do { } while (0)
#define PD_STAMP1(i) do { } while (0)
#define PD_STAMPW(i) do { } while (0)
#define PD_STAMP_DUMP() do { } while (0)
#endif

// wall-clock stamps of the consumer waves (scratch/r05_upd_stamps.hip compiles this file with -DGPT_PU_STAMPS; absent from the library):
// g_pu_stamps[(workgroup * 8 + wave) * 32 + i], 100 MHz
#ifdef GPT_PU_STAMPS
__device__ long long *g_pu_stamps;
#define PU_STAMP(i) do { if (lane == 0 && g_pu_stamps) g_pu_stamps[((long long)blockIdx.x * 8 + wave) * 32 + (i)] = (long long)wall_clock64(); } while (0)
#else
#define PU_STAMP(i) do { } while (0)
#endif

// linear index of a lower-triangle tile -> (row a, column b <= a); t is wave-uniform, so this is a scalar loop
__device__ __forceinline__ void tri_decode(int t, int &a_, int &b_)
{
    a_ = 0;
    while (t > a_) {
        t -= a_ + 1;
        a_++;
    }
    b_ = t;
}

// S(ti, tj) -= S(ti, jb) S(tj, jb)^T on 16x16 tiles; the four k-steps run as two independent accumulator chains
struct TileUpd {
    f64x4 acc, acc2;
    double av[4], bv[4];
    int ti, tj;
    __device__ __forceinline__ void load(double (*S)[PD_PITCH], int ti_, int tj_, int jb, int fr, int fk)
    {
        ti = ti_;
        tj = tj_;
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = S[ti * 16 + fk + 4 * r][tj * 16 + fr];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            av[kk] = -S[ti * 16 + fr][jb * 16 + fk + 4 * kk];
            bv[kk] = S[tj * 16 + fr][jb * 16 + fk + 4 * kk];
        }
        acc2 = f64x4{0.0, 0.0, 0.0, 0.0};
    }
    __device__ __forceinline__ void mma()
    {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], acc2, 0, 0, 0);
    }
    __device__ __forceinline__ void store(double (*S)[PD_PITCH], int fr, int fk)
    {
#pragma unroll
        for (int r = 0; r < 4; r++) S[ti * 16 + fk + 4 * r][tj * 16 + fr] = acc[r] + acc2[r];
    }
};


__device__ __forceinline__ double bcast_lane(double v, int srclane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(d) from v_rsq_f64 (relative error ~2^-26) plus one Newton step: error 1.5 * 2^-52 -- an extra rounding of
// the same size as the Cholesky's own.  The pivot chain of the factorisation is instruction-issue bound on a single
// wave, so the IEEE sqrt + divide sequences (~60 instructions) are replaced by 5; the diagonal entry sqrt(d) is
// simply d * (1/sqrt(d)).  Checked against scipy/LAPACK factors in tests/test_gpu_parity.py.
// The five steps are kept separate (struct RsqPipe) so that pivot_col can issue them one at a time BETWEEN the
// broadcast/FMA groups of the previous column: the wave issues in order, and a dependent chain placed in one piece
// would stall it for its whole latency.
struct RsqPipe {
    double d, y0, t, h, u, inv;
    __device__ __forceinline__ void step(int k)
    {
        if (k == 0) y0 = __builtin_amdgcn_rsq(d);
        else if (k == 1) { t = d * y0; h = 0.5 * y0; }
        else if (k == 2) u = fma(-t, y0, 1.0);
        else if (k == 3) inv = fma(h, u, y0);
    }
};

// Row broadcasts by DPP.  The four quarters of the wave hold the same 16 matrix rows, so "the value lane C holds" is
// a broadcast inside each row of 16 lanes: gfx950 has it as a modifier of the DP multiply-add itself,
//   v_fmac_f64_dpp  acc, src row_newbcast:C, mult      acc += (src of lane C of my row) * mult,
// one instruction where v_readlane x2 -> SGPR pair -> v_fma took three (and an SGPR round trip).  A DPP read of a VGPR
// that a VALU instruction has just written needs two wait states; the s_nop sits in the statements that can follow
// such a write directly.
template <int C>
__device__ __forceinline__ void fmac_bcast(double &acc, double src, double mult)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mult), "i"(C));
}
// acc -= (src of lane C of my row) * mult: the sign rides on the source modifier, no separate negation
template <int C>
__device__ __forceinline__ void fnmac_bcast(double &acc, double src, double mult)
{
    asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mult), "i"(C));
}
template <int C>
__device__ __forceinline__ double mov_bcast_nop(double src)
{
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "i"(C));
    return r;
}

// One column J of the pivot block; on entry rp.inv = 1/sqrt(pivot J) (the same value in every lane).  The NEXT pivot
// is a[J+1][J+1] - l^2 with l = L[J+1][J] (two row broadcasts), and its rsqrt pipeline is advanced one step per
// update group of this column.
// One column J of the pivot block; on entry rp.inv = 1/sqrt(pivot J) (the same value in every lane).  The NEXT pivot
// is a[J+1][J+1] - l^2 with l = L[J+1][J] (two row broadcasts), and its rsqrt pipeline is advanced one step per
// update group of this column.  Measured with s_memtime stamps (scratch/potf2_stamps.hip): the sixteen columns take
// ~2700 cycles, i.e. the chain is bound by the issue of its ~640 double-precision instructions (~4 cycles each), so
// what counts is their number: the sign of the update rides on a source modifier of the DPP multiply-add, and a pivot
// that is not positive is NOT tested here (the test, a select and the bookkeeping were 5 instructions per column) --
// v_rsq of a negative number is NaN and 0 * rsq(0) = 0 * inf is NaN, so the first bad column leaves a NaN on the
// diagonal of L and pivot_block_16 finds it there after the block.  (A variant built around 1/d from v_rcp_f64, whose
// dependent chain per column is 4 operations instead of 7, needs 6 more instructions per column and measured the same.)
template <int J, int C>
__device__ __forceinline__ void pivot_group(double (&a)[16], double (&x)[16], RsqPipe &np)
{
    if constexpr (C < 16) {
        fnmac_bcast<C>(a[C], a[J], a[J]);                  // a[r][C] -= L[r][J] * L[C][J]
        fnmac_bcast<C>(x[C], a[J], x[J]);                  // x[C]    -= L[C][J] * x[J]
        np.step(C - J - 1);                                // C = J+1 carries step 0 (v_rsq), ... J+4 the last
        pivot_group<J, C + 1>(a, x, np);
    }
}

template <int J>
__device__ __forceinline__ void pivot_col(double (&a)[16], double (&x)[16], RsqPipe &rp)
{
    const double inv = rp.inv;
    a[J] *= inv;                                          // lane J: d * inv = sqrt(d)
    x[J] *= inv;
    if constexpr (J + 1 < 16) {
        RsqPipe np;
        const double l = mov_bcast_nop<J + 1>(a[J]), q = mov_bcast_nop<J + 1>(a[J + 1]);
        np.d = fma(-l, l, q);
        // (no scheduling fences inside the block: measured 24.4 us per 128-block against 26.9 with one per group and
        // 27.1 with one per column -- left alone, hipcc overlaps the head of column J+1 with the tail of column J)
        pivot_group<J, J + 1>(a, x, np);
#pragma unroll
        for (int k = 15 - J; k < 4; k++) np.step(k);      // columns with fewer than four groups finish the pipeline here
        rp = np;
    }
}

// (Round 4: the same recurrence with EVERY instruction placed by hand -- volatile asm statements, the updates of column J - 1
// spread over the gaps of column J's dependent chain, the new pivot read after the first update instead of two broadcasts and
// an fma -- measured 2564-2612 cycles against 2492-2576 for what hipcc makes of the code above: no gain, not kept.  The block is
// bound by the issue of its ~416 double-precision instructions on one SIMD, ~6 cycles each beside the store wave;
// scratch/dpp_rate.hip has the issue and latency numbers, chain_slot below the hand-placed form.)
// Factor the 16x16 pivot block jb of S with one wave: lane (l & 15) keeps ROW l of the block in a[0..15] and, at
// the same time, COLUMN l of inv(L_jj) in x[0..15]; both recurrences consume the same broadcast L[c][j], so the
// inverse costs one extra FMA per broadcast.  Writes L_jj back to S and inv(L_jj) to T (LDS; one buffer per pivot
// block, copied to the global workspace by the storing wave).  FROM_GLOBAL: the rows come straight from global
// memory (first block: the pivot starts while the other waves are still staging the 128x128 block into LDS).
template <bool FROM_GLOBAL>
__device__ __forceinline__ void pivot_block_16(double (*S)[PD_PITCH], double (*T)[PD_TP], int jb, int lane,
                                               const double *Ag, int64_t lda, int32_t *info, int64_t info_col0)
{
    const int row = lane & 15;
    double a[16], x[16];
#pragma unroll
    for (int c = 0; c < 16; c++) {
        a[c] = FROM_GLOBAL ? Ag[(int64_t)row * lda + c] : S[jb * 16 + row][jb * 16 + c];
        x[c] = (c == row) ? 1.0 : 0.0;
    }
    constexpr int wave = 0;                 // (PD_STAMP)
    (void)wave;
    PD_STAMP(8 + jb * 8 + 6);
    __builtin_amdgcn_sched_barrier(0);      // measured: letting hipcc mix the block loads / stores into the column
    // the pivot chain shares its SIMD with another wave of the workgroup (trailing-tile MFMAs, LDS traffic): it issues first
    __builtin_amdgcn_s_setprio(3);
    RsqPipe rp;
    rp.d = bcast_lane(a[0], 0);
    rp.step(0);
    rp.step(1);
    rp.step(2);
    rp.step(3);
    pivot_col<0>(a, x, rp);
    pivot_col<1>(a, x, rp);
    pivot_col<2>(a, x, rp);
    pivot_col<3>(a, x, rp);
    pivot_col<4>(a, x, rp);
    pivot_col<5>(a, x, rp);
    pivot_col<6>(a, x, rp);
    pivot_col<7>(a, x, rp);
    pivot_col<8>(a, x, rp);
    pivot_col<9>(a, x, rp);
    pivot_col<10>(a, x, rp);
    pivot_col<11>(a, x, rp);
    pivot_col<12>(a, x, rp);
    pivot_col<13>(a, x, rp);
    pivot_col<14>(a, x, rp);
    pivot_col<15>(a, x, rp);
    asm volatile("" : "+v"(a[15]), "+v"(x[15]));
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
    PD_STAMP(8 + jb * 8 + 7);
    if (lane < 16) {
        // whole rows, unconditionally and 16 bytes at a time: the entries above the diagonal are dead values that nobody
        // reads as data (the column store and the packed workspace take c <= row only; the next pivot blocks read their
        // own tiles) -- the per-entry `c <= row` test cost a branch per store, 1260 cycles per pivot block
#pragma unroll
        for (int c = 0; c < 16; c += 2) {
            f64x2 w = {a[c], a[c + 1]};
            *reinterpret_cast<f64x2 *>(&S[jb * 16 + row][jb * 16 + c]) = w;
        }
        // the inverse goes out TRANSPOSED (T[l][c] = inv(L_jj)[c][l], lane l holds column l of the inverse): again 16
        // bytes per store; the readers index it as T[column][row]
#pragma unroll
        for (int c = 0; c < 16; c += 2) {
            f64x2 w = {x[c], x[c + 1]};
            *reinterpret_cast<f64x2 *>(&T[row][c]) = w;
        }
    }
}

#define PD_THREADS 512

// In-kernel side of a flag edge (EdgeSig, common.hpp) for the FIRST kernel of a factorisation: the block it factors is being
// written by a kernel of the main stream (the head columns of the K build); one thread polls the word (bounded, then an
// agent-scope acquire: common.hpp edge_poll), the workgroup follows.  The host uses this form only while the launch fits the
// CUs reserved for the panel stream (api.hip, panel_ext): every workgroup of the fused kernel holds a whole CU while it spins.
__device__ __forceinline__ void edge_wait(const unsigned *word, unsigned value, unsigned *err)
{
    if (word != nullptr) {
        if (threadIdx.x == 0) edge_poll<2, true>(word, value, err);
        __syncthreads();
    }
}
#define TP_SP 18              // pitch of the per-wave 16x16 re-layout scratch of the TRSM kernels
#define PD_WAVES (PD_THREADS / 64)

// ws store of the storing wave.  PUBLISH: agent-scope (write-through) so that workgroups of the same launch on other
// XCDs can read the value as soon as the block's flag is up (potf2_trsm_kernel); otherwise a plain store.
// The publishing form is ONE explicit store instruction per call (what hipcc emits for a relaxed agent-scope atomic store on
// gfx950: global_store_dwordx2 ... sc1): the store wave of potf2_body_la counts its stores per step in s_waitcnt vmcnt(N) below,
// and an asm statement can be neither merged nor split nor dropped by the compiler.
template <bool PUBLISH>
__device__ __forceinline__ void ws_store(double *p, double v)
{
    if (PUBLISH) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else *p = v;
}

// STAGED: the 128x128 block is already in LDS (S), written by this workgroup and followed by a barrier (the second
// diagonal block of the 256-column leaf kernel); otherwise it is staged from global memory here.
template <bool PUBLISH, bool STAGED = false>
__device__ __forceinline__ void potf2_body(double *__restrict__ A, int64_t lda, double *__restrict__ invd, int32_t *info,
                                           int64_t info_col0, unsigned *flag, unsigned flag_base)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double (*S)[PD_PITCH] = reinterpret_cast<double (*)[PD_PITCH]>(smem);
    typedef double TBuf[16][PD_TP];
    TBuf *T = reinterpret_cast<TBuf *>(smem + PD_NB * PD_PITCH);      // T[jb]: inv(L_jb,jb), one buffer per pivot block

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    constexpr int NB16 = PD_NB / 16;
    // Roles.  Wave 0 runs the pivot chain.  A workgroup's waves go to the four SIMDs round-robin, so wave 4 shares wave
    // 0's SIMD -- and the double-precision vector unit of a SIMD also executes its fp64 MFMAs: a partner busy with
    // trailing tiles slows the pivot columns by a third (3440 against 2480 cycles in the stamps).  Wave 4 therefore is
    // the STORE wave (LDS reads and global stores only); waves 1-3 and 5-7 are the six tile workers (role 1..6).
    const int role = (wave == 0) ? 0 : (wave == 4) ? -1 : (wave < 4) ? wave : wave - 1;

    // ---- stage the block into LDS (waves 1..7) while wave 0 already factors the first pivot block from global ----
    if (STAGED) {
        if (wave == 0) pivot_block_16<false>(S, T[0], 0, lane, nullptr, 0, info, info_col0);
    } else if (wave == 0) {
        pivot_block_16<true>(S, T[0], 0, lane, A, lda, info, info_col0);
    } else {
        constexpr int NCH = PD_NB * PD_NB / 2;                  // 16-byte chunks
        constexpr int PER = (NCH + (PD_THREADS - 64) - 1) / (PD_THREADS - 64);
        f64x2 v[PER];
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int idx = (tid - 64) + q * (PD_THREADS - 64);
            const int r = idx / (PD_NB / 2), c2 = (idx % (PD_NB / 2)) * 2;
            if (idx < NCH) v[q] = *reinterpret_cast<const f64x2 *>(A + (int64_t)r * lda + c2);
        }
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int idx = (tid - 64) + q * (PD_THREADS - 64);
            const int r = idx / (PD_NB / 2), c2 = (idx % (PD_NB / 2)) * 2;
            if (idx < NCH && !(r < 16 && c2 < 16)) *reinterpret_cast<f64x2 *>(&S[r][c2]) = v[q];   // tile (0,0) is wave 0's
        }
    }
    PD_STAMP(0);
    __syncthreads();
    PD_STAMP(1);

    for (int jb = 0; jb < NB16; jb++) {
        PD_STAMP(8 + jb * 8 + 0);
        // (b) strip solve: X_ti = B_ti * inv(L_jj)^T for the 16-row tiles below the pivot block (waves 0..6)
        {
            const int ti = jb + 1 + role;
            if (role >= 0 && ti < NB16) {
                f64x4 acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
                double av[4], bv[4];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    av[kk] = S[ti * 16 + fr][jb * 16 + fk + 4 * kk];
                    bv[kk] = T[jb][fk + 4 * kk][fr];
                }
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], acc, 0, 0, 0);      // two chains of two
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], acc2, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) S[ti * 16 + fk + 4 * r][jb * 16 + fr] = acc[r] + acc2[r];
            }
        }
        PD_STAMP(8 + jb * 8 + 1);
        __syncthreads();
        PD_STAMP(8 + jb * 8 + 2);
        // (c) wave 0: next pivot tile update + pivot; the six workers: the other trailing tiles; store wave: column block jb of L
        //     (now final) goes to global memory, row-major and packed, together with inv(L_jj).
        if (role < 0) {
            // (stamps: written one row group at a time, each LDS read waited for before its global store, this wave
            // took 8400 cycles at jb = 0 and set the step time; now every read of a batch is issued before the first
            // store, the packed workspace -- what the TRSM consumers wait for -- goes first, and the row-major copy moves
            // 16 bytes per lane)
            double *ip = invd + jb * 256, *lp = invd + GPT_WS_LOFF;
            {
                double tv[4];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) tv[kk] = T[jb][fk + 4 * kk][fr];             // packed element (fr, fk + 4kk)
#pragma unroll
                for (int kk = 0; kk < 4; kk++) ws_store<PUBLISH>(ip + kk * 64 + lane, tv[kk]);
            }
            for (int j0 = jb + 1; j0 < NB16; j0 += 3) {                                    // packed blocks (j, jb), j > jb
                double pv[3][4];                 // (three blocks per batch: more live values made hipcc spill the pivot's registers)
#pragma unroll
                for (int jj = 0; jj < 3; jj++)
                    if (j0 + jj < NB16) {
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) pv[jj][kk] = S[(j0 + jj) * 16 + fr][jb * 16 + fk + 4 * kk];
                    }
#pragma unroll
                for (int jj = 0; jj < 3; jj++)
                    if (j0 + jj < NB16) {
                        const int j = j0 + jj, b = j * (j - 1) / 2 + jb;
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) ws_store<PUBLISH>(lp + b * 256 + kk * 64 + lane, pv[jj][kk]);
                    }
            }
            if (PUBLISH) {
                // everything step jb of a forward substitution needs is out: inv(L_jb,jb) just now, the blocks (jb, c < jb)
                // in earlier rounds.  The packed workspace was written with agent-scope (write-through) stores by THIS
                // wave: draining them (vmcnt(0)) before the flag is the whole hand-off -- a release fence would also write
                // back every dirty line of the XCD's L2 (the row-major copies of L, 1.7-6 us per step), which the
                // consumers never read.
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(flag, flag_base + (unsigned)jb + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // not positive definite (LAPACK info = first column whose pivot was not > 0): that column's diagonal entry of
            // L is NaN (see pivot_col); looked for here, off the pivot wave
            {
                const double dg = S[jb * 16 + fr][jb * 16 + fr];
                const unsigned long long m = __ballot(!(dg > 0.0)) & 0xffffull;
                if (m != 0ull && lane == 0) atomicCAS(info, 0, (int32_t)(info_col0 + jb * 16 + __ffsll((long long)m)));
            }
            // (the row-major copy of L is written after the last step, by all eight waves: nobody reads it before the kernel
            // ends -- the TRSM consumers take the packed workspace -- and in the first steps it was this wave, not the pivot
            // chain, that the end-of-step barrier waited for: 1800-2400 cycles against 250, profiles/r02_potf2_stamps.txt)
        } else if (jb + 1 < NB16) {
            const int rem = NB16 - 1 - jb;
            const int ntile = rem * (rem + 1) / 2;
            if (wave == 0) {
                TileUpd u;                                  // the next pivot tile, then the pivot itself
                u.load(S, jb + 1, jb + 1, jb, fr, fk);
                u.mma();
                u.store(S, fr, fk);
            } else {
                // waves 1..6: tiles t = wave, wave + 6, ...; two tiles in flight (independent MFMA chains and LDS
                // round trips: the trailing tiles of the first steps, not the pivot, used to set the step time)
                constexpr int STEP = PD_WAVES - 2;
                int it_ = 0;
                (void)it_;
                if (jb == 0) PD_STAMP1(80);
                for (int t = role; t < ntile; t += 2 * STEP) {
                    int a0, b0, a1 = 0, b1 = 0;
                    tri_decode(t, a0, b0);
                    const bool two = t + STEP < ntile;
                    if (two) tri_decode(t + STEP, a1, b1);
                    TileUpd u0, u1;
                    u0.load(S, jb + 1 + a0, jb + 1 + b0, jb, fr, fk);
                    if (two) u1.load(S, jb + 1 + a1, jb + 1 + b1, jb, fr, fk);
                    u0.mma();
                    if (two) u1.mma();
                    u0.store(S, fr, fk);
                    if (two) u1.store(S, fr, fk);
                    if (jb == 0) PD_STAMP1(81 + it_);
                    it_++;
                }
            }
            PD_STAMP(8 + jb * 8 + 3);
            if (wave == 0) pivot_block_16<false>(S, T[jb + 1], jb + 1, lane, nullptr, 0, info, info_col0);
            PD_STAMP(8 + jb * 8 + 4);
        }
        if (jb == 0) PD_STAMPW(86 + wave);
        __syncthreads();
        PD_STAMP(8 + jb * 8 + 5);
    }
    // L (lower triangle of S), row-major, 16 bytes per thread and store
    {
        const bool vec = (((uintptr_t)A & 15) == 0) && ((lda & 1) == 0);
        constexpr int NCH2 = PD_NB * PD_NB / 2;
        for (int idx = tid; idx < NCH2; idx += PD_THREADS) {
            const int r = idx / (PD_NB / 2), c2 = (idx % (PD_NB / 2)) * 2;
            if (c2 > r) continue;
            const f64x2 w = *reinterpret_cast<const f64x2 *>(&S[r][c2]);
            double *dst = A + (int64_t)r * lda + c2;
            if (c2 + 1 <= r) {
                if (vec) *reinterpret_cast<f64x2 *>(dst) = w;
                else {
                    dst[0] = w[0];
                    dst[1] = w[1];
                }
            } else {
                dst[0] = w[0];
            }
        }
    }
    PD_STAMP(2);
    PD_STAMP_DUMP();
}

// ------------------------------------------------------------------------------------------------
// potf2_body_la: the same 128x128 block, "look-ahead" form (round 4).  potf2_body above is one lock-step loop: strip solve
// (all waves) | barrier | tile updates + pivot block (wave 0) | barrier, 5200 cycles per 16 columns of which the pivot
// recurrence is 2500 (profiles/r04_potf2_stamps.txt).  Here the waves are decoupled (flags in LDS, no workgroup barrier
// inside the loop) and the chain wave does nothing but the chain:
//   wave 0, CHAIN: the 16-column recurrence of the pivot tile ALONE (no inverse riding along: 120 instead of 240 DPP
//           multiply-adds) and publishes every finished column (16 entries + 1/sqrt(pivot)) to LDS as it goes; then takes
//           the next pivot tile: (jb+1, jb+1) -= X X^T with X = tile (jb+1, jb) as soon as wave 1 has stored it (4 MFMAs).
//   waves 1, 2, RIDE-ALONG: 64 DIFFERENT rows each (lane = row: the up to seven 16-row tiles below the pivot tile, and the
//           sixteen rows of the identity) follow the chain wave's columns a few hundred cycles behind with the same
//           recurrence  v[C] -= L[C][J] v[J],  v[J] /= L[J][J]  (DPP broadcast out of the published column): when the pivot
//           tile is done, so are the forward substitution of the whole strip -- no inverse, no strip phase, no barrier -- and
//           inv(L_jj) (the identity rows; what the packed workspace / the TRSM consumers want).
//   waves 3, 5, 6, 7, WORKERS: trailing tiles (a, b) -= X_a X_b^T; every tile has ONE owner for the whole kernel, so
//           a tile's successive updates need no synchronisation; per step each worker takes the tiles the next step's
//           chain / ride-along waves read FIRST (column jb+1 and tile (jb+2, jb+2)) and counts them in cF[jb].
//   wave 4, STORE: as before (packed workspace, flag for the consumers of the fused kernel, info), paced by flags.
// LDS is processed in order per wave and is one memory for the workgroup: "data stores, then the flag store" by the writer
// and "flag load, then data loads" by the reader need no fence, only that the compiler keeps the order (volatile /
// asm memory clobbers).  All flags only grow; nothing waits on a value a NaN could change (a block that is not positive
// definite runs to the end and leaves NaN on its diagonal, found by the store wave as before).
// ------------------------------------------------------------------------------------------------
struct PdSync {
    int colflag;        // chain: columns published so far (16 jb + J + 1)
    int lflag;          // chain: L_jj rows of step jb are in S (jb + 1)
    int x1flag, x2flag; // ride-along waves: strips (and inverse) of step jb stored (jb + 1)
    int cstage;         // staging waves done with the columns from 32 on (5)
    int cstage0;        // ... with the columns 0..31 (5)
    int pad_[2];
    int cF[8];          // workers: priority tiles of trailing step jb done
};
#define PD_LA_COLBUF (2 * 16 * 16)
#define PD_LA_INVBUF (2 * 16)
#define PD_LA_SMEM_BYTES ((size_t)(PD_NB * PD_PITCH + 8 * 16 * PD_TP + PD_LA_COLBUF + PD_LA_INVBUF) * sizeof(double) + sizeof(PdSync))

// (plain LDS accesses between compiler barriers, NOT volatile: a volatile access loses its address space here -- flat
// instructions with sc0 sc1 and a vmcnt(0) wait behind each, 5500 instead of 1700 cycles per pivot block)
__device__ __forceinline__ int lds_peek(const int *p)
{
    asm volatile("" ::: "memory");
    const int v = *p;
    asm volatile("" ::: "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
template <bool SLEEP>
__device__ __forceinline__ void lds_wait_ge(const int *p, int target)
{
    while (lds_peek(p) < target) {
        if (SLEEP) __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void lds_post(int *p, int value)
{
    asm volatile("" ::: "memory");
    *p = value;
    asm volatile("" ::: "memory");
}
// tiles (a, s + 2), a = s + 2 .. 7: the workers' last step on them is s (cF[s] counts them)
__device__ __forceinline__ int pd_prio_count(int s) { return (s <= 5) ? 6 - s : 0; }

// ---- the chain recurrence, scheduled by hand ----
// Numbers (scratch/dpp_rate.hip): a double-precision VALU instruction issues in ~4.5 cycles and its result is there after 8; v_rsq_f64
// 20; a DPP read of a register a VALU instruction wrote needs two wait states (16 with the s_nop for a dependent v_mov_dpp).
// One column's dependent chain is  scale (8) -> first update (8) -> broadcast of the new pivot (16) -> rsq (20) -> t (8) -> u (8)
// -> 1/sqrt (8) = ~76 cycles; everything else (the column's other 14 - J updates, 4.8 cycles each) fits in its shadows IF it is
// placed there -- left to the compiler the updates pile up between two chain steps and a block takes 2550 cycles (160 per
// column, measured alone on a CU: scratch/chain_probe.hip).  Here every instruction of the recurrence is a volatile asm
// statement (the compiler keeps their order) and slot J = the chain steps of column J with the updates of column J - 1 spread
// over its gaps.  The new pivot is read AFTER the first update (lane J+1 of a[J+1] then IS a[J+1][J+1] - l^2, the same fused
// operation as before), which saves the two broadcasts and the explicit fma.
// (chain_slots.inc, generated by scratch/gen_chain_slots.py: per column two asm blocks with the instructions in exactly this order
// -- A: scale, two updates of the column before, the first update of this column; B: the new pivot, rsq + Newton step with the other
// updates of the column before in the gaps.  One statement per instruction left room for hipcc's conservative hazard s_nops: 9 of a
// column's 33 instructions.)
template <int J> __device__ __forceinline__ void chain_slot_a(double (&a)[16], double &inv);
template <int J> __device__ __forceinline__ void chain_slot_b(double (&a)[16], double &inv);
#include "chain_slots.inc"
template <int J>
__device__ __forceinline__ void chain_slot(double (&a)[16], double &inv, double &inv_prev, double *colb, double *invb, int *colflag,
                                           int flag0, int i)
{
    chain_slot_a<J>(a, inv);
    // two columns per publication (an LDS store costs ~4 cycles of issue per register it reads: three stores per column were
    // 50 of a column's ~170 cycles): after an odd column J the pair (J - 1, J) goes out as one 16-byte store per lane, the
    // pair of 1/sqrt(pivot) as another, then the count
    if constexpr (J & 1) {
        f64x2 cw = {a[J - 1], a[J]}, iw = {inv_prev, inv};
        *reinterpret_cast<f64x2 *>(colb + (J >> 1) * 32 + i * 2) = cw;
        *reinterpret_cast<f64x2 *>(invb + (J >> 1) * 2) = iw;
        asm volatile("" ::: "memory");
        *colflag = flag0 + J + 1;
        asm volatile("" ::: "memory");
    } else {
        inv_prev = inv;
    }
    chain_slot_b<J>(a, inv);
}
__device__ __forceinline__ void chain_block(double (&a)[16], double *colb, double *invb, int *colflag, int flag0, int i)
{
    __builtin_amdgcn_sched_barrier(0);
    double inv, inv_prev = 0.0;
    {
        RsqPipe rp;
        rp.d = bcast_lane(a[0], 0);
        rp.step(0);
        rp.step(1);
        rp.step(2);
        rp.step(3);
        inv = rp.inv;
    }
    chain_slot<0>(a, inv, inv_prev, colb, invb, colflag, flag0, i);
    chain_slot<1>(a, inv, inv_prev, colb, invb, colflag, flag0, i);
    chain_slot<2>(a, inv, inv_prev, colb, invb, colflag, flag0, i);
    chain_slot<3>(a, inv, inv_prev, colb, invb, colflag, flag0, i);
    chain_slot<4>(a, inv, inv_prev, colb, invb, colflag, flag0, i);
    chain_slot<5>(a, inv, inv_prev, colb, invb, colflag, flag0, i);
    chain_slot<6>(a, inv, inv_prev, colb, invb, colflag, flag0, i);
    chain_slot<7>(a, inv, inv_prev, colb, invb, colflag, flag0, i);
    chain_slot<8>(a, inv, inv_prev, colb, invb, colflag, flag0, i);
    chain_slot<9>(a, inv, inv_prev, colb, invb, colflag, flag0, i);
    chain_slot<10>(a, inv, inv_prev, colb, invb, colflag, flag0, i);
    chain_slot<11>(a, inv, inv_prev, colb, invb, colflag, flag0, i);
    chain_slot<12>(a, inv, inv_prev, colb, invb, colflag, flag0, i);
    chain_slot<13>(a, inv, inv_prev, colb, invb, colflag, flag0, i);
    chain_slot<14>(a, inv, inv_prev, colb, invb, colflag, flag0, i);
    chain_slot<15>(a, inv, inv_prev, colb, invb, colflag, flag0, i);
    asm volatile("" : "+v"(a[15]));
    __builtin_amdgcn_sched_barrier(0);
}

// what a ride-along wave reads per PAIR of columns: the count first, then the data (returned in that order)
struct ColIn {
    int flag;
    f64x2 col, inv;
};
__device__ __forceinline__ void col_read(ColIn &ci, const int *colflag, const double *colb, const double *invb, int P, int i)
{
    asm volatile("" ::: "memory");
    ci.flag = *colflag;
    asm volatile("" ::: "memory");
    ci.col = *reinterpret_cast<const f64x2 *>(colb + P * 32 + i * 2);
    ci.inv = *reinterpret_cast<const f64x2 *>(invb + P * 2);
    asm volatile("" ::: "memory");
}
template <int J, int C>
__device__ __forceinline__ void ride_group(double (&v)[16], double col)
{
    if constexpr (C < 16) {
        fnmac_bcast<C>(v[C], col, v[J]);                   // v[row][C] -= L[C][J] * v[row][J]
        ride_group<J, C + 1>(v, col);
    }
}
template <int P>
__device__ __forceinline__ void ride_pair(double (&v)[16], ColIn &cur, ColIn &pre, const int *colflag, const double *colb,
                                          const double *invb, int flag0, int i)
{
    // `cur` was requested one pair ago (speculatively: re-read until its count covers column 2 P + 1); the next pair is
    // requested into `pre` before the arithmetic of this one
    while (__builtin_amdgcn_readfirstlane(cur.flag) < flag0 + 2 * P + 2) col_read(cur, colflag, colb, invb, P, i);
    if constexpr (P + 1 < 8) col_read(pre, colflag, colb, invb, P + 1, i);
    const double c0 = cur.col[0], c1 = cur.col[1];
    v[2 * P] *= cur.inv[0];
    ride_group<2 * P, 2 * P + 1>(v, c0);
    v[2 * P + 1] *= cur.inv[1];
    ride_group<2 * P + 1, 2 * P + 2>(v, c1);
}
// (mid(): called after column 7 -- the columns 0..7 of the rows are final then and can be stored while the ride goes on)
template <class F>
__device__ __forceinline__ void ride_block(double (&v)[16], const int *colflag, const double *colb, const double *invb,
                                           int flag0, int i, F mid)
{
    ColIn c0, c1;
    col_read(c0, colflag, colb, invb, 0, i);
    ride_pair<0>(v, c0, c1, colflag, colb, invb, flag0, i);
    ride_pair<1>(v, c1, c0, colflag, colb, invb, flag0, i);
    ride_pair<2>(v, c0, c1, colflag, colb, invb, flag0, i);
    ride_pair<3>(v, c1, c0, colflag, colb, invb, flag0, i);
    mid();
    ride_pair<4>(v, c0, c1, colflag, colb, invb, flag0, i);
    ride_pair<5>(v, c1, c0, colflag, colb, invb, flag0, i);
    ride_pair<6>(v, c0, c1, colflag, colb, invb, flag0, i);
    ride_pair<7>(v, c1, c0, colflag, colb, invb, flag0, i);
}

// Row-major copy of the columns [c_lo, c_hi) of L (lower triangle of S) to global memory by ONE wave, 16 bytes per lane and store
// (c_lo, c_hi multiples of 16).  Called by waves that have run out of work, for column blocks that are final, while the chain goes on.
__device__ __forceinline__ void pd_copy_out_cols(double (*S)[PD_PITCH], double *__restrict__ A, int64_t lda, int c_lo, int c_hi, int lane)
{
    const bool vec = (((uintptr_t)A & 15) == 0) && ((lda & 1) == 0);
    const int cw = (c_hi - c_lo) / 2;                            // 16-byte chunks per row
    for (int idx = lane; idx < (PD_NB - c_lo) * cw; idx += 64) {
        const int r = c_lo + idx / cw, c2 = c_lo + (idx % cw) * 2;
        if (c2 > r) continue;
        const f64x2 w = *reinterpret_cast<const f64x2 *>(&S[r][c2]);
        double *dst = A + (int64_t)r * lda + c2;
        if (c2 + 1 <= r) {
            if (vec) *reinterpret_cast<f64x2 *>(dst) = w;
            else {
                dst[0] = w[0];
                dst[1] = w[1];
            }
        } else {
            dst[0] = w[0];
        }
    }
}

template <bool PUBLISH>
__device__ __forceinline__ void potf2_body_la(double *__restrict__ A, int64_t lda, double *__restrict__ invd, int32_t *info,
                                              int64_t info_col0, unsigned *flag, unsigned flag_base)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double (*S)[PD_PITCH] = reinterpret_cast<double (*)[PD_PITCH]>(smem);
    typedef double TBuf[16][PD_TP];
    TBuf *T = reinterpret_cast<TBuf *>(smem + PD_NB * PD_PITCH);
    double *colbuf = smem + PD_NB * PD_PITCH + 8 * 16 * PD_TP;      // [parity][column][row]
    double *invbuf = colbuf + PD_LA_COLBUF;                          // [parity][column]
    PdSync *sy = reinterpret_cast<PdSync *>(invbuf + PD_LA_INVBUF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    constexpr int NB16 = PD_NB / 16;

    if (tid < (int)(sizeof(PdSync) / sizeof(int))) reinterpret_cast<int *>(sy)[tid] = 0;
    __syncthreads();
    PD_TRACE_DECL;
    PD_TRACE(1);

    if (wave == 0) {
        // ================================ chain ================================
        __builtin_amdgcn_s_setprio(3);
        double a[16];
        f64x4 acc;
#pragma unroll
        for (int c = 0; c < 16; c++) a[c] = A[(int64_t)fr * lda + c];
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = A[(int64_t)(16 + fk + 4 * r) * lda + 16 + fr];      // tile (1, 1)
        PD_STAMP(0);
        for (int jb = 0; jb < NB16; jb++) {
            PD_STAMP(8 + jb * 8 + 0);
            PD_TRACE(100 + jb);
            PU_STAMP(1 + 2 * jb);
            chain_block(a, colbuf + (jb & 1) * 256, invbuf + (jb & 1) * 16, &sy->colflag, jb * 16, fr);
            PU_STAMP(2 + 2 * jb);
            PD_STAMP(8 + jb * 8 + 1);
            PD_TRACE(110 + jb);
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; c += 2) {
                    f64x2 w = {a[c], a[c + 1]};
                    *reinterpret_cast<f64x2 *>(&S[jb * 16 + fr][jb * 16 + c]) = w;
                }
            }
            lds_post(&sy->lflag, jb + 1);
            if (jb + 1 == NB16) break;
            // the next pivot tile: its updates of the steps before this one (its owner's), then this step's
            const int t1 = (jb + 1) * 16;
            if (jb > 0) {
                lds_wait_ge<false>(&sy->cF[jb - 1], pd_prio_count(jb - 1));
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r] = S[t1 + fk + 4 * r][t1 + fr];
            }
            PD_STAMP(8 + jb * 8 + 2);
            PD_TRACE(120 + jb);
            f64x4 acc2 = {0.0, 0.0, 0.0, 0.0};
            lds_wait_ge<false>((jb + 1 <= 4) ? &sy->x1flag : &sy->x2flag, jb + 1);
            PD_STAMP(8 + jb * 8 + 3);
            PD_TRACE(130 + jb);
            double xv[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) xv[kk] = S[t1 + fr][jb * 16 + fk + 4 * kk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xv[0], xv[0], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xv[1], xv[1], acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xv[2], xv[2], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xv[3], xv[3], acc2, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) S[t1 + fk + 4 * r][t1 + fr] = acc[r] + acc2[r];
            PD_STAMP(8 + jb * 8 + 4);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int c = 0; c < 16; c += 2) {
                const f64x2 w = *reinterpret_cast<const f64x2 *>(&S[t1 + fr][t1 + c]);
                a[c] = w[0];
                a[c + 1] = w[1];
            }
            PD_STAMP(8 + jb * 8 + 5);
        }
        __builtin_amdgcn_s_setprio(0);
    } else if (wave == 1 || wave == 2) {
        // ================================ ride-along ================================
        // Fixed rows: wave 1 the tiles 1..4, wave 2 the tiles 5..7 and the identity (quarter 3).  After the ride of step jb
        // the wave stores its strips, applies step jb to ITS tiles of column jb+1 (what it rides next; the tiles' earlier
        // steps are the workers': cF[jb-1]) and takes their rows back -- no other wave between two rides.
        __builtin_amdgcn_s_setprio(2);
        int *myflag = (wave == 1) ? &sy->x1flag : &sy->x2flag;
        const int tq = (wave == 1) ? 1 + fk : 5 + fk;                   // this lane's tile (8: the identity)
        const bool ident = (tq == NB16);
        const int tfirst = (wave == 1) ? 1 : 5, tlast = (wave == 1) ? 4 : 7;
        const int row = (ident ? NB16 - 1 : tq) * 16 + fr;
        // the first two column blocks are staged first (coalesced; lane = row straight from global memory is 64 cache lines
        // per load instruction: the rows arrived 2000 cycles after the chain wave's tile)
        lds_wait_ge<false>(&sy->cstage0, PD_WAVES - 3);
        double v[16];
#pragma unroll
        for (int c = 0; c < 16; c += 2) {
            const f64x2 w = *reinterpret_cast<const f64x2 *>(&S[row][c]);
            v[c] = ident ? ((c == fr) ? 1.0 : 0.0) : w[0];
            v[c + 1] = ident ? ((c + 1 == fr) ? 1.0 : 0.0) : w[1];
        }
        // the wave's three tiles of the NEXT column (what it rides in the next step), C layout, once the workers are through
        // with them (cF; before the first ride: as staged)
        const int tu0 = (wave == 1) ? 2 : 5;                            // tiles tu0 .. tu0 + 2
        f64x4 accn[3];
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int r = 0; r < 4; r++) accn[q][r] = S[(tu0 + q) * 16 + fk + 4 * r][16 + fr];
        for (int jb = 0; jb < NB16; jb++) {
            const bool valid = !ident && tq > jb;
            const bool any = (wave == 2) || (jb < tlast);
            const bool upd = (jb + 1 < NB16) && (tu0 + 2 >= jb + 2);   // (some tile of this wave lies below tile jb + 1)
            const int t1 = (jb + 1) * 16;
            if (any) {
                PD_TRACE(200 + jb);
                PD_STAMPW((wave == 1 ? 80 : 104) + jb);
                // (the inverse goes out TRANSPOSED: T[l][c] = inv(L_jj)[c][l]; lane l holds column l of the inverse)
                double *dst = ident ? &T[jb][fr][0] : &S[row][jb * 16];
                const bool wr = valid || ident;
                ride_block(v, &sy->colflag, colbuf + (jb & 1) * 256, invbuf + (jb & 1) * 16, jb * 16, fr, [&]() {
                    if (wr) {
#pragma unroll
                        for (int c = 0; c < 8; c += 2) {
                            f64x2 w = {v[c], v[c + 1]};
                            *reinterpret_cast<f64x2 *>(dst + c) = w;
                        }
                    }
                });
                PD_STAMPW((wave == 1 ? 88 : 112) + jb);
                PD_TRACE(210 + jb);
                if (wr) {
#pragma unroll
                    for (int c = 8; c < 16; c += 2) {
                        f64x2 w = {v[c], v[c + 1]};
                        *reinterpret_cast<f64x2 *>(dst + c) = w;
                    }
                }
                PD_STAMPW((wave == 1 ? 72 : 96) + jb);
            }
            lds_post(myflag, jb + 1);
            PD_TRACE(220 + jb);
            // ---- this wave's tiles of column jb + 1 take step jb (all three slots computed, the ones above tile jb + 2
            // not stored); then their rows come back for the next ride
            if (upd) {
                // X(jb+1, jb) is the other wave's when tile jb+1 is (wave 1 owns 1..4)
                if ((jb + 1 <= 4) != (wave == 1)) lds_wait_ge<false>((wave == 1) ? &sy->x2flag : &sy->x1flag, jb + 1);
                if (jb > 0) {
                    lds_wait_ge<false>(&sy->cF[jb - 1], pd_prio_count(jb - 1));
#pragma unroll
                    for (int q = 0; q < 3; q++)
#pragma unroll
                        for (int r = 0; r < 4; r++) accn[q][r] = S[(tu0 + q) * 16 + fk + 4 * r][t1 + fr];
                }
                double bv[4], av[3][4];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) bv[kk] = S[t1 + fr][jb * 16 + fk + 4 * kk];
#pragma unroll
                for (int q = 0; q < 3; q++)
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) av[q][kk] = -S[(tu0 + q) * 16 + fr][jb * 16 + fk + 4 * kk];
                f64x4 c1[3];
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    c1[q] = f64x4{0.0, 0.0, 0.0, 0.0};
                    accn[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q][0], bv[0], accn[q], 0, 0, 0);
                    c1[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q][1], bv[1], c1[q], 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    accn[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q][2], bv[2], accn[q], 0, 0, 0);
                    c1[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q][3], bv[3], c1[q], 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    if (tu0 + q >= jb + 2) {
#pragma unroll
                        for (int r = 0; r < 4; r++) S[(tu0 + q) * 16 + fk + 4 * r][t1 + fr] = accn[q][r] + c1[q][r];
                    }
                }
                asm volatile("" ::: "memory");
                const bool vnext = !ident && tq > jb + 1;
#pragma unroll
                for (int c = 0; c < 16; c += 2) {
                    const f64x2 w = *reinterpret_cast<const f64x2 *>(&S[row][t1 + c]);
                    v[c] = ident ? ((c == fr) ? 1.0 : 0.0) : (vnext ? w[0] : 0.0);
                    v[c + 1] = ident ? ((c + 1 == fr) ? 1.0 : 0.0) : (vnext ? w[1] : 0.0);
                }
            } else if (wave == 2) {
#pragma unroll
                for (int c = 0; c < 16; c++) v[c] = (ident && c == fr) ? 1.0 : 0.0;
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (wave == 1) {
            // wave 1 has no rows left after step 3: it writes the row-major copy of the columns 0..63 of L (final once the strips and
            // the diagonal tiles of the steps 0..3 are stored) while the steps 4..7 run (the copy-out of the whole block after the
            // loop was 2400 of the kernel's 46 000 cycles); waves 3 and 7 take the columns 64..111 when their tiles are done
            lds_wait_ge<true>(&sy->x2flag, 4);
            lds_wait_ge<true>(&sy->lflag, 4);
            pd_copy_out_cols(S, A, lda, 0, 64, lane);
        }
    } else {
        // ---- stage the columns from 32 on into LDS (waves 3..7): the first two column blocks are read from global memory by
        // the waves that use them (chain: tiles (0, 0), (1, 1); ride-along: rows of column block 0, tiles (t, 1)) ----
        {
            // phase 1: columns 0..31 of all rows (what the ride-along waves start from), without the chain wave's tiles (0, 0)
            // and (1, 1); phase 2: the columns from 32 on of the rows from 32 on (the workers' tiles)
            constexpr int NT = PD_THREADS - 192;
            constexpr int NCH1 = PD_NB * 16, PER1 = (NCH1 + NT - 1) / NT;            // 16-byte chunks
            constexpr int CW = (PD_NB - 32) / 2;                                      // 16-byte chunks per row of phase 2
            constexpr int NCH2 = (PD_NB - 32) * CW, PER2 = (NCH2 + NT - 1) / NT;
            // (phase 2's loads only behind phase 1's LDS writes: issued together with phase 1's -- one memory round trip less for the
            // workers' tiles -- the ride-along waves get their columns later: N = 8192 4.316 -> 4.358 ms, N = 4096 1.151 -> 1.168, same box)
            {
                f64x2 v1[PER1];
#pragma unroll
                for (int q = 0; q < PER1; q++) {
                    const int idx = (tid - 192) + q * NT;
                    const int r = idx / 16, c2 = (idx % 16) * 2;
                    if (idx < NCH1) v1[q] = *reinterpret_cast<const f64x2 *>(A + (int64_t)r * lda + c2);
                }
#pragma unroll
                for (int q = 0; q < PER1; q++) {
                    const int idx = (tid - 192) + q * NT;
                    const int r = idx / 16, c2 = (idx % 16) * 2;
                    if (idx < NCH1 && r >= 16 && !(r < 32 && c2 >= 16)) *reinterpret_cast<f64x2 *>(&S[r][c2]) = v1[q];
                }
                asm volatile("" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(&sy->cstage0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("" ::: "memory");
            }
            f64x2 v2[PER2];
#pragma unroll
            for (int q = 0; q < PER2; q++) {
                const int idx = (tid - 192) + q * NT;
                const int r = 32 + idx / CW, c2 = 32 + (idx % CW) * 2;
                if (idx < NCH2) v2[q] = *reinterpret_cast<const f64x2 *>(A + (int64_t)r * lda + c2);
            }
#pragma unroll
            for (int q = 0; q < PER2; q++) {
                const int idx = (tid - 192) + q * NT;
                const int r = 32 + idx / CW, c2 = 32 + (idx % CW) * 2;
                if (idx < NCH2) *reinterpret_cast<f64x2 *>(&S[r][c2]) = v2[q];
            }
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&sy->cstage, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
        }
        if (wave == 4) {
            // ================================ store ================================
            for (int jb = 0; jb < NB16; jb++) {
                lds_wait_ge<true>(&sy->lflag, jb + 1);
                lds_wait_ge<true>(&sy->x1flag, jb + 1);
                lds_wait_ge<true>(&sy->x2flag, jb + 1);
                PD_TRACE(300 + jb);
                PU_STAMP(1 + 2 * jb);
                double *ip = invd + jb * 256, *lp = invd + GPT_WS_LOFF;
                {
                    double tv[4];
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) tv[kk] = T[jb][fk + 4 * kk][fr];             // packed element (fr, fk + 4kk)
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) ws_store<PUBLISH>(ip + kk * 64 + lane, tv[kk]);
                }
                for (int j0 = jb + 1; j0 < NB16; j0 += 3) {                                    // packed blocks (j, jb), j > jb
                    double pv[3][4];
#pragma unroll
                    for (int jj = 0; jj < 3; jj++)
                        if (j0 + jj < NB16) {
#pragma unroll
                            for (int kk = 0; kk < 4; kk++) pv[jj][kk] = S[(j0 + jj) * 16 + fr][jb * 16 + fk + 4 * kk];
                        }
#pragma unroll
                    for (int jj = 0; jj < 3; jj++)
                        if (j0 + jj < NB16) {
                            const int j = j0 + jj, b = j * (j - 1) / 2 + jb;
#pragma unroll
                            for (int kk = 0; kk < 4; kk++) ws_store<PUBLISH>(lp + b * 256 + kk * 64 + lane, pv[jj][kk]);
                        }
                }
                if (PUBLISH) {
                    // Write-through stores of THIS wave, acknowledged, then the flag -- PIPELINED (round 5): a write-through store
                    // is acknowledged ~2.5 us after its issue, longer than a step of the chain (2 us), and draining every step
                    // before the next one's stores were issued made the PUBLICATION the pace of every fused leaf (flags 3.06 us
                    // apart, the last one 7 us behind the chain; profiles/r05_upd_stamps.txt).  Stores complete in issue order
                    // (vmcnt), so once step jb's 32 - 4 jb store instructions are issued, vmcnt <= 32 - 4 jb means that everything
                    // older -- step jb - 1 -- is in memory: its flag goes up then, and the last step drains.
                    // (the count: 4 stores of inv(L_jb,jb) + 4 per packed block (j, jb), j = jb + 1 .. 7 -- ws_store<true> is one
                    // instruction by construction)
                    static_assert(NB16 == 8 && PD_NB == 128, "the vmcnt counts below are 4 * (8 - jb) stores per step of a 128-column block");
                    switch (jb) {
                    case 0: break;
                    case 1: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
                    case 2: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
                    case 3: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
                    case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
                    case 5: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
                    case 6: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                    }
                    if (jb > 0 && lane == 0) __hip_atomic_store(flag, flag_base + (unsigned)jb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    PU_STAMP(2 + 2 * jb);
                    if (jb + 1 == NB16) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (lane == 0) __hip_atomic_store(flag, flag_base + (unsigned)NB16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                {
                    const double dg = S[jb * 16 + fr][jb * 16 + fr];
                    const unsigned long long m = __ballot(!(dg > 0.0)) & 0xffffull;
                    if (m != 0ull && lane == 0) atomicCAS(info, 0, (int32_t)(info_col0 + jb * 16 + __ffsll((long long)m)));
                }
            }
        } else {
            // ================================ workers ================================
            // Unit of work: (column b >= 2, step s <= b - 2) = the worker's tiles (a, b), a >= b, take  -= X(a, s) X(b, s)^T.
            // (Step b - 1 of column b is taken by the waves that use it next: the ride-along waves, the chain wave for (b, b).)
            // A tile's steps go in order (one owner, program order): step-major, within a step the lowest column first.
            // cF[s] counts the tiles of column s + 2 that have taken step s, their last one here.
            // Ownership: column 4 is waves 3 and 7's (one SIMD), the columns 5..7 -- late deadlines -- waves 5 and 6's: those share
            // their SIMDs with the ride-along waves, and on gfx950
            // a SIMD's fp64 MFMAs run at 32 flop per cycle (64 cycles per 16x16x4): a busy partner doubled the time of the
            // ride-along waves' own tile updates between two rides (2000 against 1100 cycles), which are on the chain.
            // (columns 2 and 3 -- the first deadlines, 16 tile-steps in the first two steps -- go four ways all the same: one SIMD
            // cannot take the 36 tile updates of the first two steps; waves 3 and 7 with the columns 2..5: step 2 waits 4800 cycles)
            const int me2 = (wave == 3 || wave == 5) ? 0 : 1;       // columns >= 4: tiles (a, b) with (a + b) & 1 == me2
            const int me4 = (wave == 3) ? 0 : wave - 4;             // columns 2, 3: (a + b) & 3 == me4 (waves 3, 5, 6, 7 -> 0 .. 3)
            const int blo = (wave == 3 || wave == 7) ? 4 : 5, bhi = (wave == 3 || wave == 7) ? 4 : 7;
            lds_wait_ge<true>(&sy->cstage, PD_WAVES - 3);
            for (int st = 0; st + 2 < NB16; st++) {                 // step-major, within a step the lowest column first
                if (st + 2 > bhi) break;
                lds_wait_ge<true>(&sy->x1flag, st + 1);
                lds_wait_ge<true>(&sy->x2flag, st + 1);
                for (int b = st + 2; b <= bhi; b++) {
                    if (b > 3 && b < blo) continue;
                    const bool four = b <= 3;
                    int cnt = 0;
                    const int astep = four ? 8 : 4, apair = four ? 4 : 2;
                    for (int a0 = four ? b + ((me4 - 2 * b) & 3) : b + ((me2 + b + b) & 1); a0 < NB16; a0 += astep) {
                        const int a1 = a0 + apair;
                        PD_TRACE(1000 + b * 10 + st);
                        const bool two = a1 < NB16;
                        TileUpd u0, u1;
                        u0.load(S, a0, b, st, fr, fk);
                        if (two) u1.load(S, a1, b, st, fr, fk);
                        u0.mma();
                        if (two) u1.mma();
                        u0.store(S, fr, fk);
                        if (two) u1.store(S, fr, fk);
                        cnt += two ? 2 : 1;
                        PD_TRACE(2000 + b * 10 + st);
                    }
                    if (st == b - 2 && cnt) {
                        asm volatile("" ::: "memory");
                        if (lane == 0) __hip_atomic_fetch_add(&sy->cF[st], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        asm volatile("" ::: "memory");
                    }
                }
            }
            if (wave == 3 || wave == 7) {                           // (done with their columns after step 2)
                const int done = (wave == 3) ? 6 : 7;
                lds_wait_ge<true>(&sy->x2flag, done);
                lds_wait_ge<true>(&sy->lflag, done);
                pd_copy_out_cols(S, A, lda, wave == 3 ? 64 : 96, wave == 3 ? 96 : 112, lane);
            }
        }
    }
    __syncthreads();
    // L (lower triangle of S), row-major: what is left are the columns from 112 on (0..111: waves 1, 3 and 7, above)
    {
        const bool vec = (((uintptr_t)A & 15) == 0) && ((lda & 1) == 0);
        constexpr int NCH2 = (PD_NB - 112) * 8;
        for (int idx = tid; idx < NCH2; idx += PD_THREADS) {
            const int r = 112 + idx / 8, c2 = 112 + (idx % 8) * 2;
            if (c2 > r) continue;
            const f64x2 w = *reinterpret_cast<const f64x2 *>(&S[r][c2]);
            double *dst = A + (int64_t)r * lda + c2;
            if (c2 + 1 <= r) {
                if (vec) *reinterpret_cast<f64x2 *>(dst) = w;
                else {
                    dst[0] = w[0];
                    dst[1] = w[1];
                }
            } else {
                dst[0] = w[0];
            }
        }
    }
    PD_STAMP(2);
    PD_STAMP_DUMP();
}

template <bool LA>
__global__ __launch_bounds__(PD_THREADS) void potf2_diag_kernel(double *__restrict__ A, int64_t lda,
                                                                double *__restrict__ invd, int32_t *info,
                                                                int64_t info_col0, const unsigned *wait_word,
                                                                unsigned wait_val, unsigned *wait_err, int64_t bs_a,
                                                                int64_t bs_ws)
{
    edge_wait(wait_word, wait_val, wait_err);
    // (batched: blockIdx.y = element of a batch of independent matrices, gpt_fit_batch)
    A += (int64_t)blockIdx.y * bs_a;
    invd += (int64_t)blockIdx.y * bs_ws;
    info += blockIdx.y;
    if (LA) potf2_body_la<false>(A, lda, invd, info, info_col0, nullptr, 0u);
    else potf2_body<false>(A, lda, invd, info, info_col0, nullptr, 0u);
}

// wave-level poll of a device word that only ever goes up (signed difference against `base`); returns the value seen
template <int SLEEP>
__device__ __forceinline__ unsigned pu_poll_ge(const unsigned *w, unsigned v, unsigned base, int target)
{
    while ((int)(v - base) < target) {
        __builtin_amdgcn_s_sleep(SLEEP);
        v = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    return v;
}

// ---- the forward substitution of ONE 16-row strip behind the chain's flags (the consumer waves of the fused leaf kernels) ----
// Per step j: X_j = (B_j - sum_{c<j} X_c L_jc^T) inv(L_jj)^T; right-looking: x_j goes into every later right-hand side at
// once (the left-looking form put 28 dependent MFMAs between the last two flags).  Everything a step reads -- inv(L_jj) and the
// blocks (j', j) -- is published with flag j + 1 and read with agent-scope loads, ~1.5 us each way on this chip, so the memory
// round trips are what a step costs (profiles/r05_upd_stamps.txt: 2.9-3.1 us per step against the chain's 2.1, the last x_7
// stored 8 us behind the chain):
//   * one BATCH per step: inv(L_jj), the fold's blocks and a sample of the flag word are requested together;
//   * a wave that is BEHIND the chain (the sample taken with batch j already shows flag j + 2) requests batch j + 1 before
//     the fold of step j, so that its round trip runs under the fold's MFMAs and the next step starts with its data there;
//     a wave that has caught up polls, as before.  Every wave polls for itself: the strips share nothing (no barrier).
// Same operations in the same order on every accumulator as the first form: same bits.
struct StripBatch {
    double dv[4];
    double lv[7][4];
    unsigned fn;
};
template <int J>
__device__ __forceinline__ void strip_batch_issue(StripBatch &b, const double *__restrict__ ws, const unsigned *flag, int lane)
{
    const double *lpk = ws + GPT_WS_LOFF;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) b.dv[kk] = __hip_atomic_load(ws + J * 256 + kk * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int jp = J + 1; jp < 8; jp++)
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
            b.lv[jp - J - 1][kk] = __hip_atomic_load(lpk + (jp * (jp - 1) / 2 + J) * 256 + kk * 64 + lane, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
    // (the flag sample LAST: sampled first -- readable without waiting for the fold blocks -- it sees the next step's publication less
    // often and the strips prefetch less: N = 8192 4.328 -> 4.360 ms, N = 4096 1.153 -> 1.164, same box)
    b.fn = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// X: the wave's 16 x 16 re-layout scratch in LDS.  through: write x_j through to memory (somebody reads it before this kernel ends).
struct StripCtx {
    const double *ws;
    const unsigned *flag;
    unsigned flag_base;
    double *Brow;          // B + row0 * ldb
    int64_t ldb;
    bool through;
    int lane, fr, fk;
    unsigned fl;
    bool have;
};
// The first step that may request the NEXT step's batch under its own fold.  Step 0 does not: with both batches, seven accumulators and the
// store addresses live the strip code needed 272 VGPRs (16 spilled, their reloads in every step); without it 254, no spill.  Same box,
// separate processes: N = 4096 1.163 -> 1.129 ms, N = 8192 4.302 / 4.303; from step 2 on (232 VGPRs): 1.131 -> 1.141, 4.348 -> 4.365.
#ifndef STRIP_PREFETCH_FROM
#define STRIP_PREFETCH_FROM 1
#endif
template <int J, bool THROUGH>
__device__ __forceinline__ void strip_step(StripCtx &c, StripBatch &cur, StripBatch &nxt, f64x4 (&bt)[8], f64x4 &acc,
                                           double (*X)[TP_SP])
{
    const int lane = c.lane, fr = c.fr, fk = c.fk;
    if (!c.have) {
        c.fl = pu_poll_ge<1>(c.flag, c.fl, c.flag_base, J + 1);      // (signed difference: an earlier launch's value lies below flag_base)
        strip_batch_issue<J>(cur, c.ws, c.flag, lane);
    }
    // accumulator (C layout) -> A operand through the scratch
#pragma unroll
    for (int r = 0; r < 4; r++) X[fk + 4 * r][fr] = acc[r];
    double av[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) av[kk] = X[fr][fk + 4 * kk];
    f64x4 res = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; kk++) res = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], cur.dv[kk], res, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        double *dst = c.Brow + (int64_t)(fk + 4 * r) * c.ldb + J * 16 + fr;
        // (THROUGH is a template parameter: with the two kinds of store in a run-time branch hipcc cannot count them and waits
        // vmcnt(0) -- for the acknowledgement of THESE stores, 1-2.5 us -- in front of the flag sample below, in every step)
        if (THROUGH) __hip_atomic_store(dst, res[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *dst = res[r];
        X[fk + 4 * r][fr] = res[r];
    }
    if constexpr (J + 1 < 8) {
        double xj[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) xj[kk] = -X[fr][fk + 4 * kk];
        // behind the chain?  then the next step's batch goes out now, under this step's fold
        const unsigned f = (unsigned)__builtin_amdgcn_readfirstlane((int)cur.fn);
        c.have = (J >= STRIP_PREFETCH_FROM) && (int)(f - c.flag_base) >= J + 2;
        if (c.have) strip_batch_issue<J + 1>(nxt, c.ws, c.flag, lane);
        else c.fl = f;
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
#pragma unroll
            for (int jp = J + 1; jp < 8; jp++)
                bt[jp] = __builtin_amdgcn_mfma_f64_16x16x4f64(xj[kk], cur.lv[jp - J - 1][kk], bt[jp], 0, 0, 0);
        acc = bt[J + 1];
    }
}
template <bool THROUGH>
__device__ __forceinline__ void strip_substitution_t(StripCtx &c, f64x4 (&bt)[8], double (*X)[TP_SP])
{
    StripBatch a, b;
    f64x4 acc = bt[0];
    c.have = false;
    c.fl = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(c.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    strip_step<0, THROUGH>(c, a, b, bt, acc, X);
    strip_step<1, THROUGH>(c, b, a, bt, acc, X);
    strip_step<2, THROUGH>(c, a, b, bt, acc, X);
    strip_step<3, THROUGH>(c, b, a, bt, acc, X);
    strip_step<4, THROUGH>(c, a, b, bt, acc, X);
    strip_step<5, THROUGH>(c, b, a, bt, acc, X);
    strip_step<6, THROUGH>(c, a, b, bt, acc, X);
    strip_step<7, THROUGH>(c, b, a, bt, acc, X);
}
__device__ __forceinline__ void strip_substitution(StripCtx &c, f64x4 (&bt)[8], double (*X)[TP_SP])
{
    if (c.through) strip_substitution_t<true>(c, bt, X);
    else strip_substitution_t<false>(c, bt, X);
}

// ------------------------------------------------------------------------------------------------
// potf2_trsm_kernel: the diagonal block AND the TRSM of the rows below it in one launch, for the latency-bound end of
// the factorisation.  Workgroup 0 is potf2_body; it publishes the packed workspace block by block (agent-scope
// stores, release, flag = flag_base + blocks done).  Workgroups 1.. own 128 rows of the panel each (16 per wave, as
// trsm_panel_kernel): they load their rows while the pivots run, and take step j of the forward substitution as soon
// as flag >= flag_base + j + 1, reading inv(L_jj) and the blocks (j, c < j) straight into MFMA operand registers with
// agent-scope loads.  The substitution therefore trails the pivot chain by one step instead of starting after it:
// the pair costs ~28 us where potf2 (24.4) + trsm_panel (10, of which ~5 are launch + prologue) cost ~34.
// Workgroup 0 is dispatched first and waits for nobody, so the spinning consumers cannot starve it.  Every workgroup
// of the launch gets potf2's 135 KB of LDS, i.e. a CU to itself: used only while the panel is short (few workgroups,
// idle chip); above that the two plain kernels run (host side, api.hip).
// ------------------------------------------------------------------------------------------------
template <bool LA, int SWV>
__global__ __launch_bounds__(PD_THREADS) void potf2_trsm_kernel(double *__restrict__ A, int64_t lda,
                                                                double *__restrict__ invd, int32_t *info,
                                                                int64_t info_col0, int64_t m, double *__restrict__ B,
                                                                int64_t ldb, unsigned *flag, unsigned flag_base,
                                                                unsigned *edge, unsigned edge_val,
                                                                const unsigned *wait_word, unsigned wait_val,
                                                                unsigned *wait_err)
{
    edge_wait(wait_word, wait_val, wait_err);
    if (blockIdx.x == 0) {
        if (LA) potf2_body_la<true>(A, lda, invd, info, info_col0, flag, flag_base);
        else potf2_body<true>(A, lda, invd, info, info_col0, flag, flag_base);
        // (edge flag: "the panel is final" -- what the waiting update reads are the consumers' rows; this workgroup only
        // has to be counted)
        if (edge) edge_signal(edge, edge_val, gridDim.x);
        return;
    }
    // (the chain's kernels share CUs with the main stream's trailing update: raised wave priority, as in gemm.hip)
    __builtin_amdgcn_s_setprio(2);
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    constexpr int NB16 = PD_NB / 16;
    double (*X)[TP_SP] = reinterpret_cast<double (*)[TP_SP]>(smem + wave * 16 * TP_SP);
    // R64 (round 5): 64 rows per workgroup -- ONE strip wave per SIMD (waves 4..7 idle).  On gfx950 a SIMD's fp64 MFMAs are 64
    // cycles each: with two strips per SIMD the fold's 28 MFMAs of step 0 are 1.5 us and the strips finish 28 us after the
    // launch; with one they follow the chain's flags (x_7 stored 20 us after the launch, m = 4096; profiles/r05_upd_stamps.txt)
    // -- at twice the CUs, so only for the chain-bound end of a factorisation (api.hip: fuse_rows64).
    // (SWV = 2 / 1, option fuse_rows32 / fuse_rows16: 32 / 16 rows per workgroup -- a CU's vector loads are ~32 bytes of requests per
    // cycle whatever they hit, and every strip wave requests the same 5 + 4 j fragments per step)
    const int64_t row0 = ((int64_t)(blockIdx.x - 1) * SWV + wave) * 16;
    const bool active = wave < SWV && row0 < m;
    f64x4 bt[NB16];
    if (active) {
#pragma unroll
        for (int j = 0; j < NB16; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) bt[j][r] = B[(row0 + fk + 4 * r) * ldb + j * 16 + fr];
    }
    if (active) {
        StripCtx sc;
        sc.ws = invd;
        sc.flag = flag;
        sc.flag_base = flag_base;
        sc.Brow = B + row0 * ldb;
        sc.ldb = ldb;
        sc.through = edge != nullptr;      // (a launch that raises an edge flag writes its rows through to memory, see EdgeSig)
        sc.lane = lane;
        sc.fr = fr;
        sc.fk = fk;
        strip_substitution(sc, bt, X);
    }
    if (edge) edge_signal(edge, edge_val, gridDim.x);
}

// (Removed in round 6, sources in the history up to round 5's last commit: potf2_trsm_upd_kernel -- the leaf's rank-128 update of
// the next 128 / 256 columns inside the leaf's launch, bit-identical and slower, NOTES_r05.md section 1 -- and potf2x2_trsm_kernel --
// a 256-column leaf in one launch, round 2, slower than two 128-column leaves once the fused leaf kernel existed.)

// The 135 KB of dynamic LDS the diagonal-block kernels ask for needs hipFuncAttributeMaxDynamicSharedMemorySize, which
// is a property of the function ON ONE DEVICE: set once per (kernel, device), thread-safe (ll_batch and bench.py drive
// two contexts from two host threads; a process may hold contexts on several GPUs).
#include <mutex>
#include <cstdlib>
// GPT_POTF2_LA=0 selects the lock-step body (potf2_body) for the 128-column kernels; the look-ahead body is the default
// (same-box A/B, round 4: N = 4096 1.245 -> 1.207 ms, N = 8192 4.437 -> 4.402 ms; profiles/r04_potf2_la.txt)
static bool potf2_lookahead()
{
    static const bool on = [] { const char *e = getenv("GPT_POTF2_LA"); return e == nullptr || atoi(e) != 0; }();
    return on;
}
static int ensure_big_lds(const void *fn, int which, size_t shmem)
{
    static std::mutex mu;
    static bool done[21][64];
    int dev = 0;
    GPT_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) dev = 63;
    std::lock_guard<std::mutex> lk(mu);
    if (!done[which][dev] || dev == 63) {
        GPT_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        done[which][dev] = true;
    }
    return GPT_OK;
}

// m rows below the 128x128 diagonal block at A (B = A + 128 * lda).  `flag` is a device word only ever raised;
// flag_base must exceed every value written to it before (the caller counts: 16 per launch).
int launch_potf2_trsm(hipStream_t st, double *A, int64_t lda, double *invd, int32_t *info, int64_t info_base, int64_t m,
                      unsigned *flag, unsigned flag_base, hipEvent_t done, EdgeSig edge, EdgeSig wait, int rows64)
{
    gpt_jitter(st);
    const bool la = potf2_lookahead();
    const size_t shmem = la ? PD_LA_SMEM_BYTES : (size_t)(PD_NB * PD_PITCH + 8 * 16 * PD_TP) * sizeof(double);
    // rows64: 0 = 128 rows per consumer workgroup, 1 = 64, 2 = 32, 3 = 16
    auto kern = rows64 == 3 ? (la ? potf2_trsm_kernel<true, 1> : potf2_trsm_kernel<false, 1>)
              : rows64 == 2 ? (la ? potf2_trsm_kernel<true, 2> : potf2_trsm_kernel<false, 2>)
              : rows64 == 1 ? (la ? potf2_trsm_kernel<true, 4> : potf2_trsm_kernel<false, 4>)
                            : (la ? potf2_trsm_kernel<true, PD_WAVES> : potf2_trsm_kernel<false, PD_WAVES>);
    { int rc_ = ensure_big_lds(reinterpret_cast<const void *>(kern), (la ? 3 : 0) + (rows64 == 1 ? 9 : rows64 == 2 ? 13 : rows64 == 3 ? 17 : 0), shmem); if (rc_ != GPT_OK) return rc_; }
    const int64_t rpw = 128 >> rows64;
    const unsigned grid = 1u + (unsigned)((m + rpw - 1) / rpw);
    if (done) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(PD_THREADS), shmem, st, nullptr, done, 0, A, lda, invd,
                                    info, info_base, m, A + 128 * lda, lda, flag, flag_base, edge.word, edge.value,
                                    wait.word, wait.value, wait.err);
    else hipLaunchKernelGGL(kern, dim3(grid), dim3(PD_THREADS), shmem, st, A, lda, invd, info, info_base, m,
                            A + 128 * lda, lda, flag, flag_base, edge.word, edge.value, wait.word, wait.value, wait.err);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

int launch_potf2_diag(hipStream_t st, double *A, int64_t lda, double *invd, int32_t *info, int64_t info_base, EdgeSig wait,
                      int64_t nbatch, int64_t bstride_a, int64_t bstride_ws)
{
    gpt_jitter(st);
    const bool la = potf2_lookahead();
    const size_t shmem = la ? PD_LA_SMEM_BYTES : (size_t)(PD_NB * PD_PITCH + 8 * 16 * PD_TP) * sizeof(double);
    auto kern = la ? potf2_diag_kernel<true> : potf2_diag_kernel<false>;
    { int rc_ = ensure_big_lds(reinterpret_cast<const void *>(kern), la ? 4 : 1, shmem); if (rc_ != GPT_OK) return rc_; }
    hipLaunchKernelGGL(kern, dim3(1, (unsigned)nbatch), dim3(PD_THREADS), shmem, st, A, lda, invd, info, info_base, wait.word,
                       wait.value, wait.err, bstride_a, bstride_ws);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

// ---- panel TRSM: B (m x 128) <- B * L^-T, L = 128x128 lower, invd = the block's packed workspace (common.hpp) ----
// One wave per 16 rows, blocked forward substitution X_j = (B_j - sum_{c<j} X_c L_jc^T) inv(L_jj)^T with every
// product on MFMA.  The 28 strictly-lower 16x16 blocks of L arrive already packed in B-operand lane order (written
// by potf2_diag_kernel) and are copied once per workgroup into LDS with coalesced 16-byte loads (a fragment read is
// 512 contiguous bytes); the B tiles, the packed inv(L_jj) fragments and the A-operand form of the finished X_c
// blocks live in registers; the accumulator -> A-operand re-layout goes through a 16x16 per-wave LDS scratch.
// (A variant that streams the L fragments from global memory needs only 9 KB of LDS but measured 12.6 us against
// 10.8 us for this one, and did not improve the overlap with a concurrent trailing update.)
#define TP_WAVES 4
__global__ __launch_bounds__(64 * TP_WAVES, 2) void trsm_panel_kernel(int64_t m, const double *__restrict__ L,
                                                                      int64_t ldl, const double *__restrict__ invd,
                                                                      double *__restrict__ B, int64_t ldb, unsigned *edge,
                                                                      unsigned edge_val, int64_t bs_a, int64_t bs_ws)
{
    invd += (int64_t)blockIdx.y * bs_ws;          // (batched: blockIdx.y = element, see potf2_diag_kernel)
    B += (int64_t)blockIdx.y * bs_a;
    __shared__ __attribute__((aligned(16))) double Lp[28][4][64];
    __shared__ __attribute__((aligned(16))) double Sc[TP_WAVES][16][TP_SP];
    __builtin_amdgcn_s_setprio(2);        // (on the chain, sharing CUs with the main stream's update: see gemm.hip)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    constexpr int NB16 = PD_NB / 16;
    (void)L;
    (void)ldl;

    // Prologue: all loads are issued before the first wait.
    const double *lpk = invd + GPT_WS_LOFF;
    constexpr int NPK = 28 * 256 / 2 / (64 * TP_WAVES);      // 16-byte chunks per thread
    f64x2 v[NPK];
#pragma unroll
    for (int q = 0; q < NPK; q++) v[q] = *reinterpret_cast<const f64x2 *>(lpk + 2 * (tid + q * 64 * TP_WAVES));
    const int64_t row0 = ((int64_t)blockIdx.x * TP_WAVES + wave) * 16;
    const bool active = row0 < m;
    f64x4 bt[NB16];
    double dv[NB16][4], xa[NB16][4];
    if (active) {
#pragma unroll
        for (int j = 0; j < NB16; j++) {
#pragma unroll
            for (int r = 0; r < 4; r++) bt[j][r] = B[(row0 + fk + 4 * r) * ldb + j * 16 + fr];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) dv[j][kk] = invd[j * 256 + kk * 64 + lane];
        }
    }
    {
        double *lflat = &Lp[0][0][0];
#pragma unroll
        for (int q = 0; q < NPK; q++) *reinterpret_cast<f64x2 *>(lflat + 2 * (tid + q * 64 * TP_WAVES)) = v[q];
    }
    __syncthreads();
    double (*X)[TP_SP] = Sc[wave];
    if (active) {
#pragma unroll
    for (int j = 0; j < NB16; j++) {
        f64x4 acc = bt[j];
#pragma unroll
        for (int c = 0; c < j; c++) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[c][kk], Lp[j * (j - 1) / 2 + c][kk][lane], acc, 0, 0, 0);
        }
        // accumulator (C layout) -> A operand through the per-wave scratch
#pragma unroll
        for (int r = 0; r < 4; r++) X[fk + 4 * r][fr] = acc[r];
        double av[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) av[kk] = X[fr][fk + 4 * kk];
        f64x4 res = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; kk++) res = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], dv[j][kk], res, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            // (a launch that raises an edge flag writes its rows through to memory, see EdgeSig)
            if (edge) __hip_atomic_store(&B[(row0 + fk + 4 * r) * ldb + j * 16 + fr], res[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else B[(row0 + fk + 4 * r) * ldb + j * 16 + fr] = res[r];
            X[fk + 4 * r][fr] = res[r];
        }
        if (j + 1 < NB16) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++) xa[j][kk] = -X[fr][fk + 4 * kk];
        }
    }
    }
    if (edge) edge_signal(edge, edge_val, gridDim.x);
}

int launch_trsm_panel(hipStream_t st, int64_t m, const double *L, int64_t ldl, const double *invd, double *B,
                      int64_t ldb, hipEvent_t done, EdgeSig edge, int64_t nbatch, int64_t bstride_a, int64_t bstride_ws)
{
    gpt_jitter(st);
    if (m <= 0) {
        if (done) GPT_HIP_CHECK(hipEventRecord(done, st));
        if (edge.word) {
            gpt_set_error("trsm_panel: an empty launch cannot raise an edge flag");
            return GPT_E_ARG;
        }
        return GPT_OK;
    }
    if (m % 16) {
        gpt_set_error("trsm_panel: m must be a multiple of 16 (m=%lld)", (long long)m);
        return GPT_E_ARG;
    }
    const int64_t nwave = m / 16;
    const unsigned grid = (unsigned)((nwave + TP_WAVES - 1) / TP_WAVES);
    // `done` rides on the kernel's own completion signal (hipExtLaunchKernelGGL stop event): a separate
    // hipEventRecord would put a barrier packet -- ~6 us of command-processor time -- on the panel chain
    if (done) hipExtLaunchKernelGGL(trsm_panel_kernel, dim3(grid, (unsigned)nbatch), dim3(64 * TP_WAVES), 0, st, nullptr, done, 0, m, L, ldl,
                                    invd, B, ldb, edge.word, edge.value, bstride_a, bstride_ws);
    else hipLaunchKernelGGL(trsm_panel_kernel, dim3(grid, (unsigned)nbatch), dim3(64 * TP_WAVES), 0, st, m, L, ldl, invd, B, ldb, edge.word,
                            edge.value, bstride_a, bstride_ws);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}
