// kbuild_kernel.hpp -- the fused covariance-builder kernel (see kbuild.hip), shared by the single-matrix launcher
// (kbuild.hip) and the batched one (kbuild_batch.hip: gpt_fit_batch, one matrix per blockIdx.z).
#pragma once
#include "kpair.hpp"

#define KB_THREADS 256
#define KB_CPT 1                        // columns per lane.  2 (16-byte stores) lifts the pure write pattern from 4.6 to
                                        // 5.4 TB/s but the pair arithmetic then runs with half the waves: SE 4.65 -> 4.44
                                        // TB/s, Matern-5/2 with derivative rows 0.18 -> 0.21 ms at N=8192 (measured)
#define KB_COLS (KB_THREADS * KB_CPT)
#define KB_ROWS 32
#define KB_RATIO (KB_COLS / KB_ROWS)

// BATCH: element blockIdx.z of a batch of independent matrices over the same points -- its hyperparameters kps[z], its
// noise variance nvs[z], its matrix K + z * bstride (gpt_fit_batch); `kp_one` / `noise_one` are unused then.
template <int KID, int D, bool BATCH>
__global__ __launch_bounds__(KB_THREADS) void kbuild_kernel(
    KParams kp_one, const double *__restrict__ Xi, const int32_t *__restrict__ ni, int64_t M,
    const double *__restrict__ Xj, const int32_t *__restrict__ nj, int64_t P,
    int lower_only, int64_t i0, int64_t j0, const double *__restrict__ err_y, double noise_one,
    double diag_add, double *__restrict__ K, int64_t ldk, int accumulate, const KParams *__restrict__ kps,
    const double *__restrict__ nvs, int64_t bstride, KParams kp_two_one, const KParams *__restrict__ kps2)
{
    const KParams &kp = BATCH ? kps[blockIdx.z] : kp_one;
    const KParams &kp_two = (BATCH && kps2 != nullptr) ? kps2[blockIdx.z] : kp_two_one;      // second factor of a product term
    const double noise_var = BATCH ? nvs[blockIdx.z] : noise_one;
    if (BATCH) K += (int64_t)blockIdx.z * bstride;
    int64_t rt, ct;
    if (lower_only == 2) {
        // Triangular launch (i0 == j0, M == P): only the tiles that touch the lower triangle exist.  Row tile rt
        // (KB_ROWS rows) needs column tiles 0 .. rt / R (R = KB_COLS / KB_ROWS); groups of R row tiles share a count,
        // so with g = rt / R the tiles before group g number R g (g + 1) / 2 and the linear index inverts in closed
        // form.
        constexpr int64_t R = KB_RATIO;
        const int64_t b = blockIdx.x;
        int64_t g = (int64_t)((sqrt(1.0 + 8.0 * (double)b / (double)R) - 1.0) * 0.5);
        while (g > 0 && R * g * (g + 1) / 2 > b) g--;
        while (R * (g + 1) * (g + 2) / 2 <= b) g++;
        const int64_t rem = b - R * g * (g + 1) / 2;
        rt = R * g + rem / (g + 1);
        ct = rem % (g + 1);
        if (rt * KB_ROWS >= M) return;
    } else {
        rt = blockIdx.y;
        ct = blockIdx.x;
    }
    const int64_t rbase = rt * KB_ROWS;
    const int64_t cbase = ct * KB_COLS;
    if (lower_only && (cbase + j0 > rbase + KB_ROWS - 1 + i0)) return;     // tile strictly above the diagonal
    const int64_t jfirst = cbase + (int64_t)threadIdx.x * KB_CPT;
    double xj[KB_CPT][D];
    int njr[KB_CPT][D];
#pragma unroll
    for (int c = 0; c < KB_CPT; c++) {
        const int64_t jc = (jfirst + c < P) ? jfirst + c : (P - 1);
#pragma unroll
        for (int d = 0; d < D; d++) {
            xj[c][d] = Xj[jc * D + d];
            njr[c][d] = nj[jc * D + d];
        }
    }
    // both columns in range and the pair 16-byte aligned -> one dwordx4 store per row
    const bool vec = (jfirst + KB_CPT <= P) && ((ldk & 1) == 0) && ((((uintptr_t)K >> 3) + (uint64_t)jfirst) & 1) == 0;
    const int64_t rend = (rbase + KB_ROWS < M) ? rbase + KB_ROWS : M;
    // PLAIN tiles (round 4).  Most of a real Gram matrix pairs value rows with value columns: if none of the
    // tile's rows and none of this wave's columns carries a derivative order, the row loop needs no class logic at all -- no
    // derivative-order loads, no index selection (the ~25 scalar and ~6 vector instructions per row the general loop spends on
    // choosing among four formulas).  The test is once per tile and wave: the orders of the rows in one or two vector loads
    // and a ballot.  C3 (last quarter derivative rows): 56 % of the lower triangle's pairs.
    if constexpr ((KID == GPT_KERNEL_SE || KID == GPT_KERNEL_M52) && KB_CPT == 1) {
        int cn = 0;
#pragma unroll
        for (int d = 0; d < D; d++) cn |= njr[0][d];
        bool plain = kp.hyper_deriv < 0 && __builtin_amdgcn_ballot_w64(cn != 0) == 0;
        if (plain) {
            int rn = 0;
            const int64_t cnt = (rend - rbase) * D;
            for (int64_t q = threadIdx.x & 63; q < cnt; q += 64) rn |= ni[rbase * D + q];
            plain = __builtin_amdgcn_ballot_w64(rn != 0) == 0;
        }
        if (plain) {
            for (int64_t i = rbase; i < rend; i++) {
                double xi[D];
#pragma unroll
                for (int d = 0; d < D; d++) xi[d] = Xi[i * D + d];          // wave-uniform addresses -> scalar loads
                double v = plain_pair<KID, D>(kp, xi, xj[0]);
                if (accumulate && jfirst < P) v += K[i * ldk + jfirst];
                if (err_y != nullptr && (i + i0 == jfirst + j0)) {
                    const double e = err_y[i + i0];
                    v = ((v + noise_var) + e * e) + diag_add;
                }
#ifdef KB_DEBUG_NOCOMPUTE
                v = (double)i;
#endif
#ifdef KB_DEBUG_NOSTORE
                if (v != 12345.678) continue;
#endif
                if (jfirst < P) K[i * ldk + jfirst] = v;
            }
            return;
        }
    }
    for (int64_t i = rbase; i < rend; i++) {
        double xi[D];
        int nir[D];
#pragma unroll
        for (int d = 0; d < D; d++) {          // wave-uniform addresses -> scalar loads
            xi[d] = Xi[i * D + d];
            nir[d] = ni[i * D + d];
        }
        double v[KB_CPT];
#pragma unroll
        for (int c = 0; c < KB_CPT; c++) {
            if constexpr (KID == GPT_KERNEL_PRODUCT) v[c] = prod_pair<D>(kp, kp_two, xi, xj[c], nir, njr[c]);
            else v[c] = any_pair<KID, D>(kp, xi, xj[c], nir, njr[c]);
            // SumKernel (ref: gptools/kernel/core.py:549-584): later terms add to what the earlier passes stored
            if (accumulate && jfirst + c < P) v[c] += K[i * ldk + jfirst + c];
            if (err_y != nullptr && (i + i0 == jfirst + c + j0)) {
                const double e = err_y[i + i0];
                v[c] = ((v[c] + noise_var) + e * e) + diag_add;
            }
        }
#ifdef KB_DEBUG_NOCOMPUTE           // (measurement builds, scratch/kb_variants.sh: the store pattern alone / the arithmetic alone.
        v[0] = (double)i;           //  Round 3, N = 8192 lower triangle: stores alone 5.6 TB/s; arithmetic alone = the product's time, for
#endif                             //  SE as for Matern-5/2: the builder is bound by its ~57 VALU + ~45 SALU instructions per row and wave)
#ifdef KB_DEBUG_NOSTORE
        if (v[0] != 12345.678) continue;
#endif
        if (KB_CPT == 2 && vec) {
            f64x2 w = {v[0], v[KB_CPT - 1]};
            *reinterpret_cast<f64x2 *>(K + i * ldk + jfirst) = w;
        } else {
#pragma unroll
            for (int c = 0; c < KB_CPT; c++)
                if (jfirst + c < P) K[i * ldk + jfirst + c] = v[c];
        }
    }
}

template <int KID, int D>
__global__ __launch_bounds__(256) void kpairs_kernel(KParams kp, const double *__restrict__ Xi,
                                                     const double *__restrict__ Xj,
                                                     const int32_t *__restrict__ ni,
                                                     const int32_t *__restrict__ nj, int64_t M,
                                                     double *__restrict__ out, int accumulate, KParams kp_two)
{
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    double xi[D], xj[D];
    int nir[D], njr[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        xi[d] = Xi[m * D + d];
        xj[d] = Xj[m * D + d];
        nir[d] = ni[m * D + d];
        njr[d] = nj[m * D + d];
    }
    double v;
    if constexpr (KID == GPT_KERNEL_PRODUCT) v = prod_pair<D>(kp, kp_two, xi, xj, nir, njr);
    else v = any_pair<KID, D>(kp, xi, xj, nir, njr);
    out[m] = accumulate ? out[m] + v : v;
}
