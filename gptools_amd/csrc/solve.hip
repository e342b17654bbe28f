// solve.hip -- memory-bound helpers around the factorisation (gfx950): padding / augmentation of
// K_tot, log-determinant and quadratic-form reductions, triangular read-back, the backward
// substitution for alpha, GEMV and the predictive-variance row reduction.
//
// ref: gptools/gaussian_process.py:1462-1467 (alpha, ll), :971 (mean = Kstar^T alpha),
//      :987,1006 (cov diagonal / std).
#include "common.hpp"
#include <type_traits>
#define GPT_TRY_RC_SOLVE(expr) do { int rc_ = (expr); if (rc_ != GPT_OK) return rc_; } while (0)

// Rows [n_valid, n_pad) of the padded matrix: zero, unit diagonal.  If dy != NULL row n_valid
// carries y^T (the "augmented row": after the factorisation it holds z^T = (L^-1 y)^T) and its
// diagonal entry is `big`, so the pivot there stays positive whatever z.z is.
__global__ void fill_pad_kernel(double *__restrict__ A, int64_t lda, int64_t n_valid, int64_t n_pad,
                                const double *__restrict__ dy, double big)
{
    const int64_t row = n_valid + blockIdx.y;
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_pad || col >= n_pad) return;
    double v = 0.0;
    if (row == n_valid && dy != nullptr) {
        if (col < n_valid) v = dy[col];
        else if (col == row) v = big;
    } else if (col == row) {
        v = 1.0;
    }
    A[row * lda + col] = v;
}

// Raises an edge-flag word (EdgeSig) from the stream it is launched on: everything in front of it on that stream is complete
// and visible (kernel boundary) when it runs.
__global__ void set_flag_kernel(unsigned *word, unsigned value)
{
    // (release: this kernel has no payload of its own -- what it publishes was written by the kernels in front of it on the
    // stream, whose end-of-kernel write-back has happened -- so the fence finds nothing dirty and costs nothing)
    __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Stream-side end of a flag edge: a one-wave kernel that polls the word (bounded, then acquire; common.hpp edge_poll).  In
// place of hipStreamWaitValue32, which is a kernel of the runtime of the same cost (~5 us, measured) but with no bound.
__global__ void wait_flag_kernel(const unsigned *word, unsigned value, unsigned *err)
{
    if (threadIdx.x == 0) edge_poll<4, false>(word, value, err);
}

int launch_wait_flag(hipStream_t st, EdgeSig w)
{
    if (!w.word) return GPT_OK;
    hipLaunchKernelGGL(wait_flag_kernel, dim3(1), dim3(64), 0, st, w.word, w.value, w.err);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

int launch_set_flag(hipStream_t st, unsigned *word, unsigned value)
{
    hipLaunchKernelGGL(set_flag_kernel, dim3(1), dim3(1), 0, st, word, value);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

__global__ void jitter_kernel(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

void gpt_jitter(hipStream_t st)
{
    static int max_us = -1;
    static unsigned long long state = 0x9E3779B97F4A7C15ull;
    if (max_us < 0) {
        const char *e = getenv("GPT_JITTER");
        max_us = e ? atoi(e) : 0;
    }
    if (max_us <= 0) return;
    state = state * 6364136223846793005ull + 1442695040888963407ull;
    const double u = (double)(state >> 40) / (double)(1ull << 24);
    const long long ticks = (long long)(u * u * u * max_us * 100.0);       // wall_clock64 ticks at 100 MHz; mostly short
    hipLaunchKernelGGL(jitter_kernel, dim3(1), dim3(1), 0, st, ticks);
}

// The start of an LML evaluation in ONE launch: upload y | err_y from the pinned staging buffer (the kernel reads host memory
// directly: ncopy doubles, coalesced), zero the info word, write the padding rows with y in the augmented row.  Replaces a
// host-to-device copy, a memset and fill_pad: while the GPU is idle at the start of an evaluation every launch costs what
// the HOST needs to issue it (5-13 us each in the trace).
__global__ void upload_pad_kernel(const double *__restrict__ h_src, double *__restrict__ d_dst, int64_t ncopy,
                                  int32_t *__restrict__ info, double *__restrict__ A, int64_t lda, int64_t n_valid,
                                  int64_t n_pad, double big)
{
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.y == 0) {
        if (col == 0 && info) *info = 0;
        // (the grid's x extent covers max(ncopy, n_pad) entries)
        if (col < ncopy) d_dst[col] = h_src[col];
    }
    const int64_t row = n_valid + blockIdx.y;
    if (row >= n_pad || col >= n_pad) return;
    double v = 0.0;
    if (row == n_valid) {
        if (col < n_valid) v = h_src[col];
        else if (col == row) v = big;
    } else if (col == row) {
        v = 1.0;
    }
    A[row * lda + col] = v;
}

int launch_upload_pad(hipStream_t st, const double *h_src, double *d_dst, int64_t ncopy, int32_t *info, double *A,
                      int64_t lda, int64_t n_valid, int64_t n_pad, double big)
{
    if (n_pad <= n_valid) {
        gpt_set_error("upload_pad: the padded order must exceed the order (augmented row)");
        return GPT_E_ARG;
    }
    const int64_t w = ncopy > n_pad ? ncopy : n_pad;
    dim3 grid((unsigned)((w + 255) / 256), (unsigned)(n_pad - n_valid));
    hipLaunchKernelGGL(upload_pad_kernel, grid, dim3(256), 0, st, h_src, d_dst, ncopy, info, A, lda, n_valid, n_pad, big);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

int launch_fill_pad(hipStream_t st, double *A, int64_t lda, int64_t n_valid, int64_t n_pad, const double *dy,
                    double big)
{
    if (n_pad <= n_valid) return GPT_OK;
    dim3 grid((unsigned)((n_pad + 255) / 256), (unsigned)(n_pad - n_valid));
    hipLaunchKernelGGL(fill_pad_kernel, grid, dim3(256), 0, st, A, lda, n_valid, n_pad, dy, big);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

// A[i][i] = (A[i][i] + err[i]^2) + diag_add  (ref: gptools/gaussian_process.py:1447-1451 on an assembled matrix)
__global__ void add_diag_kernel(double *__restrict__ A, int64_t lda, int64_t n, const double *__restrict__ err,
                                double diag_add, int64_t bstride)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    A += (int64_t)blockIdx.y * bstride;                 // (batched: one matrix per blockIdx.y, the same err for all)
    const double e = err[i];
    A[i * lda + i] = (A[i * lda + i] + e * e) + diag_add;
}

int launch_add_diag(hipStream_t st, double *A, int64_t lda, int64_t n, const double *err, double diag_add, int64_t nbatch,
                    int64_t bstride)
{
    if (n <= 0 || nbatch <= 0) return GPT_OK;
    hipLaunchKernelGGL(add_diag_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)nbatch), dim3(256), 0, st, A, lda, n, err, diag_add,
                       bstride);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

// out[0] = sum_{i<n} log A[i][i] ; out[1] = sum_{c<n} A[n][c]^2 (the augmented row z) ; out[2] = *info -- everything an
// LML evaluation returns.  LD_WGS workgroups take 256 entries at a time (the diagonal is one cache line per entry: a
// single workgroup's address unit needed 4 us for the 8192 lines of n = 8192 and the kernel 15-17 us, on the tail of every
// evaluation); each leaves a partial pair, the LAST one to finish adds them up in index order (deterministic) and writes
// the three results to `out` -- which may be pinned host memory (system-scope stores): no copy kernel behind it.
#define LD_WGS 32
__global__ __launch_bounds__(256) void logdet_dot_kernel(const double *__restrict__ A, int64_t lda, int64_t n,
                                                         int has_z, const int32_t *__restrict__ info,
                                                         double *__restrict__ part, unsigned *__restrict__ count,
                                                         double *__restrict__ out, unsigned *__restrict__ edge_err)
{
    __shared__ double s0[4], s1[4];
    __shared__ bool last;
    double a = 0.0, b = 0.0;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 4 * 256 * LD_WGS) {
        double dg[4], zz[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int64_t i = i0 + (int64_t)q * 256 * LD_WGS;
            dg[q] = (i < n) ? A[i * lda + i] : 1.0;
            zz[q] = (has_z && i < n) ? A[n * lda + i] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            a += log(dg[q]);
            b = fma(zz[q], zz[q], b);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off);
        b += __shfl_down(b, off);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s0[wave] = a;
        s1[wave] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ta = ((s0[0] + s0[1]) + s0[2]) + s0[3], tb = ((s1[0] + s1[1]) + s1[2]) + s1[3];
        __hip_atomic_store(part + 2 * blockIdx.x, ta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(part + 2 * blockIdx.x + 1, tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned done = __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        last = (done == gridDim.x);
        if (last) __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (last && threadIdx.x < 64) {
        // the partial pairs in parallel (one per lane), then a fixed-shape tree: the same sum on every run
        const unsigned w = threadIdx.x;
        double ta = (w < gridDim.x) ? __hip_atomic_load(part + 2 * w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        double tb = (w < gridDim.x) ? __hip_atomic_load(part + 2 * w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        for (int off = 32; off > 0; off >>= 1) {
            ta += __shfl_down(ta, off);
            tb += __shfl_down(tb, off);
        }
        if (w == 0) {
            __hip_atomic_store(out + 0, ta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(out + 1, tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(out + 2, info ? (double)*info : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            // out[3]: a flag-edge wait of this evaluation timed out (EdgeSig, common.hpp); read and cleared
            double ee = 0.0;
            if (edge_err) {
                ee = (double)__hip_atomic_load(edge_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(edge_err, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __hip_atomic_store(out + 3, ee, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// d_part: 2 * LD_WGS doubles followed by one 32-bit counter (zero on entry, left zero); out3: FOUR doubles of device or
// pinned host memory.  ev0 / ev1 (optional): start / stop events on the dispatch packet itself.
int launch_logdet_dot(hipStream_t st, const double *A, int64_t lda, int64_t n, const int32_t *d_info, double *d_part,
                      double *out3, hipEvent_t ev0, hipEvent_t ev1, unsigned *edge_err)
{
    unsigned *count = reinterpret_cast<unsigned *>(d_part + 2 * LD_WGS);
    if (ev0 || ev1)
        hipExtLaunchKernelGGL(logdet_dot_kernel, dim3(LD_WGS), dim3(256), 0, st, ev0, ev1, 0, A, lda, n, 1, d_info, d_part, count, out3, edge_err);
    else
        hipLaunchKernelGGL(logdet_dot_kernel, dim3(LD_WGS), dim3(256), 0, st, A, lda, n, 1, d_info, d_part, count, out3, edge_err);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

// ---- batched small fits (gpt_fit_batch): one matrix per batch element, bstride elements apart -------------------------------
// Padding rows + augmented row of every element (as fill_pad_kernel), y read from the pinned host buffer h_y (nbatch x n_valid),
// and the elements' info words cleared.  blockIdx.z = element.
__global__ void batch_pad_kernel(const double *__restrict__ h_y, double *__restrict__ A, int64_t lda, int64_t bstride,
                                 int64_t n_valid, int64_t n_pad, double big, int32_t *__restrict__ info)
{
    const int64_t b = blockIdx.z;
    const int64_t row = n_valid + blockIdx.y;
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) info[b] = 0;
    if (row >= n_pad || col >= n_pad) return;
    double v = 0.0;
    if (row == n_valid) {
        if (col < n_valid) v = h_y[b * n_valid + col];
        else if (col == row) v = big;
    } else if (col == row) {
        v = 1.0;
    }
    A[b * bstride + row * lda + col] = v;
}

int launch_batch_pad(hipStream_t st, const double *h_y, int64_t nbatch, double *A, int64_t lda, int64_t bstride, int64_t n_valid,
                     int64_t n_pad, double big, int32_t *info)
{
    if (n_pad <= n_valid || nbatch <= 0) return GPT_OK;
    dim3 grid((unsigned)((n_pad + 255) / 256), (unsigned)(n_pad - n_valid), (unsigned)nbatch);
    hipLaunchKernelGGL(batch_pad_kernel, grid, dim3(256), 0, st, h_y, A, lda, bstride, n_valid, n_pad, big, info);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

// The two ll scalars + info of every element: out[4 b + {0, 1, 2}].  One workgroup per element walks the LD_WGS partial sums
// of logdet_dot_kernel one after another, each formed by the same threads in the same order and reduced by the same tree, so
// an element's scalars carry the very bits a single gpt_fit of it returns.
__global__ __launch_bounds__(256) void batch_logdet_dot_kernel(const double *__restrict__ A, int64_t lda, int64_t bstride,
                                                               int64_t n, const int32_t *__restrict__ info,
                                                               double *__restrict__ out)
{
    __shared__ double s0[4], s1[4], p0[LD_WGS], p1[LD_WGS];
    const double *Ab = A + (int64_t)blockIdx.x * bstride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int vw = 0; vw < LD_WGS; vw++) {
        double a = 0.0, b = 0.0;
        for (int64_t i0 = (int64_t)vw * 256 + threadIdx.x; i0 < n; i0 += 4 * 256 * LD_WGS) {
            double dg[4], zz[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int64_t i = i0 + (int64_t)q * 256 * LD_WGS;
                dg[q] = (i < n) ? Ab[i * lda + i] : 1.0;
                zz[q] = (i < n) ? Ab[n * lda + i] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                a += log(dg[q]);
                b = fma(zz[q], zz[q], b);
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_down(a, off);
            b += __shfl_down(b, off);
        }
        if (lane == 0) {
            s0[wave] = a;
            s1[wave] = b;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            p0[vw] = ((s0[0] + s0[1]) + s0[2]) + s0[3];
            p1[vw] = ((s1[0] + s1[1]) + s1[2]) + s1[3];
        }
        __syncthreads();
    }
    if (threadIdx.x < 64) {
        const unsigned w = threadIdx.x;
        double ta = (w < LD_WGS) ? p0[w] : 0.0, tb = (w < LD_WGS) ? p1[w] : 0.0;
        for (int off = 32; off > 0; off >>= 1) {
            ta += __shfl_down(ta, off);
            tb += __shfl_down(tb, off);
        }
        if (w == 0) {
            double *o = out + 4 * (int64_t)blockIdx.x;
            __hip_atomic_store(o + 0, ta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(o + 1, tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(o + 2, (double)info[blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

int launch_batch_logdet_dot(hipStream_t st, const double *A, int64_t lda, int64_t bstride, int64_t n, int64_t nbatch,
                            const int32_t *d_info, double *out3)
{
    if (nbatch <= 0) return GPT_OK;
    hipLaunchKernelGGL(batch_logdet_dot_kernel, dim3((unsigned)nbatch), dim3(256), 0, st, A, lda, bstride, n, d_info, out3);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

__global__ void extract_lower_kernel(const double *__restrict__ A, int64_t lda, int64_t n,
                                     double *__restrict__ out, int64_t ldo)
{
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n) return;
    for (int64_t row = blockIdx.y; row < n; row += gridDim.y)
        out[row * ldo + col] = (col <= row) ? A[row * lda + col] : 0.0;
}

int launch_extract_lower(hipStream_t st, const double *A, int64_t lda, int64_t n, double *out, int64_t ldo)
{
    if (n <= 0) return GPT_OK;
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)(n < 16384 ? n : 16384));
    hipLaunchKernelGGL(extract_lower_kernel, grid, dim3(256), 0, st, A, lda, n, out, ldo);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

__global__ void copy2d_kernel(int64_t rows, int64_t cols, const double *__restrict__ src, int64_t lds,
                              double *__restrict__ dst, int64_t ldd)
{
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= cols) return;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) dst[row * ldd + col] = src[row * lds + col];
}

int launch_copy2d(hipStream_t st, int64_t rows, int64_t cols, const double *src, int64_t lds, double *dst, int64_t ldd)
{
    if (rows <= 0 || cols <= 0) return GPT_OK;
    dim3 grid((unsigned)((cols + 255) / 256), (unsigned)(rows < 16384 ? rows : 16384));
    hipLaunchKernelGGL(copy2d_kernel, grid, dim3(256), 0, st, rows, cols, src, lds, dst, ldd);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

__global__ void zero2d_kernel(int64_t rows, int64_t cols, double *__restrict__ dst, int64_t ldd)
{
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= cols) return;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) dst[row * ldd + col] = 0.0;
}

int launch_zero2d(hipStream_t st, int64_t rows, int64_t cols, double *dst, int64_t ldd)
{
    if (rows <= 0 || cols <= 0) return GPT_OK;
    dim3 grid((unsigned)((cols + 255) / 256), (unsigned)(rows < 16384 ? rows : 16384));
    hipLaunchKernelGGL(zero2d_kernel, grid, dim3(256), 0, st, rows, cols, dst, ldd);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

// ---- backward substitution  L^T x = z  (x overwrites z), n a multiple of 128 ----------------
// Right-looking over 128-wide blocks, last block first:
//   (1) trsv_lt_diag_kernel: one wave solves L_bb^T x_b = w_b (two entries per lane, the matrix
//       row arrives from LDS, the pivot is broadcast with v_readlane; 1/L_ii comes from invd);
//   (2) trsv_lt_update_kernel: w[0 : b*128) -= L[b-block rows, 0 : b*128)^T x_b, one column per
//       lane, rows of L read as contiguous coalesced segments.
// 256 threads: the packed workspace of the diagonal block (inverses of its eight 16x16 diagonal blocks + its 28
// strictly-lower 16x16 blocks, 72 KB) is staged into LDS with coalesced loads, then the 128 unknowns fall in eight
// block steps, last block first:  x_b = inv(L_bb)^T w_b  (16 lanes, 16 FMAs),  w_j -= L_bj^T x_b for every j < b
// (16 lanes per block j, 16 FMAs).  Packed element (r, c) of a 16x16 block sits at (c >> 2) * 64 + r + 16 * (c & 3)
// (MFMA B-operand lane order, see potrf.hip), i.e. a column of the block is 16 consecutive doubles.
// (The first version walked the 128 columns one by one on a single wave that also loaded the 128 KB block by
// itself: 100 us per block, 7 ms at N = 8192 -- more than the factorisation.)
__global__ __launch_bounds__(256) void trsv_lt_diag_kernel(const double *__restrict__ Lbb, int64_t ldl,
                                                           const double *__restrict__ invd, double *__restrict__ x)
{
    (void)Lbb;
    (void)ldl;
    __shared__ double ws[GPT_WS_BLOCK];
    __shared__ double w[128], xb[16];
    const int tid = threadIdx.x;
    for (int i = tid; i < GPT_WS_BLOCK; i += 256) ws[i] = invd[i];
    if (tid < 128) w[tid] = x[tid];
    __syncthreads();
    const int j = tid >> 4, i = tid & 15;                       // lane i of 16-block j (threads 0..127)
    const int col = (i >> 2) * 64 + 16 * (i & 3);               // start of packed column i
    for (int b = 7; b >= 0; b--) {
        if (tid < 128 && j == b) {
            const double *inv = ws + b * 256 + col;            // column i of inv(L_bb): entries (k, i)
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 16; k++) acc = fma(inv[k], w[b * 16 + k], acc);
            xb[i] = acc;
        }
        __syncthreads();
        if (tid < 128) {
            if (j == b) w[tid] = xb[i];
            else if (j < b) {
                const double *lb = ws + GPT_WS_LOFF + (b * (b - 1) / 2 + j) * 256 + col;     // column i of L_bj
                double acc = w[tid];
#pragma unroll
                for (int k = 0; k < 16; k++) acc = fma(-lb[k], xb[k], acc);
                w[tid] = acc;
            }
        }
        __syncthreads();
    }
    if (tid < 128) x[tid] = w[tid];
}

__global__ __launch_bounds__(256) void trsv_lt_update_kernel(int64_t ncols, const double *__restrict__ Lrow,
                                                             int64_t ldl, const double *__restrict__ xb,
                                                             double *__restrict__ w)
{
    __shared__ double xs[128];
    if (threadIdx.x < 128) xs[threadIdx.x] = xb[threadIdx.x];
    __syncthreads();
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= ncols) return;
    double acc = 0.0;
#pragma unroll 8
    for (int r = 0; r < 128; r++) acc = fma(Lrow[(int64_t)r * ldl + col], xs[r], acc);
    w[col] -= acc;
}

// (b_lo > 0: only the blocks b >= b_lo are solved -- their updates still reach every column to their left; the caller
// continues from there with wider steps, launch_trsv_lt_wide)
int launch_trsv_lt(hipStream_t st, int64_t n, const double *L, int64_t ldl, const double *invd, double *x, int64_t b_lo)
{
    if (n % 128) {
        gpt_set_error("trsv_lt: n must be a multiple of 128");
        return GPT_E_ARG;
    }
    for (int64_t b = n / 128 - 1; b >= b_lo; b--) {
        const double *Lbb = L + (b * 128) * ldl + b * 128;
        hipLaunchKernelGGL(trsv_lt_diag_kernel, dim3(1), dim3(256), 0, st, Lbb, ldl, invd + b * GPT_WS_BLOCK, x + b * 128);
        if (b > 0) {
            const int64_t ncols = b * 128;
            hipLaunchKernelGGL(trsv_lt_update_kernel, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, st, ncols,
                               L + (b * 128) * ldl, ldl, x + b * 128, x);
        }
    }
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

// ---- inverses of the 512 x 512 diagonal blocks of the factor, all of them in ONE launch ------------------------------------------
// Rounds 2-4 built them by recursion over launches (identity through the panel TRSM, 128 -> 256 -> 512 by batched GEMMs and TRSMs,
// one transpose: 15 dependent launches, 205 us at N = 8192 whatever the block count -- half of what alpha costs after a
// factorisation).  Here a workgroup owns 16 ROWS of one block's X = L_bb^-1 (strip s = rows [16 s, 16 s + 16), non-zero in columns
// [0, 16 s + 16)) and solves X L_bb = I for them from the diagonal leftwards, right-looking, in 16-column steps against the inverted
// 16x16 diagonal blocks the factorisation left in its workspace:
//   step k = s, s - 1, .. :  X_k = R_k D_k^-1          (the wave that owns column block k; R = the running right-hand side, I at start)
//                            R_j -= X_k L_kj           for every column block j < k   (eight waves; wave w owns the j = w mod 8)
// so that a step reads ONE row block of L (16 rows, contiguous bytes along each) -- a first version that owned 16 columns read a
// 128-byte column slice of every row below k per step: 512 rows, 64 KB apart, 512 pages per step, 1.4 us per step in address
// translation alone (75 us per launch at N = 4096).  Everything is kept TRANSPOSED in the accumulators (acc = R_j^T): with the
// contraction index of MFMA u taken as g + 4u (g = lane / 16) the B operand of X_k^T = D_k^-T R_k^T IS the accumulator of R_k, and
// the solver's result registers ARE the B operand (X_k^T) of the updates R_j^T -= L_kj^T X_k^T -- the other waves fetch the same 32
// bytes per lane from a 2 KB LDS tile (two tiles, by step parity: one barrier per step).  The L fragments of step k - 2 are requested
// when step k has consumed their registers.  The wave that owns column block k - 1 updates it first and solves it at once; its other
// column blocks follow while the rest of the workgroup is already reading X_{k-1}.  Heavy strips (large s) are dealt first.
// No load of the step loop sits inside a branch (with loads inside branches hipcc waits vmcnt(0) in front of each of them and of
// every MFMA group): the loop is compiled once per number of live column blocks of a wave, the D_k^-1 come from LDS.
// Out: W = X (lower, rows of 512), U = X^T (upper), both with their zero halves.
#define TI_NB 512
#define TI_XS 513                                            // row pitch of the strip in LDS (odd: conflict-free column reads)
__global__ __launch_bounds__(512, 2) void trinv512_kernel(int nblk, const double *__restrict__ L, int64_t ldl, const double *__restrict__ invd,
                                                          double *__restrict__ U, double *__restrict__ W)
{
    __shared__ double xs[16 * TI_XS];                        // the strip of X, 16 rows (columns >= 16 s + 16 never touched)
    __shared__ double xt[2][256];                            // X_k^T in the solver's lane order: [lane][u]
    __shared__ double dl[32 * 256];                          // D_k^-1, k <= s, as the workspace holds them
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fk = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // eight waves: wave w owns the column blocks j = w mod 8
    const int s = 31 - (int)blockIdx.x / nblk, b = (int)blockIdx.x % nblk;
    const double *Lb = L + (int64_t)b * TI_NB * (ldl + 1);
    const double *wsb = invd + (int64_t)b * 4 * GPT_WS_BLOCK;

    f64x4 acc[4];                                            // acc[t][q] = R_j[fr][fk + 4 q], j = 8 t + w
#pragma unroll
    for (int t = 0; t < 4; t++) {
        acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
        if (8 * t + w == s) {
#pragma unroll
            for (int q = 0; q < 4; q++) acc[t][q] = (fk + 4 * q == fr) ? 1.0 : 0.0;      // R_s = I
        }
    }
    f64x4 la[2][4];
    // A operand of the update of column block j with row block k: L_kj^T, lane (fr, fk), MFMA u: L[16 k + fk + 4 u][16 j + fr].
    // What bounds a step is the CU's rate of vector-memory requests (a 64-lane 8-byte load every ~16 cycles whatever it hits), so only
    // the NL column blocks of this wave that are still being updated are requested: the step is compiled once per NL (below).
    auto request = [&](f64x4 (&dst)[4], int k, auto nl) {
        const int kc = k > 0 ? k : 0;
#pragma unroll
        for (int t = 0; t < decltype(nl)::value; t++) {
            const double *p = Lb + (int64_t)(16 * kc + fk) * ldl + 16 * (8 * t + w) + fr;
#pragma unroll
            for (int u = 0; u < 4; u++) dst[t][u] = p[(int64_t)(4 * u) * ldl];
        }
    };
    // X_k^T = D_k^-T R_k^T by the owner of column block k: x[q] = X[fr][16 k + fk + 4 q]
    // (A operand: D_k^-T, lane (fr, fk), MFMA u: D_k^-1[fk + 4 u][fr]; workspace order: element (r, c) at 64 (c / 4) + 16 (c % 4) + r)
    auto solve = [&](const f64x4 &r, int k) {
        const double *dk = dl + k * 256 + 64 * (fr >> 2) + 16 * (fr & 3) + fk;
        f64x4 x = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 4; u++) x = __builtin_amdgcn_mfma_f64_16x16x4f64(dk[4 * u], r[u], x, 0, 0, 0);
        *reinterpret_cast<f64x4 *>(&xt[k & 1][4 * lane]) = x;
#pragma unroll
        for (int q = 0; q < 4; q++) xs[fr * TI_XS + 16 * k + fk + 4 * q] = x[q];
    };
    // the slots of this wave that are live at step k: j = 8 t + w < k
    auto nlive = [&](int k) { const int n = (k - w + 7) >> 3; return n < 0 ? 0 : (n > 4 ? 4 : n); };

    {   // (sixteen loads in flight per thread, then sixteen LDS stores: as a plain loop hipcc waits for every load before its store,
        // sixteen dependent round trips = 13 us of the launch)
        double v[16];
        const int lim = 256 * (s + 1);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int e = tid + 512 * i, ec = e < lim ? e : 0;
            v[i] = wsb[(int64_t)(ec >> 11) * GPT_WS_BLOCK + (ec & 2047)];
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int e = tid + 512 * i;
            if (e < lim) dl[e] = v[i];
        }
    }
    request(la[0], s, std::integral_constant<int, 4>());
    request(la[1], s - 1, std::integral_constant<int, 4>());
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; t++)
        if (8 * t + w == s) solve(acc[t], s);
    __syncthreads();
    // step k: lk = the L fragments of row block k (requested two steps ago)
    auto step = [&](f64x4 (&lk)[4], int k, auto nl) {
        constexpr int NL = decltype(nl)::value;
        f64x4 xb = *reinterpret_cast<const f64x4 *>(&xt[k & 1][4 * lane]);
        xb = -xb;
        const int tn = (k - 1) >> 3;
        const bool next_owner = (w == ((k - 1) & 7));
        if (next_owner) {
#pragma unroll
            for (int t = 0; t < NL; t++)
                if (t == tn) {
#pragma unroll
                    for (int u = 0; u < 4; u++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(lk[t][u], xb[u], acc[t], 0, 0, 0);
                    solve(acc[t], k - 1);
                }
        }
#pragma unroll
        for (int t = 0; t < NL; t++) {
            if (!(next_owner && t == tn)) {
#pragma unroll
                for (int u = 0; u < 4; u++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(lk[t][u], xb[u], acc[t], 0, 0, 0);
            }
        }
        request(lk, k - 2, nl);
        __syncthreads();
    };
    // (a phase ends on an even step count, so that la[0] / la[1] keep their turns: its last step may carry one slot that has just died --
    // the block it solved in the step before; updating that accumulator further is harmless)
    int k = s;
    auto phase = [&](auto nl) {
        while (k > 0 && nlive(k) == decltype(nl)::value) {
            step(la[0], k, nl);
            if (k - 1 > 0) step(la[1], k - 1, nl);
            k -= 2;
        }
    };
    phase(std::integral_constant<int, 4>());
    phase(std::integral_constant<int, 3>());
    phase(std::integral_constant<int, 2>());
    phase(std::integral_constant<int, 1>());
    phase(std::integral_constant<int, 0>());

    // the strip goes out: W = its 16 rows of X (zeros right of column 16 s + 15), U = its 16 columns of X^T
    double *Ub = U + (int64_t)b * TI_NB * TI_NB, *Wb = W + (int64_t)b * TI_NB * TI_NB;
    const int nz = 16 * s + 16;
    {   // (all sixteen LDS reads of a thread before its sixteen stores, for the same reason)
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int e = tid + 512 * i, r = e >> 9, c = e & 511;
            v[i] = (c < nz) ? xs[r * TI_XS + c] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int e = tid + 512 * i, r = e >> 9, c = e & 511;
            Wb[(int64_t)(16 * s + r) * TI_NB + c] = v[i];
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int e = tid + 512 * i, c = e >> 4, r = e & 15;
            v[i] = (c < nz) ? xs[r * TI_XS + c] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int e = tid + 512 * i, c = e >> 4, r = e & 15;
            Ub[(int64_t)c * TI_NB + 16 * s + r] = v[i];
        }
    }
}

int launch_trinv512(hipStream_t st, int64_t nblk, const double *L, int64_t ldl, const double *invd, double *U, double *W)
{
    if (nblk <= 0) return GPT_OK;
    hipLaunchKernelGGL(trinv512_kernel, dim3((unsigned)(32 * nblk)), dim3(512), 0, st, (int)nblk, L, ldl, invd, U, W);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

// ---- the same substitution in 512-wide steps against explicit inverses of the diagonal blocks -------------------------------
// 127 dependent launches of ~8 us are what the 128-wide form costs at n = 8192 (1 ms, a fifth of an evaluation).  With
// U_j = L_jj^-T of the 512 x 512 diagonal blocks at hand (trinv512_kernel above) a step is  x_j = U_j w_j  (launch_gemv_n: one
// workgroup per row)  and  w[0 : j0) -= L[j0 : j0 + 512, 0 : j0)^T x_j  (this kernel): 2 n / 512 launches.
// A workgroup takes 16 columns: a CU turns out ~32 bytes of vector loads per cycle whatever they hit, so the 256 KB of the 64-column
// workgroups of rounds 2-4 were 3.9 us of requests alone (8-10 us per launch whatever ncols was); 16 columns are 64 KB, and j0 / 16
// workgroups fill the chip four times sooner.  Lane = (column, row mod 4); the eight waves split the 512 rows, sixteen loads in flight
// per lane (a load instruction covers four rows of 128 bytes); the 32 partial sums of a column meet in LDS in a fixed order.
#define TW_NB 512
// w_in: where the vector being updated is READ (w itself, or -- the first step of an eager alpha -- the augmented row z of the factor:
// w = z - M^T x then initialises w in passing and no init kernel sits in front of the substitution)
__global__ __launch_bounds__(512) void gemv_t_sub_kernel(int64_t ncols, const double *__restrict__ M, int64_t ldm,
                                                         const double *__restrict__ x, double *__restrict__ w, const double *__restrict__ w_in)
{
    __shared__ double xs[TW_NB];
    __shared__ double part[32][17];
    const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6, c = lane & 15, rs = lane >> 4;
    xs[tid] = x[tid];
    __syncthreads();
    const int64_t col = (int64_t)blockIdx.x * 16 + c;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (col < ncols) {
        const int r0 = g * (TW_NB / 8) + rs;                  // rows r0 + 4 i, i < 16
        const double *p = M + (int64_t)r0 * ldm + col;
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = p[(int64_t)(4 * i) * ldm];
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            a0 = fma(v[i + 0], xs[r0 + 4 * (i + 0)], a0);
            a1 = fma(v[i + 1], xs[r0 + 4 * (i + 1)], a1);
            a2 = fma(v[i + 2], xs[r0 + 4 * (i + 2)], a2);
            a3 = fma(v[i + 3], xs[r0 + 4 * (i + 3)], a3);
        }
    }
    part[g * 4 + rs][c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (tid < 16 && col < ncols) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 32; q++) t += part[q][tid];
        w[col] = w_in[col] - t;
    }
}

// Blocks [0, nwide / 512) of  L^T x = w : U = the strip of the blocks' inverse transposes (block j at rows [512 j, 512 j + 512),
// row stride 512), w (in: right-hand side, already updated by every block right of nwide; consumed) and x (out) distinct.
// alpha_init_kernel: w = z (the augmented row of the factor, n entries) padded with zeros to np, x = 0: one launch where a memset
// and a device-to-device copy of the runtime were two (~5 us each on the eager alpha's critical path)
__global__ __launch_bounds__(256) void alpha_init_kernel(int64_t n, int64_t np, const double *__restrict__ z, double *__restrict__ w,
                                                         double *__restrict__ x)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < np) {
        w[i] = i < n ? z[i] : 0.0;
        x[i] = 0.0;
    }
}
int launch_alpha_init(hipStream_t st, int64_t n, int64_t np, const double *z, double *w, double *x)
{
    hipLaunchKernelGGL(alpha_init_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, n, np, z, w, x);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

// (Round 6 also let every step write its piece of x to the pinned host buffer of the eager alpha -- no device-to-host copy behind the
// last step: same-box A/B 0.2795 against 0.2795 ms of alpha per evaluation; removed.)
int launch_trsv_lt_wide(hipStream_t st, int64_t nwide, const double *L, int64_t ldl, const double *U, double *w, double *x, const double *w0)
{
    // w0 (may be null): the right-hand side is still where it was produced (the augmented row of the factor) and w is uninitialised:
    // the first step reads w0 and its update writes w = w0 - ..., so that no copy / init kernel precedes the first step
    if (nwide % TW_NB) {
        gpt_set_error("trsv_lt_wide: the extent must be a multiple of %d", TW_NB);
        return GPT_E_ARG;
    }
    if (w0 && nwide < 2 * TW_NB) {
        gpt_set_error("trsv_lt_wide: w0 needs at least two blocks");
        return GPT_E_ARG;
    }
    // (Round 4 also built a one-launch step -- update of step j and block solve of step j - 1 in one kernel, the eight workgroups that
    // own the next block's columns dispatched first and handing over through a counter: 0.36 against 0.26 ms at n = 8192 with the
    // inverses cached, and its wait was the one unbounded spin of the library.  Removed in round 5.)
    const double *wsrc = w0 ? w0 : w;
    for (int64_t j0 = nwide - TW_NB; j0 >= 0; j0 -= TW_NB) {
        GPT_TRY_RC_SOLVE(launch_gemv_n(st, TW_NB, TW_NB, U + j0 * TW_NB, TW_NB, wsrc + j0, x + j0));
        if (j0 > 0)
            hipLaunchKernelGGL(gemv_t_sub_kernel, dim3((unsigned)(j0 / 16)), dim3(512), 0, st, j0, L + j0 * ldl, ldl, x + j0, w, wsrc);
        wsrc = w;
    }
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

// y (m) = A (m x n, row-major) * x (n), and var[i] = kdiag[i] - sum_c V[i][c]^2 (the diagonal of Kss - V V^T without forming
// the M x M product).  One wave per row while there are many rows; with few rows (predict at a handful of points: 64 rows of
// 8192 entries) a whole workgroup per row, four loads in flight per thread -- one wave per row walked 128 dependent
// iterations, 42 us for either kernel at M = 64, N = 8192.  Fixed summation order for a given shape in both forms.
template <int WAVES, bool SQ>
__global__ __launch_bounds__(256) void rowred_kernel(int64_t m, int64_t n, const double *__restrict__ A, int64_t lda,
                                                     const double *__restrict__ x, double *__restrict__ y)
{
    // SQ: y[row] = x[row] - sum_c A[row][c]^2 ; else y[row] = sum_c A[row][c] x[c].  WAVES waves share a row.
    __shared__ double part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (WAVES == 4) ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * 4 + wave;
    const bool live = row < m;
    const int t = (WAVES == 4) ? (int)threadIdx.x : lane, T = WAVES * 64;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (live) {
        const double *r = A + row * lda;
        int64_t c = t;
        for (; c + 3 * T < n; c += 4 * T) {
            const double v0 = r[c], v1 = r[c + T], v2 = r[c + 2 * T], v3 = r[c + 3 * T];
            a0 = fma(v0, SQ ? v0 : x[c], a0);
            a1 = fma(v1, SQ ? v1 : x[c + T], a1);
            a2 = fma(v2, SQ ? v2 : x[c + 2 * T], a2);
            a3 = fma(v3, SQ ? v3 : x[c + 3 * T], a3);
        }
        for (; c < n; c += T) {
            const double v0 = r[c];
            a0 = fma(v0, SQ ? v0 : x[c], a0);
        }
    }
    double acc = (a0 + a1) + (a2 + a3);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (WAVES == 4) {
        if (lane == 0) part[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0 && live) {
            const double tot = (part[0] + part[1]) + (part[2] + part[3]);
            y[row] = SQ ? x[row] - tot : tot;
        }
    } else if (lane == 0 && live) {
        y[row] = SQ ? x[row] - acc : acc;
    }
}

int launch_gemv_n(hipStream_t st, int64_t m, int64_t n, const double *A, int64_t lda, const double *x, double *y)
{
    if (m <= 0) return GPT_OK;
    if (m < 2048) hipLaunchKernelGGL((rowred_kernel<4, false>), dim3((unsigned)m), dim3(256), 0, st, m, n, A, lda, x, y);
    else hipLaunchKernelGGL((rowred_kernel<1, false>), dim3((unsigned)((m + 3) / 4)), dim3(256), 0, st, m, n, A, lda, x, y);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

int launch_rowsumsq_sub(hipStream_t st, int64_t m, int64_t n, const double *V, int64_t ldv, const double *kdiag,
                        double *var_out)
{
    if (m <= 0) return GPT_OK;
    if (m < 2048) hipLaunchKernelGGL((rowred_kernel<4, true>), dim3((unsigned)m), dim3(256), 0, st, m, n, V, ldv, kdiag, var_out);
    else hipLaunchKernelGGL((rowred_kernel<1, true>), dim3((unsigned)((m + 3) / 4)), dim3(256), 0, st, m, n, V, ldv, kdiag, var_out);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

// Completes block row [c0, c0 + w) of a symmetric matrix whose lower triangle holds the values: A[a][b] = A[b][a] for a in
// the block row and b > a (32 x 32 tiles through LDS).  Used by gpt_predict behind each block column of the covariance SYRK.
__global__ __launch_bounds__(256) void mirror_rows_kernel(double *__restrict__ A, int64_t lda, int64_t c0, int64_t w, int64_t n)
{
    __shared__ double t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t a0 = c0 + (int64_t)blockIdx.y * 32;            // rows of the block row (destination rows)
    const int64_t b0 = a0 + (int64_t)blockIdx.x * 32;            // destination columns: from the diagonal tile on
    if (b0 >= n) return;
    for (int i = ty; i < 32; i += 8)                             // source tile: rows b0.., columns a0..
        if (b0 + i < n && a0 + tx < c0 + w) t[i][tx] = A[(b0 + i) * lda + a0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int64_t a = a0 + i, b = b0 + tx;
        if (a < c0 + w && b < n && b > a) A[a * lda + b] = t[tx][i];
    }
}

int launch_mirror_rows(hipStream_t st, double *A, int64_t lda, int64_t c0, int64_t w, int64_t n)
{
    if (w <= 0 || c0 >= n) return GPT_OK;
    dim3 grid((unsigned)((n - c0 + 31) / 32), (unsigned)((w + 31) / 32));
    hipLaunchKernelGGL(mirror_rows_kernel, grid, dim3(256), 0, st, A, lda, c0, w, n);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

// out[0] = sum_{i<n} (alpha_i^2 - W_ii): the noise term of the LML gradient over the observations (one workgroup, fixed order)
__global__ __launch_bounds__(256) void alpha_trace_kernel(const double *__restrict__ alpha, const double *__restrict__ W,
                                                          int64_t ldw, int64_t n, double *__restrict__ out)
{
    __shared__ double s0[4];
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) a += alpha[i] * alpha[i] - W[i * ldw + i];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
    if ((threadIdx.x & 63) == 0) s0[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((s0[0] + s0[1]) + s0[2]) + s0[3];
}

int launch_alpha_trace(hipStream_t st, const double *alpha, const double *W, int64_t ldw, int64_t n, double *out)
{
    hipLaunchKernelGGL(alpha_trace_kernel, dim3(1), dim3(256), 0, st, alpha, W, ldw, n, out);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

__global__ void diag_gather_kernel(const double *__restrict__ A, int64_t lda, int64_t n, double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = A[i * lda + i];
}

int launch_diag_gather(hipStream_t st, const double *A, int64_t lda, int64_t n, double *out)
{
    if (n <= 0) return GPT_OK;
    hipLaunchKernelGGL(diag_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, A, lda, n, out);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

// ---- helpers of the one-process-per-GPU path (gpt_dev_pad_block / gpt_dev_panel_scalars, include/gpt_hip.h) ----
__global__ void pad_block_kernel(double *__restrict__ A, int64_t lda, int64_t c0, int64_t nb, int64_t n_valid,
                                 int64_t n_pad, const double *__restrict__ dy, double big)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = n_valid + blockIdx.y;
    if (c >= nb || row >= n_pad) return;
    const int64_t gc = c0 + c;
    double v = 0.0;
    if (row == n_valid && gc < n_valid) v = dy[gc];
    else if (gc == row) v = (row == n_valid) ? big : 1.0;
    A[row * lda + c] = v;
}

int launch_pad_block(hipStream_t st, double *A, int64_t lda, int64_t c0, int64_t nb, int64_t n_valid, int64_t n_pad,
                     const double *dy, double big)
{
    if (n_pad <= n_valid || nb <= 0) return GPT_OK;
    dim3 grid((unsigned)((nb + 255) / 256), (unsigned)(n_pad - n_valid));
    hipLaunchKernelGGL(pad_block_kernel, grid, dim3(256), 0, st, A, lda, c0, nb, n_valid, n_pad, dy, big);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

__global__ __launch_bounds__(256) void panel_scalars_kernel(const double *__restrict__ P, int64_t ldp, int64_t w,
                                                            int64_t zrow, double *__restrict__ acc)
{
    __shared__ double s0[4], s1[4];
    double a = 0.0, b = 0.0;
    for (int64_t i = threadIdx.x; i < w; i += 256) {
        a += log(P[i * ldp + i]);
        if (zrow >= 0) {
            const double z = P[zrow * ldp + i];
            b = fma(z, z, b);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off);
        b += __shfl_down(b, off);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s0[wave] = a;
        s1[wave] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        acc[0] += ((s0[0] + s0[1]) + s0[2]) + s0[3];
        if (zrow >= 0) acc[1] += ((s1[0] + s1[1]) + s1[2]) + s1[3];
    }
}

int launch_panel_scalars(hipStream_t st, const double *P, int64_t ldp, int64_t w, int64_t zrow, double *acc)
{
    if (w <= 0) return GPT_OK;
    hipLaunchKernelGGL(panel_scalars_kernel, dim3(1), dim3(256), 0, st, P, ldp, w, zrow, acc);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}

// acc[0] += sum_{c < w} row[c]^2 (one workgroup, fixed order): the z.z part of ll (ref gaussian_process.py:1463) from the piece
// of the augmented row a rank of the 2-D block-cyclic engine holds (gptools_amd/dist.py GridLML)
__global__ __launch_bounds__(256) void row_sumsq_kernel(const double *__restrict__ row, int64_t w, double *__restrict__ acc)
{
    __shared__ double s1[4];
    double b = 0.0;
    for (int64_t i = threadIdx.x; i < w; i += 256) b = fma(row[i], row[i], b);
    for (int off = 32; off > 0; off >>= 1) b += __shfl_down(b, off);
    if ((threadIdx.x & 63) == 0) s1[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) acc[0] += ((s1[0] + s1[1]) + s1[2]) + s1[3];
}

int launch_row_sumsq(hipStream_t st, const double *row, int64_t w, double *acc)
{
    if (w <= 0) return GPT_OK;
    hipLaunchKernelGGL(row_sumsq_kernel, dim3(1), dim3(256), 0, st, row, w, acc);
    GPT_LAUNCH_CHECK();
    return GPT_OK;
}
