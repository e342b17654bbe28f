// Micro-benchmark of 16x16 pivot-block variants (one wave, block in LDS), cycles per block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double bcast_lane(double v, int srclane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}
template <int C> __device__ __forceinline__ double swz_bcast(double v) {
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), (C << 5));
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), (C << 5));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void sqrt_rsqrt(double d, double &s, double &inv) {
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double r = fma(-g, h, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-g, h, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    s = g; inv = h + h;
}
template <int J, int C, int V> struct Upd {
    static __device__ __forceinline__ void run(double (&a)[16], double (&x)[16]) {
        double l;
        if (V == 3) l = swz_bcast<C>(a[J]); else l = bcast_lane(a[J], C);
        a[C] = fma(-a[J], l, a[C]);
        if (V != 1) x[C] = fma(-l, x[J], x[C]);
        if (V == 2 && ((C & 3) == 3)) asm volatile("" ::: "memory");
        if (C + 1 < 16) Upd<J, (C + 1 < 16 ? C + 1 : 15), V>::run(a, x);
    }
};
template <int J, int V> __device__ __forceinline__ void col(double (&a)[16], double (&x)[16], int row) {
    double d = bcast_lane(a[J], J);
    if (!(d > 0.0)) d = 1.0;
    double sq, inv; sqrt_rsqrt(d, sq, inv);
    a[J] = (row == J) ? sq : a[J] * inv;
    x[J] = x[J] * inv;
    if (J + 1 < 16) Upd<J, (J + 1 < 16 ? J + 1 : 15), V>::run(a, x);
}
template <int V> __global__ __launch_bounds__(64) void pivot_kernel(const double* A, double* Lout, double* Iout, long long* cyc, int reps) {
    __shared__ double S[16][18];
    const int lane = threadIdx.x, row = lane & 15;
    if (lane < 16) for (int c = 0; c < 16; c++) S[row][c] = A[row * 16 + c];
    __syncthreads();
    double a[16], x[16];
    long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; rep++) {
#pragma unroll
        for (int c = 0; c < 16; c++) { a[c] = S[row][c]; x[c] = (c == row) ? 1.0 : 0.0; }
        col<0, V>(a, x, row); col<1, V>(a, x, row); col<2, V>(a, x, row); col<3, V>(a, x, row);
        col<4, V>(a, x, row); col<5, V>(a, x, row); col<6, V>(a, x, row); col<7, V>(a, x, row);
        col<8, V>(a, x, row); col<9, V>(a, x, row); col<10, V>(a, x, row); col<11, V>(a, x, row);
        col<12, V>(a, x, row); col<13, V>(a, x, row); col<14, V>(a, x, row); col<15, V>(a, x, row);
        if (rep + 1 < reps) { if (lane < 16) S[row][0] += 1e-9 * a[15]; }   // keep the loop alive
        __syncthreads();
    }
    long long t1 = __builtin_readcyclecounter();
    if (lane < 16) for (int c = 0; c < 16; c++) { Lout[row * 16 + c] = (c <= row) ? a[c] : 0.0; Iout[c * 16 + row] = x[c]; }
    if (lane == 0) cyc[0] = (t1 - t0) / reps;
}

// ---- variant 5: row-per-lane, scalar shortcut for the next pivot, one Newton step ----
__device__ __forceinline__ double rsqrt1(double d) {
    const double y0 = __builtin_amdgcn_rsq(d);
    const double t = d * y0, h = 0.5 * y0;
    const double u = fma(-t, y0, 1.0);
    return fma(h, u, y0);
}
template <int J> struct Upd5 {
    template <int C> static __device__ __forceinline__ void run(double (&a)[16], double (&x)[16]) {
        const double l = bcast_lane(a[J], C);
        a[C] = fma(-a[J], l, a[C]);
        x[C] = fma(-l, x[J], x[C]);
        if constexpr (C + 1 < 16) run<C + 1>(a, x);
    }
};
template <int J> __device__ __forceinline__ void col5(double (&a)[16], double (&x)[16], double &d) {
    if (!(d > 0.0)) d = 1.0;
    const double inv = rsqrt1(d);
    if constexpr (J + 1 < 16) {
        const double p = bcast_lane(a[J], J + 1), q = bcast_lane(a[J + 1], J + 1);
        const double pl = p * inv;
        d = fma(-pl, pl, q);
    }
    a[J] *= inv;
    x[J] *= inv;
    if constexpr (J + 1 < 16) Upd5<J>::template run<J + 1>(a, x);
}
__global__ __launch_bounds__(512) void pivot_kernel5(const double* A, double* Lout, double* Iout, long long* cyc, int reps) {
    __shared__ double S[16][18];
    const int lane = threadIdx.x & 63, row = lane & 15;
    if (threadIdx.x >= 64) { __syncthreads(); for (int rep = 0; rep < reps; rep++) __syncthreads(); return; }
    if (lane < 16) for (int c = 0; c < 16; c++) S[row][c] = A[row * 16 + c];
    __syncthreads();
    double a[16], x[16];
    long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; rep++) {
#pragma unroll
        for (int c = 0; c < 16; c++) { a[c] = S[row][c]; x[c] = (c == row) ? 1.0 : 0.0; }
        double d = bcast_lane(a[0], 0);
        col5<0>(a, x, d); col5<1>(a, x, d); col5<2>(a, x, d); col5<3>(a, x, d); col5<4>(a, x, d); col5<5>(a, x, d); col5<6>(a, x, d); col5<7>(a, x, d);
        col5<8>(a, x, d); col5<9>(a, x, d); col5<10>(a, x, d); col5<11>(a, x, d); col5<12>(a, x, d); col5<13>(a, x, d); col5<14>(a, x, d); col5<15>(a, x, d);
        if (rep + 1 < reps) { if (lane < 16) S[row][0] += 1e-9 * a[15]; }
        __syncthreads();
    }
    long long t1 = __builtin_readcyclecounter();
    if (lane < 16) for (int c = 0; c < 16; c++) { Lout[row * 16 + c] = (c <= row) ? a[c] : 0.0; Iout[c * 16 + row] = x[c]; }
    if (lane == 0) cyc[0] = (t1 - t0) / reps;
}

// ---- variant 4: all 64 lanes: lane (i = l&15, q = l>>4) holds A[i][4t+q] and X[4t+q][i], t = 0..3 ----
__device__ __forceinline__ double shfl_d(double v, int src) {
    const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int J> __device__ __forceinline__ void col4(double (&a)[4], double (&x)[4], double &d, int i, int q) {
    constexpr int TJ = J / 4, QJ = J % 4;
    if (!(d > 0.0)) d = 1.0;
    const double inv = rsqrt1(d);
    if constexpr (J + 1 < 16) {
        const double p = bcast_lane(a[TJ], (J + 1) + 16 * QJ);
        const double qq = bcast_lane(a[(J + 1) / 4], (J + 1) + 16 * ((J + 1) % 4));
        const double pl = p * inv;
        d = fma(-pl, pl, qq);
    }
    const double sc = a[TJ] * inv, sx = x[TJ] * inv;          // valid in quarter QJ
    if (q == QJ) { a[TJ] = sc; x[TJ] = sx; }
    if constexpr (J + 1 < 16) {
        const double li = shfl_d(sc, i + 16 * QJ);             // L[i][j]
        const double xj = shfl_d(sx, i + 16 * QJ);             // X[j][i]
#pragma unroll
        for (int t = TJ; t < 4; t++) {
            const int c = 4 * t + q;
            const double lc = shfl_d(sc, c + 16 * QJ);         // L[c][j]
            const double na = fma(-li, lc, a[t]);
            const double nx = fma(-lc, xj, x[t]);
            if (t > TJ || q > QJ) { a[t] = na; x[t] = nx; }
        }
    }
}
__global__ __launch_bounds__(64) void pivot_kernel4(const double* A, double* Lout, double* Iout, long long* cyc, int reps) {
    __shared__ double S[16][18];
    const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
    if (lane < 16) for (int c = 0; c < 16; c++) S[i][c] = A[i * 16 + c];
    __syncthreads();
    double a[4], x[4];
    long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; rep++) {
#pragma unroll
        for (int t = 0; t < 4; t++) { a[t] = S[i][4 * t + q]; x[t] = (4 * t + q == i) ? 1.0 : 0.0; }
        double d = bcast_lane(a[0], 0);
        col4<0>(a, x, d, i, q); col4<1>(a, x, d, i, q); col4<2>(a, x, d, i, q); col4<3>(a, x, d, i, q);
        col4<4>(a, x, d, i, q); col4<5>(a, x, d, i, q); col4<6>(a, x, d, i, q); col4<7>(a, x, d, i, q);
        col4<8>(a, x, d, i, q); col4<9>(a, x, d, i, q); col4<10>(a, x, d, i, q); col4<11>(a, x, d, i, q);
        col4<12>(a, x, d, i, q); col4<13>(a, x, d, i, q); col4<14>(a, x, d, i, q); col4<15>(a, x, d, i, q);
        if (rep + 1 < reps) { if (lane == 0) S[0][0] += 1e-9 * a[3]; }
        __syncthreads();
    }
    long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int t = 0; t < 4; t++) { const int c = 4 * t + q; Lout[i * 16 + c] = (c <= i) ? a[t] : 0.0; Iout[c * 16 + i] = x[t]; }
    if (lane == 0) cyc[0] = (t1 - t0) / reps;
}

int main() {
    std::vector<double> A(256), L(256), I(256);
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) A[i * 16 + j] = (i == j ? 4.0 : 0) + cos(0.3 * i) * cos(0.3 * j) + 0.1 * cos(0.7 * (i + 1) * (j + 1));
    double *dA, *dL, *dI; long long* dc;
    hipMalloc(&dA, 2048); hipMalloc(&dL, 2048); hipMalloc(&dI, 2048); hipMalloc(&dc, 8);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
    for (int v = 4; v < 8; v++) {
        if (v == 0) pivot_kernel<0><<<1, 64>>>(dA, dL, dI, dc, 200);
        if (v == 1) pivot_kernel<1><<<1, 64>>>(dA, dL, dI, dc, 200);
        if (v == 2) pivot_kernel<2><<<1, 64>>>(dA, dL, dI, dc, 200);
        if (v == 3) pivot_kernel<3><<<1, 64>>>(dA, dL, dI, dc, 200);
        if (v == 4) pivot_kernel4<<<1, 64>>>(dA, dL, dI, dc, 200);
        if (v == 5) pivot_kernel5<<<1, 64>>>(dA, dL, dI, dc, 200);
        if (v == 6) pivot_kernel5<<<1, 512>>>(dA, dL, dI, dc, 1);
        if (v == 7) pivot_kernel5<<<1, 512>>>(dA, dL, dI, dc, 8);
        hipDeviceSynchronize();
        long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        hipMemcpy(L.data(), dL, 2048, hipMemcpyDeviceToHost); hipMemcpy(I.data(), dI, 2048, hipMemcpyDeviceToHost);
        // check L L^T = A(lower) for rep-1 perturbed... just check L*I = identity
        double err = 0; for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 16; k++) s += L[i * 16 + k] * I[k * 16 + j]; err = fmax(err, fabs(s - (i == j))); }
        double e2 = 0; for (int i = 0; i < 16; i++) for (int j = 0; j <= i; j++) { double s = 0; for (int k = 0; k < 16; k++) s += L[i * 16 + k] * L[j * 16 + k]; e2 = fmax(e2, fabs(s - A[i * 16 + j])); }
        printf("variant %d: %lld cycles per pivot block; |L*inv - I| = %.2e  |LL^T - A| = %.2e\n", v, c, err, e2);
    }
    return 0;
}
