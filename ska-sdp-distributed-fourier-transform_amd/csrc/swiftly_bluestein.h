// SwiFTly on MI355X: transforms of NON-power-of-two length (171 of the 244 catalogue entries have yN = 3, 5, 7 or
// 9 times a power of two, xM in {320, 384, 448}, m in {160, 192, 224}) through Bluestein's chirp-z identity
//
//     X[k] = conj(c[k]) * sum_j ( v[j] conj(c[j]) ) c[k - j],      c[j] = exp(i pi j^2 / n),
//
// i.e. one cyclic convolution of power-of-two length L >= 2n - 1, evaluated with the workgroup FFT kernels:
//   blu_load   : in -> work[row][0..L): mapped load (window, zero-pad, shift, centred-index permutation) * conj(c[j])
//   FFT_L      : in place on the work rows (mapped row FFT, identity maps)
//   blu_mul    : work *= FFT_L(c wrapped)   (filter spectrum precomputed on the host in double precision)
//   iFFT_L     : in place
//   blu_store  : out (+)= work[row][k] * conj(c[k]) * scale * windows through the store map
// A general, correct fallback (about 10x the HBM traffic of a native kernel) -- the BASELINE configurations are all
// power-of-two and never come here.
#pragma once
#include "swiftly_rows.h"

namespace swf {

template <typename R>
struct BluArgs {
    RowsArgs<R> a;     // pointers, strides, maps, flags of the primitive (transform length n, not a power of two)
    int n, L;
    const cx<R>* chirp;  // c[j], j < n
    cx<R>* work;         // [nbatch][nrows][L]
};

template <typename R>
__device__ __forceinline__ int blu_mod(int v, int n) {
    v %= n;
    return v < 0 ? v + n : v;
}

template <typename R>
__global__ void blu_load_kernel(const BluArgs<R> B, const OffTab tab) {
    const RowsArgs<R>& A = B.a;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.z;
    if (j >= B.L) return;
    const int ld_a = (tab.use & 1) ? tab.ld_a[b] : A.ld.a;
    const int ld_c = (tab.use & 2) ? tab.ld_c[b] : A.ld.c;
    for (long long row = blockIdx.y; row < A.nrows; row += gridDim.y) {
        cx<R> val = {(R)0, (R)0};
        if (j < B.n) {
            long long in_row = row;
            if (A.rm_mod > 0) {
                int r1 = (int)row + A.rm_inner;
                if (r1 >= A.rm_mod) r1 -= A.rm_mod;
                r1 += A.rm_outer;
                if (r1 >= A.rm_full) r1 -= A.rm_full;
                in_row = r1;
            }
            if (A.in_rowmap) in_row = A.in_rowmap[in_row];
            const int ci = blu_mod<R>(j + B.n / 2, B.n);  // plain index -> centred index
            const int q = blu_mod<R>(ci + ld_a, B.n);
            if (q < A.ld.len && in_row >= 0) {
                int idx = q + ld_c;
                if (idx >= A.ld.mod) idx -= A.ld.mod;
                const cx<R> x = A.in[in_row * A.in_rs + (long long)b * A.in_bs + (long long)idx * A.in_cs];
                R w = (R)1;
                if (A.ld.win) w *= A.ld.win[q];
                if (A.ld.win2) w *= A.ld.win2[q];
                const cx<R> xv = {x.x * w, (A.conj_ld ? -x.y : x.y) * w};
                const cx<R> c = B.chirp[j];
                val = cx<R>{xv.x * c.x + xv.y * c.y, xv.y * c.x - xv.x * c.y};  // * conj(c)
            }
        }
        // stored rotated by L/2: the row kernels compute CENTRED transforms (fftshift . fft . ifftshift) with identity
        // maps, which on rotated data is the plain transform
        B.work[((long long)b * A.nrows + row) * B.L + ((j + (B.L >> 1)) & (B.L - 1))] = val;
    }
}

template <typename R>
__global__ void blu_mul_kernel(cx<R>* work, const cx<R>* __restrict__ spec, long long rows, int L) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= L) return;
    const cx<R> s = spec[j];
    const int jm = (j + (L >> 1)) & (L - 1);  // rotated storage
    for (long long row = blockIdx.y; row < rows; row += gridDim.y) {
        cx<R>* p = work + row * L + jm;
        *p = cmul(*p, s);
    }
}

template <typename R>
__global__ void blu_store_kernel(const BluArgs<R> B, const OffTab tab) {
    const RowsArgs<R>& A = B.a;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.z;
    if (k >= B.n) return;
    const int st_a = (tab.use & 4) ? tab.st_a[b] : A.st.a;
    const int st_c = (tab.use & 8) ? tab.st_c[b] : A.st.c;
    const int ck = blu_mod<R>(k + B.n / 2, B.n);
    const int d = blu_mod<R>(ck + st_a, B.n);
    if (d >= A.st.len) return;
    int idx = d + st_c;
    if (idx >= A.st.mod) idx -= A.st.mod;
    if (A.st_rowmap) {
        idx = A.st_rowmap[idx];
        if (idx < 0) return;
    }
    R w = A.scale;
    if (A.st.win) w *= A.st.win[(long long)b * A.st_win_bs + d];
    if (A.st.win2) w *= A.st.win2[d];
    const cx<R> c = B.chirp[k];
    for (long long row = blockIdx.y; row < A.nrows; row += gridDim.y) {
        const cx<R> y = B.work[((long long)b * A.nrows + row) * B.L + ((k + (B.L >> 1)) & (B.L - 1))];
        cx<R> v = {y.x * c.x + y.y * c.y, y.y * c.x - y.x * c.y};  // * conj(c)
        R wr = w;
        if (A.row_win) wr *= A.row_win[row];
        v.x *= wr;
        v.y *= A.conj_st ? -wr : wr;
        cx<R>* p = A.out + row * A.out_rs + (long long)b * A.out_bs + (long long)idx * A.out_cs;
        if (A.accumulate) {
            const cx<R> old = *p;
            v.x += old.x;
            v.y += old.y;
        }
        *p = v;
    }
}

}  // namespace swf
