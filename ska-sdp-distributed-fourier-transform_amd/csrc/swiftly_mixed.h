// SwiFTly on MI355X: transforms of length n = Q * 2^k, Q in {3, 5, 7, 9} (171 of the reference's 244 catalogue entries:
// yN = 3, 5, 7 or 9 times a power of two; a few also xM = 320 / 384 / 448 and m = 160 / 192 / 224), natively.
//
// One decimation-in-frequency step of radix Q in front of the power-of-two kernels.  With M = 2^k, input index
// y = y2 + M*y1 (y1 < Q) and output index k = Q*k2 + j (j < Q):
//
//     z_j[y2]      = W_n^(y2 j) * sum_{y1 < Q} x[y2 + M y1] W_Q^(y1 j)          this file: mixed_radix_pass_kernel
//     X[Q k2 + j]  = sum_{y2 < M} z_j[y2] W_M^(y2 k2)                            Q mapped row FFTs of length M
//
// The pass is element-wise over y2 (coalesced along whichever direction is contiguous) and carries the whole LOAD side
// of the primitive -- centred-index permutation, cyclic offset, zero padding, windows, conjugation, row maps, per-item
// offsets (swiftly_rows.h) -- so the power-of-two kernels behind it read a plain scratch; their STORE side carries the
// store map with plain output index Q*e + j (RowsArgs::st_mul / st_add0, modulus RowsArgs::full_n).  Cost: the data
// crosses HBM twice more than in a fused kernel (scratch written + read), against 3 transforms of 2.7-4x the length
// plus three element-wise passes for the Bluestein path (swiftly_bluestein.h) it replaces -- and no length limit at
// 32768: every catalogue length up to 7 * 2^13 = 57344 has M <= 16384.
#pragma once
#include "swiftly_rows.h"

namespace swf {

constexpr int kMixedMaxQ = 9;

template <typename R>
struct MixedArgs {
    int Q, M, n;          // n = Q * M
    cx<R> wq[kMixedMaxQ];  // exp(-2 pi i r / Q), r < Q
    const cx<R>* tw_n;    // exp(-2 pi i r / n), r < n
    cx<R>* scratch;
    // scratch index of (batch b, row, j, y2):  b*s_b + row*s_row + j*s_j + y2*s_y
    long long s_b, s_row, s_j, s_y;
};

__device__ __forceinline__ int wrap_n(int v, int n) { return v >= n ? v - n : v; }

template <typename R, int Q>
__device__ __forceinline__ void mixed_radix_point(const RowsArgs<R>& A, const OffTab& tab, const MixedArgs<R>& X, const long long row,
                                                  const int y2, const int b) {
    long long in_row = row;
    if (A.rm_mod > 0) {
        int r1 = (int)row + A.rm_inner;
        if (r1 >= A.rm_mod) r1 -= A.rm_mod;
        r1 += A.rm_outer;
        if (r1 >= A.rm_full) r1 -= A.rm_full;
        in_row = r1;
    }
    if (A.in_rowmap) in_row = A.in_rowmap[in_row];
    const bool absent = in_row < 0;  // row absent from a compacted input: reads as zeros
    if (absent) in_row = 0;
    const cx<R>* __restrict__ in = A.in + in_row * A.in_rs + (long long)b * A.in_bs;
    const int ld_a = (tab.use & 1) ? tab.ld_a[b] : A.ld.a;
    const int ld_c = (tab.use & 2) ? tab.ld_c[b] : A.ld.c;
    const int n = X.n;
    cx<R> x[Q];
#pragma unroll
    for (int y1 = 0; y1 < Q; y1++) {
        const int pi = y2 + X.M * y1;            // plain index
        const int ci = wrap_n(pi + (n >> 1), n);  // centred index
        const int q = wrap_n(ci + ld_a, n);
        const bool ok = q < A.ld.len && !absent;
        const int qs = ok ? q : 0;
        int idx = qs + ld_c;
        if (idx >= A.ld.mod) idx -= A.ld.mod;
        cx<R> val = in[(size_t)((unsigned)idx * A.in_cs)];
        R w = ok ? (R)1 : (R)0;
        if (A.ld.win) w *= A.ld.win[qs];
        if (A.ld.win2) w *= A.ld.win2[qs];
        x[y1] = cx<R>{val.x * w, (A.conj_ld ? -val.y : val.y) * w};
    }
    cx<R>* __restrict__ out = X.scratch + (long long)b * X.s_b + row * X.s_row + (long long)y2 * X.s_y;
#pragma unroll
    for (int j = 0; j < Q; j++) {
        cx<R> z = x[0];
#pragma unroll
        for (int y1 = 1; y1 < Q; y1++) {
            const cx<R> w = X.wq[(y1 * j) % Q];  // compile-time index after unrolling
            z.x += x[y1].x * w.x - x[y1].y * w.y;
            z.y += x[y1].x * w.y + x[y1].y * w.x;
        }
        if (j > 0) z = cmul(z, X.tw_n[(unsigned)y2 * (unsigned)j]);  // y2 * j < n
        out[(long long)j * X.s_j] = z;
    }
}

template <typename R, int Q>
__global__ __launch_bounds__(256) void mixed_radix_pass_kernel(const RowsArgs<R> A, const OffTab tab, const MixedArgs<R> X) {
    // lanes along the contiguous direction: y2 for transforms along the contiguous axis, rows otherwise
    const bool rowfast = A.rowfast != 0;
    const long long ifast = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.z;
    const long long nslow = rowfast ? X.M : A.nrows;
    if (ifast >= (rowfast ? (long long)A.nrows : (long long)X.M)) return;
    for (long long islow = blockIdx.y; islow < nslow; islow += gridDim.y) mixed_radix_point<R, Q>(A, tab, X, rowfast ? ifast : islow, (int)(rowfast ? islow : ifast), b);
}

// The radix-Q pass with the GATHER-SUM load of the backward pass (swiftly_colpass.h, template GS: add_to_facet along
// the strided axis fused into the load of finish_facet, api_helper.py:142-179): logical row idx of the padded axis is the
// sum of up to two source rows  rowmap[idx], rowmap[n + idx]  (negative = none), each encoded  chunk << 20 | row  and
// read at  in + c_base[chunk] + facet * c_fs[chunk] + row * in_pitch  (chunks = pieces of a multi-GPU receive buffer).
// Lanes run over the m columns of the contributions; scratch[facet][j][y2][column].
constexpr int kMixedGsChunks = 16;
constexpr int kMixedGsRowBits = 20;
struct MixedGsArgs {
    const cx<float>* in;
    unsigned in_pitch;
    const int* rowmap;  // [2][n]
    int ncols;
    long long c_base[kMixedGsChunks], c_fs[kMixedGsChunks];
};

template <int Q>
__global__ __launch_bounds__(256) void mixed_gs_pass_kernel(const MixedGsArgs G, const MixedArgs<float> X) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.z;
    if (col >= G.ncols) return;
    const int n = X.n;
    constexpr int RM = (1 << kMixedGsRowBits) - 1;
    for (int y2 = blockIdx.y; y2 < X.M; y2 += gridDim.y) {  // uniform per workgroup: the row tables are scalar loads
        cx<float> x[Q];
#pragma unroll
        for (int y1 = 0; y1 < Q; y1++) {
            const int pi = y2 + X.M * y1;
            const int idx = wrap_n(pi + (n >> 1), n);  // centred index = logical row (identity load map)
            cx<float> val = {0.f, 0.f};
            const int r1 = G.rowmap[idx], r2 = G.rowmap[n + idx];
            if (r1 >= 0) {
                const int c = r1 >> kMixedGsRowBits;
                val = G.in[G.c_base[c] + (long long)f * G.c_fs[c] + (long long)(r1 & RM) * G.in_pitch + col];
            }
            if (r2 >= 0) {
                const int c = r2 >> kMixedGsRowBits;
                const cx<float> w = G.in[G.c_base[c] + (long long)f * G.c_fs[c] + (long long)(r2 & RM) * G.in_pitch + col];
                val.x += w.x;
                val.y += w.y;
            }
            x[y1] = val;
        }
        cx<float>* __restrict__ out = X.scratch + (long long)f * X.s_b + (long long)y2 * X.s_y + col;
#pragma unroll
        for (int j = 0; j < Q; j++) {
            cx<float> z = x[0];
#pragma unroll
            for (int y1 = 1; y1 < Q; y1++) {
                const cx<float> w = X.wq[(y1 * j) % Q];
                z.x += x[y1].x * w.x - x[y1].y * w.y;
                z.y += x[y1].x * w.y + x[y1].y * w.x;
            }
            if (j > 0) z = cmul(z, X.tw_n[(unsigned)y2 * (unsigned)j]);
            out[(long long)j * X.s_j] = z;
        }
    }
}

int launch_mixed_gs_pass(int Q, const MixedGsArgs& g, const MixedArgs<float>& x, int nfacets, hipStream_t s);
int launch_mixed_pass(int Q, const RowsArgs<float>& a, const OffTab& tab, const MixedArgs<float>& x, int nbatch, hipStream_t s);
int launch_mixed_pass(int Q, const RowsArgs<double>& a, const OffTab& tab, const MixedArgs<double>& x, int nbatch, hipStream_t s);

}  // namespace swf
