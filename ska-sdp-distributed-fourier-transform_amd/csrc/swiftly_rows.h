// SwiFTly on MI355X: the generic "mapped row FFT" kernel.
//
//   value_in[i]  = in[row, src(i)] * win(i)         (or 0 outside the map)
//   X            = FFT or iFFT of value_in           (centred, length N)
//   out[row, dst(i)] (+)= X[i] * scale * win(i)      (or skipped)
//
// Every SwiFTly primitive that contains a transform is an instance of this
// with different AxisMaps (see swiftly_abi.hip for the table that maps the
// reference's core.py:189-484 onto it).
#pragma once
#include "swiftly_fft.h"

namespace swf {

// Maps a centred transform-domain index ci in [0, N) to a memory index:
//   q = (ci + a) mod N ; valid iff q < len ; idx = (q + c) mod mod
// win / win2 are optional real windows indexed by q.
template <typename R>
struct AxisMap {
    int a;
    int len;
    int c;
    int mod;
    const R* win;
    const R* win2;
};

template <typename R>
struct RowsArgs {
    const cx<R>* in;
    cx<R>* out;
    long long in_rs, out_rs;   // element stride between rows
    unsigned in_cs, out_cs;    // element stride along the transform axis; (full length)*cs < 2^32 (host-checked)
    int nrows;
    AxisMap<R> ld, st;
    R scale;
    int accumulate;  // out += instead of out =
    int conj_ld;     // conjugate on load   } inverse transform = conj(FFT(conj(x))) * scale,
    int conj_st;     // conjugate on store  } split over two kernels for four-step transforms
    int rowfast;     // lanes run over rows (use when in_rs == 1)
    // Decomposed (four-step) transforms: the transform index i of this kernel
    // is the plain full-length index  i*ld_mul + ld_add  on load and
    // i*st_mul + st_add  on store, where *_add = (outer row index) * *_addmul.
    // Plain (non-decomposed) use: mul = 1, addmul = 0, outer = 1.
    int full_logn;       // log2 of the full (power-of-two) transform length of a decomposed transform
    // Modulus of the load / store maps: 0 = 2^full_logn.  Set to n = Q * 2^k when this launch is one of the Q
    // power-of-two sub-transforms behind the radix-Q pass of swiftly_mixed.h; the plain output index is then
    // (e*st_mul + o*st_addmul) + st_add0 with st_mul, st_addmul premultiplied by Q and st_add0 = j.
    int full_n;
    int st_add0;
    // outer_group: enumerate the workgroups so that the `outer` sub-transforms of one row (which write INTERLEAVED
    // elements Q*e + j of the same output lines) run 8 block ids apart = on the same XCD at about the same time, where
    // its L2 merges their partial lines; block id = 8*(outer*rowblock8 + o) + xcd.  (Plain order: o = row / nrows.)
    int outer_group;
    int ld_mul, ld_addmul;
    int st_mul, st_addmul;
    int outer;           // rows are (inner, outer): row = inner*outer + o  -- see kernel
    long long in_os, out_os;  // element stride of the outer index
    const cx<R>* tw;     // exp(-2 pi i k / N), k < N  (this kernel's N)
    const cx<R>* tw_full;  // exp(-2 pi i k / 2^full_logn) for the four-step twiddle, or null
    int tw_on_store;     // multiply output i by tw_full[(i * o) ...] (four-step inter-pass twiddle)
    int raw_ld, raw_st;  // bypass centred-shift + map on load / store (intermediate buffers)
    // Optional modular input-row map (fuses a row gather such as
    // extract_from_facet along the other axis into the load):
    //   in_row = (rm_outer + ((row + rm_inner) mod rm_mod)) mod rm_full   when rm_mod > 0
    int rm_mod, rm_inner, rm_outer, rm_full;
    // Batch dimension (blockIdx.y): independent problems of identical shape at
    // in + b*in_bs / out + b*out_bs; per-item map offsets come from OffTab,
    // per-item store windows (masks) from st.win + b*st_win_bs.
    long long in_bs, out_bs;
    int nbatch;
    long long st_win_bs;
    // Optional row compaction tables (device, int32): BF_F keeps only the rows some requested subgrid
    // column reads.  st_rowmap: physical output index of logical store index (negative = not stored);
    // in_rowmap: physical input row of logical input row.
    const int* st_rowmap;
    const int* in_rowmap;
    // Optional real factor per ROW of the primitive applied on store (lets K1 pre-apply the window that
    // the next primitive would apply along the other axis: windows commute with transforms along the
    // orthogonal axis).
    const R* row_win;
};

// Per-batch-item overrides of the map offsets (passed by value as a kernel
// argument; kMaxBatch items per launch).
constexpr int kMaxBatch = 64;
struct OffTab {
    int use;  // bit 0: ld_a, bit 1: ld_c, bit 2: st_a, bit 3: st_c
    int ld_a[kMaxBatch], ld_c[kMaxBatch], st_a[kMaxBatch], st_c[kMaxBatch];
};

template <typename R>
struct kOneTable {
    static __device__ const R value;
};
template <typename R>
__device__ const R kOneTable<R>::value = (R)1;

template <class G, typename R>
__global__ __launch_bounds__(G::NT) void fft_rows_kernel(const RowsArgs<R> A, const OffTab tab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int P = G::P, T = G::T, RB = G::RB;
    const int tid = threadIdx.x;
    const bool rowfast = A.rowfast != 0;
    int t, rb;
    if (rowfast) {
        rb = tid % RB;
        t = tid / RB;
    } else {
        t = tid % T;
        rb = tid / T;
    }
    // rows are enumerated as (o, inner): consecutive rows share the outer index
    bool live;
    int o;
    long long row;
    if (A.outer_group) {  // uniform
        const long long x = blockIdx.x & 7, tb = blockIdx.x >> 3;
        o = (int)(tb % A.outer);
        row = ((tb / A.outer) * 8 + x) * RB + rb;
        live = row < A.nrows;
        if (!live) row = 0;
    } else {
        const long long grow = (long long)blockIdx.x * RB + rb;
        const long long total = (long long)A.nrows * A.outer;
        live = grow < total;
        o = live ? (int)(grow / A.nrows) : 0;
        row = live ? grow % A.nrows : 0;
    }
    const int FS = 1 << A.full_logn;            // four-step twiddle period
    const int FN = A.full_n > 0 ? A.full_n : FS;  // modulus of the maps (any even length)
    auto wrapn = [FN](int v) { return v >= FN ? v - FN : v; };
    long long in_row = row;
    if (A.rm_mod > 0) {
        int r1 = (int)row + A.rm_inner;
        if (r1 >= A.rm_mod) r1 -= A.rm_mod;
        r1 += A.rm_outer;
        if (r1 >= A.rm_full) r1 -= A.rm_full;
        in_row = r1;
    }
    if (A.in_rowmap) in_row = A.in_rowmap[in_row];
    const bool absent = in_row < 0;  // row absent from a compacted input (map entry < 0): reads as zeros
    if (absent) in_row = 0;
    const int b = blockIdx.y;
    const cx<R>* __restrict__ in = A.in + in_row * A.in_rs + (long long)o * A.in_os + (long long)b * A.in_bs;
    cx<R>* __restrict__ out = A.out + row * A.out_rs + (long long)o * A.out_os + (long long)b * A.out_bs;
    const int ld_a = (tab.use & 1) ? tab.ld_a[b] : A.ld.a;
    const int ld_c = (tab.use & 2) ? tab.ld_c[b] : A.ld.c;
    const int st_a = (tab.use & 4) ? tab.st_a[b] : A.st.a;
    const int st_c = (tab.use & 8) ? tab.st_c[b] : A.st.c;
    const R* __restrict__ st_win = A.st.win ? A.st.win + (long long)b * A.st_win_bs : nullptr;
    const R csign_ld = A.conj_ld ? (R)-1 : (R)1;
    const R csign_st = A.conj_st ? (R)-1 : (R)1;

    // Loads are branch-free: out-of-map elements read a clamped (always valid)
    // address and are zeroed through the window factor.  With no data-dependent
    // control flow in the loop the P loads of a thread stay in flight together;
    // a predicated version exposes one memory latency per element.
    cx<R> x[P];
    const int ld_add = o * A.ld_addmul;
    const R* __restrict__ win1 = A.ld.win ? A.ld.win : &kOneTable<R>::value;
    const R* __restrict__ win2 = A.ld.win2 ? A.ld.win2 : &kOneTable<R>::value;
    const int w1s = A.ld.win ? 1 : 0, w2s = A.ld.win2 ? 1 : 0;
    const R live_f = (live && !absent) ? (R)1 : (R)0;
    if (A.raw_ld) {  // uniform over the launch
        static_for<0, P>([&](auto vI) {
            constexpr int v = decltype(vI)::value;
            x[v] = in[(size_t)((unsigned)(t + v * T) * A.in_cs)];
        });
        static_for<0, P>([&](auto vI) {
            constexpr int v = decltype(vI)::value;
            x[v].x *= live_f;
            x[v].y *= live_f * csign_ld;
        });
    } else {
        static_for<0, P>([&](auto vI) {
            constexpr int v = decltype(vI)::value;
            const int i = t + v * T;
            const int pi = i * A.ld_mul + ld_add;        // plain full-length index
            const int ci = wrapn(pi + (FN >> 1));  // centred index
            const int q = wrapn(ci + ld_a);
            const bool ok = q < A.ld.len;
            const int qs = ok ? q : 0;
            int idx = qs + ld_c;
            if (idx >= A.ld.mod) idx -= A.ld.mod;
            cx<R> val = in[(size_t)((unsigned)idx * A.in_cs)];
            R w = win1[qs * w1s] * win2[qs * w2s];
            w = ok ? w * live_f : (R)0;
            val.x *= w;
            val.y *= w * csign_ld;
            x[v] = val;
        });
    }

    const int st_add = o * A.st_addmul + A.st_add0;
    fft_phases<G, R, 0>(x, t, rb, rowfast, smem, A.tw, [&](int e, cx<R> v) {
        if (!live) return;
        if (A.tw_on_store) {
            // four-step twiddle exp(-/+ 2 pi i * e * o / FN)
            cx<R> w = A.tw_full[((unsigned)e * (unsigned)o) & (unsigned)(FS - 1)];
            v = cmul(v, w);
        }
        v.x *= A.scale;
        v.y *= A.scale * csign_st;
        if (A.raw_st) {
            cx<R>* p = out + (size_t)((unsigned)e * A.out_cs);
            if (A.accumulate) {
                cx<R> old = *p;
                v.x += old.x;
                v.y += old.y;
            }
            *p = v;
            return;
        }
        const int pk = e * A.st_mul + st_add;
        const int ck = wrapn(pk + (FN >> 1));
        const int d = wrapn(ck + st_a);
        if (d < A.st.len) {
            int idx = d + st_c;
            if (idx >= A.st.mod) idx -= A.st.mod;
            if (A.st_rowmap) {
                idx = A.st_rowmap[idx];
                if (idx < 0) return;
            }
            R w = (R)1;
            if (st_win) w = st_win[d];
            if (A.st.win2) w *= A.st.win2[d];
            if (A.row_win) w *= A.row_win[row];
            v.x *= w;
            v.y *= w;
            cx<R>* p = out + (size_t)((unsigned)idx * A.out_cs);
            if (A.accumulate) {
                cx<R> old = *p;
                v.x += old.x;
                v.y += old.y;
            }
            *p = v;
        }
    });
}

// Engine configuration per transform length ---------------------------------
// float : LOGP = ceil(LOGN / ceil(LOGN/4)) (radix <= 16), N = 32768 uses radix 32
//         with the split re/im exchange (256 KiB of points > 160 KiB LDS).
// double: radix 8 throughout (register budget), N <= 8192.
constexpr int logp_for(int logn, bool dbl) {
    if (dbl) return logn < 3 ? logn : 3;
    if (logn <= 4) return logn < 3 ? logn : 3;
    if (logn == 15) return 5;
    int k = (logn + 3) / 4;
    return (logn + k - 1) / k;
}
constexpr int nt_for(int logn, int logp) {
    int t = 1 << (logn - logp);
    return t > 256 ? t : 256;
}

template <typename R, int LOGN>
struct GeoFor {
    static constexpr bool DBL = sizeof(R) == 8;
    static constexpr int LOGP = logp_for(LOGN, DBL);
    static constexpr int NT = nt_for(LOGN, LOGP);
    static constexpr bool SPLIT = (!DBL && LOGN == 15);
    using type = Geo<R, LOGN, LOGP, NT, SPLIT>;
};

constexpr int kMinLogN = 3;
constexpr int kMaxLogNFloat = 15;
constexpr int kMaxLogNDouble = 13;

// implemented in fft_rows_f32.hip / fft_rows_f64.hip
int launch_fft_rows(int logn, const RowsArgs<float>& a, const OffTab& tab, hipStream_t s);
int launch_fft_rows(int logn, const RowsArgs<double>& a, const OffTab& tab, hipStream_t s);
int init_fft_rows_f32();
int init_fft_rows_f64();

}  // namespace swf
