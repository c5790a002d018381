// SwiFTly on MI355X: lean single-workgroup transforms along the CONTIGUOUS axis
// for long rows (N = 8192 .. 32768, complex64): K2 (extract_column =
// row gather + prepare_facet axis 1), the axis-1 finish_facet of the backward
// pass, and any other long contiguous-axis primitive.
//
// One workgroup = one row: the row base pointers are wave-uniform (SGPR base +
// 32-bit lane offset addressing), the lane mapping is fixed at compile time,
// all loads of a lane are issued before the first use, and the window loads
// are batched the same way.  Register budget: P complex points + P window
// values, no spills at 128 VGPRs.
#pragma once
#include <mutex>
#include <vector>

#include "swiftly_fft.h"

// groups of three loads the W4 instances of the band row kernel request before the first one is consumed (0: the
// compiler's own schedule)
#ifndef SWF_K1_PIPE
#define SWF_K1_PIPE 8
#endif
// band store of the long-row kernel through a range-checked buffer store (1, r6) or the wave-uniform all-inside / masked pair
// (0, r4b): bit-identical, K1 1.457 - 1.493 against 1.467 - 1.505 ms per facet (same box, interleaved), 484 -> ~330 scalar
// instructions per wave
#ifndef SWF_K1_BUFSTORE
#define SWF_K1_BUFSTORE 1
#endif

namespace swf {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// The band row kernel loads through BUFFER (range-checked) accesses.  Measured (r2, same box, K1 per pass): global loads +
// exec-masked stores 20.2 ms (at 86 spilled VGPRs in some variants: 51 ms -- the kernel sits at the 128-VGPR limit), buffer
// loads 20.3 ms with ~10 % fewer instructions and no spills (kept), buffer stores too 20.7 ms (not kept).

struct RowPassArgs {
    const cx<float>* in;
    cx<float>* out;
    long long in_pitch, out_pitch;  // elements between rows
    int nrows;
    // optional modular input-row map (extract_from_facet along the other axis)
    int rm_mod, rm_inner, rm_outer, rm_full;
    const int* in_rowmap;  // optional: physical input row of logical input row (compacted BF_F)
    // load map:  q = (ci + ld_a) mod N ; valid q < ld_len ; src = (q + ld_c) mod ld_mod ; window ld_win[q]
    int ld_a, ld_len, ld_c, ld_mod;
    const float* ld_win;
    // store map: d = (ck + st_a) mod N ; valid d < st_len ; dst = (d + st_c) mod st_mod ; windows st_win[d]*st_win2[d]
    int st_a, st_len, st_c, st_mod;
    const float* st_win;
    const float* st_win2;
    const cx<float>* tw;
    float scale;
    int conj_ld, conj_st, accumulate;
    // -- row_pass_band_kernel only --------------------------------------------------------------------
    // optional real factor per ROW folded into the scale (window of the OTHER axis, pre-applied: windows
    // commute with transforms along the orthogonal axis)
    const float* row_win;
    // band store: only the outputs whose centred index ck lies in the cyclic range [band_start, band_start +
    // band_len) are kept, in PARITY-SPLIT order: d = (ck - band_start) mod N lands at column
    // (d & 1) * band_half + (d >> 1).  (The two workgroups of a row produce the even / odd outputs; with this
    // layout each of them writes one contiguous run instead of every other element.)  band_len = 0: plain row.
    int band_start, band_len, band_half;
    // zero-segment skipping (NSEG instances of row_pass_band_kernel, set by the launcher): the transform input is taken
    // cyclically rotated by seg_rot segments so that every segment that can hold data is one of the first NSEG ones;
    // the outputs get the compensating phase W_nseg^(seg_rot k)
    int seg_rot;
    // W4 instances (r5, forward K1): the load window of the lane's segments re-laid out so that ONE 16-byte load fetches
    // the window pairs of TWO segments (Win4Cache below builds it per (window, facet offset, segment count)): table entry
    // (p * T + t) holds {w[s0][2t], w[s0][2t+1], w[s1][2t], w[s1][2t+1]} for the p-th segment pair (s0, s1) of the load
    // loop, zeros where the padded row has no data
    const float* ld_win4;
    // ... and the compact twiddle sections of geometries with COMPACT_TW (Win4Cache::compact_tw; swiftly_fft.h, preload_compact)
    const cx<float>* twc;
    // WINDOW-ROWS store (r6, whole-row kernel swiftly_rowwhole.h, WIN instances): one workgroup owns BOTH output parities of a
    // row, so the complete contiguous-axis half of add_to_subgrid runs in its epilogue (what swiftly_hip_finish_axis1_rows does
    // per wave in a pass of its own): the band of the row is staged in LDS (the exchange buffer, free after the last gather),
    // and for every window w   out[row][w * m + parity-split position of (kk - s_w) mod m] =
    // Fn[kk] cfft_m(window w)[(kk + win_sp) mod m]   -- the layout finish_axis1_rows produces for wave w.
    int win_full;                 // 1: this form
    const int* win_d;             // device table, nwin entries: (first logical column of window w - band_start) mod N
    int nwin, win_logm;           // windows of 2^win_logm columns, every one inside the band
    long long win_pitch;          // elements between the rows of consecutive windows in `out` (m: side by side in a row)
    int win_sp;                   // s'1 = floor(facet_off1 * xM / N) mod m of this facet
    const float* win_fn;          // Fn[m]
    const cx<float>* win_tw_m;    // twiddle table of length m and its compact sections for m / 64 points per lane
    const cx<float>* win_twc_m;
};

// physical column of logical (centred) column ck in a parity-split band buffer, or -1
__host__ __device__ inline int band_column(int ck, int n, int start, int len, int half) {
    const int d = (ck - start) & (n - 1);
    return d < len ? (d & 1) * half + (d >> 1) : -1;
}

template <int LOGN_, int LOGP_, bool SPLIT_, bool PAD_ = true>
struct RGeo {
    static constexpr int LOGN = LOGN_, LOGP = LOGP_;
    static constexpr bool SPLIT = SPLIT_;
    static constexpr int N = 1 << LOGN, P = 1 << LOGP, T = N / P, NT = T;
    static constexpr bool WAVE_ROWS = false;  // a row spans the whole workgroup
    static constexpr int RB = 1;
    static constexpr int ELEM = SPLIT ? 4 : 8;
    // PAD_ = false: exactly N elements (64 KiB for the 16384-point split exchange) so that TWO workgroups fit
    // the 128 KiB the dispatcher hands out per CU for co-resident workgroups (measured: 68 KiB -> 1 block/CU)
    static constexpr int LOGPAD = PAD_ ? eff_logpad(ELEM, LOGP) : 30;
    static constexpr int PITCH = N + (N >> LOGPAD);
    static constexpr size_t LDS_BYTES = (size_t)PITCH * ELEM;
};

// the same geometry with the inter-phase twiddle values requested before the exchange that precedes their phase
// (swiftly_fft.h, preload_tw_of): r4c, the forward K1 instances -- 1.690 -> 1.625 ms per facet (same box); the backward
// finish instances sit at 128 VGPRs and do not gain (1.80 -> 1.82 ms), so they keep the plain geometry
template <int LOGN_, int LOGP_, bool SPLIT_, bool PAD_ = true>
struct RGeoPre : RGeo<LOGN_, LOGP_, SPLIT_, PAD_> {
    static constexpr bool PRELOAD_TW = true;
};
// ... and with the preloaded values taken from the compact sections of RowPassArgs::twc (swiftly_fft.h, compact_tw_of):
// the W4 instances of the forward K1 (r5)
template <int LOGN_, int LOGP_, bool SPLIT_, bool PAD_ = true>
struct RGeoPreC : RGeoPre<LOGN_, LOGP_, SPLIT_, PAD_> {
    static constexpr bool COMPACT_TW = true;
};
// compact sections without the preload: the backward finish instances (128 VGPRs: values fetched where they are used)
template <int LOGN_, int LOGP_, bool SPLIT_, bool PAD_ = true>
struct RGeoC : RGeo<LOGN_, LOGP_, SPLIT_, PAD_> {
    static constexpr bool COMPACT_TW = true;
};

__device__ const float kRowOne = 1.f;

// (non-temporal accesses in the long-row kernels: loads measured much slower -- the input row is read by two workgroups:
// K2 25 -> 37.5 ms, r2; non-temporal band stores: K1 1.73 -> 1.95 ms per facet, r4 -- so they are plain)

// MODE 0: mapped load (window, pad, shift), identity store   -- prepare_facet / prepare_subgrid style
// MODE 1: identity load, mapped store (shift, crop, windows)  -- finish_facet / finish_subgrid style
// MODE 2: both mapped
template <class G, int MODE>
__global__ __launch_bounds__(G::NT) void row_pass_kernel(const RowPassArgs A, const cx<float>* __restrict__ gin,
                                                         cx<float>* __restrict__ gout,
                                                         const float* __restrict__ ld_win,
                                                         const float* __restrict__ st_win,
                                                         const float* __restrict__ st_win2,
                                                         const cx<float>* __restrict__ tw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int P = G::P, T = G::T, N = G::N;
    constexpr bool MAP_LD = MODE != 1, MAP_ST = MODE != 0;
    constexpr int CH = P < 8 ? P : 8;  // loads are issued in chunks of CH; products trail one chunk behind
    const int t = threadIdx.x;
    const int row = blockIdx.x;  // uniform
    int in_row = row;
    if (A.rm_mod > 0) {
        int r1 = row + A.rm_inner;
        if (r1 >= A.rm_mod) r1 -= A.rm_mod;
        r1 += A.rm_outer;
        if (r1 >= A.rm_full) r1 -= A.rm_full;
        in_row = r1;
    }
    if (A.in_rowmap) in_row = A.in_rowmap[in_row];
    const float alive = in_row < 0 ? 0.f : 1.f;  // a row absent from a compacted input (map entry < 0) reads as zeros
    if (in_row < 0) in_row = 0;
    const cx<float>* __restrict__ in = gin + (long long)in_row * A.in_pitch;  // uniform base
    cx<float>* __restrict__ out = gout + (long long)row * A.out_pitch;
    const float sg_ld = A.conj_ld ? -1.f : 1.f;
    const float sg_st = A.conj_st ? -1.f : 1.f;
    const float* __restrict__ lw = ld_win ? ld_win : &kRowOne;
    const int lws = ld_win ? 1 : 0;

    cx<float> x[P];
    if constexpr (MAP_LD) {
        // branch-free: out-of-map points read element 0 (valid) and get weight 0, so the loop body has no
        // control flow and the loads of a chunk stay in flight while the previous chunk is multiplied
        float w[P];
        static_for<0, P / CH + 1>([&](auto cI) {
            constexpr int c = decltype(cI)::value;
            if constexpr (c < P / CH) {
                static_for<c * CH, (c + 1) * CH>([&](auto vI) {
                    constexpr int v = decltype(vI)::value;
                    const int ci = (t + v * T) ^ (N >> 1);
                    const int q = (ci + A.ld_a) & (N - 1);
                    const bool ok = q < A.ld_len;
                    const int qs = ok ? q : 0;
                    unsigned idx = (unsigned)(qs + A.ld_c);
                    if (idx >= (unsigned)A.ld_mod) idx -= (unsigned)A.ld_mod;
                    x[v] = in[idx];
                    const float wv = lw[qs * lws];
                    w[v] = ok ? wv * alive : 0.f;
                });
            }
            if constexpr (c > 0) {
                static_for<(c - 1) * CH, c * CH>([&](auto vI) {
                    constexpr int v = decltype(vI)::value;
                    x[v].x *= w[v];
                    x[v].y *= w[v] * sg_ld;
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    } else {
        static_for<0, P>([&](auto vI) {
            constexpr int v = decltype(vI)::value;
            x[v] = in[(t + v * T) ^ (N >> 1)];
        });
        static_for<0, P>([&](auto vI) {
            constexpr int v = decltype(vI)::value;
            x[v].x *= alive;
            x[v].y *= sg_ld * alive;
        });
    }

    const float* __restrict__ sw1 = st_win ? st_win : &kRowOne;
    const float* __restrict__ sw2 = st_win2 ? st_win2 : &kRowOne;
    const int sw1s = st_win ? 1 : 0, sw2s = st_win2 ? 1 : 0;
    const float rscale = A.row_win ? A.scale * A.row_win[row] : A.scale;  // window of the OTHER axis, folded per row
    fft_phases<G, float, 0>(x, t, 0, false, smem, tw, [&](int e, cx<float> v) {
        const int ck = e ^ (N >> 1);
        if constexpr (!MAP_ST) {
            v.x *= rscale;
            v.y *= rscale * sg_st;
            cx<float>* p = out + ck;
            if (A.accumulate) {
                const cx<float> old = *p;
                v.x += old.x;
                v.y += old.y;
            }
            *p = v;
        } else {
            const int d = (ck + A.st_a) & (N - 1);
            const bool ok = d < A.st_len;
            const int ds = ok ? d : 0;
            unsigned idx = (unsigned)(ds + A.st_c);
            if (idx >= (unsigned)A.st_mod) idx -= (unsigned)A.st_mod;
            const float w = rscale * sw1[ds * sw1s] * sw2[ds * sw2s];
            v.x *= w;
            v.y *= w * sg_st;
            if (ok) {
                cx<float>* p = out + idx;
                if (A.accumulate) {
                    const cx<float> old = *p;
                    v.x += old.x;
                    v.y += old.y;
                }
                *p = v;
            }
        }
    });
}

// N = S*G::N transform of one row by S = 2^LOGS workgroups (radix-S decimation in
// frequency on load): part h computes the outputs k = S*k' + h,
//   X[S k'+h] = FFT_{N/S}( W_N^{h j} * sum_q x[j + q N/S] W_S^{q h} )[k'] .
// Each part is an N/S-point problem that fits comfortably (512 threads x 16
// points, 32-64 KB of LDS, ~45 VGPRs), so several workgroups are resident per
// CU and one's HBM phase overlaps another's butterflies.  (Measured on MI355X:
// a 1024-thread workgroup is never co-resident with a second one, whatever its
// LDS/VGPR footprint -- kernel time is exactly linear in ceil(workgroups/256) --
// so the single 32768-point workgroup and the 2 x 16384-point form run one
// workgroup per CU with all of its waves in the same phase.)
// The S parts of rows 8i..8i+7 are blocks 8S*i .. 8S*i+8S-1 with part = (b/8) mod S,
// so all parts of a row run on the same XCD (block b -> XCD b mod 8): the
// repeated reads of the input row and the interleaved (stride-S) output lines
// meet in one L2.  MODE 0 only (mapped load, identity store): K2.
template <class G, int LOGS, bool HAS_WIN>
__global__ __launch_bounds__(G::NT) void row_pass_split_kernel(const RowPassArgs A, const cx<float>* __restrict__ gin,
                                                               cx<float>* __restrict__ gout,
                                                               const float* __restrict__ ld_win,
                                                               const cx<float>* __restrict__ tw,
                                                               const cx<float>* __restrict__ tw_full) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int P = G::P, T = G::T, H = G::N, S = 1 << LOGS, N = S * G::N;
    const int t = threadIdx.x;
    const int b = blockIdx.x;
    const int h = (b >> 3) & (S - 1);
    const int row = ((b >> (3 + LOGS)) << 3) + (b & 7);  // uniform
    if (row >= A.nrows) return;
    int in_row = row;
    if (A.rm_mod > 0) {
        int r1 = row + A.rm_inner;
        if (r1 >= A.rm_mod) r1 -= A.rm_mod;
        r1 += A.rm_outer;
        if (r1 >= A.rm_full) r1 -= A.rm_full;
        in_row = r1;
    }
    if (A.in_rowmap) in_row = A.in_rowmap[in_row];
    const float alive = in_row < 0 ? 0.f : 1.f;  // row absent from a compacted input: zeros
    if (in_row < 0) in_row = 0;
    // row base pointers are wave-uniform: pin them to SGPRs so that every access is SGPR base + 32-bit lane offset
    // the row indices are wave-uniform: pin them to SGPRs so that the row bases are scalar and every access is
    // global_load/store  SGPR base + 32-bit lane offset  (the pointers keep their global address space;
    // rebuilding them from integers would turn every access into a FLAT one)
    in_row = __builtin_amdgcn_readfirstlane(in_row);
    const cx<float>* __restrict__ in = gin + (long long)in_row * A.in_pitch;
    cx<float>* __restrict__ out = gout + (long long)__builtin_amdgcn_readfirstlane(row) * A.out_pitch;
    const float sg_ld = A.conj_ld ? -1.f : 1.f;
    const float sg_st = A.conj_st ? -1.f : 1.f;
    // W_S^{q h} = (-i)^{(q h) mod 4 * (4/S)}: real/imag parts in {0, +-1}, uniform per workgroup
    float cr[S], ci[S];
    static_for<0, S>([&](auto qI) {
        constexpr int q = decltype(qI)::value;
        const int rot = ((q * h) * (4 / S)) & 3;
        cr[q] = rot == 0 ? 1.f : (rot == 2 ? -1.f : 0.f);
        ci[q] = rot == 1 ? -1.f : (rot == 3 ? 1.f : 0.f);
    });

    cx<float> x[P];
    static_for<0, P>([&](auto vI) {
        constexpr int v = decltype(vI)::value;
        const int j = t + v * T;
        cx<float> y = {0.f, 0.f};
        static_for<0, S>([&](auto qI) {
            constexpr int q = decltype(qI)::value;
            // plain index j + q*H -> centred index (j + q*H) ^ (N/2)
            const int qq = ((((j + q * H) ^ (N >> 1)) + A.ld_a) & (N - 1));
            const bool ok = qq < A.ld_len;
            const int qs = ok ? qq : 0;
            unsigned idx = (unsigned)(qs + A.ld_c);
            if (idx >= (unsigned)A.ld_mod) idx -= (unsigned)A.ld_mod;
            const cx<float> a = in[idx];
            float w = ok ? alive : 0.f;
            if constexpr (HAS_WIN) w *= ld_win[qs];
            const float ax = a.x * w, ay = a.y * w * sg_ld;
            y.x += ax * cr[q] - ay * ci[q];
            y.y += ax * ci[q] + ay * cr[q];
        });
        if (h) y = cmul(y, tw_full[(h * j) & (N - 1)]);
        x[v] = y;
    });

    fft_phases<G, float, 0>(x, t, 0, false, smem, tw, [&](int e, cx<float> v) {
        const int ck = (S * e + h) ^ (N >> 1);
        v.x *= A.scale;
        v.y *= A.scale * sg_st;
        out[ck] = v;
    });
}


// Full-facet contiguous-axis transform for long rows (N = 2 * G::N = 32768 with G = RGeo<14, 5, true>):
// K1 of the contiguous-axis-first pipeline (prepare_facet along the contiguous axis for every facet row,
// DESIGN.md section 4) and a drop-in for the K2 column kernel.
//
// Same radix-2 decimation-in-frequency split as row_pass_split_kernel (half h of a row = one 16384-point
// problem).  Two geometries: 512 threads x 32 points (66 KB LDS, <= 128 VGPRs: TWO workgroups resident per CU,
// one's HBM phase overlaps the other's butterflies; two exchanges, radix 32, 32, 16) and 1024 threads x 16
// points (one workgroup per CU).  Unlike the r1 kernel the load loop has no control flow and no dependent
// table load per point, so ALL loads of a lane are in flight together (r1: "L L wait L wait" per point --
// 59 % of wave cycles waiting): the inter-half twiddle W_N^j, j = t + T v, is W_N^t (ONE table load) times
// the compile-time constant W_64^(v 64 T / N).
// ST: 0 = plain row store, 1 = band store (parity-split), 2 = mapped store of a finish_* primitive (shift, crop to
// st_len, windows st_win * st_win2; st_c = 0, st_mod = st_len host-checked) -- finish_facet along the contiguous axis.
// PAIR (G = 512 x 32 only, host-checked: ld_a, ld_len and the input pitch even): the lane owns the ADJACENT points
// 2t, 2t+1 (+ 1024 r) and fetches them with ONE 16-byte load (8-byte for the window), the transform starts with the
// radix-16 phase.  The kernel without its butterflies and exchanges still took 1.73 of 1.93 ms: it is bound by the
// number of vector-memory INSTRUCTIONS (128 loads per lane: 0.42 ms for the window loads alone), not by bytes.
// (Skipping the loads whose 64 lanes are all padding -- 31 % of prepare_facet's, 64 % of finish_facet's band loads --
// with a wave-uniform branch was tried: the branches make the compiler hoist every load above the first use and wait
// at every join, 280 B/lane of spills; not kept.)
// (r3: PERSISTENT workgroups -- a grid of 2, 4 or 8 workgroups per CU, each looping over row halves, thread-index
// invariants kept out of the loop so that nothing spills -- were measured SLOWER than one workgroup per row half:
// K1 1.95 / 1.93 / 1.88 ms per facet against 1.82 ms.  The hardware dispatcher refills a CU the moment one of its two
// workgroups retires, which also keeps the two residents out of phase; a static loop does neither.  Not kept.)
// NSEG (r4): the padded row is 2P (PAIR: 32) segments of T (PAIR: 2T) consecutive points, and the lane's loads are one
// per segment.  A zero-padded row leaves whole segments empty (prepare_facet of a 22528-point facet in a 32768-point
// row: 10 of 32), but which ones depends on the facet offset.  The launcher therefore rotates the input cyclically by
// seg_rot segments so that the segments that can hold data are the first NSEG (compile time: their loads, window
// products and first-stage additions simply do not exist) and the kernel multiplies output k by W_nseg^(seg_rot k),
// which is ONE complex constant per lane (k = 2 e0 + h modulo the segment count for every output of the lane).
// NSEG = 0: all segments, no rotation (the r2/r3 kernel).  Measured (r4, same box): K1 1.79 -> 1.73 ms per facet -- the
// range-checked loads of the empty segments were already nearly free (no memory access), what goes away is their
// issue slots and the arithmetic on zeros.  (s_setprio 3 during the load phase: 1.95 ms; during the butterflies
// instead: 1.80 ms; neither kept.  tools/k1_trace.py + -DSWF_TRACE=1: a workgroup lives 46 k cycles -- loads 20 %,
// window products / first stage / inter-half twiddle 18 %, the three butterfly phases 35 %, the two exchanges 25 %
// -- with two workgroups per CU, i.e. the SIMDs issue about 55 % of the time.)
// CJ (r4b): -1 = conjugation on load / store from the runtime flags (multiplications by +-1); 1 / 0 = the launcher
// guarantees conj_ld = conj_st = 1 / 0 and the sign changes ride on the neg_hi modifier of the packed window and scale
// products (bit-identical results, 32 + 64 multiplications per lane fewer).
// Band store with ONE block per lane in the last phase (r4b): the lane's outputs are e = t + r T, i.e. cyclic band
// distance d_r = (2 t + h + N/2 - band_start + 2 T r) mod N.  Across a wave d_r spans 128 consecutive values, so for all
// but at most two r the whole wave is inside or outside the band: that decision is made on the SCALAR unit, outputs the
// wave does not keep (65 % on the 64k workload) cost two scalar instructions instead of the rotation phase product,
// the scale and five address / compare operations, and kept outputs are stored at  SGPR base + lane * 8.
// W4 (r5): the window of the pair loads comes from the re-laid-out table RowPassArgs::ld_win4 -- NS / 2 loads of 16 bytes
// per lane instead of NS of 8 bytes, the same bytes and bit-identical products; a shape probe of the kernel measured
// 1.64 against 1.73 ms per facet (tools/k1_shape_probe.hip V=64: the load phase is bound by the NUMBER of vector-memory
// instructions).
template <class G, bool HAS_WIN, int ST, bool PAIR = false, int NSEG = 0, int CJ = -1, bool W4 = false>
__global__ __launch_bounds__(G::NT, 4) void row_pass_band_kernel(const RowPassArgs A, const cx<float>* __restrict__ gin,
                                                                 cx<float>* __restrict__ gout,
                                                                 const float* __restrict__ ld_win,
                                                                 const cx<float>* __restrict__ tw,
                                                                 const cx<float>* __restrict__ tw_full) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int P = G::P, T = G::T, H = G::N, N = 2 * G::N;
    constexpr bool BAND = ST == 1;
    static_assert(64 % (N / T) == 0, "the inter-half twiddle uses W_64 constants: j = t + T v, W_N^(T v) = W_64^(v 64 T / N)");
    constexpr int WSTEP = 64 / (N / T);
    // segments: PAIR: 32 of 2T points (lane: one 16-byte load each), else 2P of T points
    constexpr int SEGLEN = PAIR ? 2 * T : T, NSEGTOT = N / SEGLEN;
    constexpr bool SEGSKIP = NSEG > 0;
    constexpr int NS = SEGSKIP ? NSEG : NSEGTOT;
    static_assert(NS <= NSEGTOT, "segment count");
    static_assert(!SEGSKIP || G::LOGN % G::LOGP == 0 || PAIR, "one output phase per lane needs a single block in the last phase");
    static_assert(!SEGSKIP || (T % P) == 0, "rotation phase must not depend on the last radix digit");
    const int t = threadIdx.x;
    const int b = blockIdx.x;
    const int h = (b >> 3) & 1;
    const int row = ((b >> 4) << 3) + (b & 7);  // uniform; both halves of a row on the same XCD (b mod 8)
    if (row >= A.nrows) return;
#if SWF_TRACE
    SWF_TRACE_POINT(0);
    if (threadIdx.x == 0 && blockIdx.x < (unsigned)kTraceBlocks) {
        swf_trace_buf[blockIdx.x * kTracePoints + 10] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) |
                                                       (unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11));
    }
#endif
    const int rot = SEGSKIP ? A.seg_rot * SEGLEN : 0;  // cyclic rotation of the transform input (points)
    int in_row = row;
    if (A.rm_mod > 0) {
        int r1 = row + A.rm_inner;
        if (r1 >= A.rm_mod) r1 -= A.rm_mod;
        r1 += A.rm_outer;
        if (r1 >= A.rm_full) r1 -= A.rm_full;
        in_row = r1;
    }
    if (A.in_rowmap) in_row = A.in_rowmap[in_row];
    const bool dead = in_row < 0;  // row not present in a compacted input: treated as zeros
    if (dead) in_row = 0;
    // the row indices are wave-uniform: pin them to SGPRs so that the row bases are scalar and every access is
    // global_load/store  SGPR base + 32-bit BYTE offset in one VGPR  (the pointers keep their global address
    // space; rebuilding them from integers would turn every access into a FLAT one, and 64-bit per-lane
    // addresses cost two VGPRs and a v_lshl_add_u64 per access)
    in_row = __builtin_amdgcn_readfirstlane(in_row);
    const char* __restrict__ inb = reinterpret_cast<const char*>(gin + (long long)in_row * A.in_pitch);
    char* __restrict__ outb = reinterpret_cast<char*>(gout + (long long)__builtin_amdgcn_readfirstlane(row) * A.out_pitch);
    const char* __restrict__ winb = reinterpret_cast<const char*>(ld_win);
    const float sg_ld = A.conj_ld ? -1.f : 1.f;
    const float sg_st = A.conj_st ? -1.f : 1.f;
    const float sgn = h ? -1.f : 1.f;   // W_2^{q h}

    // Load map of a prepare_* primitive (ld_c = 0, ld_mod = ld_len; host-checked): element q = (ci + ld_a) mod N
    // of the zero-padded row is in[q] for q < ld_len.  The loads are BUFFER loads whose descriptors cover exactly
    // the valid elements: the hardware range check returns 0 for the padding (no compare / select / clamp per
    // element, no memory access for the padding), and a row that is absent from a compacted input gets an empty
    // descriptor.  No control flow: the compiler keeps as many of the 2 P loads of a lane in flight as the
    // 128-VGPR budget (two workgroups per CU) allows.
    cx<float> x[P];
    if constexpr (PAIR) {
        static_assert(P == 32 && G::LOGN == 14, "pair loads: 512 threads x 32 points of a 16384-point half");
        constexpr int R1 = 16, SEG = H / R1;  // radix-16 first phase: points j + r*SEG, j = 2t + u
        const unsigned valid = dead ? 0u : (unsigned)A.ld_len;
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(inb), (short)0, (int)(valid << 3), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(winb), (short)0, (int)(valid << 2), 0x00020000);
        const unsigned base8 = (unsigned)((2 * t + A.ld_a + (N >> 1) + rot) & (N - 1)) << 3;
        // W4: segment pairs of the re-laid-out window table, in the order of this loop -- (r, r + 16) while segment r + 16
        // can hold data (r < NB1: both windows of an iteration in one load), then (r, r + 1) for the rest
        static_assert(!W4 || (HAS_WIN && SEGSKIP && NS >= R1 && NS % 2 == 0), "re-laid-out window: forward K1 instances");
        constexpr int NB1 = NS - R1;
        const __amdgpu_buffer_rsrc_t rs_w4 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W4 ? A.ld_win4 : nullptr), (short)0, W4 ? (NS / 2) * T * 16 : 0, 0x00020000);
        f32x4 wcarry = {0.f, 0.f, 0.f, 0.f};
#if SWF_K1_PIPE
        // (r5) EXPLICIT LOAD PIPELINE of the W4 instances.  The lane's loads come in NG groups of three 16-byte loads -- the
        // window quad of a segment pair and the pair's two data quads (segments (r, r + 16) for r < NB1, then (r, r + 1)) --
        // and 33 loads in flight would need 132 VGPRs.  Left to itself the compiler issues 20 loads, WAITS for nine of them,
        // and only then issues the other 13: two memory round trips per workgroup.  Here the first PIPE groups are requested
        // at once and every group consumed (window products, first-stage sum: 12 VGPRs shrink to 4 or 8) makes room for the
        // request of the next one, in program order pinned by scheduling barriers.
        constexpr bool PIPED = W4 && CJ == 1;
        if constexpr (PIPED) {
            constexpr int NGA = NB1, NG = NB1 + (R1 - NB1) / 2, PIPE = SWF_K1_PIPE < NG ? SWF_K1_PIPE : NG;
            f32x4 gw[NG], g0[NG], g1[NG];
            auto issue = [&](auto gI) {
                constexpr int g = decltype(gI)::value;
                constexpr int r0 = g < NGA ? g : NGA + 2 * (g - NGA);           // first slot of the group
                constexpr int s0 = r0, s1 = g < NGA ? r0 + R1 : r0 + 1;         // its two segments
                gw[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w4, (g * T + t) << 4, 0, 0));
                const unsigned o0 = (base8 + (unsigned)((s0 * SEG) << 3)) & (unsigned)((N << 3) - 1);
                const unsigned o1 = (base8 + (unsigned)((s1 * SEG) << 3)) & (unsigned)((N << 3) - 1);
                g0[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)o0, 0, 0));
                g1[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)o1, 0, 0));
            };
            auto wmul = [&](f32x4 val, f32x2 w, cx<float>& e0, cx<float>& e1) {  // (x w, -y w) per point: window + conjugation
                const f32x2 p0 = {val.x, val.y}, p1 = {val.z, val.w};
                f32x2 q0, q1;
                asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(q0) : "v"(p0), "v"(w));
                asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_hi:[1,0]" : "=v"(q1) : "v"(p1), "v"(w));
                e0 = pkc(q0);
                e1 = pkc(q1);
            };
            auto consume = [&](auto gI) {
                constexpr int g = decltype(gI)::value;
                constexpr int r0 = g < NGA ? g : NGA + 2 * (g - NGA);
                cx<float> a0[2], a1[2];
                wmul(g0[g], f32x2{gw[g].x, gw[g].y}, a0[0], a0[1]);
                wmul(g1[g], f32x2{gw[g].z, gw[g].w}, a1[0], a1[1]);
                static_for<0, 2>([&](auto uI) {
                    constexpr int u = decltype(uI)::value;
                    if constexpr (g < NGA) {
                        x[u + 2 * r0] = pkc(__builtin_elementwise_fma(pkv(a1[u]), f32x2{sgn, sgn}, pkv(a0[u])));
                    } else {
                        x[u + 2 * r0] = a0[u];
                        x[u + 2 * (r0 + 1)] = a1[u];
                    }
                });
            };
            static_for<0, PIPE>(issue);
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, NG>([&](auto gI) {
                constexpr int g = decltype(gI)::value;
                consume(gI);
                if constexpr (g + PIPE < NG) issue(std::integral_constant<int, g + PIPE>{});
                __builtin_amdgcn_sched_barrier(0);
            });
            static_for<2 * (NGA + 2 * (NG - NGA)), P>([&](auto vI) { x[decltype(vI)::value] = cx<float>{0.f, 0.f}; });
        } else
#endif
        static_for<0, R1>([&](auto rI) {
            constexpr int r = decltype(rI)::value;
            cx<float> a[2][2];  // [q][u]
            f32x2 w4[2] = {{0.f, 0.f}, {0.f, 0.f}};
            if constexpr (W4) {
                if constexpr (r < NB1) {
                    const f32x4 wv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w4, (r * T + t) << 4, 0, 0));
                    w4[0] = f32x2{wv.x, wv.y};
                    w4[1] = f32x2{wv.z, wv.w};
                } else if constexpr (((r - NB1) & 1) == 0) {
                    wcarry = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w4, ((NB1 + (r - NB1) / 2) * T + t) << 4, 0, 0));
                    w4[0] = f32x2{wcarry.x, wcarry.y};
                } else {
                    w4[0] = f32x2{wcarry.z, wcarry.w};
                }
            }
            static_for<0, 2>([&](auto qI) {
                constexpr int q = decltype(qI)::value;
                if constexpr (r + R1 * q < NS) {  // segment r + 16 q can hold data
                    const unsigned off8 = (base8 + (unsigned)((r * SEG + q * H) << 3)) & (unsigned)((N << 3) - 1);
                    const f32x4 val = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)off8, 0, 0));
                    if constexpr (HAS_WIN) {
                        f32x2 w;
                        if constexpr (W4)
                            w = w4[q];
                        else
                            w = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_w, (int)(off8 >> 1), 0, 0));
                        if constexpr (CJ >= 0) {  // one packed product per point; CJ = 1: (x w, -y w)
                            const f32x2 p0 = {val.x, val.y}, p1 = {val.z, val.w};
                            f32x2 r0, r1;
                            if constexpr (CJ == 1) {
                                asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(r0) : "v"(p0), "v"(w));
                                asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_hi:[1,0]" : "=v"(r1) : "v"(p1), "v"(w));
                            } else {
                                asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r0) : "v"(p0), "v"(w));
                                asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r1) : "v"(p1), "v"(w));
                            }
                            a[q][0] = pkc(r0);
                            a[q][1] = pkc(r1);
                        } else {
                            a[q][0] = cx<float>{val.x * w.x, val.y * w.x};
                            a[q][1] = cx<float>{val.z * w.y, val.w * w.y};
                        }
                    } else if constexpr (CJ == 1) {
                        a[q][0] = cx<float>{val.x, -val.y};
                        a[q][1] = cx<float>{val.z, -val.w};
                    } else {
                        a[q][0] = cx<float>{val.x, val.y};
                        a[q][1] = cx<float>{val.z, val.w};
                    }
                }
            });
            static_for<0, 2>([&](auto uI) {
                constexpr int u = decltype(uI)::value;
                if constexpr (CJ >= 0) {  // the conjugation already happened in the window product
                    if constexpr (r + R1 < NS)
                        x[u + 2 * r] = pkc(__builtin_elementwise_fma(pkv(a[1][u]), f32x2{sgn, sgn}, pkv(a[0][u])));
                    else if constexpr (r < NS)
                        x[u + 2 * r] = a[0][u];
                    else
                        x[u + 2 * r] = cx<float>{0.f, 0.f};
                } else if constexpr (r + R1 < NS)
                    x[u + 2 * r] = cx<float>{a[0][u].x + sgn * a[1][u].x, (a[0][u].y + sgn * a[1][u].y) * sg_ld};
                else if constexpr (r < NS)
                    x[u + 2 * r] = cx<float>{a[0][u].x, a[0][u].y * sg_ld};
                else
                    x[u + 2 * r] = cx<float>{0.f, 0.f};
            });
        });
#if SWF_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SWF_TRACE_POINT(1);
#endif
        if (h) {  // uniform: odd outputs need W_N^j, j = 2t + u + SEG r:  W_N^(2t+u) * W_64^(2r)
            // (r4c: requesting this value before the data loads -- an asm statement, the compiler sinks an ordinary load back to
            // here -- was measured SLOWER, 1.685 vs 1.653 ms per facet: four more VGPRs live through the load phase)
            const f32x4 wt = *reinterpret_cast<const f32x4*>(tw_full + 2 * t);
            const cx<float> w0 = {wt.x, wt.y}, w1 = {wt.z, wt.w};
            static_for<0, R1>([&](auto rI) {
                constexpr int r = decltype(rI)::value;
                x[2 * r] = mul_w64<float, 64 * SEG / N * r>(cmul(x[2 * r], w0));
                x[2 * r + 1] = mul_w64<float, 64 * SEG / N * r>(cmul(x[2 * r + 1], w1));
            });
        }
    } else {
    const unsigned valid = dead ? 0u : (unsigned)A.ld_len;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(inb), (short)0, (int)(valid << 3), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(winb), (short)0, (int)(valid << 2), 0x00020000);
    // byte offset of plain index j = t (centred index j ^ N/2 = j + N/2 mod N), then + (v T + q H) * 8 mod 8 N
    const unsigned base8 = (unsigned)((t + A.ld_a + (N >> 1) + rot) & (N - 1)) << 3;
    static_for<0, P>([&](auto vI) {
        constexpr int v = decltype(vI)::value;
        cx<float> a[2];
        static_for<0, 2>([&](auto qI) {
            constexpr int q = decltype(qI)::value;
            if constexpr (v + P * q < NS) {  // segment v + P q can hold data
                const unsigned off8 = (base8 + (unsigned)((v * T + q * H) << 3)) & (unsigned)((N << 3) - 1);
                const f32x2 val = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_in, (int)off8, 0, 0));
                if constexpr (HAS_WIN) {
                    const float w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_w, (int)(off8 >> 1), 0, 0));
                    a[q] = cx<float>{val.x * w, val.y * w};
                } else {
                    a[q] = cx<float>{val.x, val.y};
                }
            }
        });
        if constexpr (v + P < NS)
            x[v] = cx<float>{a[0].x + sgn * a[1].x, (a[0].y + sgn * a[1].y) * sg_ld};
        else if constexpr (v < NS)
            x[v] = cx<float>{a[0].x, a[0].y * sg_ld};
        else
            x[v] = cx<float>{0.f, 0.f};
    });
    if (h) {  // uniform: odd outputs need W_N^j = W_N^t * W_64^v
        const cx<float> wt = tw_full[t];
        static_for<0, P>([&](auto vI) {
            constexpr int v = decltype(vI)::value;
            x[v] = mul_w64<float, WSTEP * v>(cmul(x[v], wt));
        });
    }

    }
    SWF_TRACE_POINT(2);
    // rotated input: output k = 2 e + h, e = t + (last radix digit) * H / P, carries W_N^(rot k) = W_N^(rot (2 t + h))
    cx<float> rphi = {1.f, 0.f};
    // (the 32 distinct rotation phases from a compact table too: 12 bytes of scratch per lane in the NSEG = 22 instance, not kept)
    // (the 32 distinct phases from a 256-byte table behind the exchange buffer, read by the store epilogue: 12 bytes of scratch
    // in the NSEG = 22 instance again, K1 1.55 - 1.58 vs 1.51 - 1.53 ms per facet, r5; not kept)
    if constexpr (SEGSKIP) rphi = tw_full[(unsigned)(rot * (2 * t + h)) & (unsigned)(N - 1)];
    auto run_phases = [&](auto&& fin) {
        if constexpr (PAIR)
            fft_phases_pair<G, float>(x, t, smem, tw, fin, compact_tw_of<G>::value ? A.twc : nullptr);
        else
            fft_phases<G, float, 0>(x, t, 0, false, smem, tw, fin);
    };
    float scale = A.scale;
    if (A.row_win) scale *= A.row_win[row];
    const float scale_im = scale * sg_st;
    // band store: d = (ck - band_start) mod N has the parity of h ^ band_start for the whole workgroup, so the
    // destination (d & 1) * band_half + (d >> 1) is  region base + (d >> 1)
    const unsigned region = BAND ? (unsigned)(((h ^ A.band_start) & 1) * A.band_half) << 3 : 0u;
    if constexpr (ST == 2) {
        // ONE output window (the host combines mask and 1/PSWF), fetched for all P outputs of the lane before the
        // last butterflies: a load per output inside the store loop costs a full memory latency per element
        // (measured: 4.4 ms per 22528^2 facet with two dependent window loads per output)
        const unsigned vlen = (unsigned)A.st_len;
        const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.st_win), (short)0, (int)(A.st_win ? vlen << 2 : 0u), 0x00020000);
        const bool has1 = A.st_win != nullptr;  // uniform
        // outputs of a lane come in runs of CH slots whose indices differ by q << LNS (phase_scatter): the windows of
        // a run are fetched together right before its stores
        // (last phase: the remainder radix in the greedy schedule, the full radix when the short phase runs first)
        constexpr int LR = (PAIR || G::LOGN % G::LOGP == 0) ? G::LOGP : G::LOGN % G::LOGP, LNS = G::LOGN - LR;
        constexpr int CH = (1 << LR) < 16 ? (1 << LR) : 16;
        float wv[CH];
        // (r4b: the wave-uniform in-range test of the band store below was tried here too -- window loads and stores
        // inside scalar branches: 1.81 -> 1.92 ms per facet, the branches split the batches of 16 window loads; not kept)
        run_phases([&](int e, cx<float> v, auto sI) {
            constexpr int s = decltype(sI)::value;
            if constexpr (s % CH == 0) {
                static_for<0, CH>([&](auto qI) {
                    constexpr int q = decltype(qI)::value;
                    const int ckq = (2 * (e + (q << LNS)) + h) ^ (N >> 1);
                    const int dq = (ckq + A.st_a) & (N - 1);
                    wv[q] = has1 ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_w1, dq << 2, 0, 0)) : 1.f;
                });
            }
            const int ck = (2 * e + h) ^ (N >> 1);
            const int d = (ck + A.st_a) & (N - 1);
            const float w = scale * wv[s % CH];
            if constexpr (SEGSKIP) v = cmul(v, rphi);
            if (d < A.st_len) {
                f32x2 val = {v.x * w, v.y * w * sg_st};
                if (A.accumulate) {
                    const f32x2 old = *reinterpret_cast<const f32x2*>(outb + ((unsigned)d << 3));
                    val += old;
                }
                *reinterpret_cast<f32x2*>(outb + ((unsigned)d << 3)) = val;
            }
        });
        return;
    }
    constexpr bool ONEBLK = PAIR || (G::LOGN % G::LOGP == 0);  // last phase: one block per lane, outputs e = t + r T
    if constexpr (BAND && ONEBLK && CJ >= 0) {
        constexpr int LNS = G::LOGN - G::LOGP;
        static_assert(T == (1 << LNS), "e = t + (r << LNS)");
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
        const int dw = ((wave << 7) + (N >> 1) - A.band_start + h) & (N - 1);  // d of lane 0, output r = 0
        const int lane2 = (t & 63) << 1;
        const f32x2 sc = {scale, scale};
#if SWF_K1_BUFSTORE
        // (r6) WHICH LANES of a kept output are inside the band is left to the range check of a buffer store whose descriptor
        // covers exactly the kept columns of this workgroup's parity region: one store path, one scalar branch per output
        const int par = (h ^ A.band_start) & 1;
        const int ncol = (A.band_len - par + 1) >> 1;   // columns q = d >> 1 with 2 q + par < band_len
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(outb + region, (short)0, ncol << 3, 0x00020000);
        const unsigned off0 = (unsigned)(((dw + lane2) & (N - 1)) >> 1) << 3;
        run_phases([&](int, cx<float> v, auto sI) {
            constexpr int r = decltype(sI)::value;
            const int base = (dw + (r << (LNS + 1))) & (N - 1);  // wave-uniform
            const bool none_in = base >= A.band_len && base + 126 < N;
            if (none_in) return;
            if constexpr (SEGSKIP) v = cmul(v, rphi);
            f32x2 val;
            const f32x2 vv = pkv(v);
            if constexpr (CJ == 1)
                asm("v_pk_mul_f32 %0, %1, %2 neg_hi:[1,0]" : "=v"(val) : "v"(vv), "v"(sc));
            else
                val = vv * sc;
            const unsigned off = (off0 + (unsigned)(r << (LNS + 3))) & (unsigned)((N << 2) - 1);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, val), rs_out, (int)off, 0, 0);
        });
#else
        const unsigned lane8 = (unsigned)(t & 63) << 3;
        char* __restrict__ ob = outb + region;
        run_phases([&](int, cx<float> v, auto sI) {
            constexpr int r = decltype(sI)::value;
            const int base = (dw + (r << (LNS + 1))) & (N - 1);  // wave-uniform
            const bool all_in = base + 126 < A.band_len;
            const bool none_in = base >= A.band_len && base + 126 < N;
            if (none_in) return;
            if constexpr (SEGSKIP) v = cmul(v, rphi);
            f32x2 val;
            const f32x2 vv = pkv(v);
            if constexpr (CJ == 1)
                asm("v_pk_mul_f32 %0, %1, %2 neg_hi:[1,0]" : "=v"(val) : "v"(vv), "v"(sc));
            else
                val = vv * sc;
            if (all_in) {
                *reinterpret_cast<f32x2*>(ob + ((unsigned)(base >> 1) << 3) + lane8) = val;
            } else {
                const int d = (base + lane2) & (N - 1);
                if (d < A.band_len) *reinterpret_cast<f32x2*>(ob + ((unsigned)(d >> 1) << 3)) = val;
            }
        });
#endif
#if SWF_TRACE
        SWF_TRACE_POINT(7);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SWF_TRACE_POINT(8);
#endif
        return;
    }
    run_phases([&](int e, cx<float> v) {
        const int ck = (2 * e + h) ^ (N >> 1);
        if constexpr (SEGSKIP) v = cmul(v, rphi);
        const f32x2 val = {v.x * scale, v.y * scale_im};
        if constexpr (BAND) {
            const int d = (ck - A.band_start) & (N - 1);
            if (d < A.band_len)
                *reinterpret_cast<f32x2*>(outb + region + ((unsigned)(d >> 1) << 3)) = val;
        } else {
            *reinterpret_cast<f32x2*>(outb + ((unsigned)ck << 3)) = val;
        }
    });
#if SWF_TRACE
    SWF_TRACE_POINT(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SWF_TRACE_POINT(8);
#endif
}

constexpr int kRowPassMinLog = 13;
constexpr int kRowPassMaxLog = 15;
int launch_row_pass(int logn, int mode, const RowPassArgs& a, hipStream_t s);
int init_row_pass();
int row_pass_half_occupancy(int lds_bytes);
// multi-workgroup form for N = 32768 (MODE 0 only); tw14 / tw13 = tables of length 16384 / 8192, tw_full of length N
int launch_row_pass_split(const RowPassArgs& a, const cx<float>* tw14, const cx<float>* tw13, const cx<float>* tw_full,
                          hipStream_t s);
// 2 x 16384-point form with 512-thread workgroups (two per CU); a.band_len > 0 selects the band store
// Re-laid-out load windows of the W4 instances (RowPassArgs::ld_win4), built on first use per (window, position of the
// facet in the rotated padded row, facet size, segment count) and kept for the life of the owner (a handle: the tables
// are constants of the configuration like the twiddle tables; 4 KB per segment).  A table is built on the stream of the
// launch that first needs it; launches on other streams wait for its event.
struct Win4Cache {
    struct Entry {
        const float* win;
        int c, len, ns, n, seglen;
        float* tab;
        hipEvent_t ready;
        hipStream_t built_on;
        bool complete;  // the build kernel has finished (seen by hipEventQuery): later launches on other streams need no wait
    };
    std::mutex mu;
    std::vector<Entry> items;
    // compact twiddle sections (RowPassArgs::twc; swiftly_fft.h) of the 2 x 16384-point geometry with 32 points per lane,
    // set (and owned) by the owner of the cache: nullptr = the instances that gather from the plain table
    const cx<float>* twc = nullptr;
    static constexpr size_t kMaxEntries = 64;  // beyond: the plain instances (nothing is ever evicted: a kernel may be reading)
    const float* get(const float* win, int c, int len, int ns, int n, int seglen, hipStream_t s);
    void clear();  // frees the tables (owner's teardown, device idle)
};
int launch_row_pass_band(const RowPassArgs& a, const cx<float>* tw14, const cx<float>* tw_full, hipStream_t s, Win4Cache* w4 = nullptr);
int launch_row_pass_band_n(int logn, const RowPassArgs& a, const cx<float>* tw_half, const cx<float>* tw_full, hipStream_t s,
                           Win4Cache* w4 = nullptr);
int row_pass_band_occupancy();
int launch_row_pass_whole(const RowPassArgs& a, int nseg, const cx<float>* tw14, const cx<float>* tw_full, hipStream_t s);
int init_row_pass_whole();
int row_pass_whole_grid();  // workgroups of a whole-row launch (one per CU)
int row_pass_whole_stage_columns();  // physical band columns (both parities) the window-rows epilogue can stage

}  // namespace swf
