// SwiFTly on MI355X: fused "sum over facet groups + finish" along the
// contiguous axis of a subgrid (complex64).
//
// For every row of every subgrid of a wave this does, on chip,
//     acc   = sum_g  place_g( Fn * cfft_m( colacc[g][row, :] ) )        add_to_subgrid(axis=1), core.py:274-285,
//                                                                       summed over the off1 groups, api_helper.py:96-99
//     out_i = mask_i * cifft_xM(acc)[(xM/2 - xA//2 + i + off1) mod xM]  finish_subgrid(axis=1) + mask, core.py:316-323,
//                                                                       api_helper.py:110-111
// so the [xM, xM] accumulator never exists in HBM: one read of the group
// column buffers, one write of the half-finished subgrid (vs. zero-fill +
// read-modify-write per group + read/write of the finish in the unfused form).
#pragma once
#include "swiftly_fft.h"

namespace swf {

// threads per workgroup of the row-wise fused kernels (one wave per row: NT / 64 rows per workgroup).
// Measured (r2, same-box A/B): 128 gives the same time as 256 (K3-5 10.8 vs 10.7 ms per pass).
constexpr int kSumFinishThreads = 256;
// facets whose rows are in flight together in sum_finish_facets_kernel (registers: kSumFinishInFlight * m / 64 complex values)
constexpr int kSumFinishInFlight = 3;
constexpr int kSumFinishMaxGroups = 8;
constexpr int kSumFinishMaxBatch = 64;
// the facet kernels with the compact twiddle sections (0: gathers from the plain tables; A/B builds)
#ifndef SWF_SUMFINISH_COMPACT
#define SWF_SUMFINISH_COMPACT 1
#endif
template <class G>
using SFCompact = std::conditional_t<SWF_SUMFINISH_COMPACT != 0, CompactTw<G>, G>;

struct SumFinishArgs {
    const cx<float>* in;   // colacc[g][b][row][m]
    cx<float>* out;        // tmp[b][row][xA]
    long long in_gs, in_bs, in_rs;  // element strides: group, subgrid, row
    long long out_bs, out_rs;
    int nrows;             // rows per subgrid (xM)
    int ngroups, xA;
    int sp[kSumFinishMaxGroups];       // s' = floor(facet_off1 * xM / N) per group
    int st_a[kSumFinishMaxBatch];      // (-(xM/2 - xA//2 + off1_b)) mod xM per subgrid
    const float* fn;       // Fn[m]
    const float* mask;     // optional [nbatch][xA]
    long long mask_bs;
    const cx<float>* tw_m;
    const cx<float>* tw_x;
};

template <int LOGM, int LOGX>
struct SFGeo {
    // threads per row: one wave up to 2048-point rows (rows are wave-private: no workgroup barriers); FOUR waves for
    // 4096-point rows: the 35 KB accumulator row limits a CU to 3 rows, so the waves have to come from within the row
    // (measured r3, config-3 sizes, 16 facets per row: two waves per row + two rows per workgroup = 4 waves per CU,
    // 8.4 ms per wave of 16 subgrids = 1.0 TB/s).  The row then synchronises with workgroup barriers, which the
    // workgroup-uniform facet loop allows.
    static constexpr int LOGTR = LOGX >= 12 ? 8 : 6;
    static constexpr int TR = 1 << LOGTR;
    // threads per workgroup.  2048-point rows (21.7 KB of LDS per row): ONE row per workgroup so that seven rows fit
    // a CU (four rows per workgroup = 87 KB = one workgroup of four waves per CU: measured r3 on the N = 8192
    // workload, 485 us per wave of 8 subgrids); 4096-point rows: the row IS the workgroup (four waves)
    static constexpr int NT = LOGX >= 12 ? 256 : (LOGX == 11 ? 64 : kSumFinishThreads);
    using GM = Geo<float, LOGM, LOGM - LOGTR, NT, false>;
    using GX = Geo<float, LOGX, LOGX - LOGTR, NT, false>;
    static_assert(LOGM - LOGTR >= 1, "at least two points per lane");
    static_assert(GM::T == TR && GX::T == TR, "TR threads per row");
    static constexpr int RB = NT / TR;
    static constexpr size_t LDS_M = (size_t)RB * GM::PITCH * 8;
    static constexpr size_t LDS_X = (size_t)RB * GX::PITCH * 8;
    static constexpr size_t LDS_BYTES = LDS_M + LDS_X;
};

// Wave-parallel m-point transforms for the rows that are worked on by several waves (4096 points: 256 threads).  With
// all 256 threads on ONE m-point transform a lane holds 4 points: five radix-4 phases, four exchanges, nine workgroup
// barriers per group, eight groups per row on an 8x8 cover (measured r3, N = 32768 workload: 3.45 ms per wave = 2.8 TB/s,
// unchanged by prefetching the next group's rows -- the barriers, not the loads, are the cost).  Here each WAVE
// transforms a different group (16 points per lane, radix 16 x 16 x 4, wave-local exchanges in its own quarter of the
// buffer, no workgroup barrier) and the four results are added into the accumulator row together; groups whose
// placement windows overlap are put into different rounds by the host (SumFinishFacetArgs::rgroup), one barrier per round.
template <int LOGM, int LOGX>
struct SFWide {
    using S = SFGeo<LOGM, LOGX>;
    // (2048-point rows as two waves per row + this form: measured r3 on the N = 8192 workload, subgrid side 5.0 -> 5.55 ms
    // forward, 8.6 -> 8.4 ms for the backward pass: not adopted, one wave per row stays)
    static constexpr bool ON = LOGX >= 12 && LOGM >= 7;
    using GM = Geo<float, LOGM, (LOGM >= 7 ? LOGM - 6 : 1), S::NT, false>;  // 64 threads per transform
    static constexpr size_t LDS_M = GM::LDS_BYTES;
    static constexpr size_t LDS_BYTES = LDS_M + S::LDS_X;
};
template <int LOGM, int LOGX>
constexpr size_t sum_finish_facets_lds() {
    return SFWide<LOGM, LOGX>::ON ? SFWide<LOGM, LOGX>::LDS_BYTES : SFGeo<LOGM, LOGX>::LDS_BYTES;
}
// sum_finish_facets_kernel, one wave per row (r4): the accumulator row lives in REGISTERS (see the kernel), so LDS only
// holds ONE exchange buffer per row -- used by the m-point transforms first, by the xM-point transform last -- and Fn.
// (r5: re / im exchanged separately -- rows are wave-private, so it costs LDS instructions but no barrier -- halves the
// buffer and would admit 32 instead of 16 waves per CU, but the <9,10> instance needs 113 VGPRs: at 5 waves per SIMD
// (96 VGPRs, 64 B of scratch per lane) the subgrid side takes 9.58 instead of 8.98 ms per pass, at 8 waves per SIMD (64
// VGPRs, 184 B of scratch) 14.1 ms; not kept)
template <int LOGM, int LOGX>
constexpr size_t sum_finish_facets_reg_lds() {
    using S = SFGeo<LOGM, LOGX>;
    return (S::LDS_X > S::LDS_M ? S::LDS_X : S::LDS_M) + ((size_t)4 << LOGM);
}
template <int LOGM, int LOGX>
constexpr size_t sum_finish_facets_kernel_lds() {
    return SFWide<LOGM, LOGX>::ON ? SFWide<LOGM, LOGX>::LDS_BYTES : sum_finish_facets_reg_lds<LOGM, LOGX>();
}

template <int LOGM, int LOGX>
__global__ __launch_bounds__((SFGeo<LOGM, LOGX>::NT)) void sum_finish_rows_kernel(const SumFinishArgs A) {
    using S = SFGeo<LOGM, LOGX>;
    using GM = typename S::GM;
    using GX = typename S::GX;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cx<float>* ex_m = reinterpret_cast<cx<float>*>(smem);
    cx<float>* acc = reinterpret_cast<cx<float>*>(smem + S::LDS_M);
    constexpr int M = GM::N, X = GX::N, PM = GM::P, PX = GX::P, TR = S::TR;
    const int t = threadIdx.x % TR, rb = threadIdx.x / TR;
    const int b = blockIdx.y;
    const int row = blockIdx.x * S::RB + rb;
    const bool live = row < A.nrows;
    const int rrow = live ? row : 0;

    // zero the accumulator row
    static_for<0, PX>([&](auto vI) {
        constexpr int v = decltype(vI)::value;
        acc[lds_pos<GX>(rb, t + v * TR, false)] = cx<float>{0.f, 0.f};
    });
    row_sync<GX>(false);

    for (int g = 0; g < A.ngroups; g++) {
        const cx<float>* __restrict__ in = A.in + (long long)g * A.in_gs + (long long)b * A.in_bs + (long long)rrow * A.in_rs;
        cx<float> x[PM];
        static_for<0, PM>([&](auto vI) {
            constexpr int v = decltype(vI)::value;
            x[v] = in[(t + v * TR) ^ (M >> 1)];  // plain index -> centred element
        });
        const int sp = A.sp[g];
        fft_phases<GM, float, 0>(x, t, rb, false, ex_m, A.tw_m, [&](int e, cx<float> v) {
            const int ck = e ^ (M >> 1);
            const int k = (ck - sp) & (M - 1);
            const int dest = (k + (X >> 1) - (M >> 1) + sp) & (X - 1);  // centred position in the padded subgrid
            const float w = A.fn[k];
            cx<float>* p = acc + lds_pos<GX>(rb, dest ^ (X >> 1), false);  // stored at its plain iFFT index
            cx<float> o = *p;
            o.x += v.x * w;
            o.y += v.y * w;
            *p = o;
        });
        row_sync<GX>(false);  // also protects ex_m reuse by the next group
    }

    cx<float> y[PX];
    static_for<0, PX>([&](auto vI) {
        constexpr int v = decltype(vI)::value;
        cx<float> val = acc[lds_pos<GX>(rb, t + v * TR, false)];
        val.y = -val.y;  // inverse transform = conj(FFT(conj(.)))
        y[v] = val;
    });
    row_sync<GX>(false);
    cx<float>* __restrict__ out = A.out + (long long)b * A.out_bs + (long long)rrow * A.out_rs;
    const float* __restrict__ mask = A.mask ? A.mask + (long long)b * A.mask_bs : nullptr;
    const int st_a = A.st_a[b];
    const float scale = 1.f / (float)X;
    fft_phases<GX, float, 0>(y, t, rb, false, acc, A.tw_x, [&](int e, cx<float> v) {
        const int ck = e ^ (X >> 1);
        const int d = (ck + st_a) & (X - 1);
        if (d < A.xA && live) {
            float w = scale;
            if (mask) w *= mask[d];
            out[d] = cx<float>{v.x * w, -v.y * w};
        }
    });
}

// ---------------------------------------------------------------------------------------------------------
// "sum over FACETS + finish": the second half of api_helper.sum_and_finish_subgrid (api_helper.py:81-112) with
// the facet sum re-associated so that no accumulator ever lives in HBM.
//
// Input: G[f][b] = [m, m] per (facet, subgrid) = add_to_subgrid(axis 0) WITHOUT its placement:
//     G[k, j] = Fn[k] * cfft_m(contribution[:, j])[(k + s'0_f) mod m]            (swiftly_hip_transform_contributions)
// Row k of G[f][b] belongs to row  r = (k + xM/2 - m/2 + s'0_f) mod xM  of the padded subgrid.  For every padded row r
// this kernel sums, over ALL facets f whose band covers r, the axis-1 transforms of G[f][b][k_f(r), :] placed by the
// facet's off1 (core.py:274-285), then finishes axis 1 (inverse transform, crop, mask; core.py:316-323).  Each input
// row is read exactly once; there is no zero-fill and no read-modify-write (r1: colacc zero-filled, re-written by one
// launch per facet off0, re-read here).
constexpr int kSumFinishMaxFacets = 64;

struct SumFinishFacetArgs {
    const cx<float>* in;   // G[f][b][k][m]
    cx<float>* out;        // tmp[b][r][xA]
    long long in_fs, in_bs, in_rs;  // element strides: facet, subgrid, row
    long long out_bs, out_rs;
    int nrows;             // padded rows per subgrid (xM)
    int nfacets, xA;
    // The facets are listed GROUPED by their off1 (entries gstart[g] .. gstart[g+1] of fidx / base0 belong to group g):
    // the placement along this axis only depends on off1, so the rows of one group's facets are summed BEFORE the
    // m-point transform (linearity) -- one transform + one scatter per group that covers the row instead of one per
    // facet (3 instead of 4.5 per row on the 3x3 cover, 8 instead of 16 on an 8x8 cover); r3.
    int ngroups;
    int gstart[kSumFinishMaxFacets + 1];
    int fidx[kSumFinishMaxFacets];   // facet index into `in` of entry n
    int base0[kSumFinishMaxFacets];  // (xM/2 - m/2 + s'0_f) mod xM of entry n: first padded row of that facet's band
    int gsp1[kSumFinishMaxFacets];   // s'1 = floor(facet_off1 * xM / N) of GROUP g
    int st_a[kSumFinishMaxBatch];    // (-(xM/2 - xA//2 + off1_b)) mod xM per subgrid
    const float* fn;       // Fn[m]
    const float* mask;     // optional [nbatch][xA]
    long long mask_bs;
    const cx<float>* tw_m;
    const cx<float>* tw_x;
    // compact copies of the two tables (swiftly_fft.h, "compact twiddle sections": the lanes otherwise gather their table
    // values with strides of up to a cache line; r5) for GM::LOGP / GX::LOGP points per lane; host-checked non-null
    const cx<float>* twc_m;
    const cx<float>* twc_x;
    // Wave-parallel form (4096-point rows, SFWide): the groups in ROUNDS of mutually disjoint placement windows
    // (rgroup[rstart[r] .. rstart[r+1]) = the groups of round r); the waves of a workgroup transform different groups of
    // a round at the same time and add them into the shared accumulator row without conflicts
    int nrounds;
    int rstart[kSumFinishMaxFacets + 1];
    int rgroup[kSumFinishMaxFacets];
    // (r6, axis-1-first pipeline) 1 = the rows of `in` already ARE  Fn[k] * cfft_m(contribution)[(k + s'1) mod m]  along the
    // contiguous axis (axis1_rows_kernel ran before the strided-axis transforms): no m-point transform here, the rows of
    // a group are summed and placed.  Register form only.
    int placed;
};

// (register form, 1024-point rows: 4 waves per SIMD = 16 rows per CU, which the 8.7 KB of LDS per row now allow)
template <int LOGM, int LOGX>
__global__ __launch_bounds__((SFGeo<LOGM, LOGX>::NT), (LOGX <= 10 ? 4 : 1)) void sum_finish_facets_kernel(const SumFinishFacetArgs A) {
    using S = SFGeo<LOGM, LOGX>;
    using GX = typename S::GX;
    using W = SFWide<LOGM, LOGX>;
    using GM = std::conditional_t<W::ON, typename W::GM, typename S::GM>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cx<float>* ex_m = reinterpret_cast<cx<float>*>(smem);
    // (wave-parallel form: accumulator row in LDS behind the m-point exchange buffers; register form: one exchange
    // buffer per row shared by both transforms, then the Fn table)
    cx<float>* acc = reinterpret_cast<cx<float>*>(smem + (W::ON ? W::LDS_M : 0));
    constexpr int M = GM::N, X = GX::N, PM = GM::P, PX = GX::P, TR = S::TR;
    const int t = threadIdx.x % TR, rb = threadIdx.x / TR;
    const int b = blockIdx.y;
    const int row0 = blockIdx.x * S::RB;
    const int row = row0 + rb;
    const bool live = row < A.nrows;
    cx<float> y[PX];  // register form: the lane's part of the accumulator row, y[v] = row[t + v * TR]

    if constexpr (W::ON) {
        static_for<0, PX>([&](auto vI) {
            constexpr int v = decltype(vI)::value;
            acc[lds_pos<GX>(rb, t + v * TR, false)] = cx<float>{0.f, 0.f};
        });
        row_sync<GX>(false);
    }

    if constexpr (W::ON) {
        static_assert(S::RB == 1, "one row per workgroup");
        constexpr int NBW = 2;  // rows of one group in flight per wave (16 points per lane each)
        const int lane = threadIdx.x & 63;
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        constexpr int NW = S::NT / 64;
        for (int r = 0; r < A.nrounds; r++) {  // workgroup-uniform
            for (int i0 = A.rstart[r]; i0 < A.rstart[r + 1]; i0 += NW) {
                const int i = i0 + wv;  // wave-uniform
                if (i < A.rstart[r + 1]) {
                    const int g = A.rgroup[i];
                    cx<float> xs[PM];
                    static_for<0, PM>([&](auto vI) { xs[decltype(vI)::value] = cx<float>{0.f, 0.f}; });
                    bool anyg = false;
                    int n = A.gstart[g];
                    const int ne = A.gstart[g + 1];
                    while (n < ne) {
                        int fs[NBW];
                        int cnt = 0;
                        for (; n < ne && cnt < NBW; n++) {
                            const bool any = live && ((row - A.base0[n]) & (X - 1)) < M;
                            if (any) fs[cnt++] = n;
                        }
                        if (cnt == 0) break;
                        anyg = true;
                        cx<float> x[NBW][PM];
                        static_for<0, NBW>([&](auto sI) {
                            constexpr int sl = decltype(sI)::value;
                            if (sl < cnt) {  // wave-uniform
                                const int nn = fs[sl];
                                const int k = (row - A.base0[nn]) & (X - 1);
                                const cx<float>* __restrict__ in = A.in + (long long)A.fidx[nn] * A.in_fs +
                                                                   (long long)b * A.in_bs + (long long)(live ? k : 0) * A.in_rs;
                                static_for<0, PM>([&](auto vI) {
                                    constexpr int v = decltype(vI)::value;
                                    x[sl][v] = in[(lane + v * 64) ^ (M >> 1)];  // plain index -> centred element
                                });
                            }
                        });
                        const float lw = live ? 1.f : 0.f;
                        static_for<0, NBW>([&](auto sI) {
                            constexpr int sl = decltype(sI)::value;
                            if (sl < cnt) {
                                static_for<0, PM>([&](auto vI) {
                                    constexpr int v = decltype(vI)::value;
                                    xs[v].x += x[sl][v].x * lw;
                                    xs[v].y += x[sl][v].y * lw;
                                });
                            }
                        });
                    }
                    if (anyg) {
                        const int sp = A.gsp1[g];
                        fft_phases<SFCompact<GM>, float, 0>(xs, lane, wv, false, ex_m, A.tw_m, [&](int e, cx<float> v) {
                            const int ck = e ^ (M >> 1);
                            const int kk = (ck - sp) & (M - 1);
                            const int dest = (kk + (X >> 1) - (M >> 1) + sp) & (X - 1);
                            const float w = A.fn[kk];
                            cx<float>* p = acc + lds_pos<GX>(0, dest ^ (X >> 1), false);
                            cx<float> o = *p;
                            o.x += v.x * w;
                            o.y += v.y * w;
                            *p = o;
                        }, nullptr, A.twc_m);
                    }
                }
            }
            __syncthreads();  // the next round's windows overlap this round's
        }
    } else {

    // Per off1 group: the rows of the group's facets whose band covers this workgroup's rows are requested SF_NB at a
    // time (a wave pays the HBM latency once per batch, not once per facet; r2: one facet at a time -- 4.5 dependent
    // round trips per row on the 3x3 cover, SQ_WAIT_ANY 58 % of the wave cycles at 12 waves per CU) and summed in
    // registers; the sum is transformed, weighted and scattered into the accumulator row once.  (Requesting the NEXT
    // batch before the current group is transformed -- measured r3 on the 4096-point rows of the N = 32768 workload,
    // whose 8 groups per row are 8 dependent load -> transform steps: 114.5 vs 111-113 ms for the subgrid side, no
    // gain; the 4-points-per-lane m-point transforms with their 9 workgroup barriers each are the cost there.)
    // r4: ACCUMULATOR IN REGISTERS.  Output e of the m-point transform of lane t is e = t (mod 64) (Stockham: every
    // output of a lane is congruent to its lane index modulo the thread count), and its place in the padded row, taken
    // at its plain inverse-transform index p = (kk - m/2 + s') mod xM, is congruent to e modulo m -- so it lands in one
    // of lane t's OWN inputs y[v] = row[t + 64 v] of the xM-point transform, v = (e - t) / 64 + (m / 64) c with
    // c = p / m in [0, xM / m).  The placement becomes xM / m multiply-adds per output with weights Fn[kk] * (c == c')
    // instead of a read-modify-write of an LDS row: no accumulator row (13.3 -> 8.7 KB of LDS per row: 16 instead of
    // 12 waves per CU), no zero fill, no final read-back, and Fn from a copy in LDS.
    static_assert(S::LDS_X >= S::LDS_M, "one exchange buffer per row, sized by the longer transform");
    constexpr int RATIO = X / M;
    static_assert(PX == PM * RATIO, "accumulator slots");
    cx<float>* ex_row = reinterpret_cast<cx<float>*>(smem) + (size_t)rb * GX::PITCH;  // this row's exchange buffer (rb = 0 addressing)
    float* fn_l = reinterpret_cast<float*>(smem + (S::LDS_X > S::LDS_M ? S::LDS_X : S::LDS_M));
    for (int i = threadIdx.x; i < M; i += S::NT) fn_l[i] = A.fn[i];
    static_for<0, PX>([&](auto vI) { y[decltype(vI)::value] = cx<float>{0.f, 0.f}; });
    __syncthreads();
    // (8 points per lane and 1024-point rows: two facets in flight keep the kernel at 128 VGPRs without spills)
    constexpr int NB = (PM >= 8 && LOGX <= 10 && kSumFinishInFlight > 2) ? 2 : kSumFinishInFlight;
    for (int g = 0; g < A.ngroups; g++) {  // workgroup-uniform
        cx<float> xs[PM];
        static_for<0, PM>([&](auto vI) { xs[decltype(vI)::value] = cx<float>{0.f, 0.f}; });
        bool anyg = false;
        int n = A.gstart[g];
        const int ne = A.gstart[g + 1];
        while (n < ne) {
            int fs[NB];
            int cnt = 0;
            for (; n < ne && cnt < NB; n++) {  // workgroup-uniform scan
                const int base = A.base0[n];
                bool any = false;
                for (int q = 0; q < S::RB; q++) any = any || (((row0 + q - base) & (X - 1)) < M && row0 + q < A.nrows);
                if (any) fs[cnt++] = n;
            }
            if (cnt == 0) break;
            anyg = true;
            cx<float> x[NB][PM];
            float wgt[NB];
            static_for<0, NB>([&](auto sI) {
                constexpr int sl = decltype(sI)::value;
                if (sl < cnt) {  // uniform
                    const int nn = fs[sl];
                    const int k = (row - A.base0[nn]) & (X - 1);
                    const bool on = live && k < M;
                    const cx<float>* __restrict__ in =
                        A.in + (long long)A.fidx[nn] * A.in_fs + (long long)b * A.in_bs + (long long)(on ? k : 0) * A.in_rs;
                    wgt[sl] = on ? 1.f : 0.f;
                    // placed rows are indexed by kk = (centred output index - s'1) mod m of the slot the value lands in
                    const int rot = A.placed ? A.gsp1[g] : 0;
                    static_for<0, PM>([&](auto vI) {
                        constexpr int v = decltype(vI)::value;
                        x[sl][v] = in[(((t + v * TR) ^ (M >> 1)) - rot) & (M - 1)];  // plain index -> centred element
                    });
                }
            });
            static_for<0, NB>([&](auto sI) {
                constexpr int sl = decltype(sI)::value;
                if (sl < cnt) {  // uniform
                    static_for<0, PM>([&](auto vI) {
                        constexpr int v = decltype(vI)::value;
                        xs[v].x += x[sl][v].x * wgt[sl];
                        xs[v].y += x[sl][v].y * wgt[sl];
                    });
                }
            });
        }
        if (!anyg) continue;
        const int sp = A.gsp1[g];
        if (A.placed) {  // workgroup-uniform: xs[v] = sum of the group's rows at element kk_v (loaded below in that order)
            static_for<0, PM>([&](auto vI) {
                constexpr int v = decltype(vI)::value;
                const int ck = (t + v * TR) ^ (M >> 1);
                const int kk = (ck - sp) & (M - 1);
                const int p = (kk - (M >> 1) + sp) & (X - 1);
                const int c = p >> LOGM;
                static_for<0, RATIO>([&](auto cI) {
                    constexpr int cc = decltype(cI)::value;
                    const float wc = c == cc ? 1.f : 0.f;
                    y[v + PM * cc].x += xs[v].x * wc;
                    y[v + PM * cc].y += xs[v].y * wc;
                });
            });
            continue;
        }
        fft_phases<SFCompact<GM>, float, 0>(xs, t, 0, false, ex_row, A.tw_m, [&](int e, cx<float> v, auto sI) {
            // slot = u * RAD + r of the last phase (radix RAD = 2^LR, NB = PM / RAD blocks): e = t + TR * (u + NB * r)
            constexpr int LR = GM::LOGN % GM::LOGP == 0 ? GM::LOGP : GM::LOGN % GM::LOGP, RAD = 1 << LR, NBL = PM / RAD;
            constexpr int slot = (decltype(sI)::value / RAD) + NBL * (decltype(sI)::value % RAD);
            const int ck = e ^ (M >> 1);
            const int kk = (ck - sp) & (M - 1);
            const int p = (kk - (M >> 1) + sp) & (X - 1);  // plain inverse-transform index of the placed element
            const int c = p >> LOGM;
            const float w = fn_l[kk];
            static_for<0, RATIO>([&](auto cI) {
                constexpr int cc = decltype(cI)::value;
                const float wc = c == cc ? w : 0.f;
                y[slot + PM * cc].x += v.x * wc;
                y[slot + PM * cc].y += v.y * wc;
            });
        }, nullptr, A.twc_m);
        row_sync<GX>(false);  // the exchange buffer is reused by the next group
    }
    static_for<0, PX>([&](auto vI) { y[decltype(vI)::value].y = -y[decltype(vI)::value].y; });  // inverse = conj(FFT(conj(.)))
    acc = ex_row - (size_t)rb * GX::PITCH;  // the xM-point transform exchanges through the same rows

    }  // !W::ON

    if constexpr (W::ON) {
        static_for<0, PX>([&](auto vI) {
            constexpr int v = decltype(vI)::value;
            cx<float> val = acc[lds_pos<GX>(rb, t + v * TR, false)];
            val.y = -val.y;  // inverse transform = conj(FFT(conj(.)))
            y[v] = val;
        });
        row_sync<GX>(false);
    }
    cx<float>* __restrict__ out = A.out + (long long)b * A.out_bs + (long long)(live ? row : 0) * A.out_rs;
    const float* __restrict__ mask = A.mask ? A.mask + (long long)b * A.mask_bs : nullptr;
    const int st_a = A.st_a[b];
    const float scale = 1.f / (float)X;
    if constexpr (!W::ON) {
        // r4: the mask values of the lane's PX outputs are requested together BEFORE the transform (output slot s of the
        // last phase is element e = t + TR * (u + NB r)) instead of inside the store loop (fewer branches; measured r4,
        // same box: 251.9 vs 253.7 us per wave for sum_finish + K5b, i.e. no difference)
        constexpr int LRX = GX::LOGN % GX::LOGP == 0 ? GX::LOGP : GX::LOGN % GX::LOGP, RADX = 1 << LRX, NBX = PX / RADX;
        float mw[PX];
        static_for<0, PX>([&](auto sI) {
            constexpr int sl = decltype(sI)::value;
            constexpr int vi = (sl / RADX) + NBX * (sl % RADX);
            const int d = (((t + TR * vi) ^ (X >> 1)) + st_a) & (X - 1);
            const bool ok = d < A.xA && live;
            mw[sl] = mask ? mask[ok ? d : 0] : 1.f;
        });
        fft_phases<SFCompact<GX>, float, 0>(y, t, rb, false, acc, A.tw_x, [&](int e, cx<float> v, auto sI) {
            constexpr int sl = decltype(sI)::value;
            const int ck = e ^ (X >> 1);
            const int d = (ck + st_a) & (X - 1);
            const float w = scale * mw[sl];
            if (d < A.xA && live) out[d] = cx<float>{v.x * w, -v.y * w};
        }, nullptr, A.twc_x);
        return;
    }
    fft_phases<SFCompact<GX>, float, 0>(y, t, rb, false, acc, A.tw_x, [&](int e, cx<float> v) {
        const int ck = e ^ (X >> 1);
        const int d = (ck + st_a) & (X - 1);
        if (d < A.xA && live) {
            float w = scale;
            if (mask) w *= mask[d];
            out[d] = cx<float>{v.x * w, -v.y * w};
        }
    }, nullptr, A.twc_x);
}

// ---------------------------------------------------------------------------------------------------------
// AXIS-1-FIRST pipeline (r6; SwiftlyConfig(axis1_first=True)): the contiguous-axis half of add_to_subgrid (core.py:255-285)
// applied to the rows of the K1 output BEFORE the strided-axis transforms K2 / K3 -- the transforms commute
// (api_helper.py:81-99, 200-210) -- so that K2 and K3 work on data that carries ONE facet window instead of two: their
// float32 rounding then reaches the subgrid 10x weaker (tests/accuracy_model.py: 1.30e-5 -> 2.1e-6 end to end).
// Per facet f, facet row r and wave (subgrid off1, s = off1 yN / N):
//     x[(i + s) mod m] = P_f[r, (yN/2 - m/2 + i + s) mod yN]          extract_from_facet(axis 1), core.py:243-253
//     Z[k]             = Fn[k] * cfft_m(x)[(k + s'1_f) mod m]          add_to_subgrid(axis 1) without its placement
//     W_f[r, band column of window element i] = Z[(i + s) mod m]       parity-split band of exactly the window's m columns
// The output is laid out as a K1 band buffer whose band IS the window (start (yN/2 - m/2 + s) mod yN, length m), so the
// unchanged K2 gathers Z[j] as "window element" j; sum_finish_facets then runs with SumFinishFacetArgs::placed.
struct Axis1RowsArgs {
    const cx<float>* in;   // bands[f][row][band columns], parity-split
    cx<float>* out;        // W[f][row][m], parity-split band of the window
    long long in_fs, in_rs, out_fs, out_rs;
    int nrows, yN;
    int band_start, band_len, band_half;
    int c0;                // logical column of window element 0: (yN/2 - m/2 + s) mod yN
    int s;                 // s mod m
    int sp[kSumFinishMaxFacets];  // s'1 = floor(facet_off1 * xM / N) mod m per facet
    const float* fn;
    const cx<float>* tw_m;
    const cx<float>* twc_m;
};
constexpr int kAxis1Threads = 256;  // four rows per workgroup, one wave per row (wave-private exchange buffers)
template <int LOGM>
struct Axis1Geo {
    using GM = Geo<float, LOGM, LOGM - 6, kAxis1Threads, false>;
    static constexpr size_t LDS_BYTES = GM::LDS_BYTES;
};
template <int LOGM>
__global__ __launch_bounds__(kAxis1Threads) void axis1_rows_kernel(const Axis1RowsArgs A) {
    using GM = typename Axis1Geo<LOGM>::GM;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cx<float>* ex = reinterpret_cast<cx<float>*>(smem);
    constexpr int M = GM::N, PM = GM::P, TR = 64;
    const int t = threadIdx.x & 63, rb = threadIdx.x >> 6;
    const int f = blockIdx.y;
    const int row = blockIdx.x * GM::RB + rb;
    const bool live = row < A.nrows;
    const cx<float>* __restrict__ in = A.in + (long long)f * A.in_fs + (long long)(live ? row : 0) * A.in_rs;
    cx<float>* __restrict__ out = A.out + (long long)f * A.out_fs + (long long)(live ? row : 0) * A.out_rs;
    const int sp = A.sp[f];
    cx<float> x[PM];
    static_for<0, PM>([&](auto vI) {
        constexpr int v = decltype(vI)::value;
        const int ci = (t + v * TR) ^ (M >> 1);          // centred element of x (plain index t + v TR)
        const int i = (ci - A.s) & (M - 1);              // window element that lands there
        int col = A.c0 + i;
        if (col >= A.yN) col -= A.yN;
        int d = col - A.band_start;
        if (d < 0) d += A.yN;
        const bool ok = d < A.band_len;                  // (a window always lies inside the band of its plan)
        x[v] = in[ok ? (d & 1) * A.band_half + (d >> 1) : 0];
        if (!ok) x[v] = cx<float>{0.f, 0.f};
    });
    fft_phases<SFCompact<GM>, float, 0>(x, t, rb, false, ex, A.tw_m, [&](int e, cx<float> v) {
        const int ck = e ^ (M >> 1);
        const int kk = (ck - sp) & (M - 1);
        const float w = A.fn[kk];
        const int i2 = (kk - A.s) & (M - 1);
        if (live) out[(i2 & 1) * (M >> 1) + (i2 >> 1)] = cx<float>{v.x * w, v.y * w};
    }, nullptr, A.twc_m);
}

// ---------------------------------------------------------------------------------------------------------
// "prepare + split over FACETS": the mirror of sum_finish_facets_kernel for the backward pass -- the contiguous-axis
// half of api_helper.prepare_and_split_subgrid (api_helper.py:115-139) for all facets at once.
//
// Input: tmp[b] = [xM, xA] = prepare_subgrid along axis 0 of subgrid b (core.py:328-368).  For every padded row r:
//     P    = cfft_xM( pad(tmp[b][r, :], off1_b) )                          prepare_subgrid(axis 1), kept in LDS
//     E[f][b][k_f(r), :] = cifft_m( Fn * P[window of facet f's off1] )      extract_from_subgrid(axis 1), core.py:396-439
// for every facet f whose axis-0 band covers r (k_f(r) = (r - base0_f) mod xM < m).  The [xM, xM] prepared subgrid
// and the per-off0 intermediates [m, xM] never exist in HBM; the remaining axis-0 half of extract_from_subgrid
// (window Fn over k, inverse transform of length m) is one column pass over E (swiftly_hip_split_prepare_facets).
struct SplitFacetArgs {
    const cx<float>* in;   // tmp[b][r][xA]
    cx<float>* out;        // E[f][b][k][m]
    long long in_bs, in_rs;
    long long out_fs, out_bs, out_rs;
    int nrows;             // padded rows per subgrid (xM)
    int nfacets, xA;
    // facets grouped by off1 as in SumFinishFacetArgs: what is extracted for row r only depends on the facet's off1,
    // so it is computed once per group and stored into every facet of the group whose band covers the row
    int ngroups;
    int gstart[kSumFinishMaxFacets + 1];
    int fidx[kSumFinishMaxFacets];   // facet index into `out` of entry n
    int base0[kSumFinishMaxFacets];  // (xM/2 - m/2 + s'0_f) mod xM of entry n
    int gsp1[kSumFinishMaxFacets];   // s'1 of group g
    int ld_a[kSumFinishMaxBatch];    // (-(xM/2 - xA//2 + off1_b)) mod xM per subgrid
    const float* fn;
    const cx<float>* tw_m;
    const cx<float>* tw_x;
    const cx<float>* twc_m;  // compact copies, one-wave-per-row form (see SumFinishFacetArgs)
    const cx<float>* twc_x;
};

template <int LOGM, int LOGX>
__global__ __launch_bounds__((SFGeo<LOGM, LOGX>::NT)) void split_prepare_facets_kernel(const SplitFacetArgs A) {
    using S = SFGeo<LOGM, LOGX>;
    using GX = typename S::GX;
    using W = SFWide<LOGM, LOGX>;  // 4096-point rows: every wave extracts a different group (see SFWide)
    using GM = std::conditional_t<W::ON, typename W::GM, typename S::GM>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cx<float>* ex_m = reinterpret_cast<cx<float>*>(smem);
    cx<float>* acc = reinterpret_cast<cx<float>*>(smem + (W::ON ? W::LDS_M : S::LDS_M));
    constexpr int M = GM::N, X = GX::N, PM = GM::P, PX = GX::P, TR = S::TR;
    const int t = threadIdx.x % TR, rb = threadIdx.x / TR;
    const int b = blockIdx.y;
    const int row0 = blockIdx.x * S::RB;
    const int row = row0 + rb;
    const bool live = row < A.nrows;

    {   // prepare_subgrid along the row: zero-pad + shift on load, forward transform, result kept in LDS (array order)
        const cx<float>* __restrict__ in = A.in + (long long)b * A.in_bs + (long long)(live ? row : 0) * A.in_rs;
        const int lda = A.ld_a[b];
        cx<float> y[PX];
        static_for<0, PX>([&](auto vI) {
            constexpr int v = decltype(vI)::value;
            const int q = (((t + v * TR) ^ (X >> 1)) + lda) & (X - 1);
            const bool ok = live && q < A.xA;
            const cx<float> val = in[ok ? q : 0];
            y[v] = ok ? val : cx<float>{0.f, 0.f};
        });
        fft_phases<SFCompact<GX>, float, 0>(y, t, rb, false, acc, A.tw_x, [&](int e, cx<float> v) {
            acc[lds_pos<GX>(rb, e ^ (X >> 1), false)] = v;
        }, nullptr, A.twc_x);
        row_sync<GX>(false);
    }
    const float scale = 1.f / (float)M;
    constexpr int NS = 4;  // facets of a group served by one transform (more: the transform is repeated)
    if constexpr (W::ON) {
        static_assert(S::RB == 1, "one row per workgroup");
        const int lane = threadIdx.x & 63;
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        constexpr int NW = S::NT / 64;
        for (int g = wv; g < A.ngroups; g += NW) {  // wave-uniform; the prepared row in `acc` is only read
            int n = A.gstart[g];
            const int ne = A.gstart[g + 1];
            const int sp = A.gsp1[g];
            const int c1 = ((X >> 1) - (M >> 1) + sp) & (X - 1);
            while (n < ne) {
                // (slots filled through compile-time indices: a runtime index would put the pointers into scratch memory)
                cx<float>* outp[NS];
                int cnt = 0;
                static_for<0, NS>([&](auto sI) {
                    constexpr int sl = decltype(sI)::value;
                    outp[sl] = nullptr;
                    for (; n < ne && cnt == sl; n++) {  // the next covering facet of the group, if any
                        const int k = (row - A.base0[n]) & (X - 1);
                        if (live && k < M) {
                            outp[sl] = A.out + (long long)A.fidx[n] * A.out_fs + (long long)b * A.out_bs + (long long)k * A.out_rs;
                            cnt++;
                        }
                    }
                });
                if (cnt == 0) break;
                cx<float> x[PM];
                static_for<0, PM>([&](auto vI) {
                    constexpr int v = decltype(vI)::value;
                    const int q = (((lane + v * 64) ^ (M >> 1)) - sp) & (M - 1);
                    const cx<float> val = acc[lds_pos<GX>(0, (q + c1) & (X - 1), false)];
                    const float w = A.fn[q];
                    x[v] = cx<float>{val.x * w, -val.y * w};  // inverse transform = conj(FFT(conj(.)))
                });
                fft_phases<SFCompact<GM>, float, 0>(x, lane, wv, false, ex_m, A.tw_m, [&](int e, cx<float> v) {
                    const cx<float> o = cx<float>{v.x * scale, -v.y * scale};
                    static_for<0, NS>([&](auto sI) {
                        constexpr int sl = decltype(sI)::value;
                        if (outp[sl]) outp[sl][e ^ (M >> 1)] = o;
                    });
                }, nullptr, A.twc_m);
                __builtin_amdgcn_wave_barrier();  // this wave's quarter of ex_m is reused by its next transform
            }
        }
        return;
    }
    for (int g = 0; g < A.ngroups; g++) {  // workgroup-uniform
        int n = A.gstart[g];
        const int ne = A.gstart[g + 1];
        const int sp = A.gsp1[g];
        const int c1 = ((X >> 1) - (M >> 1) + sp) & (X - 1);
        while (n < ne) {
            int fs[NS];
            int cnt = 0;
            for (; n < ne && cnt < NS; n++) {  // workgroup-uniform scan
                const int base = A.base0[n];
                bool any = false;
                for (int q = 0; q < S::RB; q++) any = any || (((row0 + q - base) & (X - 1)) < M && row0 + q < A.nrows);
                if (any) fs[cnt++] = n;
            }
            if (cnt == 0) break;
            cx<float>* outp[NS];
            static_for<0, NS>([&](auto sI) {
                constexpr int sl = decltype(sI)::value;
                outp[sl] = nullptr;
                if (sl < cnt) {  // uniform
                    const int nn = fs[sl];
                    const int k = (row - A.base0[nn]) & (X - 1);
                    if (live && k < M)
                        outp[sl] = A.out + (long long)A.fidx[nn] * A.out_fs + (long long)b * A.out_bs + (long long)k * A.out_rs;
                }
            });
            cx<float> x[PM];
            static_for<0, PM>([&](auto vI) {
                constexpr int v = decltype(vI)::value;
                const int q = (((t + v * TR) ^ (M >> 1)) - sp) & (M - 1);
                const cx<float> val = acc[lds_pos<GX>(rb, (q + c1) & (X - 1), false)];
                const float w = A.fn[q];
                x[v] = cx<float>{val.x * w, -val.y * w};  // inverse transform = conj(FFT(conj(.)))
            });
            fft_phases<SFCompact<GM>, float, 0>(x, t, rb, false, ex_m, A.tw_m, [&](int e, cx<float> v) {
                const cx<float> o = cx<float>{v.x * scale, -v.y * scale};
                static_for<0, NS>([&](auto sI) {
                    constexpr int sl = decltype(sI)::value;
                    if (outp[sl]) outp[sl][e ^ (M >> 1)] = o;
                });
            }, nullptr, A.twc_m);
            row_sync<GX>(false);  // ex_m is reused by the next transform
        }
    }
}

int launch_split_prepare_facets(int logm, int logx, const SplitFacetArgs& a, int nbatch, hipStream_t s);
int launch_sum_finish_facets(int logm, int logx, const SumFinishFacetArgs& a, int nbatch, hipStream_t s);
int launch_sum_finish_rows(int logm, int logx, const SumFinishArgs& a, int nbatch, hipStream_t s);
int init_sum_finish_rows();
bool sum_finish_supported(int logm, int logx);
int launch_axis1_rows(int logm, const Axis1RowsArgs& a, int nfacets, hipStream_t s);

}  // namespace swf
