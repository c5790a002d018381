// SwiFTly on MI355X: the WHOLE-ROW form of K1 (prepare_facet along the contiguous axis of a 32768-point padded row,
// reference fourier_transform/core.py:212-222, band store of DESIGN.md section 3) -- r6.
//
// row_pass_band_kernel (swiftly_rowpass.h) works on a row as TWO workgroups, one per output parity; each of them loads
// and windows the whole row, so per row 540 KB go through the L2 -> L1 path for 180 KB of facet data, and the two
// resident workgroups of a CU overlap their load and butterfly phases only as far as the dispatcher happens to stagger
// them (r4 timeline: 40 % of a load phase).  Here ONE persistent 512-thread workgroup per CU owns whole rows:
//
//   * every element and every window value is loaded ONCE; the radix-2 decimation in frequency that splits the row
//     into its even-output half A = a + b and its odd-output half B = (a - b) W_N^j happens in registers;
//   * the two 16384-point halves share ONE complex64 exchange buffer (136 KB) alternately: while half B sits in LDS,
//     half A is in registers being transformed, and vice versa -- the LDS traffic of one half runs under the
//     butterflies of the other, which is what the second workgroup of the two-workgroup form was there for;
//   * only one half is live in registers during a butterfly phase (64 VGPRs), so at 256 VGPRs per lane (8 waves per CU)
//     there is room for the NEXT row's loads: PF of its NG load groups (window quad + two data quads each) are
//     requested right after the current row has been folded, the rest when half A has been stored -- the whole next
//     row is in flight or landed while the current one is transformed, and the loop never waits for HBM latency.
//
// Arithmetic per element is the same sequence of operations as in row_pass_band_kernel<RGeoPreC<14,5>, ..., W4>
// (same products, same tables): the outputs are bit-identical (tests/test_hip_band_pipeline_gpu.py).
#pragma once
#include "swiftly_rowpass.h"
#include "swiftly_sumfinish.h"  // SFCompact: the m-point engine of the window epilogue

// timing-only ablations of the whole-row kernel (tools/build_variant.sh ... -DSWF_WHOLE_ABL=mask; results are wrong):
// 1 = no memory behind the loads (empty descriptors), 2 = no stores, 4 = no LDS traffic (barriers stay), 8 = no butterflies
#ifndef SWF_WHOLE_ABL
#define SWF_WHOLE_ABL 0
#endif
// data-quad pairs of the next row requested under the gathers of exchange 1 of A, exchange 1 of B, exchange 2 of A
// butterfly stages of a radix-32 / radix-16 phase that run in the scheduling region of the OTHER half's scatter (with the
// inter-phase twiddles); the rest runs under that half's gather
#ifndef SWF_WHOLE_SPLIT5
#define SWF_WHOLE_SPLIT5 1
#endif
#ifndef SWF_WHOLE_SPLIT4
#define SWF_WHOLE_SPLIT4 2
#endif
// VALU instructions the scheduler places behind every LDS write of a scatter (0: its own order, all writes first)
#ifndef SWF_WHOLE_SGBW
#define SWF_WHOLE_SGBW 0
#endif
#ifndef SWF_WHOLE_E1
#define SWF_WHOLE_E1 2
#endif
#ifndef SWF_WHOLE_E2
#define SWF_WHOLE_E2 2
#endif
#ifndef SWF_WHOLE_E3
#define SWF_WHOLE_E3 2
#endif
#ifndef SWF_WHOLE_P2_RECOMPUTE
#define SWF_WHOLE_P2_RECOMPUTE 1
#endif

namespace swf {

// 512 threads x 32 points of a 16384-point half, complex64 exchange (8-byte elements, one pad element per 16)
struct RGeoWhole : RGeo<14, 5, false, true> {
    static constexpr bool PRELOAD_TW = true;
    static constexpr bool COMPACT_TW = true;
};

// In-register radix-2^LOGR DIF network of swiftly_fft.h::fft_reg, stages [S0, S1) only (stage 0 = the first, half = 2^(LOGR-1))
template <typename R, int LOGR, int STR, int OFF, int PTOT, int S0, int S1>
__device__ __forceinline__ void fft_reg_stages(cx<R> (&x)[PTOT]) {
    constexpr int RAD = 1 << LOGR;
    static_for<S0, S1>([&](auto sI) {
        constexpr int s = LOGR - 1 - decltype(sI)::value;
        constexpr int half = 1 << s;
        static_for<0, RAD / 2>([&](auto bI) {
            constexpr int b = decltype(bI)::value;
            constexpr int blk = b / half, k = b % half;
            constexpr int i0 = OFF + (blk * 2 * half + k) * STR;
            constexpr int i1 = i0 + half * STR;
            cx<R> a = x[i0], c = x[i1];
            x[i0] = a + c;
            x[i1] = sub_mul_w64<R, k*(32 / half)>(a, c);
        });
    });
}
// One Stockham phase (swiftly_fft.h::phase_compute with preloaded compact twiddle values) in two parts: PART 0 = the
// inter-phase twiddles and the first SPLIT butterfly stages, PART 1 = the remaining stages.  Same operations in the same
// order per element as phase_compute.
template <class G, int LOGNS, int LOGR, int PART, int SPLIT>
__device__ __forceinline__ void phase_compute_part(cx<float> (&x)[G::P], const cx<float>* __restrict__ tw, const cx<float>* pre) {
    constexpr int RAD = 1 << LOGR, NB = G::P / RAD;
    static_assert(LOGNS == 0 || NB == 1, "twiddled phases of this kernel have one block per lane");
    static_for<0, NB>([&](auto uI) {
        constexpr int u = decltype(uI)::value;
        if constexpr (PART == 0) {
            if constexpr (LOGNS > 0) twiddle_inputs<float, LOGR, NB, u, G::P, G::N, lean_tw_of<G>::value>(x, tw, 0, pre);
            fft_reg_stages<float, LOGR, NB, u, G::P, 0, SPLIT>(x);
        } else {
            fft_reg_stages<float, LOGR, NB, u, G::P, SPLIT, LOGR>(x);
        }
    });
}

// values the optimiser must treat as new (no common subexpressions with what was computed from the old ones)
template <int K>
__device__ __forceinline__ void opaque_values(cx<float> (&v)[K]) {
    static_for<0, K>([&](auto bI) {
        float a = v[decltype(bI)::value].x, b = v[decltype(bI)::value].y;
        asm volatile("" : "+v"(a), "+v"(b));
        v[decltype(bI)::value] = cx<float>{a, b};
    });
}

// WIN (RowPassArgs::win_full): the band of a row is staged in the exchange buffer (free once the last gather is done) and the
// epilogue finishes the contiguous axis for every planned window (see RowPassArgs) -- the K1 of the axis-1-first pipeline.
constexpr int kWholeMaxWindows = 256;
// exchange buffer | Fn | window table of the window epilogue
constexpr size_t kWholeWinLds = RGeoWhole::LDS_BYTES + 512 * sizeof(float) + kWholeMaxWindows * sizeof(int);
template <int NSEG, bool WIN = false>
__global__ __launch_bounds__(512, 2) void row_pass_whole_kernel(const RowPassArgs A, const cx<float>* __restrict__ gin,
                                                                cx<float>* __restrict__ gout,
                                                                const cx<float>* __restrict__ tw,
                                                                const cx<float>* __restrict__ tw_full,
                                                                const float* __restrict__ row_win,
                                                                const int* __restrict__ in_rowmap) {
    using G = RGeoWhole;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int P = G::P, T = G::T, H = G::N, N = 2 * G::N;
    constexpr int R1 = 16, SEG = H / R1, SEGLEN = 2 * T;      // radix-16 first phase: points j + r * SEG, j = 2t + u
    constexpr int NS = NSEG, NB1 = NS - R1;
    static_assert(P == 32 && T == 512 && NS >= R1 && NS % 2 == 0 && NS <= 32, "forward K1 geometry");
    constexpr int NGA = NB1, NG = NB1 + (R1 - NB1) / 2;       // load groups: (r, r + 16) for r < NB1, then (r, r + 1)
    constexpr int LOGR1 = G::LOGN % G::LOGP;                   // 4
    constexpr int LNS1 = LOGR1, LNS2 = LOGR1 + G::LOGP;        // phase boundaries 4 and 9
    constexpr int LNS = G::LOGN - G::LOGP;                     // 9: outputs e = t + (r << LNS)
    cx<float>* buf = reinterpret_cast<cx<float>*>(smem);
    const int t = threadIdx.x;
    const int rot = A.seg_rot * SEGLEN;

    // -- constants of the lane (the same for every row) -----------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rs_w4 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.ld_win4), (short)0, (NS / 2) * T * 16, 0x00020000);
    const unsigned base8 = (unsigned)((2 * t + A.ld_a + (N >> 1) + rot) & (N - 1)) << 3;
    // inter-half twiddle of the odd outputs: W_N^j, j = 2t + u + SEG r = W_N^(2t+u) * W_64^(2r)
    const f32x4 wt = *reinterpret_cast<const f32x4*>(tw_full + 2 * t);
    const cx<float> w0 = {wt.x, wt.y}, w1 = {wt.z, wt.w};
    // rotated input: output k = 2 e + h carries W_N^(rot k) = W_N^(rot (2 t + h))
    const cx<float> rphi0 = tw_full[(unsigned)(rot * (2 * t)) & (unsigned)(N - 1)];
    const cx<float> rphi1 = tw_full[(unsigned)(rot * (2 * t + 1)) & (unsigned)(N - 1)];
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int dw0 = ((wave << 7) + (N >> 1) - A.band_start) & (N - 1);   // d of lane 0, output r = 0, half 0
    const unsigned region0 = (unsigned)(((0 ^ A.band_start) & 1) * A.band_half) << 3;
    const unsigned region1 = (unsigned)(((1 ^ A.band_start) & 1) * A.band_half) << 3;

    auto row_rsrc = [&](int row, bool live) {
        int in_row = row;
        if (A.rm_mod > 0) {
            int r1 = row + A.rm_inner;
            if (r1 >= A.rm_mod) r1 -= A.rm_mod;
            r1 += A.rm_outer;
            if (r1 >= A.rm_full) r1 -= A.rm_full;
            in_row = r1;
        }
        if (in_rowmap) in_row = in_rowmap[in_row];
        in_row = __builtin_amdgcn_readfirstlane(in_row);   // wave-uniform: the descriptor must live in SGPRs
        const bool dead = in_row < 0 || !live;   // absent from a compacted input / no further row: an empty descriptor
        if (in_row < 0) in_row = 0;
        const char* inb = reinterpret_cast<const char*>(gin + (long long)in_row * A.in_pitch);
        const unsigned valid = (dead || (SWF_WHOLE_ABL & 1)) ? 0u : (unsigned)A.ld_len;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(inb), (short)0, (int)(valid << 3), 0x00020000);
    };

    f32x4 gw[NG], g0[NG], g1[NG];
    auto issue_win = [&](auto gI, int t) {
        constexpr int g = decltype(gI)::value;
        gw[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w4, (g * T + t) << 4, 0, 0));
    };
    auto issue_data = [&](auto gI, unsigned base8, const __amdgpu_buffer_rsrc_t& rs_in) {
        constexpr int g = decltype(gI)::value;
        constexpr int r0 = g < NGA ? g : NGA + 2 * (g - NGA);   // first slot of the group
        constexpr int s0 = r0, s1 = g < NGA ? r0 + R1 : r0 + 1; // its two segments
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

    // workgroup barrier that the instruction scheduler may not move work across: the butterflies of one half are meant to
    // run between the scatter of the other half and the barrier behind it (under the LDS writes), not behind the barrier
    auto wg_sync = [] {
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    };
    // "n loads, each followed by `valu` VALU instructions": spreads the requests of the next row through a butterfly phase (a
    // load instruction occupies the CU's address path for 16 cycles; issued back to back by all eight waves they stall)
    auto spread_loads = [](auto nI, auto vI) {
        static_for<0, decltype(nI)::value>([](auto) {
            __builtin_amdgcn_sched_group_barrier(0x002, decltype(vI)::value, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        });
    };
    // "one LDS write, then `valu` VALU instructions", 16 times: a ds_write_b128 occupies its wave for ~13 cycles; with the
    // eight waves of the workgroup in the same place of the same instruction stream a block of 16 writes stalls all of them
    auto spread_writes = [](auto vI) {
#if SWF_WHOLE_SGBW
        static_for<0, 16>([](auto) {
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, decltype(vI)::value, 0);
        });
#endif
    };
    int row = blockIdx.x;
    if (row >= A.nrows) return;
    if constexpr (WIN) {
        // Fn (m floats) BEHIND the exchange buffer: the row exchanges never touch it (kWholeWinLds bytes of LDS for WIN launches)
        float* fn_w = reinterpret_cast<float*>(smem + G::LDS_BYTES);
        if (t < 512) fn_w[t] = A.win_fn[t];
        int* wd = reinterpret_cast<int*>(fn_w + 512);   // ... and the window table (host-checked: nwin <= kWholeMaxWindows)
        if (t < A.nwin) wd[t] = A.win_d[t];
    }
#if SWF_TRACE
#define SWF_WTRACE(id)                                                                                             \
    do {                                                                                                           \
        asm volatile("" ::: "memory");                                                                             \
        if (t == 0 && row < kTraceBlocks) swf_trace_buf[row * kTracePoints + (id)] = __builtin_readcyclecounter(); \
        asm volatile("" ::: "memory");                                                                             \
    } while (0)
#else
#define SWF_WTRACE(id) ((void)0)
#endif
    float scale_next = A.scale;
    {
        const __amdgpu_buffer_rsrc_t rs0 = row_rsrc(row, true);
        static_for<0, NG>([&](auto gI) {
            issue_win(gI, t);
            issue_data(gI, base8, rs0);
        });
        if (row_win) scale_next *= row_win[row];
        __builtin_amdgcn_sched_barrier(0);
    }
    // data groups of the next row requested under the first three gathers (the rest, and the window quads, when half A of
    // the current row has been stored)
    constexpr int E1 = SWF_WHOLE_E1 < NG ? SWF_WHOLE_E1 : NG;
    constexpr int E2 = E1 + SWF_WHOLE_E2 < NG ? E1 + SWF_WHOLE_E2 : NG;
    constexpr int E3 = E2 + SWF_WHOLE_E3 < NG ? E2 + SWF_WHOLE_E3 : NG;
    for (;;) {
        // Row-invariant lane / wave values are RE-DERIVED in every iteration from opaque copies: hoisted out of the loop
        // (LDS addresses, the 33 load offsets, 64 store bases and their in-band tests) they would cost 120 VGPRs and 430
        // SGPRs of spills; recomputing them is a few dozen instructions per row.
        SWF_WTRACE(0);
        int tt = t, dwi = dw0;
        unsigned b8 = base8;
        asm volatile("" : "+v"(tt), "+v"(b8), "+s"(dwi));
        const int lane2 = (tt & 63) << 1;
        // -- fold the landed row into its two halves ---------------------------------------------------------------
        cx<float> xa[P], xb[P];
        static_for<0, NG>([&](auto gI) {
            constexpr int g = decltype(gI)::value;
            constexpr int r0 = g < NGA ? g : NGA + 2 * (g - NGA);
            cx<float> a0[2], a1[2];
            wmul(g0[g], f32x2{gw[g].x, gw[g].y}, a0[0], a0[1]);
            wmul(g1[g], f32x2{gw[g].z, gw[g].w}, a1[0], a1[1]);
            static_for<0, 2>([&](auto uI) {
                constexpr int u = decltype(uI)::value;
                if constexpr (g < NGA) {
                    xa[u + 2 * r0] = pkc(__builtin_elementwise_fma(pkv(a1[u]), f32x2{1.f, 1.f}, pkv(a0[u])));
                    xb[u + 2 * r0] = pkc(__builtin_elementwise_fma(pkv(a1[u]), f32x2{-1.f, -1.f}, pkv(a0[u])));
                } else {
                    xa[u + 2 * r0] = a0[u];
                    xa[u + 2 * (r0 + 1)] = a1[u];
                    xb[u + 2 * r0] = a0[u];
                    xb[u + 2 * (r0 + 1)] = a1[u];
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        SWF_WTRACE(1);
        // -- this row's output, the next row's input (scalar loads one row ahead) -------------------------------------
        const int urow = __builtin_amdgcn_readfirstlane(row);
        char* __restrict__ outb = reinterpret_cast<char*>(gout + (long long)urow * A.out_pitch);
        const float scale = scale_next;
        const f32x2 sc = {scale, scale};
        const int next = row + (int)gridDim.x;
        const bool more = next < A.nrows;
        const __amdgpu_buffer_rsrc_t rsn = row_rsrc(more ? next : row, more);
        scale_next = A.scale;
        if (row_win) scale_next *= row_win[more ? next : urow];

        // Band store of one half (output parity h): the lane's outputs are e = t + (r << LNS), cyclic band distance
        // d_r = (2 t + h + N/2 - band_start + 1024 r) mod N, physical column (d_r >> 1) of the half's region.  Which r a wave
        // keeps at all is decided on the scalar unit (one not-taken branch per kept output, one taken branch per dropped
        // one); WHICH LANES of a kept output are inside the band is left to the range check of a buffer store whose
        // descriptor covers exactly the half's kept columns -- no per-lane compare, no exec juggling, and one store path
        // (the two-workgroup kernel's split into a wave-uniform "all inside" path and a masked one compiled to ~25 scalar
        // instructions and three branches per output here, where only two waves per SIMD hide a taken branch).
        auto store_half = [&](const cx<float> (&x)[P], int h, cx<float> rphi, unsigned region, auto&& between) {
            const int par = (h ^ A.band_start) & 1;                         // parity of d for this half
            const int ncol = (A.band_len - par + 1) >> 1;                   // columns q = d >> 1 with 2 q + par < band_len
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(outb + region, (short)0, ncol << 3, 0x00020000);
            const int dw = (dwi + h) & (N - 1);
            const unsigned off0 = (unsigned)(((dw + lane2) & (N - 1)) >> 1) << 3;  // byte offset of output r = 0
            phase_scatter<G, float, LNS2, G::LOGP>(x, tt, [&](int, cx<float> v, auto sI) {
                constexpr int r = decltype(sI)::value;
                between(sI);
                const int base = (dw + (r << (LNS + 1))) & (N - 1);  // wave-uniform
                const bool none_in = base >= A.band_len && base + 126 < N;
                if (none_in || (SWF_WHOLE_ABL & 2)) return;
                v = cmul(v, rphi);
                f32x2 val;
                const f32x2 vv = pkv(v);
                asm("v_pk_mul_f32 %0, %1, %2 neg_hi:[1,0]" : "=v"(val) : "v"(vv), "v"(sc));
                if constexpr (WIN) {
                    // the band of the row is STAGED in the exchange buffer (free by now), laid out like a band buffer row
                    const int d = (base + lane2) & (N - 1);
                    f32x2* __restrict__ st = reinterpret_cast<f32x2*>(buf) + par * A.band_half + (d >> 1);
                    if (base + 126 < A.band_len) *st = val;          // wave-uniform: every lane inside
                    else if (d < A.band_len) *st = val;
                } else {
                    const unsigned off = (off0 + (unsigned)(r << (LNS + 3))) & (unsigned)((N << 2) - 1);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, val), rs_out, (int)off, 0, 0);
                }
            });
        };
        auto scatter1 = [&](cx<float> (&x)[P]) {  // output of the radix-16 phase (adjacent virtual threads per lane)
            if constexpr (SWF_WHOLE_ABL & 4) return;
            exchange_pass<G, float, 0, LOGR1, true>(x, tt, 0, false, buf, [](cx<float> v) { return v; }, true);
        };
        auto scatter2 = [&](cx<float> (&x)[P]) {
            if constexpr (SWF_WHOLE_ABL & 4) return;
            exchange_pass<G, float, LNS1, G::LOGP, false>(x, tt, 0, false, buf, [](cx<float> v) { return v; }, true);
        };
        auto gather = [&](cx<float> (&x)[P]) {
            if constexpr (SWF_WHOLE_ABL & 4) return;
            gather_pass<G, float>(tt, 0, false, buf, [&](auto vI, cx<float> val) { x[decltype(vI)::value] = val; });
        };
        auto butterflies = [&](auto lnsI, auto logrI, cx<float> (&x)[P], const cx<float>* pre) {
            if constexpr (!(SWF_WHOLE_ABL & 8)) phase_compute<G, float, decltype(lnsI)::value, decltype(logrI)::value>(x, tt, tw, pre, A.twc);
        };
        using I0 = std::integral_constant<int, 0>;
        using IR1 = std::integral_constant<int, LOGR1>;
        using IP = std::integral_constant<int, G::LOGP>;
        using IL1 = std::integral_constant<int, LNS1>;
        using IL2 = std::integral_constant<int, LNS2>;

        // The schedule of a row: the exchanges of the two halves alternate through the one buffer, and every GATHER (32
        // 8-byte LDS reads per lane, which run at a fraction of the LDS rate with only eight waves on the CU) has the other
        // half's butterfly phase behind it in the same scheduling region, with the next row's loads spread through it.
        auto part = [&](auto lnsI, auto logrI, auto partI, cx<float> (&x)[P], const cx<float>* pre) {
            constexpr int LR = decltype(logrI)::value;
            if constexpr (!(SWF_WHOLE_ABL & 8))
                phase_compute_part<G, decltype(lnsI)::value, LR, decltype(partI)::value, LR == G::LOGP ? SWF_WHOLE_SPLIT5 : SWF_WHOLE_SPLIT4>(x, tw, pre);
        };
        using P0_ = std::integral_constant<int, 0>;
        using P1_ = std::integral_constant<int, 1>;
        butterflies(I0{}, IR1{}, xa, nullptr);                     // phase 0 of A (radix 16, no twiddles)
        wg_sync();   // the previous row's last gather is complete in every wave
        SWF_WTRACE(2);
        scatter1(xa);                                              // ... under the inter-half twiddle + first part of phase 0 of B
        static_for<0, R1>([&](auto rI) {
            constexpr int r = decltype(rI)::value;
            xb[2 * r] = mul_w64<float, 64 * SEG / N * r>(cmul(xb[2 * r], w0));
            xb[2 * r + 1] = mul_w64<float, 64 * SEG / N * r>(cmul(xb[2 * r + 1], w1));
        });
        part(I0{}, IR1{}, P0_{}, xb, nullptr);
        spread_writes(std::integral_constant<int, SWF_WHOLE_SGBW>{});
        cx<float> nx[G::LOGP];
        load_compact<G, float, LNS1, G::LOGP>(nx, tt, A.twc);
        SWF_WTRACE(3);
        wg_sync();
        gather(xa);                                                // ... under the rest of phase 0 of B
        part(I0{}, IR1{}, P1_{}, xb, nullptr);
#ifndef SWF_WHOLE_NOPF
        static_for<0, E1>([&](auto gI) { issue_data(gI, b8, rsn); });
        spread_loads(std::integral_constant<int, 2 * E1>{}, std::integral_constant<int, 8>{});
#endif
        wg_sync();
        SWF_WTRACE(4);
        scatter1(xb);                                              // ... under twiddles + first stages of phase 1 of A
        part(IL1{}, IP{}, P0_{}, xa, nx);
        spread_writes(std::integral_constant<int, SWF_WHOLE_SGBW>{});
        SWF_WTRACE(5);
        wg_sync();
        gather(xb);                                                // ... under the rest of phase 1 of A
        part(IL1{}, IP{}, P1_{}, xa, nx);
#ifndef SWF_WHOLE_NOPF
        static_for<E1, E2>([&](auto gI) { issue_data(gI, b8, rsn); });
        spread_loads(std::integral_constant<int, 2 * (E2 - E1)>{}, std::integral_constant<int, 16>{});
#endif
        wg_sync();
        SWF_WTRACE(6);
        scatter2(xa);                                              // ... under twiddles + first stages of phase 1 of B
        // (the 26 composite twiddle powers of a radix-32 phase are identical asm statements for the two halves, which the
        // compiler would compute once and keep alive -- 52 VGPRs across the gather where both halves are in registers;
        // opaque copies of the five table values make it build them again: 52 more instructions, 52 fewer registers)
        opaque_values(nx);
        part(IL1{}, IP{}, P0_{}, xb, nx);
        spread_writes(std::integral_constant<int, SWF_WHOLE_SGBW>{});
        SWF_WTRACE(7);
        wg_sync();
        gather(xa);                                                // ... under the rest of phase 1 of B
        part(IL1{}, IP{}, P1_{}, xb, nx);
#ifndef SWF_WHOLE_NOPF
        static_for<E2, E3>([&](auto gI) { issue_data(gI, b8, rsn); });
        spread_loads(std::integral_constant<int, 2 * (E3 - E2)>{}, std::integral_constant<int, 16>{});
#endif
        load_compact<G, float, LNS2, G::LOGP>(nx, tt, A.twc);
        wg_sync();
        SWF_WTRACE(8);
        scatter2(xb);                                              // ... under twiddles + first stages of phase 2 of A
        part(IL2{}, IP{}, P0_{}, xa, nx);
        spread_writes(std::integral_constant<int, SWF_WHOLE_SGBW>{});
        SWF_WTRACE(9);
        wg_sync();
        gather(xb);                                                // ... under the rest of phase 2 + the band store of A
        part(IL2{}, IP{}, P1_{}, xa, nx);
        // The rest of the next row's requests (NL = 2 (NG - E3) data quads, NG window quads) ride on the band store of A, one
        // per output slot from slot 32 - NL - NG on: every slot frees two registers (kept or not), a request takes four, and
        // between two requests lie the ten-odd instructions of a slot -- issued as ONE block behind the store they backed
        // up the CU's address path for 3000 cycles (21 requests x 8 waves x 16 cycles), into the next row's first phase.
        constexpr int NLATE = 2 * (NG - E3) + NG, SLOT0 = 32 - NLATE;
        static_assert(SLOT0 >= 8, "the late requests need registers the band store has already freed");
        if constexpr (WIN) wg_sync();   // every wave has gathered B: the exchange buffer becomes the band stage
        store_half(xa, 0, rphi0, region0, [&](auto sI) {
#ifndef SWF_WHOLE_NOPF
            constexpr int i = decltype(sI)::value - SLOT0;
            if constexpr (i >= 0 && i < 2 * (NG - E3)) {
                constexpr int g = E3 + i / 2;
                constexpr int r0 = g < NGA ? g : NGA + 2 * (g - NGA);
                constexpr int sg = (i & 1) ? (g < NGA ? r0 + R1 : r0 + 1) : r0;
                const unsigned o = (b8 + (unsigned)((sg * SEG) << 3)) & (unsigned)((N << 3) - 1);
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsn, (int)o, 0, 0));
                if constexpr (i & 1) g1[g] = v; else g0[g] = v;
            } else if constexpr (i >= 2 * (NG - E3) && i < NLATE) {
                issue_win(std::integral_constant<int, i - 2 * (NG - E3)>{}, tt);
            }
#endif
        });
        __builtin_amdgcn_sched_barrier(0);
        SWF_WTRACE(10);
#if SWF_WHOLE_P2_RECOMPUTE
        opaque_values(nx);
#endif
        part(IL2{}, IP{}, P0_{}, xb, nx);
        part(IL2{}, IP{}, P1_{}, xb, nx);
        store_half(xb, 1, rphi1, region1, [](auto) {});
        if constexpr (WIN) {
            // -- window epilogue: the row's band is staged once every wave has passed the barrier ----------------------------
            // Nothing in the window loop reads global memory (a load behind the previous window's stores waits for THEIR
            // completion: one counter for both): Fn and the window table sit in LDS behind the exchange buffer, the six twiddle
            // table values of the lane in registers.  A round of eight windows (one per wave) costs 1.8 us per row either way --
            // ~200 vector instructions and 40 LDS operations per window with two waves per SIMD in step, i.e. the epilogue is
            // bound by its own arithmetic and exchanges, and a 25th window rides in the slack of the waves that have only three (tools/time_k1_window_rows.py).
            using GM = SFCompact<Geo<float, 9, 3, G::NT, false>>;   // m = 512 (host-checked): one wave per transform, 8 points per lane
            static_assert(GM::T == 64 && GM::WAVE_ROWS && GM::LOGP == 3, "window transform geometry");
            constexpr int M = GM::N;
            wg_sync();
            const int lane = tt & 63;
            const int wv = __builtin_amdgcn_readfirstlane(tt >> 6);
            const cx<float>* __restrict__ stage = buf;
            cx<float>* ex = buf + 2 * A.band_half;       // the m-point exchanges behind the stage (host-checked: it fits)
            const float* __restrict__ fn_l = reinterpret_cast<const float*>(smem + G::LDS_BYTES);
            const int* __restrict__ wd_l = reinterpret_cast<const int*>(fn_l + 512);
            cx<float>* __restrict__ orow = gout + (long long)urow * A.out_pitch;
            const int sp = A.win_sp;
            cx<float> pre1[3], pre2[3];
            load_compact<GM, float, 3, 3>(pre1, lane, A.win_twc_m);
            load_compact<GM, float, 6, 3>(pre2, lane, A.win_twc_m);
            float wgt[GM::P];   // Fn at the lane's outputs e = lane + 64 r
            static_for<0, GM::P>([&](auto rI) {
                constexpr int r = decltype(rI)::value;
                wgt[r] = fn_l[(((lane + 64 * r) ^ (M >> 1)) - sp) & (M - 1)];
            });
            for (int w = wv; w < A.nwin; w += G::NT / 64) {   // wave-uniform
                const int D = __builtin_amdgcn_readfirstlane(wd_l[w]);
                const int s = (A.band_start + D - (N / 2 - M / 2)) & (M - 1);   // off1 yN / N of the wave, mod m
                // The lane's eight elements (plain index lane + 64 v = centred element lane + 64 (v ^ 4)) are the window elements
                // i_v = (r0 + 64 (v ^ 4)) mod m with r0 = (lane - s) mod m = 64 h + l: they share their parity, and their
                // band distances D + i_v differ by multiples of 64, i.e. their stage columns by multiples of 32 --
                // one base address per window, three instructions per element.
                cx<float> xw[GM::P];
                {
                    const int r0 = (lane - s) & (M - 1), h = r0 >> 6, l = r0 & 63;
                    const int d0 = D + l;                              // band distance of element 64 k + l is d0 + 64 k
                    const cx<float>* __restrict__ g0 = stage + (d0 & 1) * A.band_half + (d0 >> 1);
                    static_for<0, GM::P>([&](auto vI) {
                        constexpr int v = decltype(vI)::value;
                        xw[v] = g0[((h + (v ^ 4)) & 7) << 5];
                    });
                }
                phase_compute<GM, float, 0, 3>(xw, lane, A.win_tw_m, nullptr, A.win_twc_m);
                phase_exchange<GM, float, 0, 3>(xw, lane, wv, false, ex);
                phase_compute<GM, float, 3, 3>(xw, lane, A.win_tw_m, pre1, A.win_twc_m);
                phase_exchange<GM, float, 3, 3>(xw, lane, wv, false, ex);
                phase_compute<GM, float, 6, 3>(xw, lane, A.win_tw_m, pre2, A.win_twc_m);
                // outputs e = lane + 64 r = centred lane + 64 (r ^ 4): kk = (centred - s'1) mod m, window-band position
                // i2 = (kk - s) mod m = (t0 + 64 (r ^ 4)) mod m with t0 = (lane - s'1 - s) mod m -- the same structure
                const int t0 = (lane - sp - s) & (M - 1), h2 = t0 >> 6, l2 = t0 & 63;
                cx<float>* __restrict__ o0 = orow + (long long)w * A.win_pitch + (l2 & 1) * (M >> 1) + (l2 >> 1);
                phase_scatter<GM, float, 6, 3>(xw, lane, [&](int, cx<float> v, auto rI) {
                    constexpr int r = decltype(rI)::value;
                    o0[((h2 + (r ^ 4)) & 7) << 5] = pkc(pkv(v) * f32x2{wgt[r], wgt[r]});
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        SWF_WTRACE(11);
        if (!more) break;
        row = next;
    }
}

}  // namespace swf
