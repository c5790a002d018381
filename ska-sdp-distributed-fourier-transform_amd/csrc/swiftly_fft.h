// SwiFTly on MI355X (gfx950): workgroup-level FFT engine.
//
// One workgroup transforms RB rows of length N = 2^LOGN held entirely in
// registers (P = 2^LOGP points per thread, T = N/P threads per row) with a
// Stockham autosort schedule: each phase does a radix-2^LOGR DFT in registers
// and exchanges data through LDS, so input and output are both in natural
// order and no bit-reversal pass touches memory.  The first phase reads
// straight from global memory through an index map (window x zero-pad x
// cyclic shift x centred-FFT shift all folded into the map) and the last phase
// writes straight to global memory through a second map (shift x crop x window
// x accumulate), which is how every SwiFTly primitive (reference
// fourier_transform/core.py:189-484) becomes ONE kernel with one HBM read and
// one HBM write per element.
//
// Wavefronts are 64 wide; LDS exchange buffers are padded by one element per
// 128 B (32 banks x 4 B) so that the strided scatter of a phase stays
// conflict-free for ds_write_b32/b64/b128 alike.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace swf {

template <typename R>
struct cx {
    R x, y;
};

template <typename R>
__device__ __forceinline__ cx<R> operator+(cx<R> a, cx<R> b) {
    return {a.x + b.x, a.y + b.y};
}
template <typename R>
__device__ __forceinline__ cx<R> operator-(cx<R> a, cx<R> b) {
    return {a.x - b.x, a.y - b.y};
}
template <typename R>
__device__ __forceinline__ cx<R> cmul(cx<R> a, cx<R> b) {
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------------
// Hand-packed complex64 arithmetic.  A complex value is one 64-bit VGPR pair; add / subtract are one v_pk_add_f32, a
// general product is v_pk_mul_f32 + v_pk_fma_f32 with op_sel / neg modifiers doing the swaps and sign changes (the
// compiler's own lowering of the same expressions needs a v_xor per product because it does not fold a one-lane
// negation into neg_lo), and the (a - c) * {-i, W8, W8^3} butterflies fold their rotation into the subtraction.
// VOP3P modifiers: op_sel[i] / op_sel_hi[i] pick the half of source i that feeds the low / high result lane,
// neg_lo / neg_hi negate source i for that lane.
// Measured on MI355X (r2, same-box A/B against the plain scalar forms): the band row kernel 1675 instead of 2416 VALU
// instructions per wave, 117 instead of 128 VGPRs and no spills in any geometry, K1 2.03 -> 1.93 ms per facet; column
// passes and the subgrid-side kernels unchanged (bandwidth / LDS bound).  Forms that were measured and dropped (numbers
// in DESIGN.md section 4): the compiler's own 2-vector lowering of the product (column passes 18 % slower: v_mov
// shuffles), both halves of a product in ONE asm statement (r5: drops 92 `s_nop 0` per wave, K1 1.644 vs 1.644 ms).
__device__ __forceinline__ f32x2 pkv(cx<float> a) { return f32x2{a.x, a.y}; }
__device__ __forceinline__ cx<float> pkc(f32x2 v) { return {v.x, v.y}; }
__device__ __forceinline__ cx<float> operator+(cx<float> a, cx<float> b) { return pkc(pkv(a) + pkv(b)); }
__device__ __forceinline__ cx<float> operator-(cx<float> a, cx<float> b) { return pkc(pkv(a) - pkv(b)); }
template <>
__device__ __forceinline__ cx<float> cmul<float>(cx<float> a, cx<float> b) {
    f32x2 t, r;
    const f32x2 av = pkv(a), bv = pkv(b);
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(av), "v"(bv));  // (a.x b.x, a.x b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"  // (t.x - a.y b.y, t.y + a.y b.x)
        : "=v"(r)
        : "v"(av), "v"(bv), "v"(t));
    return pkc(r);
}
// (a - c) * (-i) = (a.y - c.y, c.x - a.x)
__device__ __forceinline__ cx<float> pk_sub_mi(cx<float> a, cx<float> c) {
    f32x2 r;
    const f32x2 av = pkv(a), cv = pkv(c);
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(av), "v"(cv));
    return pkc(r);
}
// (a - c) * (+i) = (c.y - a.y, a.x - c.x)
__device__ __forceinline__ cx<float> pk_sub_pi(cx<float> a, cx<float> c) {
    f32x2 r;
    const f32x2 av = pkv(a), cv = pkv(c);
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(av), "v"(cv));
    return pkc(r);
}
// (d.x + d.y, d.y - d.x)
__device__ __forceinline__ f32x2 pk_rot8(f32x2 d) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(d));
    return r;
}
// (d.y - d.x, -d.x - d.y)
__device__ __forceinline__ f32x2 pk_rot24(f32x2 d) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[1,1]" : "=v"(r) : "v"(d));
    return r;
}

// Timeline tracing of a kernel (diagnostic builds only, -DSWF_TRACE=1; tools/k1_trace.py): thread 0 of the first
// kTraceBlocks workgroups stamps the shader clock at fixed points into a device array.
#ifndef SWF_TRACE
#define SWF_TRACE 0
#endif
#if SWF_TRACE
constexpr int kTraceBlocks = 49152, kTracePoints = 12;
extern __device__ unsigned long long swf_trace_buf[kTraceBlocks * kTracePoints];
__device__ __forceinline__ void trace_point(int id) {
    asm volatile("" ::: "memory");
    if (threadIdx.x == 0 && blockIdx.x < (unsigned)kTraceBlocks)
        swf_trace_buf[blockIdx.x * kTracePoints + id] = __builtin_readcyclecounter();
    asm volatile("" ::: "memory");
}
#define SWF_TRACE_POINT(id) trace_point(id)
#else
#define SWF_TRACE_POINT(id) ((void)0)
#endif

// compile-time loop: f(std::integral_constant<int, i>) for i in [I0, I1)
template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        static_for<I0 + 1, I1>(f);
    }
}

constexpr int bitrev(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}

// cos(2 pi k / 64), first quadrant, exact at the ends
constexpr double kCosQ[17] = {1.0,
                              0.9951847266721969,
                              0.9807852804032304,
                              0.9569403357322088,
                              0.9238795325112867,
                              0.881921264348355,
                              0.8314696123025452,
                              0.773010453362737,
                              0.7071067811865476,
                              0.6343932841636455,
                              0.5555702330196023,
                              0.4713967368259978,
                              0.38268343236508984,
                              0.29028467725446233,
                              0.19509032201612833,
                              0.09801714032956077,
                              0.0};
constexpr double cos64(int k) {
    k &= 63;
    if (k <= 16) return kCosQ[k];
    if (k <= 32) return -kCosQ[32 - k];
    if (k <= 48) return -kCosQ[k - 32];
    return kCosQ[64 - k];
}
constexpr double sin64(int k) { return cos64(k - 16); }

// d * exp(-2 pi i NUM / 64) with the trivial cases special-cased
template <typename R, int NUM>
__device__ __forceinline__ cx<R> mul_w64(cx<R> d) {
    constexpr int n = NUM & 63;
    if constexpr (n == 0) {
        return d;
    } else if constexpr (n == 16) {  // -i
        return {d.y, -d.x};
    } else if constexpr (n == 32) {
        return {-d.x, -d.y};
    } else if constexpr (n == 48) {  // +i
        return {-d.y, d.x};
    } else if constexpr (n == 8 && std::is_same<R, float>::value) {
        constexpr float c = 0.70710678118654752f;
        return pkc(pk_rot8(pkv(d)) * f32x2{c, c});
    } else if constexpr (n == 24 && std::is_same<R, float>::value) {
        constexpr float c = 0.70710678118654752f;
        return pkc(pk_rot24(pkv(d)) * f32x2{c, c});
    } else if constexpr (n == 8) {  // (1 - i)/sqrt2
        constexpr R c = (R)0.7071067811865476;
        return {(d.x + d.y) * c, (d.y - d.x) * c};
    } else if constexpr (n == 24) {  // (-1 - i)/sqrt2
        constexpr R c = (R)0.7071067811865476;
        return {(d.y - d.x) * c, -(d.x + d.y) * c};
    } else {
        constexpr R c = (R)cos64(n), s = (R)(-sin64(n));
        if constexpr (std::is_same<R, float>::value) {
            // (d.x, d.x)*(c, s) + (d.y, d.y)*(-s, c) as two packed instructions
            const f32x2 dx = {d.x, d.x}, dy = {d.y, d.y};
            const f32x2 k0 = {c, s}, k1 = {-s, c};
            const f32x2 r = __builtin_elementwise_fma(dy, k1, dx * k0);
            return {r.x, r.y};
        } else {
            return {d.x * c - d.y * s, d.x * s + d.y * c};
        }
    }
}

// (a - c) * exp(-2 pi i NUM / 64): the lower output of a DIF butterfly
template <typename R, int NUM>
__device__ __forceinline__ cx<R> sub_mul_w64(cx<R> a, cx<R> c) {
    if constexpr (std::is_same<R, float>::value && (NUM & 63) == 16) {
        return pk_sub_mi(a, c);
    } else if constexpr (std::is_same<R, float>::value && (NUM & 63) == 48) {
        return pk_sub_pi(a, c);
    } else {
        return mul_w64<R, NUM>(a - c);
    }
}

// In-register radix-2^LOGR DIF FFT over x[OFF + r*STR], r < 2^LOGR.
// Result element k ends up at x[OFF + bitrev(k)*STR].
template <typename R, int LOGR, int STR, int OFF, int PTOT>
__device__ __forceinline__ void fft_reg(cx<R> (&x)[PTOT]) {
    constexpr int RAD = 1 << LOGR;
    static_assert(LOGR <= 6, "radix too large for the constant table");
    static_for<0, LOGR>([&](auto sI) {
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

// LDS exchange buffer layout ---------------------------------------------------
// Row-per-workgroup-row layout: element e of row rb sits at rb*PITCH + e + (e >> LOGPAD): one pad element
// per 2^LOGPAD elements (128 B worth, but never more than the radix so that every phase's scatter and
// gather addresses are  base + compile-time constant  -- see phase_exchange).
constexpr int natural_logpad(int elem_bytes) { return elem_bytes == 4 ? 5 : elem_bytes == 8 ? 4 : 3; }
constexpr int eff_logpad(int elem_bytes, int logp) {
    return natural_logpad(elem_bytes) < logp ? natural_logpad(elem_bytes) : (logp > 0 ? logp : 1);
}

// Geometry of one engine configuration.
template <typename R, int LOGN_, int LOGP_, int NT_, bool SPLIT_>
struct Geo {
    static constexpr int LOGN = LOGN_, LOGP = LOGP_, NT = NT_;
    static constexpr bool SPLIT = SPLIT_;
    static constexpr int N = 1 << LOGN, P = 1 << LOGP, T = N / P;
    static_assert(T >= 1 && NT % T == 0, "bad geometry");
    static constexpr int RB = NT / T;  // rows per workgroup
    static constexpr int ELEM = SPLIT ? (int)sizeof(R) : (int)(2 * sizeof(R));
    static constexpr int LOGPAD = eff_logpad(ELEM, LOGP);
    static constexpr int PITCH = N + (N >> LOGPAD);
    static constexpr size_t LDS_BYTES = (size_t)RB * PITCH * ELEM;
    // the T threads of a row are consecutive threads of ONE wave (row-per-slice layout, rowfast = false): row_sync
    static constexpr bool WAVE_ROWS = T <= 64 && (64 % T) == 0;
};

// Position of (row rb, element e) in the exchange buffer.  rowfast: the RB rows
// of the workgroup are interleaved so that consecutive lanes (consecutive
// rows) hit consecutive banks.
template <class G>
__device__ __forceinline__ int lds_pos(int rb, int e, bool rowfast) {
    return rowfast ? e * G::RB + rb : rb * G::PITCH + e + (e >> G::LOGPAD);
}
// Offset of element e0 + d relative to element e0 when the low bits of e0 that d touches are zero
// (true for every scatter / gather of the Stockham schedule): a compile-time constant.
template <class G>
constexpr int lds_delta(int d, bool rowfast) {
    return rowfast ? d * G::RB : d + (d >> G::LOGPAD);
}

// Apply inter-phase twiddles w^r = exp(-2 pi i * kidx * r / N) to the
// butterfly inputs x[U + r*NB], r = 1..RAD-1.  Powers of two come from the
// table (tw[k] = exp(-2 pi i k / N)), the rest from at most log2(RAD)-1
// complex products, which keeps the error at a few ulp.
// Forms that were measured and dropped (r2, same-box A/B): a two-factor form (every w^r ONE product
// of two exactly rounded table values): end-to-end complex64 error 1.0137e-5 vs 1.0172e-5 -- twiddle rounding is not what
// sets the float32 error -- at +1.5 ms of K1 per pass (registers); every w^r loaded from the table (-330 VALU, +48 loads per
// lane): K1 2.06 -> 3.30 ms per facet, the dependent loads sit in the critical path of every phase.
// geometries may ask for the register-lean twiddle form at every radix (kernels that keep an accumulator in registers
// next to the transform: swiftly_groupfinish.h): `static constexpr bool LEAN_TW = true`
template <class G, class = void>
struct lean_tw_of {
    static constexpr bool value = false;
};
template <class G>
struct lean_tw_of<G, std::enable_if_t<G::LEAN_TW>> {
    static constexpr bool value = true;
};

template <typename R, int LOGR, int NB, int U, int PTOT, int N, bool LEAN = false>
__device__ __forceinline__ void twiddle_inputs(cx<R> (&x)[PTOT], const cx<R>* __restrict__ tw, int kidx,
                                               const cx<R>* pre = nullptr) {
    constexpr int RAD = 1 << LOGR;
    if constexpr (LOGR >= 5 || LEAN) {
        // register-lean form: keep only the LOGR table values alive and build each w^r from the set bits of r (the
        // partial products of different powers are the same asm statements on the same operands, which the compiler
        // merges: 26 products per radix-32 phase in the ISA, i.e. one per composite power -- an explicit tree walk
        // compiles to the same instruction counts, r5)
        cx<R> wp[LOGR];
        static_for<0, LOGR>([&](auto bI) {
            constexpr int b = decltype(bI)::value;
            wp[b] = pre ? pre[b] : tw[(kidx << b) & (N - 1)];
        });
        static_for<1, RAD>([&](auto rI) {
            constexpr int r = decltype(rI)::value;
            constexpr int hb = 31 - __builtin_clz(r);
            cx<R> w = wp[hb];
            static_for<0, hb>([&](auto cI) {
                constexpr int c = decltype(cI)::value;
                if constexpr ((r >> c) & 1) w = cmul(w, wp[c]);
            });
            x[U + r * NB] = cmul(x[U + r * NB], w);
        });
    } else {
        cx<R> w[RAD];
        static_for<0, LOGR>([&](auto bI) {
            constexpr int b = decltype(bI)::value;
            w[1 << b] = pre ? pre[b] : tw[(kidx << b) & (N - 1)];
        });
        static_for<1, RAD>([&](auto rI) {
            constexpr int r = decltype(rI)::value;
            constexpr int hb = 31 - __builtin_clz(r);
            if constexpr (r != (1 << hb)) w[r] = cmul(w[1 << hb], w[r - (1 << hb)]);
            x[U + r * NB] = cmul(x[U + r * NB], w[r]);
        });
    }
}

// COMPACT TWIDDLE SECTIONS (r5).  The table values a phase builds its inter-phase twiddles from are
// tw[(k << (LOGN - LOGNS - LOGR + b)) & (N - 1)], b < LOGR, with k = (virtual thread) mod 2^LOGNS: gathers whose lanes are
// 8 << s bytes apart -- up to one cache line per lane (K1's last phase: 124 lines per wave for 5 loads).  Geometries with
// `static constexpr bool COMPACT_TW = true` take them from a re-ordered COPY of the table instead (host-built from the
// same values, bit-identical): for every phase boundary ns = 1 .. LOGN - 1 a section of 2^ns entries of LOGP slots at
// offset LOGP * (2^ns - 2), slot i of entry k = tw[(k << (LOGN - ns - 1 - i)) & (N - 1)] (0 when the shift would be
// negative) -- independent of the schedule: a phase (LOGNS, LOGR) reads b from slot LOGR - 1 - b, and the lanes of a wave
// read consecutive entries (20 lines for K1's last phase).  Measured (r5, same box, K1 per facet): 1.61 - 1.63 ms with the
// table gathers, 1.53 - 1.55 ms with two 16-byte and one 8-byte load per section, 1.51 - 1.52 ms with five 8-byte loads
// (kept): the vector-memory path charges for the lines a wave instruction touches, not for the instruction count.
template <class G, class = void>
struct compact_tw_of {
    static constexpr bool value = false;
};
template <class G>
struct compact_tw_of<G, std::enable_if_t<G::COMPACT_TW>> {
    static constexpr bool value = true;
};
// a geometry with the compact sections switched on
template <class G>
struct CompactTw : G {
    static constexpr bool COMPACT_TW = true;
};
constexpr int compact_tw_offset(int logp, int ns) { return logp * ((1 << ns) - 2); }
constexpr int compact_tw_entries(int logn, int logp) { return compact_tw_offset(logp, logn); }
// host side: the compact copy of table `tw` (length 2^logn) for geometries with 2^logp points per lane
template <typename R>
inline void build_compact_twiddles(const cx<R>* tw, int logn, int logp, cx<R>* out) {
    for (int ns = 1; ns < logn; ns++)
        for (int k = 0; k < (1 << ns); k++)
            for (int i = 0; i < logp; i++) {
                const int sh = logn - ns - 1 - i;
                out[compact_tw_offset(logp, ns) + k * logp + i] = sh >= 0 ? tw[((long long)k << sh) & ((1 << logn) - 1)] : cx<R>{0, 0};
            }
}
// the LOGR table values of phase (LOGNS, LOGR) for virtual thread j
template <class G, typename R, int LOGNS, int LOGR>
__device__ __forceinline__ void load_compact(cx<R> (&cw)[LOGR], int j, const cx<R>* __restrict__ twc) {
    static_assert(LOGR <= G::LOGP && LOGNS >= 1, "a phase never has a larger radix than the points per lane");
    const cx<R>* p = twc + compact_tw_offset(G::LOGP, LOGNS) + (j & ((1 << LOGNS) - 1)) * G::LOGP;
    static_for<0, LOGR>([&](auto bI) {
        constexpr int b = decltype(bI)::value;
        cw[b] = p[LOGR - 1 - b];
    });
}

// One Stockham phase, compute part: twiddle + in-register DFTs.
template <class G, typename R, int LOGNS, int LOGR>
__device__ __forceinline__ void phase_compute(cx<R> (&x)[G::P], int t, const cx<R>* __restrict__ tw,
                                              const cx<R>* pre = nullptr, const cx<R>* __restrict__ twc = nullptr) {
    constexpr int RAD = 1 << LOGR, NB = G::P / RAD;
    static_for<0, NB>([&](auto uI) {
        constexpr int u = decltype(uI)::value;
        if constexpr (LOGNS > 0) {
            int j = t + u * G::T;
            int k = j & ((1 << LOGNS) - 1);
            // angle = -2 pi k r / (Ns * RAD)  ->  table index k * N/(Ns*RAD) * r
            if constexpr (compact_tw_of<G>::value) {
                // table values from the compact section of this phase (in place unless they were preloaded)
                if (NB == 1 && pre) {
                    twiddle_inputs<R, LOGR, NB, u, G::P, G::N, lean_tw_of<G>::value>(x, tw, 0, pre);
                } else {
                    cx<R> cw[LOGR];
                    load_compact<G, R, LOGNS, LOGR>(cw, j, twc);
                    twiddle_inputs<R, LOGR, NB, u, G::P, G::N, lean_tw_of<G>::value>(x, tw, 0, cw);
                }
            } else {
                twiddle_inputs<R, LOGR, NB, u, G::P, G::N, lean_tw_of<G>::value>(x, tw, k << (G::LOGN - LOGNS - LOGR),
                                                                                 NB == 1 ? pre : nullptr);
            }
        }
        fft_reg<R, LOGR, NB, u, G::P>(x);
    });
}
// Geometries with `static constexpr bool PRELOAD_TW = true` (r4c: RGeoPre = the forward K1): the LOGR table values the next
// phase's inter-phase twiddles are built from are requested BEFORE the exchange that precedes the phase, so that their
// latency (an L2 hit: the table is shared by every workgroup) hides under the exchange instead of following its last barrier
template <class G, class = void>
struct preload_tw_of {
    static constexpr bool value = false;
};
template <class G>
struct preload_tw_of<G, std::enable_if_t<G::PRELOAD_TW>> {
    static constexpr bool value = true;
};

// One Stockham phase, scatter part: f(e, value) for every output element of
// this thread, e = natural-order index within the phase's output array.
template <class G, typename R, int LOGNS, int LOGR, class F>
__device__ __forceinline__ void phase_scatter(const cx<R> (&x)[G::P], int t, F&& f) {
    constexpr int RAD = 1 << LOGR, NB = G::P / RAD;
    static_for<0, NB>([&](auto uI) {
        constexpr int u = decltype(uI)::value;
        int j = t + u * G::T;
        int k = j & ((1 << LOGNS) - 1);
        int e0 = ((j - k) << LOGR) + k;
        static_for<0, RAD>([&](auto rI) {
            constexpr int r = decltype(rI)::value;
            // f(e, value) or f(e, value, slot) with slot = u*RAD + r as a compile-time constant
            if constexpr (std::is_invocable_v<F, int, cx<R>, std::integral_constant<int, 0>>)
                f(e0 + (r << LOGNS), x[u + bitrev(r, LOGR) * NB], std::integral_constant<int, u * RAD + r>{});
            else
                f(e0 + (r << LOGNS), x[u + bitrev(r, LOGR) * NB]);
        });
    });
}

// Exchange through LDS: scatter phase output, then gather x[v] = buf[t + v*T].
// Addresses are one runtime base per butterfly (scatter) / per thread (gather)
// plus compile-time offsets, so each LDS access is a single instruction with
// an immediate offset (the padded index e + (e >> LOGPAD) is affine in the
// radix digit because the digit only fills bits that are zero in the base).
// PAIRJ: the thread's blocks are the ADJACENT virtual threads j = t*NB + u (first phase only: lets the kernel load
// the NB adjacent input points of a lane with one wide access) instead of j = t + u*T.
template <class G, typename R, int LOGNS, int LOGR, bool PAIRJ = false, class T_, class Pick>
__device__ __forceinline__ void exchange_pass(cx<R> (&x)[G::P], int t, int rb, bool rowfast, T_* buf, Pick&& pick,
                                              bool write_x_component) {
    constexpr int RAD = 1 << LOGR, NB = G::P / RAD;
    (void)write_x_component;
    static_for<0, NB>([&](auto uI) {
        constexpr int u = decltype(uI)::value;
        const int j = PAIRJ ? t * NB + u : t + u * G::T;
        const int k = j & ((1 << LOGNS) - 1);
        const int e0 = ((j - k) << LOGR) + k;
        const int p0 = lds_pos<G>(rb, e0, rowfast);
        static_for<0, RAD>([&](auto rI) {
            constexpr int r = decltype(rI)::value;
            const int off = rowfast ? lds_delta<G>(r << LOGNS, true) : lds_delta<G>(r << LOGNS, false);
            buf[p0 + off] = pick(x[u + bitrev(r, LOGR) * NB]);
        });
    });
}

template <class G, typename R, class T_, class Put>
__device__ __forceinline__ void gather_pass(int t, int rb, bool rowfast, const T_* buf, Put&& put) {
    const int p0 = lds_pos<G>(rb, t, rowfast);
    static_for<0, G::P>([&](auto vI) {
        constexpr int v = decltype(vI)::value;
        if constexpr (G::T % (1 << G::LOGPAD) == 0) {
            const int off = rowfast ? lds_delta<G>(v * G::T, true) : lds_delta<G>(v * G::T, false);
            put(vI, buf[p0 + off]);
        } else {
            put(vI, buf[lds_pos<G>(rb, t + v * G::T, rowfast)]);
        }
    });
}

// Synchronisation between the scatter and the gather of an exchange.  A row whose T threads sit inside ONE wave
// (T <= 64, row-per-wave-slice layout) needs no workgroup barrier: the LDS operations of a wave execute in order, so
// an ordering point for the compiler is enough and the waves of the workgroup stop marching in lock-step.
template <class G>
__device__ __forceinline__ void row_sync(bool rowfast) {
    if constexpr (G::WAVE_ROWS) {
        if (rowfast)
            __syncthreads();
        else
            __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

template <class G, typename R, int LOGNS, int LOGR, bool PAIRJ = false>
__device__ __forceinline__ void phase_exchange_impl(cx<R> (&x)[G::P], int t, int rb, bool rowfast, void* lds);
template <class G, typename R, int LOGNS, int LOGR, bool PAIRJ = false>
__device__ __forceinline__ void phase_exchange(cx<R> (&x)[G::P], int t, int rb, bool rowfast, void* lds) {
    SWF_TRACE_POINT(LOGNS > 0 ? 5 : 3);
    phase_exchange_impl<G, R, LOGNS, LOGR, PAIRJ>(x, t, rb, rowfast, lds);
    SWF_TRACE_POINT(LOGNS > 0 ? 6 : 4);
}
template <class G, typename R, int LOGNS, int LOGR, bool PAIRJ>
__device__ __forceinline__ void phase_exchange_impl(cx<R> (&x)[G::P], int t, int rb, bool rowfast, void* lds) {
    if constexpr (!G::SPLIT) {
        cx<R>* buf = reinterpret_cast<cx<R>*>(lds);
        exchange_pass<G, R, LOGNS, LOGR, PAIRJ>(x, t, rb, rowfast, buf, [](cx<R> v) { return v; }, true);
        row_sync<G>(rowfast);
        gather_pass<G, R>(t, rb, rowfast, buf, [&](auto vI, cx<R> val) { x[decltype(vI)::value] = val; });
        row_sync<G>(rowfast);
    } else {
        // re and im separately: halves the LDS footprint
        R* buf = reinterpret_cast<R*>(lds);
        exchange_pass<G, R, LOGNS, LOGR, PAIRJ>(x, t, rb, rowfast, buf, [](cx<R> v) { return v.x; }, true);
        row_sync<G>(rowfast);
        // every old real part is in LDS now, so x[].x can take the new ones
        gather_pass<G, R>(t, rb, rowfast, buf, [&](auto vI, R val) { x[decltype(vI)::value].x = val; });
        row_sync<G>(rowfast);
        exchange_pass<G, R, LOGNS, LOGR, PAIRJ>(x, t, rb, rowfast, buf, [](cx<R> v) { return v.y; }, false);
        row_sync<G>(rowfast);
        gather_pass<G, R>(t, rb, rowfast, buf, [&](auto vI, R val) { x[decltype(vI)::value].y = val; });
        row_sync<G>(rowfast);
    }
}

// Run all phases starting at LOGNS.  On entry x[v] = input[t + v*T] (natural
// order); the last phase hands (natural-order output index, value) to `fin`.
template <class G, typename R, int LOGNS, class F>
__device__ __forceinline__ void fft_phases(cx<R> (&x)[G::P], int t, int rb, bool rowfast, void* lds,
                                           const cx<R>* __restrict__ tw, F&& fin, const cx<R>* pre = nullptr,
                                           const cx<R>* __restrict__ twc = nullptr) {
    constexpr int REM = G::LOGN - LOGNS;
    constexpr int LOGR = REM < G::LOGP ? REM : G::LOGP;
    phase_compute<G, R, LOGNS, LOGR>(x, t, tw, pre, twc);
    if constexpr (LOGNS + LOGR == G::LOGN) {
        phase_scatter<G, R, LOGNS, LOGR>(x, t, fin);
    } else {
        constexpr int NLOGNS = LOGNS + LOGR, NREM = G::LOGN - NLOGNS, NLOGR = NREM < G::LOGP ? NREM : G::LOGP;
        if constexpr (preload_tw_of<G>::value && G::P == (1 << NLOGR) && NLOGR >= 5) {
            cx<R> nxt[NLOGR];
            if constexpr (compact_tw_of<G>::value) {
                load_compact<G, R, NLOGNS, NLOGR>(nxt, t, twc);
            } else {
                const int kidx = (t & ((1 << NLOGNS) - 1)) << (G::LOGN - NLOGNS - NLOGR);
                static_for<0, NLOGR>([&](auto bI) {
                    constexpr int b = decltype(bI)::value;
                    nxt[b] = tw[(kidx << b) & (G::N - 1)];
                });
            }
            phase_exchange<G, R, LOGNS, LOGR>(x, t, rb, rowfast, lds);
            fft_phases<G, R, NLOGNS>(x, t, rb, rowfast, lds, tw, fin, nxt, twc);
        } else {
            phase_exchange<G, R, LOGNS, LOGR>(x, t, rb, rowfast, lds);
            fft_phases<G, R, NLOGNS>(x, t, rb, rowfast, lds, tw, fin, nullptr, twc);
        }
    }
}

// Schedule with the SHORT radix first (2^LOGR1 = N / P^k, two blocks per lane) and adjacent virtual threads per
// lane: on entry x[u + NB*r] = input[(t*NB + u) + r * N / 2^LOGR1]  (u < NB = P / 2^LOGR1).
template <class G, typename R, class F>
__device__ __forceinline__ void fft_phases_pair(cx<R> (&x)[G::P], int t, void* lds, const cx<R>* __restrict__ tw, F&& fin,
                                                const cx<R>* __restrict__ twc = nullptr) {
    constexpr int LOGR1 = G::LOGN % G::LOGP;
    static_assert(LOGR1 > 0 && LOGR1 < G::LOGP, "needs a short first phase");
    phase_compute<G, R, 0, LOGR1>(x, t, tw);
    if constexpr (preload_tw_of<G>::value && G::LOGP >= 5 && G::LOGN - LOGR1 >= G::LOGP) {
        cx<R> nxt[G::LOGP];
        if constexpr (compact_tw_of<G>::value) {
            load_compact<G, R, LOGR1, G::LOGP>(nxt, t, twc);
        } else {
            const int kidx = (t & ((1 << LOGR1) - 1)) << (G::LOGN - LOGR1 - G::LOGP);
            static_for<0, G::LOGP>([&](auto bI) {
                constexpr int b = decltype(bI)::value;
                nxt[b] = tw[(kidx << b) & (G::N - 1)];
            });
        }
        phase_exchange<G, R, 0, LOGR1, true>(x, t, 0, false, lds);
        fft_phases<G, R, LOGR1>(x, t, 0, false, lds, tw, fin, nxt, twc);
    } else {
        phase_exchange<G, R, 0, LOGR1, true>(x, t, 0, false, lds);
        fft_phases<G, R, LOGR1>(x, t, 0, false, lds, tw, fin, nullptr, twc);
    }
}

}  // namespace swf
