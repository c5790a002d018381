// SwiFTly on MI355X: lean column-tile passes for transforms along the STRIDED
// axis of row-major complex64 arrays (K1 prepare_facet axis 0, K9 finish_facet
// axis 0, the axis-0 halves of K4/K5/K6/K7).
//
// A length-N transform along axis 0 is done as two passes of short transforms
// (four-step, N = n1*n2).  In each pass a workgroup owns a tile of 64 adjacent
// columns (512 B contiguous per row) and `n` rows `stride` apart; wave w of the
// workgroup owns rows {w + T*v}.  Because a whole wave works on ONE row at a
// time, every index computation (centred shift, cyclic offset, zero-pad test,
// window, four-step twiddle, output row) is wave-uniform and runs on the
// scalar unit with scalar loads; the vector unit only sees the butterflies,
// one 8-byte global access per point and the LDS exchange.
#pragma once
#include "swiftly_fft.h"

namespace swf {

// streaming (non-temporal) global accesses for the column passes: every byte is touched once per pass
// (same-box A/B on MI355X against cacheable accesses: whole pass 74.0 -> 70.6-72.0 ms, r1; K3-5 11.9 -> 12.8 ms cacheable, r2)
template <bool NT>
__device__ __forceinline__ cx<float> cp_load(const cx<float>* p) {
    if constexpr (NT) {
        const f32x2 v = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p));
        return {v.x, v.y};
    } else {
        return *p;
    }
}
template <bool NT>
__device__ __forceinline__ void cp_store(cx<float>* p, cx<float> v) {
    if constexpr (NT) {
        const f32x2 w = {v.x, v.y};
        __builtin_nontemporal_store(w, reinterpret_cast<f32x2*>(p));
    } else {
        *p = v;
    }
}

struct ColPassArgs {
    const cx<float>* in;
    cx<float>* out;
    unsigned in_pitch, out_pitch;  // elements between consecutive rows; (rows * pitch) < 2^32 (host-checked)
    long long in_bs, out_bs;        // batch strides (elements), blockIdx.z
    // two-level batch on the input side: item z reads in + (z / in_bdiv)*in_bs_hi + (z % in_bdiv)*in_bs (in_bdiv > 0)
    long long in_bs_hi;
    int in_bdiv;
    // same on the output side: item z writes out + (z / out_bdiv)*out_bs_hi + (z % out_bdiv)*out_bs (out_bdiv > 0)
    long long out_bs_hi;
    int out_bdiv;
    // optional column gather on load (fuses extract_from_facet along the contiguous axis, core.py:243-253):
    //   source column = (cz.b_base[b] + ((col + cz.b_rot[b]) mod cg_mod)) mod cg_full
    // and, when cg_band_len > 0, through the parity-split band layout of swiftly_rowpass.h (band_column)
    int cg_mod, cg_full;
    int cg_band_start, cg_band_len, cg_band_half;
    const int* ld_rowmap;  // optional (mapped load): physical input row of logical row idx (must be >= 0 for valid rows)
    // gather-sum load (template GS, backward pass: add_to_facet fused into the load of finish_facet, api_helper.py:142-179):
    // ld_rowmap is a table [2][2^full_logn]; logical row idx is the SUM of up to two source rows  ld_rowmap[idx],
    // ld_rowmap[2^full_logn + idx]  (negative = none), each encoded  chunk << 20 | row  and read at
    // in + cz.c_base[chunk] + facet * cz.c_fs[chunk] + row * in_pitch  (chunks = pieces of a multi-GPU receive buffer)
    int gs;
    // accumulate only: optional per-OUTPUT-column flags (device bytes, indexed by the column written); a column whose
    // flag is 0 has not been written yet and is stored plainly (no read-modify-write, no zero fill needed)
    const unsigned char* touched;
    int ncols;                      // columns (= rows of the primitive)
    // Sub-launches of one logical launch (r4: the chunked, two-stream four-step of col_transform): the batch items of this
    // launch are z0 + blockIdx.z (the four-step scratch pointer is pre-shifted by the host so that item z0 sits at its
    // start), and the column gather sees column col0 + (tile column) -- the launch covers columns [col0, col0 + ncols) of
    // the logical one, with the un-gathered side's pointer pre-shifted by col0
    int z0, col0;
    // raw (scratch) accesses address item z at (z - raw_z0) * {in,out}_bs: the scratch slot of a sub-launch holds only the
    // launch's own items
    int raw_z0;
    int full_logn;                  // log2 of the full (power-of-two) transform length of a decomposed transform
    // Sub-transform j of a length n = Q * 2^full_logn transform behind the radix-Q pass of swiftly_mixed.h (0 = off):
    // the input is the plain scratch of that pass (ld_plain: logical row = plain index, no map), the store map refers to
    // the full length full_n (any even number) with plain output index Q*k + j (st_qmul, st_qadd)
    int full_n, ld_plain, st_qmul, st_qadd;
    // load: raw -> row = o*in_o_rows + i*in_i_rows ; mapped -> plain index i*ld_mul + o through the map
    int raw_ld;
    int in_i_rows, in_o_rows;
    int ld_mul, ld_a, ld_len, ld_c, ld_mod;
    const float* ld_win;
    const float* ld_win2;
    // store: raw -> row = o*out_o_rows + e*out_i_rows ; mapped -> plain index e*st_mul + o through the map
    int raw_st;
    int out_i_rows, out_o_rows;
    int st_mul, st_a, st_len, st_c, st_mod;
    const float* st_win;
    const float* st_win2;
    long long st_win_bs;   // per-batch-item stride of st_win (masks), 0 = shared
    const int* st_rowmap;  // optional: physical output row of logical row idx (negative = not stored)
    long long st_rowmap_bs;  // per-subgrid-index stride of st_rowmap (one row map per batch item b), 0 = shared
    const float* col_win;  // optional real factor per column applied on store (see RowsArgs::row_win)
    const cx<float>* tw;       // exp(-2 pi i k / n), this pass's length
    const cx<float>* tw_full;  // exp(-2 pi i k / 2^full_logn)
    // float64 ARITHMETIC (r4): complex64 loads and stores, everything in between -- windows, butterflies, LDS
    // exchange, four-step twiddle -- in double, with double tables.  For the two transforms whose rounding errors are
    // amplified by BOTH facet windows before anything cancels them (K2 = the yN-point transform along the strided
    // axis, K3 = the m-point transform behind it, and their backward mirrors; tests/accuracy_model.py): end-to-end
    // complex64 error 1.3e-5 -> 3.5e-6 on the probe configuration, storage floor 3.3e-6.
    int f64;
    const cx<double>* twd;
    const cx<double>* twd_full;
    int tw_on_store;
    float scale;
    int conj_ld, conj_st, accumulate;
    // accesses to the four-step scratch (pass A stores, pass B loads) non-temporal (1, default) or cacheable (0:
    // lets the intermediate live in the 256 MiB Infinity Cache when both passes of a slab run back to back)
    int scratch_nt;
    // single 512-point pass on 32-column tiles (512 threads, 64 KiB of LDS: two workgroups per CU, or one next to the
    // workgroups of another kernel) instead of 64-column tiles (1024 threads, 128 KiB: the CU to itself); set by the
    // callers for which it was measured -- K3 of the forward wave loop, which runs next to K2 of the following waves (r5)
    int tile32;
};

// Per-batch-item parameters (by value).  Batch item z = f * nb + b  (f: facet index, b: subgrid index of the
// wave); what varies with the SUBGRID lives in the b_* tables, what varies with the FACET in the f_* tables.
constexpr int kColZB = 64;   // subgrids per launch
constexpr int kColZF = 32;   // facets per launch
constexpr int kColZC = 16;   // source chunks of a gather-sum load
constexpr int kGsRowBits = 20;
// kZColScatter: the column map of kZColGather applied to the OUTPUT column (add_to_facet along the contiguous axis
// fused into the store; cg_band_half = 0 selects a plain band: column d = (scol - cg_band_start) mod cg_full)
enum { kZColGather = 1, kZLoadB = 2, kZLoadAF = 4, kZStoreAF = 8, kZStoreAB = 16, kZOutB = 32, kZColScatter = 64 };
template <int NB_, int NF_, int NC_>
struct ColZT {
    int flags;
    int nb;                            // z = f*nb + b ; nb >= 1
    int b_rot[NB_], b_base[NB_];  // kZColGather: column gather (window) per subgrid
    int b_lda[NB_], b_ldc[NB_];   // kZLoadB: load-map offsets (ld_a, ld_c) per subgrid (row window)
    int b_sta[NB_];                  // kZStoreAB: store-map offset st_a per subgrid
    // kZOutB: output placement per subgrid -- item (f, b) writes at out + b_out_off[b] + f * b_out_fs[b] (elements):
    // lets one launch fill a rank-ordered all-to-all send buffer [dest][facet][subgrid of dest]
    long long b_out_off[NB_], b_out_fs[NB_];
    int f_lda[NF_];                  // kZLoadAF: load-map offset ld_a per facet
    int f_sta[NF_];                  // kZStoreAF: store-map offset st_a per facet
    long long c_base[NC_], c_fs[NC_];  // gather-sum load: element offset / facet stride of source chunk c
};
using ColZ = ColZT<kColZB, kColZF, kColZC>;

// COLS_ = 64: a wave works on ONE row (all row bookkeeping wave-uniform).  COLS_ = 32 (r3, P = 32 only): the tile is 32
// columns wide and a wave works on TWO rows, lanes 0-31 / 32-63 -- half the LDS per point, which lets a 1024-point
// transform run in a single pass (1024 x 32 x 4 B = 128 KiB with the re/im-split exchange) instead of a four-step
// through HBM; the row bookkeeping is per half-wave (two v_readlane + a select instead of one v_readlane).
template <int LOGN_, int LOGP_, bool SPLIT_, int COLS_ = 64, int RSZ_ = 4>
struct CGeo {
    static constexpr int LOGN = LOGN_, LOGP = LOGP_;
    static constexpr bool SPLIT = SPLIT_;
    static constexpr int N = 1 << LOGN, P = 1 << LOGP, T = N / P;  // T thread-rows per workgroup
    static constexpr bool WAVE_ROWS = false;  // a column's points are spread over the T thread-rows
    static constexpr int COLS = COLS_;
    static constexpr bool HALF = COLS_ == 32;
    static_assert(COLS_ == 64 || (COLS_ == 32 && P <= 32 && T % 2 == 0), "32-column tiles: one row slot per lane of a half-wave");
    static constexpr int NT = COLS * T;
    static constexpr int RB = COLS;  // columns per tile
    static constexpr int ELEM = SPLIT ? RSZ_ : 2 * RSZ_;  // RSZ_ = size of the real type of the exchange
    static constexpr bool LEAN_TW = RSZ_ == 8;            // double: inter-phase twiddles in the register-lean form
    static constexpr int PITCH = 0;  // unused (interleaved-rows layout)
    static constexpr int LOGPAD = 4;  // unused
    static constexpr size_t LDS_BYTES = T > 1 ? (size_t)N * RB * ELEM : 0;
    // minimum waves per SIMD the register allocator must leave room for
    static constexpr int MINW = NT >= 256 ? 4 : 1;
};

// the geometry used for a LOGN-point pass (float arithmetic)
template <int LOGN, typename RC = float>
struct CGeoFor {
    static constexpr int LOGP = LOGN < 5 ? LOGN : 5;
    // exchange re and im separately from 128 rows on: 32 KiB (n = 128) / 64 KiB (n = 256) of LDS per
    // workgroup keep 16 waves per CU resident, which is what hides the HBM latency (measured:
    // 8 waves/CU -> 84 % of wave cycles waiting, 2.2 TB/s; 16 waves/CU -> 4.7 TB/s)
    // 1024 points: 32-column tiles (two rows per wave), 128 KiB.  (The same tiles for 512 points -- 64 KiB, two
    // workgroups per CU instead of one -- were measured r3 on the 64k pass: 43.37 vs 43.29 ms, no gain; not kept.)
    using type = CGeo<LOGN, LOGP, (LOGN >= 7), (LOGN >= 10 ? 32 : 64)>;
};
// double arithmetic: 16 points per lane (64 VGPRs of data), re / im exchanged separately; 128 points: 512 threads and
// 64 KiB (two workgroups per CU), 256 points: 1024 threads and 128 KiB, 512 points: 32-column tiles, 1024 threads, 128 KiB
constexpr int kColPassMaxLogF64 = 9;
template <int LOGN>
struct CGeoFor<LOGN, double> {
    static constexpr int LOGP = LOGN < 4 ? LOGN : 4;
    // 32-column tiles for 128 points (256 threads and 32 KiB per workgroup: four workgroups per CU in different phases
    // instead of two; measured r4 on the 64k pass: 757 -> 632 us per wave) and for 512 points (1024 threads); 256 points
    // stay at 64 columns (1024 threads, 128 KiB: 608 us against 632 us with 32 columns)
    using type = CGeo<LOGN, LOGP, true, ((LOGN == 7 || LOGN >= 9) ? 32 : 64), 8>;
    // the gather-sum load (backward pass) exists for 64-column tiles only
    using type_gs = CGeo<LOGN, LOGP, true, (LOGN >= 9 ? 32 : 64), 8>;
};

// value of `val` held by the lane that describes row slot v of THIS lane's half-wave (HALF) / of the wave
template <bool HALF>
__device__ __forceinline__ int slot_bcast(int val, int v, int hw) {
    if constexpr (HALF) {
        const int a = __builtin_amdgcn_readlane(val, v), b = __builtin_amdgcn_readlane(val, 32 + v);
        return hw ? b : a;
    } else {
        return __builtin_amdgcn_readlane(val, v);
    }
}
template <bool HALF>
__device__ __forceinline__ float slot_bcast_f(float val, int v, int hw) {
    return __builtin_bit_cast(float, slot_bcast<HALF>(__builtin_bit_cast(int, val), v, hw));
}

template <bool HALF>
__device__ __forceinline__ double slot_bcast_f(double val, int v, int hw) {
    const long long b = __builtin_bit_cast(long long, val);
    const unsigned lo = (unsigned)slot_bcast<HALF>((int)b, v, hw), hi = (unsigned)slot_bcast<HALF>((int)(b >> 32), v, hw);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// MODE: 0 = pass A (mapped load, four-step twiddle, raw store to scratch)
//       1 = pass B (raw load from scratch, mapped store)
//       2 = single pass (mapped load, mapped store)
//
// Row bookkeeping is done LANE-PARALLEL once per wave: lane v works out
// source row / validity / window of the wave's v-th input row, lane s the
// destination row / window / four-step twiddle of its s-th output row (one
// coalesced vector load per table instead of P dependent scalar loads), and
// the main loops fetch the values with v_readlane.  The output-side
// bookkeeping is issued before the butterflies so its latency hides under them.
template <class G, int MODE, bool SNT, bool GS, class CZ, typename RC = float>
__device__ __forceinline__ void col_pass_body(const ColPassArgs& A, const cx<float>* __restrict__ gin,
                                              cx<float>* __restrict__ gout, const float* __restrict__ ld_win,
                                              const float* __restrict__ ld_win2, const float* __restrict__ st_win,
                                              const float* __restrict__ st_win2, const int* __restrict__ st_rowmap,
                                              const cx<RC>* __restrict__ tw, const cx<RC>* __restrict__ tw_full,
                                              const CZ& cz, const int wave, const int lane, const int bx, const int o,
                                              const int z, unsigned char* smem) {
    constexpr int P = G::P, T = G::T;
    static_assert(P <= 64, "one lane per row slot");
    constexpr bool HALF = G::HALF;
    static_assert(!(HALF && GS), "gather-sum load: 64-column tiles only");
    constexpr bool RAW_LD = MODE == 1, RAW_ST = MODE == 0;
    constexpr bool NT_LD = RAW_LD ? SNT : true, NT_ST = RAW_ST ? SNT : true;
    // last Stockham phase: radix 2^LR at stride 2^LNS  (phases are LOGP, LOGP, ..., remainder)
    constexpr int LR = G::LOGN <= G::LOGP ? G::LOGN : (G::LOGN % G::LOGP == 0 ? G::LOGP : G::LOGN % G::LOGP);
    constexpr int LNS = G::LOGN - LR;
    const int hw = HALF ? lane >> 5 : 0;              // which half-wave (32-column tiles: two rows per wave)
    const int t = HALF ? 2 * wave + hw : wave;        // thread-row id: owns rows t + v*T
    const int clane = HALF ? (lane & 31) : lane;      // column within the tile
    const int col = bx * G::COLS + clane;
    const bool live = col < A.ncols;
    const int FS = 1 << A.full_logn;              // four-step period / power-of-two length
    const int FN = A.full_n > 0 ? A.full_n : FS;  // modulus of the maps
    auto wrapn = [FN](int v) { return v >= FN ? v - FN : v; };
    const int zf = z / cz.nb, zb = z - zf * cz.nb;  // uniform
    int scol = col;
    if (cz.flags & (kZColGather | kZColScatter)) {  // uniform
        // (cg_mod = m is a power of two; cg_full = yN any even length: b_base < cg_full, i < cg_mod <= cg_full)
        const int i = (col + A.col0 + cz.b_rot[zb]) & (A.cg_mod - 1);
        scol = cz.b_base[zb] + i;
        if (scol >= A.cg_full) scol -= A.cg_full;
        if (A.cg_band_len > 0) {
            int d = scol - A.cg_band_start;
            if (d < 0) d += A.cg_full;
            if (d >= A.cg_band_len) d = 0;  // cannot happen for a window of the plan; keeps the access in bounds
            scol = A.cg_band_half > 0 ? (d & 1) * A.cg_band_half + (d >> 1) : d;
        }
    }
    const int lcol = (cz.flags & kZColGather) ? scol : col;    // column read
    const int ocol = (cz.flags & kZColScatter) ? scol : col;   // column written
    // (selects of VALUES, with in-range indices whatever the flags say: a conditional assignment from the table makes
    // the compiler select between POINTERS into `cz` and a local, which keeps a private copy of the whole table alive)
    constexpr int NBZ = sizeof(cz.b_lda) / sizeof(int), NFZ = sizeof(cz.f_lda) / sizeof(int);
    const int zbi = zb < NBZ ? zb : 0, zfi = zf < NFZ ? zf : 0;
    const int t_lda = cz.b_lda[zbi], t_ldc = cz.b_ldc[zbi], t_flda = cz.f_lda[zfi], t_fsta = cz.f_sta[zfi], t_bsta = cz.b_sta[zbi];
    int ld_a = A.ld_a, ld_c = A.ld_c, st_a = A.st_a;
    ld_a = (cz.flags & kZLoadB) ? t_lda : ld_a;
    ld_c = (cz.flags & kZLoadB) ? t_ldc : ld_c;
    ld_a = (cz.flags & kZLoadAF) ? t_flda : ld_a;
    st_a = (cz.flags & kZStoreAF) ? t_fsta : st_a;
    st_a = (cz.flags & kZStoreAB) ? t_bsta : st_a;
    const long long in_off =
        A.in_bdiv > 0 ? (long long)(z / A.in_bdiv) * A.in_bs_hi + (long long)(z % A.in_bdiv) * A.in_bs
                      : (long long)(RAW_LD ? z - A.raw_z0 : z) * A.in_bs;
    const cx<float>* __restrict__ in = gin + (GS ? 0ll : in_off) + lcol;
    const long long out_off =
        (cz.flags & kZOutB) ? cz.b_out_off[zb] + (long long)zf * cz.b_out_fs[zb]
        : A.out_bdiv > 0 ? (long long)(z / A.out_bdiv) * A.out_bs_hi + (long long)(z % A.out_bdiv) * A.out_bs
                         : (long long)(RAW_ST ? z - A.raw_z0 : z) * A.out_bs;
    cx<float>* __restrict__ out = gout + out_off + ocol;
    const RC sg_ld = A.conj_ld ? (RC)-1 : (RC)1;
    const RC sg_st = A.conj_st ? (RC)-1 : (RC)1;
    const RC col_w = (A.col_win && live) ? (RC)A.col_win[col] : (RC)1;
    bool rmw = A.accumulate != 0;
    if (!RAW_ST && A.accumulate && A.touched && live) rmw = A.touched[ocol] != 0;
    const int slot = lane & (P - 1);

    // ---- input rows: lane `slot` describes row i = t + slot*T
    int in_row;         // element offset row*pitch is formed later; -1 = zero (padding)
    long long gs_off1 = -1, gs_off2 = -1;  // GS: element offsets of the (up to) two source rows, -1 = none
    RC in_w = (RC)1;
    {
        const int i = t + slot * T;
        if constexpr (RAW_LD) {
            in_row = o * A.in_o_rows + i * A.in_i_rows;
        } else {
            const int pi = i * A.ld_mul + o;
            const int ci = wrapn(pi + (FN >> 1));
            const int q = A.ld_plain ? pi : wrapn(ci + ld_a);
            int idx = q + (A.ld_plain ? 0 : ld_c);
            if (!A.ld_plain && idx >= A.ld_mod) idx -= A.ld_mod;
            const bool ok = A.ld_plain || q < A.ld_len;
            if constexpr (GS) {
                // (resolved here, outside the load loop's lambda: a by-reference capture of `cz` with a dynamic
                // index would make the compiler copy the whole table to scratch)
                constexpr int RM = (1 << kGsRowBits) - 1;
                const int r1 = ok ? A.ld_rowmap[idx] : -1;
                const int r2 = ok ? A.ld_rowmap[FN + idx] : -1;
                if (r1 >= 0) {
                    const int c = r1 >> kGsRowBits;
                    gs_off1 = cz.c_base[c] + (long long)zf * cz.c_fs[c] + (long long)(r1 & RM) * A.in_pitch;
                }
                if (r2 >= 0) {
                    const int c = r2 >> kGsRowBits;
                    gs_off2 = cz.c_base[c] + (long long)zf * cz.c_fs[c] + (long long)(r2 & RM) * A.in_pitch;
                }
                in_row = r1;
            } else {
                if (A.ld_rowmap) idx = A.ld_rowmap[ok ? idx : 0];
                in_row = ok ? idx : -1;
            }
            const int qs = ok ? q : 0;
            if (ld_win) in_w *= (RC)ld_win[qs];
            if (ld_win2) in_w *= (RC)ld_win2[qs];
        }
    }
    // ---- output rows: lane `slot` describes the output the scatter calls slot (u, r)
    int out_row;  // -1 = not stored
    RC out_w = (RC)A.scale;
    cx<RC> out_tw = {(RC)1, (RC)0};
    {
        const int u = slot >> LR, r = slot & ((1 << LR) - 1);
        const int j = t + u * T;
        const int k = j & ((1 << LNS) - 1);
        const int e = ((j - k) << LR) + k + (r << LNS);
        if constexpr (RAW_ST) {
            out_row = o * A.out_o_rows + e * A.out_i_rows;
            out_tw = tw_full[((unsigned)e * (unsigned)o) & (unsigned)(FS - 1)];
        } else {
            int pk = e * A.st_mul + o;
            if (A.st_qmul > 0) pk = pk * A.st_qmul + A.st_qadd;
            const int ck = wrapn(pk + (FN >> 1));
            const int d = wrapn(ck + st_a);
            int idx = d + A.st_c;
            if (idx >= A.st_mod) idx -= A.st_mod;
            const bool ok = d < A.st_len;
            const int ds = ok ? d : 0;
            if (st_win) out_w *= (RC)st_win[(long long)z * A.st_win_bs + ds];
            if (st_win2) out_w *= (RC)st_win2[ds];
            int row = idx;
            if (st_rowmap) row = st_rowmap[(long long)zb * A.st_rowmap_bs + (ok ? idx : 0)];
            out_row = ok ? row : -1;
        }
    }

    // Issue ALL loads first (nothing in this loop consumes a loaded value, so the
    // P loads of a lane are in flight together), then apply windows / conjugation.
    cx<RC> x[P];
    static_for<0, P>([&](auto vI) {
        constexpr int v = decltype(vI)::value;
        const int row = slot_bcast<HALF>(in_row, v, hw);
        cx<float> val = {0.f, 0.f};
        if constexpr (GS) {
            const int lo1 = __builtin_amdgcn_readlane((int)gs_off1, v), hi1 = __builtin_amdgcn_readlane((int)(gs_off1 >> 32), v);
            if (hi1 >= 0) {  // uniform
                const long long o1 = ((long long)hi1 << 32) | (unsigned)lo1;
                if (live) val = cp_load<NT_LD>(in + o1);
            }
            const int lo2 = __builtin_amdgcn_readlane((int)gs_off2, v), hi2 = __builtin_amdgcn_readlane((int)(gs_off2 >> 32), v);
            if (hi2 >= 0) {  // uniform
                const long long o2 = ((long long)hi2 << 32) | (unsigned)lo2;
                if (live) {
                    const cx<float> w2 = cp_load<NT_LD>(in + o2);
                    if constexpr (sizeof(RC) == 8) {  // the sum of the two source rows in double
                        x[v] = cx<RC>{(RC)val.x + (RC)w2.x, (RC)val.y + (RC)w2.y};
                        return;
                    }
                    val.x += w2.x;
                    val.y += w2.y;
                }
            }
        } else {
            if (row >= 0) {  // uniform (per half-wave with 32-column tiles)
                if (live) val = cp_load<NT_LD>(in + (unsigned)row * A.in_pitch);
            }
        }
        x[v] = cx<RC>{(RC)val.x, (RC)val.y};
    });
    static_for<0, P>([&](auto vI) {
        constexpr int v = decltype(vI)::value;
        if constexpr (!RAW_LD) {
            const RC w = slot_bcast_f<HALF>(in_w, v, hw);
            x[v].x *= w;
            x[v].y *= w * sg_ld;
        } else {
            x[v].y *= sg_ld;
        }
    });

    fft_phases<G, RC, 0>(x, t, clane, true, smem, tw, [&](int, cx<RC> v, auto sI) {
        constexpr int s = decltype(sI)::value;
        const int row = slot_bcast<HALF>(out_row, s, hw);
        if (row < 0) return;  // uniform (per half-wave with 32-column tiles)
        if constexpr (RAW_ST) {
            cx<RC> w;
            w.x = slot_bcast_f<HALF>(out_tw.x, s, hw);
            w.y = slot_bcast_f<HALF>(out_tw.y, s, hw);
            v = cmul(v, w);
            v.y *= sg_st;
            if (live) cp_store<NT_ST>(out + (unsigned)row * A.out_pitch, cx<float>{(float)v.x, (float)v.y});
        } else {
            const RC w = slot_bcast_f<HALF>(out_w, s, hw) * col_w;
            v.x *= w;
            v.y *= w * sg_st;
            cx<float>* p = out + (unsigned)row * A.out_pitch;
            if (A.accumulate) {
                if (live && rmw) {
                    const cx<float> old = *p;
                    v.x += (RC)old.x;
                    v.y += (RC)old.y;
                }
            }
            if (live) cp_store<NT_ST>(p, cx<float>{(float)v.x, (float)v.y});
        }
    });
}

// (wave = workgroup-uniform wave index, lane; bx / o / z = column tile, outer index, batch item: the grid of the plain
// kernel; tools/experiments/swiftly_fourstep.h calls the body with a schedule of its own)
template <class G, int MODE, bool SNT, bool GS = false, typename RC = float>
__global__ __launch_bounds__(G::NT, G::MINW) void col_pass_kernel(const ColPassArgs A, const cx<float>* __restrict__ gin,
                                                         cx<float>* __restrict__ gout,
                                                         const float* __restrict__ ld_win,
                                                         const float* __restrict__ ld_win2,
                                                         const float* __restrict__ st_win,
                                                         const float* __restrict__ st_win2,
                                                         const int* __restrict__ st_rowmap,
                                                         const cx<RC>* __restrict__ tw,
                                                         const cx<RC>* __restrict__ tw_full, const ColZ cz) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    col_pass_body<G, MODE, SNT, GS, ColZ, RC>(A, gin, gout, ld_win, ld_win2, st_win, st_win2, st_rowmap, tw, tw_full, cz,
                                    __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63, blockIdx.x,
                                    blockIdx.y, blockIdx.z + A.z0, smem);
}

constexpr int kColPassMinLog = 2;
constexpr int kColPassMaxLog = 10;  // 1024 points: 32-column tiles

int launch_col_pass(int logn, int mode, const ColPassArgs& a, const ColZ& cz, int outer, int nbatch, hipStream_t s);
int init_col_pass();
bool col_pass_f64_supported(int logn);  // float64-arithmetic instances (ColPassArgs::f64)

}  // namespace swf
