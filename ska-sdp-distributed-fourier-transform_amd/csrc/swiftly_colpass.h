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

struct ColPassArgs {
    const cx<float>* in;
    cx<float>* out;
    unsigned in_pitch, out_pitch;  // elements between consecutive rows; (rows * pitch) < 2^32 (host-checked)
    long long in_bs, out_bs;        // batch strides (elements), blockIdx.z
    int ncols;                      // columns (= rows of the primitive)
    int full_logn;                  // log2 of the full transform length the maps refer to
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
    const cx<float>* tw;       // exp(-2 pi i k / n), this pass's length
    const cx<float>* tw_full;  // exp(-2 pi i k / 2^full_logn)
    int tw_on_store;
    float scale;
    int conj_ld, conj_st, accumulate;
};

template <int LOGN_, int LOGP_, bool SPLIT_>
struct CGeo {
    static constexpr int LOGN = LOGN_, LOGP = LOGP_;
    static constexpr bool SPLIT = SPLIT_;
    static constexpr int N = 1 << LOGN, P = 1 << LOGP, T = N / P;  // T waves per workgroup
    static constexpr int NT = 64 * T;
    static constexpr int RB = 64;  // columns per tile == lanes
    static constexpr int ELEM = SPLIT ? 4 : 8;
    static constexpr int PITCH = 0;  // unused (interleaved-rows layout)
    static constexpr size_t LDS_BYTES = T > 1 ? (size_t)N * RB * ELEM : 0;
};

// MODE: 0 = pass A (mapped load, four-step twiddle, raw store to scratch)
//       1 = pass B (raw load from scratch, mapped store)
//       2 = single pass (mapped load, mapped store)
// The read-only tables are separate __restrict__ parameters so that their
// wave-uniform loads become scalar loads.
template <class G, int MODE>
__global__ __launch_bounds__(G::NT) void col_pass_kernel(const ColPassArgs A, const cx<float>* __restrict__ gin,
                                                         cx<float>* __restrict__ gout,
                                                         const float* __restrict__ ld_win,
                                                         const float* __restrict__ ld_win2,
                                                         const float* __restrict__ st_win,
                                                         const float* __restrict__ st_win2,
                                                         const int* __restrict__ st_rowmap,
                                                         const cx<float>* __restrict__ tw,
                                                         const cx<float>* __restrict__ tw_full) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int P = G::P, T = G::T;
    constexpr bool RAW_LD = MODE == 1, RAW_ST = MODE == 0;
    // Views of the read-only tables in the constant address space: their loads
    // are invariant, so wave-uniform indices turn into scalar (SMEM) loads.
    typedef const float __attribute__((address_space(4))) * cfp;
    typedef const int __attribute__((address_space(4))) * cip;
    typedef const cx<float> __attribute__((address_space(4))) * ccp;
    const cfp c_ld_win = (cfp)(uintptr_t)ld_win, c_ld_win2 = (cfp)(uintptr_t)ld_win2;
    const cfp c_st_win = (cfp)(uintptr_t)(st_win ? st_win + (long long)blockIdx.z * A.st_win_bs : st_win);
    const cfp c_st_win2 = (cfp)(uintptr_t)st_win2;
    const cip c_rowmap = (cip)(uintptr_t)st_rowmap;
    const ccp c_twf = (ccp)(uintptr_t)tw_full;
    const int lane = threadIdx.x & 63;
    const int t = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave id: uniform
    const int o = blockIdx.y;
    const int col = blockIdx.x * 64 + lane;
    const bool live = col < A.ncols;
    const int FN = 1 << A.full_logn;
    const cx<float>* __restrict__ in = gin + (long long)blockIdx.z * A.in_bs + col;
    cx<float>* __restrict__ out = gout + (long long)blockIdx.z * A.out_bs + col;
    const float sg_ld = A.conj_ld ? -1.f : 1.f;
    const float sg_st = A.conj_st ? -1.f : 1.f;

    cx<float> x[P];
    static_for<0, P>([&](auto vI) {
        constexpr int v = decltype(vI)::value;
        const int i = t + v * T;  // uniform
        cx<float> val = {0.f, 0.f};
        if constexpr (RAW_LD) {
            const unsigned row = (unsigned)(o * A.in_o_rows + i * A.in_i_rows);
            if (live) val = in[row * A.in_pitch];
        } else {
            const int pi = i * A.ld_mul + o;
            const int ci = (pi + (FN >> 1)) & (FN - 1);
            const int q = (ci + A.ld_a) & (FN - 1);
            if (q < A.ld_len) {  // uniform branch
                int idx = q + A.ld_c;
                if (idx >= A.ld_mod) idx -= A.ld_mod;
                if (live) val = in[(unsigned)idx * A.in_pitch];
                float w = 1.f;
                if (ld_win) w *= c_ld_win[q];
                if (ld_win2) w *= c_ld_win2[q];
                val.x *= w;
                val.y *= w * sg_ld;
            }
        }
        if constexpr (RAW_LD) val.y *= sg_ld;
        x[v] = val;
    });

    fft_phases<G, float, 0>(x, t, lane, true, smem, tw, [&](int e_, cx<float> v) {
        // e_ only depends on the wave id: pin it to an SGPR so that all row bookkeeping stays scalar
        const int e = __builtin_amdgcn_readfirstlane(e_);
        if constexpr (RAW_ST) {
            {
                const unsigned ti = ((unsigned)e * (unsigned)o) & (unsigned)(FN - 1);
                const cx<float> wv = {c_twf[ti].x, c_twf[ti].y};
                v = cmul(v, wv);
            }
            v.y *= sg_st;
            const unsigned row = (unsigned)(o * A.out_o_rows + e * A.out_i_rows);
            if (live) out[row * A.out_pitch] = v;
        } else {
            const int pk = e * A.st_mul + o;
            const int ck = (pk + (FN >> 1)) & (FN - 1);
            const int d = (ck + A.st_a) & (FN - 1);
            if (d < A.st_len) {  // uniform
                int idx = d + A.st_c;
                if (idx >= A.st_mod) idx -= A.st_mod;
                float w = A.scale;
                if (st_win) w *= c_st_win[d];
                if (st_win2) w *= c_st_win2[d];
                int row = idx;
                if (st_rowmap) row = c_rowmap[idx];
                if (row >= 0) {  // uniform
                    v.x *= w;
                    v.y *= w * sg_st;
                    cx<float>* p = out + (unsigned)row * A.out_pitch;
                    if (A.accumulate) {
                        if (live) {
                            const cx<float> old = *p;
                            v.x += old.x;
                            v.y += old.y;
                        }
                    }
                    if (live) *p = v;
                }
            }
        }
    });
}

constexpr int kColPassMinLog = 2;
constexpr int kColPassMaxLog = 8;

int launch_col_pass(int logn, int mode, const ColPassArgs& a, int outer, int nbatch, hipStream_t s);
int init_col_pass();

}  // namespace swf
