// SwiFTly on MI355X: "sum over the facets of one off1 GROUP + finish" along the STRIDED axis of a subgrid (complex64).
//
// The subgrid side of the forward pass (api_helper.sum_and_finish_subgrid, api_helper.py:73-112) is separable, so the
// axis-0 half can be finished completely before the axis-1 half starts:
//
//     V_g[a, j] = mask0[a] * cifft_xM( sum_{f in g} place0_f( Fn * cfft_m( C_f[:, j] ) ) )[(xM/2 - xA//2 + a + off0) mod xM]
//     S[a, c]   = mask1[c] * cifft_xM( sum_g place1_g( Fn * cfft_m( V_g[a, :] ) ) )[(xM/2 - xA//2 + c + off1) mod xM]
//
// with g = the facets that share an off1 (a column of the facet grid), C_f the [m, m] contribution of facet f
// (extract_from_facet along both axes, core.py:243-253), place0_f / place1_g the placement of add_to_subgrid
// (core.py:274-285) and the crop of finish_subgrid (core.py:316-323).  This kernel is the first line: per 32-column
// tile of the contribution it gathers the window rows of every facet of the group straight from the wave's facet
// buffers Q_f (the contribution is never materialised), transforms them (m points, strided axis), weights with Fn,
// places them into the padded column and sums over the facets IN REGISTERS, then runs the xM-point inverse transform,
// crops to xA rows and applies mask0.  The second line is sum_finish_facets_kernel with one "facet" per group
// (swiftly_sumfinish.h, direct-row mode).
//
// Against the r2 dataflow (transform_contributions -> G[F][S][m][m] -> sum_finish_facets -> tmp[S][xM][xA] ->
// finish_subgrid axis 0) the intermediate shrinks from F*m*m + xM*xA to n_groups*xA*m complex values per subgrid
// (9*512*512 + 1024*928 -> 3*928*512 for the 3x3 cover: 26.5 -> 11.4 MB) and is written once and read once:
// 78.8 -> 48.6 MB per subgrid.
//
// Geometry: 16-column tiles (128-byte row segments), 64 thread-rows: thread (tr, c) owns rows tr + 64*v of column c
// for BOTH transforms (m/64 and xM/64 points per thread), 1024 threads.  The accumulator of a thread is the register
// image of the xM-point transform's input, so the facet sum needs no LDS accumulator: the weighted m-point outputs go
// through a compact plane [m][16] (complex) and every thread picks up the rows it owns.  (A first version with
// 32-column tiles held 32 accumulator points per thread next to the m-point transform: 236-260 bytes of spills per lane
// in every arrangement tried, 576 us per wave of the 64k workload; r3.  This form: 128 VGPRs, no spills, 240 us per wave
// = 2.5 TB/s -- one 1024-thread workgroup per CU whose phases run one after the other; together with the direct-row
// sum_finish (125 us) it equals the 385 us of the three kernels it replaces.)
#pragma once
#include "swiftly_colpass.h"

namespace swf {

constexpr int kGroupFinishMaxFacets = 64;
constexpr int kGroupFinishMaxBatch = 64;

struct GroupFinishArgs {
    const cx<float>* in;   // Q[f][row][m]: the wave's facet buffers (prepare_facet_columns)
    cx<float>* out;        // V[g][b][a][m]
    long long in_fs;       // facet stride of Q (elements); row stride = in_pitch
    unsigned in_pitch;
    long long out_gs, out_bs;  // group / subgrid strides of V (elements); row stride = m
    const int* rowmap;     // optional: physical row of Q for padded row idx (negative = absent: zeros)
    int yN, xA;
    int ncols;             // m
    int ngroups;
    int gstart[kGroupFinishMaxFacets + 1];  // facets of group g: fidx[gstart[g] .. gstart[g+1])
    int fidx[kGroupFinishMaxFacets];        // facet index into Q
    int sp0[kGroupFinishMaxFacets];         // s'0 = floor(facet_off0 * xM / N) of that facet
    int lda[kGroupFinishMaxBatch], ldc[kGroupFinishMaxBatch];  // row window of subgrid b: (-s) mod m, (yN/2 - m/2 + s) mod yN
    int st_a[kGroupFinishMaxBatch];                            // (-(xM/2 - xA//2 + off0_b)) mod xM
    const float* fn;       // Fn[m]
    const float* mask;     // optional mask0 [nbatch][xA]
    long long mask_bs;
    const cx<float>* tw_m;
    const cx<float>* tw_x;
};

template <int LOGM, int LOGX>
struct GFGeo {
    static_assert(LOGM >= 7 && LOGM < LOGX && LOGX <= 12, "m = 128 .. , m < xM <= 4096");
    static constexpr int LOGT = 6, T = 64;  // thread-rows
    template <int LOGN_>
    struct G16 {
        static constexpr int LOGN = LOGN_, LOGP = LOGN_ - LOGT;
        // interleaved (re,im) exchange: half the LDS passes and workgroup barriers of the split form (the workgroup is
        // alone on its CU either way: 1024 threads)
        static constexpr bool SPLIT = false;
        static constexpr int N = 1 << LOGN, P = 1 << LOGP, T = N / P;
        static constexpr bool WAVE_ROWS = false;
        static constexpr int COLS = 16, RB = 16, NT = COLS * T, ELEM = 8, PITCH = 0, LOGPAD = 4;
        static constexpr bool LEAN_TW = true;  // an accumulator stays live next to the transform
        static constexpr size_t LDS_BYTES = (size_t)N * RB * ELEM;
    };
    using GM = G16<LOGM>;
    using GX = G16<LOGX>;
    static constexpr int NT = 16 * T;
    // loop phase: plane [m][16] complex = the m-point exchange buffer (the last phase of a transform has no exchange and
    // follows a barrier); final phase: exchange buffer [xM][16] complex
    static constexpr size_t LDS_BYTES = GX::LDS_BYTES;
};

template <int LOGM, int LOGX>
__global__ __launch_bounds__((GFGeo<LOGM, LOGX>::NT), 4) void group_finish_kernel(const GroupFinishArgs A) {
    using S = GFGeo<LOGM, LOGX>;
    using GM = typename S::GM;
    using GX = typename S::GX;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int M = 1 << LOGM, X = 1 << LOGX, T = S::T, PM = GM::P, PX = GX::P;
    const int tr = threadIdx.x >> 4, c16 = threadIdx.x & 15;
    const int col = blockIdx.x * 16 + c16;
    const bool live = col < A.ncols;
    const int b = blockIdx.y;  // subgrid of the wave
    const int g = blockIdx.z;  // off1 group
    void* exch = smem;
    cx<float>* plane = reinterpret_cast<cx<float>*>(smem);

    // input rows of the m-point transform (the same for every facet: the window only depends on the subgrid)
    int in_row[PM];
    static_for<0, PM>([&](auto vI) {
        constexpr int v = decltype(vI)::value;
        const int i = tr + v * T;
        const int ci = i ^ (M >> 1);
        const int q = (ci + A.lda[b]) & (M - 1);
        int idx = q + A.ldc[b];
        if (idx >= A.yN) idx -= A.yN;
        in_row[v] = A.rowmap ? A.rowmap[idx] : idx;
    });

    cx<float> acc[PX];
    static_for<0, PX>([&](auto vI) { acc[decltype(vI)::value] = cx<float>{0.f, 0.f}; });

    // the rows of facet n+1 are requested before facet n is transformed (the workgroup is alone on its CU: nothing else
    // hides the HBM latency)
    auto load_rows = [&](cx<float> (&dst)[PM], int n) {
        const cx<float>* __restrict__ in = A.in + (long long)A.fidx[n] * A.in_fs + col;
        static_for<0, PM>([&](auto vI) {
            constexpr int v = decltype(vI)::value;
            cx<float> val = {0.f, 0.f};
            if (in_row[v] >= 0 && live) val = cp_load<true>(in + (unsigned)in_row[v] * A.in_pitch);
            dst[v] = val;
        });
    };
    const int n_end = A.gstart[g + 1];
    cx<float> x[PM], xn[PM];
    if (A.gstart[g] < n_end) load_rows(x, A.gstart[g]);
    for (int n = A.gstart[g]; n < n_end; n++) {  // uniform
        const int sp = A.sp0[n];
        if (n + 1 < n_end) load_rows(xn, n + 1);
        // weighted outputs go straight into the plane, indexed by the row d of G = Fn[d] * F[(d + s') mod m]
        fft_phases<GM, float, 0>(x, tr, c16, true, exch, A.tw_m, [&](int e, cx<float> v) {
            const int ck = e ^ (M >> 1);
            const int d = (ck - sp) & (M - 1);
            const float w = A.fn[d];
            plane[d * 16 + c16] = cx<float>{v.x * w, v.y * w};
        });
        __syncthreads();
        // pick up the rows this thread owns: plain row p = tr + T*v of the xM-point input is centred row rho = p ^ xM/2;
        // the facet's band covers it iff k = (rho - base) mod xM < m, and then it is row d = k of G.  Branch-free (an
        // uncovered row reads a valid address and gets weight 0), in chunks so that the loaded values do not pile up
        // next to the accumulator.
        const int base = ((X >> 1) - (M >> 1) + sp) & (X - 1);
        static_for<0, PX>([&](auto vI) {
            constexpr int v = decltype(vI)::value;
            const int p = tr + T * v;
            const int k = ((p ^ (X >> 1)) - base) & (X - 1);
            const cx<float> val = plane[(k & (M - 1)) * 16 + c16];
            const float w = k < M ? 1.f : 0.f;
            acc[v].x += val.x * w;
            acc[v].y += val.y * w;
            if constexpr (v % 8 == 7) __builtin_amdgcn_sched_barrier(0);
        });
        __syncthreads();  // the plane is the exchange buffer of the next transform
        static_for<0, PM>([&](auto vI) { x[decltype(vI)::value] = xn[decltype(vI)::value]; });
    }

    // ---- xM-point inverse transform (conj . FFT . conj), crop, mask, store
    static_for<0, PX>([&](auto vI) { acc[decltype(vI)::value].y = -acc[decltype(vI)::value].y; });
    cx<float>* __restrict__ out = A.out + (long long)g * A.out_gs + (long long)b * A.out_bs + col;
    const float* __restrict__ mask = A.mask ? A.mask + (long long)b * A.mask_bs : nullptr;
    const int st_a = A.st_a[b];
    const float scale = 1.f / (float)X;
    fft_phases<GX, float, 0>(acc, tr, c16, true, exch, A.tw_x, [&](int e, cx<float> v) {
        const int ck = e ^ (X >> 1);
        const int d = (ck + st_a) & (X - 1);
        if (d < A.xA && live) {
            float w = scale;
            if (mask) w *= mask[d];
            cp_store<true>(out + (unsigned)d * (unsigned)A.ncols, cx<float>{v.x * w, -v.y * w});
        }
    });
}

int launch_group_finish(int logm, int logx, const GroupFinishArgs& a, int nbatch, hipStream_t s);
int init_group_finish();
bool group_finish_supported(int logm, int logx);

}  // namespace swf
