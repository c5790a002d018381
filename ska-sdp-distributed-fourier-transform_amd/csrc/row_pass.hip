// instantiations + dispatch of the lean long-row kernel (complex64)
#include <cstdlib>

#include "swiftly_rowpass.h"

namespace swf {

#if SWF_TRACE
__device__ unsigned long long swf_trace_buf[kTraceBlocks * kTracePoints];
extern "C" int swiftly_hip_trace_fetch(void* host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(swf_trace_buf), bytes);
}
#endif

template <int LOGN>
struct RGeoFor {
    // 1024 threads, 8 / 16 / 32 points per thread; N = 32768 needs the split re/im exchange
    using type = RGeo<LOGN, LOGN - 10, (LOGN >= 14)>;  // split from 16384: 64 KB LDS -> 2 workgroups per CU
};

template <int LOGN, int MODE>
static int launch_mode(const RowPassArgs& a, hipStream_t s) {
    using G = typename RGeoFor<LOGN>::type;
    hipLaunchKernelGGL((row_pass_kernel<G, MODE>), dim3((unsigned)a.nrows), dim3(G::NT), G::LDS_BYTES, s, a, a.in,
                       a.out, a.ld_win, a.st_win, a.st_win2, a.tw);
    return (int)hipGetLastError();
}
template <int LOGN>
static int launch_one(int mode, const RowPassArgs& a, hipStream_t s) {
    if (mode == 0) return launch_mode<LOGN, 0>(a, s);
    if (mode == 1) return launch_mode<LOGN, 1>(a, s);
    return launch_mode<LOGN, 2>(a, s);
}
template <int LOGN, int MODE>
static int init_mode() {
    using G = typename RGeoFor<LOGN>::type;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&row_pass_kernel<G, MODE>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
}
template <int LOGN>
static int init_one() {
    int rc = init_mode<LOGN, 0>();
    if (!rc) rc = init_mode<LOGN, 1>();
    if (!rc) rc = init_mode<LOGN, 2>();
    return rc;
}

int launch_row_pass(int logn, int mode, const RowPassArgs& a, hipStream_t s) {
    if (a.nrows <= 0) return 0;
    switch (logn) {
        case 13: return launch_one<13>(mode, a, s);
        case 14: return launch_one<14>(mode, a, s);
        case 15: return launch_one<15>(mode, a, s);
        default: return -1;
    }
}
template <class G, int LOGS>
static int launch_split(const RowPassArgs& a, const cx<float>* tw_part, const cx<float>* tw_full, hipStream_t s) {
    const unsigned blocks = (unsigned)(((a.nrows + 7) / 8) * (8 << LOGS));
    if (a.ld_win)
        hipLaunchKernelGGL((row_pass_split_kernel<G, LOGS, true>), dim3(blocks), dim3(G::NT), G::LDS_BYTES, s, a, a.in,
                           a.out, a.ld_win, tw_part, tw_full);
    else
        hipLaunchKernelGGL((row_pass_split_kernel<G, LOGS, false>), dim3(blocks), dim3(G::NT), G::LDS_BYTES, s, a,
                           a.in, a.out, a.ld_win, tw_part, tw_full);
    return (int)hipGetLastError();
}
using SplitGeo2 = RGeo<14, 4, true>;   // 2 x 16384 points, 1024 threads x 16 (one workgroup per CU)
// (4 x 8192 points in 512-thread workgroups and the interleaved-exchange forms were measured slower in r1-r2)
int launch_row_pass_split(const RowPassArgs& a, const cx<float>* tw14, const cx<float>* tw13, const cx<float>* tw_full,
                          hipStream_t s) {
    (void)tw13;
    if (a.nrows <= 0) return 0;
    return launch_split<SplitGeo2, 1>(a, tw14, tw_full, s);
}
using BandGeo5 = RGeo<14, 5, true>;  // 2 x 16384 points, 512 threads x 32, 66 KB LDS: two workgroups per CU
using BandGeo4 = RGeo<14, 4, true>;  // 2 x 16384 points, 1024 threads x 16, one workgroup per CU
// Cyclic run of input segments (seglen consecutive points of the n-point transform input) that can hold data under
// the load map of a prepare_* primitive: q = (j + n/2 + ld_a) mod n < ld_len.  Returns the length of the run and its
// first segment (0, all segments: the row has no empty segment).
static int data_segment_run(const RowPassArgs& a, int n, int seglen, int* first) {
    const int nseg = n / seglen, base = (int)(((long long)a.ld_a + n / 2) % n);
    int valid[64];
    int count = 0;
    for (int r = 0; r < nseg; r++) {
        const int lo = (int)(((long long)seglen * r + base) % n);
        valid[r] = a.ld_len > 0 && (lo < a.ld_len || lo + seglen > n);
        count += valid[r];
    }
    *first = 0;
    if (count == nseg || count == 0) return count == 0 ? 1 : nseg;
    for (int r = 0; r < nseg; r++)
        if (valid[r] && !valid[(r + nseg - 1) % nseg]) *first = r;
    int run = 0;
    while (run < nseg && valid[(*first + run) % nseg]) run++;
    return run == count ? run : nseg;  // two runs cannot happen for one cyclic range; be safe
}

template <class G, bool PAIR, bool WIN, int ST, int NSEG, int CJ = -1, bool W4 = false>
static void launch_band_inst(const RowPassArgs& a, unsigned blocks, const cx<float>* tw14, const cx<float>* tw_full,
                             hipStream_t s) {
    hipLaunchKernelGGL((row_pass_band_kernel<G, WIN, ST, PAIR, NSEG, CJ, W4>), dim3(blocks), dim3(G::NT), G::LDS_BYTES, s, a,
                       a.in, a.out, a.ld_win, tw14, tw_full);
}

// -- re-laid-out load windows (Win4Cache, swiftly_rowpass.h) ------------------------------------------------------
// entry (p * T + t), T = seglen / 2 lanes: the window pairs of the p-th segment pair of the K1 load loop -- segments
// (p, p + 16) for p < ns - 16, then (s, s + 1) -- at the lane's elements q = (2 t + u + c + seglen * segment) mod n
__global__ void build_win4_kernel(const float* __restrict__ win, float* __restrict__ tab, int c, int len, int ns, int n,
                                  int seglen) {
    const int T = seglen >> 1, idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (ns >> 1) * T) return;
    const int p = idx / T, t = idx - p * T, nb1 = ns - 16;
    const int s0 = p < nb1 ? p : nb1 + 2 * (p - nb1), s1 = p < nb1 ? p + 16 : s0 + 1;
    const int q0 = (2 * t + c + seglen * s0) & (n - 1), q1 = (2 * t + c + seglen * s1) & (n - 1);
    f32x4 v;
    v.x = q0 < len ? win[q0] : 0.f;
    v.y = q0 + 1 < len ? win[q0 + 1] : 0.f;
    v.z = q1 < len ? win[q1] : 0.f;
    v.w = q1 + 1 < len ? win[q1 + 1] : 0.f;
    reinterpret_cast<f32x4*>(tab)[idx] = v;
}
static int win4_enabled() {  // SWIFTLY_K1_WIN4: 0 = the plain window loads, 1 = re-laid-out window, 2 = + compact twiddle sections (A/B runs)
    static const int v = getenv("SWIFTLY_K1_WIN4") ? atoi(getenv("SWIFTLY_K1_WIN4")) : 2;
    return v;
}
const float* Win4Cache::get(const float* win, int c, int len, int ns, int n, int seglen, hipStream_t s) {
    std::lock_guard<std::mutex> lock(mu);
    for (Entry& e : items)
        if (e.win == win && e.c == c && e.len == len && e.ns == ns && e.n == n && e.seglen == seglen) {
            // (r5 advisor) a launch on another stream than the one the table was built on waits for the build -- until the
            // build has been seen complete once; after that the table is a constant like the twiddle tables
            if (!e.complete && e.built_on != s) {
                const hipError_t q = hipEventQuery(e.ready);
                if (q == hipSuccess) {
                    e.complete = true;
                } else {
                    (void)hipGetLastError();  // hipErrorNotReady must not be mistaken for a failed launch later
                    if (hipStreamWaitEvent(s, e.ready, 0) != hipSuccess) return nullptr;
                }
            }
            return e.tab;
        }
    if (items.size() >= kMaxEntries) return nullptr;
    Entry e{win, c, len, ns, n, seglen, nullptr, nullptr, s, false};
    const int count = (ns >> 1) * (seglen >> 1);
    if (hipMalloc(&e.tab, (size_t)count * 16) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&e.ready, hipEventDisableTiming) != hipSuccess) {
        (void)hipFree(e.tab);
        return nullptr;
    }
    hipLaunchKernelGGL(build_win4_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, win, e.tab, c, len, ns, n, seglen);
    if (hipGetLastError() != hipSuccess || hipEventRecord(e.ready, s) != hipSuccess) {
        (void)hipStreamSynchronize(s);
        (void)hipEventDestroy(e.ready);
        (void)hipFree(e.tab);
        return nullptr;
    }
    items.push_back(e);
    return e.tab;
}
void Win4Cache::clear() {
    std::lock_guard<std::mutex> lock(mu);
    for (Entry& e : items) {
        (void)hipEventDestroy(e.ready);
        (void)hipFree(e.tab);
    }
    items.clear();
}

template <class G, bool PAIR = false>
static int launch_band_geo(const RowPassArgs& a0, const cx<float>* tw14, const cx<float>* tw_full, hipStream_t s,
                           Win4Cache* w4cache = nullptr) {
    RowPassArgs a = a0;
    a.seg_rot = 0;
    const unsigned blocks = (unsigned)(((a.nrows + 7) / 8) * 16);
    const bool band = a.band_len > 0;
    // instances with empty segments compiled out (PAIR geometry of the 32768-point rows, 65536-point geometry): the
    // forward K1 (window, band store) and the backward finish (no load window, mapped store)
    constexpr bool SEGS = PAIR || (G::LOGN == 15 && G::LOGP == 5);
    if constexpr (SEGS) {
        constexpr int SEGLEN = PAIR ? 2 * G::T : G::T;
        int first = 0;
        const int run = data_segment_run(a, 2 * G::N, SEGLEN, &first);
#define SWF_TRY_SEG(WIN, ST, NS, CJ)                                   \
    if (run <= NS) {                                                   \
        a.seg_rot = first;                                             \
        launch_band_inst<G, PAIR, WIN, ST, NS, CJ>(a, blocks, tw14, tw_full, s); \
        return (int)hipGetLastError();                                 \
    }
        // these instances also have the conjugations compiled in (CJ): prepare_facet is an inverse transform (both
        // set), finish_facet a forward one (neither)
        const bool inv = a.conj_ld && a.conj_st, fwd = !a.conj_ld && !a.conj_st;
        if constexpr (PAIR) {
            if (a.band_len > 0 && a.ld_win && inv) {
                // forward K1: geometry with the twiddle preload (RGeoPre)
                using GP = RGeoPre<G::LOGN, G::LOGP, G::SPLIT>;
                using GC = RGeoPreC<G::LOGN, G::LOGP, G::SPLIT>;
                // ... and, when the caller owns a table cache, the window through 16-byte loads (W4)
#define SWF_TRY_SEG_PRE(NS)                                                                    \
    if (run <= NS) {                                                                           \
        a.seg_rot = first;                                                                     \
        constexpr int n_ = 2 * G::N;                                                           \
        const int c_ = (int)(((long long)a.ld_a + n_ / 2 + (long long)first * SEGLEN) % n_);   \
        a.ld_win4 = (w4cache && win4_enabled()) ? w4cache->get(a.ld_win, c_, a.ld_len, NS, n_, SEGLEN, s) : nullptr; \
        a.twc = (a.ld_win4 && win4_enabled() >= 2) ? w4cache->twc : nullptr;               \
        {   /* whole-row form: the window-rows store (win_full), or SWIFTLY_K1_WHOLE=1 for the band store */ \
            const int ew_ = launch_row_pass_whole(a, NS, tw14, tw_full, s);                    \
            if (ew_ != -2 || a.win_full) return ew_;                                           \
        }                                                                                      \
        if (a.twc)                                                                        \
            launch_band_inst<GC, PAIR, true, 1, NS, 1, true>(a, blocks, tw14, tw_full, s);     \
        else if (a.ld_win4)                                                                    \
            launch_band_inst<GP, PAIR, true, 1, NS, 1, true>(a, blocks, tw14, tw_full, s);     \
        else                                                                                   \
            launch_band_inst<GP, PAIR, true, 1, NS, 1>(a, blocks, tw14, tw_full, s);           \
        return (int)hipGetLastError();                                                         \
    }
                SWF_TRY_SEG_PRE(16)
                SWF_TRY_SEG_PRE(22)
                SWF_TRY_SEG_PRE(24)
#undef SWF_TRY_SEG_PRE
                if (a.win_full) return -2;  // window-rows store: only the instances above
            } else if (a.band_len < 0 && fwd) {
                // backward finish: compact twiddle sections when the caller owns the tables
                using GB = RGeoC<G::LOGN, G::LOGP, G::SPLIT>;
                a.twc = (w4cache && win4_enabled() >= 2) ? w4cache->twc : nullptr;
#define SWF_TRY_SEG_C(NS)                                                              \
    if (a.twc && run <= NS) {                                                          \
        a.seg_rot = first;                                                             \
        launch_band_inst<GB, PAIR, false, 2, NS, 0>(a, blocks, tw14, tw_full, s);      \
        return (int)hipGetLastError();                                                 \
    }
                SWF_TRY_SEG_C(13)
                SWF_TRY_SEG_C(16)
#undef SWF_TRY_SEG_C
                SWF_TRY_SEG(false, 2, 13, 0)
                SWF_TRY_SEG(false, 2, 16, 0)
            }
        } else {
            if (a.band_len > 0 && a.ld_win && inv) {
                SWF_TRY_SEG(true, 1, 44, 1)
            }
        }
#undef SWF_TRY_SEG
    }
#define SWF_LAUNCH_BAND(WIN, ST) launch_band_inst<G, PAIR, WIN, ST, 0>(a, blocks, tw14, tw_full, s)
    if (a.band_len < 0) {  // mapped (crop + window) store: finish_* primitives
        SWF_LAUNCH_BAND(false, 2);
    } else if (a.ld_win) {
        if (band) SWF_LAUNCH_BAND(true, 1);
        else SWF_LAUNCH_BAND(true, 0);
    } else {
        if (band) SWF_LAUNCH_BAND(false, 1);
        else SWF_LAUNCH_BAND(false, 0);
    }
#undef SWF_LAUNCH_BAND
    return (int)hipGetLastError();
}
using BandGeo64k = RGeo<15, 5, true>;  // yN = 65536: 2 x 32768 points, 1024 threads x 32, 132 KB LDS, one workgroup per CU
using BandGeo16k = RGeo<13, 4, true>;  // yN = 16384: 2 x  8192 points,  512 threads x 16, 33 KB LDS
int launch_row_pass_band(const RowPassArgs& a, const cx<float>* tw14, const cx<float>* tw_full, hipStream_t s, Win4Cache* w4) {
    return launch_row_pass_band_n(15, a, tw14, tw_full, s, w4);
}
// logn = log2 of the full row length (14, 15 or 16); tw_half = table of length 2^(logn-1), tw_full of length 2^logn
int launch_row_pass_band_n(int logn, const RowPassArgs& a, const cx<float>* tw_half, const cx<float>* tw_full, hipStream_t s,
                           Win4Cache* w4) {
    if (a.nrows <= 0) return 0;
    if (a.win_full) {
        // window-rows store (whole-row kernel): 32768-point rows, 512-column windows, 16-byte loads
        if (logn != 15 || a.win_logm != 9 || a.nwin <= 0 || !a.win_d || a.band_len <= 0) return -2;
        const bool pair_ok_w = !(a.ld_a & 1) && !(a.ld_len & 1) && !(a.in_pitch & 1) && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0;
        if (!pair_ok_w || !a.ld_win || !(a.conj_ld && a.conj_st)) return -2;
        return launch_band_geo<BandGeo5, true>(a, tw_half, tw_full, s, w4);
    }
    if (logn == 16) return launch_band_geo<BandGeo64k>(a, tw_half, tw_full, s);
    if (logn == 14) return launch_band_geo<BandGeo16k>(a, tw_half, tw_full, s);
    if (logn != 15) return -1;
    const cx<float>* tw14 = tw_half;
    // 512 threads x 32 points (two workgroups per CU) for the band store and the mapped store; 1024 x 16 (one per CU) for
    // the plain store, where the 512-thread form spills ~49 VGPRs (1024 x 16 for the band store: 1.95 vs 1.73 ms per facet, r4)
    if (a.band_len == 0) return launch_band_geo<BandGeo4>(a, tw14, tw_full, s);
    // adjacent-point (16-byte) loads need even shifts / lengths / pitches
    const bool pair_ok = !(a.ld_a & 1) && !(a.ld_len & 1) && !(a.in_pitch & 1) && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0;
    if (pair_ok) return launch_band_geo<BandGeo5, true>(a, tw14, tw_full, s, w4);
    return launch_band_geo<BandGeo5>(a, tw14, tw_full, s);
}
int row_pass_band_occupancy() {
    int n = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, row_pass_band_kernel<BandGeo5, true, 1>, BandGeo5::NT,
                                                       BandGeo5::LDS_BYTES);
    return n;
}
template <class G, bool WIN, int ST>
static int init_band() {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&row_pass_band_kernel<G, WIN, ST>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
}
template <class G, bool WIN, int ST, int NSEG = 0, bool PAIR = true, int CJ = -1, bool W4 = false>
static int init_band_pair() {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&row_pass_band_kernel<G, WIN, ST, PAIR, NSEG, CJ, W4>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
}
template <class G>
static int init_band_geo() {
    int rc = init_band<G, true, 1>();
    if (!rc) rc = init_band<G, true, 0>();
    if (!rc) rc = init_band<G, false, 1>();
    if (!rc) rc = init_band<G, false, 0>();
    if (!rc) rc = init_band<G, false, 2>();
    return rc;
}
// occupancy query (blocks per CU) for tuning / DESIGN.md
int row_pass_half_occupancy(int lds_bytes) {
    int n = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, row_pass_split_kernel<SplitGeo2, 1, false>, SplitGeo2::NT,
                                                       lds_bytes < 0 ? SplitGeo2::LDS_BYTES : (size_t)lds_bytes);
    return n;
}
template <class G, int LOGS>
static int init_split() {
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&row_pass_split_kernel<G, LOGS, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (rc) return rc;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&row_pass_split_kernel<G, LOGS, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
}
int init_row_pass() {
    {
        int rcb = init_band_geo<BandGeo5>();
        if (!rcb) rcb = init_band_pair<BandGeo5, true, 1>();
        if (!rcb) rcb = init_band_pair<BandGeo5, true, 0>();
        if (!rcb) rcb = init_band_pair<BandGeo5, false, 1>();
        if (!rcb) rcb = init_band_pair<BandGeo5, false, 0>();
        if (!rcb) rcb = init_band_pair<BandGeo5, false, 2>();
        using BandGeo5Pre = RGeoPre<14, 5, true>;
        if (!rcb) rcb = init_band_pair<BandGeo5Pre, true, 1, 16, true, 1>();
        if (!rcb) rcb = init_band_pair<BandGeo5Pre, true, 1, 22, true, 1>();
        if (!rcb) rcb = init_band_pair<BandGeo5Pre, true, 1, 24, true, 1>();
        if (!rcb) rcb = init_band_pair<BandGeo5Pre, true, 1, 16, true, 1, true>();
        if (!rcb) rcb = init_band_pair<BandGeo5Pre, true, 1, 22, true, 1, true>();
        if (!rcb) rcb = init_band_pair<BandGeo5Pre, true, 1, 24, true, 1, true>();
        using BandGeo5PreC = RGeoPreC<14, 5, true>;
        if (!rcb) rcb = init_band_pair<BandGeo5PreC, true, 1, 16, true, 1, true>();
        if (!rcb) rcb = init_band_pair<BandGeo5PreC, true, 1, 22, true, 1, true>();
        if (!rcb) rcb = init_band_pair<BandGeo5PreC, true, 1, 24, true, 1, true>();
        if (!rcb) rcb = init_band_pair<BandGeo5, false, 2, 13, true, 0>();
        if (!rcb) rcb = init_band_pair<BandGeo5, false, 2, 16, true, 0>();
        using BandGeo5C = RGeoC<14, 5, true>;
        if (!rcb) rcb = init_band_pair<BandGeo5C, false, 2, 13, true, 0>();
        if (!rcb) rcb = init_band_pair<BandGeo5C, false, 2, 16, true, 0>();
        if (!rcb) rcb = init_band_pair<BandGeo64k, true, 1, 44, false, 1>();
        if (!rcb) rcb = init_row_pass_whole();
        if (!rcb) rcb = init_band_geo<BandGeo4>();
        if (!rcb) rcb = init_band_geo<BandGeo64k>();
        if (!rcb) rcb = init_band_geo<BandGeo16k>();
        if (rcb) return rcb;
    }
    {
        int rc0 = init_split<SplitGeo2, 1>();
        if (rc0) return rc0;
    }
    int rc = init_one<13>();
    if (!rc) rc = init_one<14>();
    if (!rc) rc = init_one<15>();
    return rc;
}

}  // namespace swf
