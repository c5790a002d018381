// instantiations + dispatch of the lean long-row kernel (complex64)
#include <cstdlib>

#include "swiftly_rowpass.h"

namespace swf {

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
template <class G>
static int launch_half(const RowPassArgs& a, const cx<float>* tw_half, const cx<float>* tw_full, hipStream_t s) {
    const unsigned blocks = (unsigned)(((a.nrows + 7) / 8) * 16);
    hipLaunchKernelGGL((row_pass_half_kernel<G>), dim3(blocks), dim3(G::NT), G::LDS_BYTES, s, a, a.in, a.out, a.ld_win,
                       tw_half, tw_full);
    return (int)hipGetLastError();
}
using HalfGeoA = RGeo<14, 4, true>;  // 1024 threads x 16 points, 2 workgroups / CU
using HalfGeoB = RGeo<14, 5, true>;  // 512 threads x 32 points (tuning alternative)
int launch_row_pass_half(const RowPassArgs& a, const cx<float>* tw_half, const cx<float>* tw_full, hipStream_t s) {
    if (a.nrows <= 0) return 0;
    static const bool alt = getenv("SWIFTLY_K2_P32") != nullptr;
    return alt ? launch_half<HalfGeoB>(a, tw_half, tw_full, s) : launch_half<HalfGeoA>(a, tw_half, tw_full, s);
}
int init_row_pass() {
    {
        int rc0 = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&row_pass_half_kernel<HalfGeoA>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)HalfGeoA::LDS_BYTES);
        if (!rc0)
            rc0 = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&row_pass_half_kernel<HalfGeoB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)HalfGeoB::LDS_BYTES);
        if (rc0) return rc0;
    }
    int rc = init_one<13>();
    if (!rc) rc = init_one<14>();
    if (!rc) rc = init_one<15>();
    return rc;
}

}  // namespace swf
