// Instantiates fft_rows_kernel for one scalar type over every supported
// transform length and provides the runtime dispatch.  Included by
// fft_rows_f32.hip and fft_rows_f64.hip (separate TUs so they build in
// parallel).
#pragma once
#include "swiftly_rows.h"

namespace swf {

template <typename R, int LOGN>
static int launch_one(const RowsArgs<R>& a, const OffTab& tab, hipStream_t s) {
    using G = typename GeoFor<R, LOGN>::type;
    const long long total = (long long)a.nrows * a.outer;
    if (total <= 0) return 0;
    unsigned grid = (unsigned)((total + G::RB - 1) / G::RB);
    if (a.outer_group) {  // row blocks padded to a multiple of 8 (one per XCD), `outer` workgroups each
        const long long rowblocks = ((long long)a.nrows + G::RB - 1) / G::RB;
        grid = (unsigned)(((rowblocks + 7) / 8) * 8 * a.outer);
    }
    hipLaunchKernelGGL((fft_rows_kernel<G, R>), dim3(grid, a.nbatch > 0 ? a.nbatch : 1), dim3(G::NT), G::LDS_BYTES, s, a, tab);
    return (int)hipGetLastError();
}

template <typename R, int LOGN>
static int init_one() {
    using G = typename GeoFor<R, LOGN>::type;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_rows_kernel<G, R>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
}

template <typename R, int LO, int HI>
struct Dispatch {
    static int launch(int logn, const RowsArgs<R>& a, const OffTab& tab, hipStream_t s) {
        if (logn == LO) return launch_one<R, LO>(a, tab, s);
        if constexpr (LO < HI) return Dispatch<R, LO + 1, HI>::launch(logn, a, tab, s);
        return -1;
    }
    static int init() {
        int rc = init_one<R, LO>();
        if (rc) return rc;
        if constexpr (LO < HI) return Dispatch<R, LO + 1, HI>::init();
        return 0;
    }
};

}  // namespace swf
