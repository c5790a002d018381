// instantiations + dispatch of the lean column-tile pass (complex64)
#include <cstdlib>

#include "swiftly_colpass.h"

namespace swf {

// The single 512-point pass on 32-column tiles (ColPassArgs::tile32; r5): 512 threads and 64 KiB per workgroup instead of
// 1024 threads and 128 KiB.  Alone on the chip the two forms are equal (r3: 43.37 vs 43.29 ms per pass); in the overlapped
// wave loop K3 shares the CUs with the K2 workgroups of the following waves, which a whole-CU workgroup cannot: 64k pass
// 38.07 / 37.84 / 38.14 -> 37.79 / 37.68 / 37.95 ms (interleaved same-box pairs, gpurun_out/r5r).  SWIFTLY_COL512_TILE32=0
// switches it off (A/B runs).  (The same tiles for the two passes of K2's four-step, same session: pass A 38.2 / 37.6 ->
// 39.1 / 39.7 ms, pass B 37.7 / 38.4 -- not kept.)
using CGeo512Half = CGeo<9, 5, true, 32>;
static int col512_tile32() {
    static const int v = getenv("SWIFTLY_COL512_TILE32") ? atoi(getenv("SWIFTLY_COL512_TILE32")) : 1;
    return v;
}

template <int LOGN, int MODE>
static int launch_mode(const ColPassArgs& a, const ColZ& cz, int outer, int nbatch, hipStream_t s) {
    using G = typename CGeoFor<LOGN>::type;
    if constexpr (LOGN == 9 && MODE == 2) {
        if (a.tile32 && !a.gs && col512_tile32()) {
            using GH = CGeo512Half;
            dim3 hgrid((unsigned)((a.ncols + GH::COLS - 1) / GH::COLS), (unsigned)outer, (unsigned)nbatch);
            hipLaunchKernelGGL((col_pass_kernel<GH, 2, true>), hgrid, dim3(GH::NT), GH::LDS_BYTES, s, a, a.in, a.out, a.ld_win,
                               a.ld_win2, a.st_win, a.st_win2, a.st_rowmap, a.tw, a.tw_full, cz);
            return (int)hipGetLastError();
        }
    }
    dim3 grid((unsigned)((a.ncols + G::COLS - 1) / G::COLS), (unsigned)outer, (unsigned)nbatch);
    if constexpr (MODE == 1 || G::HALF) {
        if (a.gs) return (int)hipErrorInvalidConfiguration;  // no gather-sum instance of this geometry: never fall through to the plain load
    }
    if constexpr (MODE != 1 && !G::HALF) {
        if (a.gs) {  // gather-sum load (backward pass); the source contributions are small and re-read: cacheable
            hipLaunchKernelGGL((col_pass_kernel<G, MODE, true, true>), grid, dim3(G::NT), G::LDS_BYTES, s, a, a.in, a.out,
                               a.ld_win, a.ld_win2, a.st_win, a.st_win2, a.st_rowmap, a.tw, a.tw_full, cz);
            return (int)hipGetLastError();
        }
    }
    if (MODE == 2 || a.scratch_nt)
        hipLaunchKernelGGL((col_pass_kernel<G, MODE, true>), grid, dim3(G::NT), G::LDS_BYTES, s, a, a.in, a.out, a.ld_win,
                           a.ld_win2, a.st_win, a.st_win2, a.st_rowmap, a.tw, a.tw_full, cz);
    else
        hipLaunchKernelGGL((col_pass_kernel<G, (MODE == 2 ? 0 : MODE), false>), grid, dim3(G::NT), G::LDS_BYTES, s, a, a.in,
                           a.out, a.ld_win, a.ld_win2, a.st_win, a.st_win2, a.st_rowmap, a.tw, a.tw_full, cz);
    return (int)hipGetLastError();
}
// float64 arithmetic (ColPassArgs::f64): lengths 32 .. 512
constexpr int kColF64MinLog = 5;
template <int LOGN, int MODE>
static int launch_mode_f64(const ColPassArgs& a, const ColZ& cz, int outer, int nbatch, hipStream_t s) {
    if constexpr (LOGN < kColF64MinLog || LOGN > kColPassMaxLogF64) {
        return (int)hipErrorInvalidConfiguration;
    } else {
        using G = typename CGeoFor<LOGN, double>::type;
        if (!a.twd || (MODE == 0 && !a.twd_full)) return (int)hipErrorInvalidValue;
        if (a.gs) {
            using GG = typename CGeoFor<LOGN, double>::type_gs;
            if constexpr (MODE != 1 && !GG::HALF) {
                dim3 ggrid((unsigned)((a.ncols + GG::COLS - 1) / GG::COLS), (unsigned)outer, (unsigned)nbatch);
                hipLaunchKernelGGL((col_pass_kernel<GG, MODE, true, true, double>), ggrid, dim3(GG::NT), GG::LDS_BYTES, s, a, a.in,
                                   a.out, a.ld_win, a.ld_win2, a.st_win, a.st_win2, a.st_rowmap, a.twd, a.twd_full, cz);
                return (int)hipGetLastError();
            } else {
                return (int)hipErrorInvalidConfiguration;
            }
        }
        dim3 grid((unsigned)((a.ncols + G::COLS - 1) / G::COLS), (unsigned)outer, (unsigned)nbatch);
        if (MODE == 2 || a.scratch_nt)
            hipLaunchKernelGGL((col_pass_kernel<G, MODE, true, false, double>), grid, dim3(G::NT), G::LDS_BYTES, s, a, a.in,
                               a.out, a.ld_win, a.ld_win2, a.st_win, a.st_win2, a.st_rowmap, a.twd, a.twd_full, cz);
        else
            hipLaunchKernelGGL((col_pass_kernel<G, (MODE == 2 ? 0 : MODE), false, false, double>), grid, dim3(G::NT),
                               G::LDS_BYTES, s, a, a.in, a.out, a.ld_win, a.ld_win2, a.st_win, a.st_win2, a.st_rowmap, a.twd,
                               a.twd_full, cz);
        return (int)hipGetLastError();
    }
}
template <int LOGN, int MODE>
static int init_mode_f64() {
    if constexpr (LOGN < kColF64MinLog || LOGN > kColPassMaxLogF64) {
        return 0;
    } else {
        using G = typename CGeoFor<LOGN, double>::type;
        int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&col_pass_kernel<G, MODE, true, false, double>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
        if (!rc && MODE != 2)
            rc = (int)hipFuncSetAttribute(
                reinterpret_cast<const void*>(&col_pass_kernel<G, (MODE == 2 ? 0 : MODE), false, false, double>),
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
        using GG = typename CGeoFor<LOGN, double>::type_gs;
        if constexpr (MODE != 1 && !GG::HALF) {
            if (!rc)
                rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&col_pass_kernel<GG, MODE, true, true, double>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)GG::LDS_BYTES);
        }
        return rc;
    }
}

template <int LOGN>
static int launch_one(int mode, const ColPassArgs& a, const ColZ& cz, int outer, int nbatch, hipStream_t s) {
    if (a.f64) {
        if (mode == 0) return launch_mode_f64<LOGN, 0>(a, cz, outer, nbatch, s);
        if (mode == 1) return launch_mode_f64<LOGN, 1>(a, cz, outer, nbatch, s);
        return launch_mode_f64<LOGN, 2>(a, cz, outer, nbatch, s);
    }
    if (mode == 0) return launch_mode<LOGN, 0>(a, cz, outer, nbatch, s);
    if (mode == 1) return launch_mode<LOGN, 1>(a, cz, outer, nbatch, s);
    return launch_mode<LOGN, 2>(a, cz, outer, nbatch, s);
}
template <int LOGN, int MODE>
static int init_mode() {
    using G = typename CGeoFor<LOGN>::type;
    if (G::LDS_BYTES == 0) return 0;
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&col_pass_kernel<G, MODE, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (!rc && MODE != 2)
        rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&col_pass_kernel<G, (MODE == 2 ? 0 : MODE), false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if constexpr (MODE != 1 && !G::HALF) {
        if (!rc)
            rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&col_pass_kernel<G, MODE, true, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    }
    return rc;
}
template <int LOGN>
static int init_one() {
    int rc = init_mode<LOGN, 0>();
    if (!rc) rc = init_mode<LOGN, 1>();
    if (!rc) rc = init_mode<LOGN, 2>();
    if (!rc) rc = init_mode_f64<LOGN, 0>();
    if (!rc) rc = init_mode_f64<LOGN, 1>();
    if (!rc) rc = init_mode_f64<LOGN, 2>();
    return rc;
}

template <int LO, int HI>
struct CDispatch {
    static int launch(int logn, int mode, const ColPassArgs& a, const ColZ& cz, int outer, int nbatch, hipStream_t s) {
        if (logn == LO) return launch_one<LO>(mode, a, cz, outer, nbatch, s);
        if constexpr (LO < HI) return CDispatch<LO + 1, HI>::launch(logn, mode, a, cz, outer, nbatch, s);
        return -1;
    }
    static int init() {
        int rc = init_one<LO>();
        if (rc) return rc;
        if constexpr (LO < HI) return CDispatch<LO + 1, HI>::init();
        return 0;
    }
};

int launch_col_pass(int logn, int mode, const ColPassArgs& a, const ColZ& cz, int outer, int nbatch, hipStream_t s) {
    return CDispatch<kColPassMinLog, kColPassMaxLog>::launch(logn, mode, a, cz, outer, nbatch, s);
}
int init_col_pass() {
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&col_pass_kernel<CGeo512Half, 2, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)CGeo512Half::LDS_BYTES);
    if (rc) return rc;
    return CDispatch<kColPassMinLog, kColPassMaxLog>::init();
}
bool col_pass_f64_supported(int logn) { return logn >= kColF64MinLog && logn <= kColPassMaxLogF64; }

}  // namespace swf
