// instantiations + dispatch of the whole-row form of K1 (swiftly_rowwhole.h)
#include <cstdlib>

#if SWF_TRACE
#define swf_trace_buf swf_wtrace_buf   // this translation unit's own stamp buffer (device symbols are per code object)
#endif
#include "swiftly_rowwhole.h"

namespace swf {

#if SWF_TRACE
__device__ unsigned long long swf_wtrace_buf[kTraceBlocks * kTracePoints];
extern "C" int swiftly_hip_wtrace_fetch(void* host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(swf_wtrace_buf), bytes);
}
#endif


// SWIFTLY_K1_WHOLE: 1 = the plain band store too runs as one persistent workgroup per CU that owns whole rows (A/B runs of
// the r6 experiment: no faster than the two-workgroup form, tools/experiments/README.md); default 0 -- the whole-row form is
// the K1 of the axis-1-first pipeline only (RowPassArgs::win_full).  Read once per process.
static int whole_enabled() {
    static const int v = getenv("SWIFTLY_K1_WHOLE") ? atoi(getenv("SWIFTLY_K1_WHOLE")) : 0;
    return v;
}
int row_pass_whole_grid() {  // one workgroup per CU (256 VGPRs x 512 threads: a CU holds exactly one)
    static const int v = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (getenv("SWIFTLY_K1_WHOLE_GRID")) cus = atoi(getenv("SWIFTLY_K1_WHOLE_GRID"));
        return cus > 0 ? cus : 256;
    }();
    return v;
}

template <int NSEG, bool WIN = false>
static int launch_whole_inst(const RowPassArgs& a, const cx<float>* tw14, const cx<float>* tw_full, hipStream_t s) {
    const int grid = a.nrows < row_pass_whole_grid() ? a.nrows : row_pass_whole_grid();
    hipLaunchKernelGGL((row_pass_whole_kernel<NSEG, WIN>), dim3((unsigned)grid), dim3(RGeoWhole::NT), WIN ? kWholeWinLds : RGeoWhole::LDS_BYTES, s, a,
                       a.in, a.out, tw14, tw_full, a.row_win, a.in_rowmap);
    return (int)hipGetLastError();
}

// forward K1 with the re-laid-out window (a.ld_win4) and the compact twiddle sections (a.twc) set, a.seg_rot chosen for
// `nseg` data segments; returns -2 when this form does not apply (the caller launches the two-workgroup kernel)
int launch_row_pass_whole(const RowPassArgs& a, int nseg, const cx<float>* tw14, const cx<float>* tw_full, hipStream_t s) {
    if (!a.ld_win4 || !a.twc || a.band_len <= 0 || !(a.conj_ld && a.conj_st)) return -2;
    if (a.win_full) {
        if (a.win_logm != 9 || !a.win_d || a.nwin <= 0 || a.nwin > kWholeMaxWindows || !a.win_fn || !a.win_tw_m || !a.win_twc_m ||
            2 * a.band_half > row_pass_whole_stage_columns())
            return -2;
        switch (nseg) {
            case 16: return launch_whole_inst<16, true>(a, tw14, tw_full, s);
            case 22: return launch_whole_inst<22, true>(a, tw14, tw_full, s);
            case 24: return launch_whole_inst<24, true>(a, tw14, tw_full, s);
            default: return -2;
        }
    }
    if (!whole_enabled()) return -2;
    switch (nseg) {
        case 16: return launch_whole_inst<16>(a, tw14, tw_full, s);
        case 22: return launch_whole_inst<22>(a, tw14, tw_full, s);
        case 24: return launch_whole_inst<24>(a, tw14, tw_full, s);
        default: return -2;
    }
}

int row_pass_whole_stage_columns() {
    using GM = Geo<float, 9, 3, RGeoWhole::NT, false>;
    return (int)((RGeoWhole::LDS_BYTES - GM::LDS_BYTES) / 8) & ~31;  // stage | m-point exchange rows
}
template <int NSEG, bool WIN>
static int init_whole_inst() {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&row_pass_whole_kernel<NSEG, WIN>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(WIN ? kWholeWinLds : RGeoWhole::LDS_BYTES));
}
int init_row_pass_whole() {
    int rc = init_whole_inst<16, false>();
    if (!rc) rc = init_whole_inst<22, false>();
    if (!rc) rc = init_whole_inst<24, false>();
    if (!rc) rc = init_whole_inst<16, true>();
    if (!rc) rc = init_whole_inst<22, true>();
    if (!rc) rc = init_whole_inst<24, true>();
    return rc;
}

}  // namespace swf
