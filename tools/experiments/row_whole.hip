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


// SWIFTLY_K1_WHOLE: 1 (default) = one persistent workgroup per CU owns whole rows, 0 = the two-workgroup form (A/B runs;
// read once per process)
static int whole_enabled() {
    static const int v = getenv("SWIFTLY_K1_WHOLE") ? atoi(getenv("SWIFTLY_K1_WHOLE")) : 1;
    return v;
}
static int whole_grid() {  // one workgroup per CU (256 VGPRs x 512 threads: a CU holds exactly one)
    static const int v = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (getenv("SWIFTLY_K1_WHOLE_GRID")) cus = atoi(getenv("SWIFTLY_K1_WHOLE_GRID"));
        return cus > 0 ? cus : 256;
    }();
    return v;
}

template <int NSEG>
static int launch_whole_inst(const RowPassArgs& a, const cx<float>* tw14, const cx<float>* tw_full, hipStream_t s) {
    const int grid = a.nrows < whole_grid() ? a.nrows : whole_grid();
    hipLaunchKernelGGL((row_pass_whole_kernel<NSEG>), dim3((unsigned)grid), dim3(RGeoWhole::NT), RGeoWhole::LDS_BYTES, s, a,
                       a.in, a.out, tw14, tw_full, a.row_win, a.in_rowmap);
    return (int)hipGetLastError();
}

// forward K1 with the re-laid-out window (a.ld_win4) and the compact twiddle sections (a.twc) set, a.seg_rot chosen for
// `nseg` data segments; returns -2 when this form does not apply (the caller launches the two-workgroup kernel)
int launch_row_pass_whole(const RowPassArgs& a, int nseg, const cx<float>* tw14, const cx<float>* tw_full, hipStream_t s) {
    if (!whole_enabled() || !a.ld_win4 || !a.twc || a.band_len <= 0 || !(a.conj_ld && a.conj_st)) return -2;
    switch (nseg) {
        case 16: return launch_whole_inst<16>(a, tw14, tw_full, s);
        case 22: return launch_whole_inst<22>(a, tw14, tw_full, s);
        case 24: return launch_whole_inst<24>(a, tw14, tw_full, s);
        default: return -2;
    }
}

template <int NSEG>
static int init_whole_inst() {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&row_pass_whole_kernel<NSEG>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)RGeoWhole::LDS_BYTES);
}
int init_row_pass_whole() {
    int rc = init_whole_inst<16>();
    if (!rc) rc = init_whole_inst<22>();
    if (!rc) rc = init_whole_inst<24>();
    return rc;
}

}  // namespace swf
