// instantiations + dispatch of the fused "sum over an off1 group + finish along the strided axis" kernel
#include "swiftly_groupfinish.h"

namespace swf {

template <int LOGM, int LOGX>
static int launch_one_g(const GroupFinishArgs& a, int nbatch, hipStream_t s) {
    using S = GFGeo<LOGM, LOGX>;
    dim3 grid((unsigned)((a.ncols + 15) / 16), (unsigned)nbatch, (unsigned)a.ngroups);
    hipLaunchKernelGGL((group_finish_kernel<LOGM, LOGX>), grid, dim3(S::NT), S::LDS_BYTES, s, a);
    return (int)hipGetLastError();
}
template <int LOGM, int LOGX>
static int init_one_g() {
    using S = GFGeo<LOGM, LOGX>;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&group_finish_kernel<LOGM, LOGX>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS_BYTES);
}

#define GF_PAIRS(X) X(7, 8) X(7, 10) X(8, 9) X(8, 10) X(9, 10)

int launch_group_finish(int logm, int logx, const GroupFinishArgs& a, int nbatch, hipStream_t s) {
#define GF_CASE(M, XX) \
    if (logm == M && logx == XX) return launch_one_g<M, XX>(a, nbatch, s);
    GF_PAIRS(GF_CASE)
#undef GF_CASE
    return -1;
}
int init_group_finish() {
    int rc = 0;
#define GF_INIT(M, XX) \
    if (!rc) rc = init_one_g<M, XX>();
    GF_PAIRS(GF_INIT)
#undef GF_INIT
    return rc;
}
bool group_finish_supported(int logm, int logx) {
#define GF_HAS(M, XX) \
    if (logm == M && logx == XX) return true;
    GF_PAIRS(GF_HAS)
#undef GF_HAS
    return false;
}

}  // namespace swf
