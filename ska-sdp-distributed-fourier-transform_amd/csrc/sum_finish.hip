// instantiations of the fused sum + finish row kernel for the (m, xM) pairs of the catalogue families
#include "swiftly_sumfinish.h"

namespace swf {

template <int LOGM, int LOGX>
static int launch_one(const SumFinishArgs& a, int nbatch, hipStream_t s) {
    using S = SFGeo<LOGM, LOGX>;
    dim3 grid((unsigned)((a.nrows + S::RB - 1) / S::RB), (unsigned)nbatch);
    hipLaunchKernelGGL((sum_finish_rows_kernel<LOGM, LOGX>), grid, dim3(S::NT), S::LDS_BYTES, s, a);
    return (int)hipGetLastError();
}
template <int LOGM, int LOGX>
static int init_one() {
    using S = SFGeo<LOGM, LOGX>;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&sum_finish_rows_kernel<LOGM, LOGX>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS_BYTES);
}

template <int LOGM, int LOGX>
static int launch_one_f(const SumFinishFacetArgs& a, int nbatch, hipStream_t s) {
    using S = SFGeo<LOGM, LOGX>;
    dim3 grid((unsigned)((a.nrows + S::RB - 1) / S::RB), (unsigned)nbatch);
    constexpr size_t lds = sum_finish_facets_kernel_lds<LOGM, LOGX>();
    hipLaunchKernelGGL((sum_finish_facets_kernel<LOGM, LOGX>), grid, dim3(S::NT), lds, s, a);
    return (int)hipGetLastError();
}
template <int LOGM, int LOGX>
static int init_one_f() {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&sum_finish_facets_kernel<LOGM, LOGX>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sum_finish_facets_kernel_lds<LOGM, LOGX>()));
}
template <int LOGM, int LOGX>
static int launch_one_s(const SplitFacetArgs& a, int nbatch, hipStream_t s) {
    using S = SFGeo<LOGM, LOGX>;
    dim3 grid((unsigned)((a.nrows + S::RB - 1) / S::RB), (unsigned)nbatch);
    constexpr size_t lds = sum_finish_facets_lds<LOGM, LOGX>();  // (the same wave-parallel geometry)
    hipLaunchKernelGGL((split_prepare_facets_kernel<LOGM, LOGX>), grid, dim3(S::NT), lds, s, a);
    return (int)hipGetLastError();
}
template <int LOGM, int LOGX>
static int init_one_s() {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&split_prepare_facets_kernel<LOGM, LOGX>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sum_finish_facets_lds<LOGM, LOGX>()));
}

template <int LOGM>
static int launch_axis1(const Axis1RowsArgs& a, int nfacets, hipStream_t s) {
    using GM = typename Axis1Geo<LOGM>::GM;
    dim3 grid((unsigned)((a.nrows + GM::RB - 1) / GM::RB), (unsigned)nfacets);
    hipLaunchKernelGGL((axis1_rows_kernel<LOGM>), grid, dim3(kAxis1Threads), Axis1Geo<LOGM>::LDS_BYTES, s, a);
    return (int)hipGetLastError();
}
int launch_axis1_rows(int logm, const Axis1RowsArgs& a, int nfacets, hipStream_t s) {
    switch (logm) {
        case 7: return launch_axis1<7>(a, nfacets, s);
        case 8: return launch_axis1<8>(a, nfacets, s);
        case 9: return launch_axis1<9>(a, nfacets, s);
        case 10: return launch_axis1<10>(a, nfacets, s);
        default: return -1;
    }
}

#define SF_PAIRS(X) X(7, 8) X(7, 10) X(8, 9) X(8, 10) X(9, 10) X(9, 11) X(10, 11) X(10, 12)

int launch_sum_finish_rows(int logm, int logx, const SumFinishArgs& a, int nbatch, hipStream_t s) {
#define SF_CASE(M, XX) \
    if (logm == M && logx == XX) return launch_one<M, XX>(a, nbatch, s);
    SF_PAIRS(SF_CASE)
#undef SF_CASE
    return -1;
}
int launch_sum_finish_facets(int logm, int logx, const SumFinishFacetArgs& a, int nbatch, hipStream_t s) {
#define SF_CASE_F(M, XX) \
    if (logm == M && logx == XX) return launch_one_f<M, XX>(a, nbatch, s);
    SF_PAIRS(SF_CASE_F)
#undef SF_CASE_F
    return -1;
}
int launch_split_prepare_facets(int logm, int logx, const SplitFacetArgs& a, int nbatch, hipStream_t s) {
#define SF_CASE_S(M, XX) \
    if (logm == M && logx == XX) return launch_one_s<M, XX>(a, nbatch, s);
    SF_PAIRS(SF_CASE_S)
#undef SF_CASE_S
    return -1;
}
int init_sum_finish_rows() {
    int rc = 0;
#define SF_INIT(M, XX)               \
    if (!rc) rc = init_one<M, XX>(); \
    if (!rc) rc = init_one_f<M, XX>(); \
    if (!rc) rc = init_one_s<M, XX>();
    SF_PAIRS(SF_INIT)
#undef SF_INIT
    return rc;
}
bool sum_finish_supported(int logm, int logx) {
#define SF_HAS(M, XX) \
    if (logm == M && logx == XX) return true;
    SF_PAIRS(SF_HAS)
#undef SF_HAS
    return false;
}

}  // namespace swf
