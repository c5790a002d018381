// instantiations of the radix-Q pass (swiftly_mixed.h)
#include "swiftly_mixed.h"

namespace swf {

template <typename R>
static int launch_any(int Q, const RowsArgs<R>& a, const OffTab& tab, const MixedArgs<R>& x, int nbatch, hipStream_t s) {
    const long long nfast = a.rowfast ? (long long)a.nrows : (long long)x.M;
    const long long nslow = a.rowfast ? (long long)x.M : (long long)a.nrows;
    dim3 grid((unsigned)((nfast + 255) / 256), (unsigned)(nslow < 65535 ? nslow : 65535), (unsigned)nbatch);
#define MX_CASE(QQ)                                                                                   \
    if (Q == QQ) {                                                                                    \
        hipLaunchKernelGGL((mixed_radix_pass_kernel<R, QQ>), grid, dim3(256), 0, s, a, tab, x);         \
        return (int)hipGetLastError();                                                                \
    }
    MX_CASE(3) MX_CASE(5) MX_CASE(7) MX_CASE(9)
#undef MX_CASE
    return (int)hipErrorInvalidValue;
}

int launch_mixed_gs_pass(int Q, const MixedGsArgs& g, const MixedArgs<float>& x, int nfacets, hipStream_t s) {
    dim3 grid((unsigned)((g.ncols + 255) / 256), (unsigned)(x.M < 65535 ? x.M : 65535), (unsigned)nfacets);
#define GS_CASE(QQ)                                                                      \
    if (Q == QQ) {                                                                       \
        hipLaunchKernelGGL((mixed_gs_pass_kernel<QQ>), grid, dim3(256), 0, s, g, x);       \
        return (int)hipGetLastError();                                                   \
    }
    GS_CASE(3) GS_CASE(5) GS_CASE(7) GS_CASE(9)
#undef GS_CASE
    return (int)hipErrorInvalidValue;
}

int launch_mixed_pass(int Q, const RowsArgs<float>& a, const OffTab& tab, const MixedArgs<float>& x, int nbatch, hipStream_t s) {
    return launch_any<float>(Q, a, tab, x, nbatch, s);
}
int launch_mixed_pass(int Q, const RowsArgs<double>& a, const OffTab& tab, const MixedArgs<double>& x, int nbatch, hipStream_t s) {
    return launch_any<double>(Q, a, tab, x, nbatch, s);
}

}  // namespace swf
