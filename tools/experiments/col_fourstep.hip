// instantiations + dispatch of the fused four-step column transform (swiftly_fourstep.h)
#include "swiftly_fourstep.h"

namespace swf {

template <int L1, int L2>
struct FsGeo {
    using GA = typename CGeoFor<L1>::type;
    using GB = typename CGeoFor<L2>::type;
    static constexpr int SUB = GB::NT / GA::NT;
    static constexpr size_t LDS = (size_t)SUB * GA::LDS_BYTES > GB::LDS_BYTES ? (size_t)SUB * GA::LDS_BYTES : GB::LDS_BYTES;
};

template <int L1, int L2, bool SNT>
static int launch_one(const ColPassArgs& a, const ColPassArgs& b, const ColZS& za, const ColZS& zb, int nbatch, int lag,
                      unsigned* counters, unsigned* err, hipStream_t s) {
    using F = FsGeo<L1, L2>;
    FourStepSched S;
    S.counters = counters;
    S.err = err;
    S.col_tiles = (a.ncols + 63) / 64;
    S.chunks = nbatch * S.col_tiles;
    S.a_per = (1 << L2) / F::SUB;  // pass A: one transform per y2 (2^l2 of them), SUB per workgroup
    S.b_per = 1 << L1;             // pass B: one transform per k1
    S.lag = lag;
    const long long blocks = (long long)(S.chunks + lag) * (S.a_per + S.b_per);
    if (blocks >= (1ll << 31)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL((col_fourstep_kernel<typename F::GA, typename F::GB, SNT>), dim3((unsigned)blocks), dim3(F::GB::NT),
                       F::LDS, s, a, b, za, zb, S);
    return (int)hipGetLastError();
}
template <int L1, int L2>
static int init_one() {
    using F = FsGeo<L1, L2>;
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&col_fourstep_kernel<typename F::GA, typename F::GB, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)F::LDS);
    if (!rc)
        rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&col_fourstep_kernel<typename F::GA, typename F::GB, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)F::LDS);
    return rc;
}

// (l1, l2) = (floor(log N / 2), rest) for N = 2^12 .. 2^17
#define FS_PAIRS(X) X(6, 6) X(6, 7) X(7, 7) X(7, 8) X(8, 8) X(8, 9)

int launch_col_fourstep(int l1, int l2, const ColPassArgs& a, const ColPassArgs& b, const ColZS& za, const ColZS& zb,
                        int nbatch, int lag, unsigned* counters, unsigned* err, hipStream_t s) {
#define FS_CASE(P, Q)                                                                                  \
    if (l1 == P && l2 == Q)                                                                            \
        return a.scratch_nt ? launch_one<P, Q, true>(a, b, za, zb, nbatch, lag, counters, err, s)      \
                            : launch_one<P, Q, false>(a, b, za, zb, nbatch, lag, counters, err, s);
    FS_PAIRS(FS_CASE)
#undef FS_CASE
    return -1;
}
int init_col_fourstep() {
    int rc = 0;
#define FS_INIT(P, Q) \
    if (!rc) rc = init_one<P, Q>();
    FS_PAIRS(FS_INIT)
#undef FS_INIT
    return rc;
}
bool col_fourstep_supported(int l1, int l2) {
#define FS_HAS(P, Q) \
    if (l1 == P && l2 == Q) return true;
    FS_PAIRS(FS_HAS)
#undef FS_HAS
    return false;
}

}  // namespace swf
