// C ABI of libswiftly_hip.so (see include/swiftly_hip.h).
//
// Each primitive of the reference (fourier_transform/core.py:189-484) is
// expressed as ONE launch of the mapped row FFT kernel (swiftly_rows.h) or of
// the modular gather/scatter kernel below:
//
//   primitive             N    load map (q = (ci+a) mod N)          store map (d = (ci+a) mod N)        dir  scale
//   prepare_facet         yN   a=-(off+yN/2-yB//2) len=yB win=1/pswf  identity                          inv  1/yN
//   add_to_subgrid        m    identity                             a=-s' len=m c=xM/2-m/2+s' mod xM Fn  fwd  1   (+=)
//   finish_subgrid        xM   identity                             a=-(xM/2-xA//2+off) len=xA [mask]   inv  1/xM
//   prepare_subgrid       xM   a=-(xM/2-xA//2+off) len=xA           identity                            fwd  1
//   extract_from_subgrid  m    a=-s' len=m c=xM/2-m/2+s' mod xM Fn  identity                            inv  1/m
//   finish_facet          yN   identity                             a=-(yN/2-yB//2+off) len=yB 1/pswf   fwd  1
//   extract_from_facet / add_to_facet: pure modular gather / scatter-add (no FFT)
//
// with s = floor(subgrid_off*yN/N), s' = floor(facet_off*xM/N), all arrays
// centred (origin at index n//2).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "../../include/swiftly_hip.h"
#include "swiftly_rows.h"

using namespace swf;

// ---------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return fail(SWIFTLY_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static inline int64_t floordiv(int64_t a, int64_t b) {
    int64_t q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) q--;
    return q;
}
static inline int pmod(int64_t a, int64_t n) {
    int64_t r = a % n;
    if (r < 0) r += n;
    return (int)r;
}
static inline int ilog2_exact(int64_t n) {
    if (n <= 0 || (n & (n - 1))) return -1;
    int l = 0;
    while ((int64_t(1) << l) < n) l++;
    return l;
}

struct swiftly_hip {
    int64_t N, yN, xM, m;
    double W;
    int device;
    int log_yN, log_xM, log_m;  // -1 when not a power of two
    float* invp_f = nullptr;    // 1/pswf[k] (k = 0 -> 0)
    double* invp_d = nullptr;
    float* fn_f = nullptr;  // Fn[k], k < m
    double* fn_d = nullptr;
    std::map<int, cx<float>*> tw_f;  // by log2(length)
    std::map<int, cx<double>*> tw_d;
    std::vector<void*> allocs;
};

template <typename R>
static const cx<R>* twiddles(const swiftly_hip* h, int logn);
template <>
const cx<float>* twiddles<float>(const swiftly_hip* h, int logn) {
    auto it = h->tw_f.find(logn);
    return it == h->tw_f.end() ? nullptr : it->second;
}
template <>
const cx<double>* twiddles<double>(const swiftly_hip* h, int logn) {
    auto it = h->tw_d.find(logn);
    return it == h->tw_d.end() ? nullptr : it->second;
}
template <typename R>
static const R* invp(const swiftly_hip* h);
template <>
const float* invp<float>(const swiftly_hip* h) { return h->invp_f; }
template <>
const double* invp<double>(const swiftly_hip* h) { return h->invp_d; }
template <typename R>
static const R* fnwin(const swiftly_hip* h);
template <>
const float* fnwin<float>(const swiftly_hip* h) { return h->fn_f; }
template <>
const double* fnwin<double>(const swiftly_hip* h) { return h->fn_d; }

template <typename T>
static int upload(swiftly_hip* h, T** dst, const std::vector<T>& v) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, v.size() * sizeof(T)));
    h->allocs.push_back(p);
    HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *dst = (T*)p;
    return 0;
}

static int make_twiddles(swiftly_hip* h, int logn) {
    if (logn < kMinLogN) return 0;
    const int n = 1 << logn;
    if (logn <= kMaxLogNFloat && !h->tw_f.count(logn)) {
        std::vector<cx<float>> t(n);
        for (int k = 0; k < n; k++) {
            // exact octant symmetry is not needed; evaluate in long double
            long double a = -2.0L * 3.14159265358979323846264338327950288L * k / n;
            t[k] = {(float)cosl(a), (float)sinl(a)};
        }
        cx<float>* d;
        if (int rc = upload(h, &d, t)) return rc;
        h->tw_f[logn] = d;
    }
    if (logn <= kMaxLogNDouble && !h->tw_d.count(logn)) {
        std::vector<cx<double>> t(n);
        for (int k = 0; k < n; k++) {
            long double a = -2.0L * 3.14159265358979323846264338327950288L * k / n;
            t[k] = {(double)cosl(a), (double)sinl(a)};
        }
        cx<double>* d;
        if (int rc = upload(h, &d, t)) return rc;
        h->tw_d[logn] = d;
    }
    return 0;
}

static bool g_inited = false;

extern "C" {

const char* swiftly_hip_last_error(void) { return g_err.c_str(); }
int swiftly_hip_version(void) { return 100; }
int swiftly_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int swiftly_hip_create(swiftly_hip_t** out, int64_t N, int64_t yN, int64_t xM, double W, const double* pswf,
                       int device) {
    if (!out || !pswf) return fail(SWIFTLY_ERR_PARAM, "null argument");
    *out = nullptr;
    // parameter checks of core.py:55-74
    if (N <= 0 || yN <= 0 || xM <= 0) return fail(SWIFTLY_ERR_PARAM, "sizes must be positive");
    if (N % yN != 0) return fail(SWIFTLY_ERR_PARAM, "Image size %lld not divisible by facet size %lld!", (long long)N, (long long)yN);
    if (N % xM != 0) return fail(SWIFTLY_ERR_PARAM, "Image size %lld not divisible by subgrid size %lld!", (long long)N, (long long)xM);
    if ((xM * yN) % N != 0)
        return fail(SWIFTLY_ERR_PARAM, "Contribution size not integer with image size %lld, subgrid size %lld and facet size %lld!",
                    (long long)N, (long long)xM, (long long)yN);
    int ndev = swiftly_hip_device_count();
    if (ndev <= 0) return fail(SWIFTLY_ERR_HIP, "no HIP device visible: the SwiFTly HIP backend has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(SWIFTLY_ERR_PARAM, "invalid device %d", device);
    HIP_TRY(hipSetDevice(device));
    if (!g_inited) {
        if (int rc = init_fft_rows_f32()) return fail(SWIFTLY_ERR_HIP, "kernel attribute setup failed (f32): %d", rc);
        if (int rc = init_fft_rows_f64()) return fail(SWIFTLY_ERR_HIP, "kernel attribute setup failed (f64): %d", rc);
        g_inited = true;
    }
    swiftly_hip* h = new (std::nothrow) swiftly_hip();
    if (!h) return fail(SWIFTLY_ERR_HIP, "out of host memory");
    h->N = N;
    h->yN = yN;
    h->xM = xM;
    h->m = xM * yN / N;
    h->W = W;
    h->device = device;
    h->log_yN = ilog2_exact(yN);
    h->log_xM = ilog2_exact(xM);
    h->log_m = ilog2_exact(h->m);
    // windows: 1/pswf (Fb, core.py:104-108) and Fn (core.py:110-117)
    std::vector<double> ip(yN);
    std::vector<float> ipf(yN);
    for (int64_t k = 0; k < yN; k++) {
        ip[k] = (k == 0 || pswf[k] == 0.0) ? 0.0 : 1.0 / pswf[k];
        ipf[k] = (float)ip[k];
    }
    const int64_t step = N / xM;
    std::vector<double> fn;
    for (int64_t k = (yN / 2) % step; k < yN; k += step) fn.push_back(pswf[k]);
    if ((int64_t)fn.size() != h->m) {
        delete h;
        return fail(SWIFTLY_ERR_PARAM, "internal: Fn length %zu != contribution size %lld", fn.size(), (long long)(xM * yN / N));
    }
    std::vector<float> fnf(fn.begin(), fn.end());
    int rc = 0;
    if (!rc) rc = upload(h, &h->invp_d, ip);
    if (!rc) rc = upload(h, &h->invp_f, ipf);
    if (!rc) rc = upload(h, &h->fn_d, fn);
    if (!rc) rc = upload(h, &h->fn_f, fnf);
    for (int l : {h->log_yN, h->log_xM, h->log_m})
        if (!rc && l >= 0) rc = make_twiddles(h, l);
    if (rc) {
        swiftly_hip_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

void swiftly_hip_destroy(swiftly_hip_t* h) {
    if (!h) return;
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
}

int64_t swiftly_hip_contribution_size(const swiftly_hip_t* h) { return h ? h->m : -1; }

}  // extern "C"

// ---------------------------------------------------------------------------
// modular gather / scatter-add (extract_from_facet / add_to_facet)
//   j < m ; i = (j - s) mod m ; big = (base + i + s) mod yN
//   gather : out[row, j]   = in[row, big]
//   scatter: out[row, big] += in[row, j]
template <typename R, bool SCATTER>
__global__ void modcopy_kernel(const cx<R>* __restrict__ in, cx<R>* __restrict__ out, long long rows, int m, int yN,
                               int s_m, int base_s, long long in_rs, long long in_cs, long long out_rs,
                               long long out_cs, int rowfast) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rows * m) return;
    long long row;
    int j;
    if (rowfast) {
        row = gid % rows;
        j = (int)(gid / rows);
    } else {
        j = (int)(gid % m);
        row = gid / m;
    }
    int i = j - s_m;
    if (i < 0) i += m;
    int big = base_s + i;  // base_s = (yN/2 - m/2 + s) mod yN
    if (big >= yN) big -= yN;
    if (SCATTER) {
        cx<R> v = in[row * in_rs + (long long)j * in_cs];
        cx<R>* p = out + row * out_rs + (long long)big * out_cs;
        cx<R> o = *p;
        o.x += v.x;
        o.y += v.y;
        *p = o;
    } else {
        out[row * out_rs + (long long)j * out_cs] = in[row * in_rs + (long long)big * in_cs];
    }
}

template <typename R, bool SCATTER>
static int run_modcopy(swiftly_hip* h, const void* in, int64_t rows, int64_t in_rs, int64_t in_cs, void* out,
                       int64_t out_rs, int64_t out_cs, int64_t subgrid_off, hipStream_t st) {
    if (rows <= 0) return 0;
    const int64_t s = floordiv(subgrid_off * h->yN, h->N);
    const int m = (int)h->m, yN = (int)h->yN;
    const int s_m = pmod(s, m);
    const int base_s = pmod(yN / 2 - m / 2 + s, yN);
    const long long total = rows * (long long)m;
    const int rowfast = (in_rs == 1 && in_cs != 1) ? 1 : 0;
    dim3 grid((unsigned)((total + 255) / 256));
    hipLaunchKernelGGL((modcopy_kernel<R, SCATTER>), grid, dim3(256), 0, st, (const cx<R>*)in, (cx<R>*)out,
                       (long long)rows, m, yN, s_m, base_s, (long long)in_rs, (long long)in_cs, (long long)out_rs,
                       (long long)out_cs, rowfast);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
template <typename R>
static AxisMap<R> identity_map(int n) {
    return AxisMap<R>{0, n, 0, n, nullptr, nullptr};
}

template <typename R>
static int run_rows(swiftly_hip* h, int logn, RowsArgs<R>& a, hipStream_t st) {
    constexpr int maxlog = sizeof(R) == 8 ? kMaxLogNDouble : kMaxLogNFloat;
    if (logn < kMinLogN || logn > maxlog)
        return fail(SWIFTLY_ERR_UNSUPPORTED,
                    "transform length %s is not supported by the HIP backend (power of two in [8, %d] required for %s)",
                    logn < 0 ? "(not a power of two)" : std::to_string(1 << logn).c_str(), 1 << maxlog,
                    sizeof(R) == 8 ? "complex128" : "complex64");
    a.tw = twiddles<R>(h, logn);
    if (!a.tw) return fail(SWIFTLY_ERR_HIP, "internal: missing twiddle table for 2^%d", logn);
    const uint64_t n = uint64_t(1) << logn;
    if (n * (uint64_t)a.in_cs >= (uint64_t(1) << 32) || n * (uint64_t)a.out_cs >= (uint64_t(1) << 32))
        return fail(SWIFTLY_ERR_PARAM, "transform length * column stride must be < 2^32");
    a.full_logn = logn;
    a.ld_mul = a.st_mul = 1;
    a.ld_addmul = a.st_addmul = 0;
    a.outer = 1;
    a.in_os = a.out_os = 0;
    a.tw_full = nullptr;
    a.tw_on_store = 0;
    a.raw_ld = a.raw_st = 0;
    if (a.nrows <= 0) return 0;
    int rc = launch_fft_rows(logn, a, st);
    if (rc) return fail(SWIFTLY_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return 0;
}

#define CHECK_COMMON()                                                                       \
    if (!h || !in || !out) return fail(SWIFTLY_ERR_PARAM, "null argument");                  \
    if (rows < 0) return fail(SWIFTLY_ERR_PARAM, "negative row count");                      \
    if (dtype != SWIFTLY_C64 && dtype != SWIFTLY_C128) return fail(SWIFTLY_ERR_PARAM, "bad dtype %d", dtype); \
    if (in_cs < 0 || out_cs < 0 || in_cs >= (int64_t(1) << 32) || out_cs >= (int64_t(1) << 32)) \
        return fail(SWIFTLY_ERR_PARAM, "column strides must be in [0, 2^32)");                 \
    if (rows > 0x7fffffff) return fail(SWIFTLY_ERR_PARAM, "too many rows");

template <typename R>
static void fill_io(RowsArgs<R>& a, const void* in, int64_t rows, int64_t in_rs, int64_t in_cs, void* out,
                    int64_t out_rs, int64_t out_cs) {
    std::memset(&a, 0, sizeof a);
    a.in = (const cx<R>*)in;
    a.out = (cx<R>*)out;
    a.in_rs = in_rs;
    a.in_cs = (unsigned)in_cs;
    a.out_rs = out_rs;
    a.out_cs = (unsigned)out_cs;
    a.nrows = (int)rows;
    a.scale = (R)1;
    // lanes run along rows when rows are the contiguous direction
    a.rowfast = (in_rs == 1 && in_cs != 1) ? 1 : 0;
}

template <typename R>
static int do_prepare_facet(swiftly_hip* h, const void* in, int64_t rows, int64_t yB, int64_t in_rs, int64_t in_cs,
                            void* out, int64_t out_rs, int64_t out_cs, int64_t off, hipStream_t st) {
    const int yN = (int)h->yN;
    RowsArgs<R> a;
    fill_io(a, in, rows, in_rs, in_cs, out, out_rs, out_cs);
    const int lo = yN / 2 - (int)(yB / 2);
    a.ld = AxisMap<R>{pmod(-(off + lo), yN), (int)yB, 0, (int)yB, invp<R>(h) + lo, nullptr};
    a.st = identity_map<R>(yN);
    a.conj_ld = a.conj_st = 1;
    a.scale = (R)(1.0 / yN);
    return run_rows(h, h->log_yN, a, st);
}

template <typename R>
static int do_add_to_subgrid(swiftly_hip* h, const void* in, int64_t rows, int64_t in_rs, int64_t in_cs, void* out,
                             int64_t out_rs, int64_t out_cs, int64_t off, hipStream_t st) {
    const int m = (int)h->m, xM = (int)h->xM;
    const int64_t sp = floordiv(off * h->xM, h->N);
    RowsArgs<R> a;
    fill_io(a, in, rows, in_rs, in_cs, out, out_rs, out_cs);
    a.ld = identity_map<R>(m);
    a.st = AxisMap<R>{pmod(-sp, m), m, pmod(xM / 2 - m / 2 + sp, xM), xM, fnwin<R>(h), nullptr};
    a.accumulate = 1;
    return run_rows(h, h->log_m, a, st);
}

template <typename R>
static int do_finish_subgrid(swiftly_hip* h, const void* in, int64_t rows, int64_t in_rs, int64_t in_cs, void* out,
                             int64_t out_rs, int64_t out_cs, int64_t off, int64_t xA, const void* mask,
                             hipStream_t st) {
    const int xM = (int)h->xM;
    RowsArgs<R> a;
    fill_io(a, in, rows, in_rs, in_cs, out, out_rs, out_cs);
    a.ld = identity_map<R>(xM);
    a.st = AxisMap<R>{pmod(-(xM / 2 - xA / 2 + off), xM), (int)xA, 0, (int)xA, (const R*)mask, nullptr};
    a.conj_ld = a.conj_st = 1;
    a.scale = (R)(1.0 / xM);
    return run_rows(h, h->log_xM, a, st);
}

template <typename R>
static int do_prepare_subgrid(swiftly_hip* h, const void* in, int64_t rows, int64_t xA, int64_t in_rs, int64_t in_cs,
                              void* out, int64_t out_rs, int64_t out_cs, int64_t off, hipStream_t st) {
    const int xM = (int)h->xM;
    RowsArgs<R> a;
    fill_io(a, in, rows, in_rs, in_cs, out, out_rs, out_cs);
    a.ld = AxisMap<R>{pmod(-(xM / 2 - xA / 2 + off), xM), (int)xA, 0, (int)xA, nullptr, nullptr};
    a.st = identity_map<R>(xM);
    return run_rows(h, h->log_xM, a, st);
}

template <typename R>
static int do_extract_from_subgrid(swiftly_hip* h, const void* in, int64_t rows, int64_t in_rs, int64_t in_cs,
                                   void* out, int64_t out_rs, int64_t out_cs, int64_t off, hipStream_t st) {
    const int m = (int)h->m, xM = (int)h->xM;
    const int64_t sp = floordiv(off * h->xM, h->N);
    RowsArgs<R> a;
    fill_io(a, in, rows, in_rs, in_cs, out, out_rs, out_cs);
    a.ld = AxisMap<R>{pmod(-sp, m), m, pmod(xM / 2 - m / 2 + sp, xM), xM, fnwin<R>(h), nullptr};
    a.st = identity_map<R>(m);
    a.conj_ld = a.conj_st = 1;
    a.scale = (R)(1.0 / m);
    return run_rows(h, h->log_m, a, st);
}

template <typename R>
static int do_finish_facet(swiftly_hip* h, const void* in, int64_t rows, int64_t in_rs, int64_t in_cs, void* out,
                           int64_t out_rs, int64_t out_cs, int64_t off, int64_t yB, const void* mask, hipStream_t st) {
    const int yN = (int)h->yN;
    const int lo = yN / 2 - (int)(yB / 2);
    RowsArgs<R> a;
    fill_io(a, in, rows, in_rs, in_cs, out, out_rs, out_cs);
    a.ld = identity_map<R>(yN);
    a.st = AxisMap<R>{pmod(-(lo + off), yN), (int)yB, 0, (int)yB, invp<R>(h) + lo, (const R*)mask};
    return run_rows(h, h->log_yN, a, st);
}

#define DISPATCH(fn, ...) (dtype == SWIFTLY_C64 ? fn<float>(__VA_ARGS__) : fn<double>(__VA_ARGS__))

extern "C" {

int swiftly_hip_prepare_facet(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t facet_size,
                              int64_t in_rs, int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs,
                              int64_t facet_off, void* stream) {
    CHECK_COMMON();
    if (facet_size <= 0 || facet_size >= h->yN)
        return fail(SWIFTLY_ERR_PARAM, "facet size %lld must be in [1, yN_size - 1 = %lld]", (long long)facet_size,
                    (long long)(h->yN - 1));
    return DISPATCH(do_prepare_facet, h, in, rows, facet_size, in_rs, in_cs, out, out_rs, out_cs, facet_off,
                    (hipStream_t)stream);
}

int swiftly_hip_extract_from_facet(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_rs,
                                   int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs, int64_t subgrid_off,
                                   void* stream) {
    CHECK_COMMON();
    if (dtype == SWIFTLY_C64)
        return run_modcopy<float, false>(h, in, rows, in_rs, in_cs, out, out_rs, out_cs, subgrid_off, (hipStream_t)stream);
    return run_modcopy<double, false>(h, in, rows, in_rs, in_cs, out, out_rs, out_cs, subgrid_off, (hipStream_t)stream);
}

int swiftly_hip_add_to_subgrid(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_rs,
                               int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs, int64_t facet_off,
                               void* stream) {
    CHECK_COMMON();
    return DISPATCH(do_add_to_subgrid, h, in, rows, in_rs, in_cs, out, out_rs, out_cs, facet_off, (hipStream_t)stream);
}

int swiftly_hip_finish_subgrid(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_rs,
                               int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs, int64_t subgrid_off,
                               int64_t subgrid_size, const void* mask, void* stream) {
    CHECK_COMMON();
    if (subgrid_size <= 0 || subgrid_size > h->xM)
        return fail(SWIFTLY_ERR_PARAM, "subgrid size %lld must be in [1, xM_size = %lld]", (long long)subgrid_size,
                    (long long)h->xM);
    return DISPATCH(do_finish_subgrid, h, in, rows, in_rs, in_cs, out, out_rs, out_cs, subgrid_off, subgrid_size, mask,
                    (hipStream_t)stream);
}

int swiftly_hip_prepare_subgrid(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t subgrid_size,
                                int64_t in_rs, int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs,
                                int64_t subgrid_off, void* stream) {
    CHECK_COMMON();
    if (subgrid_size <= 0 || subgrid_size > h->xM)
        return fail(SWIFTLY_ERR_PARAM, "subgrid size %lld must be in [1, xM_size = %lld]", (long long)subgrid_size,
                    (long long)h->xM);
    return DISPATCH(do_prepare_subgrid, h, in, rows, subgrid_size, in_rs, in_cs, out, out_rs, out_cs, subgrid_off,
                    (hipStream_t)stream);
}

int swiftly_hip_extract_from_subgrid(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_rs,
                                     int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs, int64_t facet_off,
                                     void* stream) {
    CHECK_COMMON();
    return DISPATCH(do_extract_from_subgrid, h, in, rows, in_rs, in_cs, out, out_rs, out_cs, facet_off,
                    (hipStream_t)stream);
}

int swiftly_hip_add_to_facet(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_rs, int64_t in_cs,
                             void* out, int64_t out_rs, int64_t out_cs, int64_t subgrid_off, void* stream) {
    CHECK_COMMON();
    if (dtype == SWIFTLY_C64)
        return run_modcopy<float, true>(h, in, rows, in_rs, in_cs, out, out_rs, out_cs, subgrid_off, (hipStream_t)stream);
    return run_modcopy<double, true>(h, in, rows, in_rs, in_cs, out, out_rs, out_cs, subgrid_off, (hipStream_t)stream);
}

int swiftly_hip_finish_facet(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_rs, int64_t in_cs,
                             void* out, int64_t out_rs, int64_t out_cs, int64_t facet_off, int64_t facet_size,
                             const void* mask, void* stream) {
    CHECK_COMMON();
    if (facet_size <= 0 || facet_size >= h->yN)
        return fail(SWIFTLY_ERR_PARAM, "facet size %lld must be in [1, yN_size - 1 = %lld]", (long long)facet_size,
                    (long long)(h->yN - 1));
    return DISPATCH(do_finish_facet, h, in, rows, in_rs, in_cs, out, out_rs, out_cs, facet_off, facet_size, mask,
                    (hipStream_t)stream);
}

int swiftly_hip_malloc(void** ptr, size_t bytes) {
    if (!ptr) return fail(SWIFTLY_ERR_PARAM, "null argument");
    HIP_TRY(hipMalloc(ptr, bytes));
    return 0;
}
int swiftly_hip_free(void* ptr) {
    HIP_TRY(hipFree(ptr));
    return 0;
}
int swiftly_hip_memset_async(void* ptr, int value, size_t bytes, void* stream) {
    HIP_TRY(hipMemsetAsync(ptr, value, bytes, (hipStream_t)stream));
    return 0;
}
int swiftly_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return 0;
}
int swiftly_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return 0;
}
int swiftly_hip_stream_synchronize(void* stream) {
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

}  // extern "C"
