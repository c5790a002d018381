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
#include "swiftly_abi_internal.h"

// ---------------------------------------------------------------------------
static thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

template <typename T>
static int upload(swiftly_hip* h, T** dst, const std::vector<T>& v) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, v.size() * sizeof(T)));
    h->allocs.push_back(p);
    HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *dst = (T*)p;
    return 0;
}

static std::vector<cx<float>> host_twiddles_f(int logn) {
    const int n = 1 << logn;
    std::vector<cx<float>> t(n);
    for (int k = 0; k < n; k++) {
        // exact octant symmetry is not needed; evaluate in long double
        long double a = -2.0L * 3.14159265358979323846264338327950288L * k / n;
        t[k] = {(float)cosl(a), (float)sinl(a)};
    }
    return t;
}
// compact copy of the float table of length 2^logn for kernels with 2^logp points per lane (swiftly_fft.h)
static int make_compact_twiddles(swiftly_hip* h, int logn, int logp) {
    if (logn < 2 || logp < 1 || logn > kMaxLogNFloat + 1 || h->twc_f.count({logn, logp})) return 0;
    const std::vector<cx<float>> t = host_twiddles_f(logn);
    std::vector<cx<float>> c((size_t)compact_tw_entries(logn, logp));
    build_compact_twiddles(t.data(), logn, logp, c.data());
    cx<float>* d;
    if (int rc = upload(h, &d, c)) return rc;
    h->twc_f[{logn, logp}] = d;
    return 0;
}

static int make_twiddles(swiftly_hip* h, int logn) {
    if (logn < kMinLogN) return 0;
    const int n = 1 << logn;
    if (logn <= kMaxLogNFloat + 1 && !h->tw_f.count(logn)) {  // 2^16: only the band row kernel and the column passes use it
        const std::vector<cx<float>> t = host_twiddles_f(logn);
        cx<float>* d;
        if (int rc = upload(h, &d, t)) return rc;
        h->tw_f[logn] = d;
    }
    // (double tables up to 2^16: the float64-arithmetic column passes need the four-step twiddles of the whole length)
    if (logn <= kMaxLogNFloat + 1 && !h->tw_d.count(logn)) {
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

// Bluestein tables for length n: chirp c[j] = exp(i pi j^2 / n) and the spectrum of the wrapped chirp filter of
// length L = 2^logL >= 2n - 1 (host radix-2 FFT in double precision; exact angles through j^2 mod 2n).
static int make_bluestein(swiftly_hip* h, int64_t n) {
    if (n <= 0 || ilog2_exact(n) >= 0 || h->blu.count(n)) return 0;
    int logL = 0;
    while ((int64_t(1) << logL) < 2 * n - 1) logL++;
    if (logL > kMaxLogNFloat + 1) return 0;  // no kernel for the convolution length: stays unsupported
    const int64_t L = int64_t(1) << logL;
    const long double pi = 3.14159265358979323846264338327950288L;
    std::vector<std::complex<double>> c(n), f(L, 0.0);
    for (int64_t j = 0; j < n; j++) {
        const long double ang = pi * (long double)((j * j) % (2 * n)) / (long double)n;
        c[j] = {(double)cosl(ang), (double)sinl(ang)};
    }
    f[0] = c[0];
    for (int64_t j = 1; j < n; j++) f[j] = f[L - j] = c[j];
    // iterative radix-2 decimation-in-time FFT (forward, exp(-2 pi i jk / L))
    for (int64_t i = 1, j = 0; i < L; i++) {
        int64_t bit = L >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(f[i], f[j]);
    }
    for (int64_t len = 2; len <= L; len <<= 1) {
        for (int64_t i = 0; i < L; i += len)
            for (int64_t k = 0; k < len / 2; k++) {
                const long double ang = -2.0L * pi * (long double)k / (long double)len;
                const std::complex<double> w((double)cosl(ang), (double)sinl(ang));
                const std::complex<double> u = f[i + k], v = f[i + k + len / 2] * w;
                f[i + k] = u + v;
                f[i + k + len / 2] = u - v;
            }
    }
    swiftly_hip::Blu t;
    t.logL = logL;
    std::vector<cx<float>> cf(n), sf(L);
    std::vector<cx<double>> cd(n), sd(L);
    for (int64_t j = 0; j < n; j++) {
        cd[j] = {c[j].real(), c[j].imag()};
        cf[j] = {(float)c[j].real(), (float)c[j].imag()};
    }
    for (int64_t j = 0; j < L; j++) {
        sd[j] = {f[j].real(), f[j].imag()};
        sf[j] = {(float)f[j].real(), (float)f[j].imag()};
    }
    int rc = upload(h, &t.chirp_f, cf);
    if (!rc) rc = upload(h, &t.spec_f, sf);
    if (!rc && logL <= kMaxLogNDouble) {
        rc = upload(h, &t.chirp_d, cd);
        if (!rc) rc = upload(h, &t.spec_d, sd);
    }
    if (!rc) rc = make_twiddles(h, logL);
    if (!rc && logL == 16) rc = make_twiddles(h, 15);
    if (!rc && logL == 15) {
        rc = make_twiddles(h, 14);
        if (!rc) rc = make_twiddles(h, 13);
    }
    if (!rc) h->blu[n] = t;
    return rc;
}

// n = Q * 2^k with Q in {3, 5, 7, 9} and 2^k a length the power-of-two kernels take: Q and k, else false
bool mixed_factor(int64_t n, int* Q, int* logM) {
    if (n <= 0) return false;
    for (int q : {3, 5, 7, 9}) {
        if (n % q) continue;
        const int l = ilog2_exact(n / q);
        if (l >= kMinLogN) {
            *Q = q;
            *logM = l;
            return true;
        }
    }
    return false;
}

// Tables for a length n = Q * 2^k (swiftly_mixed.h): the full-length twiddles and the power-of-two tables of the
// sub-transforms (with the halves their strided four-step form uses).
static int make_mixed(swiftly_hip* h, int64_t n) {
    int Q = 0, logM = 0;
    if (!mixed_factor(n, &Q, &logM) || h->mixed.count(n)) return 0;
    if (logM > kMaxLogNFloat) return 0;  // stays with the fallbacks
    swiftly_hip::Mixed t;
    t.Q = Q;
    t.logM = logM;
    std::vector<cx<float>> tf((size_t)n);
    std::vector<cx<double>> td((size_t)n);
    for (int64_t k = 0; k < n; k++) {
        const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)n;
        td[(size_t)k] = {(double)cosl(a), (double)sinl(a)};
        tf[(size_t)k] = {(float)cosl(a), (float)sinl(a)};
    }
    int rc = upload(h, &t.tw_f, tf);
    if (!rc && logM <= kMaxLogNDouble) rc = upload(h, &t.tw_d, td);
    if (!rc) rc = make_twiddles(h, logM);
    if (!rc && logM >= kTwoPassMinLog) rc = make_twiddles(h, logM / 2);
    if (!rc && logM >= kTwoPassMinLog) rc = make_twiddles(h, logM - logM / 2);
    if (!rc && logM == 15) rc = make_twiddles(h, 14);
    if (!rc && logM == 15) rc = make_twiddles(h, 13);
    if (!rc) h->mixed[n] = t;
    return rc;
}

// Kernel attributes (max dynamic LDS) and the memory-pool release threshold are per DEVICE state: they are set
// once for every device a handle is created on, under a lock (handles may be created from several host threads).
static std::mutex g_init_mutex;
static std::vector<char> g_device_inited;


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
    DeviceGuard guard(device);
    if (guard.rc) return fail(SWIFTLY_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString((hipError_t)guard.rc));
    std::lock_guard<std::mutex> init_lock(g_init_mutex);
    if ((int)g_device_inited.size() < ndev) g_device_inited.resize(ndev, 0);
    if (!g_device_inited[device]) {
        if (int rc = init_fft_rows_f32()) return fail(SWIFTLY_ERR_HIP, "kernel attribute setup failed (f32): %d", rc);
        if (int rc = init_fft_rows_f64()) return fail(SWIFTLY_ERR_HIP, "kernel attribute setup failed (f64): %d", rc);
        if (int rc = init_col_pass()) return fail(SWIFTLY_ERR_HIP, "kernel attribute setup failed (col pass): %d", rc);
        if (int rc = init_row_pass()) return fail(SWIFTLY_ERR_HIP, "kernel attribute setup failed (row pass): %d", rc);
        if (int rc = init_sum_finish_rows()) return fail(SWIFTLY_ERR_HIP, "kernel attribute setup failed (sum finish): %d", rc);
        // keep freed scratch (the four-step intermediate, up to yN*yB*8 bytes) in the stream-ordered pool instead of
        // returning it to the driver at every synchronisation point
        hipMemPool_t pool;
        if (hipDeviceGetDefaultMemPool(&pool, device) == hipSuccess) {
            uint64_t keep = UINT64_MAX;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        }
        (void)hipGetLastError();
        g_device_inited[device] = 1;
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
    if (const char* e = getenv("SWIFTLY_COL_F64")) h->col_f64 = atoi(e) != 0;
    if (const char* e = getenv("SWIFTLY_COL_F64_STAGES")) h->col_f64_stages = atoi(e) & 7;
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
        if (!rc && l >= 0) {
            rc = make_twiddles(h, l);
            if (!rc && l >= kTwoPassMinLog) rc = make_twiddles(h, l / 2);
            if (!rc && l >= kTwoPassMinLog) rc = make_twiddles(h, l - l / 2);
            if (!rc && l == 15) rc = make_twiddles(h, 14);  // multi-workgroup row kernels
            if (!rc && l == 15) rc = make_twiddles(h, 13);
            if (!rc && (l == 16 || l == 14)) rc = make_twiddles(h, l - 1);  // band row kernel halves
        }
    // compact copies for the contiguous-axis kernels whose lanes gather table values (swiftly_fft.h): the forward K1 / backward
    // finish of 32768-point rows (2 x 16384 points, 32 per lane) and the facet kernels of the subgrid side (64 lanes per transform)
    if (!rc && h->log_yN == 15) {
        rc = make_compact_twiddles(h, 14, 5);
        if (!rc) h->win4.twc = compact_twiddles(h, 14, 5);
    }
    if (!rc && sum_finish_supported(h->log_m, h->log_xM)) {
        rc = make_compact_twiddles(h, h->log_m, h->log_m - 6);
        if (!rc) rc = make_compact_twiddles(h, h->log_xM, h->log_xM - (h->log_xM >= 12 ? 8 : 6));
    }
    for (int64_t len : {yN, xM, h->m})
        if (!rc) rc = make_bluestein(h, len);
    for (int64_t len : {yN, xM, h->m})
        if (!rc) rc = make_mixed(h, len);
    if (rc) {
        swiftly_hip_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

void swiftly_hip_destroy(swiftly_hip_t* h) {
    if (!h) return;
    DeviceGuard guard(h->device);
    for (hipStream_t st : h->chunk_st)
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    for (void* p : h->allocs) (void)hipFree(p);
    h->win4.clear();
    delete h;
}


int64_t swiftly_hip_contribution_size(const swiftly_hip_t* h) { return h ? h->m : -1; }

}  // extern "C"

// ---------------------------------------------------------------------------
// Batch descriptor shared by all *_batch entry points: nbatch independent
// problems of identical shape at in + b*in_bs / out + b*out_bs.  `offs`
// (host array, may be null) gives a per-item offset, otherwise `off` applies
// to every item.
struct Batch {
    int64_t n = 1, in_bs = 0, out_bs = 0;
    const int64_t* offs = nullptr;
    int64_t mask_bs = 0;
    int64_t off_of(int64_t b, int64_t off) const { return offs ? offs[b] : off; }
};

// modular gather / scatter-add (extract_from_facet / add_to_facet)
//   j < m ; i = (j - s) mod m ; big = (base + i + s) mod yN
//   gather : out[row, j]   = in[row, big]
//   scatter: out[row, big] += in[row, j]
struct ModTab {
    int s_m[kMaxBatch], base_s[kMaxBatch];
};

template <typename R, bool SCATTER>
__global__ void modcopy_kernel(const cx<R>* __restrict__ in, cx<R>* __restrict__ out, long long rows, int m, int yN,
                               const ModTab tab, long long in_rs, long long in_cs, long long out_rs,
                               long long out_cs, long long in_bs, long long out_bs, int rowfast) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rows * m) return;
    const int b = blockIdx.y;
    in += (long long)b * in_bs;
    out += (long long)b * out_bs;
    long long row;
    int j;
    if (rowfast) {
        row = gid % rows;
        j = (int)(gid / rows);
    } else {
        j = (int)(gid % m);
        row = gid / m;
    }
    int i = j - tab.s_m[b];
    if (i < 0) i += m;
    int big = tab.base_s[b] + i;  // base_s = (yN/2 - m/2 + s) mod yN
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
                       int64_t out_rs, int64_t out_cs, int64_t subgrid_off, const Batch& bt, hipStream_t st) {
    if (rows <= 0 || bt.n <= 0) return 0;
    const int m = (int)h->m, yN = (int)h->yN;
    const long long total = rows * (long long)m;
    const int rowfast = (in_rs == 1 && in_cs != 1) ? 1 : 0;
    for (int64_t b0 = 0; b0 < bt.n; b0 += kMaxBatch) {
        const int nb = (int)std::min<int64_t>(kMaxBatch, bt.n - b0);
        ModTab tab;
        for (int b = 0; b < nb; b++) {
            const int64_t s = floordiv(bt.off_of(b0 + b, subgrid_off) * h->yN, h->N);
            tab.s_m[b] = pmod(s, m);
            tab.base_s[b] = pmod(yN / 2 - m / 2 + s, yN);
        }
        dim3 grid((unsigned)((total + 255) / 256), nb);
        hipLaunchKernelGGL((modcopy_kernel<R, SCATTER>), grid, dim3(256), 0, st,
                           (const cx<R>*)in + b0 * bt.in_bs, (cx<R>*)out + b0 * bt.out_bs, (long long)rows, m, yN, tab,
                           (long long)in_rs, (long long)in_cs, (long long)out_rs, (long long)out_cs,
                           (long long)bt.in_bs, (long long)bt.out_bs, rowfast);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// ---------------------------------------------------------------------------
template <typename R>
static AxisMap<R> identity_map(int n) {
    return AxisMap<R>{0, n, 0, n, nullptr, nullptr};
}

template <typename R>
static int launch_checked(int logn, const RowsArgs<R>& a, const OffTab& tab, hipStream_t st) {
    int rc = launch_fft_rows(logn, a, tab, st);
    if (rc) return fail(SWIFTLY_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return 0;
}

ColZ plain_colz() {
    ColZ z;
    std::memset(&z, 0, sizeof z);
    z.nb = 1;
    return z;
}

int launch_col_checked(int lg, int mode, const ColPassArgs& args, const ColZ& cz, int outer, int nb, hipStream_t st) {
    int e = launch_col_pass(lg, mode, args, cz, outer, nb, st);
    if (e) return fail(SWIFTLY_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    return 0;
}

// Strided-axis transform of length 2^logn over `W` adjacent columns and `nb` batch items with the column-tile
// passes: one pass up to 512 points, otherwise four-step (N = n1*n2, input index y = y1*n2 + y2, output index
// k = k1 + n1*k2) through a stream-ordered scratch [nb][N][W]:
//   pass A (length n1 over y1, one per y2): scratch[k1*n2 + y2] = W_N^(y2 k1) * sum_y1 x[y1 n2 + y2] W_n1^(y1 k1)
//   pass B (length n2 over y2, one per k1): X[k1 + n1 k2]       = sum_y2 scratch[k1*n2 + y2] W_n2^(y2 k2)
// `c` carries the load / store maps, windows, conjugation flags, scale, batch strides, column gather and row
// maps of the whole transform; in/out pitches and pointers are given separately.  Returns -1 when the length
// is outside the column-pass range (caller falls back), else a status code.
// Scratch handed down by an entry point for the duration of one ABI call on this host thread (the batch entry
// points have no workspace parameter): col_transform prefers it to a stream-ordered allocation.  Measured on
// MI355X: hipMallocAsync with a size that changes from call to call costs ~2 ms of HOST time per call (the pool
// does not reuse a smaller free block), which made a 25-wave pass host-bound.
thread_local void* t_call_ws = nullptr;
thread_local size_t t_call_ws_bytes = 0;

// Sub-transform form (qmul = Q > 0, swiftly_mixed.h): the input is the plain scratch of the radix-Q pass (element y of
// the length-2^logn sub-transform j = qadd at row y of `c.in`), the store map of `c` refers to the full length full_n
// with plain output index Q*k + j.
int col_transform(swiftly_hip* h, int logn, const ColPassArgs& c, const ColZ& cz, int W, int nb, hipStream_t st, void* ws,
                  size_t ws_bytes, int qmul, int qadd, int full_n) {
    if (!ws && t_call_ws) {
        ws = t_call_ws;
        ws_bytes = t_call_ws_bytes;
    }
    // single pass up to 1024 points; the gather-sum load (c.gs, backward pass) exists for 64-column tiles only, i.e. up to
    // 512 points: longer gather-sum transforms go through the four-step, whose pass A carries the load (r3 bug: a
    // 1024-point gather-sum transform ran the plain 32-column kernel, which read the encoded table as a row map)
    const bool two = logn > (c.gs ? 9 : kColPassMaxLog);
    const int l1 = two ? logn / 2 : logn, l2 = logn - l1;  // (32768 = 128 x 256; 256 x 128 and 64 x 512 measured slower, r4)
    if (l1 < kColPassMinLog || l1 > kColPassMaxLog || (two && (l2 < kColPassMinLog || l2 > kColPassMaxLog))) return -1;
    const uint64_t n = uint64_t(1) << logn;
    // float64 arithmetic where the caller asks for it and the instances exist (else float32, silently: same results to
    // float32 rounding)
    const bool f64 = c.f64 && (two ? (col_pass_f64_supported(l1) && col_pass_f64_supported(l2))
                                   : (col_pass_f64_supported(logn) && !(c.gs && logn > 8)));
    if (!two) {
        ColPassArgs one = c;
        one.tw = twiddles<float>(h, logn);
        if (!one.tw) return -1;
        one.f64 = (f64 && (h->col_f64_stages & 4)) ? 1 : 0;
        one.twd = f64 ? twiddles<double>(h, logn) : nullptr;
        one.twd_full = one.twd;
        if (f64 && !one.twd) return -1;
        if (qmul > 0) {
            one.full_logn = logn; one.full_n = full_n; one.ld_plain = 1; one.st_qmul = qmul; one.st_qadd = qadd;
            one.ld_mul = one.st_mul = 1;
        }
        return launch_col_checked(logn, 2, one, cz, 1, nb, st);
    }
    const int n1 = 1 << l1, n2 = 1 << l2;
    const cx<float>* tw1 = twiddles<float>(h, l1);
    const cx<float>* tw2 = twiddles<float>(h, l2);
    const cx<float>* twf = twiddles<float>(h, logn);
    if (!tw1 || !tw2 || !twf) return -1;
    const cx<double>* twd1 = f64 ? twiddles<double>(h, l1) : nullptr;
    const cx<double>* twd2 = f64 ? twiddles<double>(h, l2) : nullptr;
    const cx<double>* twdf = f64 ? twiddles<double>(h, logn) : nullptr;
    if (f64 && (!twd1 || !twd2 || !twdf)) return -1;
    if (n * (uint64_t)W >= (uint64_t(1) << 32)) return -1;
    const bool gathered = (cz.flags & kZColGather) != 0;
    const long long Ws = (long long)W;  // scratch row width (column slabs that keep the intermediate cache-sized: no gain, r2-r4)
    const size_t scratch_bytes = (size_t)nb * n * (size_t)Ws * sizeof(cx<float>);
    void* scratch = nullptr;
    hipError_t he = hipSuccess;
    // caller-provided workspace (deterministic; the stream-ordered pool reuses memory across STREAMS only
    // opportunistically, which made the two-stream schedule fall back to fresh multi-GB allocations on some runs)
    const bool own = !(ws && ws_bytes >= scratch_bytes);
    if (own) {
        he = hipMallocAsync(&scratch, scratch_bytes, st);
        if (he != hipSuccess) return fail(SWIFTLY_ERR_HIP, "hipMallocAsync(two-pass scratch): %s", hipGetErrorString(he));
    } else {
        scratch = ws;
    }
    int rc = 0;
    // Layout of the intermediate (r4): row y2 * n1 + k1 -- a pass-A workgroup (one y2) WRITES n1 consecutive rows and a
    // pass-B workgroup (one k1) reads a comb -- instead of row k1 * n2 + y2 (comb written, consecutive rows read).  HBM
    // writes are the expensive direction on this chip (tools/mall_pipe.hip: a comb costs 7 % on the write side and nothing
    // on the read side): pass A 425 -> 395 us per wave as a pure copy, 427 -> 386 us for the kernel.
    constexpr bool y2_major = true;
    // (r5: a TILE-major scratch -- [item][64-column tile][row][64], the n1 rows of a pass-A workgroup one contiguous run
    // of n1 * 512 bytes -- measured the same within the run-to-run spread: 38.87 / 39.64 / 39.33 against 39.29 / 38.75 /
    // 38.58 ms per pass, interleaved on one box; not kept)
    const int a_i_rows = y2_major ? 1 : n2, a_o_rows = y2_major ? n1 : 1;  // pass A: row of (e = k1, o = y2)
    const int b_i_rows = y2_major ? n1 : 1, b_o_rows = y2_major ? 1 : n2;  // pass B: row of (i = y2, o = k1)
    // Chunked, two-stream form (r4; K2 = the gathered forward transform of several facets): the batch items are worked
    // on in chunks of `zc` items x `Wc` columns whose two passes run back to back, chunks alternating between two
    // internal streams, each stream re-using ONE chunk-sized slot of the scratch: pass A of one chunk (HBM reads, scratch
    // writes) runs next to pass B of the other (scratch reads that can still hit the 256 MiB Infinity Cache).  Measured on
    // the 64k workload (interleaved repeats on one box, gpurun_out/s3k, s3l): 40.7 -> 39.7 ms per pass with chunks of
    // 2 facets x 256 columns (134 MB); 1 x 512: 40.0; 1 x 256, 4 x 128, 2 x 128, 3 x 256, 2 x 512: no gain or worse.  The
    // gain is the overlap of the two kinds of pass, not cache residency: pure-copy stand-ins of the two passes bound it at
    // 8 % of K2 (tools/mall_pipe.hip) -- the cache does not absorb the scratch WRITES.
    // SWIFTLY_K2_CHUNK = "cols[,items]" | 0 (off) | unset: chunks of ~128 MB when the whole intermediate exceeds 256 MB.
    static const char* chunk_env = getenv("SWIFTLY_K2_CHUNK");
    int chunk_cols = chunk_env ? atoi(chunk_env) : -1;
    int chunk_items = (chunk_env && strchr(chunk_env, ',')) ? std::max(1, atoi(strchr(chunk_env, ',') + 1)) : 1;
    if (chunk_cols < 0) {  // automatic
        chunk_cols = 0;
        if (scratch_bytes > (size_t(256) << 20) && W >= 256) {
            chunk_cols = 256;
            chunk_items = (int)std::max<uint64_t>(1, (uint64_t(128) << 20) / (n * 256 * sizeof(cx<float>)));
        }
    }
    // (r5: three or four chunk streams instead of two -- 41.4-42.1 ms per pass against 38.9-39.8 with the default chunks,
    // 39.5-39.7 against 39.1-39.3 with chunks of 1 facet x 256 columns; one stream: 41.7 -- two streams stay)
    if (chunk_cols >= 64 && gathered && !(cz.flags & kZColScatter) && !c.gs && qmul == 0 &&
        (nb > chunk_items || W > chunk_cols) && !own) {
        const int Wc = std::min<int>((chunk_cols / 64) * 64, W), zc = std::min(chunk_items, nb);
        const size_t slot_elems = (size_t)n * (size_t)Wc * (size_t)zc;
        if (2 * slot_elems * sizeof(cx<float>) <= ws_bytes) {
            {
                std::lock_guard<std::mutex> lock(h->chunk_mu);
                for (hipStream_t& s2 : h->chunk_st)
                    if (!s2) {
                        he = hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
                        if (he != hipSuccess) return fail(SWIFTLY_ERR_HIP, "hipStreamCreateWithFlags: %s", hipGetErrorString(he));
                    }
            }
            // (events are per call: two host threads may drive the same handle on different streams)
            hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
            auto drop_events = [&ev]() {
                for (hipEvent_t& e : ev)
                    if (e) {
                        (void)hipEventDestroy(e);
                        e = nullptr;
                    }
            };
            for (hipEvent_t& e : ev) {
                he = hipEventCreateWithFlags(&e, hipEventDisableTiming);
                if (he != hipSuccess) {
                    e = nullptr;
                    drop_events();
                    return fail(SWIFTLY_ERR_HIP, "hipEventCreateWithFlags: %s", hipGetErrorString(he));
                }
            }
            // fork: the chunk streams start behind everything queued on `st` (a failed record / wait would let them
            // run ahead of the producer of the input: nothing has been launched yet, so just report it)
            // (swiftly_hip_chain_chunk_streams: the caller vouches for the inputs; the chunk streams run on from the chunks
            // of the previous call -- no pipeline drain and no idle event hops between consecutive waves, r5)
            he = hipSuccess;
            // (a chained call still forks when the un-chunked path has used THIS workspace since the last fork)
            void* plain_ws = scratch;
            const bool plain_pending = h->ws_plain_pending.compare_exchange_strong(plain_ws, nullptr);
            if (!g_chain_chunk_streams || plain_pending) {
                he = hipEventRecord(ev[0], st);
                for (hipStream_t s2 : h->chunk_st)
                    if (he == hipSuccess) he = hipStreamWaitEvent(s2, ev[0], 0);
            }
            if (he != hipSuccess) {
                drop_events();
                return fail(SWIFTLY_ERR_HIP, "chunked four-step, fork: %s", hipGetErrorString(he));
            }
            int i = 0;
            for (int z0 = 0; z0 < nb && !rc; z0 += zc) {
                const int nz = std::min(zc, nb - z0);
                for (int c0 = 0; c0 < W && !rc; c0 += Wc, i++) {
                    const int wc = std::min(Wc, W - c0);
                    hipStream_t s2 = h->chunk_st[i & 1];
                    // item z of the launch sits at slot + (z - z0) * n * Wc (raw_z0: the kernels address the scratch
                    // with the item index relative to the launch's first item)
                    cx<float>* slot = (cx<float>*)scratch + (size_t)(i & 1) * slot_elems;
                    ColPassArgs A = c;
                    A.scratch_nt = 0;
                    A.ncols = wc; A.col0 = c0; A.z0 = z0; A.raw_z0 = z0;
                    A.out = slot; A.out_pitch = (unsigned)Wc; A.out_bs = (long long)(n * Wc);
                    A.out_bdiv = 0; A.out_bs_hi = 0;
                    A.ld_mul = n2;
                    A.out_i_rows = a_i_rows; A.out_o_rows = a_o_rows;
                    A.tw = tw1; A.tw_full = twf;
                    A.f64 = (f64 && (h->col_f64_stages & 1)) ? 1 : 0; A.twd = twd1; A.twd_full = twdf;
                    A.conj_st = 0; A.accumulate = 0; A.scale = 1.f;
                    A.col_win = nullptr; A.st_rowmap = nullptr; A.st_win = nullptr; A.st_win2 = nullptr;
                    rc = launch_col_checked(l1, 0, A, cz, n2, nz, s2);
                    if (rc) break;
                    ColPassArgs B = c;
                    B.scratch_nt = 0;
                    ColZ zb = cz;
                    zb.flags &= ~(kZColGather | kZLoadB | kZLoadAF);  // the scratch is read plainly
                    B.ncols = wc; B.col0 = 0; B.z0 = z0; B.raw_z0 = z0;
                    B.in = slot; B.in_pitch = (unsigned)Wc; B.in_bs = (long long)(n * Wc);
                    B.in_bdiv = 0; B.in_bs_hi = 0;
                    B.in_i_rows = b_i_rows; B.in_o_rows = b_o_rows;
                    B.ld_rowmap = nullptr; B.ld_win = nullptr; B.ld_win2 = nullptr; B.gs = 0;
                    B.out = c.out + c0;
                    B.st_mul = n1;
                    B.tw = tw2; B.tw_full = twf;
                    B.f64 = (f64 && (h->col_f64_stages & 2)) ? 1 : 0; B.twd = twd2; B.twd_full = twdf;
                    B.conj_ld = 0;
                    if (B.col_win) B.col_win += c0;
                    rc = launch_col_checked(l2, 1, B, zb, n1, nz, s2);
                }
            }
            // join: `st` continues behind both chunk streams.  If the join cannot be queued, the consumer of the output
            // on `st` must not start early: wait for the chunk streams on the host instead
            for (int k = 0; k < 2; k++) {
                he = hipEventRecord(ev[1 + k], h->chunk_st[k]);
                if (he == hipSuccess) he = hipStreamWaitEvent(st, ev[1 + k], 0);
                if (he != hipSuccess) {
                    (void)hipStreamSynchronize(h->chunk_st[k]);
                    if (!rc) rc = fail(SWIFTLY_ERR_HIP, "chunked four-step, join: %s", hipGetErrorString(he));
                }
            }
            drop_events();
            return rc;
        }
    }
    // scratch accesses: a small intermediate is left cacheable so that pass B finds it in the 256 MiB Infinity
    // Cache (measured: the 160 MB of a K5b wave, K3-5 12.5 -> 11.6 ms per pass); a large one is streamed
    // non-temporally (measured: K2, 1.2 GB per wave, 18.5 ms vs 19.6 ms cacheable).
    const int scratch_nt = scratch_bytes > (size_t(192) << 20) ? 1 : 0;
    if (!own) h->ws_plain_pending.store(scratch);  // `ws` is written on `st` below: a later chained chunked call has to wait for it
    for (long long c0 = 0; c0 < (long long)W && !rc; c0 += Ws) {
        const int wc = (int)std::min<long long>(Ws, (long long)W - c0);
        // pass A: length n1 over y1 (input index y1*n2 + y2), outer = y2; scratch row k1*n2 + y2
        ColPassArgs A = c;
        A.scratch_nt = scratch_nt;
        A.ncols = wc;
        A.in = c.in + c0;
        A.out = (cx<float>*)scratch; A.out_pitch = (unsigned)Ws; A.out_bs = (long long)(n * Ws);
        A.out_bdiv = 0; A.out_bs_hi = 0;
        A.ld_mul = n2;
        A.out_i_rows = a_i_rows; A.out_o_rows = a_o_rows;
        A.tw = tw1; A.tw_full = twf;
        A.f64 = (f64 && (h->col_f64_stages & 1)) ? 1 : 0; A.twd = twd1; A.twd_full = twdf;
        if (qmul > 0) {
            A.full_logn = logn; A.full_n = 0; A.ld_plain = 1; A.st_qmul = 0;
        }
        A.conj_st = 0; A.accumulate = 0; A.scale = 1.f;
        A.col_win = nullptr; A.st_rowmap = nullptr; A.st_win = nullptr; A.st_win2 = nullptr;
        ColZ za = cz;
        za.flags &= ~kZColScatter;  // the scratch is written plainly
        rc = launch_col_checked(l1, 0, A, za, n2, nb, st);
        if (rc) break;
        // pass B: length n2 over y2, outer = k1; output index k1 + n1*k2
        ColPassArgs B = c;
        B.scratch_nt = scratch_nt;
        ColZ zb = cz;
        zb.flags &= ~(kZColGather | kZLoadB | kZLoadAF);  // the scratch is read plainly
        B.ncols = wc;
        B.in = (const cx<float>*)scratch; B.in_pitch = (unsigned)Ws; B.in_bs = (long long)(n * Ws);
        B.in_bdiv = 0; B.in_bs_hi = 0;
        B.in_i_rows = b_i_rows; B.in_o_rows = b_o_rows;
        B.ld_rowmap = nullptr; B.ld_win = nullptr; B.ld_win2 = nullptr; B.gs = 0;
        B.out = c.out + c0;
        B.st_mul = n1;
        B.tw = tw2; B.tw_full = twf;
        B.f64 = (f64 && (h->col_f64_stages & 2)) ? 1 : 0; B.twd = twd2; B.twd_full = twdf;
        if (qmul > 0) {
            B.full_logn = logn; B.full_n = full_n; B.st_qmul = qmul; B.st_qadd = qadd;
        }
        B.conj_ld = 0;
        if (B.col_win) B.col_win += c0;
        rc = launch_col_checked(l2, 1, B, zb, n1, nb, st);
    }
    if (own) {
        he = hipFreeAsync(scratch, st);
        if (!rc && he != hipSuccess) rc = fail(SWIFTLY_ERR_HIP, "hipFreeAsync: %s", hipGetErrorString(he));
    }
    return rc;
}

// Lean path for complex64 transforms along the strided axis of row-major
// arrays (columns contiguous): swiftly_colpass.h.  Returns false when the call
// does not fit its constraints (the generic kernel handles it then).
static bool try_col_pass(swiftly_hip* h, int logn, const RowsArgs<float>& a, const OffTab& tab, hipStream_t st,
                         int* rc_out) {
    if (!a.rowfast || a.in_rs != 1 || a.out_rs != 1 || (tab.use & 8) || a.rm_mod > 0 || a.in_rowmap) return false;
    const uint64_t n = uint64_t(1) << logn, lim = uint64_t(1) << 32;
    const uint64_t W = (uint64_t)a.nrows;
    // all element offsets inside one batch item are 32 bit
    if (n * (uint64_t)a.in_cs + W >= lim || n * (uint64_t)a.out_cs + W >= lim || n * W >= lim) return false;
    const int nb = a.nbatch > 0 ? a.nbatch : 1;
    ColPassArgs c;
    std::memset(&c, 0, sizeof c);
    c.ncols = a.nrows;
    c.full_logn = logn;
    c.in = a.in; c.out = a.out;
    c.in_pitch = a.in_cs; c.out_pitch = a.out_cs;
    c.in_bs = a.in_bs;
    c.out_bs = a.out_bs;
    c.ld_a = a.ld.a; c.ld_len = a.ld.len; c.ld_c = a.ld.c; c.ld_mod = a.ld.mod;
    c.ld_win = a.ld.win; c.ld_win2 = a.ld.win2;
    c.st_a = a.st.a; c.st_len = a.st.len; c.st_c = a.st.c; c.st_mod = a.st.mod;
    c.st_win = a.st.win; c.st_win2 = a.st.win2; c.st_win_bs = a.st_win_bs;
    c.st_rowmap = a.st_rowmap;
    c.col_win = a.row_win;
    c.scale = a.scale;
    c.conj_ld = a.conj_ld; c.conj_st = a.conj_st; c.accumulate = a.accumulate;
    c.ld_mul = c.st_mul = 1;
    // per-item map offsets (OffTab) -> per-subgrid tables of the column pass (batch item z = b)
    ColZ cz = plain_colz();
    if (tab.use) {
        static_assert(kMaxBatch <= kColZB, "per-item tables");
        cz.nb = nb;
        if (tab.use & 3) {
            cz.flags |= kZLoadB;
            for (int b = 0; b < nb; b++) {
                cz.b_lda[b] = (tab.use & 1) ? tab.ld_a[b] : a.ld.a;
                cz.b_ldc[b] = (tab.use & 2) ? tab.ld_c[b] : a.ld.c;
            }
        }
        if (tab.use & 4) {
            cz.flags |= kZStoreAB;
            for (int b = 0; b < nb; b++) cz.b_sta[b] = tab.st_a[b];
        }
    }
    const int rc = col_transform(h, logn, c, cz, (int)W, nb, st);
    if (rc == -1) return false;
    *rc_out = rc;
    return true;
}


__global__ void mul_windows_kernel(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}

// The mapped-store form of the band row kernel takes ONE output window: fold st_win * st_win2 into a
// stream-ordered temporary (freed by the caller with hipFreeAsync after the launch is queued).
static int single_store_window(RowPassArgs& r, hipStream_t st, void** tmp) {
    *tmp = nullptr;
    if (r.st_win && r.st_win2) {
        HIP_TRY(hipMallocAsync(tmp, (size_t)r.st_len * sizeof(float), st));
        hipLaunchKernelGGL(mul_windows_kernel, dim3((unsigned)((r.st_len + 255) / 256)), dim3(256), 0, st, (float*)*tmp,
                           r.st_win, r.st_win2, r.st_len);
        HIP_TRY(hipGetLastError());
        r.st_win = (const float*)*tmp;
    } else if (r.st_win2) {
        r.st_win = r.st_win2;
    }
    r.st_win2 = nullptr;
    return 0;
}

// Lean path for long complex64 transforms along the contiguous axis (one
// workgroup per row): swiftly_rowpass.h.  Returns false when the call does not
// fit (generic kernel handles it).
static bool try_row_pass(swiftly_hip* h, int logn, const RowsArgs<float>& a, const OffTab& tab, hipStream_t st,
                         int* rc_out) {
    if (a.rowfast || a.in_cs != 1 || a.out_cs != 1 || tab.use != 0 || (a.nbatch > 1)) return false;
    if (logn < kRowPassMinLog || logn > kRowPassMaxLog + 1) return false;
    if (a.ld.win2 || a.st_win_bs != 0) return false;
    const int n = 1 << logn;
    const bool ident_ld = a.ld.a == 0 && a.ld.len == n && a.ld.c == 0 && !a.ld.win;
    const bool ident_st = a.st.a == 0 && a.st.len == n && a.st.c == 0 && !a.st.win && !a.st.win2;
    RowPassArgs r;
    std::memset(&r, 0, sizeof r);
    r.in = a.in; r.out = a.out;
    r.in_pitch = a.in_rs; r.out_pitch = a.out_rs;
    r.nrows = a.nrows;
    r.rm_mod = a.rm_mod; r.rm_inner = a.rm_inner; r.rm_outer = a.rm_outer; r.rm_full = a.rm_full;
    r.in_rowmap = a.in_rowmap;
    if (a.st_rowmap) return false;
    r.row_win = a.row_win;
    if (a.row_win && logn > kRowPassMaxLog) return false;  // the two-workgroup kernels take it through prepare_facet_band only
    r.ld_a = a.ld.a; r.ld_len = a.ld.len; r.ld_c = a.ld.c; r.ld_mod = a.ld.mod; r.ld_win = a.ld.win;
    r.st_a = a.st.a; r.st_len = a.st.len; r.st_c = a.st.c; r.st_mod = a.st.mod; r.st_win = a.st.win; r.st_win2 = a.st.win2;
    r.tw = twiddles<float>(h, logn);
    if (!r.tw) return false;
    r.scale = a.scale; r.conj_ld = a.conj_ld; r.conj_st = a.conj_st; r.accumulate = a.accumulate;
    const int mode = ident_st ? 0 : (ident_ld ? 1 : 2);
    // N = 32768: the two-workgroup band kernel for the load map of a prepare_* primitive, else the r1 split kernel
    const bool prep_ld = r.ld_c == 0 && r.ld_mod == r.ld_len;    // load map of a prepare_* primitive (or identity)
    const bool fin_st = r.st_c == 0 && r.st_mod == r.st_len;     // store map of a finish_* primitive
    if (logn == 15 && mode == 0 && !a.accumulate) {
        const cx<float>* tw14 = twiddles<float>(h, 14);
        const cx<float>* tw13 = twiddles<float>(h, 13);
        if (tw14 && prep_ld) {
            int e2 = launch_row_pass_band(r, tw14, r.tw, st);
            *rc_out = e2 ? fail(SWIFTLY_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e2)) : 0;
            return true;
        }
        if (tw14 && tw13) {
            int e2 = launch_row_pass_split(r, tw14, tw13, r.tw, st);
            *rc_out = e2 ? fail(SWIFTLY_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e2)) : 0;
            return true;
        }
    }
    if (logn == 15 && mode != 0 && fin_st && prep_ld && !r.ld_win) {  // finish_* along the contiguous axis
        const cx<float>* tw14 = twiddles<float>(h, 14);
        if (tw14) {
            r.band_len = -1;  // selects the mapped-store variant
            void* tmp = nullptr;
            if (int rc = single_store_window(r, st, &tmp)) {
                *rc_out = rc;
                return true;
            }
            int e2 = launch_row_pass_band(r, tw14, r.tw, st, &h->win4);
            if (tmp) (void)hipFreeAsync(tmp, st);
            *rc_out = e2 ? fail(SWIFTLY_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e2)) : 0;
            return true;
        }
    }
    if (logn == 16) {  // 65536-point rows: only the two-workgroup kernel exists (prepare_* loads / finish_* stores)
        const cx<float>* tw15 = twiddles<float>(h, 15);
        const bool ok0 = mode == 0 && prep_ld && !a.accumulate, ok1 = mode != 0 && fin_st && prep_ld && !r.ld_win;
        if (!tw15 || !(ok0 || ok1)) return false;
        void* tmp = nullptr;
        if (ok1) {
            r.band_len = -1;
            if (int rc = single_store_window(r, st, &tmp)) {
                *rc_out = rc;
                return true;
            }
        }
        int e2 = launch_row_pass_band_n(16, r, tw15, r.tw, st);
        if (tmp) (void)hipFreeAsync(tmp, st);
        *rc_out = e2 ? fail(SWIFTLY_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e2)) : 0;
        return true;
    }
    int e = launch_row_pass(logn, mode, r, st);
    *rc_out = e ? fail(SWIFTLY_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e)) : 0;
    return true;
}


// Launch the mapped row FFT for `a` (batch of a.nbatch <= kMaxBatch items with
// per-item offsets in `tab`).  Transforms of length >= 2^kTwoPassMinLog along
// a strided axis are decomposed (four-step) through a stream-ordered scratch.
// Sub-transform form (swiftly_mixed.h): qmul = Q > 1 says that `a` is sub-transform j = qadd of a length-Q*2^logn
// transform whose radix-Q pass has run: the input is the plain scratch of that pass (a.raw_ld set by the caller, element
// e at in + e*in_cs), the store map refers to the full length a.full_n with plain output index Q*e + j.
template <typename R>
static int run_rows_chunk(swiftly_hip* h, int logn, RowsArgs<R>& a, const OffTab& tab, hipStream_t st, int qmul = 1,
                          int qadd = 0) {
    a.tw = twiddles<R>(h, logn);
    if (!a.tw) return fail(SWIFTLY_ERR_HIP, "internal: missing twiddle table for 2^%d", logn);
    const bool sub = qmul > 1;
    const bool all_j = sub && qadd < 0;  // all Q sub-transforms in this launch: outer index = j, input at in + j*in_os
    a.full_logn = logn;
    if (!sub) a.full_n = 0;
    a.ld_mul = 1;
    a.st_mul = qmul;
    a.st_add0 = all_j ? 0 : qadd;
    a.ld_addmul = 0;
    a.st_addmul = all_j ? 1 : 0;
    a.outer = all_j ? qmul : 1;
    // the Q sub-transforms of a row write interleaved elements of the same lines: their workgroups are enumerated 8 block
    // ids apart = on one XCD, whose L2 merges the partial lines (K1 of 24k[1]-n12k-1k 1.52 -> 0.88 ms per facet, r3)
    a.outer_group = all_j ? 1 : 0;
    if (!all_j) a.in_os = 0;
    a.out_os = 0;
    a.tw_full = nullptr;
    a.tw_on_store = 0;
    a.raw_st = 0;
    if (!sub) a.raw_ld = 0;
    if constexpr (std::is_same<R, float>::value) {
        int rc = 0;
        if (!sub && try_col_pass(h, logn, a, tab, st, &rc)) return rc;
        if (!sub && try_row_pass(h, logn, a, tab, st, &rc)) return rc;
    }
    if (logn > kMaxLogNFloat && sizeof(R) == 4)
        return fail(SWIFTLY_ERR_UNSUPPORTED, "transform length 65536 is only supported for complex64 prepare_* / finish_* "
                    "calls with unit stride along the transform axis or along a strided axis of contiguous rows");
    if (!(a.rowfast && logn >= kTwoPassMinLog)) return launch_checked(logn, a, tab, st);

    // ---- four-step: N = n1 * n2, input index y = y1*n2 + y2, output index k = k1 + n1*k2
    //   pass A (length n1 over y1, one per y2): scratch[k1*n2 + y2] = W_N^(y2 k1) * sum_y1 x[y1 n2 + y2] W_n1^(y1 k1)
    //   pass B (length n2 over y2, one per k1): X[k1 + n1 k2]       = sum_y2 scratch[k1*n2 + y2] W_n2^(y2 k2)
    const uint64_t n = uint64_t(1) << logn;
    const int l1 = logn / 2, l2 = logn - l1;
    const int n1 = 1 << l1, n2 = 1 << l2;
    const cx<R>* tw1 = twiddles<R>(h, l1);
    const cx<R>* tw2 = twiddles<R>(h, l2);
    if (!tw1 || !tw2) return fail(SWIFTLY_ERR_HIP, "internal: missing twiddle tables for the two-pass transform");
    const long long W = a.nrows;  // scratch row width (rows of the op are contiguous in memory)
    if (n * (uint64_t)W >= (uint64_t(1) << 32))
        return fail(SWIFTLY_ERR_PARAM, "two-pass scratch exceeds 2^32 elements; split the call by columns");
    const int nb = a.nbatch > 0 ? a.nbatch : 1;
    void* scratch = nullptr;
    HIP_TRY(hipMallocAsync(&scratch, (size_t)nb * n * (size_t)W * sizeof(cx<R>), st));
    RowsArgs<R> A = a;
    A.tw = tw1;
    A.tw_full = a.tw;
    A.tw_on_store = 1;
    A.outer = n2;
    A.ld_mul = n2;
    A.ld_addmul = 1;
    if (sub) {  // raw input: element y1*n2 + y2 of the sub-transform, outer index y2
        if ((uint64_t)a.in_cs * n >= (uint64_t(1) << 32)) {
            (void)hipFreeAsync(scratch, st);
            return fail(SWIFTLY_ERR_PARAM, "transform length * column stride must be < 2^32");
        }
        A.in_cs = a.in_cs * (unsigned)n2;
        A.in_os = (long long)a.in_cs;
    }
    A.st_mul = 1;
    A.st_add0 = 0;
    A.raw_st = 1;
    A.out = (cx<R>*)scratch;
    A.out_rs = 1;
    A.out_cs = (unsigned)(n2 * W);
    A.out_os = W;
    A.out_bs = (long long)(n * W);
    A.conj_st = 0;
    A.scale = (R)1;
    A.accumulate = 0;
    A.st_rowmap = nullptr;  // output-side maps / windows belong to pass B
    A.row_win = nullptr;
    int rc = launch_checked(l1, A, tab, st);
    if (!rc) {
        RowsArgs<R> B = a;
        B.tw = tw2;
        B.in = (const cx<R>*)scratch;
        B.raw_ld = 1;
        B.in_rs = 1;
        B.in_cs = (unsigned)W;
        B.in_os = (long long)n2 * W;
        B.in_bs = (long long)(n * W);
        B.rm_mod = 0;
        B.in_rowmap = nullptr;  // the scratch is indexed by plain (k1, y2), never through the compaction map
        B.outer = n1;
        B.st_mul = qmul * n1;
        B.st_addmul = qmul;
        B.st_add0 = qadd;
        B.conj_ld = 0;
        rc = launch_checked(l2, B, tab, st);
    }
    hipError_t e = hipFreeAsync(scratch, st);
    if (!rc && e != hipSuccess) return fail(SWIFTLY_ERR_HIP, "hipFreeAsync: %s", hipGetErrorString(e));
    return rc;
}

// Transform length n that is NOT a power of two: Bluestein through two power-of-two transforms of length
// L >= 2n - 1 on a stream-ordered work buffer (swiftly_bluestein.h).
template <typename R>
static int run_rows_bluestein(swiftly_hip* h, int64_t n, RowsArgs<R>& a, const OffTab& tab, hipStream_t st) {
    auto it = h->blu.find(n);
    const cx<R>* chirp = nullptr;
    const cx<R>* spec = nullptr;
    int logL = 0;
    if (it != h->blu.end()) {
        logL = it->second.logL;
        if constexpr (sizeof(R) == 4) {
            chirp = it->second.chirp_f;
            spec = it->second.spec_f;
        } else {
            chirp = it->second.chirp_d;
            spec = it->second.spec_d;
        }
    }
    if (!chirp || !spec)
        return fail(SWIFTLY_ERR_UNSUPPORTED,
                    "transform length %lld (not a power of two) needs a convolution of length >= %lld, beyond the %s kernels",
                    (long long)n, (long long)(2 * n - 1), sizeof(R) == 8 ? "complex128 (8192)" : "complex64 (65536)");
    const int64_t L = int64_t(1) << logL;
    const int nb = a.nbatch > 0 ? a.nbatch : 1;
    const int64_t rows_total = (int64_t)nb * a.nrows;
    if (rows_total > 0x7fffffff) return fail(SWIFTLY_ERR_PARAM, "too many rows");
    void *w1 = nullptr, *w2 = nullptr;
    const size_t bytes = (size_t)rows_total * (size_t)L * sizeof(cx<R>);
    HIP_TRY(hipMallocAsync(&w1, bytes, st));
    hipError_t e2 = hipMallocAsync(&w2, bytes, st);
    if (e2 != hipSuccess) {
        (void)hipFreeAsync(w1, st);
        return fail(SWIFTLY_ERR_HIP, "hipMallocAsync(Bluestein work): %s", hipGetErrorString(e2));
    }
    BluArgs<R> B;
    B.a = a;
    B.n = (int)n;
    B.L = (int)L;
    B.chirp = chirp;
    B.work = (cx<R>*)w1;
    const unsigned gy = (unsigned)std::min<int64_t>(a.nrows, 65535);
    hipLaunchKernelGGL((blu_load_kernel<R>), dim3((unsigned)((L + 255) / 256), gy, (unsigned)nb), dim3(256), 0, st, B, tab);
    int rc = (int)hipGetLastError() ? fail(SWIFTLY_ERR_HIP, "kernel launch failed (blu_load)") : 0;
    RowsArgs<R> f;
    OffTab none;
    none.use = 0;
    auto fft_L = [&](const void* src, void* dst, bool inverse) -> int {
        std::memset(&f, 0, sizeof f);
        f.in = (const cx<R>*)src;
        f.out = (cx<R>*)dst;
        f.in_rs = f.out_rs = L;
        f.in_cs = f.out_cs = 1;
        f.nrows = (int)rows_total;
        f.ld = identity_map<R>((int)L);
        f.st = identity_map<R>((int)L);
        f.scale = inverse ? (R)(1.0 / (double)L) : (R)1;
        f.conj_ld = f.conj_st = inverse ? 1 : 0;
        f.nbatch = 1;
        return run_rows_chunk(h, logL, f, none, st);
    };
    if (!rc) rc = fft_L(w1, w2, false);
    if (!rc) {
        const unsigned gy2 = (unsigned)std::min<int64_t>(rows_total, 65535);
        hipLaunchKernelGGL((blu_mul_kernel<R>), dim3((unsigned)((L + 255) / 256), gy2), dim3(256), 0, st, (cx<R>*)w2, spec,
                           (long long)rows_total, (int)L);
        if (hipGetLastError() != hipSuccess) rc = fail(SWIFTLY_ERR_HIP, "kernel launch failed (blu_mul)");
    }
    if (!rc) rc = fft_L(w2, w1, true);
    if (!rc) {
        hipLaunchKernelGGL((blu_store_kernel<R>), dim3((unsigned)((n + 255) / 256), gy, (unsigned)nb), dim3(256), 0, st, B, tab);
        if (hipGetLastError() != hipSuccess) rc = fail(SWIFTLY_ERR_HIP, "kernel launch failed (blu_store)");
    }
    (void)hipFreeAsync(w1, st);
    (void)hipFreeAsync(w2, st);
    return rc;
}

// Transform length n = Q * 2^k (Q in {3, 5, 7, 9}): one radix-Q pass (swiftly_mixed.h) into a stream-ordered scratch,
// then the Q power-of-two sub-transforms with the store map of the primitive.
template <typename R>
static int run_rows_mixed(swiftly_hip* h, int64_t n, const swiftly_hip::Mixed& mx, RowsArgs<R>& a, const OffTab& tab,
                          hipStream_t st) {
    const cx<R>* tw_n;
    if constexpr (sizeof(R) == 4) tw_n = mx.tw_f; else tw_n = mx.tw_d;
    if (!tw_n) return fail(SWIFTLY_ERR_UNSUPPORTED, "transform length %lld: no %s tables", (long long)n, sizeof(R) == 8 ? "complex128" : "complex64");
    const int Q = mx.Q, logM = mx.logM;
    const long long M = 1ll << logM;
    const int nb = a.nbatch > 0 ? a.nbatch : 1;
    const long long W = a.nrows;
    if ((uint64_t)n * (uint64_t)W >= (uint64_t(1) << 32))
        return fail(SWIFTLY_ERR_PARAM, "rows * transform length must be < 2^32 for lengths that are not a power of two; split the call");
    void* scratch = nullptr;
    HIP_TRY(hipMallocAsync(&scratch, (size_t)nb * (size_t)n * (size_t)W * sizeof(cx<R>), st));
    MixedArgs<R> X;
    std::memset(&X, 0, sizeof X);
    X.Q = Q; X.M = (int)M; X.n = (int)n;
    for (int r = 0; r < Q; r++) {
        const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)r / (long double)Q;
        X.wq[r] = cx<R>{(R)cosl(ang), (R)sinl(ang)};
    }
    X.tw_n = tw_n;
    X.scratch = (cx<R>*)scratch;
    X.s_b = (long long)n * W;
    if (a.rowfast) {  // rows are the contiguous direction: scratch[b][j][y2][row]
        X.s_row = 1; X.s_y = W; X.s_j = M * W;
    } else {          // scratch[b][row][j][y2]
        X.s_row = n; X.s_j = M; X.s_y = 1;
    }
    RowsArgs<R> P = a;
    P.full_n = (int)n;
    int e = launch_mixed_pass(Q, P, tab, X, nb, st);
    int rc = e ? fail(SWIFTLY_ERR_HIP, "kernel launch failed (radix-%d pass): %s", Q, hipGetErrorString((hipError_t)e)) : 0;
    // the Q sub-transforms: ONE launch with j as the kernel's outer index when the sub-transform is a single pass
    // (qadd = -1), one four-step pair per j along a strided axis
    const bool one_launch = !(a.rowfast && logM >= kTwoPassMinLog);
    for (int j = 0; j < (one_launch ? 1 : Q) && !rc; j++) {
        RowsArgs<R> B = a;
        B.in = (const cx<R>*)scratch + (long long)j * X.s_j;
        B.in_rs = X.s_row;
        B.in_cs = (unsigned)X.s_y;
        B.in_bs = X.s_b;
        B.raw_ld = 1;
        B.conj_ld = 0;        // applied by the pass
        B.rm_mod = 0;
        B.in_rowmap = nullptr;
        B.full_n = (int)n;
        B.in_os = X.s_j;      // (used by the one-launch form only)
        rc = run_rows_chunk(h, logM, B, tab, st, Q, one_launch ? -1 : j);
    }
    hipError_t e2 = hipFreeAsync(scratch, st);
    if (!rc && e2 != hipSuccess) rc = fail(SWIFTLY_ERR_HIP, "hipFreeAsync: %s", hipGetErrorString(e2));
    return rc;
}

// `fill(b, tab_index)` sets the per-item offsets of batch item b into the
// OffTab; called once per item per chunk.  `nlen` = transform length, `logn` = its log2 or -1.
template <typename R, class Fill>
static int run_rows(swiftly_hip* h, int64_t nlen, int logn, RowsArgs<R>& a, const Batch& bt, int use_bits, Fill&& fill,
                    hipStream_t st) {
    constexpr int maxlog = sizeof(R) == 8 ? kMaxLogNDouble : kMaxLogNFloat + 1;  // 2^16: lean complex64 kernels only
    if (logn >= 0 && (logn < kMinLogN || logn > maxlog))
        return fail(SWIFTLY_ERR_UNSUPPORTED,
                    "transform length %lld is not supported by the HIP backend (power of two in [8, %d], or a length "
                    "whose Bluestein convolution fits, for %s)",
                    (long long)nlen, 1 << maxlog, sizeof(R) == 8 ? "complex128" : "complex64");
    const uint64_t n = (uint64_t)nlen;
    if (n * (uint64_t)a.in_cs >= (uint64_t(1) << 32) || n * (uint64_t)a.out_cs >= (uint64_t(1) << 32))
        return fail(SWIFTLY_ERR_PARAM, "transform length * column stride must be < 2^32");
    if (a.nrows <= 0 || bt.n <= 0) return 0;
    const cx<R>* in0 = a.in;
    cx<R>* out0 = a.out;
    const R* win0 = a.st.win;
    for (int64_t b0 = 0; b0 < bt.n; b0 += kMaxBatch) {
        const int nb = (int)std::min<int64_t>(kMaxBatch, bt.n - b0);
        OffTab tab;
        tab.use = bt.offs ? use_bits : 0;
        if (tab.use)
            for (int b = 0; b < nb; b++) fill(b0 + b, b, tab);
        RowsArgs<R> c = a;
        c.in = in0 + b0 * bt.in_bs;
        c.out = out0 + b0 * bt.out_bs;
        c.in_bs = bt.in_bs;
        c.out_bs = bt.out_bs;
        c.nbatch = nb;
        c.st_win_bs = bt.mask_bs;
        if (win0) c.st.win = win0 + b0 * bt.mask_bs;
        int rc;
        if (logn >= 0) {
            rc = run_rows_chunk(h, logn, c, tab, st);
        } else {
            // SWIFTLY_NO_MIXED=1 (read per call): Bluestein for every length that is not a power of two (A/B, tests)
            const char* no_mixed = getenv("SWIFTLY_NO_MIXED");
            auto it = h->mixed.find(nlen);
            const bool have = it != h->mixed.end() && (sizeof(R) == 4 ? it->second.tw_f != nullptr : it->second.tw_d != nullptr);
            rc = (have && !(no_mixed && atoi(no_mixed))) ? run_rows_mixed(h, nlen, it->second, c, tab, st)
                                                          : run_rows_bluestein(h, nlen, c, tab, st);
        }
        if (rc) return rc;
    }
    return 0;
}

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

static const auto kNoFill = [](int64_t, int, OffTab&) {};

// row_gather_off: when not INT64_MIN, the rows of `in` are gathered like
// extract_from_facet(subgrid_off = row_gather_off) would along the OTHER axis
// (fuses api_helper.extract_column, api_helper.py:200-210, into one kernel).
template <typename R>
static int do_prepare_facet(swiftly_hip* h, const void* in, int64_t rows, int64_t yB, int64_t in_rs, int64_t in_cs,
                            void* out, int64_t out_rs, int64_t out_cs, int64_t off, int64_t row_gather_off,
                            const int32_t* out_rowmap, const int32_t* in_rowmap, int fold_other, int no_window,
                            hipStream_t st) {
    const int yN = (int)h->yN, m = (int)h->m;
    RowsArgs<R> a;
    fill_io(a, in, rows, in_rs, in_cs, out, out_rs, out_cs);
    const int lo = yN / 2 - (int)(yB / 2);
    a.ld = AxisMap<R>{pmod(-(off + lo), yN), (int)yB, 0, (int)yB, invp<R>(h) + lo, nullptr};
    a.st = identity_map<R>(yN);
    a.conj_ld = a.conj_st = 1;
    a.scale = (R)(1.0 / yN);
    a.st_rowmap = out_rowmap;
    a.in_rowmap = in_rowmap;
    if (no_window) a.ld.win = nullptr;
    if (fold_other) a.row_win = invp<R>(h) + (yN / 2 - (int)(rows / 2));  // the rows of this call are the other axis
    if (row_gather_off != INT64_MIN) {
        const int64_t s = floordiv(row_gather_off * h->yN, h->N);
        a.rm_mod = m;
        a.rm_inner = pmod(-s, m);
        a.rm_outer = pmod(yN / 2 - m / 2 + s, yN);
        a.rm_full = yN;
    }
    return run_rows(h, h->yN, h->log_yN, a, Batch{}, 0, kNoFill, st);
}

template <typename R>
static int do_add_to_subgrid(swiftly_hip* h, const void* in, int64_t rows, int64_t in_rs, int64_t in_cs, void* out,
                             int64_t out_rs, int64_t out_cs, int64_t off, const Batch& bt, hipStream_t st) {
    const int m = (int)h->m, xM = (int)h->xM;
    auto maps = [&](int64_t o, int& a_, int& c_) {
        const int64_t sp = floordiv(o * h->xM, h->N);
        a_ = pmod(-sp, m);
        c_ = pmod(xM / 2 - m / 2 + sp, xM);
    };
    RowsArgs<R> a;
    fill_io(a, in, rows, in_rs, in_cs, out, out_rs, out_cs);
    a.ld = identity_map<R>(m);
    int a0, c0;
    maps(off, a0, c0);
    a.st = AxisMap<R>{a0, m, c0, xM, fnwin<R>(h), nullptr};
    a.accumulate = 1;
    return run_rows(h, h->m, h->log_m, a, bt, 4 | 8,
                    [&](int64_t gb, int b, OffTab& t) { maps(bt.offs[gb], t.st_a[b], t.st_c[b]); }, st);
}

template <typename R>
static int do_finish_subgrid(swiftly_hip* h, const void* in, int64_t rows, int64_t in_rs, int64_t in_cs, void* out,
                             int64_t out_rs, int64_t out_cs, int64_t off, int64_t xA, const void* mask,
                             const Batch& bt, hipStream_t st) {
    const int xM = (int)h->xM;
    RowsArgs<R> a;
    fill_io(a, in, rows, in_rs, in_cs, out, out_rs, out_cs);
    a.ld = identity_map<R>(xM);
    a.st = AxisMap<R>{pmod(-(xM / 2 - xA / 2 + off), xM), (int)xA, 0, (int)xA, (const R*)mask, nullptr};
    a.conj_ld = a.conj_st = 1;
    a.scale = (R)(1.0 / xM);
    return run_rows(h, h->xM, h->log_xM, a, bt, 4,
                    [&](int64_t gb, int b, OffTab& t) { t.st_a[b] = pmod(-(xM / 2 - xA / 2 + bt.offs[gb]), xM); }, st);
}

template <typename R>
static int do_prepare_subgrid(swiftly_hip* h, const void* in, int64_t rows, int64_t xA, int64_t in_rs, int64_t in_cs,
                              void* out, int64_t out_rs, int64_t out_cs, int64_t off, const Batch& bt,
                              hipStream_t st) {
    const int xM = (int)h->xM;
    RowsArgs<R> a;
    fill_io(a, in, rows, in_rs, in_cs, out, out_rs, out_cs);
    a.ld = AxisMap<R>{pmod(-(xM / 2 - xA / 2 + off), xM), (int)xA, 0, (int)xA, nullptr, nullptr};
    a.st = identity_map<R>(xM);
    return run_rows(h, h->xM, h->log_xM, a, bt, 1,
                    [&](int64_t gb, int b, OffTab& t) { t.ld_a[b] = pmod(-(xM / 2 - xA / 2 + bt.offs[gb]), xM); }, st);
}

template <typename R>
static int do_extract_from_subgrid(swiftly_hip* h, const void* in, int64_t rows, int64_t in_rs, int64_t in_cs,
                                   void* out, int64_t out_rs, int64_t out_cs, int64_t off, const Batch& bt,
                                   hipStream_t st) {
    const int m = (int)h->m, xM = (int)h->xM;
    auto maps = [&](int64_t o, int& a_, int& c_) {
        const int64_t sp = floordiv(o * h->xM, h->N);
        a_ = pmod(-sp, m);
        c_ = pmod(xM / 2 - m / 2 + sp, xM);
    };
    RowsArgs<R> a;
    fill_io(a, in, rows, in_rs, in_cs, out, out_rs, out_cs);
    int a0, c0;
    maps(off, a0, c0);
    a.ld = AxisMap<R>{a0, m, c0, xM, fnwin<R>(h), nullptr};
    a.st = identity_map<R>(m);
    a.conj_ld = a.conj_st = 1;
    a.scale = (R)(1.0 / m);
    return run_rows(h, h->m, h->log_m, a, bt, 1 | 2,
                    [&](int64_t gb, int b, OffTab& t) { maps(bt.offs[gb], t.ld_a[b], t.ld_c[b]); }, st);
}

template <typename R>
static int do_finish_facet(swiftly_hip* h, const void* in, int64_t rows, int64_t in_rs, int64_t in_cs, void* out,
                           int64_t out_rs, int64_t out_cs, int64_t off, int64_t yB, const void* mask, const Batch& bt,
                           hipStream_t st, int64_t band_start = 0, int64_t band_len = -1) {
    const int yN = (int)h->yN;
    const int lo = yN / 2 - (int)(yB / 2);
    RowsArgs<R> a;
    fill_io(a, in, rows, in_rs, in_cs, out, out_rs, out_cs);
    a.ld = identity_map<R>(yN);
    // band input: element d of a row is column (band_start + d) mod yN of the padded facet, the rest is zero
    if (band_len >= 0) a.ld = AxisMap<R>{pmod(-band_start, yN), (int)band_len, 0, (int)band_len, nullptr, nullptr};
    // mask goes to `win` (it is the per-item one), the PSWF window to `win2`
    a.st = AxisMap<R>{pmod(-(lo + off), yN), (int)yB, 0, (int)yB, (const R*)mask, invp<R>(h) + lo};
    return run_rows(h, h->yN, h->log_yN, a, bt, 4,
                    [&](int64_t gb, int b, OffTab& t) { t.st_a[b] = pmod(-(lo + bt.offs[gb]), yN); }, st);
}

extern "C" {

int swiftly_hip_prepare_facet(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t facet_size,
                              int64_t in_rs, int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs,
                              int64_t facet_off, void* stream) {
    CHECK_COMMON();
    CHECK_FACET_SIZE();
    return DISPATCH(do_prepare_facet, h, in, rows, facet_size, in_rs, in_cs, out, out_rs, out_cs, facet_off, INT64_MIN,
                    nullptr, nullptr, 0, 0, (hipStream_t)stream);
}

int swiftly_hip_prepare_facet_rows(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t facet_size,
                                   int64_t in_rs, int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs,
                                   int64_t facet_off, const int32_t* out_rowmap, int fold_other_axis_window,
                                   void* stream) {
    CHECK_COMMON();
    CHECK_FACET_SIZE();
    if (fold_other_axis_window && (rows <= 0 || rows >= h->yN))
        return fail(SWIFTLY_ERR_PARAM, "other-axis facet size %lld must be in [1, yN_size - 1]", (long long)rows);
    return DISPATCH(do_prepare_facet, h, in, rows, facet_size, in_rs, in_cs, out, out_rs, out_cs, facet_off, INT64_MIN,
                    out_rowmap, nullptr, fold_other_axis_window, 0, (hipStream_t)stream);
}

int swiftly_hip_extract_column(swiftly_hip_t* h, int dtype, const void* in, int64_t facet_size, int64_t in_rs,
                               int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs, int64_t subgrid_off0,
                               int64_t facet_off1, void* stream) {
    const int64_t rows = h ? h->m : 0;
    CHECK_COMMON();
    CHECK_FACET_SIZE();
    return DISPATCH(do_prepare_facet, h, in, rows, facet_size, in_rs, in_cs, out, out_rs, out_cs, facet_off1,
                    subgrid_off0, nullptr, nullptr, 0, 0, (hipStream_t)stream);
}

int swiftly_hip_extract_column_rows(swiftly_hip_t* h, int dtype, const void* in, int64_t facet_size, int64_t in_rs,
                                    int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs, int64_t subgrid_off0,
                                    int64_t facet_off1, const int32_t* in_rowmap, int prewindowed, void* stream) {
    const int64_t rows = h ? h->m : 0;
    CHECK_COMMON();
    CHECK_FACET_SIZE();
    return DISPATCH(do_prepare_facet, h, in, rows, facet_size, in_rs, in_cs, out, out_rs, out_cs, facet_off1,
                    subgrid_off0, nullptr, in_rowmap, 0, prewindowed, (hipStream_t)stream);
}

int swiftly_hip_extract_from_facet_batch(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_rs,
                                         int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs,
                                         int64_t subgrid_off, int64_t nbatch, int64_t in_bs, int64_t out_bs,
                                         const int64_t* subgrid_offs, void* stream) {
    CHECK_COMMON();
    CHECK_BATCH();
    Batch bt{nbatch, in_bs, out_bs, subgrid_offs, 0};
    if (dtype == SWIFTLY_C64)
        return run_modcopy<float, false>(h, in, rows, in_rs, in_cs, out, out_rs, out_cs, subgrid_off, bt, (hipStream_t)stream);
    return run_modcopy<double, false>(h, in, rows, in_rs, in_cs, out, out_rs, out_cs, subgrid_off, bt, (hipStream_t)stream);
}
int swiftly_hip_extract_from_facet(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_rs,
                                   int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs, int64_t subgrid_off,
                                   void* stream) {
    return swiftly_hip_extract_from_facet_batch(h, dtype, in, rows, in_rs, in_cs, out, out_rs, out_cs, subgrid_off, 1,
                                                0, 0, nullptr, stream);
}

int swiftly_hip_add_to_subgrid_batch(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_rs,
                                     int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs, int64_t facet_off,
                                     int64_t nbatch, int64_t in_bs, int64_t out_bs, const int64_t* facet_offs,
                                     void* stream) {
    CHECK_COMMON();
    CHECK_BATCH();
    CHECK_ACCUMULATE_BATCH();
    Batch bt{nbatch, in_bs, out_bs, facet_offs, 0};
    return DISPATCH(do_add_to_subgrid, h, in, rows, in_rs, in_cs, out, out_rs, out_cs, facet_off, bt, (hipStream_t)stream);
}
int swiftly_hip_add_to_subgrid(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_rs,
                               int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs, int64_t facet_off,
                               void* stream) {
    return swiftly_hip_add_to_subgrid_batch(h, dtype, in, rows, in_rs, in_cs, out, out_rs, out_cs, facet_off, 1, 0, 0,
                                            nullptr, stream);
}

int swiftly_hip_finish_subgrid_batch(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_rs,
                                     int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs, int64_t subgrid_off,
                                     int64_t subgrid_size, const void* mask, int64_t nbatch, int64_t in_bs,
                                     int64_t out_bs, const int64_t* subgrid_offs, int64_t mask_bs, void* stream) {
    CHECK_COMMON();
    CHECK_BATCH();
    CHECK_SUBGRID_SIZE();
    Batch bt{nbatch, in_bs, out_bs, subgrid_offs, mask ? mask_bs : 0};
    return DISPATCH(do_finish_subgrid, h, in, rows, in_rs, in_cs, out, out_rs, out_cs, subgrid_off, subgrid_size, mask,
                    bt, (hipStream_t)stream);
}
int swiftly_hip_finish_subgrid(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_rs,
                               int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs, int64_t subgrid_off,
                               int64_t subgrid_size, const void* mask, void* stream) {
    return swiftly_hip_finish_subgrid_batch(h, dtype, in, rows, in_rs, in_cs, out, out_rs, out_cs, subgrid_off,
                                            subgrid_size, mask, 1, 0, 0, nullptr, 0, stream);
}

int swiftly_hip_prepare_subgrid_batch(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t subgrid_size,
                                      int64_t in_rs, int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs,
                                      int64_t subgrid_off, int64_t nbatch, int64_t in_bs, int64_t out_bs,
                                      const int64_t* subgrid_offs, void* stream) {
    CHECK_COMMON();
    CHECK_BATCH();
    CHECK_SUBGRID_SIZE();
    Batch bt{nbatch, in_bs, out_bs, subgrid_offs, 0};
    return DISPATCH(do_prepare_subgrid, h, in, rows, subgrid_size, in_rs, in_cs, out, out_rs, out_cs, subgrid_off, bt,
                    (hipStream_t)stream);
}
int swiftly_hip_prepare_subgrid(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t subgrid_size,
                                int64_t in_rs, int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs,
                                int64_t subgrid_off, void* stream) {
    return swiftly_hip_prepare_subgrid_batch(h, dtype, in, rows, subgrid_size, in_rs, in_cs, out, out_rs, out_cs,
                                             subgrid_off, 1, 0, 0, nullptr, stream);
}

// Swiftly.prepare_subgrid_inplace(data[rows, xM], subgrid_off) (core.py:837-840): `data` holds the subgrid already
// zero-padded to xM (pad_mid: centred); in place it becomes fft(roll(data, subgrid_off)) along the strided/contiguous
// axis given by the two strides.  = prepare_subgrid with subgrid_size = xM_size and in == out: every kernel reads all
// of a row (column tile) before it writes any of it, and the four-step passes go through their own scratch.
int swiftly_hip_prepare_subgrid_inplace(swiftly_hip_t* h, int dtype, void* data, int64_t rows, int64_t row_stride,
                                        int64_t col_stride, int64_t subgrid_off, void* stream) {
    if (!h) return fail(SWIFTLY_ERR_PARAM, "null argument");
    return swiftly_hip_prepare_subgrid(h, dtype, data, rows, h->xM, row_stride, col_stride, data, row_stride, col_stride,
                                       subgrid_off, stream);
}
// Swiftly.prepare_subgrid_inplace_2d(data[xM, xM], off0, off1) (core.py:851-853): both axes, axis 1 first
int swiftly_hip_prepare_subgrid_inplace_2d(swiftly_hip_t* h, int dtype, void* data, int64_t row_stride, int64_t col_stride,
                                           int64_t subgrid_off0, int64_t subgrid_off1, void* stream) {
    if (!h) return fail(SWIFTLY_ERR_PARAM, "null argument");
    int rc = swiftly_hip_prepare_subgrid_inplace(h, dtype, data, h->xM, row_stride, col_stride, subgrid_off1, stream);
    if (rc) return rc;
    return swiftly_hip_prepare_subgrid_inplace(h, dtype, data, h->xM, col_stride, row_stride, subgrid_off0, stream);
}
// Swiftly.add_to_subgrid_2d(in[m, m], out[xM, xM], facet_off0, facet_off1) (core.py:752-778; numpy form = two
// add_to_subgrid calls, core.py:274-285): ACCUMULATES.  Axis 0 first, as api_helper.py:85-99 does it (the float32
// rounding of the two orders differs: 5.6e-7 vs 9.0e-7 on the reference-generated golden contribution), into a
// stream-ordered [xM, m] intermediate, then axis 1 straight into `out`.
int swiftly_hip_add_to_subgrid_2d(swiftly_hip_t* h, int dtype, const void* in, int64_t in_row_stride,
                                  int64_t in_col_stride, void* out, int64_t out_row_stride, int64_t out_col_stride,
                                  int64_t facet_off0, int64_t facet_off1, void* stream) {
    const int64_t rows = h ? h->m : 0, in_cs = in_col_stride, out_cs = out_col_stride;
    CHECK_COMMON();
    const int64_t m = h->m, xM = h->xM;
    const size_t esz = dtype == SWIFTLY_C64 ? 8 : 16;
    void* tmp = nullptr;
    HIP_TRY(hipMallocAsync(&tmp, (size_t)m * (size_t)xM * esz, (hipStream_t)stream));
    HIP_TRY(hipMemsetAsync(tmp, 0, (size_t)m * (size_t)xM * esz, (hipStream_t)stream));  // add_to_subgrid accumulates
    // axis 0: the transform runs along the rows of `in` = its strided axis; the "rows" of the primitive are the m columns
    int rc = swiftly_hip_add_to_subgrid(h, dtype, in, m, in_col_stride, in_row_stride, tmp, 1, m, facet_off0, stream);
    if (!rc) rc = swiftly_hip_add_to_subgrid(h, dtype, tmp, xM, m, 1, out, out_row_stride, out_col_stride, facet_off1, stream);
    hipError_t e = hipFreeAsync(tmp, (hipStream_t)stream);
    if (!rc && e != hipSuccess) return fail(SWIFTLY_ERR_HIP, "hipFreeAsync: %s", hipGetErrorString(e));
    return rc;
}

int swiftly_hip_extract_from_subgrid_batch(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_rs,
                                           int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs,
                                           int64_t facet_off, int64_t nbatch, int64_t in_bs, int64_t out_bs,
                                           const int64_t* facet_offs, void* stream) {
    CHECK_COMMON();
    CHECK_BATCH();
    Batch bt{nbatch, in_bs, out_bs, facet_offs, 0};
    return DISPATCH(do_extract_from_subgrid, h, in, rows, in_rs, in_cs, out, out_rs, out_cs, facet_off, bt,
                    (hipStream_t)stream);
}
int swiftly_hip_extract_from_subgrid(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_rs,
                                     int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs, int64_t facet_off,
                                     void* stream) {
    return swiftly_hip_extract_from_subgrid_batch(h, dtype, in, rows, in_rs, in_cs, out, out_rs, out_cs, facet_off, 1,
                                                  0, 0, nullptr, stream);
}

int swiftly_hip_add_to_facet_batch(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_rs,
                                   int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs, int64_t subgrid_off,
                                   int64_t nbatch, int64_t in_bs, int64_t out_bs, const int64_t* subgrid_offs,
                                   void* stream) {
    CHECK_COMMON();
    CHECK_BATCH();
    CHECK_ACCUMULATE_BATCH();
    Batch bt{nbatch, in_bs, out_bs, subgrid_offs, 0};
    if (dtype == SWIFTLY_C64)
        return run_modcopy<float, true>(h, in, rows, in_rs, in_cs, out, out_rs, out_cs, subgrid_off, bt, (hipStream_t)stream);
    return run_modcopy<double, true>(h, in, rows, in_rs, in_cs, out, out_rs, out_cs, subgrid_off, bt, (hipStream_t)stream);
}
int swiftly_hip_add_to_facet(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_rs, int64_t in_cs,
                             void* out, int64_t out_rs, int64_t out_cs, int64_t subgrid_off, void* stream) {
    return swiftly_hip_add_to_facet_batch(h, dtype, in, rows, in_rs, in_cs, out, out_rs, out_cs, subgrid_off, 1, 0, 0,
                                          nullptr, stream);
}

int swiftly_hip_finish_facet_batch(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_rs,
                                   int64_t in_cs, void* out, int64_t out_rs, int64_t out_cs, int64_t facet_off,
                                   int64_t facet_size, const void* mask, int64_t nbatch, int64_t in_bs,
                                   int64_t out_bs, const int64_t* facet_offs, int64_t mask_bs, void* stream) {
    CHECK_COMMON();
    CHECK_BATCH();
    CHECK_FACET_SIZE();
    Batch bt{nbatch, in_bs, out_bs, facet_offs, mask ? mask_bs : 0};
    return DISPATCH(do_finish_facet, h, in, rows, in_rs, in_cs, out, out_rs, out_cs, facet_off, facet_size, mask, bt,
                    (hipStream_t)stream);
}
int swiftly_hip_finish_facet(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_rs, int64_t in_cs,
                             void* out, int64_t out_rs, int64_t out_cs, int64_t facet_off, int64_t facet_size,
                             const void* mask, void* stream) {
    return swiftly_hip_finish_facet_batch(h, dtype, in, rows, in_rs, in_cs, out, out_rs, out_cs, facet_off, facet_size,
                                          mask, 1, 0, 0, nullptr, 0, stream);
}

int swiftly_hip_prepare_facet_band(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t facet_size,
                                   int64_t in_row_stride, void* out, int64_t out_row_stride, int64_t facet_off,
                                   int64_t band_start, int64_t band_len, int fold_other_axis_window, void* stream) {
    return swiftly_hip_prepare_facet_band_rows(h, dtype, in, rows, facet_size, in_row_stride, out, out_row_stride, facet_off,
                                               band_start, band_len, fold_other_axis_window ? rows : 0, 0, stream);
}

// prepare_facet_band for the rows [other_axis_row0, other_axis_row0 + rows) of a facet whose other axis has
// other_axis_size pixels: the folded window of the other axis is that facet's, not the one of a `rows`-pixel facet
// (cooperative facets of the multi-GPU pass: every rank transforms a block of rows of the same facet);
// other_axis_size = 0: no window of the other axis.
static int prepare_facet_band_rows_impl(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t facet_size,
                                        int64_t in_row_stride, void* out, int64_t out_row_stride, int64_t facet_off,
                                        int64_t band_start, int64_t band_len, int64_t other_axis_size,
                                        int64_t other_axis_row0, const int32_t* win_d, int64_t nwin, int64_t win_full,
                                        void* stream) {
    const int fold_other_axis_window = other_axis_size > 0;
    if (!h || !in || !out) return fail(SWIFTLY_ERR_PARAM, "null argument");
    if (fold_other_axis_window && (other_axis_row0 < 0 || other_axis_row0 + rows > other_axis_size))
        return fail(SWIFTLY_ERR_PARAM, "rows [%lld, +%lld) are not inside a facet of %lld rows", (long long)other_axis_row0,
                    (long long)rows, (long long)other_axis_size);
    DeviceGuard device_guard_(h->device);
    if (dtype != SWIFTLY_C64) return fail(SWIFTLY_ERR_UNSUPPORTED, "prepare_facet_band: complex64 only");
    CHECK_FACET_SIZE();
    const int yN = (int)h->yN;
    const bool mixed_yN = h->log_yN < 0 && h->mixed.count(h->yN) && h->mixed.at(h->yN).tw_f;
    if (!mixed_yN && (h->log_yN < 3 || h->log_yN > 16))
        return fail(SWIFTLY_ERR_UNSUPPORTED, "prepare_facet_band: padded facet size %d not supported (power of two 8 .. 65536, or Q * 2^k with Q = 3, 5, 7, 9)", yN);
    if (rows < 0 || rows > 0x7fffffff) return fail(SWIFTLY_ERR_PARAM, "bad row count");
    if (band_len <= 0 || band_len > yN || band_start < 0 || band_start >= yN)
        return fail(SWIFTLY_ERR_PARAM, "band [%lld, +%lld) is not a cyclic range of [0, %d)", (long long)band_start, (long long)band_len, yN);
    if (fold_other_axis_window && (other_axis_size <= 0 || other_axis_size >= h->yN))
        return fail(SWIFTLY_ERR_PARAM, "other-axis facet size %lld must be in [1, yN_size - 1]", (long long)other_axis_size);
    if (rows == 0) return 0;
    if (!band_is_split(h)) {
        // short rows: the generic contiguous-axis prepare_facet, whole padded axis, plain column order
        if (band_start != 0 || band_len != yN)
            return fail(SWIFTLY_ERR_UNSUPPORTED, "prepare_facet_band: yN_size %d keeps the whole padded axis (band must be (0, yN_size))", yN);
        if (fold_other_axis_window && (other_axis_row0 != 0 || other_axis_size != rows))
            return fail(SWIFTLY_ERR_UNSUPPORTED, "prepare_facet_band_rows: row blocks need the split band layout (yN_size 16384 .. 65536)");
        return do_prepare_facet<float>(h, in, rows, facet_size, in_row_stride, 1, out, out_row_stride, 1, facet_off, INT64_MIN,
                                       nullptr, nullptr, fold_other_axis_window, 0, (hipStream_t)stream);
    }
    const int lo = yN / 2 - (int)(facet_size / 2);
    RowPassArgs r;
    std::memset(&r, 0, sizeof r);
    r.in = (const cx<float>*)in; r.out = (cx<float>*)out;
    r.in_pitch = in_row_stride; r.out_pitch = out_row_stride;
    r.nrows = (int)rows;
    r.ld_a = pmod(-(facet_off + lo), yN); r.ld_len = (int)facet_size; r.ld_c = 0; r.ld_mod = (int)facet_size;
    r.ld_win = h->invp_f + lo;
    r.st_len = yN; r.st_mod = yN;
    r.scale = (float)(1.0 / yN);
    r.conj_ld = r.conj_st = 1;
    r.row_win = fold_other_axis_window ? h->invp_f + (yN / 2 - (int)(other_axis_size / 2) + (int)other_axis_row0) : nullptr;
    r.band_start = (int)band_start; r.band_len = (int)band_len; r.band_half = (int)band_half_columns(band_len);
    const cx<float>* twh = twiddles<float>(h, h->log_yN - 1);
    const cx<float>* twf = twiddles<float>(h, h->log_yN);
    if (!twh || !twf) return fail(SWIFTLY_ERR_HIP, "internal: missing twiddle tables");
    if (win_d && win_full) {  // complete window rows (whole-row kernel)
        r.win_full = 1;
        r.win_pitch = win_full;  // (the flag carries the window stride)
        r.win_d = win_d; r.nwin = (int)nwin; r.win_logm = h->log_m;
        r.win_sp = pmod(floordiv(facet_off * h->xM, h->N), (int)h->m);
        r.win_fn = h->fn_f;
        r.win_tw_m = twiddles<float>(h, h->log_m);
        r.win_twc_m = h->log_m >= 6 ? compact_twiddles(h, h->log_m, h->log_m - 6) : nullptr;
    }
    int e = launch_row_pass_band_n(h->log_yN, r, twh, twf, (hipStream_t)stream, &h->win4);
    if (e == -2)
        return fail(SWIFTLY_ERR_UNSUPPORTED, "prepare_facet_window_rows: needs yN_size 32768, contribution size 512, even facet "
                    "size / offset / row stride and a band of at most %d physical columns", row_pass_whole_stage_columns());
    if (e) return fail(SWIFTLY_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    return 0;
}

int swiftly_hip_prepare_facet_band_rows(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t facet_size,
                                        int64_t in_row_stride, void* out, int64_t out_row_stride, int64_t facet_off,
                                        int64_t band_start, int64_t band_len, int64_t other_axis_size,
                                        int64_t other_axis_row0, void* stream) {
    return prepare_facet_band_rows_impl(h, dtype, in, rows, facet_size, in_row_stride, out, out_row_stride, facet_off, band_start,
                                        band_len, other_axis_size, other_axis_row0, nullptr, 0, 0, stream);
}

// (r6, axis-1-first pipeline) prepare_facet_band_rows that finishes the contiguous axis COMPLETELY inside K1: one persistent
// workgroup per CU owns whole rows (both output parities; swiftly_rowwhole.h), stages the band of a row in LDS and stores, for every
// window w, what swiftly_hip_finish_axis1_rows would produce for wave w from that band:
// out[row][w*m ..] = parity-split window band of  Fn[k] cfft_m(window w)[(k + s'1) mod m].
int swiftly_hip_prepare_facet_window_rows(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t facet_size,
                                          int64_t in_row_stride, void* out, int64_t out_row_stride, int64_t facet_off,
                                          int64_t band_start, int64_t band_len, int64_t other_axis_size,
                                          int64_t other_axis_row0, const int32_t* window_starts, int64_t nwindows,
                                          int64_t out_window_stride, void* stream) {
    if (!window_starts || nwindows <= 0) return fail(SWIFTLY_ERR_PARAM, "prepare_facet_window_rows: no windows");
    if (h && (out_window_stride < h->m || out_row_stride < h->m ||
              (out_window_stride < rows * out_row_stride && out_row_stride < nwindows * out_window_stride)))
        return fail(SWIFTLY_ERR_PARAM, "prepare_facet_window_rows: the window rows of the output overlap (row stride %lld, window stride %lld)",
                    (long long)out_row_stride, (long long)out_window_stride);
    return prepare_facet_band_rows_impl(h, dtype, in, rows, facet_size, in_row_stride, out, out_row_stride, facet_off, band_start,
                                        band_len, other_axis_size, other_axis_row0, window_starts, nwindows, out_window_stride, stream);
}

int swiftly_hip_finish_facet_band(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_row_stride,
                                  int64_t band_start, int64_t band_len, void* out, int64_t out_row_stride,
                                  int64_t facet_off, int64_t facet_size, const void* mask, void* stream) {
    const int64_t in_cs = 1, out_cs = 1;
    CHECK_COMMON();
    CHECK_FACET_SIZE();
    if (band_len <= 0 || band_len > h->yN || band_start < 0 || band_start >= h->yN)
        return fail(SWIFTLY_ERR_PARAM, "band [%lld, +%lld) is not a cyclic range of [0, %lld)", (long long)band_start,
                    (long long)band_len, (long long)h->yN);
    Batch bt{1, 0, 0, nullptr, 0};
    return DISPATCH(do_finish_facet, h, in, rows, in_row_stride, in_cs, out, out_row_stride, out_cs, facet_off, facet_size,
                    mask, bt, (hipStream_t)stream, band_start, band_len);
}


}  // extern "C"
