// Internals shared by the translation units of the C ABI (swiftly_abi.hip: handles + the eight primitives and their
// batch forms; swiftly_abi_pipeline.hip: the fused / per-wave entry points of the streaming classes;
// swiftly_abi_util.hip: device memory, stream and diagnostic helpers).  Not installed: include/swiftly_hip.h is the
// public header.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <atomic>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/swiftly_hip.h"
#include "swiftly_colpass.h"
#include "swiftly_rowpass.h"
#include "swiftly_sumfinish.h"
#include "swiftly_rows.h"
#include "swiftly_bluestein.h"
#include "swiftly_mixed.h"
#include <complex>

using namespace swf;

// ---------------------------------------------------------------------------
// error state: integer status + thread-local message (swiftly_hip_last_error)
int fail(int code, const char* fmt, ...);
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
    // compact copies of the float tables (swiftly_fft.h, "compact twiddle sections") by (log2(length), log2(points per lane))
    std::map<std::pair<int, int>, cx<float>*> twc_f;
    // Bluestein tables for transform lengths that are not a power of two (swiftly_bluestein.h), by length
    struct Blu {
        int logL = 0;
        cx<float>* chirp_f = nullptr;
        cx<float>* spec_f = nullptr;
        cx<double>* chirp_d = nullptr;
        cx<double>* spec_d = nullptr;
    };
    std::map<int64_t, Blu> blu;
    // lengths n = Q * 2^k, Q in {3, 5, 7, 9} (swiftly_mixed.h): exp(-2 pi i r / n), r < n
    struct Mixed {
        int Q = 0, logM = 0;
        cx<float>* tw_f = nullptr;
        cx<double>* tw_d = nullptr;
    };
    std::map<int64_t, Mixed> mixed;
    std::vector<void*> allocs;
    // float64 arithmetic in the column passes of the band pipelines (K2, K3 and their backward mirrors; complex64 data):
    // 0 (default: float32 arithmetic everywhere) | 1 (swiftly_hip_set_column_precision, env SWIFTLY_COL_F64)
    int col_f64 = 0;
    // which of the column-pass stages compute in float64 when col_f64 is set (SWIFTLY_COL_F64_STAGES, default 7 = all; for
    // the measured error / cost table of DESIGN.md section 2): 1 = pass A of a four-step, 2 = pass B, 4 = single-pass
    // transforms (K3 and its backward mirror)
    int col_f64_stages = 7;
    // two internal streams of the chunked four-step (col_transform): created on first use, under `chunk_mu` (the fork /
    // join events are per call: two host threads may drive one handle on different streams)
    hipStream_t chunk_st[2] = {nullptr, nullptr};
    std::mutex chunk_mu;
    // the caller's four-step workspace was last used by the UN-chunked path (on the caller's stream): the next chunked call
    // must fork its chunk streams behind that stream even when swiftly_hip_chain_chunk_streams is set (r5 advisor: a trailing
    // facet group below the chunk threshold, > 32 facets)
    std::atomic<void*> ws_plain_pending{nullptr};  // that workspace
    // re-laid-out load windows of the forward K1 (swiftly_rowpass.h), built on first use per facet offset
    Win4Cache win4;
};

// swiftly_hip_chain_chunk_streams (include/swiftly_hip.h): the calling thread's chunked four-step launches skip their fork
extern thread_local int g_chain_chunk_streams;

template <typename R>
inline const cx<R>* twiddles(const swiftly_hip* h, int logn);
template <>
inline const cx<float>* twiddles<float>(const swiftly_hip* h, int logn) {
    auto it = h->tw_f.find(logn);
    return it == h->tw_f.end() ? nullptr : it->second;
}
template <>
inline const cx<double>* twiddles<double>(const swiftly_hip* h, int logn) {
    auto it = h->tw_d.find(logn);
    return it == h->tw_d.end() ? nullptr : it->second;
}
inline const cx<float>* compact_twiddles(const swiftly_hip* h, int logn, int logp) {
    auto it = h->twc_f.find({logn, logp});
    return it == h->twc_f.end() ? nullptr : it->second;
}
template <typename R>
inline const R* invp(const swiftly_hip* h);
template <>
inline const float* invp<float>(const swiftly_hip* h) { return h->invp_f; }
template <>
inline const double* invp<double>(const swiftly_hip* h) { return h->invp_d; }
template <typename R>
inline const R* fnwin(const swiftly_hip* h);
template <>
inline const float* fnwin<float>(const swiftly_hip* h) { return h->fn_f; }
template <>
inline const double* fnwin<double>(const swiftly_hip* h) { return h->fn_d; }
// RAII: make `device` current for the duration of one ABI call and restore the caller's (torch's) current
// device afterwards.  Streams handed in by the caller belong to the handle's device.
struct DeviceGuard {
    int prev = -1, rc = 0;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) rc = (int)hipSetDevice(device);
        else prev = -1;  // nothing to restore
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
// Transforms of length >= 2^kTwoPassMinLog along a STRIDED axis (rows contiguous) are decomposed into
// two passes of short transforms so that every access is >= 128 B contiguous.
static const int kTwoPassMinLog = 9;

// column-tile passes (swiftly_abi.hip)
ColZ plain_colz();
// single-pass launch of length 2^logn: float64 arithmetic when the handle asks for it and the instance exists
inline void set_col_precision(const swiftly_hip* h, ColPassArgs& c, int logn) {
    c.f64 = 0;
    if (!h->col_f64 || !(h->col_f64_stages & 4) || !col_pass_f64_supported(logn) || c.gs) return;
    const cx<double>* t = twiddles<double>(h, logn);
    if (!t) return;
    c.f64 = 1;
    c.twd = t;
    c.twd_full = t;
}
int launch_col_checked(int lg, int mode, const ColPassArgs& args, const ColZ& cz, int outer, int nb, hipStream_t st);
// scratch handed down by an entry point for the duration of one ABI call on this host thread (see col_transform)
extern thread_local void* t_call_ws;
extern thread_local size_t t_call_ws_bytes;
struct CallWorkspace {
    CallWorkspace(void* p, size_t bytes) { t_call_ws = p; t_call_ws_bytes = p ? bytes : 0; }
    ~CallWorkspace() { t_call_ws = nullptr; t_call_ws_bytes = 0; }
};
int col_transform(swiftly_hip* h, int logn, const ColPassArgs& c, const ColZ& cz, int W, int nb, hipStream_t st,
                  void* ws = nullptr, size_t ws_bytes = 0, int qmul = 0, int qadd = 0, int full_n = 0);
// rows kernels / radix-Q pass (swiftly_abi.hip)
bool mixed_factor(int64_t n, int* Q, int* logM);

// Facet tables of the row-wise fused kernels (swiftly_sumfinish.h), grouped by off1: entries of one group are adjacent.
template <class Args>
static inline void fill_facet_groups(Args& a, const swiftly_hip* h, int64_t nfacets, const int64_t* facet_off0s,
                                     const int64_t* facet_off1s) {
    const int xM = (int)h->xM, m = (int)h->m;
    std::vector<int> order((size_t)nfacets);
    for (int f = 0; f < nfacets; f++) order[(size_t)f] = f;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return facet_off1s[x] < facet_off1s[y]; });
    a.ngroups = 0;
    for (int n = 0; n < nfacets; n++) {
        const int f = order[(size_t)n];
        if (n == 0 || facet_off1s[f] != facet_off1s[order[(size_t)n - 1]]) {
            a.gstart[a.ngroups] = n;
            a.gsp1[a.ngroups] = (int)floordiv(facet_off1s[f] * h->xM, h->N);
            a.ngroups++;
        }
        a.fidx[n] = f;
        const int sp0 = (int)floordiv(facet_off0s[f] * h->xM, h->N);
        a.base0[n] = pmod(xM / 2 - m / 2 + sp0, xM);  // first padded-subgrid row the facet contributes to
    }
    a.gstart[a.ngroups] = (int)nfacets;
}

// Rounds of the wave-parallel sum_finish_facets (SFWide): groups whose placement windows [start, start + m) on the ring
// of xM positions are mutually disjoint share a round (greedy, in group order: deterministic).
static inline void fill_group_rounds(SumFinishFacetArgs& a, const swiftly_hip* h) {
    const int xM = (int)h->xM, m = (int)h->m;
    std::vector<std::vector<int>> rounds;
    auto overlap = [&](int g1, int g2) {
        const int s1 = pmod(xM / 2 - m / 2 + a.gsp1[g1], xM), s2 = pmod(xM / 2 - m / 2 + a.gsp1[g2], xM);
        const int d = pmod(s2 - s1, xM);
        return d < m || xM - d < m;
    };
    for (int g = 0; g < a.ngroups; g++) {
        size_t r = 0;
        for (; r < rounds.size(); r++) {
            bool ok = true;
            for (int o : rounds[r]) ok = ok && !overlap(g, o);
            if (ok) break;
        }
        if (r == rounds.size()) rounds.emplace_back();
        rounds[r].push_back(g);
    }
    a.nrounds = (int)rounds.size();
    int k = 0;
    for (size_t r = 0; r < rounds.size(); r++) {
        a.rstart[r] = k;
        for (int g : rounds[r]) a.rgroup[k++] = g;
    }
    a.rstart[rounds.size()] = k;
}

#define CHECK_COMMON()                                                                       \
    if (!h || !in || !out) return fail(SWIFTLY_ERR_PARAM, "null argument");                  \
    DeviceGuard device_guard_(h->device);                                                    \
    if (device_guard_.rc) return fail(SWIFTLY_ERR_HIP, "hipSetDevice(%d) failed", h->device); \
    if (rows < 0) return fail(SWIFTLY_ERR_PARAM, "negative row count");                      \
    if (dtype != SWIFTLY_C64 && dtype != SWIFTLY_C128) return fail(SWIFTLY_ERR_PARAM, "bad dtype %d", dtype); \
    if (in_cs < 0 || out_cs < 0 || in_cs >= (int64_t(1) << 32) || out_cs >= (int64_t(1) << 32)) \
        return fail(SWIFTLY_ERR_PARAM, "column strides must be in [0, 2^32)");                 \
    if (rows > 0x7fffffff) return fail(SWIFTLY_ERR_PARAM, "too many rows");
#define CHECK_BATCH()                                                                        \
    if (nbatch < 0 || in_bs < 0 || out_bs < 0) return fail(SWIFTLY_ERR_PARAM, "bad batch description");
// The accumulating entry points read-modify-write their output non-atomically and run the batch items
// concurrently: items that share output elements would lose updates.
#define CHECK_ACCUMULATE_BATCH()                                                             \
    if (nbatch > 1 && out_bs == 0)                                                           \
        return fail(SWIFTLY_ERR_PARAM, "accumulating batch items must not share output elements (out_batch_stride = 0)");

#define DISPATCH(fn, ...) (dtype == SWIFTLY_C64 ? fn<float>(__VA_ARGS__) : fn<double>(__VA_ARGS__))
#define CHECK_FACET_SIZE()                                                                                     \
    if (facet_size <= 0 || facet_size >= h->yN)                                                                \
        return fail(SWIFTLY_ERR_PARAM, "facet size %lld must be in [1, yN_size - 1 = %lld]", (long long)facet_size, \
                    (long long)(h->yN - 1));
#define CHECK_SUBGRID_SIZE()                                                                                      \
    if (subgrid_size <= 0 || subgrid_size > h->xM)                                                                \
        return fail(SWIFTLY_ERR_PARAM, "subgrid size %lld must be in [1, xM_size = %lld]", (long long)subgrid_size, \
                    (long long)h->xM);


// ---------------------------------------------------------------------------------------------------------
// band buffers of the contiguous-axis-first pipeline
// half of a parity-split band buffer, in columns: a multiple of 16 (128 bytes) so that BOTH parity runs of a 64-column
// tile of the column pass start on a cache line (r2: (band_len + 1) / 2 = 5736 left every odd run 64 bytes off a line:
// 5 lines fetched per 4 lines' worth, FETCH_SIZE of K2 pass A 1.04 GB per wave against 0.83 GB)
static inline int64_t band_half_columns(int64_t band_len) { return (((band_len + 1) / 2) + 15) & ~int64_t(15); }
// Band layout of a handle: parity-split where the two-workgroup long-row kernel produces the band (yN >= 16384), else
// PLAIN (half = 0: logical column d of the band at physical column d; K1 is the generic contiguous-axis transform and
// keeps the whole padded axis).
static inline bool band_is_split(const swiftly_hip* h) { return h->log_yN >= 14 && h->log_yN <= 16; }
static inline int band_half_of(const swiftly_hip* h, int64_t band_len) {
    return band_is_split(h) ? (int)band_half_columns(band_len) : 0;
}
