// C ABI of libswiftly_hip.so, part 2: the fused and per-wave entry points the streaming classes use
// (include/swiftly_hip.h, sections "fused kernels", "contiguous-axis-first forward pipeline" and "backward pass with
// band accumulators").  Each one is a short launch sequence over the column-tile passes (swiftly_colpass.h) and the
// row-wise fused kernels (swiftly_sumfinish.h).
#include "swiftly_abi_internal.h"

// touched[d] = 1 for the band columns d of one wave's window (see accumulate_facet_columns)
__global__ void mark_columns_kernel(unsigned char* __restrict__ touched, int m, int rot, int base, int yN, int band_start,
                                    int band_len) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= m) return;
    int scol = base + ((col + rot) & (m - 1));  // m a power of two, yN any length: base < yN
    if (scol >= yN) scol -= yN;
    int d = scol - band_start;
    if (d < 0) d += yN;
    if (d < band_len) touched[d] = 1;
}
// zero the band columns no wave has written (rows x band_len, row stride `pitch`)
__global__ void zero_untouched_kernel(cx<float>* __restrict__ band, const unsigned char* __restrict__ touched, long long rows,
                                      long long pitch, int band_len) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= band_len || touched[d]) return;
    for (long long r = blockIdx.y; r < rows; r += gridDim.y) band[r * pitch + d] = cx<float>{0.f, 0.f};
}


extern "C" {

int swiftly_hip_sum_finish_rows(swiftly_hip_t* h, int dtype, const void* in, int64_t ngroups, int64_t in_group_stride,
                                int64_t in_batch_stride, int64_t in_row_stride, const int64_t* group_facet_offs,
                                void* out, int64_t out_batch_stride, int64_t out_row_stride,
                                const int64_t* subgrid_offs, int64_t subgrid_size, const void* mask,
                                int64_t mask_batch_stride, int64_t nbatch, void* stream) {
    if (!h || !in || !out || !group_facet_offs || !subgrid_offs) return fail(SWIFTLY_ERR_PARAM, "null argument");
    DeviceGuard device_guard_(h->device);
    CHECK_SUBGRID_SIZE();
    if (dtype != SWIFTLY_C64) return fail(SWIFTLY_ERR_UNSUPPORTED, "sum_finish_rows: complex64 only");
    if (ngroups <= 0 || ngroups > kSumFinishMaxGroups)
        return fail(SWIFTLY_ERR_UNSUPPORTED, "sum_finish_rows: 1..%d facet groups supported", kSumFinishMaxGroups);
    if (!sum_finish_supported(h->log_m, h->log_xM))
        return fail(SWIFTLY_ERR_UNSUPPORTED, "sum_finish_rows: (m, xM) = (%lld, %lld) not instantiated", (long long)h->m,
                    (long long)h->xM);
    if (nbatch <= 0) return 0;
    const int xM = (int)h->xM, xA = (int)subgrid_size;
    SumFinishArgs a;
    std::memset(&a, 0, sizeof a);
    a.in_gs = in_group_stride;
    a.in_bs = in_batch_stride;
    a.in_rs = in_row_stride;
    a.out_bs = out_batch_stride;
    a.out_rs = out_row_stride;
    a.nrows = xM;
    a.ngroups = (int)ngroups;
    a.xA = xA;
    for (int g = 0; g < ngroups; g++) a.sp[g] = (int)floordiv(group_facet_offs[g] * h->xM, h->N);
    a.fn = h->fn_f;
    a.mask_bs = mask ? mask_batch_stride : 0;
    a.tw_m = twiddles<float>(h, h->log_m);
    a.tw_x = twiddles<float>(h, h->log_xM);
    if (!a.tw_m || !a.tw_x) return fail(SWIFTLY_ERR_HIP, "internal: missing twiddle tables");
    for (int64_t b0 = 0; b0 < nbatch; b0 += kSumFinishMaxBatch) {
        const int nb = (int)std::min<int64_t>(kSumFinishMaxBatch, nbatch - b0);
        a.in = (const cx<float>*)in + b0 * in_batch_stride;
        a.out = (cx<float>*)out + b0 * out_batch_stride;
        a.mask = mask ? (const float*)mask + b0 * mask_batch_stride : nullptr;
        for (int b = 0; b < nb; b++) a.st_a[b] = pmod(-(xM / 2 - xA / 2 + subgrid_offs[b0 + b]), xM);
        int e = launch_sum_finish_rows(h->log_m, h->log_xM, a, nb, (hipStream_t)stream);
        if (e) return fail(SWIFTLY_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    }
    return 0;
}

int swiftly_hip_add_to_subgrid_from_columns(swiftly_hip_t* h, int dtype, const void* in, int64_t in_row_stride,
                                            int64_t in_facet_stride, int64_t nfacets, void* out,
                                            int64_t out_col_stride, int64_t out_batch_stride, int64_t facet_off0,
                                            int64_t nsub, const int64_t* subgrid_off1s, void* stream) {
    if (!h || !in || !out || !subgrid_off1s) return fail(SWIFTLY_ERR_PARAM, "null argument");
    DeviceGuard device_guard_(h->device);
    if (dtype != SWIFTLY_C64) return fail(SWIFTLY_ERR_UNSUPPORTED, "add_to_subgrid_from_columns: complex64 only");
    const int m = (int)h->m, xM = (int)h->xM, yN = (int)h->yN;
    if (h->log_m < kColPassMinLog || h->log_m > kColPassMaxLog || h->log_yN < 0)
        return fail(SWIFTLY_ERR_UNSUPPORTED, "add_to_subgrid_from_columns: contribution size %d not supported", m);
    if (nfacets <= 0 || nsub <= 0) return 0;
    if ((uint64_t)m * (uint64_t)in_row_stride + (uint64_t)yN >= (uint64_t(1) << 32) ||
        (uint64_t)xM * (uint64_t)out_col_stride + (uint64_t)m >= (uint64_t(1) << 32))
        return fail(SWIFTLY_ERR_PARAM, "strides too large for 32-bit offsets");
    const int64_t sp = floordiv(facet_off0 * h->xM, h->N);
    ColPassArgs c;
    std::memset(&c, 0, sizeof c);
    c.ncols = m;
    c.full_logn = h->log_m;
    c.in_pitch = (unsigned)in_row_stride;
    c.out_pitch = (unsigned)out_col_stride;
    c.ld_mul = c.st_mul = 1;
    c.ld_a = 0; c.ld_len = m; c.ld_c = 0; c.ld_mod = m;
    c.st_a = pmod(-sp, m); c.st_len = m; c.st_c = pmod(xM / 2 - m / 2 + sp, xM); c.st_mod = xM;
    c.st_win = h->fn_f;
    c.scale = 1.f;
    c.accumulate = 1;
    c.tw = twiddles<float>(h, h->log_m);
    if (!c.tw) return fail(SWIFTLY_ERR_HIP, "internal: missing twiddle table");
    c.cg_mod = m;
    c.cg_full = yN;
    c.in_bs_hi = in_facet_stride;
    c.in_bs = 0;
    c.out_bs = out_batch_stride;
    // batch item z = f*nb + b  (f: facet / group index, b: subgrid of this chunk); item (f, b) reads facet f's
    // column buffer and adds into out + (f*nsub + b0 + b) * out_batch_stride
    if (nfacets > kColZF) return fail(SWIFTLY_ERR_UNSUPPORTED, "add_to_subgrid_from_columns: too many facets per call");
    for (int64_t b0 = 0; b0 < nsub; b0 += kColZB) {
        const int nb = (int)std::min<int64_t>(kColZB, nsub - b0);
        ColZ cz = plain_colz();
        cz.flags = kZColGather;
        cz.nb = nb;
        for (int b = 0; b < nb; b++) {
            const int64_t s = floordiv(subgrid_off1s[b0 + b] * h->yN, h->N);
            cz.b_rot[b] = pmod(-s, m);
            cz.b_base[b] = pmod(yN / 2 - m / 2 + s, yN);
        }
        c.in = (const cx<float>*)in;
        c.in_bdiv = nb;
        c.out = (cx<float>*)out + b0 * out_batch_stride;
        c.out_bdiv = nb;
        c.out_bs_hi = nsub * out_batch_stride;
        if (int rc = launch_col_checked(h->log_m, 2, c, cz, 1, (int)nfacets * nb, (hipStream_t)stream)) return rc;
    }
    return 0;
}


int64_t swiftly_hip_band_columns(int64_t band_len) { return 2 * band_half_columns(band_len); }
int64_t swiftly_hip_band_columns_for(const swiftly_hip_t* h, int64_t band_len) {
    if (!h) return -1;
    return band_is_split(h) ? 2 * band_half_columns(band_len) : band_len;
}

} // extern "C" (helpers follow)
// K2 for yN = Q * 2^k (swiftly_mixed.h): per wave, the radix-Q pass over the window columns of every facet (the
// generic pass kernel with lanes along the columns: window gather through the modular row map, zero padding and facet
// offset through the load map, one batch item per facet) into a scratch [facet][j][y2][column], then the Q
// power-of-two sub-transforms along the strided axis with the column-tile passes, whose store side carries the row
// map of the wave with plain output index Q*k + j.  PLAIN band layout only (the parity-split layout belongs to the
// power-of-two long-row kernel).
static int prepare_facet_columns_mixed(swiftly_hip_t* h, const void* in, int64_t rows, int64_t in_row_stride,
                                       int64_t in_facet_stride, int64_t nfacets, const int64_t* facet_off0s,
                                       int64_t band_start, int64_t band_len, int64_t nwaves, const int64_t* wave_off1s,
                                       void* out, int64_t out_row_stride, int64_t out_facet_stride,
                                       int64_t out_wave_stride, const int32_t* rowmaps, int64_t rowmap_stride,
                                       void* stream, void* ws, size_t ws_bytes) {
    const int yN = (int)h->yN, m = (int)h->m;
    auto it = h->mixed.find(h->yN);
    if (it == h->mixed.end() || !it->second.tw_f)
        return fail(SWIFTLY_ERR_UNSUPPORTED, "prepare_facet_columns: padded facet size %d is neither a power of two nor Q * 2^k (Q = 3, 5, 7, 9)", yN);
    if (band_is_split(h)) return fail(SWIFTLY_ERR_UNSUPPORTED, "prepare_facet_columns: internal: split band layout");
    const int Q = it->second.Q, logM = it->second.logM;
    const long long M = 1ll << logM;
    if ((uint64_t)rows * (uint64_t)in_row_stride >= (uint64_t(1) << 32) || (uint64_t)yN * (uint64_t)out_row_stride >= (uint64_t(1) << 32) ||
        (uint64_t)yN * (uint64_t)m >= (uint64_t(1) << 32))
        return fail(SWIFTLY_ERR_PARAM, "strides too large for 32-bit offsets");
    const int lo = yN / 2 - (int)(rows / 2);
    hipStream_t st = (hipStream_t)stream;
    // workspace: [radix-Q scratch: nf * yN * m][four-step scratch of the sub-transforms: nf * M * m]
    const int per_f_cap = (int)std::min<int64_t>(kMaxBatch, kColZF);
    for (int64_t f0 = 0; f0 < nfacets; f0 += per_f_cap) {
        const int nf = (int)std::min<int64_t>(per_f_cap, nfacets - f0);
        const size_t radix_bytes = (size_t)nf * (size_t)yN * (size_t)m * sizeof(cx<float>);
        const size_t sub_bytes = (size_t)nf * (size_t)M * (size_t)m * sizeof(cx<float>) + 4096;
        void* own = nullptr;
        char* base = (char*)ws;
        if (!ws || ws_bytes < radix_bytes + sub_bytes) {
            HIP_TRY(hipMallocAsync(&own, radix_bytes + sub_bytes, st));
            base = (char*)own;
        }
        int rc = 0;
        for (int64_t w = 0; w < nwaves && !rc; w++) {
            const int64_t s1 = floordiv(wave_off1s[w] * h->yN, h->N);
            RowsArgs<float> a;
            std::memset(&a, 0, sizeof a);
            a.in = (const cx<float>*)in + f0 * in_facet_stride;
            a.in_rs = 1;                            // "rows" of the pass = the m window columns (contiguous)
            a.in_cs = (unsigned)in_row_stride;      // transform index = facet row
            a.in_bs = in_facet_stride;
            a.nrows = m;
            a.nbatch = nf;
            a.rowfast = 1;
            a.ld = AxisMap<float>{0, (int)rows, 0, (int)rows, nullptr, nullptr};  // window already applied by prepare_facet_band
            a.conj_ld = 1;
            a.scale = 1.f;
            a.rm_mod = m;
            a.rm_inner = pmod(-s1, m);
            a.rm_outer = pmod(yN / 2 - m / 2 + s1 - band_start, yN);  // plain band: physical column = logical - band_start
            a.rm_full = yN;
            a.full_n = yN;
            OffTab tab;
            tab.use = 1;
            for (int f = 0; f < nf; f++) tab.ld_a[f] = pmod(-(facet_off0s[f0 + f] + lo), yN);
            MixedArgs<float> X;
            std::memset(&X, 0, sizeof X);
            X.Q = Q; X.M = (int)M; X.n = yN;
            for (int r = 0; r < Q; r++) {
                const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)r / (long double)Q;
                X.wq[r] = cx<float>{(float)cosl(ang), (float)sinl(ang)};
            }
            X.tw_n = it->second.tw_f;
            X.scratch = (cx<float>*)base;
            X.s_row = 1; X.s_y = m; X.s_j = M * m; X.s_b = (long long)yN * m;
            int e = launch_mixed_pass(Q, a, tab, X, nf, st);
            if (e) { rc = fail(SWIFTLY_ERR_HIP, "kernel launch failed (radix-%d pass): %s", Q, hipGetErrorString((hipError_t)e)); break; }
            for (int j = 0; j < Q && !rc; j++) {
                ColPassArgs c;
                std::memset(&c, 0, sizeof c);
                c.ncols = m;
                c.in = (const cx<float>*)base + (long long)j * X.s_j;
                c.in_pitch = (unsigned)m;
                c.in_bs = X.s_b;
                c.out = (cx<float>*)out + f0 * out_facet_stride + w * out_wave_stride;
                c.out_pitch = (unsigned)out_row_stride;
                c.out_bs = out_facet_stride;
                c.ld_mul = c.st_mul = 1;
                c.st_a = 0; c.st_len = yN; c.st_c = 0; c.st_mod = yN;
                c.scale = (float)(1.0 / yN);
                c.conj_ld = 0; c.conj_st = 1;
                c.st_rowmap = rowmaps ? rowmaps + w * rowmap_stride : nullptr;
                c.st_rowmap_bs = 0;
                const int r2 = col_transform(h, logM, c, plain_colz(), m, nf, st, base + radix_bytes, sub_bytes, Q, j, yN);
                if (r2 == -1) rc = fail(SWIFTLY_ERR_UNSUPPORTED, "prepare_facet_columns: sub-transform length %lld not supported", M);
                else rc = r2;
            }
        }
        if (own) {
            hipError_t e2 = hipFreeAsync(own, st);
            if (!rc && e2 != hipSuccess) rc = fail(SWIFTLY_ERR_HIP, "hipFreeAsync: %s", hipGetErrorString(e2));
        }
        if (rc) return rc;
    }
    return 0;
}

// K2 for `nfacets` facets x `nwaves` waves: item (f, w) gathers the window of wave_off1s[w] from band buffer f and
// writes out + f*out_facet_stride + w*out_wave_stride through row map  rowmaps + w*rowmap_stride  (or none).
static int prepare_facet_columns_impl(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_row_stride,
                                      int64_t in_facet_stride, int64_t nfacets, const int64_t* facet_off0s,
                                      int64_t band_start, int64_t band_len, int64_t nwaves, const int64_t* wave_off1s,
                                      void* out, int64_t out_row_stride, int64_t out_facet_stride,
                                      int64_t out_wave_stride, const int32_t* rowmaps, int64_t rowmap_stride,
                                      void* stream, void* ws = nullptr, size_t ws_bytes = 0) {
    if (!h || !in || !out || !facet_off0s || !wave_off1s) return fail(SWIFTLY_ERR_PARAM, "null argument");
    if (dtype != SWIFTLY_C64) return fail(SWIFTLY_ERR_UNSUPPORTED, "prepare_facet_columns: complex64 only");
    const int yN = (int)h->yN, m = (int)h->m;
    if (h->log_m < 6) return fail(SWIFTLY_ERR_UNSUPPORTED, "prepare_facet_columns: sizes not supported");
    if (rows <= 0 || rows >= yN) return fail(SWIFTLY_ERR_PARAM, "facet size %lld must be in [1, yN_size - 1]", (long long)rows);
    if (band_len <= 0 || band_len > yN || band_start < 0 || band_start >= yN) return fail(SWIFTLY_ERR_PARAM, "bad band");
    if (nfacets <= 0 || nwaves <= 0) return 0;
    if (h->log_yN < 0)
        return prepare_facet_columns_mixed(h, in, rows, in_row_stride, in_facet_stride, nfacets, facet_off0s, band_start,
                                           band_len, nwaves, wave_off1s, out, out_row_stride, out_facet_stride,
                                           out_wave_stride, rowmaps, rowmap_stride, stream, ws, ws_bytes);
    // only `rows` input rows are ever read (the rest of the padded axis is zero fill)
    if ((uint64_t)rows * (uint64_t)in_row_stride >= (uint64_t(1) << 32) || (uint64_t)yN * (uint64_t)out_row_stride >= (uint64_t(1) << 32))
        return fail(SWIFTLY_ERR_PARAM, "strides too large for 32-bit offsets");
    const int lo = yN / 2 - (int)(rows / 2);
    ColPassArgs c;
    std::memset(&c, 0, sizeof c);
    c.ncols = m;
    c.full_logn = h->log_yN;
    c.in_pitch = (unsigned)in_row_stride;
    c.out_pitch = (unsigned)out_row_stride;
    c.ld_mul = c.st_mul = 1;
    c.ld_a = 0; c.ld_len = (int)rows; c.ld_c = 0; c.ld_mod = (int)rows;  // window already applied by prepare_facet_band
    c.st_a = 0; c.st_len = yN; c.st_c = 0; c.st_mod = yN;
    c.scale = (float)(1.0 / yN);
    c.conj_ld = c.conj_st = 1;
    c.cg_mod = m; c.cg_full = yN;
    c.cg_band_start = (int)band_start; c.cg_band_len = (int)band_len; c.cg_band_half = band_half_of(h, band_len);
    c.f64 = h->col_f64;  // (col_transform falls back to float32 where the instances do not exist)
    const int per_f = kColZF;  // facets per launch group (smaller groups: no gain, r4)
    // keep the four-step scratch of one launch group below ~4 GB
    const int64_t group_cap = ws ? (int64_t)ws_bytes : (int64_t(4) << 30);
    const int64_t per_w_cap = std::max<int64_t>(1, group_cap / ((int64_t)yN * m * 8) / std::min<int64_t>(per_f, nfacets));
    const int per_w = (int)std::min<int64_t>(kColZB, per_w_cap);
    for (int64_t f0 = 0; f0 < nfacets; f0 += per_f) {
        const int nf = (int)std::min<int64_t>(per_f, nfacets - f0);
        for (int64_t w0 = 0; w0 < nwaves; w0 += per_w) {
            const int nw = (int)std::min<int64_t>(per_w, nwaves - w0);
            ColZ cz = plain_colz();
            cz.flags = kZColGather | kZLoadAF;
            cz.nb = nw;
            for (int w = 0; w < nw; w++) {
                const int64_t s = floordiv(wave_off1s[w0 + w] * h->yN, h->N);
                cz.b_rot[w] = pmod(-s, m);
                cz.b_base[w] = pmod(yN / 2 - m / 2 + s, yN);
            }
            for (int f = 0; f < nf; f++) cz.f_lda[f] = pmod(-(facet_off0s[f0 + f] + lo), yN);
            // item z = f*nw + w reads band buffer f, writes out[f][w]
            c.in = (const cx<float>*)in + f0 * in_facet_stride;
            c.in_bdiv = nw; c.in_bs_hi = in_facet_stride; c.in_bs = 0;
            c.out = (cx<float>*)out + f0 * out_facet_stride + w0 * out_wave_stride;
            c.out_bdiv = nw; c.out_bs_hi = out_facet_stride; c.out_bs = out_wave_stride;
            c.st_rowmap = rowmaps ? rowmaps + w0 * rowmap_stride : nullptr;
            c.st_rowmap_bs = rowmaps ? rowmap_stride : 0;
            const int rc = col_transform(h, h->log_yN, c, cz, m, nf * nw, (hipStream_t)stream, ws, ws_bytes);
            if (rc == -1) return fail(SWIFTLY_ERR_UNSUPPORTED, "prepare_facet_columns: padded facet size %d not supported", yN);
            if (rc) return rc;
        }
    }
    return 0;
}

extern "C" {

int swiftly_hip_prepare_facet_columns(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_row_stride,
                                      int64_t in_facet_stride, int64_t nfacets, const int64_t* facet_off0s,
                                      int64_t band_start, int64_t band_len, int64_t subgrid_off1, void* out,
                                      int64_t out_row_stride, int64_t out_facet_stride, const int32_t* out_rowmap,
                                      void* stream) {
    if (!h) return fail(SWIFTLY_ERR_PARAM, "null argument");
    DeviceGuard device_guard_(h->device);
    return prepare_facet_columns_impl(h, dtype, in, rows, in_row_stride, in_facet_stride, nfacets, facet_off0s, band_start,
                                      band_len, 1, &subgrid_off1, out, out_row_stride, out_facet_stride, 0, out_rowmap, 0,
                                      stream);
}

int swiftly_hip_prepare_facet_columns_waves(swiftly_hip_t* h, int dtype, const void* in, int64_t rows,
                                            int64_t in_row_stride, int64_t in_facet_stride, int64_t nfacets,
                                            const int64_t* facet_off0s, int64_t band_start, int64_t band_len,
                                            int64_t nwaves, const int64_t* wave_off1s, void* out, int64_t out_row_stride,
                                            int64_t out_facet_stride, int64_t out_wave_stride, const int32_t* rowmaps,
                                            int64_t rowmap_stride, void* workspace, int64_t workspace_bytes,
                                            void* stream) {
    if (!h) return fail(SWIFTLY_ERR_PARAM, "null argument");
    DeviceGuard device_guard_(h->device);
    return prepare_facet_columns_impl(h, dtype, in, rows, in_row_stride, in_facet_stride, nfacets, facet_off0s, band_start,
                                      band_len, nwaves, wave_off1s, out, out_row_stride, out_facet_stride, out_wave_stride,
                                      rowmaps, rowmap_stride, stream, workspace, workspace ? (size_t)workspace_bytes : 0);
}

} // extern "C" (helper follows)
// out_offs / out_fstrides (optional, per subgrid): item (f, b) is written at out + out_offs[b] + f*out_fstrides[b]
// instead of out + f*out_facet_stride + b*out_sub_stride
static int transform_contributions_impl(swiftly_hip_t* h, int dtype, const void* in, int layout, int64_t in_row_stride,
                                        int64_t in_facet_stride, int64_t in_sub_stride, const int32_t* in_rowmap,
                                        int64_t band_start, int64_t band_len, int64_t nfacets,
                                        const int64_t* facet_off0s, int64_t nsub, const int64_t* subgrid_offs,
                                        void* out, int64_t out_facet_stride, int64_t out_sub_stride,
                                        const int64_t* out_offs, const int64_t* out_fstrides, void* stream) {
    if (!h || !in || !out || !facet_off0s) return fail(SWIFTLY_ERR_PARAM, "null argument");
    if (dtype != SWIFTLY_C64) return fail(SWIFTLY_ERR_UNSUPPORTED, "transform_contributions: complex64 only");
    if (layout < 0 || layout > 2) return fail(SWIFTLY_ERR_PARAM, "bad layout %d", layout);
    if (layout != 2 && !subgrid_offs) return fail(SWIFTLY_ERR_PARAM, "null argument");
    const int m = (int)h->m, yN = (int)h->yN;
    // (layout 0 gathers columns with masks of the padded facet size; the row-window layouts take any yN)
    if (h->log_m < kColPassMinLog || h->log_m > kColPassMaxLog || (layout == 0 && h->log_yN < 0))
        return fail(SWIFTLY_ERR_UNSUPPORTED, "transform_contributions: contribution size %d not supported", m);
    if (nfacets <= 0 || nsub <= 0) return 0;
    if ((uint64_t)yN * (uint64_t)in_row_stride + (uint64_t)yN >= (uint64_t(1) << 32))
        return fail(SWIFTLY_ERR_PARAM, "strides too large for 32-bit offsets");
    ColPassArgs c;
    std::memset(&c, 0, sizeof c);
    c.ncols = m;
    c.full_logn = h->log_m;
    c.in_pitch = (unsigned)in_row_stride;
    c.out_pitch = (unsigned)m;
    c.ld_mul = c.st_mul = 1;
    c.ld_a = 0; c.ld_len = m; c.ld_c = 0; c.ld_mod = m;
    c.st_a = 0; c.st_len = m; c.st_c = 0; c.st_mod = m;   // no placement: out[k] = Fn[k] * F[(k + s') mod m]
    c.st_win = h->fn_f;
    c.scale = 1.f;
    c.tw = twiddles<float>(h, h->log_m);
    if (!c.tw) return fail(SWIFTLY_ERR_HIP, "internal: missing twiddle table");
    if (layout == 0) {  // in[f] = [m, yN] column buffers: window gather along the contiguous axis
        c.cg_mod = m; c.cg_full = yN;
        c.cg_band_start = (int)band_start; c.cg_band_len = (int)band_len; c.cg_band_half = band_half_of(h, band_len);
    } else if (layout == 1) {  // in[f] = [kept rows of yN, m]: window gather along the strided axis
        c.ld_mod = yN;
        c.ld_rowmap = in_rowmap;
        c.tile32 = 1;  // K3 of the band pipeline: runs next to K2 of the following waves (col_pass.hip)
    }
    for (int64_t f0 = 0; f0 < nfacets; f0 += kColZF) {
        const int nf = (int)std::min<int64_t>(kColZF, nfacets - f0);
        for (int64_t b0 = 0; b0 < nsub; b0 += kColZB) {
            const int nb = (int)std::min<int64_t>(kColZB, nsub - b0);
            ColZ cz = plain_colz();
            cz.nb = nb;
            cz.flags = kZStoreAF;
            for (int f = 0; f < nf; f++) cz.f_sta[f] = pmod(-floordiv(facet_off0s[f0 + f] * h->xM, h->N), m);
            for (int b = 0; b < nb && layout != 2; b++) {
                const int64_t s = floordiv(subgrid_offs[b0 + b] * h->yN, h->N);
                if (layout == 0) {
                    cz.b_rot[b] = pmod(-s, m);
                    cz.b_base[b] = pmod(yN / 2 - m / 2 + s, yN);
                } else {
                    cz.b_lda[b] = pmod(-s, m);
                    cz.b_ldc[b] = pmod(yN / 2 - m / 2 + s, yN);
                }
            }
            if (layout == 0) cz.flags |= kZColGather;
            if (layout == 1) cz.flags |= kZLoadB;
            // item z = f*nb + b reads in + f*in_facet_stride (+ b*in_sub_stride for layout 2), writes out[f][b]
            c.in = (const cx<float>*)in + f0 * in_facet_stride + (layout == 2 ? b0 * in_sub_stride : 0);
            c.in_bdiv = nb; c.in_bs_hi = in_facet_stride; c.in_bs = layout == 2 ? in_sub_stride : 0;
            c.out = (cx<float>*)out + f0 * out_facet_stride + b0 * out_sub_stride;
            c.out_bdiv = nb; c.out_bs_hi = out_facet_stride; c.out_bs = out_sub_stride;
            if (out_offs) {
                cz.flags |= kZOutB;
                c.out = (cx<float>*)out;
                for (int b = 0; b < nb; b++) {
                    cz.b_out_fs[b] = out_fstrides[b0 + b];
                    cz.b_out_off[b] = out_offs[b0 + b] + f0 * out_fstrides[b0 + b];
                }
            }
            set_col_precision(h, c, h->log_m);
            if (int rc = launch_col_checked(h->log_m, 2, c, cz, 1, nf * nb, (hipStream_t)stream)) return rc;
        }
    }
    return 0;
}

extern "C" {
int swiftly_hip_transform_contributions(swiftly_hip_t* h, int dtype, const void* in, int layout, int64_t in_row_stride,
                                        int64_t in_facet_stride, int64_t in_sub_stride, const int32_t* in_rowmap,
                                        int64_t band_start, int64_t band_len, int64_t nfacets,
                                        const int64_t* facet_off0s, int64_t nsub, const int64_t* subgrid_offs,
                                        void* out, int64_t out_facet_stride, int64_t out_sub_stride, void* stream) {
    if (!h) return fail(SWIFTLY_ERR_PARAM, "null argument");
    DeviceGuard device_guard_(h->device);
    return transform_contributions_impl(h, dtype, in, layout, in_row_stride, in_facet_stride, in_sub_stride, in_rowmap,
                                        band_start, band_len, nfacets, facet_off0s, nsub, subgrid_offs, out,
                                        out_facet_stride, out_sub_stride, nullptr, nullptr, stream);
}

static int sum_finish_facets_impl(swiftly_hip_t* h, int dtype, const void* in, int64_t nfacets, int64_t in_facet_stride,
                                  int64_t in_sub_stride, int64_t in_row_stride, const int64_t* facet_off0s,
                                  const int64_t* facet_off1s, void* out, int64_t out_sub_stride, int64_t out_row_stride,
                                  const int64_t* subgrid_off1s, int64_t subgrid_size, const void* mask,
                                  int64_t mask_batch_stride, int64_t nsub, int placed, void* stream) {
    if (!h || !in || !out || !facet_off0s || !facet_off1s || !subgrid_off1s) return fail(SWIFTLY_ERR_PARAM, "null argument");
    DeviceGuard device_guard_(h->device);
    CHECK_SUBGRID_SIZE();
    if (dtype != SWIFTLY_C64) return fail(SWIFTLY_ERR_UNSUPPORTED, "sum_finish_facets: complex64 only");
    if (nfacets <= 0 || nfacets > kSumFinishMaxFacets)
        return fail(SWIFTLY_ERR_UNSUPPORTED, "sum_finish_facets: 1..%d facets supported", kSumFinishMaxFacets);
    if (!sum_finish_supported(h->log_m, h->log_xM))
        return fail(SWIFTLY_ERR_UNSUPPORTED, "sum_finish_facets: (m, xM) = (%lld, %lld) not instantiated", (long long)h->m,
                    (long long)h->xM);
    if (nsub <= 0) return 0;
    const int xM = (int)h->xM, xA = (int)subgrid_size;
    SumFinishFacetArgs a;
    std::memset(&a, 0, sizeof a);
    a.in_fs = in_facet_stride; a.in_bs = in_sub_stride; a.in_rs = in_row_stride;
    a.out_bs = out_sub_stride; a.out_rs = out_row_stride;
    a.nrows = xM;
    a.nfacets = (int)nfacets;
    a.xA = xA;
    fill_facet_groups(a, h, nfacets, facet_off0s, facet_off1s);
    fill_group_rounds(a, h);
    a.placed = placed ? 1 : 0;
    if (placed && h->log_xM >= 12)
        return fail(SWIFTLY_ERR_UNSUPPORTED, "axis-1-first pipeline: rows of %lld points run the wave-parallel sum_finish form, "
                    "which has no placed mode", (long long)h->xM);
    a.fn = h->fn_f;
    a.mask_bs = mask ? mask_batch_stride : 0;
    a.tw_m = twiddles<float>(h, h->log_m);
    a.tw_x = twiddles<float>(h, h->log_xM);
    if (!a.tw_m || !a.tw_x) return fail(SWIFTLY_ERR_HIP, "internal: missing twiddle tables");
    // table values from the compact copies (m-point transforms: 64 lanes each; rows: 64 lanes, 256 from 4096 points on)
    a.twc_m = compact_twiddles(h, h->log_m, h->log_m - 6);
    a.twc_x = compact_twiddles(h, h->log_xM, h->log_xM - (h->log_xM >= 12 ? 8 : 6));
    if (!a.twc_m || !a.twc_x) return fail(SWIFTLY_ERR_HIP, "internal: missing compact twiddle tables");
    for (int64_t b0 = 0; b0 < nsub; b0 += kSumFinishMaxBatch) {
        const int nb = (int)std::min<int64_t>(kSumFinishMaxBatch, nsub - b0);
        a.in = (const cx<float>*)in + b0 * in_sub_stride;
        a.out = (cx<float>*)out + b0 * out_sub_stride;
        a.mask = mask ? (const float*)mask + b0 * mask_batch_stride : nullptr;
        for (int b = 0; b < nb; b++) a.st_a[b] = pmod(-(xM / 2 - xA / 2 + subgrid_off1s[b0 + b]), xM);
        int e = launch_sum_finish_facets(h->log_m, h->log_xM, a, nb, (hipStream_t)stream);
        if (e) return fail(SWIFTLY_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    }
    return 0;
}

int swiftly_hip_sum_finish_facets(swiftly_hip_t* h, int dtype, const void* in, int64_t nfacets, int64_t in_facet_stride,
                                  int64_t in_sub_stride, int64_t in_row_stride, const int64_t* facet_off0s,
                                  const int64_t* facet_off1s, void* out, int64_t out_sub_stride, int64_t out_row_stride,
                                  const int64_t* subgrid_off1s, int64_t subgrid_size, const void* mask,
                                  int64_t mask_batch_stride, int64_t nsub, void* stream) {
    return sum_finish_facets_impl(h, dtype, in, nfacets, in_facet_stride, in_sub_stride, in_row_stride, facet_off0s, facet_off1s,
                                  out, out_sub_stride, out_row_stride, subgrid_off1s, subgrid_size, mask, mask_batch_stride, nsub,
                                  0, stream);
}

/* AXIS-1-FIRST pipeline (r6), step R: the contiguous-axis half of add_to_subgrid (core.py:255-285) on the rows of the K1
 * band buffers of all facets, for the wave `wave_off1`, BEFORE the strided-axis transforms (swiftly_sumfinish.h,
 * axis1_rows_kernel).  out[f] = [rows, m] in the parity-split layout of a band that is exactly the wave's window
 * (start (yN/2 - m/2 + s) mod yN, length m): hand it to prepare_facet_columns / wave_facet_side with that band. */
int swiftly_hip_finish_axis1_rows(swiftly_hip_t* h, int dtype, const void* bands, int64_t rows, int64_t band_row_stride,
                                  int64_t band_facet_stride, int64_t nfacets, const int64_t* facet_off1s,
                                  int64_t band_start, int64_t band_len, int64_t wave_off1, void* out,
                                  int64_t out_row_stride, int64_t out_facet_stride, void* stream) {
    if (!h || !bands || !out || !facet_off1s) return fail(SWIFTLY_ERR_PARAM, "null argument");
    DeviceGuard device_guard_(h->device);
    if (dtype != SWIFTLY_C64) return fail(SWIFTLY_ERR_UNSUPPORTED, "finish_axis1_rows: complex64 only");
    if (!band_is_split(h) || h->log_m < 7 || h->log_m > 10)
        return fail(SWIFTLY_ERR_UNSUPPORTED, "finish_axis1_rows: needs the parity-split band layout (yN_size 16384 .. 65536) and "
                    "m = 128 .. 1024");
    if (nfacets <= 0 || nfacets > kSumFinishMaxFacets)
        return fail(SWIFTLY_ERR_UNSUPPORTED, "finish_axis1_rows: 1..%d facets supported", kSumFinishMaxFacets);
    const int yN = (int)h->yN, m = (int)h->m;
    if (band_len <= 0 || band_len > yN || band_start < 0 || band_start >= yN)
        return fail(SWIFTLY_ERR_PARAM, "band [%lld, +%lld) is not a cyclic range of [0, %d)", (long long)band_start, (long long)band_len, yN);
    if (rows <= 0) return 0;
    Axis1RowsArgs a;
    std::memset(&a, 0, sizeof a);
    a.in = (const cx<float>*)bands; a.out = (cx<float>*)out;
    a.in_fs = band_facet_stride; a.in_rs = band_row_stride; a.out_fs = out_facet_stride; a.out_rs = out_row_stride;
    a.nrows = (int)rows; a.yN = yN;
    a.band_start = (int)band_start; a.band_len = (int)band_len; a.band_half = (int)band_half_columns(band_len);
    const int64_t s = floordiv(wave_off1 * h->yN, h->N);
    a.c0 = (int)pmod(yN / 2 - m / 2 + s, yN);
    a.s = (int)pmod(s, m);
    // every column of the window must lie inside the band
    if (pmod(a.c0 - band_start, yN) + m > band_len)
        return fail(SWIFTLY_ERR_PARAM, "finish_axis1_rows: the window of off1 = %lld is not inside the band", (long long)wave_off1);
    for (int64_t f = 0; f < nfacets; f++) a.sp[f] = (int)pmod(floordiv(facet_off1s[f] * h->xM, h->N), m);
    a.fn = h->fn_f;
    a.tw_m = twiddles<float>(h, h->log_m);
    a.twc_m = compact_twiddles(h, h->log_m, h->log_m - 6);
    if (!a.tw_m || !a.twc_m) return fail(SWIFTLY_ERR_HIP, "internal: missing twiddle tables");
    int e = launch_axis1_rows(h->log_m, a, (int)nfacets, (hipStream_t)stream);
    if (e) return fail(SWIFTLY_ERR_HIP, "kernel launch failed: %s", e > 0 ? hipGetErrorString((hipError_t)e) : "no instance");
    return 0;
}

/* Backward subgrid side, contiguous-axis half + axis-0 remainder (see swiftly_sumfinish.h): in[b] = [xM, xA] =
 * prepare_subgrid(axis 0) of subgrid b; out[f][b] = [m, m] = the contribution of subgrid b to facet f
 * (api_helper.prepare_and_split_subgrid, api_helper.py:115-139). */
int swiftly_hip_split_prepare_facets(swiftly_hip_t* h, int dtype, const void* in, int64_t in_sub_stride,
                                     int64_t in_row_stride, int64_t subgrid_size, int64_t nsub,
                                     const int64_t* subgrid_off1s, int64_t nfacets, const int64_t* facet_off0s,
                                     const int64_t* facet_off1s, void* out, int64_t out_facet_stride,
                                     int64_t out_sub_stride, void* stream) {
    if (!h || !in || !out || !facet_off0s || !facet_off1s || !subgrid_off1s) return fail(SWIFTLY_ERR_PARAM, "null argument");
    DeviceGuard device_guard_(h->device);
    CHECK_SUBGRID_SIZE();
    if (dtype != SWIFTLY_C64) return fail(SWIFTLY_ERR_UNSUPPORTED, "split_prepare_facets: complex64 only");
    if (nfacets <= 0 || nfacets > kSumFinishMaxFacets)
        return fail(SWIFTLY_ERR_UNSUPPORTED, "split_prepare_facets: 1..%d facets supported", kSumFinishMaxFacets);
    if (!sum_finish_supported(h->log_m, h->log_xM) || h->log_m > kColPassMaxLog)
        return fail(SWIFTLY_ERR_UNSUPPORTED, "split_prepare_facets: (m, xM) = (%lld, %lld) not instantiated", (long long)h->m,
                    (long long)h->xM);
    if (nsub <= 0) return 0;
    const int xM = (int)h->xM, xA = (int)subgrid_size, m = (int)h->m;
    SplitFacetArgs a;
    std::memset(&a, 0, sizeof a);
    a.in_bs = in_sub_stride; a.in_rs = in_row_stride;
    a.out_fs = out_facet_stride; a.out_bs = out_sub_stride; a.out_rs = m;
    a.nrows = xM;
    a.nfacets = (int)nfacets;
    a.xA = xA;
    fill_facet_groups(a, h, nfacets, facet_off0s, facet_off1s);
    a.fn = h->fn_f;
    a.tw_m = twiddles<float>(h, h->log_m);
    a.tw_x = twiddles<float>(h, h->log_xM);
    if (!a.tw_m || !a.tw_x) return fail(SWIFTLY_ERR_HIP, "internal: missing twiddle tables");
    // table values from the compact copies (m-point transforms: 64 lanes each; rows: 64 lanes, 256 from 4096 points on)
    a.twc_m = compact_twiddles(h, h->log_m, h->log_m - 6);
    a.twc_x = compact_twiddles(h, h->log_xM, h->log_xM - (h->log_xM >= 12 ? 8 : 6));
    if (!a.twc_m || !a.twc_x) return fail(SWIFTLY_ERR_HIP, "internal: missing compact twiddle tables");
    for (int64_t b0 = 0; b0 < nsub; b0 += kSumFinishMaxBatch) {
        const int nb = (int)std::min<int64_t>(kSumFinishMaxBatch, nsub - b0);
        a.in = (const cx<float>*)in + b0 * in_sub_stride;
        a.out = (cx<float>*)out + b0 * out_sub_stride;
        for (int b = 0; b < nb; b++) a.ld_a[b] = pmod(-(xM / 2 - xA / 2 + subgrid_off1s[b0 + b]), xM);
        int e = launch_split_prepare_facets(h->log_m, h->log_xM, a, nb, (hipStream_t)stream);
        if (e) return fail(SWIFTLY_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    }
    // axis-0 remainder of extract_from_subgrid, in place: out[f][b][:, j] = cifft_m( Fn[k] * E[f][b][(k - s'0_f) ..., j] )
    ColPassArgs c;
    std::memset(&c, 0, sizeof c);
    c.ncols = m;
    c.full_logn = h->log_m;
    c.in_pitch = c.out_pitch = (unsigned)m;
    c.ld_mul = c.st_mul = 1;
    c.ld_a = 0; c.ld_len = m; c.ld_c = 0; c.ld_mod = m;
    c.ld_win = h->fn_f;
    c.st_a = 0; c.st_len = m; c.st_c = 0; c.st_mod = m;
    c.conj_ld = c.conj_st = 1;
    c.scale = 1.f / (float)m;
    c.tw = a.tw_m;
    for (int64_t f0 = 0; f0 < nfacets; f0 += kColZF) {
        const int nf = (int)std::min<int64_t>(kColZF, nfacets - f0);
        for (int64_t b0 = 0; b0 < nsub; b0 += kColZB) {
            const int nb = (int)std::min<int64_t>(kColZB, nsub - b0);
            ColZ cz = plain_colz();
            cz.nb = nb;
            cz.flags = kZLoadAF;
            for (int f = 0; f < nf; f++) cz.f_lda[f] = pmod(-floordiv(facet_off0s[f0 + f] * h->xM, h->N), m);
            c.in = (const cx<float>*)out + f0 * out_facet_stride + b0 * out_sub_stride;
            c.in_bdiv = nb; c.in_bs_hi = out_facet_stride; c.in_bs = out_sub_stride;
            c.out = (cx<float>*)out + f0 * out_facet_stride + b0 * out_sub_stride;
            c.out_bdiv = nb; c.out_bs_hi = out_facet_stride; c.out_bs = out_sub_stride;
            set_col_precision(h, c, h->log_m);
            if (int rc = launch_col_checked(h->log_m, 2, c, cz, 1, nf * nb, (hipStream_t)stream)) return rc;
        }
    }
    return 0;
}

/* The whole subgrid side of one backward wave natively, without stream-ordered allocations: prepare_subgrid along
 * axis 0 (four-step through `work`) + split_prepare_facets.  work: device scratch of >= 2 * nsub * xM * subgrid_size
 * complex64 elements (first half: tmp[nsub][xM][subgrid_size], second half: four-step scratch). */
int swiftly_hip_wave_split_subgrids(swiftly_hip_t* h, int dtype, const void* subgrids, int64_t subgrid_size, int64_t nsub,
                                    const int64_t* subgrid_off0s, const int64_t* subgrid_off1s, int64_t nfacets,
                                    const int64_t* facet_off0s, const int64_t* facet_off1s, void* work,
                                    int64_t work_elems, void* out, int64_t out_facet_stride, int64_t out_sub_stride,
                                    void* stream) {
    if (!h || !subgrids || !work || !out || !subgrid_off0s || !subgrid_off1s) return fail(SWIFTLY_ERR_PARAM, "null argument");
    DeviceGuard device_guard_(h->device);
    CHECK_SUBGRID_SIZE();
    if (dtype != SWIFTLY_C64) return fail(SWIFTLY_ERR_UNSUPPORTED, "wave_split_subgrids: complex64 only");
    if (nsub <= 0 || nfacets <= 0) return 0;
    const int xM = (int)h->xM, xA = (int)subgrid_size;
    const int64_t half = nsub * (int64_t)xM * xA;
    if (work_elems < 2 * half) return fail(SWIFTLY_ERR_PARAM, "work holds %lld elements, %lld needed", (long long)work_elems, (long long)(2 * half));
    cx<float>* tmp = (cx<float>*)work;
    ColPassArgs c;
    std::memset(&c, 0, sizeof c);
    c.ncols = xA;
    c.full_logn = h->log_xM;
    c.in_pitch = c.out_pitch = (unsigned)xA;
    c.ld_mul = c.st_mul = 1;
    c.ld_a = 0; c.ld_len = xA; c.ld_c = 0; c.ld_mod = xA;
    c.st_a = 0; c.st_len = xM; c.st_c = 0; c.st_mod = xM;
    c.scale = 1.f;
    for (int64_t b0 = 0; b0 < nsub; b0 += kColZB) {
        const int nb = (int)std::min<int64_t>(kColZB, nsub - b0);
        ColZ cz = plain_colz();
        cz.nb = nb;
        cz.flags = kZLoadB;
        for (int b = 0; b < nb; b++) {
            cz.b_lda[b] = pmod(-(xM / 2 - xA / 2 + subgrid_off0s[b0 + b]), xM);
            cz.b_ldc[b] = 0;
        }
        c.in = (const cx<float>*)subgrids + b0 * (int64_t)xA * xA;
        c.in_bs = (long long)xA * xA;
        c.out = tmp + b0 * (int64_t)xM * xA;
        c.out_bs = (long long)xM * xA;
        const int rc = col_transform(h, h->log_xM, c, cz, xA, nb, (hipStream_t)stream, tmp + half, (size_t)half * sizeof(cx<float>));
        if (rc == -1) return fail(SWIFTLY_ERR_UNSUPPORTED, "wave_split_subgrids: padded subgrid size %d not supported", xM);
        if (rc) return rc;
    }
    return swiftly_hip_split_prepare_facets(h, dtype, tmp, (int64_t)xM * xA, xA, subgrid_size, nsub, subgrid_off1s, nfacets,
                                            facet_off0s, facet_off1s, out, out_facet_stride, out_sub_stride, stream);
}


/* One forward wave in two calls (the whole launch sequence runs natively: per-wave host work is two ABI calls). */
int swiftly_hip_wave_facet_side(swiftly_hip_t* h, int dtype, const void* bands, int64_t rows, int64_t band_row_stride,
                                int64_t band_facet_stride, int64_t nfacets, const int64_t* facet_off0s,
                                int64_t band_start, int64_t band_len, int64_t wave_off1, const int32_t* rowmap,
                                int64_t n_rows, void* q_work, int64_t q_facet_stride, int compute_q, int64_t nsub,
                                const int64_t* sub_off0s, void* g_out,
                                int64_t g_facet_stride, int64_t g_sub_stride, const int64_t* g_offsets,
                                const int64_t* g_facet_strides, void* scratch, int64_t scratch_bytes, void* stream) {
    if (!h || (!bands && compute_q) || !q_work || !g_out || !facet_off0s || !sub_off0s) return fail(SWIFTLY_ERR_PARAM, "null argument");
    if (nfacets <= 0 || nsub <= 0) return 0;
    const int64_t m = h->m;
    if (n_rows <= 0 || n_rows > h->yN || q_facet_stride < n_rows * m)
        return fail(SWIFTLY_ERR_PARAM, "bad row count %lld / facet stride %lld", (long long)n_rows, (long long)q_facet_stride);
    // K2: Q[f] = [n_rows, m] (skipped when the caller still holds the wave's Q: compute_q = 0)
    DeviceGuard device_guard_(h->device);
    if (compute_q) {
        int rc = prepare_facet_columns_impl(h, dtype, bands, rows, band_row_stride, band_facet_stride, nfacets, facet_off0s,
                                            band_start, band_len, 1, &wave_off1, q_work, m, q_facet_stride, 0, rowmap, 0,
                                            stream, scratch, scratch ? (size_t)scratch_bytes : 0);
        if (rc) return rc;
    }
    // K3 + K4a from Q (layout 1)
    return transform_contributions_impl(h, dtype, q_work, 1, m, q_facet_stride, 0, rowmap, 0, 0, nfacets, facet_off0s, nsub,
                                        sub_off0s, g_out, g_facet_stride, g_sub_stride, g_offsets, g_facet_strides, stream);
}

static int wave_subgrid_side_impl(swiftly_hip_t* h, int dtype, const void* g, int64_t nfacets, int64_t g_facet_stride,
                                  int64_t g_sub_stride, const int64_t* facet_off0s, const int64_t* facet_off1s,
                                  int64_t nsub, const int64_t* sub_off0s, const int64_t* sub_off1s, int64_t subgrid_size,
                                  const void* mask0, int64_t mask0_bs, const void* mask1, int64_t mask1_bs,
                                  void* tmp_work, void* out, void* scratch, int64_t scratch_bytes, int placed, void* stream) {
    if (!h || !g || !tmp_work || !out || !sub_off0s || !sub_off1s) return fail(SWIFTLY_ERR_PARAM, "null argument");
    if (nsub <= 0) return 0;
    const int64_t m = h->m, xM = h->xM, xA = subgrid_size;
    // K4b + K5a: tmp[b] = [xM, xA]
    int rc = sum_finish_facets_impl(h, dtype, g, nfacets, g_facet_stride, g_sub_stride, m, facet_off0s, facet_off1s,
                                    tmp_work, xM * xA, xA, sub_off1s, subgrid_size, mask1, mask1_bs, nsub, placed, stream);
    if (rc) return rc;
    // K5b: finish_subgrid along axis 0 (strided): rows of the op = xA columns
    CallWorkspace call_ws(scratch, scratch ? (size_t)scratch_bytes : 0);
    return swiftly_hip_finish_subgrid_batch(h, dtype, tmp_work, xA, 1, xA, out, 1, xA, 0, subgrid_size, mask0, nsub,
                                            xM * xA, xA * xA, sub_off0s, mask0 ? mask0_bs : 0, stream);
}

int swiftly_hip_wave_subgrid_side(swiftly_hip_t* h, int dtype, const void* g, int64_t nfacets, int64_t g_facet_stride,
                                  int64_t g_sub_stride, const int64_t* facet_off0s, const int64_t* facet_off1s,
                                  int64_t nsub, const int64_t* sub_off0s, const int64_t* sub_off1s, int64_t subgrid_size,
                                  const void* mask0, int64_t mask0_bs, const void* mask1, int64_t mask1_bs,
                                  void* tmp_work, void* out, void* scratch, int64_t scratch_bytes, void* stream) {
    return wave_subgrid_side_impl(h, dtype, g, nfacets, g_facet_stride, g_sub_stride, facet_off0s, facet_off1s, nsub, sub_off0s,
                                  sub_off1s, subgrid_size, mask0, mask0_bs, mask1, mask1_bs, tmp_work, out, scratch, scratch_bytes,
                                  0, stream);
}

/* wave_subgrid_side of the AXIS-1-FIRST pipeline: the blocks g[f][b] come from band buffers that went through
 * swiftly_hip_finish_axis1_rows, i.e. their rows already are Fn * cfft_m along the contiguous axis: sum_finish_facets
 * places and sums them without its m-point transforms. */
int swiftly_hip_wave_subgrid_side_placed(swiftly_hip_t* h, int dtype, const void* g, int64_t nfacets, int64_t g_facet_stride,
                                         int64_t g_sub_stride, const int64_t* facet_off0s, const int64_t* facet_off1s,
                                         int64_t nsub, const int64_t* sub_off0s, const int64_t* sub_off1s,
                                         int64_t subgrid_size, const void* mask0, int64_t mask0_bs, const void* mask1,
                                         int64_t mask1_bs, void* tmp_work, void* out, void* scratch, int64_t scratch_bytes,
                                         void* stream) {
    return wave_subgrid_side_impl(h, dtype, g, nfacets, g_facet_stride, g_sub_stride, facet_off0s, facet_off1s, nsub, sub_off0s,
                                  sub_off1s, subgrid_size, mask0, mask0_bs, mask1, mask1_bs, tmp_work, out, scratch, scratch_bytes,
                                  1, stream);
}


int swiftly_hip_accumulate_facet_columns(swiftly_hip_t* h, int dtype, const void* parts, int64_t part_row_stride,
                                         int64_t nchunks, const int64_t* chunk_offsets,
                                         const int64_t* chunk_facet_strides, const int32_t* row_sources,
                                         int64_t nfacets, const int64_t* facet_off0s, int64_t facet_size,
                                         const float* masks, int64_t subgrid_off1, void* bands, int64_t band_row_stride,
                                         int64_t band_facet_stride, int64_t band_start, int64_t band_len,
                                         unsigned char* touched, void* workspace, int64_t workspace_bytes,
                                         void* stream) {
    if (!h || !parts || !bands || !chunk_offsets || !chunk_facet_strides || !row_sources || !facet_off0s)
        return fail(SWIFTLY_ERR_PARAM, "null argument");
    DeviceGuard device_guard_(h->device);
    if (dtype != SWIFTLY_C64) return fail(SWIFTLY_ERR_UNSUPPORTED, "accumulate_facet_columns: complex64 only");
    CHECK_FACET_SIZE();
    const int yN = (int)h->yN, m = (int)h->m;
    const swiftly_hip::Mixed* mx = nullptr;  // yN = Q * 2^k: radix-Q pass with the gather-sum load + sub-transforms
    if (h->log_yN < 0) {
        auto it = h->mixed.find(h->yN);
        if (it != h->mixed.end() && it->second.tw_f && !band_is_split(h)) mx = &it->second;
    }
    if ((h->log_yN < 0 && !mx) || h->log_m < 0 || h->log_yN > 18)
        return fail(SWIFTLY_ERR_UNSUPPORTED, "accumulate_facet_columns: sizes not supported (yN a power of two or Q * 2^k, m a power of two)");
    if (nchunks <= 0 || nchunks > kColZC) return fail(SWIFTLY_ERR_PARAM, "1..%d source chunks", kColZC);
    if (band_len <= 0 || band_len > yN || band_start < 0 || band_start >= yN) return fail(SWIFTLY_ERR_PARAM, "bad band");
    if (nfacets <= 0) return 0;
    if ((uint64_t)facet_size * (uint64_t)band_row_stride >= (uint64_t(1) << 32) ||
        (uint64_t)part_row_stride << kGsRowBits >= (uint64_t(1) << 32))
        return fail(SWIFTLY_ERR_PARAM, "strides too large for 32-bit offsets");
    const int lo = yN / 2 - (int)(facet_size / 2);
    ColPassArgs c;
    std::memset(&c, 0, sizeof c);
    c.ncols = m;
    c.full_logn = h->log_yN;
    c.in = (const cx<float>*)parts;
    c.in_pitch = (unsigned)part_row_stride;
    c.out_pitch = (unsigned)band_row_stride;
    c.ld_mul = c.st_mul = 1;
    c.ld_a = 0; c.ld_len = yN; c.ld_c = 0; c.ld_mod = yN;
    c.ld_rowmap = row_sources; c.gs = 1;
    c.f64 = h->col_f64;
    c.st_a = 0; c.st_len = (int)facet_size; c.st_c = 0; c.st_mod = (int)facet_size;
    c.st_win = masks; c.st_win_bs = masks ? facet_size : 0;
    c.st_win2 = h->invp_f + lo;
    c.scale = 1.f;
    c.accumulate = 1;
    c.touched = touched;
    c.cg_mod = m; c.cg_full = yN;
    c.cg_band_start = (int)band_start; c.cg_band_len = (int)band_len; c.cg_band_half = 0;
    const int64_t cap = workspace ? workspace_bytes : (int64_t(4) << 30);
    const int64_t per_item = mx ? ((int64_t)yN + (int64_t(1) << mx->logM)) * m * 8 + 4096 : (int64_t)yN * m * 8;
    const int per_f = (int)std::max<int64_t>(1, std::min<int64_t>(kColZF, cap / per_item));
    const int64_t s1 = floordiv(subgrid_off1 * h->yN, h->N);
    for (int64_t f0 = 0; f0 < nfacets; f0 += per_f) {
        const int nf = (int)std::min<int64_t>(per_f, nfacets - f0);
        ColZ cz = plain_colz();
        cz.flags = kZColScatter | kZStoreAF;
        cz.nb = 1;
        cz.b_rot[0] = pmod(-s1, m);
        cz.b_base[0] = pmod(yN / 2 - m / 2 + s1, yN);
        for (int f = 0; f < nf; f++) cz.f_sta[f] = pmod(-(lo + facet_off0s[f0 + f]), yN);
        for (int k = 0; k < nchunks; k++) {
            cz.c_base[k] = chunk_offsets[k] + f0 * chunk_facet_strides[k];
            cz.c_fs[k] = chunk_facet_strides[k];
        }
        c.out = (cx<float>*)bands + f0 * band_facet_stride;
        c.out_bs = band_facet_stride;
        if (masks) c.st_win = masks + f0 * facet_size;
        int rc = 0;
        if (!mx) {
            rc = col_transform(h, h->log_yN, c, cz, m, nf, (hipStream_t)stream, workspace,
                               workspace ? (size_t)workspace_bytes : 0);
        } else {
            // workspace: [radix-Q scratch nf * yN * m][four-step scratch of the sub-transforms nf * M * m]
            const int Q = mx->Q, logM = mx->logM;
            const long long M = 1ll << logM;
            const size_t radix_bytes = (size_t)nf * (size_t)yN * (size_t)m * sizeof(cx<float>);
            const size_t sub_bytes = (size_t)nf * (size_t)M * (size_t)m * sizeof(cx<float>) + 4096;
            hipStream_t st = (hipStream_t)stream;
            void* own = nullptr;
            char* base = (char*)workspace;
            if (!workspace || (size_t)workspace_bytes < radix_bytes + sub_bytes) {
                HIP_TRY(hipMallocAsync(&own, radix_bytes + sub_bytes, st));
                base = (char*)own;
            }
            MixedGsArgs g;
            std::memset(&g, 0, sizeof g);
            g.in = (const cx<float>*)parts;
            g.in_pitch = (unsigned)part_row_stride;
            g.rowmap = row_sources;
            g.ncols = m;
            for (int k = 0; k < nchunks; k++) {
                g.c_base[k] = cz.c_base[k];
                g.c_fs[k] = cz.c_fs[k];
            }
            MixedArgs<float> X;
            std::memset(&X, 0, sizeof X);
            X.Q = Q; X.M = (int)M; X.n = yN;
            for (int r = 0; r < Q; r++) {
                const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)r / (long double)Q;
                X.wq[r] = cx<float>{(float)cosl(ang), (float)sinl(ang)};
            }
            X.tw_n = mx->tw_f;
            X.scratch = (cx<float>*)base;
            X.s_row = 1; X.s_y = m; X.s_j = M * m; X.s_b = (long long)yN * m;
            int e = launch_mixed_gs_pass(Q, g, X, nf, st);
            if (e) rc = fail(SWIFTLY_ERR_HIP, "kernel launch failed (radix-%d gather-sum pass): %s", Q, hipGetErrorString((hipError_t)e));
            for (int j = 0; j < Q && !rc; j++) {
                ColPassArgs cs = c;  // the store side of the primitive; plain load from the pass's scratch
                cs.in = (const cx<float>*)base + (long long)j * X.s_j;
                cs.in_pitch = (unsigned)m;
                cs.in_bs = X.s_b; cs.in_bdiv = 0; cs.in_bs_hi = 0;
                cs.ld_rowmap = nullptr; cs.gs = 0;
                rc = col_transform(h, logM, cs, cz, m, nf, st, base + radix_bytes, sub_bytes, Q, j, yN);
            }
            if (own) {
                hipError_t e2 = hipFreeAsync(own, st);
                if (!rc && e2 != hipSuccess) rc = fail(SWIFTLY_ERR_HIP, "hipFreeAsync: %s", hipGetErrorString(e2));
            }
        }
        if (rc == -1) return fail(SWIFTLY_ERR_UNSUPPORTED, "accumulate_facet_columns: padded facet size %d not supported", yN);
        if (rc) return rc;
    }
    if (touched) {
        hipLaunchKernelGGL(mark_columns_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, (hipStream_t)stream, touched,
                           m, pmod(-s1, m), pmod(yN / 2 - m / 2 + s1, yN), yN, (int)band_start, (int)band_len);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

int swiftly_hip_band_zero_untouched(swiftly_hip_t* h, int dtype, void* bands, int64_t rows, int64_t band_row_stride,
                                    int64_t band_len, const unsigned char* touched, void* stream) {
    if (!h || !bands || !touched) return fail(SWIFTLY_ERR_PARAM, "null argument");
    DeviceGuard device_guard_(h->device);
    if (dtype != SWIFTLY_C64) return fail(SWIFTLY_ERR_UNSUPPORTED, "band_zero_untouched: complex64 only");
    if (rows <= 0 || band_len <= 0) return 0;
    dim3 grid((unsigned)((band_len + 63) / 64), (unsigned)std::min<int64_t>(rows, 1024));
    hipLaunchKernelGGL(zero_untouched_kernel, grid, dim3(64), 0, (hipStream_t)stream, (cx<float>*)bands, touched,
                       (long long)rows, (long long)band_row_stride, (int)band_len);
    HIP_TRY(hipGetLastError());
    return 0;
}


}  // extern "C"
