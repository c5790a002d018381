/*
 * swiftly_hip.h -- C ABI of libswiftly_hip.so, the MI355X (gfx950) native
 * implementation of the SwiFTly facet<->subgrid primitives.
 *
 * Drop-in boundary.  The reference (ska_sdp_exec_swiftly 1.0.0) reaches its
 * native backend through
 *     ska_sdp_func.fourier_transforms.swiftly.Swiftly(N, yN_size, xM_size, W)
 * (src/ska_sdp_exec_swiftly/fourier_transform/core.py:508-510) and then calls
 * one method per primitive on 2-D arrays, always along the LAST axis, passing
 * transposed views for axis 0 (core.py:577-630).  Every entry point below
 * replaces exactly one of those methods and keeps its contract: a batch of
 * `rows` independent 1-D problems along an axis with arbitrary element
 * strides (so "pass a .T view" becomes "swap the two strides"), overwrite
 * for the operations the reference allocates with numpy.empty and
 * ACCUMULATE for the ones it allocates with numpy.zeros (core.py:685, 714,
 * 743, 871, 896, 922).
 *
 * Conventions
 *  - plain C, no C++/torch types; all array pointers are DEVICE pointers
 *    (HIP), `stream` is a hipStream_t passed as void* (NULL = default stream);
 *    work is enqueued asynchronously on that stream.
 *  - dtype: SWIFTLY_C64 (interleaved float re,im) or SWIFTLY_C128.
 *  - element (r, i) of an array lives at base + r*row_stride + i*col_stride,
 *    strides counted in complex elements.  (transform length)*col_stride
 *    must be < 2^32.
 *  - offsets are in full-resolution pixels exactly as in the reference
 *    (facet_off a multiple of N/xM, subgrid_off a multiple of N/yN); they may
 *    be negative or >= N.
 *  - return value 0 = success; non-zero = failure, message available from
 *    swiftly_hip_last_error() (thread local).  Nothing throws across the ABI.
 *    SWIFTLY_ERR_UNSUPPORTED is returned for transform lengths that are not
 *    a power of two in [8, 32768] (complex64) / [8, 8192] (complex128).
 *  - a handle is immutable after creation and may be used concurrently from
 *    several host threads / streams (the reference scatters one core object
 *    to all Dask worker threads, api.py:145-147).
 */
#ifndef SWIFTLY_HIP_H
#define SWIFTLY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct swiftly_hip swiftly_hip_t;

enum { SWIFTLY_C64 = 0, SWIFTLY_C128 = 1 };
enum {
    SWIFTLY_OK = 0,
    SWIFTLY_ERR_PARAM = 1,       /* -> ValueError in the Python mirror */
    SWIFTLY_ERR_UNSUPPORTED = 2, /* -> NotImplementedError */
    SWIFTLY_ERR_HIP = 3          /* -> RuntimeError */
};

const char* swiftly_hip_last_error(void);
int swiftly_hip_version(void);
/* Number of visible HIP devices (0 when there is no GPU; never fails). */
int swiftly_hip_device_count(void);

/*
 * Replaces Swiftly(N, yN_size, xM_size, W) (core.py:508-510) and the parameter
 * checks of core.py:55-74.  `pswf` is the length-yN host array
 * pro_ang1(0, 0, pi*W/2, 2*(k - yN/2)/yN) with pswf[0] = 0 (core.py:119-150);
 * the window constants Fb = 1/pswf[1:] and Fn = pswf[(yN/2)%(N/xM)::N/xM]
 * (core.py:104-117) are derived from it and uploaded in both precisions.
 */
int swiftly_hip_create(swiftly_hip_t** out, int64_t N, int64_t yN_size, int64_t xM_size, double W,
                       const double* pswf, int device);
void swiftly_hip_destroy(swiftly_hip_t* h);
int64_t swiftly_hip_contribution_size(const swiftly_hip_t* h); /* xM*yN/N, core.py:48 */

/* -- facet -> subgrid ------------------------------------------------------ */

/* Swiftly.prepare_facet(in[rows, facet_size], out[rows, yN], facet_off)
 * (core.py:686; numpy form core.py:212-222).  Overwrites out. */
int swiftly_hip_prepare_facet(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t facet_size,
                              int64_t in_row_stride, int64_t in_col_stride, void* out, int64_t out_row_stride,
                              int64_t out_col_stride, int64_t facet_off, void* stream);

/* Swiftly.extract_from_facet(in[rows, yN], out[rows, m], subgrid_off)
 * (core.py:715; numpy form core.py:243-253).  Overwrites out.  Bit-exact. */
int swiftly_hip_extract_from_facet(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_row_stride,
                                   int64_t in_col_stride, void* out, int64_t out_row_stride, int64_t out_col_stride,
                                   int64_t subgrid_off, void* stream);

/* Swiftly.add_to_subgrid(in[rows, m], out[rows, xM], facet_off)
 * (core.py:744; numpy form core.py:274-285).  ACCUMULATES into out. */
int swiftly_hip_add_to_subgrid(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_row_stride,
                               int64_t in_col_stride, void* out, int64_t out_row_stride, int64_t out_col_stride,
                               int64_t facet_off, void* stream);

/* Swiftly.finish_subgrid(in[rows, xM], out[rows, subgrid_size], subgrid_off)
 * (core.py:798-811; numpy form core.py:316-323), one axis per call.
 * `mask` (optional, device, real of matching precision, length subgrid_size)
 * folds the subgrid mask multiply of api_helper.py:107-112 into the store.
 * Overwrites out. */
int swiftly_hip_finish_subgrid(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_row_stride,
                               int64_t in_col_stride, void* out, int64_t out_row_stride, int64_t out_col_stride,
                               int64_t subgrid_off, int64_t subgrid_size, const void* mask, void* stream);

/* -- subgrid -> facet ------------------------------------------------------ */

/* Swiftly.prepare_subgrid_inplace (core.py:837, 852; numpy form
 * core.py:357-366), one axis per call, out of place with the pad_mid of
 * core.py:836/849 folded in: in[rows, subgrid_size] -> out[rows, xM].
 * Overwrites out. */
int swiftly_hip_prepare_subgrid(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t subgrid_size,
                                int64_t in_row_stride, int64_t in_col_stride, void* out, int64_t out_row_stride,
                                int64_t out_col_stride, int64_t subgrid_off, void* stream);

/* Swiftly.extract_from_subgrid(in[rows, xM], out[rows, m], facet_off)
 * (core.py:873; numpy form core.py:390-405).  Overwrites out. */
int swiftly_hip_extract_from_subgrid(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_row_stride,
                                     int64_t in_col_stride, void* out, int64_t out_row_stride,
                                     int64_t out_col_stride, int64_t facet_off, void* stream);

/* Swiftly.add_to_facet(in[rows, m], out[rows, yN], subgrid_off)
 * (core.py:897; numpy form core.py:430-446).  ACCUMULATES.  Exact adds. */
int swiftly_hip_add_to_facet(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_row_stride,
                             int64_t in_col_stride, void* out, int64_t out_row_stride, int64_t out_col_stride,
                             int64_t subgrid_off, void* stream);

/* Swiftly.finish_facet(in[rows, yN], out[rows, facet_size], facet_off)
 * (core.py:923; numpy form core.py:475-483).  `mask` (optional) folds the
 * facet mask multiply of api_helper.py:175-176 / 195-196.  Overwrites out. */
int swiftly_hip_finish_facet(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_row_stride,
                             int64_t in_col_stride, void* out, int64_t out_row_stride, int64_t out_col_stride,
                             int64_t facet_off, int64_t facet_size, const void* mask, void* stream);


/* -- fused / batched forms used by the streaming classes ---------------------
 *
 * *_batch: `nbatch` independent problems of identical shape; item b reads
 * in + b*in_batch_stride and writes out + b*out_batch_stride (strides in
 * complex elements; in_batch_stride = 0 shares one input).  `offs` is a HOST
 * array of nbatch per-item offsets, or NULL to use the scalar offset for all
 * items.  `mask` (finish_*): item b uses mask + b*mask_batch_stride (real
 * elements; 0 = one shared mask).  One launch covers up to 64 items. */

/* api_helper.extract_column (api_helper.py:200-210) as one kernel:
 * prepare_facet(extract_from_facet(BF_F, subgrid_off0, axis=0), facet_off1,
 * axis=1).  in = BF_F[yN, facet_size] (row stride in_row_stride), out =
 * NMBF_BF[m, yN].  The row gather is folded into the load. */
int swiftly_hip_extract_column(swiftly_hip_t* h, int dtype, const void* in, int64_t facet_size,
                               int64_t in_row_stride, int64_t in_col_stride, void* out, int64_t out_row_stride,
                               int64_t out_col_stride, int64_t subgrid_off0, int64_t facet_off1, void* stream);

/* Row-compacted BF_F (sparse subgrid sets): `rowmap` is a DEVICE int32 array of length yN; entry k is the
 * physical row of BF_F that holds logical row k, or negative when no requested subgrid column reads row k.
 * prepare_facet_rows = prepare_facet whose output index along the transform axis goes through the map
 * (unmapped rows are never written; rowmap may be NULL); extract_column_rows = extract_column reading such a
 * compacted BF_F.  fold_other_axis_window != 0 makes prepare_facet_rows also multiply row r of the call (= index r
 * along the OTHER axis, facet size `rows`) by 1/pswf -- the window the axis-1 prepare_facet of extract_column
 * applies; windows commute with transforms along the orthogonal axis, so extract_column_rows(prewindowed != 0)
 * then skips its window loads. */
int swiftly_hip_prepare_facet_rows(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t facet_size,
                                   int64_t in_row_stride, int64_t in_col_stride, void* out, int64_t out_row_stride,
                                   int64_t out_col_stride, int64_t facet_off, const int32_t* rowmap,
                                   int fold_other_axis_window, void* stream);
int swiftly_hip_extract_column_rows(swiftly_hip_t* h, int dtype, const void* in, int64_t facet_size,
                                    int64_t in_row_stride, int64_t in_col_stride, void* out, int64_t out_row_stride,
                                    int64_t out_col_stride, int64_t subgrid_off0, int64_t facet_off1,
                                    const int32_t* rowmap, int prewindowed, void* stream);

int swiftly_hip_extract_from_facet_batch(swiftly_hip_t* h, int dtype, const void* in, int64_t rows,
                                         int64_t in_row_stride, int64_t in_col_stride, void* out,
                                         int64_t out_row_stride, int64_t out_col_stride, int64_t subgrid_off,
                                         int64_t nbatch, int64_t in_batch_stride, int64_t out_batch_stride,
                                         const int64_t* subgrid_offs, void* stream);
int swiftly_hip_add_to_subgrid_batch(swiftly_hip_t* h, int dtype, const void* in, int64_t rows,
                                     int64_t in_row_stride, int64_t in_col_stride, void* out, int64_t out_row_stride,
                                     int64_t out_col_stride, int64_t facet_off, int64_t nbatch,
                                     int64_t in_batch_stride, int64_t out_batch_stride, const int64_t* facet_offs,
                                     void* stream);
int swiftly_hip_finish_subgrid_batch(swiftly_hip_t* h, int dtype, const void* in, int64_t rows,
                                     int64_t in_row_stride, int64_t in_col_stride, void* out, int64_t out_row_stride,
                                     int64_t out_col_stride, int64_t subgrid_off, int64_t subgrid_size,
                                     const void* mask, int64_t nbatch, int64_t in_batch_stride,
                                     int64_t out_batch_stride, const int64_t* subgrid_offs,
                                     int64_t mask_batch_stride, void* stream);
int swiftly_hip_prepare_subgrid_batch(swiftly_hip_t* h, int dtype, const void* in, int64_t rows,
                                      int64_t subgrid_size, int64_t in_row_stride, int64_t in_col_stride, void* out,
                                      int64_t out_row_stride, int64_t out_col_stride, int64_t subgrid_off,
                                      int64_t nbatch, int64_t in_batch_stride, int64_t out_batch_stride,
                                      const int64_t* subgrid_offs, void* stream);
int swiftly_hip_extract_from_subgrid_batch(swiftly_hip_t* h, int dtype, const void* in, int64_t rows,
                                           int64_t in_row_stride, int64_t in_col_stride, void* out,
                                           int64_t out_row_stride, int64_t out_col_stride, int64_t facet_off,
                                           int64_t nbatch, int64_t in_batch_stride, int64_t out_batch_stride,
                                           const int64_t* facet_offs, void* stream);
int swiftly_hip_add_to_facet_batch(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_row_stride,
                                   int64_t in_col_stride, void* out, int64_t out_row_stride, int64_t out_col_stride,
                                   int64_t subgrid_off, int64_t nbatch, int64_t in_batch_stride,
                                   int64_t out_batch_stride, const int64_t* subgrid_offs, void* stream);
int swiftly_hip_finish_facet_batch(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_row_stride,
                                   int64_t in_col_stride, void* out, int64_t out_row_stride, int64_t out_col_stride,
                                   int64_t facet_off, int64_t facet_size, const void* mask, int64_t nbatch,
                                   int64_t in_batch_stride, int64_t out_batch_stride, const int64_t* facet_offs,
                                   int64_t mask_batch_stride, void* stream);

/* Fused second half of api_helper.sum_and_finish_subgrid (api_helper.py:96-112) along the contiguous axis,
 * complex64: for every row of every subgrid b of a wave
 *     out[b][row, :] = mask_b * finish_subgrid_axis1( sum_g add_to_subgrid_axis1(in[g][b][row, :], group_facet_offs[g]) )
 * in[g][b] = [xM, m] (row stride in_row_stride), out[b] = [xM, subgrid_size].  Host arrays: group_facet_offs
 * (ngroups <= 8 facet off1 values), subgrid_offs (nbatch subgrid off1 values).  The [xM, xM] accumulator never
 * reaches HBM.  SWIFTLY_ERR_UNSUPPORTED for complex128 or (m, xM) pairs that are not instantiated. */
int swiftly_hip_sum_finish_rows(swiftly_hip_t* h, int dtype, const void* in, int64_t ngroups, int64_t in_group_stride,
                                int64_t in_batch_stride, int64_t in_row_stride, const int64_t* group_facet_offs,
                                void* out, int64_t out_batch_stride, int64_t out_row_stride,
                                const int64_t* subgrid_offs, int64_t subgrid_size, const void* mask,
                                int64_t mask_batch_stride, int64_t nbatch, void* stream);

/* K3 + K4a fused (complex64): for facet index f < nfacets and subgrid b < nsub
 *     out[f][b] (+)= add_to_subgrid_axis0( extract_from_facet_axis1(in[f], subgrid_off1s[b]), facet_off0 )
 * with in[f] = NMBF_BF [m, yN] at in + f*in_facet_stride (row stride in_row_stride) -- the output of
 * extract_column -- and out[f][b] = [xM, m] at out + (f*nsub + b)*out_batch_stride (element (r, c) at
 * r*out_col_stride + c).  The window gather of core.py:243-253 is folded into the load, so the [m, m]
 * contribution never goes through HBM (single-GPU path; the multi-GPU path materialises it for the
 * all-to-all).  All nfacets facets must share facet_off0.  ACCUMULATES. */
int swiftly_hip_add_to_subgrid_from_columns(swiftly_hip_t* h, int dtype, const void* in, int64_t in_row_stride,
                                            int64_t in_facet_stride, int64_t nfacets, void* out,
                                            int64_t out_col_stride, int64_t out_batch_stride, int64_t facet_off0,
                                            int64_t nsub, const int64_t* subgrid_off1s, void* stream);

/* -- device memory helpers for callers that do not bring their own allocator
 *    (the Python mirror uses torch for device memory and never calls these) -- */
int swiftly_hip_malloc(void** ptr, size_t bytes);
int swiftly_hip_free(void* ptr);
int swiftly_hip_memset_async(void* ptr, int value, size_t bytes, void* stream);
int swiftly_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int swiftly_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int swiftly_hip_stream_synchronize(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SWIFTLY_HIP_H */
