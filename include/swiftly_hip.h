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
 *  - transform lengths (yN, xM, m = xM*yN/N): powers of two from 4 to 32768 run on native kernels in both
 *    precisions (complex128 up to 8192 along a unit-stride axis); 65536 is supported in complex64 for prepare_* /
 *    finish_* along a unit-stride axis and along the strided axis of contiguous rows.  Lengths Q * 2^k with
 *    Q in {3, 5, 7, 9} -- every other length of the reference's parameter catalogue, yN up to 57344 -- run natively
 *    too: one radix-Q pass in front of the power-of-two kernels (2^k <= 32768 in complex64, <= 8192 in complex128;
 *    rows * length < 2^32 per call).  Any other length goes through a Bluestein fallback on the power-of-two kernels
 *    (correct, ~10x the traffic) as long as 2^ceil(log2(2n-1)) <= 65536 (complex64) / 8192 (complex128); beyond
 *    that the call returns SWIFTLY_ERR_UNSUPPORTED (handle creation still succeeds).
 *  - a handle is immutable after creation and may be used concurrently from
 *    several host threads / streams (the reference scatters one core object
 *    to all Dask worker threads, api.py:145-147).  Handles may be created
 *    concurrently and on several devices of one process; every entry point
 *    makes the handle's device current for the duration of the call and
 *    restores the caller's current device afterwards (`stream` must belong to
 *    the handle's device).
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
/* Identity of the build: the first 16 hex digits of the SHA-256 of the kernel sources (the headers and .hip files
 * of csrc, this header, the Makefile) the library was compiled from.  The per-kernel counter summaries under profiles/
 * record it and bench.py compares it with the running build. */
const char* swiftly_hip_build_id(void);

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
/* Arithmetic of the column passes of the band pipelines on complex64 data (K2 = prepare_facet along the strided axis of
 * a wave, K3 = the m-point transform of add_to_subgrid behind it, and their backward mirrors): 32 (default) = float32
 * throughout; 64 = loads and stores in complex64, windows / butterflies / exchanges / four-step twiddles in float64.
 * These are the transforms whose rounding errors are amplified by BOTH facet windows (1/pswf, up to 90 per axis) before
 * anything cancels them; in float64 the end-to-end complex64 error against the numpy reference drops from 1.0e-5 to
 * 2.8e-6 forward and from 2.1e-5 to 5.1e-6 backward on the N = 65536 workload (storage floor 3.3e-6 on the probe
 * configuration of tests/accuracy_model.py), for 1.5x the pass time.  Initial value from the
 * environment variable SWIFTLY_COL_F64 (0 | 1).  (No reference counterpart: numpy computes in complex128 throughout,
 * core.py:212-222.) */
int swiftly_hip_set_column_precision(swiftly_hip_t* h, int bits);
int swiftly_hip_get_column_precision(const swiftly_hip_t* h);
/* Chained four-step launches (calling thread only; default 0).  A strided-axis transform whose intermediate exceeds
 * 256 MB runs in chunks on two internal streams of the handle; each call FORKS them behind everything queued on the
 * caller's stream and JOINS them back into it.  Between two such calls that follow each other on one stream the join /
 * fork pair is a full pipeline drain plus two cross-stream event hops (measured on the 64k workload: 40 us with nothing
 * running on the GPU, once per wave).  With chain = 1 the calling thread promises that the INPUTS of its next calls were
 * already complete when an earlier (forking) call on the same handle was queued, and that nobody else still uses the
 * output buffer: the fork is skipped, the chunk streams run on from the previous call's chunks, the join is unchanged
 * (the caller's stream still continues behind the whole transform).  SwiftlyForward's planned-wave prefetch (K2 of
 * consecutive waves read the same band buffers) sets it around the second and later calls of a pass.  (No reference
 * counterpart: scheduling of this implementation.) */
void swiftly_hip_chain_chunk_streams(int chain);

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

/* Swiftly.add_to_subgrid_2d(in[m, m], out[xM, xM], facet_off0, facet_off1) (core.py:752-778): both axes in one call;
 * ACCUMULATES into out.  (Axis 0 into a stream-ordered [xM, m] intermediate, then axis 1 into out.) */
int swiftly_hip_add_to_subgrid_2d(swiftly_hip_t* h, int dtype, const void* in, int64_t in_row_stride,
                                  int64_t in_col_stride, void* out, int64_t out_row_stride, int64_t out_col_stride,
                                  int64_t facet_off0, int64_t facet_off1, void* stream);

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

/* Swiftly.prepare_subgrid_inplace(data[rows, xM], subgrid_off) (core.py:837-840): `data` already holds the subgrid
 * zero-padded to xM (pad_mid: centred); in place it becomes fft(roll(data, subgrid_off)) along the axis whose element
 * stride is col_stride.  ..._2d(data[xM, xM], off0, off1) (core.py:851-853) does both axes. */
int swiftly_hip_prepare_subgrid_inplace(swiftly_hip_t* h, int dtype, void* data, int64_t rows, int64_t row_stride,
                                        int64_t col_stride, int64_t subgrid_off, void* stream);
int swiftly_hip_prepare_subgrid_inplace_2d(swiftly_hip_t* h, int dtype, void* data, int64_t row_stride, int64_t col_stride,
                                           int64_t subgrid_off0, int64_t subgrid_off1, void* stream);

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
 * complex elements; in_batch_stride = 0 shares one input; the ACCUMULATING
 * entry points add_to_subgrid_batch / add_to_facet_batch run the items
 * concurrently with plain read-modify-write, so items must not share output
 * elements: out_batch_stride = 0 with nbatch > 1 is SWIFTLY_ERR_PARAM).  `offs` is a HOST
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

/* -- contiguous-axis-first forward pipeline ----------------------------------------------------------------
 *
 * The 2-D transforms of the reference are separable, so the order in which api_helper.extract_column
 * (api_helper.py:200-210) visits the axes is free (it does axis 0 on the whole facet, then axis 1 per subgrid
 * column).  On a row-major facet the cheap order is the other one: the full-facet pass runs along the CONTIGUOUS
 * axis in ONE kernel (no four-step scratch), keeps only the output columns some planned subgrid reads, and the
 * strided-axis transform moves to the per-wave stage where it acts on m columns only.  complex64.
 *
 *   P_f   = prepare_facet_band(facet_f)                      once per facet       [yB, band]      K1
 *   Q_f   = prepare_facet_columns(P_f, wave off1)            per subgrid wave     [rows kept, m]  K2
 *   G_f,b = transform_contributions(Q_f, subgrid b)          per (facet, subgrid) [m, m]          K3+K4a
 *   T_b   = sum_finish_facets(G_.,b)                         per subgrid          [xM, xA]        K4b+K5a
 *   S_b   = finish_subgrid_batch(T_b, axis 0)                per subgrid          [xA, xA]        K5b
 */

/* Number of physical columns of a band buffer holding `band_len` logical columns (parity-split layout: the
 * logical column with cyclic distance d from band_start lives at (d & 1) * H + (d >> 1), H = swiftly_hip_band_columns / 2
 * = (band_len + 1) / 2 rounded up to a multiple of 16 columns, so that both parity runs start on a 128-byte line). */
int64_t swiftly_hip_band_columns(int64_t band_len);
/* The same for the band layout a HANDLE uses: parity-split as above for yN = 16384, 32768, 65536 (where the
 * two-workgroup long-row kernel produces the band); PLAIN for shorter padded facets (logical column d at physical
 * column d, band_columns = band_len; K1 is then the generic contiguous-axis transform and the band must be the
 * whole padded axis (0, yN)). */
int64_t swiftly_hip_band_columns_for(const swiftly_hip_t* h, int64_t band_len);

/* K1: Swiftly.prepare_facet(in[rows, facet_size], ., facet_off) (core.py:686; numpy form core.py:212-222) along
 * the contiguous axis for every row of a facet, keeping only the centred output indices in the cyclic range
 * [band_start, band_start + band_len) of [0, yN) (band_len = yN keeps everything), parity-split (see above);
 * out[rows, band_columns(band_len)], row stride out_row_stride.  fold_other_axis_window != 0 also multiplies row r
 * by 1/pswf of the OTHER axis (facet size `rows`) -- the window prepare_facet applies along that axis later;
 * windows commute with transforms along the orthogonal axis.  yN = 16384, 32768 or 65536 (SWIFTLY_ERR_UNSUPPORTED else). */
int swiftly_hip_prepare_facet_band(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t facet_size,
                                   int64_t in_row_stride, void* out, int64_t out_row_stride, int64_t facet_off,
                                   int64_t band_start, int64_t band_len, int fold_other_axis_window, void* stream);
/* The same for a BLOCK OF ROWS [other_axis_row0, other_axis_row0 + rows) of a facet with other_axis_size rows (`in` points
 * at the first row of the block): the folded window of the other axis is that facet's 1/pswf at those rows, so that the
 * band rows of several blocks -- computed by different GPUs -- assemble to the band buffer of the whole facet
 * (distributed.py: cooperative facets when the facet count does not divide by the number of ranks).
 * other_axis_size = 0: no window of the other axis.  Split band layout only (yN_size 16384 .. 65536). */
int swiftly_hip_prepare_facet_band_rows(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t facet_size,
                                        int64_t in_row_stride, void* out, int64_t out_row_stride, int64_t facet_off,
                                        int64_t band_start, int64_t band_len, int64_t other_axis_size,
                                        int64_t other_axis_row0, void* stream);

/* K2: for every facet f < nfacets: extract_from_facet(P_f, subgrid_off1, axis=1) (core.py:715) folded into the load
 * of prepare_facet(., facet_off0s[f], axis=0) WITHOUT its window (pre-applied by prepare_facet_band).
 * in + f*in_facet_stride = band buffer P_f[rows, band], out + f*out_facet_stride = Q_f[., m] with row stride
 * out_row_stride; out_rowmap (device int32[yN], optional): physical output row of logical row k, negative = not
 * stored (the rows no subgrid of the plan reads).  facet_off0s is a HOST array. */
int swiftly_hip_prepare_facet_columns(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_row_stride,
                                      int64_t in_facet_stride, int64_t nfacets, const int64_t* facet_off0s,
                                      int64_t band_start, int64_t band_len, int64_t subgrid_off1, void* out,
                                      int64_t out_row_stride, int64_t out_facet_stride, const int32_t* out_rowmap,
                                      void* stream);

/* K2 for several waves at once (facet-major schedule: all waves of one facet right after its K1, on a second
 * stream, so that this bandwidth-bound work overlaps the issue-bound K1 of the next facet): item (f, w) is written at
 * out + f*out_facet_stride + w*out_wave_stride through the row map rowmaps + w*rowmap_stride (device int32[yN] each;
 * NULL = all rows).  workspace (optional, device, workspace_bytes >= 8*yN*m per (facet, wave) of a launch group):
 * the four-step scratch; with NULL it comes from the stream-ordered pool, whose reuse across streams is only
 * opportunistic.  wave_off1s / facet_off0s are HOST arrays. */
int swiftly_hip_prepare_facet_columns_waves(swiftly_hip_t* h, int dtype, const void* in, int64_t rows,
                                            int64_t in_row_stride, int64_t in_facet_stride, int64_t nfacets,
                                            const int64_t* facet_off0s, int64_t band_start, int64_t band_len,
                                            int64_t nwaves, const int64_t* wave_off1s, void* out, int64_t out_row_stride,
                                            int64_t out_facet_stride, int64_t out_wave_stride, const int32_t* rowmaps,
                                            int64_t rowmap_stride, void* workspace, int64_t workspace_bytes,
                                            void* stream);

/* K3 + K4a: out[f][b][k, j] = Fn[k] * cfft_m(C_{f,b}[:, j])[(k + s'0_f) mod m]  --  add_to_subgrid(axis 0)
 * (core.py:744, numpy form core.py:274-285) of the [m, m] contribution C_{f,b} WITHOUT its placement into the
 * padded subgrid (row k belongs to padded row (k + xM/2 - m/2 + s'0_f) mod xM; sum_finish_facets resolves that).
 * The contribution is gathered on load (extract_from_facet, core.py:243-253, never materialised):
 *   layout 0: in + f*in_facet_stride = NMBF_BF[m, yN or band] (output of extract_column[_rows]); window of
 *             subgrid_offs[b] (an off1) along the contiguous axis; band_len > 0: parity-split band columns.
 *   layout 1: in + f*in_facet_stride = Q_f[rows kept, m] (output of prepare_facet_columns); window of
 *             subgrid_offs[b] (an off0) along the strided axis through in_rowmap (device int32[yN] or NULL).
 *   layout 2: in + f*in_facet_stride + b*in_sub_stride = C_{f,b}[m, m] materialised (multi-GPU path).
 * out[f][b] at out + f*out_facet_stride + b*out_sub_stride, [m, m] row-major.  Overwrites.  Host offset arrays. */
int swiftly_hip_transform_contributions(swiftly_hip_t* h, int dtype, const void* in, int layout, int64_t in_row_stride,
                                        int64_t in_facet_stride, int64_t in_sub_stride, const int32_t* in_rowmap,
                                        int64_t band_start, int64_t band_len, int64_t nfacets,
                                        const int64_t* facet_off0s, int64_t nsub, const int64_t* subgrid_offs,
                                        void* out, int64_t out_facet_stride, int64_t out_sub_stride, void* stream);

/* K4b + K5a: for every padded row r < xM of every subgrid b:
 *   out[b][r, :] = mask_b * finish_subgrid_axis1( sum_f add_to_subgrid_axis1( G[f][b][k_f(r), :], facet_off1s[f] ) )
 * over the facets whose axis-0 band covers r (k_f(r) = (r - (xM/2 - m/2 + s'0_f)) mod xM < m): the facet sums of
 * api_helper.sum_and_finish_subgrid (api_helper.py:81-99) re-associated so that no accumulator reaches HBM, plus
 * finish_subgrid along axis 1 and the mask (api_helper.py:101-111).  in = G from transform_contributions,
 * out[b] = [xM, subgrid_size].  Up to 64 facets. */
int swiftly_hip_sum_finish_facets(swiftly_hip_t* h, int dtype, const void* in, int64_t nfacets, int64_t in_facet_stride,
                                  int64_t in_sub_stride, int64_t in_row_stride, const int64_t* facet_off0s,
                                  const int64_t* facet_off1s, void* out, int64_t out_sub_stride, int64_t out_row_stride,
                                  const int64_t* subgrid_off1s, int64_t subgrid_size, const void* mask,
                                  int64_t mask_batch_stride, int64_t nsub, void* stream);

/* One forward wave as TWO native calls (per-wave host work = two ABI calls; what the streaming classes use):
 *
 * wave_facet_side: K2 (prepare_facet_columns into the workspace q_work[nfacets][n_rows][m], facet stride
 *   q_facet_stride elements; skipped when compute_q == 0 and q_work still holds the wave's result -- e.g. a slice
 *   of what prepare_facet_columns_waves precomputed) followed by K3 + K4a (transform_contributions, layout 1)
 *   for the subgrids sub_off0s[nsub] of the wave.  Block (f, b) is written at
 *       g_out + g_offsets[b] + f * g_facet_strides[b]            when g_offsets != NULL
 *       g_out + f * g_facet_stride + b * g_sub_stride             otherwise
 *   (elements) -- the first form fills a rank-ordered all-to-all send buffer [dest][facet][subgrid of dest] in place.
 * wave_subgrid_side: K4b + K5a (sum_finish_facets into tmp_work[nsub][xM][subgrid_size]) and K5b
 *   (finish_subgrid along axis 0 with mask0) -> out[nsub][subgrid_size][subgrid_size] for the blocks g[f][b] of ALL
 *   facets (facet order = order of facet_off0s / facet_off1s, e.g. arrival order of the exchange).
 * Host arrays: facet / subgrid offsets, g_offsets, g_facet_strides.  masks: device, real, [nsub][subgrid_size] with
 * batch strides (0 = shared), or NULL.  scratch / scratch_bytes: optional device scratch for the four-step
 * intermediates (facet side: nfacets*yN*m*8 bytes, subgrid side: min(nsub, 64)*xM*subgrid_size*8 bytes); NULL or too
 * small = stream-ordered allocation, which costs ~2 ms of host time per call when the size changes between calls. */
int swiftly_hip_wave_facet_side(swiftly_hip_t* h, int dtype, const void* bands, int64_t rows, int64_t band_row_stride,
                                int64_t band_facet_stride, int64_t nfacets, const int64_t* facet_off0s,
                                int64_t band_start, int64_t band_len, int64_t wave_off1, const int32_t* rowmap,
                                int64_t n_rows, void* q_work, int64_t q_facet_stride, int compute_q, int64_t nsub,
                                const int64_t* sub_off0s, void* g_out, int64_t g_facet_stride, int64_t g_sub_stride,
                                const int64_t* g_offsets, const int64_t* g_facet_strides, void* scratch,
                                int64_t scratch_bytes, void* stream);
int swiftly_hip_wave_subgrid_side(swiftly_hip_t* h, int dtype, const void* g, int64_t nfacets, int64_t g_facet_stride,
                                  int64_t g_sub_stride, const int64_t* facet_off0s, const int64_t* facet_off1s,
                                  int64_t nsub, const int64_t* sub_off0s, const int64_t* sub_off1s, int64_t subgrid_size,
                                  const void* mask0, int64_t mask0_bs, const void* mask1, int64_t mask1_bs,
                                  void* tmp_work, void* out, void* scratch, int64_t scratch_bytes, void* stream);

/* AXIS-1-FIRST forward pipeline (r6; opt-in: the accuracy mode that replaces float64 column passes).  The transforms of
 * the two axes commute (api_helper.py:81-99, 200-210), so the contiguous-axis half of add_to_subgrid (core.py:255-285:
 * m-point transform x Fn, the step that takes the axis-1 facet window out again) can run on the K1 output BEFORE the
 * strided-axis transforms K2 / K3, which then work on data that carries ONE facet window instead of two: their float32
 * rounding reaches the subgrid an order of magnitude weaker.
 *
 * finish_axis1_rows: for every facet f and row r of its K1 band buffer (bands + f*band_facet_stride, [rows, band columns],
 *   parity-split) and the wave `wave_off1` (s = wave_off1 * yN / N):
 *       x[(i + s) mod m] = P_f[r, (yN/2 - m/2 + i + s) mod yN]            extract_from_facet(axis 1), core.py:243-253
 *       Z[k] = Fn[k] * cfft_m(x)[(k + s'1_f) mod m]                        add_to_subgrid(axis 1) without its placement
 *   out + f*out_facet_stride = [rows, m] in the parity-split layout of a band that is exactly the wave's window
 *   (band_start' = (yN/2 - m/2 + s) mod yN, band_len' = m; logical column i holds Z[(i + s) mod m]): pass it to
 *   prepare_facet_columns / wave_facet_side as `bands` with that band, and the contributions come out with Z along their
 *   contiguous axis.  facet_off1s: host array.  yN_size 16384 .. 65536, m 128 .. 1024, complex64.
 * wave_subgrid_side_placed: wave_subgrid_side for blocks produced that way -- sum_finish_facets places and sums the rows
 *   without its m-point transforms (xM <= 2048). */
int swiftly_hip_finish_axis1_rows(swiftly_hip_t* h, int dtype, const void* bands, int64_t rows, int64_t band_row_stride,
                                  int64_t band_facet_stride, int64_t nfacets, const int64_t* facet_off1s,
                                  int64_t band_start, int64_t band_len, int64_t wave_off1, void* out,
                                  int64_t out_row_stride, int64_t out_facet_stride, void* stream);
int swiftly_hip_wave_subgrid_side_placed(swiftly_hip_t* h, int dtype, const void* g, int64_t nfacets, int64_t g_facet_stride,
                                         int64_t g_sub_stride, const int64_t* facet_off0s, const int64_t* facet_off1s,
                                         int64_t nsub, const int64_t* sub_off0s, const int64_t* sub_off1s,
                                         int64_t subgrid_size, const void* mask0, int64_t mask0_bs, const void* mask1,
                                         int64_t mask1_bs, void* tmp_work, void* out, void* scratch, int64_t scratch_bytes,
                                         void* stream);

/* AXIS-1-FIRST pipeline with the contiguous-axis finish FUSED INTO THE FORWARD K1 (r6): prepare_facet_band_rows as one
 * persistent workgroup per CU that owns whole rows -- both output parities -- stages the band of a row in LDS (it never
 * reaches memory) and stores for each of the `nwindows` contribution windows of the plan (window w = the m columns from
 * logical column band_start + window_starts[w], all inside the band) exactly what finish_axis1_rows produces for wave w:
 *     out[row*out_row_stride + w*out_window_stride ..] = parity-split window band of  Fn[k] cfft_m(window w)[(k + s'1) mod m]
 * (wave-major [nwindows][rows][m]: out_row_stride = m, out_window_stride = rows*m -- what the pipeline uses: K2 of wave w then
 * reads one contiguous block; or side by side in a row: out_row_stride >= nwindows*m, out_window_stride = m)
 * so the rows of window w go to prepare_facet_columns / wave_facet_side with the band (window start, m) and the
 * blocks to wave_subgrid_side_placed.  No band buffer, no row pass per wave.  window_starts: DEVICE int32 table.
 * yN_size 32768, m = 512, even facet size / offset / row stride, a band of at most 12800 physical columns (the LDS stage);
 * SWIFTLY_ERR_UNSUPPORTED otherwise: use prepare_facet_band + finish_axis1_rows.  complex64. */
int swiftly_hip_prepare_facet_window_rows(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t facet_size,
                                          int64_t in_row_stride, void* out, int64_t out_row_stride, int64_t facet_off,
                                          int64_t band_start, int64_t band_len, int64_t other_axis_size,
                                          int64_t other_axis_row0, const int32_t* window_starts, int64_t nwindows,
                                          int64_t out_window_stride, void* stream);
/* Backward subgrid side for all facets at once (mirror of sum_finish_facets): in[b] = [xM, subgrid_size] =
 * prepare_subgrid of subgrid b along axis 0 ONLY (core.py:328-368); out[f][b] = [m, m] contiguous = the contribution
 * of subgrid b to facet f, i.e. api_helper.prepare_and_split_subgrid (api_helper.py:115-139): prepare_subgrid along
 * axis 1 and both extract_from_subgrid (core.py:396-439).  The prepared [xM, xM] subgrid and the per-off0
 * intermediates never reach HBM.  (m, xM) pairs of sum_finish_facets; up to 64 facets. */
int swiftly_hip_split_prepare_facets(swiftly_hip_t* h, int dtype, const void* in, int64_t in_sub_stride,
                                     int64_t in_row_stride, int64_t subgrid_size, int64_t nsub,
                                     const int64_t* subgrid_off1s, int64_t nfacets, const int64_t* facet_off0s,
                                     const int64_t* facet_off1s, void* out, int64_t out_facet_stride,
                                     int64_t out_sub_stride, void* stream);

/* The subgrid side of one backward wave in ONE native call without stream-ordered allocations: prepare_subgrid along
 * axis 0 of subgrids[nsub][subgrid_size][subgrid_size] (contiguous) followed by split_prepare_facets.  work: device
 * scratch of work_elems >= 2 * nsub * xM * subgrid_size complex64 elements. */
int swiftly_hip_wave_split_subgrids(swiftly_hip_t* h, int dtype, const void* subgrids, int64_t subgrid_size, int64_t nsub,
                                    const int64_t* subgrid_off0s, const int64_t* subgrid_off1s, int64_t nfacets,
                                    const int64_t* facet_off0s, const int64_t* facet_off1s, void* work,
                                    int64_t work_elems, void* out, int64_t out_facet_stride, int64_t out_sub_stride,
                                    void* stream);

/* Backward pass with the contiguous-axis transform LAST (mirror of prepare_facet_band / prepare_facet_columns;
 * waves = subgrids sharing off1).  Replaces, for all facets of a wave, api_helper.accumulate_column +
 * accumulate_facet (api_helper.py:142-179) with the two axes swapped (they commute):
 *
 * accumulate_facet_columns: for every facet f
 *     bands[f][r, d] += mask0_f[r] * finish_facet_axis0( sum_b add_to_facet_axis0( C[f][b], sub_off0[b] ), facet_off0s[f] )[r, j]
 *   where column j < m of the wave lands at band column d = (col(j, subgrid_off1) - band_start) mod yN of the padded
 *   facet (add_to_facet along axis 1, core.py:441-478).  The sum over the wave's subgrids is taken while LOADING:
 *   row_sources (device int32 [2][yN]) names for every padded row up to two source rows (negative = none), each
 *   encoded  chunk << 20 | row : read at parts + chunk_offsets[chunk] + f * chunk_facet_strides[chunk] + row *
 *   part_row_stride (elements; row = b*m + k for contiguous [m, m] blocks).  Chunks (<= 16) are the pieces of a
 *   multi-GPU receive buffer; a single-process caller passes one chunk.  masks: device float [nfacets][facet_size]
 *   or NULL.  bands[f] = [facet_size rows][band_len] plain column order, read-modify-written (zero it first).
 *   touched: optional device bytes [band_len], zero before the first call: band columns whose byte is 0 have not
 *   been written yet and are stored plainly (bands need no zero fill, no read of the old value); the call then marks
 *   its columns.  band_zero_untouched clears the columns no call has written (bands[rows][band_len], rows = all
 *   facets' rows when the facets are contiguous) -- call it before finish_facet_band.  NULL: always read-modify-write.
 *   workspace: optional device scratch (nfacets*yN*m*8 bytes used when large enough), else stream-ordered allocation.
 * finish_facet_band: finish_facet (core.py:481-510) along the contiguous axis of a band accumulator row:
 *     out[r, :] = mask * Fb * crop( FFT_yN( band row r placed at columns band_start + d, zero elsewhere ) ). */
int swiftly_hip_accumulate_facet_columns(swiftly_hip_t* h, int dtype, const void* parts, int64_t part_row_stride,
                                         int64_t nchunks, const int64_t* chunk_offsets,
                                         const int64_t* chunk_facet_strides, const int32_t* row_sources,
                                         int64_t nfacets, const int64_t* facet_off0s, int64_t facet_size,
                                         const float* masks, int64_t subgrid_off1, void* bands, int64_t band_row_stride,
                                         int64_t band_facet_stride, int64_t band_start, int64_t band_len,
                                         unsigned char* touched, void* workspace, int64_t workspace_bytes,
                                         void* stream);
int swiftly_hip_band_zero_untouched(swiftly_hip_t* h, int dtype, void* bands, int64_t rows, int64_t band_row_stride,
                                    int64_t band_len, const unsigned char* touched, void* stream);
int swiftly_hip_finish_facet_band(swiftly_hip_t* h, int dtype, const void* in, int64_t rows, int64_t in_row_stride,
                                  int64_t band_start, int64_t band_len, void* out, int64_t out_row_stride,
                                  int64_t facet_off, int64_t facet_size, const void* mask, void* stream);

/* -- device memory helpers for callers that do not bring their own allocator
 *    (the Python mirror uses torch for device memory and never calls these) -- */
int swiftly_hip_malloc(void** ptr, size_t bytes);
int swiftly_hip_free(void* ptr);
int swiftly_hip_memset_async(void* ptr, int value, size_t bytes, void* stream);
int swiftly_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int swiftly_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int swiftly_hip_stream_synchronize(void* stream);

/* -- CU-partitioned streams: two kernels of different character (the full-facet row transform: latency / issue bound;
 *    the per-wave column passes: bandwidth bound) can share the chip on disjoint sets of compute units.
 *    stream_create_cu_mask: a HIP stream whose kernels only run on the CUs whose bit is set in cu_mask (nwords 32-bit
 *    words, bit i of word i/32 = CU i in the runtime's enumeration; hipExtStreamCreateWithCUMask).  cu_census: launches
 *    `nblocks` one-wave workgroups on `stream` and writes per block  xcc_id << 16 | se_id << 8 | cu_id  to the device
 *    array out[nblocks] -- how a mask maps to XCDs is not documented, so callers measure it. */
int swiftly_hip_stream_create_cu_mask(void** stream, const uint32_t* cu_mask, int nwords);
int swiftly_hip_stream_destroy(void* stream);
int swiftly_hip_cu_census(int32_t* out, int nblocks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SWIFTLY_HIP_H */
