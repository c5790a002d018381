// C ABI of libswiftly_hip.so, part 3: device memory / stream helpers for callers without their own allocator,
// CU-partitioned streams and diagnostics (include/swiftly_hip.h, last section).
#include "swiftly_abi_internal.h"
#include "build/build_id.h"

// where does this workgroup run?  xcc_id << 16 | se_id << 8 | cu_id  (HW_REG_XCC_ID, HW_REG_HW_ID of gfx9);
// the workgroup lingers for a few microseconds so that a census grid spreads over all CUs its stream may use
__global__ void cu_census_kernel(int* __restrict__ out) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 20000) {
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(((xcc & 0xf) << 16) | (((hw >> 13) & 0x7) << 8) | ((hw >> 8) & 0xf));
}


thread_local int g_chain_chunk_streams = 0;

extern "C" {

const char* swiftly_hip_build_id(void) { return SWF_SRC_HASH; }

void swiftly_hip_chain_chunk_streams(int chain) { g_chain_chunk_streams = chain ? 1 : 0; }

int swiftly_hip_set_column_precision(swiftly_hip_t* h, int bits) {
    if (!h) return fail(SWIFTLY_ERR_PARAM, "null argument");
    if (bits != 32 && bits != 64) return fail(SWIFTLY_ERR_PARAM, "column precision must be 32 or 64");
    h->col_f64 = bits == 64;
    return 0;
}
int swiftly_hip_get_column_precision(const swiftly_hip_t* h) { return h ? (h->col_f64 ? 64 : 32) : -1; }

int swiftly_hip_debug_row_band_occupancy(void) { return swf::row_pass_band_occupancy(); }

int swiftly_hip_debug_occupancy(int lds_bytes) { return swf::row_pass_half_occupancy(lds_bytes); }

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

int swiftly_hip_stream_create_cu_mask(void** stream, const uint32_t* cu_mask, int nwords) {
    if (!stream || !cu_mask || nwords <= 0) return fail(SWIFTLY_ERR_PARAM, "null argument");
    hipStream_t st = nullptr;
    HIP_TRY(hipExtStreamCreateWithCUMask(&st, (uint32_t)nwords, cu_mask));
    *stream = (void*)st;
    return 0;
}
int swiftly_hip_stream_destroy(void* stream) {
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return 0;
}
int swiftly_hip_cu_census(int32_t* out, int nblocks, void* stream) {
    if (!out || nblocks <= 0) return fail(SWIFTLY_ERR_PARAM, "null argument");
    hipLaunchKernelGGL(cu_census_kernel, dim3((unsigned)nblocks), dim3(64), 0, (hipStream_t)stream, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
