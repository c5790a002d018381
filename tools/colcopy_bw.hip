// Microbenchmark (r4): what rate can a column-tile pass reach on the MI355X?  Copies a [Z][ROWS][COLS] complex64 array
// (the shape of the K2 four-step scratch: 9 x 32768 x 512 = 1.2 GB) with the access patterns of the two passes of
// swiftly_colpass.h -- every lane issues all its loads, then all its stores -- varying only HOW a wave touches memory:
//   8 B per lane, 64 lanes per row (the kernels today: 512-B segments),
//   16 B per lane, 32 lanes per row (same 512-B segment, a wave covers two rows),
//   16 B per lane, 64 lanes per row (1-KB segments, 128-column tiles),
// for consecutive rows (pass B: rows k1*256 + 0..255) and for a comb of rows (pass A: rows y2 + 256*y1), with plain and
// non-temporal accesses, next to a flat float4 copy of the same bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/colcopy_bw.hip -o /tmp/colcopy_bw && /tmp/colcopy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float2 cx;
constexpr int Z = 9, ROWS = 32768, COLS = 512;

typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void copy_flat(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        if constexpr (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i);
        else b[i] = a[i];
    }
}

// VEC complex elements per lane per access; LPR lanes per row segment; a wave covers 64 / LPR rows at a time;
// WROWS rows per workgroup, NTHR threads; rows of the workgroup: base + rstride * (0 .. WROWS-1).
// COMB = false: base = o * WROWS, rstride = 1 (pass B);  COMB = true: base = o, rstride = ROWS / WROWS (pass A).
// Writes: WFRAC of 4 row groups are written (4 = all; 1 = every fourth row of the workgroup: pass B keeps 28 %).
template <int VEC, int LPR, int WROWS, int NTHR, bool COMB, bool NT, int WFRAC>
__global__ __launch_bounds__(NTHR) void tile_copy(const cx* __restrict__ src, cx* __restrict__ dst) {
    constexpr int RPW = 64 / LPR;              // rows per wave-access
    constexpr int NW = NTHR / 64;              // waves
    constexpr int P = WROWS / (NW * RPW);      // accesses per lane
    static_assert(P >= 1 && P * NW * RPW == WROWS, "geometry");
    constexpr int TILEC = LPR * VEC;
    typedef float __attribute__((ext_vector_type(2 * VEC))) vec_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x, o = blockIdx.y, z = blockIdx.z;
    const int c = tile * TILEC + (lane % LPR) * VEC;
    const int rsub = lane / LPR;
    const size_t zoff = (size_t)z * ROWS * COLS;
    const cx* __restrict__ sz = src + zoff;
    cx* __restrict__ dz = dst + zoff;
    constexpr int rstride = COMB ? ROWS / WROWS : 1;
    const int base = COMB ? o : o * WROWS;
    vec_t x[P];
#pragma unroll
    for (int v = 0; v < P; v++) {
        const int r = base + rstride * (wave * RPW + rsub + NW * RPW * v);
        const vec_t* p = reinterpret_cast<const vec_t*>(sz + ((unsigned)r * (unsigned)COLS + (unsigned)c));
        if constexpr (NT) x[v] = __builtin_nontemporal_load(p);
        else x[v] = *p;
    }
    __builtin_amdgcn_sched_barrier(0);  // like the transforms: every load of the lane is issued before the first store
#pragma unroll
    for (int v = 0; v < P; v++) {
        if ((v & 3) >= WFRAC) continue;
        const int r = base + rstride * (wave * RPW + rsub + NW * RPW * v);
        vec_t* p = reinterpret_cast<vec_t*>(dz + ((unsigned)r * (unsigned)COLS + (unsigned)c));
        if constexpr (NT) __builtin_nontemporal_store(x[v], p);
        else *p = x[v];
    }
}

template <int VEC, int LPR, int WROWS, int NTHR, bool COMB, bool NT, int WFRAC>
static void run(const char* name, const cx* a, cx* b) {
    constexpr int TILEC = LPR * VEC;
    dim3 grid(COLS / TILEC, ROWS / WROWS, Z);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((tile_copy<VEC, LPR, WROWS, NTHR, COMB, NT, WFRAC>), grid, dim3(NTHR), 0, 0, a, b);
    CK(hipEventRecord(e0));
    const int it = 6;
    for (int w = 0; w < it; w++) hipLaunchKernelGGL((tile_copy<VEC, LPR, WROWS, NTHR, COMB, NT, WFRAC>), grid, dim3(NTHR), 0, 0, a, b);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
    const double gb = (1.0 + WFRAC / 4.0) * Z * (double)ROWS * COLS * 8 / 1e9;
    printf("%-12s %2dB/lane seg=%4dB rows/WG=%3d thr=%4d %s %s wr=%d/4 : %7.3f ms  %7.1f GB/s\n", name, 8 * VEC, TILEC * 8, WROWS, NTHR,
           COMB ? "comb" : "cons", NT ? "nt" : "pl", WFRAC, ms, gb / ms * 1e3);
    fflush(stdout);
}

int main() {
    const size_t n = (size_t)Z * ROWS * COLS;
    cx *a, *b;
    CK(hipMalloc(&a, n * sizeof(cx))); CK(hipMalloc(&b, n * sizeof(cx)));
    CK(hipMemset(a, 1, n * sizeof(cx))); CK(hipMemset(b, 0, n * sizeof(cx)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nt = 0; nt < 2; nt++)
        for (int g : {4096, 65536}) {
            auto k = nt ? copy_flat<true> : copy_flat<false>;
            hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, (const f4*)a, (f4*)b, n / 2);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 6; i++) hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, (const f4*)a, (f4*)b, n / 2);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 6;
            printf("flat float4 copy grid=%6d %s: %7.3f ms %7.1f GB/s\n", g, nt ? "nt" : "pl", ms, 2.0 * n * 8 / 1e9 / ms * 1e3);
        }
    // pass B shape: 256 consecutive rows per workgroup, 512 threads
    run<1, 64, 256, 512, false, true, 4>("B today", a, b);
    run<1, 64, 256, 512, false, false, 4>("B today", a, b);
    run<2, 32, 256, 512, false, true, 4>("B 2row", a, b);
    run<2, 64, 128, 512, false, true, 4>("B wide128", a, b);
    run<2, 64, 256, 1024, false, true, 4>("B wide1024", a, b);
    run<2, 16, 256, 512, false, true, 4>("B 4row", a, b);
    run<1, 64, 256, 512, false, true, 1>("B today", a, b);
    run<2, 32, 256, 512, false, true, 1>("B 2row", a, b);
    run<2, 64, 256, 1024, false, true, 1>("B wide1024", a, b);
    // pass A shape: 128 rows 256 apart per workgroup, 256 threads
    run<1, 64, 128, 256, true, true, 4>("A today", a, b);
    run<1, 64, 128, 256, true, false, 4>("A today", a, b);
    run<2, 32, 128, 256, true, true, 4>("A 2row", a, b);
    run<2, 64, 128, 512, true, true, 4>("A wide512", a, b);
    run<2, 16, 128, 256, true, true, 4>("A 4row", a, b);
    // single-pass shapes (K3: 512 rows, 1024 threads; K5b-like: 1024 rows of 32 columns)
    run<1, 64, 512, 1024, false, true, 4>("K3 today", a, b);
    run<2, 32, 512, 1024, false, true, 4>("K3 2row", a, b);
    run<1, 32, 1024, 1024, false, true, 4>("K5b today", a, b);
    run<2, 16, 1024, 1024, false, true, 4>("K5b 4row", a, b);
    return 0;
}
