// Microbenchmark: how fast can the MI355X move a [rows, cols] complex64 array
// when every workgroup touches `nseg` row segments of `segw` columns that are
// `rstride` rows apart (the access pattern of one pass of the two-pass axis-0
// transform), compared with a plain streaming copy.
//   hipcc --offload-arch=gfx950 -O3 tools/strided_bw.hip -o /tmp/strided_bw && /tmp/strided_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float2 cx;

__global__ void copy_plain(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) b[i] = a[i];
}

// WG = 256 threads.  VEC = complex elements per lane per access (1 -> 8 B, 2 -> 16 B).
// Tile: segw = (256 / TROWS) * VEC columns; thread (lane-in-row, trow) handles rows
// (trow + TROWS*v) * rstride + y2, v < P.  blockIdx -> (tile column, y2) with colfast choice.
template <int VEC, int TROWS, int P>
__global__ __launch_bounds__(256) void copy_strided(const cx* __restrict__ src, cx* __restrict__ dst, int cols,
                                                    int pitch, int rstride, int ntilec, int n_y2, int colfast) {
    constexpr int LPR = 256 / TROWS;  // lanes per row
    int tile, y2;
    if (colfast) { tile = blockIdx.x % ntilec; y2 = blockIdx.x / ntilec; }
    else { y2 = blockIdx.x % n_y2; tile = blockIdx.x / n_y2; }
    const int lane = threadIdx.x % LPR, trow = threadIdx.x / LPR;
    const int c = (tile * LPR + lane) * VEC;
    if (c >= cols) return;
    typedef float __attribute__((ext_vector_type(2 * VEC))) vec_t;
    vec_t x[P];
#pragma unroll
    for (int v = 0; v < P; v++) {
        size_t row = (size_t)(trow + TROWS * v) * rstride + y2;
        x[v] = *reinterpret_cast<const vec_t*>(src + row * pitch + c);
    }
#pragma unroll
    for (int v = 0; v < P; v++) {
        size_t row = (size_t)(trow + TROWS * v) * rstride + y2;
        *reinterpret_cast<vec_t*>(dst + row * pitch + c) = x[v];
    }
}

template <int VEC, int TROWS, int P>
static void run(const char* name, const cx* a, cx* b, int rows, int cols, int pitch, int colfast) {
    constexpr int LPR = 256 / TROWS;
    const int nrows_wg = TROWS * P;        // rows per WG (= n1)
    const int rstride = rows / nrows_wg;   // = n2
    const int ntilec = (cols + LPR * VEC - 1) / (LPR * VEC);
    dim3 grid((unsigned)ntilec * rstride);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++)
        hipLaunchKernelGGL((copy_strided<VEC, TROWS, P>), grid, dim3(256), 0, 0, a, b, cols, pitch, rstride, ntilec, rstride, colfast);
    CK(hipEventRecord(e0));
    const int it = 5;
    for (int w = 0; w < it; w++)
        hipLaunchKernelGGL((copy_strided<VEC, TROWS, P>), grid, dim3(256), 0, 0, a, b, cols, pitch, rstride, ntilec, rstride, colfast);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
    double gb = 2.0 * rows * (double)cols * 8 / 1e9;
    printf("%-34s segw=%4d B rows/WG=%3d rstride=%5d pitch=%6d colfast=%d : %7.3f ms  %7.1f GB/s\n", name, LPR * VEC * 8,
           nrows_wg, rstride, pitch, colfast, ms, gb / ms * 1e3);
}

int main() {
    const int rows = 32768, cols = 22528;
    for (int pitch : {22528, 22528 + 48, 24576}) {
        size_t n = (size_t)rows * pitch;
        cx *a, *b;
        CK(hipMalloc(&a, n * sizeof(cx))); CK(hipMalloc(&b, n * sizeof(cx)));
        CK(hipMemset(a, 1, n * sizeof(cx))); CK(hipMemset(b, 0, n * sizeof(cx)));
        if (pitch == 22528) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            size_t n4 = n / 2;
            for (int g : {2048, 8192, 65536}) {
                hipLaunchKernelGGL(copy_plain, dim3(g), dim3(256), 0, 0, (const float4*)a, (float4*)b, n4);
                CK(hipEventRecord(e0));
                for (int i = 0; i < 5; i++) hipLaunchKernelGGL(copy_plain, dim3(g), dim3(256), 0, 0, (const float4*)a, (float4*)b, n4);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
                printf("plain float4 copy grid=%6d: %7.3f ms %7.1f GB/s\n", g, ms, 2.0 * n * 8 / 1e9 / ms * 1e3);
            }
        }
        for (int colfast : {1, 0}) {
            run<1, 8, 16>("8B/lane 32col x128rows (pass A)", a, b, rows, cols, pitch, colfast);
            run<2, 8, 16>("16B/lane 64col x128rows", a, b, rows, cols, pitch, colfast);
            run<2, 4, 16>("16B/lane 128col x64rows", a, b, rows, cols, pitch, colfast);
            run<2, 16, 16>("16B/lane 32col x256rows", a, b, rows, cols, pitch, colfast);
            run<1, 16, 16>("8B/lane 16col x256rows (pass B)", a, b, rows, cols, pitch, colfast);
            run<2, 8, 8>("16B/lane 64col x64rows", a, b, rows, cols, pitch, colfast);
            run<2, 2, 16>("16B/lane 256col x32rows", a, b, rows, cols, pitch, colfast);
        }
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
