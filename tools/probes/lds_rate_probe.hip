// LDS rates with ONE 512-thread workgroup per CU (8 waves, 2 per SIMD: the whole-row K1's occupancy): 128 KB moved per
// step, by instruction width and layout.  hipcc --offload-arch=gfx950 -O3 -o lds_rate_probe lds_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: 32 x ds_write_b64, natural layout (lane-consecutive elements, stride 544 elements between the lane's values)
// MODE 1: 16 x ds_write_b128 (two adjacent elements), lane stride 34 elements (exchange 1 of the kernel today)
// MODE 2: 32 x ds_write_b64, lane-major destination (stride 272 B between consecutive lanes)
// MODE 3: 32 x ds_read_b64 natural (gather today)
// MODE 4: 16 x ds_read_b128 lane-major (32 consecutive elements per lane, lane pitch 272 B)
// MODE 5: 32 x ds_read_b64 lane-major
template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(float* out, int iters, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    f2 x[32];
    for (int i = 0; i < 32; i++) x[i] = f2{(float)(t + i), (float)i};
    f2* b2 = reinterpret_cast<f2*>(smem);
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if constexpr (MODE == 0) {
            const int p0 = t + (t >> 4);
#pragma unroll
            for (int v = 0; v < 32; v++) b2[p0 + v * 544] = x[v];
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f4 q = {x[u * 16 + r].x, x[u * 16 + r].y, x[u * 16 + r + 1].x, x[u * 16 + r + 1].y};
                    *reinterpret_cast<f4*>(&b2[34 * t + 17 * u + r]) = q;
                }
        } else if constexpr (MODE == 2) {
            const int k = t & 15, a = t >> 4;
            unsigned char* base = smem + k * 272 + a * 8;
#pragma unroll
            for (int r = 0; r < 32; r++) *reinterpret_cast<f2*>(base + r * 4352 + (r >> 1) * 16) = x[r];
        } else if constexpr (MODE == 3) {
            const int p0 = t + (t >> 4);
#pragma unroll
            for (int v = 0; v < 32; v++) x[v] += b2[p0 + v * 544];
        } else if constexpr (MODE == 4) {
            const unsigned char* base = smem + t * 272 + (t >> 5) * 16;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                f4 q = *reinterpret_cast<const f4*>(base + 16 * i);
                x[2 * i] += f2{q.x, q.y};
                x[2 * i + 1] += f2{q.z, q.w};
            }
        } else {
            const unsigned char* base = smem + t * 272 + (t >> 5) * 16;
#pragma unroll
            for (int i = 0; i < 32; i++) x[i] += *reinterpret_cast<const f2*>(base + 8 * i);
        }
        __syncthreads();
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 32; i++) s += x[i].x + x[i].y;
    out[blockIdx.x * 512 + t] = s + b2[t].x;
    if (t == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 400, lds = 139520 + 512;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), lds, 0, out, 10, cyc);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), lds, 0, out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += (double)v; c /= 256.0 * iters;
    printf("%-58s %8.0f cycles per 128 KB step  (%5.1f B/clk per CU)  %.3f us per step\n", name, c, 131072.0 / c, ms * 1e3 / iters);
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    run<0>("32 x ds_write_b64 natural (lane-consecutive)", out, cyc);
    run<1>("16 x ds_write_b128 exchange-1 layout", out, cyc);
    run<2>("32 x ds_write_b64 lane-major destination", out, cyc);
    run<3>("32 x ds_read_b64 natural (gather today)", out, cyc);
    run<4>("16 x ds_read_b128 lane-major", out, cyc);
    run<5>("32 x ds_read_b64 lane-major", out, cyc);
    return 0;
}
