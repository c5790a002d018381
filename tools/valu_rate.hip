// VALU issue-rate microbenchmark for gfx950: scalar-float vs packed-float (v_pk_*) add / fma throughput.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate.bin valu_rate.hip && ./valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int NACC = 16, ITERS = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
    float s[NACC];
    f32x2 p[NACC];
    for (int i = 0; i < NACC; i++) {
        s[i] = threadIdx.x * 1e-3f + i;
        p[i] = f32x2{s[i], s[i] + 0.5f};
    }
    const f32x2 pa = {a, a * 1.5f}, pb = {b, b * 0.5f};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            if (MODE == 0) s[i] = __builtin_fmaf(s[i], a, b);                 // v_fma_f32
            if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], pa, pb);    // v_pk_fma_f32
            if (MODE == 2) s[i] = s[i] + a;                                   // v_add_f32
            if (MODE == 3) p[i] = p[i] + pa;                                  // v_pk_add_f32
            if (MODE == 4) s[i] = s[i] * a;                                   // v_mul_f32
            if (MODE == 5) p[i] = p[i] * pa;                                  // v_pk_mul_f32
        }
    }
    float r = 0;
    for (int i = 0; i < NACC; i++) r += s[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
double run(const char* name, int lanes_ops) {
    float* d;
    const int blocks = 256 * 8, threads = 256;
    hipMalloc(&d, blocks * threads * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 1.0001f, 1e-7f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 1.0001f, 1e-7f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double winstr = (double)blocks * (threads / 64) * NACC * ITERS;  // wave-instructions
    const double per_simd_per_clk = winstr / (ms * 1e-3) / (256 * 4) / 2.4e9;
    printf("%-14s %8.3f ms  %7.1f G wave-instr/s  %.3f wave-instr/clk/SIMD (at 2.4 GHz) = %.1f cycles/instr  %6.1f T lane-ops/s\n", name, ms,
           winstr / (ms * 1e-3) / 1e9, per_simd_per_clk, 1.0 / per_simd_per_clk, winstr * 64 * lanes_ops / (ms * 1e-3) / 1e12);
    hipFree(d);
    return ms;
}

int main() {
    run<0>("v_fma_f32", 1);
    run<1>("v_pk_fma_f32", 2);
    run<2>("v_add_f32", 1);
    run<3>("v_pk_add_f32", 2);
    run<4>("v_mul_f32", 1);
    run<5>("v_pk_mul_f32", 2);
    return 0;
}
