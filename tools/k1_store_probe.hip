// Microbenchmark (r5): could K1 store its band TRANSPOSED, so that the per-wave strided-axis transform (K2, a four-step
// with a 60 GB scratch round trip per pass) becomes a contiguous-axis row transform without scratch?
//
// A K1 workgroup holds ONE half row, so a transposed store is 5744 single 8-byte elements, each into a different 128-byte
// line; a line of the transposed band (16 consecutive facet rows of one column) is complete only when the workgroups of
// 16 rows have written it.  The question is whether the XCD's L2 merges those partial writes before they reach HBM.  This
// probe has K1's launch shape (one workgroup per half row, 512 threads, two per CU, 22 x 16-byte loads per lane of the
// facet row, a delay loop standing in for the transform with +-25 % jitter per workgroup) and three store forms:
//   0  today: contiguous parity-split band row                           (rows -> XCDs as today: row mod 8)
//   1  16-row interleaved layout [row / 16][column][row % 16], the 16 rows of a group on ONE XCD (blockIdx mapping)
//   2  the same layout with today's row -> XCD mapping (consecutive rows on different XCDs: no L2 can merge)
// Run under rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE to see the bytes that reach HBM.
//   hipcc --offload-arch=gfx950 -O3 tools/k1_store_probe.hip -o tools/k1_store_probe.bin && tools/k1_store_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));
constexpr int YB = 22528, BAND = 11488, HALF = 5744, NSEG = 22;

template <int MODE>
__global__ __launch_bounds__(512, 4) void probe(const v2* __restrict__ facet, v2* __restrict__ out, int spin, int nt_load) {
    const int t = threadIdx.x;
    const int b = blockIdx.x;
    int row, h;
    if (MODE == 1) {  // 16 consecutive rows (32 workgroups) on one XCD
        const int xcd = b & 7, q = b >> 3;
        const int g = xcd + 8 * (q >> 5), j = q & 31;
        row = 16 * g + (j >> 1);
        h = j & 1;
    } else {
        h = (b >> 3) & 1;
        row = ((b >> 4) << 3) + (b & 7);
    }
    if (row >= YB) return;
    const v4* __restrict__ in = reinterpret_cast<const v4*>(facet + (size_t)row * YB);
    v4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NSEG; s++) {
        const v4 x = nt_load ? __builtin_nontemporal_load(in + t + 512 * s) : in[t + 512 * s];
        acc += x;
    }
    // stand-in for the transform: a dependent chain, +-25 % jitter per workgroup
    const int n = spin + ((spin >> 2) * (int)((b * 2654435761u >> 13) & 3) >> 1) - (spin >> 2);
    float z = acc.x;
    for (int i = 0; i < n; i++) z = __builtin_fmaf(z, 1.0000001f, 1e-9f);
    const v2 val = {z + acc.y, acc.z + acc.w};
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const int c = t + 512 * i;  // column of this parity
        if (c < HALF) {
            if (MODE == 0) {
                out[(size_t)row * BAND + (size_t)h * HALF + c] = val;
            } else {
                out[((size_t)(row >> 4) * BAND + (size_t)(2 * c + h)) * 16 + (row & 15)] = val;
            }
        }
    }
}

int main(int argc, char** argv) {
    const int spin = argc > 1 ? atoi(argv[1]) : 3000;
    v2 *facet, *out;
    CK(hipMalloc(&facet, (size_t)YB * YB * sizeof(v2)));
    CK(hipMalloc(&out, (size_t)(YB + 16) * BAND * sizeof(v2)));
    CK(hipMemset(facet, 0, (size_t)YB * YB * sizeof(v2)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned blocks = (unsigned)((YB + 15) / 16 * 32 + 8 * 32);  // covers both mappings (mode 1 needs whole groups of 8 XCDs)
    auto run = [&](const char* name, auto kernel, int nt) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0, 0));
            for (int k = 0; k < 3; k++) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(512), 0, 0, facet, out, spin, nt);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms / 3 < best) best = ms / 3;
        }
        printf("%-78s %7.3f ms per facet\n", name, best);
        fflush(stdout);
    };
    printf("spin %d\n", spin);
    run("0: contiguous parity-split band row (today)", probe<0>, 0);
    run("1: 16-row interleaved transposed layout, the 16 rows of a line on one XCD", probe<1>, 0);
    run("2: 16-row interleaved transposed layout, today's row -> XCD mapping", probe<2>, 0);
    run("1 with non-temporal facet loads", probe<1>, 1);
    run("0 with non-temporal facet loads", probe<0>, 1);
    return 0;
}
