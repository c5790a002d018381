// Microbenchmark (r4 session 3): can the four-step intermediate of K2 live in the 256 MiB Infinity Cache?
// Pure-copy stand-ins for the two column passes with K2's access patterns and byte counts on the 64k workload
// (9 facets, 22528 data rows of 512 gathered columns out of band rows of 11488 columns, 32768-row intermediate, 28 % of
// the output rows kept), run (a) as today -- pass A of all facets into a 1.2 GB scratch, then pass B -- and (b) in
// chunks of one facet x Wc columns whose two passes run back to back, chunks alternating between two streams, each
// stream re-using one chunk-sized slot.  What the chunked schedule reaches with NO arithmetic at all bounds what the
// real kernels can gain from it.
//   hipcc --offload-arch=gfx950 -O3 tools/mall_pipe.hip -o tools/mall_pipe.bin && tools/mall_pipe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float2 cx;
typedef float v2 __attribute__((ext_vector_type(2)));
constexpr int Z = 9, YB = 22528, YN = 32768, M = 512, BANDC = 11488, N1 = 128, N2 = 256, QROWS = 9216;

template <bool NT> __device__ __forceinline__ v2 ld(const cx* p) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const v2*>(p));
    else return *reinterpret_cast<const v2*>(p);
}
template <bool NT> __device__ __forceinline__ void st(cx* p, v2 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<v2*>(p));
    else *reinterpret_cast<v2*>(p) = v;
}

// pass A stand-in: workgroup (tile of 64 columns, y2, z): reads band rows y2 + 256*y1 (y1 < 88: data rows), writes scratch
// rows k1*256 + y2 (k1 < 128).  256 threads = 4 waves, 32 slots per lane.
// RC / WC: comb (true) or consecutive (false) rows on the read / write side (a permuted band row order / a [y2][k1]
// scratch layout would make them consecutive)
template <bool NTS, bool RC = true, bool WC = true>
__global__ __launch_bounds__(256) void pass_a(const cx* __restrict__ band, cx* __restrict__ scr, int col0, int z0, unsigned spitch,
                                              long long sbs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int y2 = blockIdx.y, z = blockIdx.z + z0;
    const int col = blockIdx.x * 64 + lane;
    const cx* __restrict__ src = band + (size_t)z * YB * BANDC + 3000 + col0 + col;
    cx* __restrict__ dst = scr + (long long)blockIdx.z * sbs + col;
    v2 x[32];
#pragma unroll
    for (int v = 0; v < 32; v++) {
        const int y1 = wave + 4 * v;
        x[v] = v2{0.f, 0.f};
        if (y1 < YB / N2) x[v] = ld<true>(src + (size_t)(RC ? y2 + N2 * y1 : y2 * (YB / N2) + y1) * BANDC);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 0; v < 32; v++) {
        const int k1 = wave + 4 * v;
        st<NTS>(dst + (unsigned)(WC ? k1 * N2 + y2 : y2 * N1 + k1) * spitch, x[v] + v2{1.f, 1.f});
    }
}
// pass B stand-in: workgroup (tile, k1, z): reads scratch rows k1*256 + y2 (y2 < 256), writes 72 of them (28 %) to Q.
// 512 threads = 8 waves, 32 slots per lane.
template <bool NTS, bool RC = false>
__global__ __launch_bounds__(512) void pass_b(const cx* __restrict__ scr, cx* __restrict__ q, int col0, int z0, unsigned spitch,
                                              long long sbs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k1 = blockIdx.y, z = blockIdx.z + z0;
    const int col = blockIdx.x * 64 + lane;
    const cx* __restrict__ src = scr + (long long)blockIdx.z * sbs + col;
    cx* __restrict__ dst = q + (size_t)z * QROWS * M + col0 + col;
    v2 x[32];
#pragma unroll
    for (int v = 0; v < 32; v++) x[v] = ld<NTS>(src + (unsigned)(RC ? (wave + 8 * v) * N1 + k1 : k1 * N2 + wave + 8 * v) * spitch);
    __builtin_amdgcn_sched_barrier(0);
    v2 acc = v2{0.f, 0.f};
#pragma unroll
    for (int v = 0; v < 32; v++) {
        if (v < 9) st<true>(dst + (size_t)(k1 * 72 + wave + 8 * v) * M, x[v]);  // 72 of 256 rows kept
        else acc += x[v];
    }
    if (acc.x == 12345.f) st<true>(dst, acc);  // keeps the other loads alive
}

int main() {
    cx *band, *scr, *q;
    CK(hipMalloc(&band, (size_t)Z * YB * BANDC * sizeof(cx)));
    CK(hipMalloc(&scr, (size_t)Z * YN * M * sizeof(cx)));
    CK(hipMalloc(&q, (size_t)Z * QROWS * M * sizeof(cx)));
    CK(hipMemset(band, 0, (size_t)Z * YB * BANDC * sizeof(cx)));
    CK(hipMemset(scr, 0, (size_t)Z * YN * M * sizeof(cx)));
    hipStream_t s[2];
    for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    hipEvent_t e0, e1, ej[2];
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& x : ej) CK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
    const int waves = 6;
    auto timeit = [&](const char* name, auto&& body) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0, 0));
            for (int w = 0; w < waves; w++) body();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("%-58s %8.1f us per wave\n", name, 1e3 * best / waves);
        fflush(stdout);
    };
    // (a) today: one launch pair for all facets, non-temporal scratch
    timeit("monolithic, 1.2 GB scratch, non-temporal scratch", [&] {
        hipLaunchKernelGGL(pass_a<true>, dim3(M / 64, N2, Z), dim3(256), 0, 0, band, scr, 0, 0, (unsigned)M, (long long)YN * M);
        hipLaunchKernelGGL(pass_b<true>, dim3(M / 64, N1, Z), dim3(512), 0, 0, scr, q, 0, 0, (unsigned)M, (long long)YN * M);
    });
    timeit("monolithic, 1.2 GB scratch, cacheable scratch", [&] {
        hipLaunchKernelGGL(pass_a<false>, dim3(M / 64, N2, Z), dim3(256), 0, 0, band, scr, 0, 0, (unsigned)M, (long long)YN * M);
        hipLaunchKernelGGL(pass_b<false>, dim3(M / 64, N1, Z), dim3(512), 0, 0, scr, q, 0, 0, (unsigned)M, (long long)YN * M);
    });
    timeit("pass A alone (monolithic, nt)", [&] {
        hipLaunchKernelGGL(pass_a<true>, dim3(M / 64, N2, Z), dim3(256), 0, 0, band, scr, 0, 0, (unsigned)M, (long long)YN * M);
    });
    timeit("pass B alone (monolithic, nt)", [&] {
        hipLaunchKernelGGL(pass_b<true>, dim3(M / 64, N1, Z), dim3(512), 0, 0, scr, q, 0, 0, (unsigned)M, (long long)YN * M);
    });
    // row patterns: which side pays for a comb?
#define RUN_A(RC, WC) timeit("pass A alone: " #RC " read, " #WC " write (1 = comb)", [&] { \
        hipLaunchKernelGGL((pass_a<true, RC, WC>), dim3(M / 64, N2, Z), dim3(256), 0, 0, band, scr, 0, 0, (unsigned)M, (long long)YN * M); });
    RUN_A(1, 1) RUN_A(0, 1) RUN_A(1, 0) RUN_A(0, 0)
    timeit("pass B alone: comb read, consecutive write", [&] {
        hipLaunchKernelGGL((pass_b<true, true>), dim3(M / 64, N1, Z), dim3(512), 0, 0, scr, q, 0, 0, (unsigned)M, (long long)YN * M);
    });
    timeit("A(cons, cons) + B(comb read)", [&] {
        hipLaunchKernelGGL((pass_a<true, false, false>), dim3(M / 64, N2, Z), dim3(256), 0, 0, band, scr, 0, 0, (unsigned)M, (long long)YN * M);
        hipLaunchKernelGGL((pass_b<true, true>), dim3(M / 64, N1, Z), dim3(512), 0, 0, scr, q, 0, 0, (unsigned)M, (long long)YN * M);
    });
    timeit("A(cons read, comb write) + B(cons read)", [&] {
        hipLaunchKernelGGL((pass_a<true, false, true>), dim3(M / 64, N2, Z), dim3(256), 0, 0, band, scr, 0, 0, (unsigned)M, (long long)YN * M);
        hipLaunchKernelGGL((pass_b<true, false>), dim3(M / 64, N1, Z), dim3(512), 0, 0, scr, q, 0, 0, (unsigned)M, (long long)YN * M);
    });
    // (b) chunks of zc facets x Wc columns, two streams, one slot per stream
    for (int nstreams : {1, 2})
        for (int Wc : {128, 256, 512})
            for (int zc : {1, 3}) {
                if (zc == 3 && Wc != 128) continue;
                char name[128];
                snprintf(name, sizeof name, "chunks of %d facet(s) x %3d columns (%3.0f MB), %d stream(s), cacheable", zc, Wc,
                         zc * YN * (double)Wc * 8 / 1e6, nstreams);
                timeit(name, [&] {
                    CK(hipEventRecord(ej[0], 0));
                    for (int k = 0; k < nstreams; k++) CK(hipStreamWaitEvent(s[k], ej[0], 0));
                    int i = 0;
                    for (int z0 = 0; z0 < Z; z0 += zc)
                        for (int c0 = 0; c0 < M; c0 += Wc, i++) {
                            hipStream_t st2 = s[i % nstreams];
                            cx* slot = scr + (size_t)(i % nstreams) * zc * YN * Wc;
                            hipLaunchKernelGGL(pass_a<false>, dim3(Wc / 64, N2, zc), dim3(256), 0, st2, band, slot, c0, z0, (unsigned)Wc,
                                               (long long)YN * Wc);
                            hipLaunchKernelGGL(pass_b<false>, dim3(Wc / 64, N1, zc), dim3(512), 0, st2, slot, q, c0, z0, (unsigned)Wc,
                                               (long long)YN * Wc);
                        }
                    for (int k = 0; k < nstreams; k++) {
                        CK(hipEventRecord(ej[k], s[k]));
                        CK(hipStreamWaitEvent(0, ej[k], 0));
                    }
                });
            }
    return 0;
}
