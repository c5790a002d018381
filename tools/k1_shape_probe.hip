// Microbenchmark (r5): would K1 gain from FOUR quarter-row workgroups per CU instead of TWO half-row workgroups?
//
// K1's time is (rows per CU) x (lifetime of a row's workgroups): the register file of a CU holds one 32768-point row, as
// two 512-thread workgroups that march through load / butterfly / exchange phases nearly in step, so that the vector
// ALUs (60 % busy), the LDS (18 %) and the vector-memory path (21 %) are used one after the other.  Four 256-thread
// workgroups (radix-4 decimation in frequency across workgroups, 8192-point problems, 33 KB of LDS each) hold the same
// row in the same registers but are four independent instruction streams per SIMD instead of two, with barriers among
// four waves instead of eight -- at the price of loading (and windowing) the row four times instead of twice.
//
// The probe keeps what decides the timing and drops what does not: the REAL transform engine of the library
// (swiftly_fft.h: butterflies, twiddles, LDS exchanges, barriers) on 16384 / 8192 points, the real number and width of the
// vector-memory instructions per lane (S = 2: 22 x 16 B data + 22 x 8 B window; S = 4: 44 x 16 B + 22 x 16 B), the
// first-stage additions, the inter-part twiddle products, and a band store of 12 of the lane's 32 outputs.  The values
// are meaningless (no load map, no band arithmetic).  S = 2 is the calibration: it should take what K1 takes.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/k1_shape_probe.hip -o tools/k1_shape_probe.bin
// Second use (ablations of the S = 2 shape, V = bit mask): 1 no facet / window loads, 2 no stores, 4 no butterflies,
// 8 no LDS exchanges, 16 every workgroup loads only HALF of the segments (what a load shared between the two halves of a
// row would cost), 32 no window loads; -DPROBE_NOBAR: workgroup barriers replaced by wave barriers (wrong results; timing).
#include <hip/hip_runtime.h>
#ifdef PROBE_NOBAR
#define __syncthreads() __builtin_amdgcn_wave_barrier()
#endif

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../ska-sdp-distributed-fourier-transform_amd/csrc/swiftly_rowpass.h"

using namespace swf;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

constexpr int YB = 22528, NFULL = 32768, NSEG = 22, OUT_PITCH = 12288, KEEP = 12;

template <int S>
struct Shape;
template <>
struct Shape<2> {
    using G = RGeoPre<14, 5, true>;
};
template <>
struct Shape<4> {
    using G = RGeoPre<13, 5, true>;
};

template <int S, int V = 0>
__global__ __launch_bounds__(Shape<S>::G::NT, 4) void probe(const cx<float>* __restrict__ facet, const float* __restrict__ win,
                                                            cx<float>* __restrict__ out, const cx<float>* __restrict__ tw_part,
                                                            const cx<float>* __restrict__ tw_full, int nrows, float scale) {
    using G = typename Shape<S>::G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int P = G::P, T = G::T, H = G::N;
    constexpr int LOGR1 = G::LOGN % G::LOGP, R1 = 1 << LOGR1, NB = P / R1, SEG = H / R1;  // S=2: 16, 2, 1024; S=4: 8, 4, 1024
    static_assert(SEG == 1024 && NB * T == SEG, "segments of 1024 points, NB adjacent points per lane");
    constexpr int LOGS = S == 2 ? 1 : 2;
    const int t = threadIdx.x, b = blockIdx.x;
    const int h = (b >> 3) & (S - 1);
    int row = ((b >> (3 + LOGS)) << 3) + (b & 7);
    if (row >= nrows) return;
    row = __builtin_amdgcn_readfirstlane(row);
    const char* inb = reinterpret_cast<const char*>(facet + (long long)row * YB);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(inb), (short)0, YB << 3, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(win), (short)0, YB << 2, 0x00020000);
    const float sgn = (h & 1) ? -1.f : 1.f, sgn2 = (h & 2) ? -1.f : 1.f;

    cx<float> x[P];
    float wpairq[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    static_for<0, R1>([&](auto rI) {
        constexpr int r = decltype(rI)::value;
        cx<float> acc[NB];
        static_for<0, NB>([&](auto uI) { acc[decltype(uI)::value] = cx<float>{0.f, 0.f}; });
        static_for<0, S>([&](auto qI) {
            constexpr int q = decltype(qI)::value;
            constexpr int seg = r + R1 * q;
            if constexpr (seg < NSEG && !(V & 1) && (!(V & 16) || (seg & 1) == 0)) {
                const unsigned e0 = (unsigned)(NB * t + SEG * seg);  // first of the lane's NB adjacent elements
                float w[NB];
                if constexpr (V & 32) {
                    static_for<0, NB>([&](auto uI) { w[decltype(uI)::value] = scale; });
                } else if constexpr ((V & 64) && NB == 2) {
                    // window table re-laid out per pair of segments: ONE 16-byte load per lane and segment pair
                    // ([pair][t][seg parity][u]) instead of two 8-byte loads -- half the window load instructions, same bytes
                    if constexpr ((seg & 1) == 0) {
                        const f32x4 wv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(((seg >> 1) * 512 + t) << 4), 0, 0));
                        wpairq[q][0] = wv.z; wpairq[q][1] = wv.w;
                        w[0] = wv.x; w[1] = wv.y;
                    } else {
                        w[0] = wpairq[q][0]; w[1] = wpairq[q][1];
                    }
                } else if constexpr (NB == 2) {
                    const f32x2 wv = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_w, (int)(e0 << 2), 0, 0));
                    w[0] = wv.x; w[1] = wv.y;
                } else {
                    const f32x4 wv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(e0 << 2), 0, 0));
                    w[0] = wv.x; w[1] = wv.y; w[2] = wv.z; w[3] = wv.w;
                }
                static_for<0, NB / 2>([&](auto pI) {
                    constexpr int p = decltype(pI)::value;
                    const f32x4 val = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)((e0 + 2 * p) << 3), 0, 0));
                    const f32x2 p0 = {val.x, val.y}, p1 = {val.z, val.w};
                    const f32x2 w0 = {w[2 * p], w[2 * p]}, w1 = {w[2 * p + 1], w[2 * p + 1]};
                    // window product, then the first-stage term (+-1, +-i of the real kernel: one packed operation)
                    const f32x2 s = (q & 1) ? f32x2{sgn, sgn} : ((q & 2) ? f32x2{sgn2, sgn2} : f32x2{1.f, 1.f});
                    acc[2 * p] = pkc(__builtin_elementwise_fma(p0 * w0, s, pkv(acc[2 * p])));
                    acc[2 * p + 1] = pkc(__builtin_elementwise_fma(p1 * w1, s, pkv(acc[2 * p + 1])));
                });
            }
        });
        static_for<0, NB>([&](auto uI) {
            constexpr int u = decltype(uI)::value;
            x[u + NB * r] = acc[u];
            if constexpr (V & 1) x[u + NB * r] = cx<float>{scale * (float)(t + u + r), scale * (float)(t - r)};
        });
    });
    if (h) {  // inter-part twiddle: W_N^(h (NB t + u)) from the table, times a compile-time constant per r
        cx<float> wt[NB];
        static_for<0, NB / 2>([&](auto pI) {
            constexpr int p = decltype(pI)::value;
            const f32x4 v = *reinterpret_cast<const f32x4*>(tw_full + ((h * NB * t + 2 * p) & (NFULL - 1)));
            wt[2 * p] = cx<float>{v.x, v.y};
            wt[2 * p + 1] = cx<float>{v.z, v.w};
        });
        static_for<0, R1>([&](auto rI) {
            constexpr int r = decltype(rI)::value;
            static_for<0, NB>([&](auto uI) {
                constexpr int u = decltype(uI)::value;
                x[u + NB * r] = mul_w64<float, (64 / R1 / S) * r>(cmul(x[u + NB * r], wt[u]));
            });
        });
    }
    const cx<float> rphi = tw_full[(unsigned)(1024 * (S * t + h)) & (unsigned)(NFULL - 1)];
    cx<float>* orow = out + (long long)row * OUT_PITCH + h * (OUT_PITCH / S);
    const f32x2 sc = {scale, scale};
    auto fin = [&](int, cx<float> v, auto sI) {
        constexpr int s = decltype(sI)::value;
        if constexpr (s < KEEP) {
            v = cmul(v, rphi);
            const f32x2 val = pkv(v) * sc;
            if constexpr (V & 2) {
                if (val.x == 123456.75f) *reinterpret_cast<f32x2*>(orow + s * T + t) = val;  // never: keeps the value alive
            } else {
                *reinterpret_cast<f32x2*>(orow + s * T + t) = val;
            }
        }
    };
    if constexpr (V == 0) {
        fft_phases_pair<G, float>(x, t, smem, tw_part, fin);
    } else {  // the same schedule (swiftly_fft.h: fft_phases_pair / fft_phases with the twiddle preload), stage by stage
        constexpr int L1 = LOGR1, L2 = L1 + G::LOGP;
        static_assert(L2 + G::LOGP == G::LOGN, "three phases");
        if constexpr (!(V & 4)) phase_compute<G, float, 0, L1>(x, t, tw_part);
        cx<float> nxt[G::LOGP];
        static_for<0, G::LOGP>([&](auto bI) {
            constexpr int b2 = decltype(bI)::value;
            nxt[b2] = tw_part[(((t & ((1 << L1) - 1)) << (G::LOGN - L1 - G::LOGP)) << b2) & (G::N - 1)];
        });
        if constexpr (!(V & 8)) phase_exchange<G, float, 0, L1, true>(x, t, 0, false, smem);
        if constexpr (!(V & 4)) phase_compute<G, float, L1, G::LOGP>(x, t, tw_part, nxt);
        static_for<0, G::LOGP>([&](auto bI) {
            constexpr int b2 = decltype(bI)::value;
            nxt[b2] = tw_part[(((t & ((1 << L2) - 1)) << (G::LOGN - L2 - G::LOGP)) << b2) & (G::N - 1)];
        });
        if constexpr (!(V & 8)) phase_exchange<G, float, L1, G::LOGP>(x, t, 0, false, smem);
        if constexpr (!(V & 4)) phase_compute<G, float, L2, G::LOGP>(x, t, tw_part, nxt);
        if constexpr (V & 4) {  // keep every register alive without the butterflies
            static_for<1, P>([&](auto vI) { x[0] = x[0] + x[decltype(vI)::value]; });
        }
        phase_scatter<G, float, L2, G::LOGP>(x, t, fin);
    }
}


// Third use (r5): ONE 1024-thread workgroup per row -- waves 0-7 transform the even outputs, waves 8-15 the odd ones, and the
// two halves SHARE the load phase: every lane loads (and windows) half of the lane pair's 16 first-stage slots, forms
// a + b and a - b for them, keeps the one of its own half and hands the other to its partner lane through the idle
// exchange buffer of the partner's half (8 x 16 bytes out, 8 x 16 bytes in, two workgroup barriers).  Per row: 22 x 16-byte
// data loads + 22 x 8-byte window loads per lane PAIR instead of per lane, the same HBM bytes; the price: one workgroup per
// CU with all 16 waves in the same phase, barriers among 16 waves, 128 KB through the LDS.  W = 1: window through 16-byte
// loads of a re-laid-out table on top.
constexpr int pair_owner(int r) {
    constexpr int NB1 = NSEG - 16;
    return r < NB1 ? (r < NB1 / 2 ? 0 : 1) : ((r - NB1) < (16 - NB1) / 2 ? 0 : 1);
}
constexpr int pair_slot_index(int r) {  // position of slot r among the slots of its owner
    int k = 0;
    for (int i = 0; i < r; i++) k += pair_owner(i) == pair_owner(r);
    return k;
}
template <int W>
__global__ __launch_bounds__(1024) void probe_pair(const cx<float>* __restrict__ facet, const float* __restrict__ win,
                                                    cx<float>* __restrict__ out, const cx<float>* __restrict__ tw_part,
                                                    const cx<float>* __restrict__ tw_full, int nrows, float scale) {
    using G = Shape<2>::G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int P = G::P, T = G::T, R1 = 16, SEG = 1024, NB1 = NSEG - R1;  // NB1 slots have an a and a b segment
    const int h = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 9), t = threadIdx.x & 511;
    const int row = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
    if (row >= nrows) return;
    const char* inb = reinterpret_cast<const char*>(facet + (long long)row * YB);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(inb), (short)0, YB << 3, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(win), (short)0, YB << 2, 0x00020000);
    unsigned char* mine = smem + h * G::LDS_BYTES;          // exchange buffer of this half
    unsigned char* theirs = smem + (1 - h) * G::LDS_BYTES;  // ... of the partner half
    // slot r of the first stage (points 2t, 2t+1 of segments r and r + 16) belongs to half OWNER(r): 3 + 5 slots each
    cx<float> x[P];
    constexpr auto owner = [](int r) constexpr { return pair_owner(r); };
    constexpr auto slot_index = [](int r) constexpr { return pair_slot_index(r); };
    // both halves run the same code on their own slots: hand-unrolled per half so that the slot numbers stay compile-time
    auto load_phase = [&](auto HI) {
        constexpr int HH = decltype(HI)::value;
        f32x4 wcarry = {0.f, 0.f, 0.f, 0.f};
        int wcount = 0;
        (void)wcount;
        static_for<0, R1>([&](auto rI) {
            constexpr int r = decltype(rI)::value;
            if constexpr (owner(r) == HH) {
                constexpr int k = slot_index(r);
                cx<float> a[2][2];
                static_for<0, 2>([&](auto qI) {
                    constexpr int q = decltype(qI)::value;
                    constexpr int seg = r + R1 * q;
                    if constexpr (seg < NSEG) {
                        const unsigned e0 = (unsigned)(2 * t + SEG * seg);
                        f32x2 wv;
                        if constexpr (W) {
                            if constexpr (r < NB1) {  // one 16-byte load: the windows of segments r and r + 16
                                if constexpr (q == 0) {
                                    wcarry = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)((r * 512 + t) << 4), 0, 0));
                                    wv = f32x2{wcarry.x, wcarry.y};
                                } else {
                                    wv = f32x2{wcarry.z, wcarry.w};
                                }
                            } else if constexpr ((k & 1) == 1) {  // a-only slots: one load per two slots of the lane
                                wcarry = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)((r * 512 + t) << 4), 0, 0));
                                wv = f32x2{wcarry.x, wcarry.y};
                            } else {
                                wv = f32x2{wcarry.z, wcarry.w};
                            }
                        } else {
                            wv = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_w, (int)(e0 << 2), 0, 0));
                        }
                        const f32x4 val = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)(e0 << 3), 0, 0));
                        a[q][0] = pkc(f32x2{val.x, val.y} * f32x2{wv.x, wv.x});
                        a[q][1] = pkc(f32x2{val.z, val.w} * f32x2{wv.y, wv.y});
                    }
                });
                cx<float> keep[2], give[2];
                static_for<0, 2>([&](auto uI) {
                    constexpr int u = decltype(uI)::value;
                    if constexpr (r < NB1) {
                        const cx<float> sum = a[0][u] + a[1][u], dif = a[0][u] - a[1][u];
                        keep[u] = HH ? dif : sum;
                        give[u] = HH ? sum : dif;
                    } else {
                        keep[u] = a[0][u];
                        give[u] = a[0][u];
                    }
                    x[u + 2 * r] = keep[u];
                });
                *reinterpret_cast<f32x4*>(theirs + ((k * 512 + t) << 4)) = f32x4{give[0].x, give[0].y, give[1].x, give[1].y};
            }
        });
    };
    auto take_phase = [&](auto HI) {  // the slots the partner loaded
        constexpr int HH = decltype(HI)::value;
        static_for<0, R1>([&](auto rI) {
            constexpr int r = decltype(rI)::value;
            if constexpr (owner(r) != HH) {
                constexpr int k = slot_index(r);
                const f32x4 v = *reinterpret_cast<const f32x4*>(mine + ((k * 512 + t) << 4));
                x[2 * r] = cx<float>{v.x, v.y};
                x[2 * r + 1] = cx<float>{v.z, v.w};
            }
        });
    };
    if (h == 0) load_phase(std::integral_constant<int, 0>{}); else load_phase(std::integral_constant<int, 1>{});
    __syncthreads();
    if (h == 0) take_phase(std::integral_constant<int, 0>{}); else take_phase(std::integral_constant<int, 1>{});
    __syncthreads();
    if (h) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(tw_full + ((2 * t) & (NFULL - 1)));
        const cx<float> w0 = {v.x, v.y}, w1 = {v.z, v.w};
        static_for<0, R1>([&](auto rI) {
            constexpr int r = decltype(rI)::value;
            x[2 * r] = mul_w64<float, 2 * r>(cmul(x[2 * r], w0));
            x[2 * r + 1] = mul_w64<float, 2 * r>(cmul(x[2 * r + 1], w1));
        });
    }
    const cx<float> rphi = tw_full[(unsigned)(1024 * (2 * t + h)) & (unsigned)(NFULL - 1)];
    cx<float>* orow = out + (long long)row * OUT_PITCH + h * (OUT_PITCH / 2);
    const f32x2 sc = {scale, scale};
    fft_phases_pair<G, float>(x, t, mine, tw_part, [&](int, cx<float> v, auto sI) {
        constexpr int s = decltype(sI)::value;
        if constexpr (s < KEEP) {
            v = cmul(v, rphi);
            *reinterpret_cast<f32x2*>(orow + s * T + t) = pkv(v) * sc;
        }
    });
}

template <int W>
static float run_pair(const cx<float>* facet, const float* win, cx<float>* out, const cx<float>* tw_part, const cx<float>* tw_full,
                      int launches) {
    using G = Shape<2>::G;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_pair<W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * G::LDS_BYTES)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < launches; i++)
        hipLaunchKernelGGL((probe_pair<W>), dim3(YB), dim3(1024), 2 * G::LDS_BYTES, 0, facet, win, out, tw_part, tw_full, YB, 0.5f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / launches;
}

__global__ void fill(float* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (float)((i * 2654435761u >> 8) & 0xffff) * (1.f / 65536.f) - 0.5f;
}

static std::vector<cx<float>> table(int n) {
    std::vector<cx<float>> tw(n);
    for (int k = 0; k < n; k++) {
        const double a = -2.0 * M_PI * k / n;
        tw[k] = cx<float>{(float)cos(a), (float)sin(a)};
    }
    return tw;
}

template <int S, int V = 0>
static float run(const cx<float>* facet, const float* win, cx<float>* out, const cx<float>* tw_part, const cx<float>* tw_full,
                 int launches) {
    using G = typename Shape<S>::G;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<S, V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
    const unsigned blocks = (unsigned)((YB + 7) / 8 * 8 * S);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < launches; i++)
        hipLaunchKernelGGL((probe<S, V>), dim3(blocks), dim3(G::NT), G::LDS_BYTES, 0, facet, win, out, tw_part, tw_full, YB, 0.5f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / launches;
}

int main() {
    cx<float>*facet, *out, *tw14, *tw13, *twf;
    float* win;
    CK(hipMalloc(&facet, (size_t)YB * YB * 8));
    CK(hipMalloc(&out, (size_t)YB * OUT_PITCH * 8));
    CK(hipMalloc(&win, (size_t)YB * 4));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (float*)facet, (size_t)YB * YB * 2);
    hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, 0, win, (size_t)YB);
    auto t14 = table(16384), t13 = table(8192), tf = table(NFULL);
    CK(hipMalloc(&tw14, t14.size() * 8));
    CK(hipMalloc(&tw13, t13.size() * 8));
    CK(hipMalloc(&twf, tf.size() * 8));
    CK(hipMemcpy(tw14, t14.data(), t14.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(tw13, t13.data(), t13.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(twf, tf.data(), tf.size() * 8, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    int occ2 = 0, occ4 = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ2, probe<2>, Shape<2>::G::NT, Shape<2>::G::LDS_BYTES));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ4, probe<4>, Shape<4>::G::NT, Shape<4>::G::LDS_BYTES));
    printf("workgroups per CU: S=2 (512 threads, %zu B LDS): %d   S=4 (256 threads, %zu B LDS): %d\n", (size_t)Shape<2>::G::LDS_BYTES,
           occ2, (size_t)Shape<4>::G::LDS_BYTES, occ4);
    run<2>(facet, win, out, tw14, twf, 2);
    run<4>(facet, win, out, tw13, twf, 2);
    for (int rep = 0; rep < 3; rep++) {
        const float a = run<2>(facet, win, out, tw14, twf, 9);
        const float b = run<4>(facet, win, out, tw13, twf, 9);
        printf("rep %d: two half-row workgroups %.4f ms per facet    four quarter-row workgroups %.4f ms per facet\n", rep, a, b);
    }
    for (int rep = 0; rep < 3; rep++) {
        run_pair<0>(facet, win, out, tw14, twf, 2);
        const float a = run<2>(facet, win, out, tw14, twf, 9), b = run_pair<0>(facet, win, out, tw14, twf, 9);
        run_pair<1>(facet, win, out, tw14, twf, 2);
        const float c = run<2, 64>(facet, win, out, tw14, twf, 9), d = run_pair<1>(facet, win, out, tw14, twf, 9);
        printf("rep %d: two 512-thread workgroups %.4f   one 1024-thread workgroup, shared loads %.4f   |  16-byte window loads: %.4f  %.4f ms per facet\n", rep, a, b, c, d);
    }
    if (getenv("PROBE_PAIR_ONLY")) return 0;
#define ABL(V, what)                                                                                   \
    {                                                                                                  \
        run<2, V>(facet, win, out, tw14, twf, 2);                                                      \
        const float a = run<2, V>(facet, win, out, tw14, twf, 9), b2 = run<2, V>(facet, win, out, tw14, twf, 9); \
        printf("V=%2d %-72s %.4f %.4f ms per facet\n", V, what, a, b2);                                \
    }
    ABL(0, "full")
    ABL(16, "each workgroup loads half of the segments")
    ABL(32, "no window loads")
    ABL(48, "half of the segments, no window loads")
    ABL(64, "window loads as 11 x 16 bytes (re-laid-out table) instead of 22 x 8 bytes")
    ABL(28, "half of the segments, no butterflies, no exchanges")
    ABL(44, "no window loads, no butterflies, no exchanges")
    ABL(14, "no stores, no butterflies, no exchanges (loads + first stage)")
    ABL(13, "no loads, no butterflies, no exchanges (stores only)")
    ABL(1, "no facet / window loads")
    ABL(2, "no stores")
    ABL(3, "no loads, no stores (transform only)")
    ABL(8, "no LDS exchanges")
    ABL(4, "no butterflies")
    ABL(12, "no butterflies, no exchanges (memory skeleton + first stage)")
    ABL(11, "butterflies only (no loads, stores, exchanges)")
    ABL(7, "exchanges only (no loads, stores, butterflies)")
    ABL(0, "full")
    return 0;
}
