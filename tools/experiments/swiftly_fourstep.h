// SwiFTly on MI355X: both passes of a four-step column transform in ONE launch, the intermediate handed over in flight.
//
// col_transform (swiftly_abi.hip) runs a strided-axis transform of more than 1024 points as pass A (length n1, writes
// the scratch [batch][N][W]) and pass B (length n2, reads it back): for K2 of the 64k workload that is 1.21 GB written
// and 1.21 GB re-read per wave, a third of the pass's HBM traffic, long after the 256 MiB Infinity Cache has lost it.
// Here the two passes are workgroups of one grid.  The unit of hand-over is a CHUNK = one (batch item, 64-column tile):
// 2^l2 / SUB pass-A workgroups write its N x 64 intermediate (16.8 MB at N = 32768), 2^l1 pass-B workgroups read it.
// The grid is a sequence of slots; slot s holds the pass-A workgroups of chunk s and the pass-B workgroups of chunk
// s - LAG, so the intermediate of a chunk is read back a few hundred workgroups after it was written, while it still
// sits in the Infinity Cache.
//
// Hand-over (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"): a pass-A workgroup
// drains its stores, one lane issues an agent-scope release (the XCD's L2 is not coherent with the other seven) and
// bumps the chunk's arrival counter; a pass-B workgroup polls the counter with relaxed loads (one lane, s_sleep),
// issues ONE agent-scope acquire, then reads with plain loads.  Nothing reads a chunk's intermediate earlier in the
// launch, so no L2 holds a stale line of it.
//
// Forward progress: pass-B workgroups of chunk c have larger block ids than every pass-A workgroup of chunk c.  The
// hardware hands out the blocks of a 1-D grid in id order round-robin over the XCDs; the block with the smallest id
// not yet started can only be kept waiting by resident blocks with smaller ids, and those are either pass-A blocks
// (never wait) or pass-B blocks waiting for pass-A blocks with still smaller ids (started, by minimality) -- so it
// starts, by induction every block does.  HIP does not PROMISE that order, so the wait is bounded: after
// kFourStepTimeoutTicks of the 100 MHz wall clock a waiting workgroup raises *err (host-visible, sticky; every later
// ABI call on the handle fails) and goes on -- a wrong result that is reported, never a hang.
//
// MEASURED (r3, MI355X, K2 of the 64k workload, 25 waves): two launches 17.4 ms; this kernel 35.5 / 31.0 / 32.0 ms at
// LAG = 2 / 4 / 8.  With the release fence compiled out (wrong results, timing only) 17.3 ms, with neither fence nor
// wait 15.9 ms: the 9216 release fences per wave (each writes back its XCD's whole L2, which other workgroups keep
// dirtying) cost more than the Infinity Cache returns, and even a free hand-over would only be worth 1.5 ms per pass
// -- pass B is not HBM-bound once its reads hit the cache.  Kept as an opt-in (SWIFTLY_FOURSTEP_FUSED=1) with a parity
// test; the two-launch form stays the default.
#pragma once
#include "swiftly_colpass.h"

namespace swf {

// per-batch-item tables of the fused launch: at most 4 subgrids per facet and no gather-sum chunks (kernel arguments
// are limited to 4 KiB; two ColPassArgs + two of these = 1.7 KiB)
constexpr int kFsZB = 4;
using ColZS = ColZT<kFsZB, kColZF, 1>;

constexpr long long kFourStepTimeoutTicks = 200000000ll;  // 2 s of s_memrealtime (100 MHz)

struct FourStepSched {
    unsigned* counters;   // [chunks] arrivals of pass-A workgroups, zero at launch
    unsigned* err;        // host-visible word, set to 1 when a wait timed out
    int chunks;           // batch items x column tiles
    int col_tiles;
    int a_per, b_per;     // pass-A / pass-B workgroups per chunk
    int lag;              // slots between a chunk's pass A and its pass B
};

template <class GA, class GB, bool SNT>
__global__ __launch_bounds__(GB::NT, 4) void col_fourstep_kernel(
    const ColPassArgs A, const ColPassArgs B, const ColZS za, const ColZS zb, const FourStepSched S) {
    static_assert(GA::COLS == 64 && GB::COLS == 64, "64-column tiles");
    static_assert(GB::NT % GA::NT == 0, "a pass-A workgroup is a whole number of pass-A tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SUB = GB::NT / GA::NT;  // pass-A tiles (outer indices) per workgroup
    const int per = S.a_per + S.b_per;
    const int slot = blockIdx.x / per, r = blockIdx.x - slot * per;  // uniform
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (r < S.a_per) {
        const int c = slot;
        if (c >= S.chunks) return;
        const int z = c / S.col_tiles, bx = c - z * S.col_tiles;
        const int sub = wave / GA::T, w = wave - sub * GA::T;  // GA::T waves per 64-column tile
        col_pass_body<GA, 0, SNT, false>(A, A.in, A.out, A.ld_win, A.ld_win2, A.st_win, A.st_win2, A.st_rowmap, A.tw,
                                         A.tw_full, za, w, lane, bx, r * SUB + sub, z, smem + (size_t)sub * GA::LDS_BYTES);
        // publish: stores drained by every wave, one agent-scope release, then the arrival
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(S.counters + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        const int c = slot - S.lag;
        if (c < 0 || c >= S.chunks) return;
        const int z = c / S.col_tiles, bx = c - z * S.col_tiles;
        if (threadIdx.x == 0) {
            const unsigned want = (unsigned)S.a_per;
            if (__hip_atomic_load(S.counters + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                const long long t0 = wall_clock64();
                while (__hip_atomic_load(S.counters + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(16);
                    if (wall_clock64() - t0 > kFourStepTimeoutTicks) {
                        __hip_atomic_store(S.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        col_pass_body<GB, 1, SNT, false>(B, B.in, B.out, B.ld_win, B.ld_win2, B.st_win, B.st_win2, B.st_rowmap, B.tw,
                                         B.tw_full, zb, wave, lane, bx, r - S.a_per, z, smem);
    }
}

// -1: no fused instance for (l1, l2); else a hipError_t
int launch_col_fourstep(int l1, int l2, const ColPassArgs& a, const ColPassArgs& b, const ColZS& za, const ColZS& zb,
                        int nbatch, int lag, unsigned* counters, unsigned* err, hipStream_t s);
int init_col_fourstep();
bool col_fourstep_supported(int l1, int l2);

}  // namespace swf
