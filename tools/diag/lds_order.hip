// lds_order.hip -- does ds_add_rtn_u32 hand out its results in lane order?
//
// When several lanes of ONE wave instruction add to the same LDS word, each gets a different "old" value; the ISA
// manual does not say in which order.  The block sort's counting pass (rcx_bwt.hpp) can use the atomic in place of its
// 8-ballot lane match only if the order is ascending lane order.  This probe runs many digit patterns through both
// and counts disagreements (0 on the MI355X it was run on: profiles/r02_lds_order.txt).
//   hipcc --offload-arch=gfx950 -O3 -o build/lds_order tools/diag/lds_order.hip && build/lds_order
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32;
typedef uint64_t u64;

__device__ __forceinline__ void match8(u32 d, u32& below, u32& total)
{
    u32 lo = ~0u, hi = ~0u;
#pragma unroll
    for (u32 b = 0; b < 8; ++b) {
        const u32 mine = (u32)((int32_t)(d << (31u - b)) >> 31);
        const u64 bal = __builtin_amdgcn_ballot_w64(mine != 0);
        lo &= ~((u32)bal ^ mine);
        hi &= ~((u32)(bal >> 32) ^ mine);
    }
    below = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
    total = (u32)__popc(lo) + (u32)__popc(hi);
}

__device__ __forceinline__ u32 mixbits(u32 x)
{
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

// 16 waves per workgroup, like the sort: every wave its own 128 words (digits d and d + 128 share a word: +1 / +65536)
__global__ __launch_bounds__(1024) void probe(u32 rounds, u32* bad, u32* seen_collisions)
{
    __shared__ u32 tab[16][128];
    __shared__ u32 ref[16][256];
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    for (u32 i = lane; i < 128; i += 64) tab[w][i] = 0;
    for (u32 i = lane; i < 256; i += 64) ref[w][i] = 0;
    u32 wrong = 0, coll = 0;
    for (u32 r = 0; r < rounds; ++r) {
        const u32 kind = (r + blockIdx.x) % 7u;
        const u32 h = mixbits(r * 0x9E3779B9u + blockIdx.x * 977u + w * 131u + lane * 0x85EBCA6Bu);
        u32 d;
        switch (kind) {
        case 0: d = h & 0xFFu; break;                      // all digits
        case 1: d = h & 3u; break;                         // four values
        case 2: d = 7u; break;                             // all lanes the same word and half
        case 3: d = (h & 1u) ? 200u : 72u; break;          // two values in ONE word, different halves
        case 4: d = (lane >> 2) + ((h >> 9) & 1u) * 128u; break; // neighbours collide
        case 5: d = (h % 5u) * 32u; break;                 // same bank, different words
        default: d = ((h >> 3) & 0x7Fu) | ((r & 1u) << 7); break;
        }
        // the reference: ballots + a plain counter
        u32 below, total;
        match8(d, below, total);
        const u32 before = ref[w][d];
        if (below + 1 == total) ref[w][d] = before + total;
        // the candidate: one atomic
        const u32 shift = (d >> 7) * 16u;
        const u32 old = __hip_atomic_fetch_add(&tab[w][d & 127u], 1u << shift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const u32 got = (old >> shift) & 0xFFFFu;
        wrong += got != ((before + below) & 0xFFFFu) ? 1u : 0u;
        coll += total > 1 ? 1u : 0u;
        if ((r & 1023u) == 1023u) { // keep the 16-bit halves from running over
            for (u32 i = lane; i < 128; i += 64) tab[w][i] = 0;
            for (u32 i = lane; i < 256; i += 64) ref[w][i] = 0;
        }
    }
    atomicAdd(bad, wrong);
    atomicAdd(seen_collisions, coll);
}

int main()
{
    u32 *bad, *coll;
    (void)hipMalloc(&bad, 4);
    (void)hipMalloc(&coll, 4);
    (void)hipMemset(bad, 0, 4);
    (void)hipMemset(coll, 0, 4);
    const u32 rounds = 1u << 14;
    hipLaunchKernelGGL(probe, dim3(512), dim3(1024), 0, 0, rounds, bad, coll);
    (void)hipDeviceSynchronize();
    u32 b = 0, c = 0;
    (void)hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&c, coll, 4, hipMemcpyDeviceToHost);
    printf("ds_add_rtn_u32 vs lane-ordered counts: %llu lane-instructions checked, %u (mod 2^32) of them shared their digit, %u disagreements\n",
           512ull * 1024ull * rounds, c, b);
    return b != 0;
}
