// rcx_bwt.hpp -- the reference's block sort (blksort.h) on gfx950: the Burrows-Wheeler transform of 32 KiB blocks
// that its harness puts in front of the entropy coders (test/main.cpp:961-986), and the inverse.
//
//   rcx_bwt_fwd_k   one workgroup of 16 waves per block: the 32768 rotations are sorted by prefix doubling inside
//                   the CU's LDS, the last column and the row of rotation 0 leave as blksort.h:511-518 writes them
//   rcx_bwt_tie_k   periodic blocks only (rotations that tie): one lane replays the reference's unstable sort on
//                   row classes for the row index it would store (rcx_bwt_tie.hpp)
//   rcx_bwt_inv_k   one workgroup per block: stable counting sort of the column (blksort.h:365-397), then the
//                   walk of blksort.h:663-667 cut into 1024 pieces that are walked at once
//
// What is sorted and how.  SA[k] = the rotation in row k, RK[i] = the first row of the group rotation i is in.  Start:
// rows ordered by two bytes (two stable 8-bit counting passes).  Round h = 2, 4, ...: the rows are in h-order, so the
// sequence SA[k] - h (k = 0, 1, ...) lists the rotations by their SECOND h bytes; sorted stably by RK[.] -- the group
// of their FIRST h bytes, a 15-bit key, two 8-bit passes -- it is the 2h-order (Manber-Myers).  Then the groups are
// split where neighbours differ in (RK[i], RK[i + h]).  All indices are mod 32768: these are rotations, not suffixes.
// It ends when every row is alone in its group -- or when a round splits nothing, which happens exactly when the
// block is periodic (if h bytes and 2h bytes give the same classes, so does any depth): the number of groups then is
// the period, the column is right whatever the order inside the ties, and the row index is rcx_bwt_tie_k's.
//
// A block that is mostly runs of one byte starts from deeper keys than two bytes (byte, how its run ends, the run's
// length, the byte behind the run: rcx_bwt_fwd_k), which order the rotations inside a run at once; the rounds work on
// ranks of unequal depth as long as they agree with the true order and are at least h bytes deep at the round with shift h.
//
// Rounds get cheaper: a rotation that is alone in its group has its final row (RK carries a flag for it), and once the
// rotations that are not fit a list in LDS (11264 entries; text gets there after two or three rounds) a round only
// touches those: the SA[k] - h that are still open are collected in row order, that list is sorted by group with the
// same two counting passes -- a third, a tenth or a thirtieth as long -- and lands in the open rows, which it fills exactly.
//
// A stable counting pass of 32768 keys by 1024 lanes: wave w owns keys [2048 w, 2048 w + 2048), 64 at a time in index
// order; per wave and digit a running count in LDS gives every key its rank among the wave's earlier keys with the
// same digit -- within the 64 of a batch either by 8 ballots (lanes with the same digit find each other, the lowest
// ones first) or by one LDS atomic per key where the device serves a wave's lanes in lane order (rcx_bwt_pass);
// after an exclusive scan of the 16 x 256 counts in (digit, wave) order every key knows its place.  The keys stay in
// registers between the count and the scatter, so one array is sorted in place.
//
// LDS (forward): SA 64 KiB | RK 64 KiB | counts 8 KiB | 256 B | list 22 KiB.  RK's space first holds the block itself
// (the two-byte keys are bytes of it) and at the end again (the column is gathered from it); SA's space ends as the
// staging buffer the 32770 output bytes leave from in aligned 16-byte pieces.  One workgroup per CU.
#pragma once
#include <hip/hip_runtime.h>

#include "rcx_bwt_tie.hpp"
#include "rcx_lane.hpp"

#define RCX_BWT_BLOCK 32768u   /* blksort.h:82 */
#define RCX_BWT_MASK 32767u
#define RCX_BWT_ENCODED 32770u /* blksort.h:85 */
#define RCX_BWT_THREADS 1024u
#define RCX_BWT_WAVES 16u

#define RCX_BWT_FWD_SA 0u
#define RCX_BWT_FWD_RK 65536u
#define RCX_BWT_FWD_CNT 131072u
#define RCX_BWT_FWD_MISC (131072u + 8192u)
#define RCX_BWT_FWD_LIST (RCX_BWT_FWD_MISC + 256u)
#define RCX_BWT_FINAL 0x8000u  /* in RK: the rotation is alone in its group, its row is final */
#define RCX_BWT_LIST_BIG 11u   /* a list of up to 16 waves x 11 x 64 = 11264 rotations ... */
#define RCX_BWT_LIST_SMALL 3u  /* ... or of up to 3072 */
#define RCX_BWT_FWD_LDS (RCX_BWT_FWD_LIST + 2u * 1024u * RCX_BWT_LIST_BIG)

#define RCX_BWT_INV_NEXT 0u
#define RCX_BWT_INV_ENC 65536u                 /* 32770 bytes + up to 15 of alignment */
#define RCX_BWT_INV_OUT (65536u + 32832u)      /* 32768 bytes + up to 15 */
#define RCX_BWT_INV_CNT (RCX_BWT_INV_OUT + 32800u)
#define RCX_BWT_INV_LINK (RCX_BWT_INV_CNT + 8192u) /* u32[1024] */
#define RCX_BWT_INV_MISC (RCX_BWT_INV_LINK + 4096u)
/* behind the counting pass its 8 KiB of counts hold what the walks note: u32 mark_of[1024] | u16 mark_row[1024] | u16 lens[1024] */
#define RCX_BWT_INV_LDS (RCX_BWT_INV_MISC + 256u)

#define RCX_BWT_TIE_ROWS 0u
#define RCX_BWT_TIE_WORD 65536u
#define RCX_BWT_TIE_RANK (65536u + 16384u)              /* u16 per class: its rotation's place (rank_classes) */
#define RCX_BWT_TIE_HIST (RCX_BWT_TIE_RANK + 32768u)     /* 256 x 64 u16 counts of a counting pass / the next ranks */
#define RCX_BWT_TIE_STACK (RCX_BWT_TIE_HIST + 32768u)
#define RCX_BWT_TIE_SUMS (RCX_BWT_TIE_STACK + RCX_TIE_STACK * 16u)
#define RCX_BWT_TIE_LDS (RCX_BWT_TIE_SUMS + 128u + 64u)

// threadIdx.x as a value the compiler treats as new each time: what is computed from it (row numbers, list positions,
// LDS addresses: one multiply-add each) is then computed where it is used.  Otherwise all of it is hoisted out of the
// loop over blocks, does not fit the 128 registers a lane of a 16-wave workgroup has, and goes through scratch memory
// (measured: 27 GiB of HBM traffic for 1 GiB of blocks).
__device__ __forceinline__ u32 rcx_bwt_tid()
{
    u32 t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
// A value every lane holds alike (read from LDS, so in a vector register): moved to a scalar register, where it costs
// no vector register and branches on it are scalar branches.
__device__ __forceinline__ u32 rcx_bwt_same(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }

// ---------------------------------------------------------------------------
// global <-> LDS copies of one block.  The LDS image starts at lds + (address & 15), so that both sides of the 16-byte
// pieces in the middle are aligned whatever the caller's pointer is; a few threads move the ragged ends bytewise.
// ---------------------------------------------------------------------------
__device__ __forceinline__ u32 rcx_bwt_stage_in(u8* lds, const u8* g, u32 bytes)
{
    const u32 tid = rcx_bwt_tid();
    const u32 shift = (u32)(reinterpret_cast<uintptr_t>(g) & 15u);
    const u32 head = (16u - shift) & 15u;
    const u32 pieces = (bytes - head) >> 4, tail = (bytes - head) & 15u;
    if (tid < head) lds[shift + tid] = g[tid];
    for (u32 i = tid; i < pieces; i += RCX_BWT_THREADS)
        *reinterpret_cast<U4*>(lds + shift + head + 16u * i) = *reinterpret_cast<const U4*>(g + head + 16u * i);
    if (tid < tail) lds[shift + head + 16u * pieces + tid] = g[head + 16u * pieces + tid];
    return shift;
}
// the image must have been built at lds + (g & 15)
__device__ __forceinline__ void rcx_bwt_stage_out(u8* g, const u8* lds, u32 bytes)
{
    const u32 tid = rcx_bwt_tid();
    const u32 shift = (u32)(reinterpret_cast<uintptr_t>(g) & 15u);
    const u32 head = (16u - shift) & 15u;
    const u32 pieces = (bytes - head) >> 4, tail = (bytes - head) & 15u;
    if (tid < head) g[tid] = lds[shift + tid];
    for (u32 i = tid; i < pieces; i += RCX_BWT_THREADS)
        *reinterpret_cast<U4*>(g + head + 16u * i) = *reinterpret_cast<const U4*>(lds + shift + head + 16u * i);
    if (tid < tail) g[head + 16u * pieces + tid] = lds[shift + head + 16u * pieces + tid];
}

// Lanes of the wave holding the same 8-bit digit as this one: how many of them are below this lane, and how many
// there are.  Per bit: the lanes that have it set (one compare = one ballot), and every lane keeps those that agree
// with it (xnor with its own bit spread over the word).
__device__ __forceinline__ void rcx_bwt_match8(u32 d, u32& below, u32& total)
{
    u32 lo = ~0u, hi = ~0u;
#pragma unroll
    for (u32 b = 0; b < 8; ++b) {
        const u32 mine = (u32)((s32)(d << (31u - b)) >> 31); // bit b of d on every bit
        u64 bal;
        // (written out: as a ballot builtin the compiler tests a second, shifted copy of d; volatile: the result depends
        // on which lanes are active, which the compiler does not see in the operands)
        asm volatile("v_cmp_ne_u32_e64 %0, 0, %1" : "=s"(bal) : "v"(mine));
        // keep & ~(ballot ^ mine) in one v_bitop3_b32 each (table 0x90 = a & ~(b ^ c))
        lo = __builtin_amdgcn_bitop3_b32(lo, (u32)bal, mine, 0x90);
        hi = __builtin_amdgcn_bitop3_b32(hi, (u32)(bal >> 32), mine, 0x90);
    }
    below = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
    total = (u32)__popc(lo) + (u32)__popc(hi);
}

// Inclusive scans over the 64 lanes with DPP moves (no LDS, no index registers): each lane first gathers the three
// lanes before it in its row of 16, then whole quarters and halves of the row, then the totals of the rows before.
// A lane the move does not reach sees 0, the identity of both operations.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ u32 rcx_bwt_dpp0(u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ u32 rcx_bwt_wave_incl_sum(u32 v)
{
    v += rcx_bwt_dpp0<0x111, 0xF, 0xF>(v) + rcx_bwt_dpp0<0x112, 0xF, 0xF>(v) + rcx_bwt_dpp0<0x113, 0xF, 0xF>(v); // row_shr:1..3
    v += rcx_bwt_dpp0<0x114, 0xF, 0xE>(v); // row_shr:4, lanes 4..15 of a row
    v += rcx_bwt_dpp0<0x118, 0xF, 0xC>(v); // row_shr:8, lanes 8..15
    v += rcx_bwt_dpp0<0x142, 0xA, 0xF>(v); // row_bcast:15 into rows 1 and 3
    v += rcx_bwt_dpp0<0x143, 0xC, 0xF>(v); // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ u32 rcx_bwt_max(u32 a, u32 b) { return a > b ? a : b; }
__device__ __forceinline__ u32 rcx_bwt_wave_incl_max(u32 v)
{
    v = rcx_bwt_max(rcx_bwt_max(v, rcx_bwt_dpp0<0x111, 0xF, 0xF>(v)), rcx_bwt_max(rcx_bwt_dpp0<0x112, 0xF, 0xF>(v), rcx_bwt_dpp0<0x113, 0xF, 0xF>(v)));
    v = rcx_bwt_max(v, rcx_bwt_dpp0<0x114, 0xF, 0xE>(v));
    v = rcx_bwt_max(v, rcx_bwt_dpp0<0x118, 0xF, 0xC>(v));
    v = rcx_bwt_max(v, rcx_bwt_dpp0<0x142, 0xA, 0xF>(v));
    v = rcx_bwt_max(v, rcx_bwt_dpp0<0x143, 0xC, 0xF>(v));
    return v;
}
// the value of the lane before (0 for lane 0): wave_shr:1
__device__ __forceinline__ u32 rcx_bwt_wave_prev(u32 v) { return rcx_bwt_dpp0<0x138, 0xF, 0xF>(v); }

// exclusive prefix sum over the workgroup (all 1024 threads call it); misc: 16 dwords, which the caller must not write
// again before its next barrier (both callers have one right behind)
__device__ __forceinline__ u32 rcx_bwt_block_excl(u32 v, u32* misc)
{
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const u32 incl = rcx_bwt_wave_incl_sum(v);
    if (lane == 63) misc[w] = incl;
    __syncthreads();
    u32 base = 0;
#pragma unroll
    for (u32 i = 0; i < RCX_BWT_WAVES; ++i) {
        const u32 t = misc[i];
        base += i < w ? t : 0u;
    }
    return base + incl - v;
}

// One stable counting pass over arr[0 .. 1024 ITERS): the elements end up ordered by their digit, equal digits in the
// order they were in.  ELEM(x) turns what is read from arr into the element that is stored back, DIGIT(e) is its
// digit.  Both run on whole batches of ITERS per lane before anything is counted, so their LDS reads overlap.
//
// Counting a batch of 64 keys has two forms.  The lane match above: ~60 vector instructions whatever the digits are.
// Or ONE ds_add_rtn_u32 on the digit's count (two 16-bit counts to a dword): the value a lane gets back is the count
// of earlier keys INCLUDING the lanes below it in this batch -- if the LDS hands the lanes of one instruction their
// results in ascending lane order.  The ISA manual does not promise that: rcx_bwt_lds_order_k checks it on the
// device (tools/diag/lds_order.hip is the long form: 2^33 lane-instructions without a disagreement on the MI355X),
// and with ATOMIC = false only the match is compiled.  The atomic is fast when the 64 digits differ and slow when they
// pile up (lanes adding to one word are served one after the other, and the LDS is the whole CU's): so every eighth
// batch is matched, and if no digit had more than RCX_BWT_PILE lanes there the next seven use the atomic.
// Measured, 1 GiB forward, match only / atomic only / this: uniform 13.2 / 10.5 / 9.5 ms, Zipf 13.6 / 13.1 / 11.0,
// Canterbury 30.7 / 38.9 / 28.4, runs 77.4 / 123.7 / 78.5; thresholds of 24, 32 and 48 lanes measure the same, 6 and 12
// worse (text: 30.2, 29.0): it takes most of the wave on one word to make the atomic the slower of the two.
#if !defined(RCX_BWT_PILE)
#define RCX_BWT_PILE 32u
#endif
// (the counts are read and written as u16 by the match and added to as dwords by the atomic: both through types that
// may alias anything, so the compiler keeps the accesses in program order)
typedef uint16_t __attribute__((may_alias)) RcxCount16;
typedef u32 __attribute__((may_alias)) RcxCount32;
template <u32 ITERS, bool ATOMIC, class Elem, class Digit>
__device__ __forceinline__ void rcx_bwt_pass(uint16_t* arr, uint16_t* cnt, u32* misc, Elem elem, Digit digit)
{
    const u32 tid = rcx_bwt_tid(), lane = tid & 63u, w = tid >> 6;
    RcxCount16* mine = reinterpret_cast<RcxCount16*>(cnt + 256u * w);
    RcxCount32* mine32 = reinterpret_cast<RcxCount32*>(cnt + 256u * w);
    mine32[lane] = 0;
    mine32[lane + 64] = 0;
    u32 held[ITERS]; // element | rank among the wave's earlier keys with the same digit << 16
    u32 digits[(ITERS + 3) / 4];
    const uint16_t* in = arr + 64u * ITERS * w + lane;
#pragma unroll
    for (u32 it = 0; it < ITERS; ++it) held[it] = in[64u * it];
#pragma unroll
    for (u32 it = 0; it < ITERS; ++it) held[it] = elem(held[it]);
    {
        u32 dg[ITERS];
#pragma unroll
        for (u32 it = 0; it < ITERS; ++it) dg[it] = digit(held[it]);
#pragma unroll
        for (u32 q = 0; q < (ITERS + 3) / 4; ++q) digits[q] = 0;
#pragma unroll
        for (u32 it = 0; it < ITERS; ++it) digits[it >> 2] |= dg[it] << (8u * (it & 3u));
    }
#define RCX_BWT_COUNT_MATCHED(IT, PILED)                                                                                 \
    {                                                                                                                  \
        const u32 d_ = (digits[(IT) >> 2] >> (8u * ((IT) & 3u))) & 0xFFu;                                               \
        u32 below_, total_;                                                                                            \
        rcx_bwt_match8(d_, below_, total_);                                                                            \
        const u32 old_ = mine[d_];                                                                                     \
        /* the highest of the peers writes -- or, where the match only runs on piled-up batches, all of them (the same) */ \
        if (ATOMIC || below_ + 1 == total_) mine[d_] = (uint16_t)(old_ + total_);                                      \
        held[IT] |= (old_ + below_) << 16;                                                                             \
        if (ATOMIC) PILED = __builtin_amdgcn_ballot_w64(total_ > RCX_BWT_PILE) != 0;                                   \
    }
#define RCX_BWT_COUNT_ATOMIC(IT)                                                                                         \
    {                                                                                                                  \
        const u32 d_ = (digits[(IT) >> 2] >> (8u * ((IT) & 3u))) & 0xFFu, half_ = 16u * (d_ & 1u);                      \
        const u32 old_ = __hip_atomic_fetch_add(&mine32[d_ >> 1], 1u << half_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); \
        held[IT] |= (old_ >> half_) << 16; /* (<< 16 drops the neighbour digit's count when this one is the lower half) */ \
    }
#pragma unroll
    for (u32 first = 0; first < ITERS; first += 8) {
        bool piled = true;
        RCX_BWT_COUNT_MATCHED(first, piled);
        if (!ATOMIC || piled) {
#pragma unroll
            for (u32 it = first + 1; it < ITERS && it < first + 8; ++it) {
                bool unused = true;
                RCX_BWT_COUNT_MATCHED(it, unused);
                (void)unused;
            }
        } else {
#pragma unroll
            for (u32 it = first + 1; it < ITERS && it < first + 8; ++it) RCX_BWT_COUNT_ATOMIC(it);
        }
    }
#undef RCX_BWT_COUNT_MATCHED
#undef RCX_BWT_COUNT_ATOMIC
    __syncthreads();
    { // 4096 counts -> their exclusive prefix in (digit, wave) order
        const u32 d = tid >> 2, w0 = 4u * (tid & 3u);
        const u32 v0 = cnt[256u * (w0 + 0) + d], v1 = cnt[256u * (w0 + 1) + d], v2 = cnt[256u * (w0 + 2) + d], v3 = cnt[256u * (w0 + 3) + d];
        const u32 base = rcx_bwt_block_excl(v0 + v1 + v2 + v3, misc);
        cnt[256u * (w0 + 0) + d] = (uint16_t)base;
        cnt[256u * (w0 + 1) + d] = (uint16_t)(base + v0);
        cnt[256u * (w0 + 2) + d] = (uint16_t)(base + v0 + v1);
        cnt[256u * (w0 + 3) + d] = (uint16_t)(base + v0 + v1 + v2);
    }
    __syncthreads();
#pragma unroll
    for (u32 q = 0; q < (ITERS + 3) / 4; ++q) asm volatile("" : "+v"(digits[q])); // (addresses are formed again here, not kept from above)
    {
        u32 base[ITERS];
#pragma unroll
        for (u32 it = 0; it < ITERS; ++it) base[it] = mine[(digits[it >> 2] >> (8u * (it & 3u))) & 0xFFu];
#pragma unroll
        for (u32 it = 0; it < ITERS; ++it) arr[base[it] + (held[it] >> 16)] = (uint16_t)held[it];
    }
    __syncthreads();
}

// 32 consecutive rows of SA into registers
__device__ __forceinline__ void rcx_bwt_rows32(const uint16_t* sa, u32 k0, u32 (&s)[32])
{
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
        const U4 v = reinterpret_cast<const U4*>(sa + k0)[q];
        s[8 * q + 0] = v.x & 0xFFFFu;
        s[8 * q + 1] = v.x >> 16;
        s[8 * q + 2] = v.y & 0xFFFFu;
        s[8 * q + 3] = v.y >> 16;
        s[8 * q + 4] = v.z & 0xFFFFu;
        s[8 * q + 5] = v.z >> 16;
        s[8 * q + 6] = v.w & 0xFFFFu;
        s[8 * q + 7] = v.w >> 16;
    }
}

// sum over the workgroup; misc: 16 dwords (not to be written again before the caller's next barrier)
__device__ __forceinline__ u32 rcx_bwt_block_sum(u32 v, u32* misc)
{
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    v = rcx_bwt_wave_incl_sum(v);
    if (lane == 63) misc[w] = v;
    __syncthreads();
    u32 all = 0;
#pragma unroll
    for (u32 i = 0; i < RCX_BWT_WAVES; ++i) all += misc[i];
    return all;
}

// New groups: row k starts one if it did already (a sort moves rotations inside their groups only, so the rows where
// groups start stay: `starts` / `behind` are this thread's 32 bits of that and the bit of the row behind them, kept in
// registers from one round to the next) or if KEY(SA[k]) differs from KEY(SA[k - 1]).  RK[SA[k]] = the first row of k's
// group, with RCX_BWT_FINAL if the group is that one row.  Thread t looks at rows [32 t, 32 t + 32).  Returns the
// number of groups, `open` = the rotations that are not final.  misc: 48 dwords.
template <class Key>
__device__ __forceinline__ u32 rcx_bwt_rerank(const uint16_t* sa, uint16_t* rk, u32* misc, Key key, u32& open, u32& starts, u32& behind)
{
    const u32 tid = rcx_bwt_tid(), lane = tid & 63u, w = tid >> 6, k0 = 32u * tid;
    u32 s[32];
    rcx_bwt_rows32(sa, k0, s);
    u32 prev = key((u32)sa[(k0 + RCX_BWT_MASK) & RCX_BWT_MASK]);
    u32 bits = starts;
#pragma unroll
    for (u32 i = 0; i < 32; ++i) {
        const u32 kk = key(s[i]);
        bits |= (kk != prev ? 1u : 0u) << i;
        prev = kk;
    }
    if (tid == 0) bits |= 1u;
    // does the row behind this thread's start a group?  (behind the last row: yes)
    behind |= (tid == RCX_BWT_THREADS - 1u || key((u32)sa[(k0 + 32u) & RCX_BWT_MASK]) != prev) ? 1u : 0u;
    starts = bits;
    const u32 alone = bits & ((bits >> 1) | (behind << 31));
    // the last group start at or before each row: inside the thread from `bits`, before it a running maximum
    const u32 last = bits ? k0 + 31u - (u32)__clz(bits) : 0u;
    const u32 run = rcx_bwt_wave_incl_max(last);
    // (two 16-bit sums in one word: at most 2048 each per wave)
    const u32 sum = rcx_bwt_wave_incl_sum((u32)__popc(bits) | ((32u - (u32)__popc(alone)) << 16));
    const u32 before = rcx_bwt_wave_prev(run);
    if (lane == 63) {
        misc[w] = run;
        misc[16 + w] = sum & 0xFFFFu;
        misc[32 + w] = sum >> 16;
    }
    __syncthreads(); // (every key has been read: RK may be rewritten)
    u32 carry = before, groups = 0, left = 0;
#pragma unroll
    for (u32 i = 0; i < RCX_BWT_WAVES; ++i) {
        const u32 m = misc[i];
        if (i < w) carry = carry > m ? carry : m;
        groups += misc[16 + i];
        left += misc[32 + i];
    }
#pragma unroll
    for (u32 i = 0; i < 32; ++i) {
        const u32 m = bits & ((2u << i) - 1u);
        rk[s[i]] = (uint16_t)((m ? k0 + 31u - (u32)__clz(m) : carry) | (((alone >> i) & 1u) ? RCX_BWT_FINAL : 0u));
    }
    __syncthreads();
    open = rcx_bwt_same(left);
    return rcx_bwt_same(groups);
}

// The rotations SA[k] - h that are not final, in the order of k, into lst[0 .. count); lst is padded with 0xFFFF up to
// `padded`.  Returns the count.  misc: 24 dwords.
__device__ __forceinline__ u32 rcx_bwt_collect(const uint16_t* sa, const uint16_t* rk, uint16_t* lst, u32* misc, u32 h, u32 padded)
{
    const u32 tid = rcx_bwt_tid(), k0 = 32u * tid;
    u32 s[32];
    rcx_bwt_rows32(sa, k0, s);
    u32 pick = 0;
#pragma unroll
    for (u32 i = 0; i < 32; ++i) {
        s[i] = (s[i] - h) & RCX_BWT_MASK;
        pick |= ((u32)rk[s[i]] < RCX_BWT_FINAL ? 1u : 0u) << i;
    }
    const u32 mine = (u32)__popc(pick);
    const u32 at = rcx_bwt_block_excl(mine, misc);
    if (tid == RCX_BWT_THREADS - 1u) misc[20] = at + mine;
#pragma unroll
    for (u32 i = 0; i < 32; ++i)
        if ((pick >> i) & 1u) lst[at + (u32)__popc(pick & ((1u << i) - 1u))] = (uint16_t)s[i];
    __syncthreads();
    const u32 count = rcx_bwt_same(misc[20]);
    for (u32 p = count + tid; p < padded; p += RCX_BWT_THREADS) lst[p] = 0xFFFFu;
    __syncthreads();
    return count;
}

// lst[0 .. count) holds the open rotations ordered by group, inside a group by their second h bytes: they go to the
// rows of their groups in that order, and the groups are split where neighbours differ in RK[. + h].  Thread t handles
// list entries [L t, L t + L).  splits = how many groups were added, returns the rotations still open.  misc: 16 dwords.
template <u32 L>
__device__ __forceinline__ u32 rcx_bwt_place(uint16_t* sa, uint16_t* rk, const uint16_t* lst, u32* misc, u32 h, u32 count, u32& splits)
{
    const u32 tid = rcx_bwt_tid(), q0 = L * tid;
    u32 j[L + 2], g[L + 2], k2[L + 2]; // entries q0 - 1 .. q0 + L
#pragma unroll
    for (u32 i = 0; i < L + 2; ++i) {
        const u32 q = q0 + i - 1u;
        const bool there = q < count; // (q0 - 1 wraps for the first thread)
        j[i] = there ? (u32)lst[q] : 0u;
    }
#pragma unroll
    for (u32 i = 0; i < L + 2; ++i) {
        const u32 q = q0 + i - 1u;
        const bool there = q < count;
        g[i] = there ? (u32)rk[j[i]] & RCX_BWT_MASK : 0xFFFFFFFFu;
        k2[i] = there ? (u32)rk[(j[i] + h) & RCX_BWT_MASK] & RCX_BWT_MASK : 0xFFFFFFFFu;
    }
    // bit i - 1: entry q0 + i - 1 ... starts an old group / starts a new group (entries past the list count as starts)
    u32 olds = 0, news = 0;
#pragma unroll
    for (u32 i = 1; i < L + 2; ++i) {
        const u32 q = q0 + i - 1u;
        const u32 o = (q == 0 || q >= count || g[i] != g[i - 1]) ? 1u : 0u;
        const u32 n = (o || k2[i] != k2[i - 1]) ? 1u : 0u;
        olds |= o << (i - 1);
        news |= n << (i - 1);
    }
    const u32 own = (1u << L) - 1u; // this thread's entries
    u32 in_list = 0;
#pragma unroll
    for (u32 i = 0; i < L; ++i) in_list |= (q0 + i < count ? 1u : 0u) << i;
    // index + 1 of the last old / new start at or before each entry: inside the thread from the bits, before it a running maximum
    const u32 o_own = olds & own & in_list, n_own = news & own & in_list;
    const u32 alone = news & (news >> 1) & own & in_list;
    // two running maxima and two sums across the workgroup in one exchange (misc: 48 dwords, one pair of barriers)
    u32 o_before, n_before, sums;
    {
        const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
        const u32 o_run = rcx_bwt_wave_incl_max(o_own ? q0 + 32u - (u32)__clz(o_own) : 0u);
        const u32 n_run = rcx_bwt_wave_incl_max(n_own ? q0 + 32u - (u32)__clz(n_own) : 0u);
        const u32 s_run = rcx_bwt_wave_incl_sum(((u32)__popc(n_own) - (u32)__popc(o_own)) | (((u32)__popc(in_list) - (u32)__popc(alone)) << 16));
        o_before = rcx_bwt_wave_prev(o_run);
        n_before = rcx_bwt_wave_prev(n_run);
        if (lane == 63) {
            misc[w] = o_run;
            misc[16 + w] = n_run;
            misc[32 + w] = s_run;
        }
        __syncthreads(); // (every RK has been read)
        sums = 0;
#pragma unroll
        for (u32 i = 0; i < RCX_BWT_WAVES; ++i) {
            const u32 o = misc[i], n = misc[16 + i];
            if (i < w) {
                o_before = o_before > o ? o_before : o;
                n_before = n_before > n ? n_before : n;
            }
            sums += misc[32 + i];
        }
        // (misc is next written behind the barrier at the end of this function)
    }
#pragma unroll
    for (u32 i = 0; i < L; ++i) {
        if ((in_list >> i) & 1u) {
            const u32 q = q0 + i;
            const u32 om = o_own & ((2u << i) - 1u), nm = n_own & ((2u << i) - 1u);
            const u32 first_old = om ? q0 + 31u - (u32)__clz(om) : o_before - 1u;
            const u32 first_new = nm ? q0 + 31u - (u32)__clz(nm) : n_before - 1u;
            const u32 row = g[i + 1] + (q - first_old);
            sa[row] = (uint16_t)j[i + 1];
            rk[j[i + 1]] = (uint16_t)((row - (q - first_new)) | (((alone >> i) & 1u) ? RCX_BWT_FINAL : 0u));
        }
    }
    __syncthreads();
    splits = rcx_bwt_same(sums & 0xFFFFu);
    return rcx_bwt_same(sums >> 16);
}

// Does this device's LDS hand the lanes of one ds_add_rtn_u32 their results in ascending lane order?  16 waves run digit
// patterns of several kinds (all values, few values, one value, two digits of one dword, neighbours colliding, one
// bank) through the ballot match and through the atomic, as the counting pass would; *bad counts the disagreements.
__global__ __launch_bounds__(1024) void rcx_bwt_lds_order_k(u32 rounds, u32* bad)
{
    __shared__ u32 tab[16][128];
    __shared__ uint16_t ref[16][256];
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    u32 wrong = 0;
    for (u32 r = 0; r < rounds; ++r) {
        if ((r & 511u) == 0) { // (the 16-bit counts must not run over)
            for (u32 i = lane; i < 128; i += 64) tab[w][i] = 0;
            for (u32 i = lane; i < 256; i += 64) ref[w][i] = 0;
        }
        u32 h = r * 0x9E3779B9u + w * 131u + lane * 0x85EBCA6Bu + blockIdx.x * 977u;
        h ^= h >> 16;
        h *= 0x7FEB352Du;
        h ^= h >> 15;
        h *= 0x846CA68Bu;
        h ^= h >> 16;
        u32 d;
        switch (r % 7u) {
        case 0: d = h & 0xFFu; break;
        case 1: d = h & 3u; break;
        case 2: d = 7u; break;
        case 3: d = (h & 1u) ? 201u : 200u; break;
        case 4: d = (lane >> 2) + ((h >> 9) & 1u) * 128u; break;
        case 5: d = (h % 4u) * 64u; break;
        default: d = ((h >> 3) & 0x7Fu) | ((r & 1u) << 7); break;
        }
        u32 below, total;
        rcx_bwt_match8(d, below, total);
        const u32 before = ref[w][d];
        if (below + 1 == total) ref[w][d] = (uint16_t)(before + total);
        const u32 half = 16u * (d & 1u);
        const u32 old = __hip_atomic_fetch_add(&tab[w][d >> 1], 1u << half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        wrong += ((old >> half) & 0xFFFFu) != before + below ? 1u : 0u;
    }
    if (wrong) atomicAdd(bad, wrong);
}

// The grid is one workgroup per CU; each takes the next block off a counter until none is left (a workgroup's 158 KiB
// of LDS would otherwise be handed out again for every block: measured 9 us a block).
// ties: [0] = count of periodic blocks with a period above 1, [1] = the forward kernel's block counter, [2] = the
// inverse kernel's, [3] unused, then the (block, period) pairs
#define RCX_BWT_TIES_HEAD 4u
#if !defined(RCX_BWT_RUNNY)
#define RCX_BWT_RUNNY (RCX_BWT_BLOCK / 8) /* fewer places than this where a byte differs from the next: the block starts from run keys */
/* (measured on the Canterbury GiB: 27.2 ms with 16384, 26.9 with 4096, 27.0 with 1024; the runs workload takes the run keys with any of them) */
#endif
#if defined(RCX_BWT_STAMP) /* diagnostic build only (tools/diag/stamp_bwt.py): cycles per phase, summed over blocks by thread 0 of each workgroup */
static __device__ unsigned long long rcx_bwt_stamp_out[16];
#define RCX_BWT_PHASE(i)                                                         \
    {                                                                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();            \
        stamp_[i] += now_ - mark_;                                               \
        mark_ = now_;                                                            \
    }
#else
#define RCX_BWT_PHASE(i)
#endif
template <bool ATOMIC>
__global__ __launch_bounds__(1024) void rcx_bwt_fwd_k(const u8* __restrict__ src, u64 nblocks, u8* __restrict__ dst, u32* __restrict__ ties, u32* status)
{
    extern __shared__ __attribute__((aligned(16))) u8 rcx_bwt_lds[];
    u8* lds = rcx_bwt_lds;
    uint16_t* sa = reinterpret_cast<uint16_t*>(lds + RCX_BWT_FWD_SA);
    uint16_t* rk = reinterpret_cast<uint16_t*>(lds + RCX_BWT_FWD_RK);
    uint16_t* cnt = reinterpret_cast<uint16_t*>(lds + RCX_BWT_FWD_CNT);
    u32* misc = reinterpret_cast<u32*>(lds + RCX_BWT_FWD_MISC);
    uint16_t* lst = reinterpret_cast<uint16_t*>(lds + RCX_BWT_FWD_LIST);
    const u32 tid = threadIdx.x;
#if defined(RCX_BWT_STAMP)
    unsigned long long stamp_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, mark_ = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        RCX_BWT_PHASE(11) // between blocks
        if (tid == 0) misc[49] = atomicAdd(&ties[1], 1u);
        __syncthreads();
        const u64 b = rcx_bwt_same(misc[49]);
        if (b >= nblocks) break;
        const u8* in = src + b * RCX_BWT_BLOCK;
        u8* out = dst + b * RCX_BWT_ENCODED;
        u8* text = lds + RCX_BWT_FWD_RK;
        const u32 shift = rcx_bwt_stage_in(text, in, RCX_BWT_BLOCK);
        __syncthreads();
        RCX_BWT_PHASE(0) // block in
        u32 open = 0, groups = RCX_BWT_BLOCK, starts = 0, behind = 0;
        // Where does a byte differ from the next one?  If in fewer than an eighth of the places, the block is mostly runs of
        // one byte, which prefix doubling alone resolves one doubling per round (a run of 4000: 12 rounds): such a block
        // starts from deeper keys instead -- (byte, how the run ends, its length, the byte behind it), rcx_bwt_run_key --
        // which order the rotations inside a run at once.  Ranks of unequal depth are fine for the rounds as long as
        // they agree with the true order and every one is at least h bytes deep when the round with shift h starts.
        u32* change = reinterpret_cast<u32*>(lst);                    // 1024 dwords: bit i of dword t = byte 32 t + i differs from the next
        uint16_t* next_change = reinterpret_cast<uint16_t*>(lst) + 2048; // 1025 entries: first such place at or behind 32 t (+ 32768 when it wraps)
        bool by_runs = false;
#if !defined(RCX_BWT_NO_RUN_START) && !defined(RCX_BWT_PROBE_NO_SORT) && !defined(RCX_BWT_PROBE_PASSES)
        {
            const u32 k0 = 32u * rcx_bwt_tid();
            u32 differs = 0;
            if (shift == 0) { // (the usual case: the block sits aligned in LDS and is read eight bytes a step)
                const U4 lo = *reinterpret_cast<const U4*>(text + k0), hi = *reinterpret_cast<const U4*>(text + k0 + 16);
                const u32 w[9] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w, (u32)text[(k0 + 32u) & RCX_BWT_MASK]};
#pragma unroll
                for (u32 q = 0; q < 8; ++q) {
                    const u32 x = w[q] ^ ((w[q] >> 8) | (w[q + 1] << 24)); // byte j of x: byte j against byte j + 1
                    differs |= (((x & 0xFFu) ? 1u : 0u) | ((x & 0xFF00u) ? 2u : 0u) | ((x & 0xFF0000u) ? 4u : 0u) | ((x >> 24) ? 8u : 0u)) << (4u * q);
                }
            } else {
                u32 before = text[shift + k0];
#pragma unroll
                for (u32 i = 0; i < 32; ++i) {
                    const u32 here = text[shift + ((k0 + i + 1u) & RCX_BWT_MASK)];
                    differs |= (here != before ? 1u : 0u) << i;
                    before = here;
                }
            }
            change[tid] = differs;
            next_change[tid] = (uint16_t)(differs ? k0 + (u32)__builtin_ctz(differs) : 0xFFFFu);
            const u32 changes = rcx_bwt_same(rcx_bwt_block_sum((u32)__popc(differs), misc));
            by_runs = changes < RCX_BWT_RUNNY;
        }
#endif
        if (by_runs) {
            // the first change at or behind every 32nd place: a minimum over the places behind, by doubling
            u32 mine = next_change[tid];
#pragma nounroll
            for (u32 step = 1; step < RCX_BWT_THREADS; step <<= 1) {
                const u32 other = tid + step < RCX_BWT_THREADS ? (u32)next_change[tid + step] : 0xFFFFu;
                __syncthreads();
                mine = mine < other ? mine : other;
                next_change[tid] = (uint16_t)mine;
                __syncthreads();
            }
            const u32 first_change = rcx_bwt_same(next_change[0]); // 0xFFFF: the block is one byte over and over
            __syncthreads();
            if (mine == 0xFFFFu) next_change[tid] = (uint16_t)(RCX_BWT_BLOCK + first_change);
            if (tid == 0) next_change[RCX_BWT_THREADS] = (uint16_t)(RCX_BWT_BLOCK + first_change);
            {
                const u32 k0 = 32u * rcx_bwt_tid();
#pragma unroll
                for (u32 i = 0; i < 32; ++i) sa[k0 + i] = (uint16_t)(k0 + i);
            }
            __syncthreads();
            if (first_change == 0xFFFFu) {
                groups = 1; // all rotations are equal: any order is the order, row 0 is where the reference leaves it
                open = 0;
            } else {
                // (byte << 24) | (how the run ends and how long it is: 16 bits) << 8 | the byte behind the run
                const auto run_key = [&](u32 p) {
                    const u32 rest = change[p >> 5] >> (p & 31u);
                    const u32 last = rest ? p + (u32)__builtin_ctz(rest) : (u32)next_change[(p >> 5) + 1u]; // the run's last place
                    const u32 len = last - p + 1u;
                    const u32 c = text[shift + p], b = text[shift + ((p + len) & RCX_BWT_MASK)];
                    // a run ended by a smaller byte sorts before every longer run, shortest first; by a larger one behind, longest first
                    return (c << 24) | ((b > c ? 65535u - len : len) << 8) | b;
                };
#pragma nounroll
                for (u32 down = 0; down < 32; down += 8)
                    rcx_bwt_pass<32, ATOMIC>(sa, cnt, misc, [](u32 x) { return x; }, [&](u32 e) { return (run_key(e) >> down) & 0xFFu; });
                RCX_BWT_PHASE(1)
                groups = rcx_bwt_rerank(sa, rk, misc, run_key, open, starts, behind);
                RCX_BWT_PHASE(2)
            }
        } else {
            // rows by their first two bytes: the second byte first (rows in index order), then the first
            {
                const u32 k0 = 32u * rcx_bwt_tid();
#pragma unroll
                for (u32 i = 0; i < 32; ++i) sa[k0 + i] = (uint16_t)(k0 + i);
            }
            __syncthreads();
#if defined(RCX_BWT_PROBE_NO_SORT) /* diagnostic build: what everything around the sort costs (the output is NOT the transform) */
            if (src == nullptr) {
#endif
#pragma nounroll
            for (u32 second = 1; second < 2; --second)
                rcx_bwt_pass<32, ATOMIC>(sa, cnt, misc, [](u32 x) { return x; }, [&](u32 e) { return (u32)text[shift + ((e + second) & RCX_BWT_MASK)]; });
#if defined(RCX_BWT_PROBE_PASSES) /* diagnostic build: a stable pass by the same digit again changes nothing, it only costs its time */
#pragma nounroll
            for (u32 again = 0; again < RCX_BWT_PROBE_PASSES; ++again)
                rcx_bwt_pass<32, ATOMIC>(sa, cnt, misc, [](u32 x) { return x; }, [&](u32 e) { return (u32)text[shift + (e & RCX_BWT_MASK)]; });
#endif
            RCX_BWT_PHASE(1) // the two passes of the start
            groups = rcx_bwt_rerank(sa, rk, misc, [&](u32 s) { return ((u32)text[shift + s] << 8) | text[shift + ((s + 1u) & RCX_BWT_MASK)]; }, open, starts, behind);
            RCX_BWT_PHASE(2) // regrouping
#if defined(RCX_BWT_PROBE_NO_SORT)
            }
#endif
        }
        for (u32 h = 2; open > 0 && h < RCX_BWT_BLOCK; h <<= 1) {
            if (open > 1024u * RCX_BWT_LIST_BIG) {
#pragma nounroll
                for (u32 high = 0; high < 2; ++high) {
                    const u32 back = high ? 0u : h, down = 8u * high;
                    rcx_bwt_pass<32, ATOMIC>(sa, cnt, misc, [&](u32 x) { return (x - back) & RCX_BWT_MASK; },
                                     [&](u32 e) { return (((u32)rk[e] & RCX_BWT_MASK) >> down) & 0xFFu; });
                }
                RCX_BWT_PHASE(3) // full passes
                // (inside a group of the h-order only the second h bytes can tell two rotations apart)
                const u32 now = rcx_bwt_rerank(sa, rk, misc, [&](u32 s) { return (u32)rk[(s + h) & RCX_BWT_MASK] & RCX_BWT_MASK; }, open, starts, behind);
                RCX_BWT_PHASE(2)
                if (now == groups) break; // nothing split: the block is periodic, `groups` is its period
                groups = now;
            } else {
                // (three lengths of list -- 11264, 3072, 1024 entries -- so that the last rounds, with a few hundred
                // rotations open, pay for a few hundred)
                const u32 iters = open > 1024u * RCX_BWT_LIST_SMALL ? RCX_BWT_LIST_BIG : open > 1024u ? RCX_BWT_LIST_SMALL : 1u;
                const u32 count = rcx_bwt_collect(sa, rk, lst, misc, h, 1024u * iters);
                RCX_BWT_PHASE(4) // collect
                // by group: a padding entry (0xFFFF) sorts behind everything
                const auto same = [](u32 x) { return x; };
                const auto by_group = [&](u32 down) { return [&, down](u32 e) { return e == 0xFFFFu ? 0xFFu : (((u32)rk[e & RCX_BWT_MASK] & RCX_BWT_MASK) >> down) & 0xFFu; }; };
                u32 splits;
                if (iters == RCX_BWT_LIST_BIG) {
#pragma nounroll
                    for (u32 down = 0; down < 16; down += 8) rcx_bwt_pass<RCX_BWT_LIST_BIG, ATOMIC>(lst, cnt, misc, same, by_group(down));
                    RCX_BWT_PHASE(5) // list passes (long list)
                    open = rcx_bwt_place<RCX_BWT_LIST_BIG>(sa, rk, lst, misc, h, count, splits);
                } else if (iters == RCX_BWT_LIST_SMALL) {
#pragma nounroll
                    for (u32 down = 0; down < 16; down += 8) rcx_bwt_pass<RCX_BWT_LIST_SMALL, ATOMIC>(lst, cnt, misc, same, by_group(down));
                    RCX_BWT_PHASE(6) // list passes (short list)
                    open = rcx_bwt_place<RCX_BWT_LIST_SMALL>(sa, rk, lst, misc, h, count, splits);
                } else {
#pragma nounroll
                    for (u32 down = 0; down < 16; down += 8) rcx_bwt_pass<1u, ATOMIC>(lst, cnt, misc, same, by_group(down));
                    RCX_BWT_PHASE(6)
                    open = rcx_bwt_place<1u>(sa, rk, lst, misc, h, count, splits);
                }
                RCX_BWT_PHASE(7) // place
                if (splits == 0) break; // periodic (cannot happen with a list this short, but it is the same test)
                groups += splits;
            }
        }
#if defined(RCX_BWT_PROBE_RERANKS) /* diagnostic build, data whose rows are all final by now: the same ranks again, for their time */
#pragma nounroll
        for (u32 again = 0; again < RCX_BWT_PROBE_RERANKS; ++again)
            (void)rcx_bwt_rerank(sa, rk, misc, [&](u32 s) { return (u32)rk[(s + 2u) & RCX_BWT_MASK] & RCX_BWT_MASK; }, open, starts, behind);
#endif
        RCX_BWT_PHASE(8) // (loop ends)
        // the last column (blksort.h:511-518): byte in front of every row's rotation
        u32 s[32];
        {
            const u32 k0 = 32u * rcx_bwt_tid();
            rcx_bwt_rows32(sa, k0, s);
#pragma unroll
            for (u32 i = 0; i < 32; ++i)
                if (s[i] == 0) misc[48] = k0 + i;
        }
        __syncthreads(); // SA is in registers, RK is done with
        (void)rcx_bwt_stage_in(text, in, RCX_BWT_BLOCK);
        __syncthreads();
        u8* stage = lds + RCX_BWT_FWD_SA;
        const u32 oshift = (u32)(reinterpret_cast<uintptr_t>(out) & 15u);
        {
            u8* mine = stage + oshift + 32u * rcx_bwt_tid();
#pragma unroll
            for (u32 i = 0; i < 32; ++i) mine[i] = text[shift + ((s[i] + RCX_BWT_MASK) & RCX_BWT_MASK)];
        }
        if (tid == 0) {
            // all rotations equal: the reference's sort moves nothing and row 0 stays where it is (rcx_bwt_tie.hpp)
            const u32 row = groups == 1 ? 0u : misc[48];
            stage[oshift + RCX_BWT_BLOCK] = (u8)(row & 0xFFu); // a uint16_t copied on a little-endian host (blksort.h:518)
            stage[oshift + RCX_BWT_BLOCK + 1] = (u8)(row >> 8);
            if (groups > 1 && groups < RCX_BWT_BLOCK) {
                if (groups & (groups - 1u)) {
                    rcx_flag(status, RCX_ST_CORRUPT, b); // a period divides 32768: anything else is a bug here, not data
                } else {
                    const u32 at = atomicAdd(&ties[0], 1u);
                    ties[RCX_BWT_TIES_HEAD + 2 * at] = (u32)b;
                    ties[RCX_BWT_TIES_HEAD + 2 * at + 1] = groups;
                }
            }
        }
        __syncthreads();
        RCX_BWT_PHASE(9) // column gathered
        rcx_bwt_stage_out(out, stage, RCX_BWT_ENCODED);
        __syncthreads();
        RCX_BWT_PHASE(10) // block out
#if defined(RCX_BWT_STAMP)
        stamp_[8] += 0; // (slot 8 is time between the last round and the column: loop control)
#endif
    }
#if defined(RCX_BWT_STAMP)
    if (tid == 0)
        for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&rcx_bwt_stamp_out[i_], stamp_[i_]);
#endif
}

// Periodic blocks: the row index as the reference's sort leaves it.  One wave per listed block; all its lanes run the
// replay together (rcx_bwt_tie.hpp).
__global__ __launch_bounds__(64) void rcx_bwt_tie_k(const u8* __restrict__ src, u8* __restrict__ dst, const u32* __restrict__ ties, u32* status)
{
    extern __shared__ __attribute__((aligned(16))) u8 rcx_bwt_lds[];
    uint16_t* rows = reinterpret_cast<uint16_t*>(rcx_bwt_lds + RCX_BWT_TIE_ROWS);
    u8* word = rcx_bwt_lds + RCX_BWT_TIE_WORD;
    RcxTieSort::Part* stack = reinterpret_cast<RcxTieSort::Part*>(rcx_bwt_lds + RCX_BWT_TIE_STACK);
    const u32 lane = threadIdx.x;
    const u32 count = ties[0];
    for (u32 i = blockIdx.x; i < count; i += gridDim.x) {
        const u64 b = ties[RCX_BWT_TIES_HEAD + 2 * i];
        const u32 p = ties[RCX_BWT_TIES_HEAD + 2 * i + 1];
        const u8* in = src + b * RCX_BWT_BLOCK;
        for (u32 r = lane; r < p; r += 64) word[r] = in[r];
        __syncthreads();
        RcxTieSort t{rows, word, p - 1u, RCX_BWT_BLOCK};
        // the classes' places among the rotations of the word first (the rows' space is free until then): less() is then
        // one compare instead of a walk along two rows
        uint16_t* rank = reinterpret_cast<uint16_t*>(rcx_bwt_lds + RCX_BWT_TIE_RANK);
        uint16_t* hist = reinterpret_cast<uint16_t*>(rcx_bwt_lds + RCX_BWT_TIE_HIST);
        t.rank_classes(rank, hist, rows, rows + 16384, hist, reinterpret_cast<uint16_t*>(rcx_bwt_lds + RCX_BWT_TIE_SUMS));
        t.rank = rank;
        for (u32 r = lane; r < RCX_BWT_BLOCK; r += 64) rows[r] = (uint16_t)r;
        __syncthreads();
        const bool done = t.run(stack);
        if (lane == 0 && !done) rcx_flag(status, RCX_ST_CORRUPT, b); // (cannot happen: the stack bound)
        __syncthreads();
        for (u32 r = lane; r < RCX_BWT_BLOCK; r += 64)
            if (rows[r] == 0) {
                u8* out = dst + b * RCX_BWT_ENCODED + RCX_BWT_BLOCK;
                out[0] = (u8)(r & 0xFFu);
                out[1] = (u8)(r >> 8);
            }
        __syncthreads();
    }
}

// The inverse (blksort.h:543-679).  next[r] = where the r-th smallest byte of the column sits (equal bytes in their
// order): the stable counting pass above with the column byte as the digit.  The reference then walks
// p = next[top]; out[i] = column[p]; p = next[p] for 32768 steps.  Here the rows congruent to next[top] mod 32 are
// 1024 starting points; every thread walks from its start to the next start it meets (about 32 steps, all threads at
// once; the longest piece, about 250, sets the time), the pieces are put in order by pointer jumping over the 1024
// (start -> start it ran into) links, and a second walk writes the bytes where they belong -- dealt out in stretches
// of 32 rows (the first walk notes every 32nd row it passes), so that it does not wait for the longest piece again.
// More and shorter pieces (2 or 4 walks per thread, a start at every 16th or 8th row) shorten the first walk too, but
// the pointer jumping grows with the number of pieces and measured it costs more than it saves (inverse of 1 GiB, 1 / 2
// / 4 walks: uniform 7.90 / 8.64 / 8.38 ms, Canterbury 10.7 / 13.3 / 15.2 ms).  `next` is a permutation whatever the input, so every walk ends;
// if the walk from next[top] closes after C < 32768 steps (a periodic block, or garbage) the reference keeps going
// round, and so do the writes here (position + m C).
template <bool ATOMIC>
__global__ __launch_bounds__(1024) void rcx_bwt_inv_k(const u8* __restrict__ src, u64 nblocks, u8* __restrict__ dst, u32* __restrict__ work, u32* status)
{
    extern __shared__ __attribute__((aligned(16))) u8 rcx_bwt_lds[];
    u8* lds = rcx_bwt_lds;
    uint16_t* next = reinterpret_cast<uint16_t*>(lds + RCX_BWT_INV_NEXT);
    uint16_t* cnt = reinterpret_cast<uint16_t*>(lds + RCX_BWT_INV_CNT);
    u32* link = reinterpret_cast<u32*>(lds + RCX_BWT_INV_LINK);
    u32* mark_of = reinterpret_cast<u32*>(lds + RCX_BWT_INV_CNT);
    uint16_t* mark_row = reinterpret_cast<uint16_t*>(lds + RCX_BWT_INV_CNT + 4096u);
    uint16_t* lens = reinterpret_cast<uint16_t*>(lds + RCX_BWT_INV_CNT + 6144u);
    u32* misc = reinterpret_cast<u32*>(lds + RCX_BWT_INV_MISC);
    const u32 tid = threadIdx.x;
    for (;;) {
        if (tid == 0) misc[49] = atomicAdd(work, 1u);
        __syncthreads();
        const u64 b = rcx_bwt_same(misc[49]);
        if (b >= nblocks) break;
        const u8* in = src + b * RCX_BWT_ENCODED;
        u8* out = dst + b * RCX_BWT_BLOCK;
        u8* enc = lds + RCX_BWT_INV_ENC;
        const u32 shift = rcx_bwt_stage_in(enc, in, RCX_BWT_ENCODED);
        __syncthreads();
        const u8* col = enc + shift;
        if (tid == 0) misc[40] = 0; // how many stretches the first walk notes (the pass's scans use misc[0..15]; its barriers order this)
        {
            const u32 k0 = 32u * rcx_bwt_tid();
#pragma unroll
            for (u32 i = 0; i < 32; ++i) next[k0 + i] = (uint16_t)(k0 + i);
        }
        __syncthreads();
        rcx_bwt_pass<32, ATOMIC>(next, cnt, misc, [](u32 x) { return x; }, [&](u32 e) { return (u32)col[e]; });
        u32 top = rcx_bwt_same((u32)col[RCX_BWT_BLOCK] | ((u32)col[RCX_BWT_BLOCK + 1] << 8));
        if (top >= RCX_BWT_BLOCK) { // the reference would read outside its arrays (blksort.h:663)
            if (tid == 0) rcx_flag(status, RCX_ST_CORRUPT, b);
            top &= RCX_BWT_MASK;
        }
        const u32 x0 = rcx_bwt_same(next[top]);
        // piece v starts at row residue + 32 v and is walked by thread v; every 32nd row on the way is noted, so that
        // the second walk can be dealt out in stretches of 32 whatever the pieces' lengths are
        const u32 residue = x0 & 31u, first = x0 >> 5;
        u32 r = residue + 32u * tid, steps = 0;
        for (;;) {
            r = next[r];
            ++steps;
            // (a permutation comes back to its start: the bound on the steps never cuts in)
            if ((r & 31u) == residue || steps >= RCX_BWT_BLOCK) break;
            if ((steps & 31u) == 0) { // at most 1024 of these in all: the pieces have 32768 rows between them
                const u32 m = atomicAdd(&misc[40], 1u);
                mark_row[m] = (uint16_t)r;
                mark_of[m] = (tid << 16) | (steps >> 5);
            }
        }
        lens[tid] = (uint16_t)steps; // (32768 = 0x8000 fits)
        // distance from every piece to the first one along the links (the first piece is made a sink): pointer jumping
        // on one word per piece, link << 16 | distance (16 bits hold every distance along the walk, at most 32768;
        // off the walk they may wrap into nothing that is used: the link field is re-masked)
        u32 mine = tid == first ? first << 16 : ((r >> 5) << 16) | steps;
        link[tid] = mine;
        __syncthreads();
#pragma nounroll
        for (u32 round = 0; round < 10; ++round) {
            const u32 there = link[mine >> 16];
            mine = (there & 0xFFFF0000u) | ((mine + there) & 0xFFFFu);
            __syncthreads();
            link[tid] = mine;
            __syncthreads();
        }
        if (tid == first) misc[41] = steps + ((r >> 5) == first ? 0u : link[r >> 5] & 0xFFFFu); // the length of the whole walk's cycle
        __syncthreads();
        const u32 cycle = rcx_bwt_same(misc[41]), marks = rcx_bwt_same(misc[40]);
        u8* stage = lds + RCX_BWT_INV_OUT;
        const u32 oshift = (u32)(reinterpret_cast<uintptr_t>(out) & 15u);
        // second walk: the head of the thread's own piece, and one noted stretch
#pragma nounroll
        for (u32 job = 0; job < 2; ++job) {
            u32 v = tid, from = residue + 32u * tid, skip = 0;
            bool have = job == 0;
            if (job == 1 && tid < marks) {
                v = mark_of[tid] >> 16;
                skip = 32u * (mark_of[tid] & 0xFFFFu);
                from = mark_row[tid];
                have = true;
            }
            const u32 there = link[v];
            if (have && (v == first || (there >> 16) == first)) {
                const u32 at = (v == first ? 0u : cycle - (there & 0xFFFFu)) + skip;
                u32 n = (u32)lens[v] - skip;
                n = n < 32u ? n : 32u;
                u32 row = from;
                for (u32 i = 0; i < n; ++i) {
                    const u8 c = col[row];
                    for (u32 p = at + i; p < RCX_BWT_BLOCK; p += cycle) stage[oshift + p] = c;
                    row = next[row];
                }
            }
        }
        __syncthreads();
        rcx_bwt_stage_out(out, stage, RCX_BWT_BLOCK);
        __syncthreads();
    }
}
