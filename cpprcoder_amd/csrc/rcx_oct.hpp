// rcx_oct.hpp -- the many-lane adaptive coder kernels for gfx950 that the library runs by default:
//   rcx_dec_quad_k   decode, 4 lanes per block (16 blocks per wave)
//   rcx_enc_mc5_k    encode, five waves per 64 blocks (model x3 / arithmetic / writer)
// plus the cross-lane helpers they share with the rANS kernels.  The superseded kernels (8 lanes per block for both
// directions, the four-wave encoder) stay selectable for comparison and live in variants/rcx_variants.hpp.
#pragma once
// included at the end of rcx_kernels.hpp (uses rcx_flag, rcx_wave_max, rcx_byte_of from there)

template <int CTRL>
__device__ __forceinline__ u32 rcx_dpp(u32 x)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true);
}
// sum over the 8 lanes of an octet, result in all of them
__device__ __forceinline__ u32 rcx_oct_sum(u32 x)
{
    x += rcx_dpp<0xB1>(x);  // quad_perm [1,0,3,2]
    x += rcx_dpp<0x4E>(x);  // quad_perm [2,3,0,1]
    x += rcx_dpp<0x141>(x); // row_half_mirror: lane i <-> 7-i
    return x;
}
// exclusive prefix over the 8 lanes of an octet (lane j gets x_0 + ... + x_{j-1})
__device__ __forceinline__ u32 rcx_oct_excl_scan(u32 x, u32 m1, u32 m2, u32 m4)
{
    u32 tot = x, pre = 0, o;
    o = rcx_dpp<0xB1>(tot);
    pre += o & m1;
    tot += o;
    o = rcx_dpp<0x4E>(tot);
    pre += o & m2;
    tot += o;
    o = rcx_dpp<0x141>(tot); // the other quad's total
    pre += o & m4;
    return pre;
}

// The multi-wave encoders: 16 symbols per pipeline step, double-buffered rings between the waves
#define RCX_MC_CHUNK 16
#define RCX_MC_THREADS 256
#define RCX_MC_RING_U4 (2 * RCX_MC_CHUNK * RCX_LANES)
#define RCX_MC_LDS_U4 (RCX_LDS_U4 + RCX_MC_RING_U4)

__device__ __forceinline__ void rcx_lds_barrier()
{
    // LDS hand-off between the waves of one workgroup: drain this wave's LDS operations, then
    // meet.  Deliberately not __syncthreads(): that also waits for vmcnt(0) and would stall the
    // coder wave on its own in-flight global stores every chunk.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#if defined(RCX_STAMP) /* diagnostic build only (tools/diag/stamp_encode.py) */
static __device__ unsigned long long rcx_stamp_out[16];
#define rcx_stamp_wait stamp_wait_
#endif
// ===========================================================================
// Decode, 4 lanes per block ("quad"): 16 blocks per wave, 1024 waves for 1 GiB of 64 KiB
// blocks = one wave per SIMD.  A lone wave issues one instruction -- vector, scalar, s_nop or
// s_waitcnt alike -- every 4 cycles and hides no latency (tools/diag/ubench.hip), so the kernel is
// written for the fewest instructions per symbol of any kind, and for work between the one
// dependent LDS read and its use.
//
// Model (cpprcoder.h:1094-1243): the alphabet is split 16 nodes x 16 symbols.  Lane j of the
// block's quad keeps in REGISTERS U1..U4 = the counts of all symbols below node 4j+1, ..., 4j+4
// (absolute cumulative sums: the upper bounds of its four nodes; U4 of the last lane is the
// total); the 256 counts live in LDS, node n as 64 contiguous bytes of which lane j reads counts
// 4j..4j+3 (one ds_read_b128).
//
// find() (cpprcoder.h:1220-1242) in the scaled domain (see DecLane), without selects:
//   x_k = low - U_k*t wraps past zero exactly for the bounds above low, so
//     * the number of bounds that do NOT borrow, summed over the quad, is the node index,
//     * the unsigned minimum of low and all x_k over the quad is low - cum(node)*t;
//   round 2 is the same over the node's 16 counts, and the unsigned maximum of the x over the
//   quad is the (wrapped) distance to the smallest bound above, so the new range count*t is
//   min - max (mod 2^32): cum(c+1)*t - cum(c)*t, no multiply, no select of the count.
// A target at or past the total (corrupt input only) leaves no borrow in round 1 and "node 16",
// whose counts are a scratch area behind the block's table.  Such a block is detected, not
// decoded: it is marked in `redo` and decoded again by rcx_dec_adaptive_k, which has the
// reference's fall-through for that case.
//
// Input: no bit window.  The position in the stream is a bit offset `bp8`; the two ring dwords
// around it are read right after each renormalisation (for the NEXT symbol, so their latency is
// never waited for) and the next four bytes are one v_alignbit + one byte swap away.
// ===========================================================================
#define RCX_QUAD_BLOCKS 16
#define RCX_QUAD_STAGE 16 /* divisor entries staged per refill: one top-up interval */
// LDS of one wave: the staged divisors | four 4352-byte table groups | sixteen 144-byte input rings.
//   A ds_read_b128 is served in four groups of 16 lanes -- quads {0,3,5,6}, {1,2,4,7}, {8,11,13,14},
//   {9,10,12,15} -- one LDS cycle each if the group's four 64-byte reads fall into four different quarters
//   of the 256-byte bank row (MI355X_MICROARCH.md, LDS).  So the four blocks of such a group share a table
//   group: row n (256 bytes) holds node n of all four, block s in quarter s; whatever nodes the four quads
//   ask for, they read different quarters.  (One table per block: 16 of the 36 LDS cycles per symbol were
//   bank conflicts.)  Row 16 is scratch ("node 16"): per block three 16-byte groups of decoded output
//   waiting for the fourth, then 16 bytes where skipped ring writes go.
//   A ring is 32 dwords + slot 32 (repeats slot 0) + 12 spare bytes; 36 dwords apart, the rings of the 8
//   quads of a half-wave start 4 banks apart.
#define RCX_QUAD_GROUP_BYTES 4352
#define RCX_QUAD_RING_BYTES 144
#define RCX_QUAD_LDS_BYTES (RCX_QUAD_STAGE * 16 + 4 * RCX_QUAD_GROUP_BYTES + RCX_QUAD_BLOCKS * RCX_QUAD_RING_BYTES) /* 19.5 KiB: two 4-wave workgroups per CU */

// LDS addresses computed inside the instruction sequences come back as 32-bit offsets
typedef u32 RcxV4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) RcxV4 RcxLdsV4;
typedef __attribute__((address_space(3))) u32 RcxLdsU32;
typedef RcxV4 RcxDivQv;          // a staged divisor as four dwords: mul, shift (| total << 5), addend low, addend high
typedef RcxLdsV4 RcxLdsDivQ;

// divisor entry as the quad decoder stages it: the 64-bit addend is read as a register pair
struct alignas(16) DivQ {
    u32 mul, st; // st = total << 5 | shift
    u64 add;
};

__device__ __forceinline__ u32 rcx_quad_sum(u32 x)
{
    x += rcx_dpp<0xB1>(x); // quad_perm [1,0,3,2]
    x += rcx_dpp<0x4E>(x); // quad_perm [2,3,0,1]
    return x;
}
__device__ __forceinline__ u32 rcx_umin(u32 a, u32 b) { return a < b ? a : b; }
__device__ __forceinline__ u32 rcx_umax(u32 a, u32 b) { return a > b ? a : b; }
__device__ __forceinline__ u32 rcx_quad_min(u32 x)
{
    x = rcx_umin(x, rcx_dpp<0xB1>(x));
    x = rcx_umin(x, rcx_dpp<0x4E>(x));
    return x;
}
__device__ __forceinline__ u32 rcx_quad_max(u32 x)
{
    x = rcx_umax(x, rcx_dpp<0xB1>(x));
    x = rcx_umax(x, rcx_dpp<0x4E>(x));
    return x;
}
__device__ __forceinline__ u32 rcx_quad_or(u32 x)
{
    x |= rcx_dpp<0xB1>(x);
    x |= rcx_dpp<0x4E>(x);
    return x;
}
__device__ __forceinline__ u32 rcx_quad_excl_scan(u32 x, u32 m1, u32 m2)
{
    u32 o = rcx_dpp<0xB1>(x);
    u32 pre = o & m1;
    const u32 tot = x + o;
    o = rcx_dpp<0x4E>(tot);
    pre += o & m2;
    return pre;
}

// The compressed stream of one block as its quad reads it (all 4 lanes hold the same state and
// store the same values).
// Ring: dword d of the stream (counted from `origin`, the 16-byte aligned address at or below the
// first payload byte) lives in ring[d % 32]; ring[32] repeats ring[0] so that the pair (d, d+1)
// is always one ds_read2_b32.  Every 16 symbols (which consume at most 12 dwords) topup() moves the 16-byte
// piece it requested the time before into the ring and requests the next one -- unconditionally: a piece the
// ring has no room for is written to the scratch area instead and asked for again.  One piece per 16 symbols is
// one byte per symbol: a quad that needs more (expanding data for a while, or a burst of improbable symbols)
// falls behind and is refilled synchronously, up to 24 dwords ahead, on a cold branch.
#if !defined(RCX_TOPUP_PIECES)
#define RCX_TOPUP_PIECES 1 /* 16-byte pieces requested per top-up on the fast path (2: 1 % slower on 1.0-ratio data) */
#endif
struct QuadInput {
    u32 low, range;
    u32 bp8;        // bits of the stream consumed, counted from `origin`
    u32 w0, w1;     // ring dwords (bp8 >> 5) and (bp8 >> 5) + 1, raw (memory order)
    u32 n4;         // the four stream bytes at bp8, first one on top
    u32* ring;      // this block's ring
    U4* skipped;    // 16 bytes that take the ring writes that are skipped
    u32 wr;         // dwords written to the ring so far
    u32 nfit;       // how many of pendA, pendB (requested at the last top-up) the ring has room for
    U4 pendA, pendB;
    const u8* origin;
    u32 last_off;   // byte offset from origin of the last 16-byte piece that may be loaded
    u32 body8;      // bit offset of the first payload byte (after the 8 header bytes)

    // a piece at or past the end of the stream repeats the stream's last piece (never used by a valid
    // stream; the piece holding the last byte stays inside that byte's page)
    __device__ __forceinline__ U4 load16(u32 off) const
    {
        return *reinterpret_cast<const U4*>(origin + (off < last_off ? off : last_off));
    }
    __device__ __forceinline__ void ring_put(const U4& piece, bool really)
    {
        const u32 slot = wr % RCX_RING_DW; // a multiple of 4: the piece never wraps
        *(really ? reinterpret_cast<U4*>(ring + slot) : skipped) = piece;
        ring[really && slot == 0 ? RCX_RING_DW : RCX_RING_DW + 1] = piece.x;
        wr += really ? 4u : 0u;
    }
    __device__ __forceinline__ void fetch_pair()
    {
        const u32* at = ring + ((bp8 >> 5) % RCX_RING_DW);
        w0 = at[0];
        w1 = at[1];
    }
    // cpprcoder.h:877-896 + :859-870; `s` must hold at least 8 bytes.  Returns the declared size.
    __device__ __forceinline__ u32 begin(const u8* s, const u8* stream_end, u32* block_ring, U4* scratch16)
    {
        skipped = scratch16;
        const u32 declared = (u32)s[0] | ((u32)s[1] << 8) | ((u32)s[2] << 16) | ((u32)s[3] << 24);
        low = ((u32)s[4] << 24) | ((u32)s[5] << 16) | ((u32)s[6] << 8) | (u32)s[7];
        range = 0x00FFFFFFu;
        ring = block_ring;
        const u8* body = s + 8;
        origin = body - ((uintptr_t)body & 15);
        last_off = (u32)(stream_end - 1 - origin) & ~15u;
        wr = 0;
        for (u32 r = 0; r < 6; ++r) ring_put(load16(16 * r), true); // prologue: 24 dwords, synchronously
        nfit = 0;
        pendA.x = pendA.y = pendA.z = pendA.w = 0;
        pendB = pendA;
        body8 = 8u * (u32)(body - origin);
        bp8 = body8;
        fetch_pair();
        n4 = rcx_bswap(rcx_funnel_shr(w1, w0, bp8));
        return declared;
    }
    // a lane without a block: reads 16 bytes at the start of the compressed buffer, over and over
    __device__ __forceinline__ void idle(const u8* anywhere, u32* block_ring, U4* scratch16)
    {
        skipped = scratch16;
        low = 0;
        range = 0x01000000u;
        ring = block_ring;
        origin = anywhere - ((uintptr_t)anywhere & 15);
        last_off = 0;
        wr = 24;
        nfit = 0;
        pendA.x = pendA.y = pendA.z = pendA.w = 0;
        pendB = pendA;
        body8 = bp8 = 0;
        w0 = w1 = n4 = 0;
    }
    __device__ __forceinline__ void topup()
    {
        ring_put(pendA, nfit >= 1);
#if RCX_TOPUP_PIECES > 1
        ring_put(pendB, nfit >= 2);
#endif
        const u32 rd = bp8 >> 5; // ring[rd % 32 ...] are unread
        if (rcx_any(wr - rd < 14u)) { // the next 16 symbols may need 12 dwords and the pair after them
            asm volatile("" ::: "memory"); // keep this a branch: taken only on a run of very improbable symbols
            while (__any(wr - rd <= 20u)) ring_put(load16(4 * wr), wr - rd <= 20u);
        }
        const u32 room = (rd + RCX_RING_DW - wr) >> 2;
        nfit = room < (u32)RCX_TOPUP_PIECES ? room : (u32)RCX_TOPUP_PIECES;
        pendA = load16(4 * wr);
#if RCX_TOPUP_PIECES > 1
        pendB = load16(4 * wr + 16);
#endif
    }
    // stream bytes consumed so far, header included (cpprcoder.h:901-903)
    __device__ __forceinline__ u64 taken() const { return 8 + (u64)((bp8 - body8) >> 3); }
};

// A workgroup is WAVES independent waves: with few blocks, 4 waves per workgroup land one on each SIMD of a
// CU (single-wave workgroups do not: measured 25.4 -> 19.3 ms per GiB at 16384 blocks).
#define RCX_QUAD_DEC_WAVES 4
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rcx_dec_quad_k(const u8* __restrict__ comp, u64 comp_size, const u64* __restrict__ offsets,
                                                             u64 nblocks, u32 block, u64 n, u8* __restrict__ dst,
                                                             const DivEntry* __restrict__ divtab, u32* status,
                                                             u32* __restrict__ redo, u32 quads_used)
{
    __shared__ __attribute__((aligned(256))) u8 lds_all[WAVES * RCX_QUAD_LDS_BYTES];
    const u32 lane = threadIdx.x & 63u;
    const u32 wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u8* lds = lds_all + wave_in_wg * RCX_QUAD_LDS_BYTES;
    const u32 j = lane & 3u, quad = lane >> 2;
    // quads_used (1, 2, 4, 8 or 16) of the wave's 16 quads carry a block (rcx_api.hip picks it from the block count
    // so that every SIMD of the machine has a wave before any wave carries 16 blocks).  The other quads decode
    // the block of quad (quad mod quads_used) along with it -- same instructions, well-defined state -- and store nothing.
    const bool in_use = quad < quads_used;
    const u64 blk = ((u64)blockIdx.x * WAVES + wave_in_wg) * quads_used + (quad & (quads_used - 1u));
    bool live = blk < nblocks;
    const u64 at = live ? blk * (u64)block : 0;
    u32 len = live ? (u32)((n - at) < (u64)block ? (n - at) : (u64)block) : 0u;

    DivQ* stage = reinterpret_cast<DivQ*>(lds);
    // table group and quarter of this quad (see the layout above)
    const u32 group = 2u * (quad >> 3) + ((0x96u >> (quad & 7u)) & 1u), quarter = (quad & 7u) >> 1;
    u8* mine = lds + RCX_QUAD_STAGE * 16 + group * RCX_QUAD_GROUP_BYTES + quarter * 64;
    U4* leaves = reinterpret_cast<U4*>(mine) + j; // node n: leaves[n * 16]
    U4* parked = reinterpret_cast<U4*>(mine + 16 * 256);
    u32* block_ring = reinterpret_cast<u32*>(lds + RCX_QUAD_STAGE * 16 + 4 * RCX_QUAD_GROUP_BYTES + quad * RCX_QUAD_RING_BYTES);
    // model: cpprcoder.h:1094-1132, every count 1
    {
        U4 v;
        v.x = v.y = v.z = v.w = 1;
#pragma unroll
        for (u32 q = 0; q < 17; ++q) leaves[q * 16] = v; // 16 nodes + the scratch row
    }
    u32 U1 = 64u * j + 16, U2 = U1 + 16, U3 = U1 + 32, U4_ = U1 + 48;
    const u32 T0 = 4u * j;
    const u32 m1 = (j & 1u) ? ~0u : 0u, m2 = (j & 2u) ? ~0u : 0u;

    QuadInput in;
    u64 stream_len = 0;
    if (live) {
        const u64 s0 = offsets[blk], s1 = offsets[blk + 1];
        stream_len = s1 - s0;
        if (s1 < s0 || s1 > comp_size || stream_len < 9) {
            if (j == 0 && in_use) rcx_flag(status, RCX_ST_CORRUPT, blk);
            live = false;
            len = 0;
        } else {
            const u32 declared = in.begin(comp + s0, comp + s1, block_ring, parked + 3);
            if (declared != len) {
                if (j == 0 && in_use) rcx_flag(status, RCX_ST_CORRUPT, blk);
                live = false;
                len = 0;
            }
        }
    }
    if (!live) in.idle(comp, block_ring, parked + 3);

    const u32 maxlen = rcx_wave_max(len);
    u8* out = dst + at;
    // The fast loop (16 symbols at a go, no per-symbol length test, 16-byte stores) runs as far as every block of the
    // wave has whole groups of 16 and its output is 16-byte aligned; the rest -- the ragged end of a buffer's last
    // block, the whole wave if an output is unaligned -- is decoded symbol by symbol behind it.
    u32 fast_end;
    {
        u32 mine = live ? (len & ~15u) : 0xFFFFFFF0u; // (a quad without a block sets no limit)
        if (live && (reinterpret_cast<uintptr_t>(out) & 15u) != 0) mine = 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const u32 other = (u32)__shfl_xor((int)mine, o, 64);
            mine = mine < other ? mine : other;
        }
        fast_end = mine == 0xFFFFFFF0u ? 0u : mine;
    }
    const bool leader = live && in_use && j == 0;

    // One symbol; the owning lane ORs it into WORD at bit SHIFT (the quad's other lanes OR in 0).
    //
    // The three arithmetic cores are written out as instruction sequences: a lone wave pays 4 cycles
    // for every s_nop the compiler has to put between a compare and the use of its mask, or between
    // a vector write and a DPP read of it (2 wait states each on gfx950), so compares go to four
    // different mask registers before any is used, and every DPP step has two independent
    // instructions in front of it.  Only register-to-register vector instructions are in there;
    // LDS and global accesses stay with the compiler (and its s_waitcnt placement).
    const u32 T0p3 = T0 + 3;
    const u32 leaves_lds = (u32)reinterpret_cast<uintptr_t>(leaves); // low half of a flat LDS address = the LDS offset
    const u32 ring_lds = (u32)reinterpret_cast<uintptr_t>(block_ring);
#if defined(RCX_STAMP_DEC) /* diagnostic build only (tools/diag/stamp_quad.py): where does one symbol's time go? */
#define RCX_QUAD_STAMP(i) if (stamp_now_) stamp_t_[stamp_at_ + (i)] = __builtin_amdgcn_s_memtime();
#else
#define RCX_QUAD_STAMP(i)
#endif
#define RCX_QP1 "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define RCX_QP2 "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
// What a symbol leaves for the one behind it (below): its number (valid in the lane that owns it), that lane's mask, and
// the LDS address of the lane's four counts of the node.
    u32 p_sym_ = 0, p_la_ = leaves_lds;
    u64 p_own_ = 0;
// One symbol.  Its byte and the +1 on its count are NOT made here but by the next symbol (HP = 1: PWORD, PSHIFT are that
// earlier symbol's word and bit position) or by RCX_QUAD_DEC_FINISH: nothing the coder state needs depends on them, so
// they fill the slots the node index's steps across the quad need anyway (the ds_add still comes before the next leaf
// read: LDS serves a wave's operations in order) and the wait for the leaf read.
#define RCX_QD_PREV_A_0 "s_nop 1\n\t"
#define RCX_QD_PREV_A_1 "v_cndmask_b32_e64 %[pown], 0, 1, %[pc]\n\t"                                                \
                        "v_and_b32 %[pad], 3, %[psym]\n\t"                                                         \
                        "v_lshl_add_u32 %[pad], %[pad], 2, %[pla]\n\t" /* LDS address of the earlier symbol's count */
#define RCX_QD_PREV_S_0
#define RCX_QD_PREV_S_1 "\n\tv_cndmask_b32_e64 %[pye], 0, %[psym], %[pc]\n\t"                                       \
                        "v_lshl_or_b32 %[pword], %[pye], %[psh], %[pword]"
#define RCX_QUAD_DEC_SYMBOL(K, HP, PWORD, PSHIFT)                                                           \
    {                                                                                                      \
        /* cpprcoder.h:926-940 (in.n4 = the next four stream bytes, ready since the previous symbol) */    \
        const u32 k8_ = rcx_clz(in.range) & 0x18u;                                                         \
        in.low = (u32)((((u64)in.low << 32) | in.n4) << k8_ >> 32);                                        \
        in.range <<= k8_;                                                                                  \
        const DivQ k_ = (K);                                                                               \
        const u32 t_ = (u32)(((u64)in.range * k_.mul + k_.add) >> 32) >> (k_.st & 31u); /* :904 */          \
        /* round 1: which of the 16 nodes.  node = bounds at or below low, rem = low - the largest */      \
        const u32 a1_ = rcx_mul24(U1, t_), a2_ = rcx_mul24(U2, t_), a3_ = rcx_mul24(U3, t_);               \
        const u32 a4_ = rcx_mul24(U4_, t_);                                                                \
        u32 node_, rem_, ro_, la_, x1_, x2_, x3_, x4_, pown_, pad_, pye_;                                  \
        u64 c1_, c2_, c3_, c4_, cz_;                                                                       \
        asm volatile("v_sub_co_u32_e64 %[x1], %[c1], %[low], %[a1]\n\t"                                    \
                     "v_sub_co_u32_e64 %[x2], %[c2], %[low], %[a2]\n\t"                                    \
                     "v_sub_co_u32_e64 %[x3], %[c3], %[low], %[a3]\n\t"                                    \
                     "v_sub_co_u32_e64 %[x4], %[c4], %[low], %[a4]\n\t"                                    \
                     "v_subb_co_u32_e64 %[nd], %[cz], 4, 0, %[c1]\n\t" /* (the borrows stay: the update below) */ \
                     "v_subb_co_u32_e64 %[nd], %[cz], %[nd], 0, %[c2]\n\t"                                 \
                     "v_subb_co_u32_e64 %[nd], %[cz], %[nd], 0, %[c3]\n\t"                                 \
                     "v_subb_co_u32_e64 %[nd], %[cz], %[nd], 0, %[c4]\n\t"                                 \
                     RCX_QD_PREV_A_##HP                                                                    \
                     : [nd] "=&v"(node_), [x1] "=&v"(x1_), [x2] "=&v"(x2_), [x3] "=&v"(x3_), [x4] "=&v"(x4_), \
                       [c1] "=&s"(c1_), [c2] "=&s"(c2_), [c3] "=&s"(c3_), [c4] "=&s"(c4_), [cz] "=&s"(cz_), \
                       [pown] "=&v"(pown_), [pad] "=&v"(pad_)                                              \
                     : [low] "v"(in.low), [a1] "v"(a1_), [a2] "v"(a2_), [a3] "v"(a3_), [a4] "v"(a4_),      \
                       [pc] "s"(p_own_), [psym] "v"(p_sym_), [pla] "v"(p_la_));                            \
        if (HP) (void)__hip_atomic_fetch_add(reinterpret_cast<RcxLdsU32*>(pad_), pown_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); /* :916, the earlier symbol's */ \
        asm volatile("v_add_u32_dpp %[nd], %[nd], %[nd] " RCX_QP1                                          \
                     "v_add_u32 %[bp], %[bp], %[k8]\n\t" /* the stream position moves on */                \
                     "v_bfe_u32 %[ro], %[bp], 5, 5\n\t"  /* ring slot of the next pair ... */               \
                     "v_lshl_add_u32 %[ro], %[ro], 2, %[rb]" /* ... and its LDS address (formed here: a vector instruction right \
                                                               behind the sequence that reads a register of it costs an s_nop) */ \
                     : [nd] "+v"(node_), [ro] "=&v"(ro_), [bp] "+v"(in.bp8)                                 \
                     : [k8] "v"(k8_), [rb] "v"(ring_lds));                                                 \
        {                                                                                                  \
            const RcxLdsU32* at_ = reinterpret_cast<const RcxLdsU32*>(ro_); /* the stream bytes of the next symbol */ \
            in.w0 = at_[0];                                                                                \
            in.w1 = at_[1];                                                                                \
        }                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        asm volatile("v_add_u32_dpp %[nd], %[nd], %[nd] " RCX_QP2                                          \
                     "v_lshl_add_u32 %[la], %[nd], 8, %[lvb]" /* LDS address of the lane's 4 counts of the node */ \
                     : [nd] "+v"(node_), [la] "=&v"(la_)                                                    \
                     : [lvb] "v"(leaves_lds));                                                             \
        /* round 2: which of the node's 16 symbols */                                                      \
        RCX_QUAD_STAMP(0);                                                                                 \
        const RcxV4 l_ = *reinterpret_cast<const RcxLdsV4*>(la_);                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        /* behind the read: the remainder (round 1's other result) across the quad; cpprcoder.h:1134-1177, +1 on every \
           cumulative sum above the node -- the bounds whose subtraction borrowed in round 1 (bound above low <=> its \
           node number above the symbol's node; a target past the total leaves no borrow and raises none, as find()'s \
           fall-through needs it); the earlier symbol's byte; the next symbol's stream bytes */            \
        u32 sb_;                                                                                           \
        asm volatile("v_min3_u32 %[rm], %[x1], %[x2], %[x3]\n\t"                                           \
                     "v_min3_u32 %[rm], %[rm], %[x4], %[low]\n\t"                                          \
                     "v_addc_co_u32_e64 %[u1], %[c1], 0, %[u1], %[c1]\n\t"                                 \
                     "v_addc_co_u32_e64 %[u2], %[c2], 0, %[u2], %[c2]\n\t"                                 \
                     "v_min_u32_dpp %[rm], %[rm], %[rm] " RCX_QP1                                          \
                     "v_addc_co_u32_e64 %[u3], %[c3], 0, %[u3], %[c3]\n\t"                                 \
                     "v_addc_co_u32_e64 %[u4], %[c4], 0, %[u4], %[c4]\n\t"                                 \
                     "v_min_u32_dpp %[rm], %[rm], %[rm] " RCX_QP2                                          \
                     "v_lshl_add_u32 %[sb], %[n], 4, %[t0p3]" /* symbol, if none of the lane's bounds is above */ \
                     RCX_QD_PREV_S_##HP                                                                    \
                     : [u1] "+v"(U1), [u2] "+v"(U2), [u3] "+v"(U3), [u4] "+v"(U4_), [rm] "=&v"(rem_),       \
                       [sb] "=&v"(sb_), [c1] "+s"(c1_), [c2] "+s"(c2_), [c3] "+s"(c3_), [c4] "+s"(c4_),    \
                       [pye] "=&v"(pye_), [pword] "+v"(PWORD)                                              \
                     : [n] "v"(node_), [t0p3] "v"(T0p3), [low] "v"(in.low), [x1] "v"(x1_), [x2] "v"(x2_),  \
                       [x3] "v"(x3_), [x4] "v"(x4_), [pc] "s"(p_own_), [psym] "v"(p_sym_), [psh] "n"(PSHIFT)); \
        u32 lo_, rg_, qa_, qb_, qc_, qe_, tot_, pre_, o2_, d2_, ya_, yb_, yc_, ye_, hi_;                   \
        asm volatile("v_mul_u32_u24 %[qa], %[lx], %[t]\n\t"      /* the lane's four inclusive sums, scaled: every */ \
                     "v_mad_u32_u24 %[qb], %[ly], %[t], %[qa]\n\t" /* sum is below total x t <= range < 2^32 */      \
                     "v_mad_u32_u24 %[qc], %[lz], %[t], %[qb]\n\t"                                         \
                     "v_mad_u32_u24 %[qe], %[lw], %[t], %[qc]\n\t"                                         \
                     "v_alignbit_b32 %[n4], %[w1], %[w0], %[bp]\n\t" /* (the next symbol's 4 stream bytes at bp8 ... */ \
                     "v_perm_b32 %[n4], %[n4], %[n4], %[swap]\n\t" /* ... first one on top: here they separate qe from its use across the quad) */ \
                     "v_add_u32_dpp %[tot], %[qe], %[qe] " RCX_QP1 /* the lanes' sums are scanned scaled: (a+b)t = at+bt */ \
                     "v_and_b32_dpp %[pre], %[qe], %[m1] " RCX_QP1                                         \
                     "v_sub_u32 %[d2], %[rem], %[pre]\n\t"                                                 \
                     "v_and_b32_dpp %[o2], %[tot], %[m2] " RCX_QP2                                         \
                     "v_sub_u32 %[d2], %[d2], %[o2]\n\t"     /* rem - t x the counts of the node's symbols in lower lanes */ \
                     "v_sub_co_u32_e64 %[ya], %[c1], %[d2], %[qa]\n\t"                                     \
                     "v_sub_co_u32_e64 %[yb], %[c2], %[d2], %[qb]\n\t"                                     \
                     "v_sub_co_u32_e64 %[yc], %[c3], %[d2], %[qc]\n\t"                                     \
                     "v_sub_co_u32_e64 %[ye], %[own], %[d2], %[qe]\n\t" /* borrows in the lane that owns the symbol */ \
                     "v_min3_u32 %[lo], %[d2], %[ya], %[yb]\n\t"                                           \
                     "v_max3_u32 %[hi], %[ya], %[yb], %[yc]\n\t"                                           \
                     "v_min_u32 %[lo], %[lo], %[yc]\n\t"                                                   \
                     "v_max_u32 %[hi], %[hi], %[ye]\n\t"                                                   \
                     "v_subb_co_u32_e64 %[sym], %[c1], %[sb], 0, %[c1]\n\t"                                \
                     "v_min_u32_dpp %[lo], %[lo], %[lo] " RCX_QP1                                          \
                     "v_subb_co_u32_e64 %[sym], %[c2], %[sym], 0, %[c2]\n\t"                               \
                     "v_max_u32_dpp %[hi], %[hi], %[hi] " RCX_QP1                                          \
                     "v_subb_co_u32_e64 %[sym], %[c3], %[sym], 0, %[c3]\n\t"                               \
                     "v_min_u32_dpp %[lo], %[lo], %[lo] " RCX_QP2                                          \
                     "v_max_u32_dpp %[hi], %[hi], %[hi] " RCX_QP2                                          \
                     "v_sub_u32 %[rg], %[lo], %[hi]"                                                       \
                     : [lo] "=&v"(lo_), [rg] "=&v"(rg_), [sym] "=&v"(p_sym_), [own] "=&s"(p_own_), [n4] "=&v"(in.n4), \
                       [qa] "=&v"(qa_), [qb] "=&v"(qb_), [qc] "=&v"(qc_),                                   \
                       [qe] "=&v"(qe_), [tot] "=&v"(tot_), [pre] "=&v"(pre_), [o2] "=&v"(o2_), [d2] "=&v"(d2_), \
                       [ya] "=&v"(ya_), [yb] "=&v"(yb_), [yc] "=&v"(yc_), [ye] "=&v"(ye_), [hi] "=&v"(hi_),  \
                       [c1] "=&s"(c1_), [c2] "=&s"(c2_), [c3] "=&s"(c3_)                                   \
                     : [lx] "v"(l_.x), [ly] "v"(l_.y), [lz] "v"(l_.z), [lw] "v"(l_.w), [t] "v"(t_),        \
                       [rem] "v"(rem_), [m1] "v"(m1), [m2] "v"(m2), [sb] "v"(sb_),                         \
                       [w0] "v"(in.w0), [w1] "v"(in.w1), [bp] "v"(in.bp8), [swap] "s"(0x00010203u));       \
        RCX_QUAD_STAMP(1);                                                                                 \
        in.low = lo_;   /* :906 */                                                                         \
        in.range = rg_; /* :907 */                                                                         \
        p_la_ = la_;                                                                                       \
    }
// The byte and the count of the last symbol decoded (no symbol follows that would make them): into WORD at bit SHIFT.
#define RCX_QUAD_DEC_FINISH(WORD, SHIFT)                                                                    \
    {                                                                                                      \
        u32 pown_, pad_, pye_;                                                                             \
        asm volatile("v_cndmask_b32_e64 %[pown], 0, 1, %[pc]\n\t"                                          \
                     "v_and_b32 %[pad], 3, %[psym]\n\t"                                                    \
                     "v_cndmask_b32_e64 %[pye], 0, %[psym], %[pc]\n\t"                                     \
                     "v_lshl_add_u32 %[pad], %[pad], 2, %[pla]\n\t"                                        \
                     "v_lshl_or_b32 %[pword], %[pye], %[psh], %[pword]"                                    \
                     : [pown] "=&v"(pown_), [pad] "=&v"(pad_), [pye] "=&v"(pye_), [pword] "+v"(WORD)        \
                     : [pc] "s"(p_own_), [psym] "v"(p_sym_), [pla] "v"(p_la_), [psh] "n"(SHIFT));          \
        (void)__hip_atomic_fetch_add(reinterpret_cast<RcxLdsU32*>(pad_), pown_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); /* :916 */ \
    }

    // The divisors of the next 16 symbols are converted and written to LDS at every top-up, and the 16 after
    // those requested from the table right after the top-up's own loads: every s_waitcnt vmcnt in the loop
    // then waits for loads issued 16 symbols earlier, never for one that has just been issued.
    DivEntry ahead = divtab[lane % RCX_QUAD_STAGE];
#define RCX_QUAD_STAGE_PUT()                                                                               \
    {                                                                                                      \
        DivQ q_;                                                                                           \
        q_.mul = ahead.mul;                                                                                \
        q_.st = (ahead.total << 5) | ahead.shift;                                                          \
        q_.add = ahead.add;                                                                                \
        stage[lane % RCX_QUAD_STAGE] = q_; /* four lanes per entry, same value */                          \
    }
#define RCX_QUAD_STAGE_GET(I0) ahead = divtab[(I0) + RCX_QUAD_STAGE + lane % RCX_QUAD_STAGE];
    {
        // 64 decoded bytes leave as four back-to-back 16-byte stores, so that L2 sees whole 64-byte pieces
        // (16-byte pieces 16 symbols apart were written to HBM one by one: 4x WRITE_SIZE).  Groups 0..2 wait
        // in the block's scratch area, group 3 in registers, and the stores are issued right AFTER the next
        // top-up: its s_waitcnt vmcnt for the input pieces would otherwise wait for these stores as well.
        U4 o_last;
        o_last.x = o_last.y = o_last.z = o_last.w = 0;
#if defined(RCX_STAMP_DEC)
        unsigned long long stamp_sum_[4] = {0, 0, 0, 0};
#endif
        for (u32 i0 = 0; i0 < fast_end; i0 += 16) {
            RCX_QUAD_STAGE_PUT();
            in.topup();
            RCX_QUAD_STAGE_GET(i0);
            const u32 g = (i0 >> 4) & 3u;
            if (g == 0 && i0 != 0 && leader) {
                U4* o4 = reinterpret_cast<U4*>(out + (i0 - 64));
                const U4 p0 = parked[0], p1 = parked[1], p2 = parked[2];
                o4[0] = p0;
                o4[1] = p1;
                o4[2] = p2;
                o4[3] = o_last;
            }
            u32 w0_ = 0, w1_ = 0, w2_ = 0, w3_ = 0;
            DivQ k_next = stage[0];
#if defined(RCX_STAMP_DEC)
            unsigned long long stamp_t_[4] = {0, 0, 0, 0};
#define RCX_QUAD_STEP(S, HP, PW)                                 \
    {                                                            \
        const DivQ kk = k_next;                                  \
        if ((S) + 1 < 16) k_next = stage[(S) + 1];               \
        const bool stamp_now_ = (S) == 8 || (S) == 9;            \
        const int stamp_at_ = (S) == 8 ? 0 : 2;                  \
        RCX_QUAD_DEC_SYMBOL(kk, HP, PW, (8 * (((S) + 3) & 3))); \
    }
#else
#define RCX_QUAD_STEP(S, HP, PW)                                 \
    {                                                            \
        const DivQ kk = k_next;                                  \
        if ((S) + 1 < 16) k_next = stage[(S) + 1];               \
        RCX_QUAD_DEC_SYMBOL(kk, HP, PW, (8 * (((S) + 3) & 3))); \
    }
#endif
            // (a step makes the byte of the step before it: RCX_QUAD_DEC_SYMBOL)
            RCX_QUAD_STEP(0, 0, w0_) RCX_QUAD_STEP(1, 1, w0_) RCX_QUAD_STEP(2, 1, w0_) RCX_QUAD_STEP(3, 1, w0_)
            RCX_QUAD_STEP(4, 1, w0_) RCX_QUAD_STEP(5, 1, w1_) RCX_QUAD_STEP(6, 1, w1_) RCX_QUAD_STEP(7, 1, w1_)
            RCX_QUAD_STEP(8, 1, w1_) RCX_QUAD_STEP(9, 1, w2_) RCX_QUAD_STEP(10, 1, w2_) RCX_QUAD_STEP(11, 1, w2_)
            RCX_QUAD_STEP(12, 1, w2_) RCX_QUAD_STEP(13, 1, w3_) RCX_QUAD_STEP(14, 1, w3_) RCX_QUAD_STEP(15, 1, w3_)
            RCX_QUAD_DEC_FINISH(w3_, 24)
#undef RCX_QUAD_STEP
#if defined(RCX_STAMP_DEC)
            stamp_sum_[0] += stamp_t_[1] - stamp_t_[0]; // symbol 8: leaf read issued -> round 2 done
            stamp_sum_[1] += stamp_t_[2] - stamp_t_[1]; // -> leaf read of symbol 9 issued
            stamp_sum_[2] += stamp_t_[3] - stamp_t_[2]; // symbol 9: leaf read issued -> round 2 done
            stamp_sum_[3] += 1;
#endif
            U4 o;
            o.x = rcx_quad_or(w0_);
            o.y = rcx_quad_or(w1_);
            o.z = rcx_quad_or(w2_);
            o.w = rcx_quad_or(w3_);
            if (g == 3) o_last = o;
            else parked[g] = o; // the quad's 4 lanes store the same 16 bytes
        }
#if defined(RCX_STAMP_DEC)
        if (blockIdx.x == 7 && threadIdx.x == 0)
            for (int i_ = 0; i_ < 4; ++i_) rcx_dec_stamp_out[i_] = stamp_sum_[i_];
#endif
        if (leader && fast_end != 0) { // what is still parked: the last 16..64 bytes of the fast region
            const u32 groups = ((fast_end - 1) >> 4 & 3u) + 1;
            U4* o4 = reinterpret_cast<U4*>(out + ((fast_end - 1) & ~63u));
            o4[0] = parked[0];
            if (groups > 1) o4[1] = parked[1];
            if (groups > 2) o4[2] = parked[2];
            if (groups > 3) o4[3] = o_last;
        }
    }
    {
        for (u32 i = fast_end; i < maxlen; ++i) { // (fast_end is a multiple of 16: the top-ups stay 16 symbols apart)
            if ((i & 15u) == 0) {
                RCX_QUAD_STAGE_PUT();
                in.topup();
                RCX_QUAD_STAGE_GET(i);
            }
            const DivQ k = stage[i % RCX_QUAD_STAGE];
            if (i < len) { // the 4 lanes of a quad agree
                u32 part = 0;
#if defined(RCX_STAMP_DEC)
                const bool stamp_now_ = false;
                const int stamp_at_ = 0;
                unsigned long long stamp_t_[4];
#endif
                RCX_QUAD_DEC_SYMBOL(k, 0, part, 0);
                RCX_QUAD_DEC_FINISH(part, 0);
                part = rcx_quad_or(part);
                if (leader) out[i] = (u8)part;
            }
        }
    }
#undef RCX_QUAD_DEC_SYMBOL
#undef RCX_QUAD_DEC_FINISH
#undef RCX_QD_PREV_A_0
#undef RCX_QD_PREV_A_1
#undef RCX_QD_PREV_S_0
#undef RCX_QD_PREV_S_1
#undef RCX_QUAD_STAGE_PUT
#undef RCX_QUAD_STAGE_GET
#undef RCX_QP1
#undef RCX_QP2
    // A symbol past the table ("node 16") is the only one that raises none of the cumulative sums: the last
    // lane's U4 -- the total, 256 + the symbols decoded (cpprcoder.h:1096, :1138) -- then falls short.
    // A marked block is judged (truncated or not) by the kernel that decodes it again.
    const bool marked = live && rcx_quad_or(j == 3 && U4_ != 256u + len ? 1u : 0u) != 0;
    if (leader && !marked && in.taken() > stream_len) rcx_flag(status, RCX_ST_CORRUPT, blk);
    if (leader) redo[blk] = marked ? 1u : 0u;
    else if (j == 0 && in_use && blk < nblocks) redo[blk] = 0;
}

// ===========================================================================
// Encode, pass 1, five-wave split ("MC5"): as rcx_enc_mc_k, with the coder itself cut in two
// (EncLane::arith / EncLane::emit): the interval arithmetic (state low, range) and the byte
// writer (state acc / nacc8 / pos, the global stores) are separate waves connected by a second
// LDS ring of one record per symbol.  Three pipeline stages, one chunk apart:
//   waves M1..M3 (model, chunk k) -> wave A (arithmetic, chunk k-1) -> wave W (writer, chunk k-2)
// Five waves on four SIMDs: the two lightest (A and the level-1 model wave) are meant to share one.
//
// The writer does not store to global memory: scattered 4-byte stores under an EXEC mask were its most
// expensive step.  Its bytes go to a per-block ring in LDS (StagedWriter: a window of the newest eight bytes mirrored
// there, two words a symbol), and the level-1 model wave drains the rings once per chunk with 16-byte stores (the leaf
// wave or the writer itself measured no better: a drain costs its wave the same wherever it runs).
// The drain keeps the newest RCX_OUT_MARGIN bytes back, so that a carry that runs through more than the newest four
// bytes (cpprcoder.h:767-781) is resolved in LDS; a run of 0xFF bytes longer than that margin cannot be,
// and such a block is marked in `redo` and encoded again by rcx_enc_adaptive_k.
// Level 3 of the model (one group per block) lives in registers of the levels-3+2 wave, the other levels in LDS.
// ===========================================================================
// Which wave drains the output rings: 5 = a sixth wave that does nothing else (default), 4 = the level-1 model wave (shares
// its SIMD with the arithmetic wave), 3 = the leaf wave.  The kernel's wave ROLES are numbered 0 arithmetic, 1 writer,
// 2 model levels 3+2, 3 model leaf level, 4 model level 1, 5 drain; with six waves the hardware's waves 1 and 3 swap roles,
// so that (a workgroup's wave i runs on SIMD i mod 4) the drain shares a SIMD with the leaf wave, which has the most room.
#if !defined(RCX_DRAIN_WAVE)
#define RCX_DRAIN_WAVE 5
#endif
#if RCX_DRAIN_WAVE == 5
#define RCX_MC5_THREADS 384
#define RCX_MC5_ROLE(HW) ((HW) == 1u ? 3u : (HW) == 3u ? 1u : (HW))
#else
#define RCX_MC5_THREADS 320
#define RCX_MC5_ROLE(HW) (HW)
#endif
// The ring between the model waves and the arithmetic wave: four dwords per symbol and lane.  Kept as 16 contiguous
// bytes per lane (one ds_read_b128 for the arithmetic wave; the model waves' 4-byte stores hit each bank four times)
// or, RCX_RING_PLANAR=1, as four dword planes (conflict-free stores, two ds_read2st64_b32).
#if !defined(RCX_RING_PLANAR)
#define RCX_RING_PLANAR 0
#endif
#if !defined(RCX_MODEL_AHEAD)
#define RCX_MODEL_AHEAD 2 /* symbols the model waves' LDS reads and updates run ahead of their sums */
#endif
#if !defined(RCX_ARITH_AHEAD)
#define RCX_ARITH_AHEAD 1 /* symbols the arithmetic wave's ring and divisor reads run ahead */
#endif
#if RCX_RING_PLANAR
#define RCX_RING_LANE 1
#define RCX_RING_AT(s, f) ((4 * (s) + (f)) * RCX_LANES)
#else
#define RCX_RING_LANE 4
#define RCX_RING_AT(s, f) (4 * (s) * RCX_LANES + (f))
#endif
#define RCX_MC5_RING2_DW (2 * RCX_MC_CHUNK * RCX_LANES)
#define RCX_OUT_RING_WORDS 64 /* per block: 256 bytes of output waiting in LDS */
#define RCX_OUT_MARGIN 32     /* bytes kept back from the drain */
#define RCX_MC5_OUT_DW (RCX_OUT_RING_WORDS * RCX_LANES + 3 * RCX_LANES)
#define RCX_MC5_LDS_U4 (RCX_MC_LDS_U4 + RCX_MC5_RING2_DW / 4 + RCX_LANES / 4 + RCX_MC5_OUT_DW / 4)

struct __attribute__((packed, aligned(4))) RcxU4Unaligned {
    u32 x, y, z, w;
};

// `extra` carries ran off the bytes the writer holds in registers: add them into the bytes already in the
// ring, newest first (cpprcoder.h:767-781).  Returns 1 if the carry wants to go below `safe_from`, where the
// bytes may have left for global memory already.  Rare (about 9e-5 per symbol on random data): out of line.
__device__ __attribute__((noinline, cold)) u32 rcx_stage_carry(u32* ring_lane, u32 pos, u32 safe_from, u32 extra)
{
    while (extra != 0 && pos > safe_from) {
        --pos;
        u32* w = ring_lane + ((pos >> 2) % RCX_OUT_RING_WORDS) * RCX_LANES;
        const u32 sh = 8u * (pos & 3u); // the ring holds memory-order dwords
        const u32 old = *w;
        const u32 v = ((old >> sh) & 0xFFu) + extra;
        *w = (old & ~(0xFFu << sh)) | ((v & 0xFFu) << sh);
        extra = v >> 8;
    }
    return extra != 0 && pos != 0 ? 1u : 0u; // (a carry out of the very first byte cannot happen: it starts as 0)
}

// A carry ran through all of the newest four bytes (StagedWriter::emit; `acc` has it already): it goes on in the ring, byte
// by byte.  First the two newest words go to their OWN slots as they were (the newest may so far only be in slot 64);
// everything older is in the ring already, and current: a byte stops changing, these paths apart, once it is no longer
// among the newest four.  Out of line: practically never on random data (tests/carry_runs.py builds the inputs).
__device__ __attribute__((noinline, cold)) u32 rcx_stage_far_carry(u32* ring_lane, u64 acc_after, u32 pos8, u32 safe_from, u32 far)
{
    if (!far) return 0;
    const u64 before = acc_after - 1;
    const u32 sh8 = (0u - pos8) & 24u;
    const u64 t = before << sh8;
    const u32 w = (pos8 - 8u) >> 5; // the word of the newest byte
    ring_lane[((w - 1u) % RCX_OUT_RING_WORDS) * RCX_LANES] = rcx_bswap((u32)(t >> 32));
    ring_lane[(w % RCX_OUT_RING_WORDS) * RCX_LANES] = rcx_bswap((u32)t);
    return rcx_stage_carry(ring_lane, pos8 >> 3, safe_from, 1u);
}

// The drain of a block's output ring (both multi-wave encoders): `piece` = the four ring words at `drained`, read earlier;
// it leaves if it lies below `lim`.  The predicated store is written out -- the compiler's version of
// `if (...) store` around three conditional pieces was forty instructions of execution-mask bookkeeping a chunk, on the
// wave that shares its SIMD with the arithmetic wave -- and a block that is more than one piece behind (a chunk makes 48
// bytes at most) goes out of line, where a loop may be a loop (in front of one the compiler waits for every store in flight).
__device__ __attribute__((noinline, cold)) u32 rcx_drain_more(const u32* ring_lane, u8* payload, u32 drained, u32 lim, bool live)
{
    while (__any(live && drained + 16 <= lim)) {
        if (live && drained + 16 <= lim) {
            const u32* w = ring_lane + ((drained >> 2) % RCX_OUT_RING_WORDS) * RCX_LANES; // (drained is a multiple of 16)
            RcxU4Unaligned piece;
            piece.x = w[0];
            piece.y = w[RCX_LANES];
            piece.z = w[2 * RCX_LANES];
            piece.w = w[3 * RCX_LANES];
            *reinterpret_cast<RcxU4Unaligned*>(payload + drained) = piece;
            drained += 16;
        }
    }
    return drained;
}
__device__ __forceinline__ u32 rcx_drain_piece(const u32* ring_lane, u8* payload, u32 drained, u32 lim, bool live, const RcxU4Unaligned& piece)
{
    const bool go = live && drained + 16 <= lim;
    {
        const u64 lanes = __ballot(go);
        u8* at = payload + drained;
        RcxV4 data;
        data.x = piece.x, data.y = piece.y, data.z = piece.z, data.w = piece.w;
        u64 saved;
        asm volatile("s_and_saveexec_b64 %[sv], %[go]\n\t"
                     "global_store_dwordx4 %[at], %[data], off\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [sv] "=&s"(saved)
                     : [go] "s"(lanes), [at] "v"(at), [data] "v"(data)
                     : "memory");
    }
    drained += go ? 16u : 0u;
    if (rcx_any(live && drained + 16 <= lim)) drained = rcx_drain_more(ring_lane, payload, drained, lim, live);
    return drained;
}

// The byte writer of the multi-wave encoders.  EncLane::emit gathers bytes in a register and lets four of them go when
// it holds five or more; that is a dozen selects a symbol.  Here the register is a WINDOW -- the newest eight bytes of
// the payload, newest lowest -- and the block's ring in LDS MIRRORS it: every symbol writes the two aligned words that hold
// the newest five to eight bytes (one ds_write2st64_b32: consecutive words of a block are 64 dwords apart), whether
// they are complete or not.  A carry (cpprcoder.h:767-781) is a 64-bit add on the window, made BEFORE the symbol's words
// are written, so whatever it changes among the newest four bytes is simply written again; only a carry that runs
// through all four of them -- the low half of the add overflows, which costs no instruction to notice -- has to go on
// in the ring itself (rcx_stage_carry; practically never on random data, adversarial inputs: tests/carry_runs.py).
// Ring slot 64 repeats slot 0 for the pair (63, 64): a word that lands there is the newest one, and it is written to
// its own slot as the older word of the next pair before anything reads it (the drain keeps RCX_OUT_MARGIN bytes back,
// finish() takes the newest word from the window).
struct StagedWriter {
    u64 acc;        // the newest 8 bytes of the payload as a number, newest byte lowest (before the stream: zeroes)
    u32 pos8;       // 8 x the payload bytes produced so far (the reference's initial buffer_ = 0 is the first: EncLane)
    u32 pos;        // pos8 / 8 as of the last chunk_begins() / chunk_ends()
    u32 safe_from;  // bytes below this may have been drained (pos at the start of the chunk - RCX_OUT_MARGIN)
    u32 redo;
    u32* ring_lane; // word w of this block: ring_lane[(w % RCX_OUT_RING_WORDS) * RCX_LANES]; slot RCX_OUT_RING_WORDS: see above

    __device__ __forceinline__ void begin(u32* ring, u32* /*slot 64 follows the ring*/, u32 lane)
    {
        acc = 0;
        pos8 = 8;
        pos = 1;
        safe_from = 0;
        redo = 0;
        ring_lane = ring + lane;
    }
    __device__ __forceinline__ void chunk_begins()
    {
        pos = pos8 >> 3;
        safe_from = pos > RCX_OUT_MARGIN ? pos - RCX_OUT_MARGIN : 0u;
    }
    __device__ __forceinline__ u32 chunk_ends()
    {
        pos = pos8 >> 3;
        return pos;
    }
    // the two aligned words that hold the newest 5..8 bytes, from the window
    __device__ __forceinline__ void mirror()
    {
        const u32 sh8 = (0u - pos8) & 24u;            // the window's end moved up to a word boundary
        const u32 s0 = ((pos8 - 40u) >> 5) % RCX_OUT_RING_WORDS; // slot of the older word: ((pos - 1) / 4 - 1) mod 64
        mirror_at(sh8, (u32)reinterpret_cast<uintptr_t>(ring_lane + s0 * RCX_LANES));
    }
    __device__ __forceinline__ void mirror_at(u32 sh8, u32 at_lds) // (an LDS address as a number: it passes through an asm statement)
    {
        const u64 t = acc << sh8;
        RcxLdsU32* at = reinterpret_cast<RcxLdsU32*>(at_lds);
        at[0] = rcx_bswap((u32)(t >> 32));
        at[RCX_LANES] = rcx_bswap((u32)t);
    }
    __device__ __forceinline__ void emit(u32 rec)
    {
        const u32 lo0 = (u32)acc;
        const u32 lo1 = lo0 + (rec & 1u);                  // cpprcoder.h:767-781
        const bool far = lo1 < lo0;                        // ... through all of the newest four bytes
        // (where the words go depends on the position alone: worked out between the two halves of the add, whose second
        // half may not follow the first at once)
        u32 sh8 = (0u - pos8) & 24u;
        u32 at = (u32)reinterpret_cast<uintptr_t>(ring_lane + (((pos8 - 40u) >> 5) % RCX_OUT_RING_WORDS) * RCX_LANES);
        asm volatile("" : "+v"(sh8), "+v"(at));
        acc = ((u64)((u32)(acc >> 32) + (far ? 1u : 0u)) << 32) | lo1;
        if (rcx_any(far)) redo |= rcx_stage_far_carry(ring_lane, acc, pos8, safe_from, far ? 1u : 0u);
        mirror_at(sh8, at);
        const u32 k8 = rec & 0x18u;
        acc = (acc << k8) | __builtin_amdgcn_ubfe(rec, 32u - k8, k8); // the k8 / 8 bytes that leave through the top of low
        pos8 += k8;
    }
    // After the last symbol: the words below the newest one are in the ring (returns how many bytes that is); the newest
    // 1..4 bytes are handed over as EncLane's held bytes.
    __device__ __forceinline__ u32 finish(EncLane& enc)
    {
        mirror();
        pos = pos8 >> 3;
        const u32 flushed = ((pos - 1u) >> 2) << 2;
        enc.nacc8 = 8u * (pos - flushed);
        enc.acc = acc & ((1ull << enc.nacc8) - 1ull);
        enc.pos = flushed;
        return flushed;
    }
};

template <bool FULL>
__device__ __forceinline__ void rcx_mc5_pipeline(u32 wave, u32 lane, u32 len, u32 nchunks, const u8* in,
                                                 const DivEntry* __restrict__ divtab, const Tree& tree, DivEntry* stage,
                                                 u32* ring, u32* ring2, EncLane& enc, DivEntry& ahead, StagedWriter& wr,
                                                 u32* out_pos, u32& drained, u8* payload, u32 cap, bool live)
{
    // wave roles: 0 arithmetic, 1 writer, 2 model levels 3+2, 3 model leaf level, 4 model level 1 + drain
#if defined(RCX_STAMP)
    unsigned long long stamp_wait_ = 0;
    const unsigned long long stamp_begin_ = __builtin_amdgcn_s_memtime();
#endif
    U4 piece_ahead;
    piece_ahead.x = piece_ahead.y = piece_ahead.z = piece_ahead.w = 0;
    if (FULL && wave >= 2 && wave <= 4 && nchunks > 0) piece_ahead = *reinterpret_cast<const U4*>(in);
    u32 l3a = 64, l3b = 128, l3c = 192; // wave 2: the level-3 sums (cpprcoder.h:1094-1132: every count 1)
    for (u32 k = 0; k <= nchunks + 1; ++k) {
        if (wave == 0) {
            // ---- arithmetic: chunk k-1, records into ring2[(k-1)&1] ----
            if (k >= 1 && k <= nchunks) {
                const u32 i0 = (k - 1) * RCX_MC_CHUNK;
                if ((i0 % RCX_STAGE) == 0) {
                    // staged with the 64-bit addend of the multiply-add as a register pair (see DivQ)
                    DivQ q;
                    q.mul = ahead.mul;
                    q.st = ahead.shift;
                    q.add = ahead.add;
                    reinterpret_cast<DivQ*>(stage)[lane] = q;
                    ahead = divtab[i0 + RCX_STAGE + lane];
                }
                // the model waves' answers: field f of symbol s of lane l at dword RCX_RING_AT(s, f) + RCX_RING_LANE * l
                const u32* rs = ring + ((k - 1) & 1u) * (4 * RCX_MC_CHUNK * RCX_LANES) + RCX_RING_LANE * lane;
                u32* ws2 = ring2 + ((k - 1) & 1u) * (RCX_MC_CHUNK * RCX_LANES) + lane;
                // the chunk's divisors through one vector base register and immediate offsets (a wave-uniform
                // address would be rebuilt in a scalar register and moved over for every read)
                u32 st_lds = (u32)reinterpret_cast<uintptr_t>(stage + (i0 % RCX_STAGE));
                asm volatile("" : "+v"(st_lds));
                const RcxLdsDivQ* st = reinterpret_cast<const RcxLdsDivQ*>(st_lds);
                U4 eq[RCX_MC_CHUNK];
                RcxDivQv kq[RCX_MC_CHUNK];
                u32 rec_even = 0;
#define RCX_A_ISSUE(T)                                                                                              \
    {                                                                                                               \
        eq[T].x = rs[RCX_RING_AT((T), 0)], eq[T].y = rs[RCX_RING_AT((T), 1)], eq[T].z = rs[RCX_RING_AT((T), 2)];    \
        eq[T].w = rs[RCX_RING_AT((T), 3)];                                                                          \
        kq[T] = st[T];                                                                                              \
    }
#pragma unroll
                for (u32 t = 0; t < RCX_ARITH_AHEAD; ++t) RCX_A_ISSUE(t);
#pragma unroll
                for (u32 s = 0; s < RCX_MC_CHUNK; ++s) {
                    if (s + RCX_ARITH_AHEAD < RCX_MC_CHUNK) RCX_A_ISSUE(s + RCX_ARITH_AHEAD);
                    const U4 e = eq[s];
                    const RcxDivQv kk = kq[s];
                    u32 rec = 0; // past the end of a short block: a record that does nothing
                    if (FULL || i0 + s < len) rec = enc.arith_q(e.x + e.y + e.z, e.w, kk.x, kk.y, ((u64)kk.w << 32) | kk.z);
                    // two symbols' records leave as one ds_write2st64_b32 (consecutive symbols are 64 dwords apart): an LDS
                    // instruction costs a lone wave 12-16 cycles of issue whatever it carries (tools/diag/ubench.hip k_t_*)
                    if ((s & 1u) == 0) rec_even = rec;
                    else {
                        ws2[(s - 1) * RCX_LANES] = rec_even;
                        ws2[s * RCX_LANES] = rec;
                    }
                }
#undef RCX_A_ISSUE
            }
        } else if (wave == 1) {
            // ---- writer: chunk k-2 from ring2[(k-2)&1] ----
            if (k >= 2) {
                const u32* rs2 = ring2 + ((k - 2) & 1u) * (RCX_MC_CHUNK * RCX_LANES) + lane;
                u32 ra_next = rs2[0], rb_next = rs2[RCX_LANES]; // (pairs: one ds_read2st64_b32)
                wr.chunk_begins();
#pragma unroll
                for (u32 s = 0; s < RCX_MC_CHUNK; s += 2) {
                    const u32 ra = ra_next, rb = rb_next;
                    if (s + 2 < RCX_MC_CHUNK) ra_next = rs2[(s + 2) * RCX_LANES], rb_next = rs2[(s + 3) * RCX_LANES];
                    wr.emit(ra);
                    wr.emit(rb);
                }
                out_pos[lane] = wr.chunk_ends(); // for the drain of the next iteration
            }
        } else {
          // ---- drain: whole 16-byte pieces below (the writer's position one barrier ago - margin) ----
          // Asked for here, stored behind the chunk's model work: the writer's position and -- before it is known whether they
          // may leave -- the next four words of the block's ring.  (Read and stored in one go, the five LDS reads' latency
          // was this wave's, which shares its SIMD with the arithmetic wave: 28 cycles a symbol.)
          u32 drain_p = 0;
          RcxU4Unaligned drain_piece;
          drain_piece.x = drain_piece.y = drain_piece.z = drain_piece.w = 0;
          if (wave == RCX_DRAIN_WAVE) {
            drain_p = out_pos[lane];
            const u32* w = wr.ring_lane + ((drained >> 2) % RCX_OUT_RING_WORDS) * RCX_LANES; // (drained is a multiple of 16: no wrap inside the piece)
            drain_piece.x = w[0];
            drain_piece.y = w[RCX_LANES];
            drain_piece.z = w[2 * RCX_LANES];
            drain_piece.w = w[3 * RCX_LANES];
          }
          // (up to three pieces a chunk: 16 symbols make at most 48 bytes.  No loop: in front of a loop the compiler waits for
          // every store in flight, and a store's round trip is a quarter of a chunk's time.)
          auto drain_store = [&]() {
            const u32 limit = drain_p > RCX_OUT_MARGIN ? (drain_p - RCX_OUT_MARGIN) & ~15u : 0u;
            drained = rcx_drain_piece(wr.ring_lane, payload, drained, limit < cap ? limit : cap, live, drain_piece);
          };
          if ((k >= nchunks || RCX_DRAIN_WAVE == 5) && wave == RCX_DRAIN_WAVE) drain_store(); // (a wave that only drains: nothing to wait for)
          if (k < nchunks && wave <= 4) {
            // ---- model: chunk k ----
            const u32 i0 = k * RCX_MC_CHUNK;
            u32* ws = ring + (k & 1u) * (4 * RCX_MC_CHUNK * RCX_LANES) + RCX_RING_LANE * lane;
            U4 piece;
            if (FULL) piece = piece_ahead;
            if (RCX_DRAIN_WAVE != 5 && wave == RCX_DRAIN_WAVE) {
                // The stores go out between the wait for this chunk's input (asked for a chunk ago: it is there) and the
                // request for the next chunk's: memory operations complete in order as far as s_waitcnt vmcnt can tell, so
                // a wait for input behind a store just issued would wait for that store (that was 28 cycles a symbol on
                // the SIMD this wave shares with the arithmetic wave).
                if (FULL) asm volatile("" ::"v"(piece.x), "v"(piece.y), "v"(piece.z), "v"(piece.w));
                drain_store();
            }
            if (FULL) {
                if (k + 1 < nchunks) piece_ahead = *reinterpret_cast<const U4*>(in + i0 + RCX_MC_CHUNK);
            } else if (i0 + RCX_MC_CHUNK <= len && (reinterpret_cast<uintptr_t>(in) & 15u) == 0) {
                piece = *reinterpret_cast<const U4*>(in + i0); // a whole, aligned chunk of a ragged block (or of a single stream)
            } else {
                u32 w[4] = {0, 0, 0, 0};
                for (u32 s = 0; s < RCX_MC_CHUNK; ++s)
                    if (i0 + s < len) w[s >> 2] |= (u32)in[i0 + s] << (8 * (s & 3));
                piece.x = w[0];
                piece.y = w[1];
                piece.z = w[2];
                piece.w = w[3];
            }
            // Software pipeline, RCX_MODEL_AHEAD symbols deep: the group reads AND the ds_add updates of symbol
            // s + AHEAD are issued before the sums of symbol s are formed.  LDS executes a wave's operations in order,
            // so the reads of a later symbol still see the updates of the earlier ones, and their latency -- 60 cycles
            // and more with five waves on the LDS unit, i.e. more than one symbol of a light wave -- hides behind the
            // arithmetic of the symbols in between (the updates need only the symbol, not the read data).
            U4 ga[RCX_MC_CHUNK], gb[RCX_MC_CHUNK]; // (indices are compile-time constants: registers)
            u32 held = 0, held_f = 0;
            if (wave == 2) {
#define RCX_M2_ISSUE(T)                                                       \
    {                                                                         \
        const u32 c_ = rcx_byte_of(piece, (T));                               \
        gb[T] = tree.group(RCX_G_L2 + (c_ >> 6));                             \
        if (FULL || i0 + (T) < len) tree.bump(RCX_G_L2 + (c_ >> 6), (c_ >> 4) & 3); \
    }
#pragma unroll
                for (u32 t = 0; t < RCX_MODEL_AHEAD; ++t) RCX_M2_ISSUE(t);
#pragma unroll
                for (u32 s = 0; s < RCX_MC_CHUNK; ++s) {
                    if (s + RCX_MODEL_AHEAD < RCX_MC_CHUNK) RCX_M2_ISSUE(s + RCX_MODEL_AHEAD);
                    const u32 cc = rcx_byte_of(piece, s);
                    // Level 3 -- one group per block -- lives in registers as three prefix sums (symbols below 64, 128,
                    // 192 so far): three compares serve both the select and the update (an LDS read, a ds_add and the
                    // masked sum cost this wave, the kernel's busiest, 20 cycles a symbol more).
                    u32 cum3;
                    if (FULL) {
                        u64 m1_, m2_, m3_, cz_;
                        asm volatile("v_cmp_gt_u32_e64 %[m3], %[k192], %[c]\n\t" /* (no literals in this encoding: 192, 128 from registers) */
                                     "v_cmp_gt_u32_e64 %[m2], %[k128], %[c]\n\t"
                                     "v_cmp_gt_u32_e64 %[m1], 64, %[c]\n\t"
                                     "v_cndmask_b32_e64 %[x], %[pc], %[pb], %[m3]\n\t"
                                     "v_cndmask_b32_e64 %[x], %[x], %[pa], %[m2]\n\t"
                                     "v_cndmask_b32_e64 %[x], %[x], 0, %[m1]\n\t"
                                     "v_addc_co_u32_e64 %[pc], %[cz], %[pc], 0, %[m3]\n\t"
                                     "v_addc_co_u32_e64 %[pb], %[cz], %[pb], 0, %[m2]\n\t"
                                     "v_addc_co_u32_e64 %[pa], %[cz], %[pa], 0, %[m1]"
                                     : [x] "=&v"(cum3), [pa] "+v"(l3a), [pb] "+v"(l3b), [pc] "+v"(l3c), [m1] "=&s"(m1_), [m2] "=&s"(m2_),
                                       [m3] "=&s"(m3_), [cz] "=&s"(cz_)
                                     : [c] "v"(cc), [k192] "s"(192u), [k128] "s"(128u));
                    } else {
                        cum3 = cc < 64u ? 0u : (cc < 128u ? l3a : (cc < 192u ? l3b : l3c));
                        if (i0 + s < len) l3a += cc < 64u ? 1u : 0u, l3b += cc < 128u ? 1u : 0u, l3c += cc < 192u ? 1u : 0u;
                    }
                    const u32 sum32 = cum3 + rcx_pre4(gb[s], (cc >> 4) & 3);
                    if (FULL) { // (pairs: one ds_write2st64_b32)
                        if ((s & 1u) == 0) held = sum32;
                        else ws[RCX_RING_AT(s - 1, 0)] = held, ws[RCX_RING_AT(s, 0)] = sum32;
                    } else if (i0 + s < len) ws[RCX_RING_AT(s, 0)] = sum32;
                }
#undef RCX_M2_ISSUE
            } else if (wave == 4) {
#define RCX_M1_ISSUE(T)                                                       \
    {                                                                         \
        const u32 c_ = rcx_byte_of(piece, (T));                               \
        ga[T] = tree.group(RCX_G_L1 + (c_ >> 4));                             \
        if (FULL || i0 + (T) < len) tree.bump(RCX_G_L1 + (c_ >> 4), (c_ >> 2) & 3); \
    }
#pragma unroll
                for (u32 t = 0; t < RCX_MODEL_AHEAD; ++t) RCX_M1_ISSUE(t);
#pragma unroll
                for (u32 s = 0; s < RCX_MC_CHUNK; ++s) {
                    if (s + RCX_MODEL_AHEAD < RCX_MC_CHUNK) RCX_M1_ISSUE(s + RCX_MODEL_AHEAD);
                    const u32 cc = rcx_byte_of(piece, s);
                    const u32 sum1 = rcx_pre4(ga[s], (cc >> 2) & 3);
                    if (FULL) { // (pairs: one ds_write2st64_b32)
                        if ((s & 1u) == 0) held = sum1;
                        else ws[RCX_RING_AT(s - 1, 1)] = held, ws[RCX_RING_AT(s, 1)] = sum1;
                    } else if (i0 + s < len) ws[RCX_RING_AT(s, 1)] = sum1;
                }
#undef RCX_M1_ISSUE
            } else {
#define RCX_M0_ISSUE(T)                                                       \
    {                                                                         \
        const u32 c_ = rcx_byte_of(piece, (T));                               \
        ga[T] = tree.group(RCX_G_L0 + (c_ >> 2));                             \
        if (FULL || i0 + (T) < len) tree.bump(RCX_G_L0 + (c_ >> 2), c_ & 3);  \
    }
#pragma unroll
                for (u32 t = 0; t < RCX_MODEL_AHEAD; ++t) RCX_M0_ISSUE(t);
#pragma unroll
                for (u32 s = 0; s < RCX_MC_CHUNK; ++s) {
                    if (s + RCX_MODEL_AHEAD < RCX_MC_CHUNK) RCX_M0_ISSUE(s + RCX_MODEL_AHEAD);
                    const u32 cc = rcx_byte_of(piece, s);
                    const u32 sum0 = rcx_pre4(ga[s], cc & 3), f0 = rcx_sel4(ga[s], cc & 3);
                    if (FULL) { // (pairs: two ds_write2st64_b32 for two symbols)
                        if ((s & 1u) == 0) held = sum0, held_f = f0;
                        else {
                            ws[RCX_RING_AT(s - 1, 2)] = held, ws[RCX_RING_AT(s, 2)] = sum0;
                            ws[RCX_RING_AT(s - 1, 3)] = held_f, ws[RCX_RING_AT(s, 3)] = f0;
                        }
                    } else if (i0 + s < len) {
                        ws[RCX_RING_AT(s, 2)] = sum0;
                        ws[RCX_RING_AT(s, 3)] = f0;
                    }
                }
#undef RCX_M0_ISSUE
            }
          }
        }
#if defined(RCX_STAMP)
        const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
        rcx_lds_barrier();
        stamp_wait_ += __builtin_amdgcn_s_memtime() - t0_;
#else
        rcx_lds_barrier();
#endif
    }
#if defined(RCX_STAMP)
    if (blockIdx.x == 7 && lane == 0) {
        rcx_stamp_out[wave * 2] = __builtin_amdgcn_s_memtime() - stamp_begin_;
        rcx_stamp_out[wave * 2 + 1] = stamp_wait_;
    }
#endif
}

__global__ __launch_bounds__(RCX_MC5_THREADS) void rcx_enc_mc5_k(const u8* __restrict__ src, u64 n, u32 block, u64 nblocks,
                                                                u8* __restrict__ slots, u64 slot, u32* __restrict__ sizes,
                                                                const DivEntry* __restrict__ divtab, u32* status,
                                                                u32* __restrict__ redo, u32 lanes_used)
{
    __shared__ U4 lds[RCX_MC5_LDS_U4];
    const u32 lane = threadIdx.x & 63u;
    const u32 wave = RCX_MC5_ROLE(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)); // the wave's ROLE (see RCX_DRAIN_WAVE)
    // lanes_used (1..64) of the 64 lanes carry a block; the others idle along (rcx_api.hip picks it from the
    // block count so that every CU has a workgroup before any workgroup carries 64 blocks)
    const bool in_use = lane < lanes_used;
    const u64 blk = in_use ? (u64)blockIdx.x * lanes_used + lane : nblocks;
    const bool live = blk < nblocks;
    const u64 at = live ? blk * (u64)block : 0;
    const u32 len = live ? (u32)((n - at) < (u64)block ? (n - at) : (u64)block) : 0u;

    Tree tree{reinterpret_cast<u32*>(lds) + (RCX_TREE_PLANAR ? 1 : 4) * lane};
    DivEntry* stage = reinterpret_cast<DivEntry*>(lds + RCX_GROUPS * RCX_LANES);
    u32* ring = reinterpret_cast<u32*>(lds + RCX_LDS_U4);
    u32* ring2 = reinterpret_cast<u32*>(lds + RCX_MC_LDS_U4);
    u32* final_low = ring2 + RCX_MC5_RING2_DW; // 64 dwords: the arithmetic wave's last low, for the writer's finish()
    u32* out_ring = final_low + RCX_LANES;     // the writer's words on their way to global memory
    u32* out_dummy = out_ring + RCX_OUT_RING_WORDS * RCX_LANES; // (ring slot 64: repeats slot 0 for the writer's pair (63, 64))
    u32* out_pos = out_dummy + RCX_LANES;      // writer -> drain: bytes in the ring so far
    u32* out_drained = out_pos + RCX_LANES;    // drain -> writer's finish: bytes stored so far

    const u32 maxlen = rcx_wave_max(len);
    const bool full = __all(!in_use || (live && len == block)) && (block % 16u == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    const u8* in = src + at; // (a lane without a block reads the first block along)
    const u32 nchunks = (maxlen + RCX_MC_CHUNK - 1) / RCX_MC_CHUNK;

    EncLane enc;
    u8* wave_slots = slots + (u64)blockIdx.x * lanes_used * slot;
    enc.idle(wave_slots); // wave 0 uses low/range; wave 1 takes over for finish()
    StagedWriter wr;
    wr.begin(out_ring, out_dummy, lane);
    u32 drained = 0;
    DivEntry ahead;
    ahead.mul = ahead.add = ahead.shift = ahead.total = 0;
    U4 v;
    if (wave == 0) {
        ahead = divtab[lane];
    } else if (wave == 1) {
        if (live) enc.begin(wave_slots, lane * (u32)slot, (u32)slot, len);
        out_pos[lane] = 0;
    } else if (wave == 2) { // cpprcoder.h:1094-1132: every count 1
        v.x = v.y = v.z = v.w = 16;
        for (u32 g = RCX_G_L2; g < RCX_G_L1; ++g) tree.store(g, v);
    } else if (wave == 4) {
        v.x = v.y = v.z = v.w = 4;
        for (u32 g = RCX_G_L1; g < RCX_G_L0; ++g) tree.store(g, v);
    } else if (wave == 3) {
        v.x = v.y = v.z = v.w = 1;
        for (u32 g = RCX_G_L0; g < RCX_GROUPS; ++g) tree.store(g, v);
    }
    rcx_lds_barrier();

    u8* payload = wave_slots + (u64)lane * slot + 4;
    const u32 cap = ((u32)slot - 4) & ~3u; // as EncLane::begin
    if (full) rcx_mc5_pipeline<true>(wave, lane, len, nchunks, in, divtab, tree, stage, ring, ring2, enc, ahead, wr, out_pos, drained, payload, cap, live);
    else rcx_mc5_pipeline<false>(wave, lane, len, nchunks, in, divtab, tree, stage, ring, ring2, enc, ahead, wr, out_pos, drained, payload, cap, live);

    if (wave == 0) final_low[lane] = enc.low;
    if (wave == RCX_DRAIN_WAVE) {
        out_drained[lane] = drained;
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the drained pieces are in memory before wave 1 may read them
    }
    rcx_lds_barrier();
    if (wave == 1 && live) {
        // what is still in the ring, then cpprcoder.h:744-762 as in the one-wave coder
        u32 at = out_drained[lane];
        const u32 flushed = wr.finish(enc);
        const u32 end = flushed < cap ? flushed : cap;
        for (; at < end; at += 4) *reinterpret_cast<u32*>(payload + at) = wr.ring_lane[((at >> 2) % RCX_OUT_RING_WORDS) * RCX_LANES];
        enc.low = final_low[lane];
        const u32 bytes = enc.finish();
        sizes[blk] = enc.overflow ? (u32)slot : bytes;
        if (enc.overflow) rcx_flag(status, RCX_ST_CAPACITY, blk);
        redo[blk] = (wr.redo != 0 && !enc.overflow) ? 1u : 0u;
    } else if (wave == 1 && blk < nblocks) {
        redo[blk] = 0;
    }
}
