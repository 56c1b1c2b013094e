// rcx_rans.hpp -- the reference's rANS coders (cppans.h) as many-block gfx950 kernels: SURVEY.md section 8(f)-4.
//
// Two stream formats, both [u32 LE n][257 x u32 LE scaled cumulative counts][payload] per block:
//   RCX_CODER_RANS   rANS::encode / decode            cppans.h:497-563   one 32-bit state, 14-bit probabilities, byte
//                                                     renormalisation; payload = [u32 LE final state][bytes ...]
//   RCX_CODER_RANS8  rANS::encode_simd / decode_simd  cppans.h:567-649   eight interleaved states (symbol i belongs to
//                                                     state i & 7), 12-bit probabilities, 16-bit renormalisation;
//                                                     payload = [8 x u32 LE states][u16 LE words ...]
// Both encoders run over the block backwards and write backwards from the END of the block's scratch slot (the
// reference writes backwards from the end of its destination, test/main.cpp:384-387); the compacting scatter then
// takes each stream from where it starts (`starts[]`).
//
// Mapping.  The eight-state format: a block is an octet of lanes, lane j IS state j, a wave is 8 blocks -- unlike the
// range coders, a block is 8 chains here, not one.  The one-state format is one chain per block again: its model is
// still built by an octet (histogram, scaling), its coding loops run one lane per block (encode) and four lanes per
// block (decode), see the second half of this file; a single stream (rcx_stream_encode) is coded by one octet whose
// lanes all carry the state.
//
// Included at the end of rcx_kernels.hpp.
#pragma once
#include "rcx_divtab.hpp"

#define RCX_RANS_HEADER 1032u /* 258 dwords: cppans.h:521, :598 */
#define RCX_RANS_BLOCKS 8     /* blocks per wave */

// LDS of one block while encoding: cum[257] (kept for the header) | table[256] = start | freq << 16 | 64 staged input bytes |
// 256 bytes of output on its way to memory + 16 spare bytes (the eight-state encoder); a multiple of 16 bytes
#define RCX_RANS_ENC_LDS_DW (264 + 256 + 16 + 64 + 4)
// (the decoders' LDS: see rcx_dec_rans8_k and rcx_dec_rans1_quad_k)

// Lanes of one octet talk through LDS without a barrier: a wave's LDS operations execute in order.  This keeps the
// compiler from moving or caching LDS accesses across the hand-over (it emits no instruction).
__device__ __forceinline__ void rcx_octet_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// the 8 bits of a wave-wide ballot that belong to this lane's octet (the lanes of an octet always branch together)
__device__ __forceinline__ u32 rcx_octet_ballot(bool p, u32 lane)
{
    const u64 all = __ballot(p);
    return (u32)(all >> (lane & ~7u)) & 0xFFu;
}
__device__ __forceinline__ u32 rcx_octet_min(u32 x)
{
    x = rcx_umin(x, rcx_dpp<0xB1>(x));  // quad_perm [1,0,3,2]
    x = rcx_umin(x, rcx_dpp<0x4E>(x));  // quad_perm [2,3,0,1]
    x = rcx_umin(x, rcx_dpp<0x141>(x)); // row_half_mirror
    return x;
}
// exact x / d for x < 2^21 * d (what the encoders need: a renormalised state is below x_max = d << 17 or d << 20):
// f32 estimate of a quotient below 2^21 is off by less than 1, one correction step settles it
__device__ __forceinline__ u32 rcx_div_small_quotient(u32 x, u32 d, u32& rem)
{
    u32 q = (u32)((float)x * rcx_rcp((float)d));
    u32 r = x - q * d;
    if ((s32)r < 0) {
        q -= 1;
        r += d;
    } else if (r >= d) {
        q += 1;
        r -= d;
    }
    rem = r;
    return q;
}

// ---------------------------------------------------------------------------------------------------------
// The model of one block, by its octet: count (cppans.h:102-128), cumulative (:130-136), normalize (:138-178).
// On return cum[0..256] holds the scaled cumulative counts and table[s] = cum[s] | (cum[s+1] - cum[s]) << 16.
// ---------------------------------------------------------------------------------------------------------
template <u32 PROB_BITS>
__device__ __forceinline__ void rcx_rans_model(u32* cum, u32* table, const u8* in, u32 len, u32 j, bool aligned16)
{
    // count(): into table[] for now
    for (u32 i = j; i < 256; i += 8) table[i] = 0;
    u32 at = 0;
    if (aligned16) {
        for (; at + 128 <= len; at += 128) { // 16 bytes per lane
            const U4 piece = *reinterpret_cast<const U4*>(in + at + 16 * j);
#pragma unroll
            for (u32 s = 0; s < 16; ++s) rcx_lds_inc(table + rcx_byte_of(piece, s));
        }
    }
    for (u32 i = at + j; i < len; i += 8) rcx_lds_inc(table + in[i]);
    rcx_octet_sync();
    // cumulative(): each lane sums its 32 counts, the octet scans, each lane writes its 32 entries
    u32 mine = 0;
    for (u32 i = 0; i < 32; ++i) mine += table[32 * j + i];
    u32 run = rcx_oct_excl_scan(mine, (j & 1u) ? ~0u : 0u, (j & 2u) ? ~0u : 0u, (j & 4u) ? ~0u : 0u);
    for (u32 i = 0; i < 32; ++i) {
        cum[32 * j + i] = run;
        run += table[32 * j + i];
    }
    if (j == 7) cum[256] = run; // = len
    rcx_octet_sync();
    // normalize(), cppans.h:140-143: scale (entry 0 stays 0)
    const u32 current_total = len;
    for (u32 i = j + 1; i < 257; i += 8) cum[i] = (u32)((((u64)cum[i]) << PROB_BITS) / current_total);
    rcx_octet_sync();
    // cppans.h:144-167: every symbol that occurs but lost its range takes one slot from the symbol with the
    // smallest range above 1 (the first such in index order); in order of i, each step seeing the previous ones
    for (u32 i = 0; i < 256; ++i) {
        if (table[i] != 0 && cum[i + 1] == cum[i]) { // (the octet's lanes agree)
            u32 best = 0xFFFFFFFFu;
            for (u32 k = 0; k < 32; ++k) { // lane j looks at symbols 32j .. 32j+31, lowest index first
                const u32 idx = 32 * j + k;
                const u32 freq = cum[idx + 1] - cum[idx];
                const u32 key = (freq << 8) | idx;
                if (freq > 1 && key < best) best = key;
            }
            best = rcx_octet_min(best); // smallest range, then smallest index
            const u32 steal = best & 0xFFu;
            if (steal < i) { // :156-159
                for (u32 k = steal + 1 + j; k <= i; k += 8) cum[k] -= 1;
            } else {         // :160-165
                for (u32 k = i + 1 + j; k <= steal; k += 8) cum[k] += 1;
            }
            rcx_octet_sync();
        }
    }
    // the coding table (cppans.h:176 + the symbol's start)
    for (u32 s = j; s < 256; s += 8) {
        const u32 lo = cum[s], hi = cum[s + 1];
        table[s] = lo | ((hi - lo) << 16);
    }
    rcx_octet_sync();
}

// The block's header (cppans.h:521-527 / :598-604): u32 n, then the 257 scaled cumulative counts.  `at` is 4-byte
// aligned (the slot is 16-byte aligned and everything written behind it came in multiples of 2 with an even count...
// not necessarily of 4: stored bytewise when it is not).
__device__ __forceinline__ void rcx_rans_write_header(u8* at, u32 n, const u32* cum, u32 j)
{
    if ((reinterpret_cast<uintptr_t>(at) & 3u) == 0) {
        u32* h = reinterpret_cast<u32*>(at);
        if (j == 0) h[0] = n;
        for (u32 i = j; i < 257; i += 8) h[1 + i] = cum[i];
    } else {
        for (u32 i = j; i < 258; i += 8) {
            const u32 v = i == 0 ? n : cum[i - 1];
            at[4 * i + 0] = (u8)v;
            at[4 * i + 1] = (u8)(v >> 8);
            at[4 * i + 2] = (u8)(v >> 16);
            at[4 * i + 3] = (u8)(v >> 24);
        }
    }
}

// The eight-state encoder's table: 8 bytes per symbol, shift | inc << 5 | start << 6 | freq << 19 (start and freq can both be
// 4096: 13 bits each) and the multiplier of
// x / freq = mulhi(x + inc, mul) >> shift (rcx_make_div_entry: exact for every 32-bit x; x + 1 cannot wrap here, a state is
// below freq << 20).  It takes the place of cum[] and table[] once the model is built.
__device__ __forceinline__ void rcx_rans_write_header_ent(u8* at, u32 n, const u64* ent, u32 total, u32 j)
{
    for (u32 i = j; i < 258; i += 8) {
        const u32 v = i == 0 ? n : (i == 257 ? total : ((u32)ent[i - 1] >> 6) & 0x1FFFu);
        if ((reinterpret_cast<uintptr_t>(at) & 3u) == 0) reinterpret_cast<u32*>(at)[i] = v;
        else {
            at[4 * i + 0] = (u8)v;
            at[4 * i + 1] = (u8)(v >> 8);
            at[4 * i + 2] = (u8)(v >> 16);
            at[4 * i + 3] = (u8)(v >> 24);
        }
    }
}

// ===========================================================================
// Encode, pass 1: every block's stream ends at the end of its slot; sizes[b] = its length, starts[b] = where it
// begins in the slot.  WORD = the eight-state format.
// ===========================================================================
template <bool WORD>
__global__ __launch_bounds__(256) void rcx_enc_rans_k(const u8* __restrict__ src, u64 n, u32 block, u64 nblocks, u8* __restrict__ slots,
                                                      u64 slot, u32* __restrict__ sizes, u32* __restrict__ starts, u32* status)
{
    __shared__ __attribute__((aligned(16))) u32 lds_all[4 * RCX_RANS_BLOCKS * RCX_RANS_ENC_LDS_DW];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u32 j = lane & 7u, oct = lane >> 3;
    const u64 blk = ((u64)blockIdx.x * 4 + wave) * RCX_RANS_BLOCKS + oct;
    const bool live = blk < nblocks;
    const u64 at = live ? blk * (u64)block : 0;
    const u32 len = live ? (u32)((n - at) < (u64)block ? (n - at) : (u64)block) : 0u;
    u32* cum = lds_all + (wave * RCX_RANS_BLOCKS + oct) * RCX_RANS_ENC_LDS_DW;
    u32* table = cum + 264;
    u8* stage = reinterpret_cast<u8*>(table + 256);
    const u8* in = src + at;
    constexpr u32 PROB_BITS = WORD ? 12u : 14u;

    if (len != 0) rcx_rans_model<PROB_BITS>(cum, table, in, len, j, (reinterpret_cast<uintptr_t>(in) & 15u) == 0);
    // (a wave's LDS operations execute in order and an octet lives in one wave: no barrier needed)
    u64* const ent = reinterpret_cast<u64*>(cum); // WORD: see rcx_rans_write_header_ent
    if (WORD) {
        // lane j turns symbols 32j .. 32j+31 over: all of them read before any is written (the new table lies over the old)
        u32 old[32];
#pragma unroll
        for (u32 i = 0; i < 32; ++i) old[i] = len != 0 ? table[32 * j + i] : 0u;
        rcx_octet_sync();
#pragma unroll 4
        for (u32 i = 0; i < 32; ++i) {
            const u32 freq = old[i] >> 16, start = old[i] & 0xFFFFu;
            const DivEntry d = rcx_make_div_entry(freq ? freq : 1u);
            ent[32 * j + i] = ((u64)d.mul << 32) | d.shift | (d.add ? 32u : 0u) | (start << 6) | (freq << 19);
        }
        rcx_octet_sync();
    }

    u8* const slot_base = slots + blk * slot; // only dereferenced by live lanes
    u32 ptr = (u32)slot;                      // byte offset in the slot: everything from here up is written
    u32 x = WORD ? (1u << 16) : (1u << 23);   // cppans.h:336-339 / :260-263
    const u32 floor_ = RCX_RANS_HEADER + (WORD ? 32u : 4u);
    bool overflow = false;

    // rounds of 8 symbols, from the end of the block
    const u32 rounds = (len + 7) >> 3;
    u32 next_byte = 0;
    if (rounds != 0) {
        const u32 i = 8 * (rounds - 1) + j;
        next_byte = i < len ? in[i] : 0u;
    }
    u8* const ring = stage + 64; // WORD: the stream's newest 256 bytes, byte a of the slot at ring[a & 255]
    u32 drained = (u32)slot;     // WORD: everything from here up is in memory
    if (WORD) {
        // cppans.h:591-594: symbol i goes to state i & 7 = lane j, so every lane codes the byte of its own column.  The eight
        // puts of a round are independent but for the order of their words: written backwards in the order 7 .. 0, i.e.
        // ascending by lane in memory -- an octet ballot gives every emitting lane its place.  The words go into the
        // block's ring in LDS (to a spare halfword for a lane that emits nothing: no branch) and leave for memory as whole
        // 16-byte pieces every eight rounds: 2-byte stores straight to memory were one store instruction a round and
        // kept the wave waiting (profiles/r03_rans8_pmc.json: 45 % of its cycles in s_waitcnt).
#define RCX_RANS8_PUT(ACTIVE, SYM)                                                                                   \
    {                                                                                                                \
        const u64 e_ = ent[(SYM)];                                                                                   \
        const u32 e0_ = (u32)e_, freq_ = e0_ >> 19, start_ = (e0_ >> 6) & 0x1FFFu;                                    \
        const u32 x_max_ = freq_ << 20; /* cppans.h:357: ((2^16 >> 12) << 16) * freq in u32 -- wraps to 0 for freq = 4096 */ \
        const bool emit_ = (ACTIVE) && x_max_ <= x;                                                                  \
        const u32 mask_ = rcx_octet_ballot(emit_, lane);                                                             \
        const u32 words_ = (u32)__popc(mask_), before_ = (u32)__popc(mask_ & ((1u << j) - 1u));                      \
        const bool room_ = ptr >= floor_ + 2 * words_;                                                               \
        overflow = overflow || !room_;                                                                               \
        ptr -= room_ ? 2 * words_ : 0u;                                                                              \
        u8* const where_ = (emit_ && !overflow) ? ring + ((ptr + 2 * before_) & 255u) : ring + 256 + 2 * j;         \
        *reinterpret_cast<unsigned short*>(where_) = (unsigned short)(x & 0xFFFFu);                                  \
        x = emit_ ? x >> 16 : x;                                                                                     \
        const u32 q_ = __umulhi(x + ((e0_ >> 5) & 1u), (u32)(e_ >> 32)) >> (e0_ & 31u); /* x / freq, cppans.h:363 */ \
        const u32 rem_ = x - rcx_mul24(q_, freq_); /* (q < 2^20, freq <= 2^12) */                                    \
        x = (ACTIVE) ? (q_ << 12) + rem_ + start_ : x;                                                               \
    }
        // whole 16-byte pieces of the ring go to memory, the highest first: lane k takes the k-th (at most 8 are due: eight
        // rounds make at most 128 bytes and less than 16 stay behind)
#define RCX_RANS8_DRAIN()                                                                                            \
    {                                                                                                                \
        rcx_octet_sync();                                                                                            \
        const u32 lowest_ = (ptr + 15u) & ~15u;                                                                      \
        const u32 due_ = drained > lowest_ ? (drained - lowest_) >> 4 : 0u;                                          \
        if (live && j < due_) {                                                                                      \
            const u32 a_ = drained - 16u * (j + 1u);                                                                 \
            *reinterpret_cast<U4*>(slot_base + a_) = *reinterpret_cast<const U4*>(ring + (a_ & 255u));               \
        }                                                                                                            \
        drained -= 16u * due_;                                                                                       \
        rcx_octet_sync();                                                                                            \
    }
        // the byte of round k for this lane; rounds below 0 (the queue runs ahead) read the block's first byte
        auto fetch = [&](u32 k) -> u32 { return in[k < rounds ? 8 * k + j : 0u]; };
        u32 r = rounds;
        if (r != 0) { // the block's last round: it may be short
            --r;
            const u32 have = len - 8 * r;
            RCX_RANS8_PUT(live && j < have, next_byte);
        }
        // whole rounds, as many as every block of the wave still has: no tests for a block that has ended
        u32 common = live ? r : 0xFFFFFFFFu;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const u32 other = (u32)__shfl_xor((int)common, o, 64);
            common = common < other ? common : other;
        }
        if (common == 0xFFFFFFFFu) common = 0;
        if (!live) r = common; // (an octet without a block steps along, reading the start of the buffer; it stores nothing)
        // Rounds above a multiple of eight, one by one (their bytes loaded as they come) ...
        const bool by_eights = __all((reinterpret_cast<uintptr_t>(in) & 7u) == 0);
        u32 t = 0;
        {
            const u32 odd = by_eights ? (r & 7u) : common;
            u32 q0 = fetch(r - 1);
            for (; t < odd && t < common; ++t) {
                const u32 q1 = fetch(r - 2);
                RCX_RANS8_PUT(live, q0);
                q0 = q1;
                --r;
                RCX_RANS8_DRAIN();
            }
        }
        // ... then eight rounds = 64 bytes of the block at a time: lane k loads round (base + k)'s eight bytes -- two groups
        // ahead, 16 rounds: a byte loaded four rounds ahead arrived late --, the octet turns them over through LDS (row k =
        // round base + k, lane j reads column j), one round ahead of its use
        if (t < common) {
            auto load8 = [&](u32 base) -> u64 { // base = first round of the group, a multiple of 8 (or below 0: the block's first bytes)
                return *reinterpret_cast<const u64*>(in + 8 * ((base < rounds ? base : 0u) + j));
            };
            u64 g0 = load8(r - 8), g1 = load8(r - 16);
            while (t + 8 <= common) {
                rcx_octet_sync();
                *reinterpret_cast<u64*>(stage + 8 * j) = g0; // rounds r-8 .. r-1
                g0 = g1;
                g1 = load8(r - 24);
                rcx_octet_sync();
                u32 q = stage[8 * 7 + j];
#pragma unroll
                for (u32 i = 8; i-- > 0;) {
                    const u32 cur_ = q;
                    if (i != 0) q = stage[8 * (i - 1) + j];
                    RCX_RANS8_PUT(live, cur_);
                }
                r -= 8;
                t += 8;
                RCX_RANS8_DRAIN();
            }
        }
        // what is left: blocks longer than the wave's shortest, and streams that are not 8-byte aligned
        {
            u32 q0 = fetch(r - 1);
            while (r != 0) {
                const u32 q1 = fetch(r - 2);
                RCX_RANS8_PUT(live, q0);
                q0 = q1;
                --r;
                RCX_RANS8_DRAIN();
            }
        }
        // the bytes that never made a whole piece
        rcx_octet_sync();
        if (live && !overflow)
            for (u32 a = ptr + j; a < drained; a += 8) slot_base[a] = ring[a & 255u];
#undef RCX_RANS8_DRAIN
#undef RCX_RANS8_PUT
    } else {
        // cppans.h:516-519: one state; every lane of the octet carries it, lane 0 stores; the octet stages symbols
        // [8r, 8r+8) in LDS (lane j brings byte j)
        for (u32 r = rounds; r-- > 0;) {
            rcx_octet_sync();
            stage[j] = (u8)next_byte;
            rcx_octet_sync();
            if (r != 0) next_byte = in[8 * (r - 1) + j]; // the next round's byte is on its way while this one is coded
            const u32 have = (len - 8 * r) < 8u ? (len - 8 * r) : 8u; // symbols in this round (only the last one is short)
            for (u32 k = have; k-- > 0;) {
                const u32 e = table[stage[k]];
                const u32 freq = e >> 16, start = e & 0xFFFFu;
                const u32 x_max = freq << 17; // cppans.h:203: ((2^23 >> 14) << 8) * freq
#pragma unroll
                for (u32 t = 0; t < 2; ++t) { // cppans.h:272-279: at most two bytes leave (x < 2^31, x_max >= 2^17)
                    if (live && x_max <= x) {
                        if (ptr <= floor_) overflow = true;
                        else {
                            ptr -= 1;
                            if (j == 0) slot_base[ptr] = (u8)(x & 0xFFu);
                        }
                        x >>= 8;
                    }
                }
                if (live) { // cppans.h:285-286
                    u32 rem;
                    const u32 q = rcx_div_small_quotient(x, freq, rem);
                    x = (q << 14) + rem + start;
                }
            }
        }
    }
    if (!live) return;
    // flush: cppans.h:595-597 (state 0 lowest in memory) / :289-299, then the header
    if (WORD) {
        ptr -= 32;
        u8* p = slot_base + ptr + 4 * j;
        p[0] = (u8)x, p[1] = (u8)(x >> 8), p[2] = (u8)(x >> 16), p[3] = (u8)(x >> 24);
    } else {
        ptr -= 4;
        if (j == 0) {
            u8* p = slot_base + ptr;
            p[0] = (u8)x, p[1] = (u8)(x >> 8), p[2] = (u8)(x >> 16), p[3] = (u8)(x >> 24);
        }
    }
    ptr -= RCX_RANS_HEADER;
    if (WORD) rcx_rans_write_header_ent(slot_base + ptr, len, ent, 1u << PROB_BITS, j);
    else rcx_rans_write_header(slot_base + ptr, len, cum, j);
    if (j == 0) {
        sizes[blk] = overflow ? 0u : (u32)slot - ptr;
        starts[blk] = overflow ? 0u : ptr;
        if (overflow) rcx_flag(status, RCX_ST_CAPACITY, blk);
    }
}

// ===========================================================================
// Decode
// ===========================================================================
__device__ __forceinline__ u32 rcx_load_le32(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }

// ---------------------------------------------------------------------------------------------------------
// The eight-state format: lane j is state j (cppans.h:609-649).
//
// What the step of a lone wave costs is its chain of dependent memory reads, so everything on that chain is in LDS
// and as much of the machine's waves as possible are resident (2.4 KiB of LDS per block: 8 waves per CU):
//   * symbol lookup (the reference's 16 KiB slots_ + 4 KiB slot2symbol_ per stream, cppans.h:59-63) in two reads:
//     first[slot >> 2] = index, among the symbols that occur, of the one holding slot 4*(slot >> 2); the symbol is
//     that one or one of the next three (4 slots hold at most 4 symbols), whose packed entries
//     start | (freq - 1) << 12 | symbol << 24 come back as one pair of ds_read2_b32;
//   * the word stream through a 256-byte ring per block, filled 128 bytes at a time by the octet's 16-byte loads,
//     issued a step before they are written to LDS and seven steps before they can be needed;
//   * the symbols through a 64-byte buffer per block: eight steps are written out as one 8-byte store per lane.
// ---------------------------------------------------------------------------------------------------------
#define RCX_R8_FIRST_BYTES 1024
#define RCX_R8_TABLE_DW 260 /* at most 256 symbols occur + 3 copies of the last (the 4-entry window never leaves the table) */
#define RCX_R8_RING_BYTES 256
#define RCX_R8_LDS_BYTES (RCX_R8_FIRST_BYTES + 4 * RCX_R8_TABLE_DW + RCX_R8_RING_BYTES + 64)

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rcx_dec_rans8_k(const u8* __restrict__ comp, u64 comp_size, const u64* __restrict__ offsets,
                                                             u64 nblocks, u32 block, u64 n, u8* __restrict__ dst, u32* status)
{
    __shared__ __attribute__((aligned(16))) u8 lds_all[WAVES * RCX_RANS_BLOCKS * RCX_R8_LDS_BYTES];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u32 j = lane & 7u, oct = lane >> 3;
    const u64 blk = ((u64)blockIdx.x * WAVES + wave) * RCX_RANS_BLOCKS + oct;
    bool live = blk < nblocks;
    const u64 at = live ? blk * (u64)block : 0;
    u32 len = live ? (u32)((n - at) < (u64)block ? (n - at) : (u64)block) : 0u;
    u8* mine = lds_all + (wave * RCX_RANS_BLOCKS + oct) * RCX_R8_LDS_BYTES;
    u8* first = mine;
    u32* table = reinterpret_cast<u32*>(mine + RCX_R8_FIRST_BYTES);
    u8* ring = mine + RCX_R8_FIRST_BYTES + 4 * RCX_R8_TABLE_DW;
    u8* obuf = ring + RCX_R8_RING_BYTES;

    const u8* s = comp;
    u64 stream_len = 0;
    if (live) {
        const u64 s0 = offsets[blk], s1 = offsets[blk + 1];
        stream_len = s1 - s0;
        s = comp + s0;
        bool good = s1 >= s0 && s1 <= comp_size && stream_len >= RCX_RANS_HEADER + 32;
        if (good) good = rcx_load_le32(s) == len; // cppans.h:616-620: the declared size (the layout says len)
        // The table (cppans.h:621-626).  Lane j takes symbols 32j .. 32j+31: checks that their bounds form a scaled
        // cumulative table (the reference trusts it and would leave its arrays), counts the symbols that occur, and
        // -- once the octet knows how many occur before its range -- writes their packed entries and the cells of
        // `first` whose first slot they hold.
        u32 bounds_ok = 1, mine_count = 0;
        if (good) {
            u32 lo = rcx_load_le32(s + 4 + 4 * (32 * j));
            if (j == 0 && lo != 0) bounds_ok = 0;
            for (u32 k = 0; k < 32; ++k) {
                const u32 hi = rcx_load_le32(s + 4 + 4 * (32 * j + k + 1));
                if (hi < lo || hi > 4096u) bounds_ok = 0;
                mine_count += hi > lo ? 1u : 0u;
                lo = hi;
            }
            if (j == 7 && lo != 4096u) bounds_ok = 0;
        }
        good = good && rcx_octet_ballot(bounds_ok == 0, lane) == 0;
        if (good) {
            u32 rank = rcx_oct_excl_scan(mine_count, (j & 1u) ? ~0u : 0u, (j & 2u) ? ~0u : 0u, (j & 4u) ? ~0u : 0u);
            const u32 total = rcx_oct_sum(mine_count);
            u32 lo = rcx_load_le32(s + 4 + 4 * (32 * j));
            u32 last_entry = 0;
            for (u32 k = 0; k < 32; ++k) {
                const u32 hi = rcx_load_le32(s + 4 + 4 * (32 * j + k + 1));
                if (hi > lo) {
                    const u32 entry = lo | ((hi - lo - 1) << 12) | ((32 * j + k) << 24);
                    table[rank] = entry;
                    last_entry = entry;
                    for (u32 c = (lo + 3) >> 2; c <= (hi - 1) >> 2; ++c) first[c] = (u8)rank; // cells whose slot 4c it holds
                    rank += 1;
                }
                lo = hi;
            }
            // three copies of the last entry behind it (written by the lane that owns it)
            if (mine_count != 0 && rank == total) table[total] = table[total + 1] = table[total + 2] = last_entry;
        }
        if (!good) {
            if (j == 0) rcx_flag(status, RCX_ST_CORRUPT, blk);
            live = false;
            len = 0;
        }
    }
    rcx_octet_sync();

    // the word stream: `origin` = the 16-byte aligned address at or below the first word; p = offset of the next word
    const u8* const comp_end = comp + comp_size;
    const u8* const words = s + RCX_RANS_HEADER + 32;
    const u8* const origin = words - (reinterpret_cast<uintptr_t>(words) & 15u);
    u32 p = (u32)(words - origin);
    u32 filled = 0; // ring holds [filled - 256, filled) of the stream (offsets from origin), as far as it was loaded
    auto load16 = [&](u32 off) -> U4 { // lane j's piece of the 128 bytes at `off`; bytes past the buffer read as zero
        const u8* a = origin + off + 16 * j;
        U4 z;
        z.x = z.y = z.z = z.w = 0;
        if (!live || a >= comp_end) return z;
        if (__builtin_expect(a + 16 <= comp_end, 1)) return *reinterpret_cast<const U4*>(a);
        u32 w[4] = {0, 0, 0, 0}; // the buffer's last, partial piece
        for (u32 k = 0; a + k < comp_end; ++k) w[k >> 2] |= (u32)a[k] << (8 * (k & 3));
        z.x = w[0], z.y = w[1], z.z = w[2], z.w = w[3];
        return z;
    };
    if (live) { // prologue: both halves
        *reinterpret_cast<U4*>(ring + 16 * j) = load16(0);
        *reinterpret_cast<U4*>(ring + 128 + 16 * j) = load16(128);
    }
    filled = 256;
    U4 pend;
    pend.x = pend.y = pend.z = pend.w = 0;
    bool pending = false;
    rcx_octet_sync();

    // Streams compacted by the encoder have even sizes, so in an even-aligned buffer every 16-bit word is aligned;
    // a stream at an odd address has its words read as two bytes (the whole wave then does).
    const bool even = __all(!live || (reinterpret_cast<uintptr_t>(words) & 1u) == 0);
    u32 x = live ? rcx_load_le32(s + RCX_RANS_HEADER + 4 * j) : (1u << 16); // cppans.h:405-409
    u8* out = dst + at;
    const bool out8 = (reinterpret_cast<uintptr_t>(out) & 7u) == 0;
    const u32 groups = len >> 3;
    const u32 max_groups = rcx_wave_max(groups);
    // The fast loop: as many steps (a multiple of 8) as every block of the wave has, words aligned, output 8-byte
    // aligned -- no per-step tests for a block that has ended, no unaligned paths; one wave-uniform branch for the
    // ring's refill.  Whatever is left (the ragged end of a buffer's last block, odd alignments) takes the loop below.
    u32 fast_groups;
    {
        u32 mine = live ? (groups & ~7u) : 0xFFFFFFF8u; // (an octet without a block sets no limit)
        if (live && !out8) mine = 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const u32 other = (u32)__shfl_xor((int)mine, o, 64);
            mine = mine < other ? mine : other;
        }
        fast_groups = (mine == 0xFFFFFFF8u || !even) ? 0u : mine;
    }
    // (in the fast loop the eight symbols a lane decodes in eight steps -- one column of the octet's 8 x 8 bytes -- stay in
    // two registers and the octet turns the 8 x 8 over in registers: lane t ends up with the eight symbols of step t, its
    // 8 bytes of the output.  Through LDS that was a byte write per step and a read per eight, and the LDS unit is what
    // bounds this kernel: profiles/r03_rans8_pmc.json, 5.7 LDS instructions a step, 55 % of its cycles bank conflicts)
    const u32 sel_half = (j & 2u) ? 0x03020706u : 0x05040100u; // v_perm selectors: own 16 bits kept, the partner's (lane ^ 2) taken
    const u32 sel_byte = (j & 1u) ? 0x03070105u : 0x06020400u; // own bytes kept, the partner's (lane ^ 1) taken
    for (u32 g8 = 0; g8 < fast_groups; g8 += 8) {
        u32 col_lo = 0, col_hi = 0;
#pragma unroll
        for (u32 st = 0; st < 8; ++st) {
            if (rcx_any(live && (pending || p + 128 >= filled))) { // some octet has a piece to put into its ring, or to ask for
                if (pending) {
                    *reinterpret_cast<U4*>(ring + ((filled + 16 * j) & (RCX_R8_RING_BYTES - 1))) = pend;
                    filled += 128;
                }
                pending = live && p + 128 >= filled;
                if (pending) pend = load16(filled);
            }
            // cppans.h:636-639 (simdDecSym :412-440)
            const u32 slot_ = x & 4095u;
            const u32 f = first[slot_ >> 2];
            const u32 e0 = table[f], e1 = table[f + 1], e2 = table[f + 2], e3 = table[f + 3];
            u32 e = e0;
            if ((e1 & 4095u) <= slot_) e = e1;
            if ((e2 & 4095u) <= slot_) e = e2;
            if ((e3 & 4095u) <= slot_) e = e3;
            if (st < 4) col_lo |= (e >> 24) << (8 * st);
            else col_hi |= (e >> 24) << (8 * (st - 4));
            x = (((e >> 12) & 4095u) + 1u) * (x >> 12) + slot_ - (e & 4095u); // freq * (x >> 12) + bias
            // cppans.h:640-641 (simdDecRenorm :443-488): the states below 2^16 take one word each, in state order
            const bool need = x < (1u << 16);
            const u32 mask = rcx_octet_ballot(need, lane);
            const u32 o = (p + 2 * (u32)__popc(mask & ((1u << j) - 1u))) & (RCX_R8_RING_BYTES - 1);
            const u32 word = *reinterpret_cast<const unsigned short*>(ring + o);
            x = need ? ((x << 16) | word) : x;
            p += 2 * (u32)__popc(mask);
        }
        // the 8 x 8 turned over: 4-byte blocks between lanes j and j ^ 4, 2-byte blocks between j and j ^ 2, bytes between j and j ^ 1
        {
            const u32 send = (j & 4u) ? col_lo : col_hi;
            const u32 got = rcx_dpp<0x1B>(rcx_dpp<0x141>(send)); // row_half_mirror then quad_perm [3,2,1,0]: lane j ^ 4
            col_lo = (j & 4u) ? got : col_lo;
            col_hi = (j & 4u) ? col_hi : got;
            col_lo = rcx_perm(rcx_dpp<0x4E>(col_lo), col_lo, sel_half); // quad_perm [2,3,0,1]: lane j ^ 2
            col_hi = rcx_perm(rcx_dpp<0x4E>(col_hi), col_hi, sel_half);
            col_lo = rcx_perm(rcx_dpp<0xB1>(col_lo), col_lo, sel_byte); // quad_perm [1,0,3,2]: lane j ^ 1
            col_hi = rcx_perm(rcx_dpp<0xB1>(col_hi), col_hi, sel_byte);
        }
        if (live) *reinterpret_cast<u64*>(out + 8 * g8 + 8 * j) = (u64)col_lo | ((u64)col_hi << 32); // step g8 + j's eight symbols
    }
    for (u32 g = fast_groups; g < max_groups; ++g) {
        const bool on = g < groups;
        // the 128 bytes asked for in the previous step go into the half of the ring that has been used up
        if (pending) {
            *reinterpret_cast<U4*>(ring + ((filled + 16 * j) & (RCX_R8_RING_BYTES - 1))) = pend;
            filled += 128;
            pending = false;
        }
        if (on && p + 128 >= filled) { // the older half is behind p: ask for what follows (>= 7 steps before it can be needed)
            pend = load16(filled);
            pending = true;
        }
        u32 sym = 0;
        if (on) { // cppans.h:636-639 (simdDecSym :412-440)
            const u32 slot_ = x & 4095u;
            const u32 f = first[slot_ >> 2];
            const u32 e0 = table[f], e1 = table[f + 1], e2 = table[f + 2], e3 = table[f + 3];
            u32 e = e0;
            if ((e1 & 4095u) <= slot_) e = e1;
            if ((e2 & 4095u) <= slot_) e = e2;
            if ((e3 & 4095u) <= slot_) e = e3;
            sym = e >> 24;
            x = (((e >> 12) & 4095u) + 1u) * (x >> 12) + slot_ - (e & 4095u); // freq * (x >> 12) + bias
        }
        if (on) obuf[8 * (g & 7u) + j] = (u8)sym;
        // cppans.h:640-641 (simdDecRenorm :443-488): the states below 2^16 take one word each, in state order
        const bool need = on && x < (1u << 16);
        const u32 mask = rcx_octet_ballot(need, lane);
        const u32 o = (p + 2 * (u32)__popc(mask & ((1u << j) - 1u))) & (RCX_R8_RING_BYTES - 1);
        u32 word;
        if (even) word = *reinterpret_cast<const unsigned short*>(ring + o);
        else word = (u32)ring[o] | ((u32)ring[(o + 1) & (RCX_R8_RING_BYTES - 1)] << 8);
        if (need) x = (x << 16) | word;
        p += 2 * (u32)__popc(mask);
        if ((g & 7u) == 7u) { // eight steps = 64 symbols of the block: 8 bytes per lane
            rcx_octet_sync();
            if (on) {
                if (out8) *reinterpret_cast<u64*>(out + 8 * (g - 7) + 8 * j) = *reinterpret_cast<const u64*>(obuf + 8 * j);
                else
                    for (u32 k = 0; k < 8; ++k) out[8 * (g - 7) + 8 * j + k] = obuf[8 * j + k];
            }
            rcx_octet_sync();
        }
    }
    // the steps since the last full group of eight
    if (live) {
        const u32 done = groups & ~7u;
        for (u32 g = done; g < groups; ++g) out[8 * g + j] = obuf[8 * (g & 7u) + j];
        // cppans.h:643-647: the last n mod 8 symbols step without renormalising
        if (8 * groups + j < len) {
            const u32 slot_ = x & 4095u;
            const u32 f = first[slot_ >> 2];
            u32 e = table[f];
            if ((table[f + 1] & 4095u) <= slot_) e = table[f + 1];
            if ((table[f + 2] & 4095u) <= slot_) e = table[f + 2];
            if ((table[f + 3] & 4095u) <= slot_) e = table[f + 3];
            out[8 * groups + j] = (u8)(e >> 24);
        }
        // a valid stream holds every word that was taken (cppans.h:479-481 reads on regardless)
        if (j == 0 && (u64)(words - s) + (p - (u32)(words - origin)) > stream_len) rcx_flag(status, RCX_ST_CORRUPT, blk);
    }
}

// ===========================================================================
// The one-state format at speed.  A block is ONE chain here (as with the range coders), so the octet mapping above
// wastes seven lanes in eight on the coding loop; these kernels are what the block entry points run:
//   rcx_rans_model_k     the model of every block (count / cumulative / normalize), by octets as above -> `models`
//   rcx_enc_rans1_k      one lane per block: table lookups in LDS, bytes gathered into dwords, written backwards
//   rcx_dec_rans1_quad_k 4 lanes per block with the range decoders' machinery (rcx_oct.hpp: table groups, input
//                        ring): the symbol of a slot = the number of cumulative bounds at or below it, counted in
//                        two rounds of 16; slot - start is the minimum of the wrapped differences and the frequency
//                        is minimum - maximum, as in rcx_dec_static_quad_k -- no multiplication at all.
// ===========================================================================
#define RCX_RANS_MODEL_DW 776 /* per block in `models`: cum[257] (+ pad to 264) | 256 x {reciprocal, packed} for rcx_enc_rans1_k */

template <u32 PROB_BITS>
__global__ __launch_bounds__(256) void rcx_rans_model_k(const u8* __restrict__ src, u64 n, u32 block, u64 nblocks, u32* __restrict__ models)
{
    __shared__ u32 lds_all[4 * RCX_RANS_BLOCKS * RCX_RANS_ENC_LDS_DW];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u32 j = lane & 7u, oct = lane >> 3;
    const u64 blk = ((u64)blockIdx.x * 4 + wave) * RCX_RANS_BLOCKS + oct;
    if (blk >= nblocks) return; // (an octet leaves together)
    const u64 at = blk * (u64)block;
    const u32 len = (u32)((n - at) < (u64)block ? (n - at) : (u64)block);
    u32* cum = lds_all + (wave * RCX_RANS_BLOCKS + oct) * RCX_RANS_ENC_LDS_DW;
    u32* table = cum + 264;
    const u8* in = src + at;
    rcx_rans_model<PROB_BITS>(cum, table, in, len, j, (reinterpret_cast<uintptr_t>(in) & 15u) == 0);
    u32* out = models + blk * RCX_RANS_MODEL_DW;
    for (u32 i = j; i < 264; i += 8) out[i] = cum[i];
    // The encoder's per-symbol constants, the reference's EncSymbol (cppans.h:180-250): q = mulhi(x, rcp) >> shift is
    // the exact x / freq for every state the encoder holds (Alverson; rcp = ceil(2^(shift+32) / freq) with
    // shift = ceil(log2 freq) - 1), except freq = 1: rcp = 2^32 - 1, shift = 0 gives q = x - 1 and the bias makes up
    // for it.  Packed: complement of the frequency | shift << 14 | start << 18.
    for (u32 sy = j; sy < 256; sy += 8) {
        const u32 e = table[sy];
        const u32 freq = e >> 16, start = e & 0xFFFFu;
        u32 rcp = 0xFFFFFFFFu, shift = 0;
        if (freq >= 2) {
            u32 up = 0;
            while (freq > (1u << up)) ++up;
            rcp = (u32)(((1ull << (up + 31)) + freq - 1) / freq);
            shift = up - 1;
        }
        out[264 + 2 * sy] = rcp;
        out[264 + 2 * sy + 1] = ((1u << PROB_BITS) - freq) | (shift << 14) | (start << 18);
    }
}

// One lane per block; `lanes_used` = 1 << lanes_shift of the wave's 64 lanes carry a block.  LDS (dynamic, 2 KiB per lane
// in use): entry s of lane l at 8 * (s * lanes_used + l) -- a shift, not a multiplication: the index is formed per symbol.
struct alignas(8) RcxRansSym {
    u32 rcp, packed; // see rcx_rans_model_k
};
// A workgroup is four independent waves (they land one on each SIMD of a CU; single-wave workgroups cluster).
#define RCX_RANS1_ENC_WAVES 4
__global__ __launch_bounds__(64 * RCX_RANS1_ENC_WAVES) void rcx_enc_rans1_k(const u8* __restrict__ src, u64 n, u32 block, u64 nblocks,
                                                                           const u32* __restrict__ models, u8* __restrict__ slots, u64 slot,
                                                                           u32* __restrict__ sizes, u32* __restrict__ starts, u32* status,
                                                                           u32 lanes_shift)
{
    extern __shared__ RcxRansSym rcx_rans1_lds[];
    const u32 lanes_used = 1u << lanes_shift;
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool in_use = lane < lanes_used;
    const u64 blk = in_use ? ((u64)blockIdx.x * RCX_RANS1_ENC_WAVES + wave) * lanes_used + lane : nblocks;
    const bool live = blk < nblocks;
    const u64 at = live ? blk * (u64)block : 0;
    const u32 len = live ? (u32)((n - at) < (u64)block ? (n - at) : (u64)block) : 0u;
    RcxRansSym* table = rcx_rans1_lds + wave * 256u * lanes_used + (in_use ? lane : 0u);
    if (live) {
        const RcxRansSym* m = reinterpret_cast<const RcxRansSym*>(models + blk * RCX_RANS_MODEL_DW + 264);
        for (u32 sy = 0; sy < 256; ++sy) table[sy << lanes_shift] = m[sy];
    }
    const u8* in = src + at;
    u8* const slot_base = slots + (live ? blk : 0) * slot;
    u32 ptr = (u32)slot;      // byte offset in the slot; everything from here up is written
    u32 x = 1u << 23;         // cppans.h:260-263
    u64 acc = 0;              // bytes not yet stored, the newest lowest (it will sit at the lowest address)
    u32 nacc = 0;             // how many: 0..3 between symbols
    const u32 floor_ = RCX_RANS_HEADER + 8; // (the stores below are whole dwords)
    bool overflow = false;

    // One symbol with its constants E in hand (cppans.h:265-287): up to two bytes leave (x < 2^31, x_max >= 2^17), then
    // x = C(s, x) = x + bias + q * (M - freq), q = x / freq (cppans.h:285-286)
#define RCX_RANS1_PUT(E)                                                                                \
    {                                                                                                   \
        const u32 cmpl_ = (E).packed & 0x3FFFu, shift_ = ((E).packed >> 14) & 15u, start_ = (E).packed >> 18; \
        const u32 x_max_ = (16384u - cmpl_) << 17;                                                      \
        const u32 k_ = (x_max_ <= x ? 1u : 0u) + (x_max_ <= (x >> 8) ? 1u : 0u); /* bytes leaving */    \
        /* they go out lowest first: x & 0xFF, then (x >> 8) & 0xFF; the later one ends up lower */     \
        const u32 two_ = ((x & 0xFFu) << 8) | ((x >> 8) & 0xFFu);                                       \
        const u32 out_ = k_ == 2 ? two_ : (x & 0xFFu);                                                  \
        acc = (acc << (8 * k_)) | (k_ ? out_ : 0u);                                                     \
        nacc += k_;                                                                                     \
        x >>= 8 * k_;                                                                                   \
        /* with 4 or 5 bytes held the four oldest leave as one dword, 4-byte aligned (the slot end is): everything  \
         * is computed and selected, the store is the only predicated piece */                          \
        const bool due_ = nacc >= 4;                                                                    \
        const u32 keep8_ = (8 * nacc) & 8u; /* 8 * (nacc - 4) when due: 0 or 8 */                       \
        const u32 word_ = (u32)(acc >> keep8_);                                                         \
        const bool room_ = ptr >= floor_ + 4;                                                           \
        overflow = overflow || (due_ && !room_);                                                        \
        ptr -= (due_ && room_) ? 4u : 0u;                                                               \
        if (due_ && room_) *reinterpret_cast<u32*>(slot_base + ptr) = word_;                            \
        acc = due_ ? (acc & ((1ull << keep8_) - 1ull)) : acc;                                           \
        nacc -= due_ ? 4u : 0u;                                                                         \
        const u32 q_ = __umulhi(x, (E).rcp) >> shift_;                                                  \
        const u32 bias_ = start_ + (cmpl_ == 16383u ? 16383u : 0u); /* freq = 1: cppans.h:232-234 */     \
        x = x + bias_ + rcx_mul24(q_, cmpl_);                                                           \
    }

    if (live) {
        u32 i = len;
        const bool aligned = (reinterpret_cast<uintptr_t>(in) & 15u) == 0;
        while (i != 0 && (!aligned || (i & 15u) != 0)) { // the ragged end (and everything, if the block is not aligned)
            --i;
            const RcxRansSym e = table[(u32)in[i] << lanes_shift];
            RCX_RANS1_PUT(e);
        }
        if (i != 0) {
            U4 cur = *reinterpret_cast<const U4*>(in + i - 16);
            RcxRansSym e_next = table[rcx_byte_of(cur, 15) << lanes_shift];
            while (i != 0) {
                i -= 16;
                U4 nxt = cur;
                if (i != 0) nxt = *reinterpret_cast<const U4*>(in + i - 16);
#pragma unroll
                for (u32 k = 16; k-- > 0;) {
                    const RcxRansSym e = e_next; // the next symbol's constants are on their way while this one is coded
                    e_next = table[(k != 0 ? rcx_byte_of(cur, k - 1) : rcx_byte_of(nxt, 15)) << lanes_shift];
                    RCX_RANS1_PUT(e);
                }
                cur = nxt;
            }
        }
        // what is still held, then the state (cppans.h:289-299), then the header (:521-527)
        for (u32 k = nacc; k-- > 0;) { // oldest first: it sits highest
            ptr -= 1;
            slot_base[ptr] = (u8)(acc >> (8 * k));
        }
        ptr -= 4;
        slot_base[ptr] = (u8)x, slot_base[ptr + 1] = (u8)(x >> 8), slot_base[ptr + 2] = (u8)(x >> 16), slot_base[ptr + 3] = (u8)(x >> 24);
        ptr -= RCX_RANS_HEADER;
        const u32* cum = models + blk * RCX_RANS_MODEL_DW;
        u8* h = slot_base + ptr;
        for (u32 w = 0; w < 258; ++w) {
            const u32 v = w == 0 ? len : cum[w - 1];
            h[4 * w] = (u8)v, h[4 * w + 1] = (u8)(v >> 8), h[4 * w + 2] = (u8)(v >> 16), h[4 * w + 3] = (u8)(v >> 24);
        }
        sizes[blk] = overflow ? 0u : (u32)slot - ptr;
        starts[blk] = overflow ? 0u : ptr;
        if (overflow) rcx_flag(status, RCX_ST_CAPACITY, blk);
    }
#undef RCX_RANS1_PUT
}

// ---------------------------------------------------------------------------------------------------------
// The same encoder as two waves per 64 blocks (default): what one symbol costs a lone wave is the number of instructions
// it issues (rcx_enc_rans1_k: 49 vector + 7 other instructions a symbol = 311 cycles, profiles/r03_rans1_pmc.json), and
// half of them only move bytes.  So the work is cut where rANS allows it -- the state never needs anything back from the
// bytes it has put out (no carry, unlike the range coder):
//   wave 0, the coder:  table lookup (one symbol ahead), the renormalisation test, x = C(s, x); per symbol it hands over
//                       one word: the state's low 16 bits before the step (the bytes that may leave) and how many leave;
//   wave 1, the writer: writes BOTH bytes into a 64-byte ring per block in LDS, at the place of the next byte and the one
//                       after, and moves on by the count -- a byte that did not leave is overwritten by the next one
//                       that does: two byte writes and an add per symbol, no register to shift and mask, no select --
//                       and, once per chunk of 16 symbols, stores what has become whole 16-byte pieces, backwards from
//                       the end of the slot (no store, and no branch, per symbol).
// They meet once per chunk (LDS records, double-buffered; s_waitcnt lgkmcnt(0) + s_barrier).  LDS: 128 KiB of tables
// (2 KiB a block, as before) + 8 KiB of records + 4 KiB of rings: one workgroup of 64 blocks per CU, 256 for a GiB.
// Measured (1 GiB of Zipf bytes, 64 KiB blocks): 8.49 -> 5.71 ms with a writer that gathered dwords in a register (each
// wave about 27 instructions a symbol at about 7 cycles each; the two land on different SIMDs: an idle wave between them
// changes nothing), then with the byte ring and the coder relieved of arranging the bytes: see DESIGN.md section 3.6.
// ---------------------------------------------------------------------------------------------------------
#define RCX_R1W_CHUNK 16u
#define RCX_R1W_RING_WORDS 16u /* 64 bytes: a chunk makes at most 32, fewer than 16 stay behind after a drain, 2 are written ahead */
#define RCX_R1W_TABLE_BYTES (256u * 64u * 8u)
#define RCX_R1W_REC_DW (2u * RCX_R1W_CHUNK * 64u)
#define RCX_R1W_LDS_BYTES (RCX_R1W_TABLE_BYTES + 4u * RCX_R1W_REC_DW + 4u * (RCX_R1W_RING_WORDS + 1u) * 64u + 4u * 64u)

template <bool FULL>
__device__ __forceinline__ void rcx_rans1w_pipeline(u32 wave, u32 lane, bool live, u32 len, u32 nchunks, const u8* in, const RcxRansSym* table,
                                                    u32* rec, u32* oring, u8* slot_base, u32 slot, u32& x, u32& count, u32& drained, bool& overflow)
{
    const u32 max_bytes = slot - (RCX_RANS_HEADER + 8u); // (the tail -- state, header -- has its room below)
    // the coder's look-ahead: the 16 bytes of the chunk it is about to code, and the first symbol's constants
    U4 cur, nxt;
    cur.x = cur.y = cur.z = cur.w = 0;
    nxt = cur;
    RcxRansSym e_next;
    e_next.rcp = 0xFFFFFFFFu;
    e_next.packed = 16383u; // (freq 1, start 0: harmless for a lane that codes nothing)
    auto byte_at = [&](u32 i) -> u32 { return (live && i < len) ? (u32)in[i] : 0u; };
    if (wave == 0 && nchunks != 0) {
        const u32 i0 = (nchunks - 1) * RCX_R1W_CHUNK;
        if (FULL) cur = *reinterpret_cast<const U4*>(in + i0);
        if (FULL && nchunks > 1) nxt = *reinterpret_cast<const U4*>(in + i0 - RCX_R1W_CHUNK);
        e_next = table[(FULL ? rcx_byte_of(cur, 15) : byte_at(i0 + 15)) * 64u];
    }
    for (u32 k = 0; k <= nchunks; ++k) {
        if (wave == 0) {
            if (k < nchunks) { // ---- the coder: chunk c, its symbols from the last to the first ----
                const u32 c = nchunks - 1 - k;
                const u32 i0 = c * RCX_R1W_CHUNK;
                u32* out = rec + (k & 1u) * (RCX_R1W_CHUNK * 64u) + lane;
                U4 ahead = nxt; // the chunk after the next
                if (FULL && c >= 2) ahead = *reinterpret_cast<const U4*>(in + i0 - 2 * RCX_R1W_CHUNK);
#pragma unroll
                for (u32 t = 0; t < RCX_R1W_CHUNK; ++t) {
                    const u32 s = RCX_R1W_CHUNK - 1 - t;
                    const RcxRansSym e = e_next; // the next symbol's constants are on their way while this one is coded
                    {
                        u32 b;
                        if (FULL) b = s != 0 ? rcx_byte_of(cur, s - 1) : rcx_byte_of(nxt, 15);
                        else b = (i0 + s) != 0 ? byte_at(i0 + s - 1) : 0u;
                        e_next = table[b * 64u];
                    }
                    // cppans.h:265-287 with the reference's EncSymbol constants (see rcx_rans_model_k)
                    const u32 cmpl = e.packed & 0x3FFFu, shift = (e.packed >> 14) & 15u, start = e.packed >> 18;
                    const u32 x_max = (16384u - cmpl) << 17;
                    const u32 n_out = (x_max <= x ? 1u : 0u) + (x_max <= (x >> 8) ? 1u : 0u); // bytes leaving: x & 0xFF, then (x >> 8) & 0xFF
                    const bool on = FULL || (live && i0 + s < len);
                    out[t * 64u] = on ? ((x & 0xFFFFu) | (n_out << 16)) : 0u;
                    const u32 xs = x >> (8 * n_out);
                    const u32 q = __umulhi(xs, e.rcp) >> shift;
                    const u32 bias = start + (cmpl == 16383u ? 16383u : 0u); // freq = 1: cppans.h:232-234
                    x = on ? xs + bias + rcx_mul24(q, cmpl) : x;
                }
                cur = nxt;
                nxt = ahead;
            }
        } else if (k >= 1) { // ---- the writer: the records of the step before ----
            const u32* rs = rec + ((k - 1) & 1u) * (RCX_R1W_CHUNK * 64u) + lane;
            u8* const ring = reinterpret_cast<u8*>(oring + lane); // byte e of the stream (counted from the slot's end) at ring_at(e)
            auto ring_at = [&](u32 e) -> u8* { return ring + (((e << 6) & 0xF00u) | (e & 3u)); }; // dword (e >> 2) & 15 of this lane, byte e & 3
            u32 r_next = rs[0];
#pragma unroll
            for (u32 t = 0; t < RCX_R1W_CHUNK; ++t) {
                const u32 r = r_next;
                if (t + 1 < RCX_R1W_CHUNK) r_next = rs[(t + 1) * 64u];
                *ring_at(count) = (u8)r;            // x & 0xFF: leaves first, so it lies highest
                *ring_at(count + 1) = (u8)(r >> 8); // (x >> 8) & 0xFF
                count += r >> 16;                   // 0, 1 or 2 of them count
            }
            overflow = overflow || count > max_bytes; // (cannot happen with the slot rcx_block_bound_for gives: 2 bytes a symbol at most)
            // whole 16-byte pieces go to memory: bytes 16 p .. 16 p + 15 lie at slot - 16 (p + 1) .., the highest-numbered lowest
            while (__any(live && !overflow && 16u * (drained + 1u) <= count)) {
                if (live && !overflow && 16u * (drained + 1u) <= count) {
                    const u32* w = oring + lane;
                    const u32 d = 4u * drained;
                    U4 piece;
                    piece.x = rcx_bswap(w[((d + 3) % RCX_R1W_RING_WORDS) * 64u]);
                    piece.y = rcx_bswap(w[((d + 2) % RCX_R1W_RING_WORDS) * 64u]);
                    piece.z = rcx_bswap(w[((d + 1) % RCX_R1W_RING_WORDS) * 64u]);
                    piece.w = rcx_bswap(w[((d + 0) % RCX_R1W_RING_WORDS) * 64u]);
                    *reinterpret_cast<U4*>(slot_base + slot - 16u * (drained + 1u)) = piece;
                    drained += 1;
                }
            }
        }
        rcx_lds_barrier();
    }
}

__global__ __launch_bounds__(128) void rcx_enc_rans1w_k(const u8* __restrict__ src, u64 n, u32 block, u64 nblocks, const u32* __restrict__ models,
                                                        u8* __restrict__ slots, u64 slot, u32* __restrict__ sizes, u32* __restrict__ starts,
                                                        u32* status)
{
    extern __shared__ __attribute__((aligned(16))) u8 rcx_r1w_lds[];
    RcxRansSym* table_all = reinterpret_cast<RcxRansSym*>(rcx_r1w_lds);
    u32* rec = reinterpret_cast<u32*>(rcx_r1w_lds + RCX_R1W_TABLE_BYTES);
    u32* oring = rec + RCX_R1W_REC_DW;
    u32* final_x = oring + (RCX_R1W_RING_WORDS + 1u) * 64u;
    const u32 lane = threadIdx.x & 63u;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u64 blk = (u64)blockIdx.x * 64u + lane;
    const bool live = blk < nblocks;
    const u64 at = live ? blk * (u64)block : 0;
    const u32 len = live ? (u32)((n - at) < (u64)block ? (n - at) : (u64)block) : 0u;
    const u8* in = src + at;
    RcxRansSym* table = table_all + lane; // entry s at table[64 s]
    if (live) {
        const RcxRansSym* m = reinterpret_cast<const RcxRansSym*>(models + blk * RCX_RANS_MODEL_DW + 264);
        for (u32 sy = wave; sy < 256; sy += 2) table[sy * 64u] = m[sy];
    }
    const u32 maxlen = rcx_wave_max(len);
    const bool full = __all(live && len == block) && (block % 16u == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    const u32 nchunks = (maxlen + RCX_R1W_CHUNK - 1) / RCX_R1W_CHUNK;
    rcx_lds_barrier();

    u8* const slot_base = slots + (live ? blk : 0) * slot;
    u32 x = 1u << 23; // cppans.h:260-263
    u32 count = 0, drained = 0; // the writer's: bytes put out, 16-byte pieces stored
    bool overflow = false;
    if (full) rcx_rans1w_pipeline<true>(wave, lane, live, len, nchunks, in, table, rec, oring, slot_base, (u32)slot, x, count, drained, overflow);
    else rcx_rans1w_pipeline<false>(wave, lane, live, len, nchunks, in, table, rec, oring, slot_base, (u32)slot, x, count, drained, overflow);
    if (wave == 0) final_x[lane] = x;
    rcx_lds_barrier();
    if (wave == 1 && live) {
        // what the ring still holds, the state (cppans.h:289-299), the header (:521-527)
        const u8* ring = reinterpret_cast<const u8*>(oring + lane);
        if (overflow) count = 0;
        for (u32 e = 16u * drained; e < count; ++e) slot_base[(u32)slot - 1u - e] = ring[((e << 6) & 0xF00u) | (e & 3u)];
        u32 ptr = (u32)slot - count;
        const u32 xf = final_x[lane];
        ptr -= 4;
        slot_base[ptr] = (u8)xf, slot_base[ptr + 1] = (u8)(xf >> 8), slot_base[ptr + 2] = (u8)(xf >> 16), slot_base[ptr + 3] = (u8)(xf >> 24);
        ptr -= RCX_RANS_HEADER;
        const u32* cum = models + blk * RCX_RANS_MODEL_DW;
        u8* h = slot_base + ptr;
        for (u32 w = 0; w < 258; ++w) {
            const u32 v = w == 0 ? len : cum[w - 1];
            h[4 * w] = (u8)v, h[4 * w + 1] = (u8)(v >> 8), h[4 * w + 2] = (u8)(v >> 16), h[4 * w + 3] = (u8)(v >> 24);
        }
        sizes[blk] = overflow ? 0u : (u32)slot - ptr;
        starts[blk] = overflow ? 0u : ptr;
        if (overflow) rcx_flag(status, RCX_ST_CAPACITY, blk);
    }
}

// Decode, 4 lanes per block.  LDS per wave: four table groups (rcx_oct.hpp: node n of the four blocks of a group in
// the four quarters of a 256-byte row) | sixteen input rings.  Node n of a block = its cumulative bounds
// cum[16n+1 .. 16n+16]; lane j keeps cum[16(4j+1)] .. cum[16(4j+4)] in registers.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rcx_dec_rans1_quad_k(const u8* __restrict__ comp, u64 comp_size, const u64* __restrict__ offsets,
                                                                  u64 nblocks, u32 block, u64 n, u8* __restrict__ dst, u32* status,
                                                                  u32 quads_used, u32* track)
{
    __shared__ __attribute__((aligned(256))) u8 lds_all[WAVES * RCX_SQUAD_LDS_BYTES];
    const u32 lane = threadIdx.x & 63u;
    const u32 wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u8* lds = lds_all + wave_in_wg * RCX_SQUAD_LDS_BYTES;
    const u32 j = lane & 3u, quad = lane >> 2;
    const bool in_use = quad < quads_used; // see rcx_dec_quad_k: the other quads decode along and store nothing
    const u64 blk = ((u64)blockIdx.x * WAVES + wave_in_wg) * quads_used + (quad & (quads_used - 1u));
    bool live = blk < nblocks;
    const u64 at = live ? blk * (u64)block : 0;
    u32 len = live ? (u32)((n - at) < (u64)block ? (n - at) : (u64)block) : 0u;
    const u32 group = 2u * (quad >> 3) + ((0x96u >> (quad & 7u)) & 1u), quarter = (quad & 7u) >> 1;
    u8* mine = lds + group * RCX_QUAD_GROUP_BYTES + quarter * 64;
    U4* leaves = reinterpret_cast<U4*>(mine) + j;
    U4* scratch = reinterpret_cast<U4*>(mine + 16 * 256);
    u32* block_ring = reinterpret_cast<u32*>(lds + 4 * RCX_QUAD_GROUP_BYTES + quad * RCX_QUAD_RING_BYTES);

    QuadInput in;
    u64 stream_len = 0;
    u32 U1 = 1, U2 = 2, U3 = 3, U4_ = 16384;
    if (live) {
        const u64 s0 = offsets[blk], s1 = offsets[blk + 1];
        stream_len = s1 - s0;
        const u8* s = comp + s0;
        bool good = s1 >= s0 && s1 <= comp_size && stream_len >= RCX_RANS_HEADER + 4;
        if (good) good = rcx_load_le32(s) == len; // cppans.h:540-543: the declared size (the layout says len)
        u32 ok = 1;
        if (good) {
            // lane j takes cum[64j+1 .. 64j+64] = nodes 4j .. 4j+3; the table must be a scaled cumulative one (the
            // reference trusts it, cppans.h:544, and would leave its arrays)
            u32 prev = rcx_load_le32(s + 4 + 4 * (64 * j));
            if (j == 0 && prev != 0) ok = 0;
            for (u32 i = 0; i < 64; ++i) {
                const u32 v = rcx_load_le32(s + 4 + 4 * (64 * j + i + 1));
                if (v < prev || v > 16384u) ok = 0;
                reinterpret_cast<u32*>(mine + (4 * j + (i >> 4)) * 256)[i & 15u] = v;
                if (i == 15) U1 = v;
                if (i == 31) U2 = v;
                if (i == 47) U3 = v;
                if (i == 63) U4_ = v;
                prev = v;
            }
            if (j == 3 && prev != 16384u) ok = 0;
        }
        good = good && rcx_quad_or(ok ? 0u : 1u) == 0;
        if (good) {
            // QuadInput::begin wants 4 size bytes + 4 state bytes in front of the payload: the state (u32 LE at
            // 1032) is what it reads big-endian into `low`
            in.begin(s + RCX_RANS_HEADER - 4, comp + s1, block_ring, scratch + 3);
        } else {
            if (j == 0 && in_use) rcx_flag(status, RCX_ST_CORRUPT, blk);
            live = false;
            len = 0;
        }
    }
    if (!live) {
        in.idle(comp, block_ring, scratch + 3);
        U1 = 1, U2 = 2, U3 = 3, U4_ = 16384;
        U4 v;
        v.x = v.y = v.z = v.w = 16384;
        for (u32 q = 0; q < 16; ++q) leaves[q * 16] = v;
    }
    u32 x = live ? rcx_bswap(in.low) : (1u << 23); // cppans.h:303-310
    const u32 maxlen = rcx_wave_max(len);
    const bool full = __all(live && len == block) && (block % 16u == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0);
    u8* out = dst + at;
    const bool leader = live && in_use && j == 0;
    const u32 leaves_lds = (u32)reinterpret_cast<uintptr_t>(leaves);

    const u32 ring_lds = (u32)reinterpret_cast<uintptr_t>(block_ring);
    // One symbol (cppans.h:556-561): get, the symbol of the slot, advance, renormalise by at most two bytes.
    // Written as instruction sequences like rcx_dec_quad_k / rcx_dec_static_quad_k (a lone wave: every compare result and
    // every DPP source at least two instructions old, nothing but the node index's own steps in front of the leaf read);
    // bounds at or below the slot are counted as the subtractions that do not borrow, slot - start is the unsigned
    // minimum of the differences and the frequency minimum - maximum (mod 2^32).  What no later step of the chain
    // needs is made by the NEXT symbol behind its leaf read (HP = 1: PWORD, PSHIFT are the earlier symbol's word and
    // bit position) or by RCX_RANS1_FINISH: the symbol's byte, the stream position and the read of the ring pair there.
    u32 p_nd_ = 0, p_nb_ = 0, p_r8_ = 0;
#define RCX_QP1 "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define RCX_QP2 "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define RCX_R1_PREV_0
#define RCX_R1_PREV_1 "\n\tv_lshl_add_u32 %[ps], %[pnd], 4, %[pnb]\n\t"                                             \
                      "v_lshl_or_b32 %[pword], %[ps], %[psh], %[pword]"
#define RCX_RANS1_SYMBOL(HP, PWORD, PSHIFT)                                                                  \
    {                                                                                                        \
        u32 slot_, xs_, node_, rem_, la_, ro_, x1_, x2_, x3_, x4_, ps_;                                      \
        u64 c1_, c2_, c3_, c4_, cz_;                                                                         \
        asm volatile("v_and_b32 %[sl], 0x3fff, %[x]\n\t"                                                     \
                     "v_sub_co_u32_e64 %[x1], %[c1], %[sl], %[u1]\n\t"                                       \
                     "v_sub_co_u32_e64 %[x2], %[c2], %[sl], %[u2]\n\t"                                       \
                     "v_sub_co_u32_e64 %[x3], %[c3], %[sl], %[u3]\n\t"                                       \
                     "v_sub_co_u32_e64 %[x4], %[c4], %[sl], %[u4]\n\t"                                       \
                     "v_subb_co_u32_e64 %[nd], %[cz], 4, 0, %[c1]\n\t"                                       \
                     "v_subb_co_u32_e64 %[nd], %[cz], %[nd], 0, %[c2]\n\t"                                   \
                     "v_subb_co_u32_e64 %[nd], %[cz], %[nd], 0, %[c3]\n\t"                                   \
                     "v_subb_co_u32_e64 %[nd], %[cz], %[nd], 0, %[c4]\n\t"                                   \
                     "v_min3_u32 %[rm], %[x1], %[x2], %[x3]\n\t"                                             \
                     "v_min3_u32 %[rm], %[rm], %[x4], %[sl]\n\t"                                             \
                     "v_add_u32_dpp %[nd], %[nd], %[nd] " RCX_QP1                                            \
                     "v_lshrrev_b32 %[xs], 14, %[x]\n\t"                                                     \
                     "v_min_u32_dpp %[rm], %[rm], %[rm] " RCX_QP1                                            \
                     "v_add_u32_dpp %[nd], %[nd], %[nd] " RCX_QP2                                            \
                     "v_lshl_add_u32 %[la], %[nd], 8, %[lvb]"                                                \
                     : [sl] "=&v"(slot_), [xs] "=&v"(xs_), [nd] "=&v"(node_), [rm] "=&v"(rem_), [la] "=&v"(la_), \
                       [x1] "=&v"(x1_), [x2] "=&v"(x2_), [x3] "=&v"(x3_), [x4] "=&v"(x4_),                   \
                       [c1] "=&s"(c1_), [c2] "=&s"(c2_), [c3] "=&s"(c3_), [c4] "=&s"(c4_), [cz] "=&s"(cz_)   \
                     : [x] "v"(x), [u1] "v"(U1), [u2] "v"(U2), [u3] "v"(U3), [u4] "v"(U4_), [lvb] "v"(leaves_lds)); \
        const RcxV4 l_ = *reinterpret_cast<const RcxLdsV4*>(la_);                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        asm volatile("v_min_u32_dpp %[rm], %[rm], %[rm] " RCX_QP2                                            \
                     "v_add_u32 %[bp], %[bp], %[pr8]\n\t" /* the stream position after the earlier symbol */ \
                     "v_bfe_u32 %[ro], %[bp], 5, 5\n\t"                                                      \
                     "v_lshl_add_u32 %[ro], %[ro], 2, %[rb]"                                                 \
                     RCX_R1_PREV_##HP                                                                        \
                     : [rm] "+v"(rem_), [bp] "+v"(in.bp8), [ro] "=&v"(ro_), [ps] "=&v"(ps_), [pword] "+v"(PWORD) \
                     : [pr8] "v"(p_r8_), [rb] "v"(ring_lds), [pnd] "v"(p_nd_), [pnb] "v"(p_nb_), [psh] "n"(PSHIFT)); \
        {                                                                                                    \
            const RcxLdsU32* at_ = reinterpret_cast<const RcxLdsU32*>(ro_); /* the bytes this symbol's renormalisation may take */ \
            in.w0 = at_[0];                                                                                  \
            in.w1 = at_[1];                                                                                  \
        }                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        u32 lo_, rg_, nb_, hi_, y1_, y2_, y3_, y4_;                                                          \
        asm volatile("v_sub_co_u32_e64 %[y1], %[c1], %[sl], %[lx]\n\t"                                       \
                     "v_sub_co_u32_e64 %[y2], %[c2], %[sl], %[ly]\n\t"                                       \
                     "v_sub_co_u32_e64 %[y3], %[c3], %[sl], %[lz]\n\t"                                       \
                     "v_sub_co_u32_e64 %[y4], %[c4], %[sl], %[lw]\n\t"                                       \
                     "v_subb_co_u32_e64 %[nb], %[c1], 4, 0, %[c1]\n\t"                                       \
                     "v_min3_u32 %[lo], %[y1], %[y2], %[y3]\n\t"                                             \
                     "v_subb_co_u32_e64 %[nb], %[c2], %[nb], 0, %[c2]\n\t"                                   \
                     "v_max3_u32 %[hi], %[y1], %[y2], %[y3]\n\t"                                             \
                     "v_subb_co_u32_e64 %[nb], %[c3], %[nb], 0, %[c3]\n\t"                                   \
                     "v_min3_u32 %[lo], %[lo], %[y4], %[rem]\n\t"                                            \
                     "v_subb_co_u32_e64 %[nb], %[c4], %[nb], 0, %[c4]\n\t"                                   \
                     "v_max_u32 %[hi], %[hi], %[y4]\n\t"                                                     \
                     "v_min_u32_dpp %[lo], %[lo], %[lo] " RCX_QP1                                            \
                     "v_add_u32_dpp %[nb], %[nb], %[nb] " RCX_QP1                                            \
                     "v_max_u32_dpp %[hi], %[hi], %[hi] " RCX_QP1                                            \
                     "v_min_u32_dpp %[lo], %[lo], %[lo] " RCX_QP2                                            \
                     "v_add_u32_dpp %[nb], %[nb], %[nb] " RCX_QP2                                            \
                     "v_max_u32_dpp %[hi], %[hi], %[hi] " RCX_QP2                                            \
                     "v_sub_u32 %[rg], %[lo], %[hi]"                                                         \
                     : [lo] "=&v"(lo_), [rg] "=&v"(rg_), [nb] "=&v"(nb_), [hi] "=&v"(hi_), [y1] "=&v"(y1_),   \
                       [y2] "=&v"(y2_), [y3] "=&v"(y3_), [y4] "=&v"(y4_), [c1] "=&s"(c1_), [c2] "=&s"(c2_),  \
                       [c3] "=&s"(c3_), [c4] "=&s"(c4_)                                                      \
                     : [sl] "v"(slot_), [lx] "v"(l_.x), [ly] "v"(l_.y), [lz] "v"(l_.z), [lw] "v"(l_.w),      \
                       [rem] "v"(rem_));                                                                     \
        p_nd_ = node_;                                                                                       \
        p_nb_ = nb_;                                                                                         \
        x = rcx_mul24(rg_, xs_) + lo_; /* freq * (x >> 14) + slot - start (cppans.h:326) */                   \
        const u32 r8_ = (rcx_clz(x) - 1u) & 0x18u; /* cppans.h:328-332: x >= 2^7 here; 16 bits below 2^15, 8 below 2^23 */ \
        in.n4 = rcx_bswap(rcx_funnel_shr(in.w1, in.w0, in.bp8));                                             \
        x = (u32)(((((u64)x) << 32) | in.n4) << r8_ >> 32);                                                  \
        p_r8_ = r8_;                                                                                         \
    }
    // the byte of the last symbol decoded and the stream position behind it
#define RCX_RANS1_FINISH(WORD, SHIFT)                                                                        \
    {                                                                                                        \
        (WORD) |= ((p_nd_ << 4) + p_nb_) << (SHIFT);                                                         \
        in.bp8 += p_r8_;                                                                                     \
        p_r8_ = 0;                                                                                           \
    }

    if (full) {
        // (Four pieces kept back and stored together as 64 bytes, as rcx_dec_quad_k does, bring the HBM writes down from 4.4 GB
        // per GiB decoded to the bytes themselves but cost this kernel 4 % in time -- 11.3 -> 11.8 ms: it is bound by its
        // vector instructions, not by memory -- so the pieces go one by one.)
        for (u32 i0 = 0; i0 < maxlen; i0 += 16) {
            in.topup();
            u32 w0_ = 0, w1_ = 0, w2_ = 0, w3_ = 0;
            RCX_RANS1_SYMBOL(0, w0_, 0) RCX_RANS1_SYMBOL(1, w0_, 0) RCX_RANS1_SYMBOL(1, w0_, 8) RCX_RANS1_SYMBOL(1, w0_, 16)
            RCX_RANS1_SYMBOL(1, w0_, 24) RCX_RANS1_SYMBOL(1, w1_, 0) RCX_RANS1_SYMBOL(1, w1_, 8) RCX_RANS1_SYMBOL(1, w1_, 16)
            RCX_RANS1_SYMBOL(1, w1_, 24) RCX_RANS1_SYMBOL(1, w2_, 0) RCX_RANS1_SYMBOL(1, w2_, 8) RCX_RANS1_SYMBOL(1, w2_, 16)
            RCX_RANS1_SYMBOL(1, w2_, 24) RCX_RANS1_SYMBOL(1, w3_, 0) RCX_RANS1_SYMBOL(1, w3_, 8) RCX_RANS1_SYMBOL(1, w3_, 16)
            RCX_RANS1_FINISH(w3_, 24)
            if (leader) {
                U4 o;
                o.x = w0_, o.y = w1_, o.z = w2_, o.w = w3_;
                *reinterpret_cast<U4*>(out + i0) = o;
            }
        }
    } else {
        for (u32 i = 0; i < maxlen; ++i) {
            if ((i & 15u) == 0) in.topup();
            if (i < len) { // the 4 lanes of a quad agree
                u32 sym = 0;
                RCX_RANS1_SYMBOL(0, sym, 0);
                RCX_RANS1_FINISH(sym, 0);
                if (leader) out[i] = (u8)sym;
            }
        }
    }
#undef RCX_RANS1_SYMBOL
#undef RCX_RANS1_FINISH
#undef RCX_R1_PREV_0
#undef RCX_R1_PREV_1
#undef RCX_QP1
#undef RCX_QP2
    // a valid stream holds every byte that was taken
    const u64 taken = RCX_RANS_HEADER - 4 + in.taken(); // QuadInput counts from 8 bytes into what it was given
    if (leader && taken > stream_len) rcx_flag(status, RCX_ST_CORRUPT, blk);
    // the single-stream call wants what rANS::decode returns: the payload bytes consumed (cppans.h:562)
    if (track && leader && blk == 0) track[0] = (u32)(taken - RCX_RANS_HEADER);
}

