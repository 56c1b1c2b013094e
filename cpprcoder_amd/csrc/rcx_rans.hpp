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
// Mapping: a block is an octet of lanes, a wave is 8 blocks.  With the eight-state format lane j IS state j, so every
// lane codes; with the one-state format the octet's lanes all carry the same state (SIMT makes that free) and share
// the table work (histogram, scaling, the symbol search of the decoder).  The static model lives in LDS per block;
// nothing is adaptive, so unlike the range coders a block of the eight-state format is 8 chains, not one.
//
// Included at the end of rcx_kernels.hpp.
#pragma once

#define RCX_RANS_HEADER 1032u /* 258 dwords: cppans.h:521, :598 */
#define RCX_RANS_BLOCKS 8     /* blocks per wave */

// LDS of one block while encoding: cum[257] (kept for the header) | table[256] = start | freq << 16 | 64 staged input bytes
#define RCX_RANS_ENC_LDS_DW (264 + 256 + 16)
// ... while decoding, one-state format: cum[257] as u16 (+ pad) | first[64] = the symbol holding slot 256*k (the octet
// scans on from there); eight-state format: see rcx_dec_rans8_k
#define RCX_RANS_DEC_CUM_BYTES 528
#define RCX_RANS1_DEC_LDS_BYTES (RCX_RANS_DEC_CUM_BYTES + 64)

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

// ===========================================================================
// Encode, pass 1: every block's stream ends at the end of its slot; sizes[b] = its length, starts[b] = where it
// begins in the slot.  WORD = the eight-state format.
// ===========================================================================
template <bool WORD>
__global__ __launch_bounds__(256) void rcx_enc_rans_k(const u8* __restrict__ src, u64 n, u32 block, u64 nblocks, u8* __restrict__ slots,
                                                      u64 slot, u32* __restrict__ sizes, u32* __restrict__ starts, u32* status)
{
    __shared__ u32 lds_all[4 * RCX_RANS_BLOCKS * RCX_RANS_ENC_LDS_DW];
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

    u8* const slot_base = slots + blk * slot; // only dereferenced by live lanes
    u32 ptr = (u32)slot;                      // byte offset in the slot: everything from here up is written
    u32 x = WORD ? (1u << 16) : (1u << 23);   // cppans.h:336-339 / :260-263
    const u32 floor_ = RCX_RANS_HEADER + (WORD ? 32u : 4u);
    bool overflow = false;

    // rounds of 8 symbols, from the end of the block: the octet stages symbols [8r, 8r+8) in LDS (lane j brings byte j)
    const u32 rounds = (len + 7) >> 3;
    u32 next_byte = 0;
    if (rounds != 0) {
        const u32 i = 8 * (rounds - 1) + j;
        next_byte = i < len ? in[i] : 0u;
    }
    for (u32 r = rounds; r-- > 0;) {
        rcx_octet_sync();
        stage[j] = (u8)next_byte;
        rcx_octet_sync();
        if (r != 0) next_byte = in[8 * (r - 1) + j]; // the next round's byte is on its way while this one is coded
        const u32 have = (len - 8 * r) < 8u ? (len - 8 * r) : 8u; // symbols in this round (only the last one is short)
        if (WORD) {
            // cppans.h:591-594: symbol i goes to state i & 7 = lane j; the eight puts of a round are independent but
            // for the order of their words: written backwards in the order 7 .. 0, i.e. ascending by lane in memory
            const bool active = live && j < have;
            const u32 e = table[stage[j]];
            const u32 freq = e >> 16, start = e & 0xFFFFu;
            const u32 x_max = freq << 20; // cppans.h:357: ((2^16 >> 12) << 16) * freq in u32 -- wraps to 0 for freq = 4096
            const bool emit = active && x_max <= x;
            const u32 mask = rcx_octet_ballot(emit, lane);
            const u32 words = (u32)__popc(mask), before = (u32)__popc(mask & ((1u << j) - 1u));
            if (ptr < floor_ + 2 * words) overflow = true;
            else ptr -= 2 * words;
            if (emit && !overflow) *reinterpret_cast<unsigned short*>(slot_base + ptr + 2 * before) = (unsigned short)(x & 0xFFFFu);
            if (emit) x >>= 16;
            if (active) { // cppans.h:363
                u32 rem;
                const u32 q = rcx_div_small_quotient(x, freq, rem);
                x = (q << 12) + rem + start;
            }
        } else {
            // cppans.h:516-519: one state; every lane of the octet carries it, lane 0 stores
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
    rcx_rans_write_header(slot_base + ptr, len, cum, j);
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

// Reads and checks a block's header; the scaled cumulative counts go to cum16[0..256].  The reference trusts the
// table (cppans.h:544, :621); a table that is not a scaled cumulative table would send it out of its arrays, so such a
// block is reported as corrupt instead.  Returns true if the block can be decoded.
template <u32 PROB_BITS>
__device__ __forceinline__ bool rcx_rans_read_header(const u8* s, u64 stream_len, u32 len, unsigned short* cum16, u32 j, u32 lane, u32 payload_min)
{
    bool good = stream_len >= RCX_RANS_HEADER + payload_min;
    if (good) good = rcx_load_le32(s) == len; // cppans.h:540-543: the declared size (the layout says len)
    bool mine_ok = true;
    if (good) {
        for (u32 i = j; i < 257; i += 8) { // lane j checks entries j, j+8, ... against their predecessors
            const u32 v = rcx_load_le32(s + 4 + 4 * i);
            const u32 before = i == 0 ? 0u : rcx_load_le32(s + 4 * i);
            if (v > (1u << PROB_BITS) || v < before) mine_ok = false;
            if (i == 0 && v != 0) mine_ok = false;
            if (i == 256 && v != (1u << PROB_BITS)) mine_ok = false;
            cum16[i] = (unsigned short)v;
        }
    }
    const bool all_ok = rcx_octet_ballot(!mine_ok, lane) == 0;
    rcx_octet_sync();
    return good && all_ok;
}

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
        if (a + 16 <= comp_end) return *reinterpret_cast<const U4*>(a);
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
    for (u32 g = 0; g < max_groups; ++g) {
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

// The one-state format (cppans.h:532-564): all lanes of the octet carry the state; the symbol of a slot is found by
// the octet together: first[k] = the symbol holding slot 256k, and from there eight candidates at a time are tested
// against their upper bounds (cum2sym of cppans.h:545-550 would be 16 KiB per block).
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rcx_dec_rans1_k(const u8* __restrict__ comp, u64 comp_size, const u64* __restrict__ offsets,
                                                             u64 nblocks, u32 block, u64 n, u8* __restrict__ dst, u32* status, u32* track)
{
    __shared__ __attribute__((aligned(16))) u8 lds_all[WAVES * RCX_RANS_BLOCKS * RCX_RANS1_DEC_LDS_BYTES];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u32 j = lane & 7u, oct = lane >> 3;
    const u64 blk = ((u64)blockIdx.x * WAVES + wave) * RCX_RANS_BLOCKS + oct;
    bool live = blk < nblocks;
    const u64 at = live ? blk * (u64)block : 0;
    u32 len = live ? (u32)((n - at) < (u64)block ? (n - at) : (u64)block) : 0u;
    u8* mine = lds_all + (wave * RCX_RANS_BLOCKS + oct) * RCX_RANS1_DEC_LDS_BYTES;
    unsigned short* cum16 = reinterpret_cast<unsigned short*>(mine);
    u8* first = mine + RCX_RANS_DEC_CUM_BYTES;

    const u8* s = comp;
    u64 stream_len = 0;
    if (live) {
        const u64 s0 = offsets[blk], s1 = offsets[blk + 1];
        stream_len = s1 - s0;
        s = comp + s0;
        bool good = s1 >= s0 && s1 <= comp_size;
        if (good) good = rcx_rans_read_header<14>(s, stream_len, len, cum16, j, lane, 4);
        if (!good) {
            if (j == 0) rcx_flag(status, RCX_ST_CORRUPT, blk);
            live = false;
            len = 0;
        }
    }
    if (live) {
        // first[k]: the symbol with cum[sym] <= 256k < cum[sym+1]; lane j fills cells j, j+8, ... by walking the table
        u32 sym = 0;
        for (u32 k = j; k < 64; k += 8) {
            while (cum16[sym + 1] <= 256u * k) ++sym; // cum[256] = 2^14 > 256k: ends
            first[k] = (u8)sym;
        }
    }
    rcx_octet_sync();
    const u8* const comp_end = comp + comp_size;
    u32 x = live ? rcx_load_le32(s + RCX_RANS_HEADER) : (1u << 23); // cppans.h:303-310
    u64 rp = RCX_RANS_HEADER + 4;                                    // the next payload byte, as an offset in the stream
    u8* out = dst + at;
    const u32 max_len = rcx_wave_max(len);
    bool ran_dry = false;
    for (u32 i = 0; i < max_len; ++i) {
        const bool on = i < len;
        // the next two bytes, asked for before the search (never past the compressed buffer)
        const u8* ra = s + rp;
        const u32 b0 = live && ra < comp_end ? ra[0] : 0u, b1 = live && ra + 1 < comp_end ? ra[1] : 0u;
        const u32 slot_ = x & 16383u; // cppans.h:313-316
        u32 sym = first[on ? (slot_ >> 8) : 0u];
        // the octet tests candidates sym + j: the symbol is the first whose upper bound lies above the slot
        for (;;) { // (octets leave this loop one by one: the lanes of an octet always agree)
            const u32 c = sym + j;
            const bool hit = !on || (c < 256u && cum16[c + 1] > slot_);
            const u32 m = rcx_octet_ballot(hit, lane);
            if (m != 0) {
                sym += (u32)__ffs((int)m) - 1;
                break;
            }
            sym += 8;
        }
        if (on) {
            const u32 lo = cum16[sym], hi = cum16[sym + 1];
            if (j == 0) out[i] = (u8)sym;
            x = (hi - lo) * (x >> 14) + slot_ - lo; // cppans.h:326
            // cppans.h:328-332: at most two bytes come in (x >= 2^9 after the step)
            if (x < (1u << 23)) {
                x = (x << 8) | b0;
                rp += 1;
                if (x < (1u << 23)) {
                    x = (x << 8) | b1;
                    rp += 1;
                }
            }
            if (rp > stream_len) ran_dry = true;
        }
    }
    if (live && j == 0 && ran_dry) rcx_flag(status, RCX_ST_CORRUPT, blk);
    // the single-stream call wants what rANS::decode returns: the payload bytes consumed (cppans.h:562)
    if (track && live && j == 0 && blk == 0) track[0] = (u32)(rp - RCX_RANS_HEADER);
}
