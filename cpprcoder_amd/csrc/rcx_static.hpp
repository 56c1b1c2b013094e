// rcx_static.hpp -- the static (two-pass) range coder RangeEncoder<T> on gfx950: one lane per block
// (rcx_enc_static_k, rcx_dec_static_k), and the many-wave / four-lane kernels built on rcx_oct.hpp's machinery
// (rcx_enc_static3_k, rcx_dec_static_quad_k).
//
// Reference: cpprcoder.h:321-619.  Stream of one block:
//   [u32 LE n][256 x u16 LE counts][0x00 lead-in][payload ...][u32 BE low]      (516-byte header, :331)
// encode (:375-458): count() histogram with the order-dependent 16-bit squeeze (:543-571), write16
// (:604-619), calcCumulatives (:573-583), then the same divide / multiply / carry / renormalise loop
// as the adaptive coder with a FIXED table, range starting at 0xFFFFFFFF (:382) and the tail rule
// "low == 0xFFFFFFFF bumps the held byte and fills with 0x00" (:439-451).
// decode (:460-519): table from the header, low from bytes 1..4 after it (:494-498), per symbol
// t = range/total, find(low/t) (:521-535), then renormalise.
//
// The table of a block is 257 dwords in LDS, dword-interleaved over the 64 lanes of the wave
// (entry i of lane l at (i*64 + l)*4), so any per-lane index is bank-conflict free.  The divisor
// (total) is fixed per block but differs between lanes: its multiply-add magic is computed once per
// block on the device.  Included at the end of rcx_kernels.hpp.
#pragma once

#define RCX_STATIC_HEADER 516u
#define RCX_STATIC_LDS_DW (257 * RCX_LANES)

struct StaticTable {
    u32* col; // this lane's column
    __device__ __forceinline__ u32 get(u32 i) const { return col[i * RCX_LANES]; }
    __device__ __forceinline__ void set(u32 i, u32 v) const { col[i * RCX_LANES] = v; }
    __device__ __forceinline__ void inc(u32 i) const { rcx_lds_inc(col + i * RCX_LANES); }
    // cpprcoder.h:573-583: counts -> exclusive running sums, entry 256 = total
    __device__ __forceinline__ u32 accumulate() const
    {
        u32 run = 0;
        for (u32 i = 0; i < 256; ++i) {
            const u32 c = get(i);
            set(i, run);
            run += c;
        }
        set(256, run);
        return run;
    }
};


// count() (cpprcoder.h:543-571) for the symbols [from, to) of every lane's block, with the order-dependent 16-bit
// squeeze (:549-555): when a count is 0xFFFF or more BEFORE its increment, every non-zero count becomes
// (c >> 1) | 1 first.  PIECES: the blocks are whole and 16-byte aligned and `from` is a multiple of 16, so the input
// is taken 16 bytes at a time: the 16 increments are issued as returning LDS atomics, and only if one of them
// found 0xFFFF or more (a block dominated by one symbol) are they taken back and redone one by one.
__device__ __forceinline__ void rcx_static_squeeze(const StaticTable& tab)
{
    for (u32 q = 0; q < 256; ++q) {
        const u32 c = tab.get(q);
        if (c > 0) tab.set(q, (c >> 1) | 1u);
    }
}
__device__ __forceinline__ void rcx_static_count_one(const StaticTable& tab, u32 b)
{
    if (tab.get(b) >= 0xFFFFu) rcx_static_squeeze(tab);
    tab.set(b, tab.get(b) + 1);
}
template <bool PIECES>
__device__ __forceinline__ void rcx_static_count_checked(const StaticTable& tab, const u8* in, u32 from, u32 to, u32 len)
{
    u32 i = from;
    if (PIECES) {
        for (; i + 16 <= to; i += 16) {
            const U4 piece = *reinterpret_cast<const U4*>(in + i);
            u32 worst = 0;
#pragma unroll
            for (u32 s = 0; s < 16; ++s) {
                const u32 old = __hip_atomic_fetch_add(tab.col + rcx_byte_of(piece, s) * RCX_LANES, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                worst = worst > old ? worst : old;
            }
            if (rcx_any(worst >= 0xFFFFu)) {
                if (worst >= 0xFFFFu) {
                    for (u32 s = 0; s < 16; ++s) tab.set(rcx_byte_of(piece, s), tab.get(rcx_byte_of(piece, s)) - 1);
                    for (u32 s = 0; s < 16; ++s) rcx_static_count_one(tab, rcx_byte_of(piece, s));
                }
            }
        }
    }
    for (; i < to; ++i)
        if (i < len) rcx_static_count_one(tab, in[i]);
}
// The largest count of the lane's table: no squeeze can fire while it stays below 0xFFFF.
__device__ __forceinline__ u32 rcx_static_max_count(const StaticTable& tab)
{
    u32 m = 0;
    for (u32 q = 0; q < 256; ++q) {
        const u32 c = tab.get(q);
        m = m > c ? m : c;
    }
    return m;
}

// ===========================================================================
// Static encode, pass 1 (scan + scatter are shared with the adaptive coder)
// ===========================================================================
__global__ __launch_bounds__(64) void rcx_enc_static_k(const u8* __restrict__ src, u64 n, u32 block, u64 nblocks,
                                                       u8* __restrict__ slots, u64 slot, u32* __restrict__ sizes, u32* status,
                                                       const u32* __restrict__ only)
{
    __shared__ u32 lds[RCX_STATIC_LDS_DW];
    const u32 lane = threadIdx.x;
    const u64 blk = (u64)blockIdx.x * RCX_LANES + lane;
    bool live = blk < nblocks;
    // second pass behind rcx_enc_static3_k: only the blocks it marked (none, on ordinary data)
    if (only) {
        live = live && only[blk] != 0;
        if (!__any(live)) return;
    }
    const u64 at = live ? blk * (u64)block : 0;
    const u32 len = live ? (u32)((n - at) < (u64)block ? (n - at) : (u64)block) : 0u;
    const u8* in = src + at;
    StaticTable tab{lds + lane};

    // ---- count(), cpprcoder.h:543-571 ----
    for (u32 i = 0; i <= 256; ++i) tab.set(i, 0);
    const u32 maxlen = rcx_wave_max(len);
    const bool full = __all(live && len == block) && (block % 16u == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    // A count can only be 0xFFFF before its increment once 65535 earlier symbols exist, so the squeeze
    // test is not needed for the first 65535 symbols of a block.
    const u32 easy = maxlen < 65535u ? maxlen : 65535u;
    if (full) {
        const u32 easy16 = easy & ~15u;
        for (u32 i = 0; i < easy16; i += 16) {
            const U4 piece = *reinterpret_cast<const U4*>(in + i);
#pragma unroll
            for (u32 s = 0; s < 16; ++s) tab.inc(rcx_byte_of(piece, s));
        }
        for (u32 i = easy16; i < easy; ++i) tab.inc(in[i]);
    } else {
        for (u32 i = 0; i < easy; ++i)
            if (i < len) tab.inc(in[i]);
    }
    if (full) {
        const u32 from = (easy + 15u) & ~15u;
        rcx_static_count_checked<false>(tab, in, easy, from < maxlen ? from : maxlen, len);
        if (from < maxlen) rcx_static_count_checked<true>(tab, in, from, maxlen, len);
    } else {
        rcx_static_count_checked<false>(tab, in, easy, maxlen, len);
    }
    // The second rescale of count() (cpprcoder.h:561-570): more than 2^24 symbols (single streams only: blocks are
    // at most RCX_MAX_BLOCK).  Entry 0 is skipped, as the reference's loop starting at 1 does.
    if (len > (1u << 24)) {
        u32 sz = len, shift = 0;
        while (sz > (1u << 24)) {
            sz >>= 1;
            ++shift;
        }
        for (u32 q = 1; q < 256; ++q) {
            const u32 c = tab.get(q);
            tab.set(q, c != 0 ? ((c >> shift) | 1u) : 0u);
        }
    }

    // ---- header: u32 LE n + 256 u16 counts (cpprcoder.h:386-397, :604-619), then the cumulative table ----
    u8* wave_slots = slots + (u64)blockIdx.x * RCX_LANES * slot;
    EncLane enc;
    if (live) {
        enc.begin(wave_slots, lane * (u32)slot, (u32)slot, len);
        u32* hdr = reinterpret_cast<u32*>(wave_slots + lane * (u32)slot + 4);
        for (u32 i = 0; i < 256; i += 2) hdr[i >> 1] = (tab.get(i) & 0xFFFFu) | (tab.get(i + 1) << 16);
        enc.off += RCX_STATIC_HEADER - 4;
        enc.cap -= RCX_STATIC_HEADER - 4;
    } else {
        enc.idle(wave_slots);
    }
    enc.range = 0xFFFFFFFFu; // cpprcoder.h:382
    const u32 total = tab.accumulate();
    const DivEntry k = rcx_make_div_entry(total ? total : 1u);

    // ---- the coding loop, cpprcoder.h:400-436 ----
    if (full) {
        U4 cur = *reinterpret_cast<const U4*>(in);
        for (u32 i = 0; i < maxlen; i += 16) {
            U4 nxt = cur;
            if (i + 16 < maxlen) nxt = *reinterpret_cast<const U4*>(in + i + 16);
#pragma unroll
            for (u32 s = 0; s < 16; ++s) {
                const u32 b = rcx_byte_of(cur, s);
                const u32 lo = tab.get(b), hi = tab.get(b + 1);
                enc.template code<false, true>(lo, hi - lo, k);
            }
            cur = nxt;
        }
    } else {
        for (u32 i = 0; i < maxlen; ++i) {
            if (i < len) {
                const u32 b = in[i];
                const u32 lo = tab.get(b), hi = tab.get(b + 1);
                enc.template code<false, true>(lo, hi - lo, k);
            }
        }
    }

    if (live) {
        if (enc.low == 0xFFFFFFFFu) enc.acc += 1; // cpprcoder.h:439-443: bump the held byte, pending 0xFF -> 0x00
        const u32 bytes = enc.finish() + (RCX_STATIC_HEADER - 4);
        sizes[blk] = enc.overflow ? (u32)slot : bytes;
        if (enc.overflow) rcx_flag(status, RCX_ST_CAPACITY, blk);
    }
}

// ===========================================================================
// Static encode, pass 1, three waves per 64 blocks: the five-wave adaptive encoder (rcx_enc_mc5_k, rcx_oct.hpp)
// without its model waves -- the table does not change while coding.  wave 0 = arithmetic (EncLane::arith with
// full 32-bit multiplies), wave 1 = writer (StagedWriter: words through LDS rings), wave 2 = table lookups for
// the next chunk + the drain of the rings.  The histogram (cpprcoder.h:543-571) is counted by all three waves
// with LDS atomics for the first 65535 symbols of a block, where its 16-bit squeeze cannot fire.
// A carry through more output bytes than the rings keep back marks the block in `redo` for rcx_enc_static_k.
// ===========================================================================
#define RCX_ST3_THREADS 192
#define RCX_ST3_LDS_DW (RCX_STATIC_LDS_DW + 4 * RCX_MC_RING_U4 + RCX_MC5_RING2_DW + RCX_LANES + RCX_MC5_OUT_DW)

// WIDE: full 32-bit multiplies.  With a total below 256, t = range / total can exceed 24 bits; from 256 on it cannot,
// and cum and count never do (total <= 2^24), so the 24-bit multiplies of the adaptive coder are exact.
template <bool FULL, bool WIDE>
__device__ __forceinline__ void rcx_static3_pipeline(u32 wave, u32 lane, u32 len, u32 nchunks, const u8* in, const StaticTable& tab,
                                                     U4* ring, u32* ring2, EncLane& enc, const DivEntry& kdiv, StagedWriter& wr,
                                                     u32* out_pos, u32& drained, u8* payload, u32 cap, bool live)
{
    U4 piece_ahead;
    piece_ahead.x = piece_ahead.y = piece_ahead.z = piece_ahead.w = 0;
    if (FULL && wave == 2 && nchunks > 0) piece_ahead = *reinterpret_cast<const U4*>(in);
    for (u32 k = 0; k <= nchunks + 1; ++k) {
        if (wave == 0) {
            // ---- arithmetic: chunk k-1 ----
            if (k >= 1 && k <= nchunks) {
                const u32 i0 = (k - 1) * RCX_MC_CHUNK;
                const U4* rs = ring + ((k - 1) & 1u) * (RCX_MC_CHUNK * RCX_LANES) + lane;
                u32* ws2 = ring2 + ((k - 1) & 1u) * (RCX_MC_CHUNK * RCX_LANES) + lane;
                U4 e_next = rs[0];
                u32 rec_even = 0;
#pragma unroll
                for (u32 s = 0; s < RCX_MC_CHUNK; ++s) {
                    const U4 e = e_next;
                    if (s + 1 < RCX_MC_CHUNK) e_next = rs[(s + 1) * RCX_LANES];
                    u32 rec = 0;
                    if (FULL || i0 + s < len) rec = enc.template arith<WIDE>(e.x, e.w, kdiv); // cpprcoder.h:402-408
                    if ((s & 1u) == 0) rec_even = rec; // (two symbols' records per LDS instruction: see rcx_mc5_pipeline)
                    else {
                        ws2[(s - 1) * RCX_LANES] = rec_even;
                        ws2[s * RCX_LANES] = rec;
                    }
                }
            }
        } else if (wave == 1) {
            // ---- writer: chunk k-2 ----
            if (k >= 2) {
                const u32* rs2 = ring2 + ((k - 2) & 1u) * (RCX_MC_CHUNK * RCX_LANES) + lane;
                u32 ra_next = rs2[0], rb_next = rs2[RCX_LANES];
                wr.chunk_begins();
#pragma unroll
                for (u32 s = 0; s < RCX_MC_CHUNK; s += 2) {
                    const u32 ra = ra_next, rb = rb_next;
                    if (s + 2 < RCX_MC_CHUNK) ra_next = rs2[(s + 2) * RCX_LANES], rb_next = rs2[(s + 3) * RCX_LANES];
                    wr.emit(ra);
                    wr.emit(rb);
                }
                out_pos[lane] = wr.chunk_ends();
            }
        } else {
            // ---- drain (see rcx_mc5_pipeline: the reads here, the stores between the wait for this chunk's input and the
            // request for the next chunk's, no loop) ----
            const u32 drain_p = out_pos[lane];
            RcxU4Unaligned drain_piece;
            {
                const u32* w = wr.ring_lane + ((drained >> 2) % RCX_OUT_RING_WORDS) * RCX_LANES; // (drained is a multiple of 16)
                drain_piece.x = w[0];
                drain_piece.y = w[RCX_LANES];
                drain_piece.z = w[2 * RCX_LANES];
                drain_piece.w = w[3 * RCX_LANES];
            }
            auto drain_store = [&]() {
                const u32 limit = drain_p > RCX_OUT_MARGIN ? (drain_p - RCX_OUT_MARGIN) & ~15u : 0u;
                drained = rcx_drain_piece(wr.ring_lane, payload, drained, limit < cap ? limit : cap, live, drain_piece);
            };
            if (k >= nchunks) drain_store();
            // ---- lookups: chunk k ----
            if (k < nchunks) {
                const u32 i0 = k * RCX_MC_CHUNK;
                U4* ws = ring + (k & 1u) * (RCX_MC_CHUNK * RCX_LANES) + lane;
                U4 piece;
                if (FULL) {
                    piece = piece_ahead;
                    asm volatile("" ::"v"(piece.x), "v"(piece.y), "v"(piece.z), "v"(piece.w)); // (the input has arrived before a store is issued)
                }
                drain_store();
                if (FULL) {
                    if (k + 1 < nchunks) piece_ahead = *reinterpret_cast<const U4*>(in + i0 + RCX_MC_CHUNK);
                } else {
                    u32 w[4] = {0, 0, 0, 0};
                    for (u32 s = 0; s < RCX_MC_CHUNK; ++s)
                        if (i0 + s < len) w[s >> 2] |= (u32)in[i0 + s] << (8 * (s & 3));
                    piece.x = w[0];
                    piece.y = w[1];
                    piece.z = w[2];
                    piece.w = w[3];
                }
                u32 b = rcx_byte_of(piece, 0);
                u32 lo = tab.get(b), hi = tab.get(b + 1);
#pragma unroll
                for (u32 s = 0; s < RCX_MC_CHUNK; ++s) {
                    const u32 lo_s = lo, hi_s = hi;
                    if (s + 1 < RCX_MC_CHUNK) {
                        b = rcx_byte_of(piece, s + 1);
                        lo = tab.get(b);
                        hi = tab.get(b + 1);
                    }
                    u32* e = reinterpret_cast<u32*>(&ws[s * RCX_LANES]);
                    e[0] = lo_s;        // cum
                    e[3] = hi_s - lo_s; // count
                }
            }
        }
        rcx_lds_barrier();
    }
}

__global__ __launch_bounds__(RCX_ST3_THREADS) void rcx_enc_static3_k(const u8* __restrict__ src, u64 n, u32 block, u64 nblocks,
                                                                    u8* __restrict__ slots, u64 slot, u32* __restrict__ sizes,
                                                                    u32* status, u32* __restrict__ redo, u32 lanes_used)
{
    __shared__ __attribute__((aligned(16))) u32 lds[RCX_ST3_LDS_DW];
    const u32 lane = threadIdx.x & 63u;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool in_use = lane < lanes_used; // see rcx_enc_mc5_k
    const u64 blk = in_use ? (u64)blockIdx.x * lanes_used + lane : nblocks;
    const bool live = blk < nblocks;
    const u64 at = live ? blk * (u64)block : 0;
    const u32 len = live ? (u32)((n - at) < (u64)block ? (n - at) : (u64)block) : 0u;
    const u8* in = src + at;
    StaticTable tab{lds + lane};
    U4* ring = reinterpret_cast<U4*>(lds + RCX_STATIC_LDS_DW); // 16-byte aligned: RCX_STATIC_LDS_DW = 257 * 64 dwords
    u32* ring2 = reinterpret_cast<u32*>(ring + RCX_MC_RING_U4);
    u32* final_low = ring2 + RCX_MC5_RING2_DW;
    u32* out_ring = final_low + RCX_LANES;
    u32* out_dummy = out_ring + RCX_OUT_RING_WORDS * RCX_LANES;
    u32* out_pos = out_dummy + RCX_LANES;
    u32* out_drained = out_pos + RCX_LANES;

    const u32 maxlen = rcx_wave_max(len);
    const bool full = __all(!in_use || (live && len == block)) && (block % 16u == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    const u32 nchunks = (maxlen + RCX_MC_CHUNK - 1) / RCX_MC_CHUNK;

    // ---- count(), cpprcoder.h:543-571 ----
    for (u32 i = wave; i <= 256; i += 3) tab.set(i, 0);
    {
        U4 z; // the ring entries' two middle dwords stay 0: the arithmetic wave adds x + y + z
        z.x = z.y = z.z = z.w = 0;
        for (u32 i = wave; i < 2 * RCX_MC_CHUNK; i += 3) ring[i * RCX_LANES + lane] = z;
    }
    rcx_lds_barrier();
    // a count can only be 0xFFFF before its increment once 65535 earlier symbols exist: no squeeze test before
    const u32 easy = maxlen < 65535u ? maxlen : 65535u;
    const u32 easy16 = full ? easy & ~15u : 0u;
    for (u32 i = 16 * wave; i < easy16; i += 48) {
        const U4 piece = *reinterpret_cast<const U4*>(in + i);
#pragma unroll
        for (u32 s = 0; s < 16; ++s) tab.inc(rcx_byte_of(piece, s));
    }
    rcx_lds_barrier();
    if (easy < maxlen) {
        // Past symbol 65535.  If no count can reach 0xFFFF before the block ends (largest count so far + the
        // symbols left: every wave computes the same answer from the same table) the three waves go on as
        // before; otherwise the order matters and wave 0 counts the rest alone, in order, with the test.
        const u32 mx = rcx_static_max_count(tab);
        const bool calm = __all(mx + (len > easy16 ? len - easy16 : 0u) < 0xFFFFu);
        rcx_lds_barrier(); // (wave 0 may not start changing the table while another wave still reads it)
        if (wave == 0) {
            for (u32 i = easy16; i < easy; ++i)
                if (i < len) tab.inc(in[i]);
        }
        if (calm && full) {
            const u32 from = (easy + 15u) & ~15u; // 65536
            const u32 to16 = maxlen & ~15u;
            if (wave == 0)
                for (u32 i = easy; i < from && i < maxlen; ++i) tab.inc(in[i]);
            for (u32 i = from + 16 * wave; i < to16; i += 48) {
                const U4 piece = *reinterpret_cast<const U4*>(in + i);
#pragma unroll
                for (u32 s = 0; s < 16; ++s) tab.inc(rcx_byte_of(piece, s));
            }
            if (wave == 0)
                for (u32 i = (to16 > from ? to16 : from); i < maxlen; ++i) tab.inc(in[i]);
        } else if (wave == 0) {
            if (full) {
                const u32 from = (easy + 15u) & ~15u;
                rcx_static_count_checked<false>(tab, in, easy, from < maxlen ? from : maxlen, len);
                if (from < maxlen) rcx_static_count_checked<true>(tab, in, from, maxlen, len);
            } else {
                rcx_static_count_checked<false>(tab, in, easy, maxlen, len);
            }
        }
    } else if (wave == 0) {
        for (u32 i = easy16; i < easy; ++i)
            if (i < len) tab.inc(in[i]);
    }
    rcx_lds_barrier();

    // ---- header: u32 LE n + 256 u16 counts (cpprcoder.h:386-397, :604-619) ----
    u8* wave_slots = slots + (u64)blockIdx.x * lanes_used * slot;
    EncLane enc;
    enc.idle(wave_slots);
    StagedWriter wr;
    wr.begin(out_ring, out_dummy, lane);
    u32 drained = 0;
    if (wave == 1) {
        if (live) {
            enc.begin(wave_slots, lane * (u32)slot, (u32)slot, len);
            u32* hdr = reinterpret_cast<u32*>(wave_slots + lane * (u32)slot + 4);
            for (u32 i = 0; i < 256; i += 2) hdr[i >> 1] = (tab.get(i) & 0xFFFFu) | (tab.get(i + 1) << 16);
            enc.off += RCX_STATIC_HEADER - 4;
            enc.cap -= RCX_STATIC_HEADER - 4;
        }
        out_pos[lane] = 0;
    }
    rcx_lds_barrier();
    if (wave == 2) (void)tab.accumulate(); // cpprcoder.h:573-583; entry 256 = total
    rcx_lds_barrier();
    const u32 total = tab.get(256);
    const DivEntry k = rcx_make_div_entry(total ? total : 1u);
    enc.range = 0xFFFFFFFFu; // cpprcoder.h:382

    u8* payload = wave_slots + (u64)lane * slot + RCX_STATIC_HEADER;
    const u32 cap = (((u32)slot - 4) & ~3u) - (RCX_STATIC_HEADER - 4);
    const bool narrow = __all(total >= 256u); // every wave sees the same table
    if (full && narrow) rcx_static3_pipeline<true, false>(wave, lane, len, nchunks, in, tab, ring, ring2, enc, k, wr, out_pos, drained, payload, cap, live);
    else if (full) rcx_static3_pipeline<true, true>(wave, lane, len, nchunks, in, tab, ring, ring2, enc, k, wr, out_pos, drained, payload, cap, live);
    else rcx_static3_pipeline<false, true>(wave, lane, len, nchunks, in, tab, ring, ring2, enc, k, wr, out_pos, drained, payload, cap, live);

    if (wave == 0) final_low[lane] = enc.low;
    if (wave == 2) {
        out_drained[lane] = drained;
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    }
    rcx_lds_barrier();
    if (wave == 1 && live) {
        u32 at2 = out_drained[lane];
        const u32 flushed = wr.finish(enc);
        const u32 end = flushed < cap ? flushed : cap;
        for (; at2 < end; at2 += 4) *reinterpret_cast<u32*>(payload + at2) = wr.ring_lane[((at2 >> 2) % RCX_OUT_RING_WORDS) * RCX_LANES];
        enc.low = final_low[lane];
        if (enc.low == 0xFFFFFFFFu) enc.acc += 1; // cpprcoder.h:439-443
        const u32 bytes = enc.finish() + (RCX_STATIC_HEADER - 4);
        sizes[blk] = enc.overflow ? (u32)slot : bytes;
        if (enc.overflow) rcx_flag(status, RCX_ST_CAPACITY, blk);
        redo[blk] = (wr.redo != 0 && !enc.overflow) ? 1u : 0u;
    } else if (wave == 1 && blk < nblocks) {
        redo[blk] = 0;
    }
}

// ===========================================================================
// Static decode
// ===========================================================================
// STREAM = the single-stream entry point: one block whose symbol count n the host took from the header;
// track[0] = first symbol whose renormalisation ran out of input (cpprcoder.h:506-509), or 0xFFFFFFFF.
template <bool STREAM>
__global__ __launch_bounds__(64) void rcx_dec_static_k(const u8* __restrict__ comp, u64 comp_size, const u64* __restrict__ offsets, u64 nblocks,
                                                       u32 block, u64 n, u8* __restrict__ dst, u32* status, u32* track,
                                                       const u32* __restrict__ only)
{
    __shared__ u32 lds[RCX_STATIC_LDS_DW + RCX_RING_DW * RCX_LANES];
    const u32 lane = threadIdx.x;
    const u64 blk = (u64)blockIdx.x * RCX_LANES + lane;
    bool live = blk < nblocks;
    // second pass behind rcx_dec_static_quad_k: only the blocks it marked (none, on valid input)
    if (only) {
        live = live && only[blk] != 0;
        if (!__any(live)) return;
    }
    const u64 at = live ? blk * (u64)block : 0;
    u32 len = live ? (u32)((n - at) < (u64)block ? (n - at) : (u64)block) : 0u;
    StaticTable tab{lds + lane};
    u32* ring_col = lds + RCX_STATIC_LDS_DW + lane;

    DecLane dec;
    u64 stream_len = 0;
    u32 total = 1;
    for (u32 i = 0; i <= 256; ++i) tab.set(i, 0);
    if (live) {
        const u64 s0 = offsets[blk], s1 = offsets[blk + 1];
        stream_len = s1 - s0;
        const u8* s = comp + s0;
        // cpprcoder.h:474-493: at least the header, one more byte, then 5 bytes for the lead-in and low
        bool good = s1 >= s0 && s1 <= comp_size && stream_len >= RCX_STATIC_HEADER + 5; // (an offset table that points past the buffer is not followed)
        if (good) {
            const u32 declared = (u32)s[0] | ((u32)s[1] << 8) | ((u32)s[2] << 16) | ((u32)s[3] << 24);
            good = STREAM || declared == len;
        }
        if (good) {
            for (u32 i = 0; i < 256; ++i) tab.set(i, (u32)s[4 + 2 * i] | ((u32)s[5 + 2 * i] << 8)); // :585-602
            total = tab.accumulate();
            good = total != 0; // the reference would divide by zero
        }
        if (good) {
            // DecLane::begin expects 4 size bytes + 4 bytes of low; the static stream has its lead-in byte
            // in between (low = bytes[1..4], cpprcoder.h:494-498): start it 3 bytes early and fix low up.
            const u8* h = s + RCX_STATIC_HEADER - 3;
            dec.begin(h, comp + s1, ring_col);
            dec.low = ((u32)h[4] << 24) | ((u32)h[5] << 16) | ((u32)h[6] << 8) | (u32)h[7];
            dec.range = 0xFFFFFFFFu;
        } else {
            if (!STREAM) rcx_flag(status, RCX_ST_CORRUPT, blk);
            live = false;
            len = 0;
        }
    }
    if (!live) {
        dec.idle(comp, ring_col);
        total = 1;
    }
    const DivEntry k = rcx_make_div_entry(total);
    // cum[16], cum[32], ..., cum[240] never change: keep them in registers for the first search level
    u32 coarse[15];
#pragma unroll
    for (u32 q = 0; q < 15; ++q) coarse[q] = tab.get(16 * (q + 1));

    const u32 maxlen = rcx_wave_max(len);
    const bool full = !STREAM && __all(live && len == block) && (block % 16u == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0);
    u8* out = dst + at;
    bool bad = false;
    u32 short_at = 0xFFFFFFFFu;

    // One symbol, cpprcoder.h:500-517.  find() (:521-535) returns the number of entries cum[1..255] that
    // are <= target (the table is non-decreasing), counted here in two levels of 15 probes.
#define RCX_STATIC_SYMBOL(SYM)                                                                   \
    {                                                                                            \
        const u32 t_ = rcx_div(dec.range, k);                                                    \
        /* target = low / t, exact: f32 estimate (|error| < 1 for quotients < 2^21) + correction */ \
        u32 q_ = (u32)((float)dec.low * rcx_rcp((float)t_));                                     \
        {                                                                                        \
            const u64 prod_ = (u64)q_ * t_; /* t can be anything up to 2^32-1 here: 64-bit check */ \
            if (prod_ > dec.low) q_ -= 1;                                                        \
            else if (dec.low - prod_ >= t_) q_ += 1;                                             \
        }                                                                                        \
        if (dec.low >= total * t_) q_ = 0xFFFFFFFFu; /* corrupt: past the table, find() says 255 */ \
        u32 chunk_ = 0;                                                                          \
        _Pragma("unroll") for (u32 q = 0; q < 15; ++q) chunk_ += coarse[q] <= q_ ? 1u : 0u;      \
        u32 c_ = chunk_ * 16;                                                                    \
        u32 fine_ = 0;                                                                           \
        _Pragma("unroll") for (u32 q = 1; q < 16; ++q) fine_ += tab.get(c_ + q) <= q_ ? 1u : 0u; \
        c_ += fine_;                                                                             \
        if (c_ > 255u) c_ = 255u;                                                                \
        const u32 lo_ = tab.get(c_), hi_ = tab.get(c_ + 1);                                      \
        dec.low -= lo_ * t_;                                                                     \
        dec.range = (hi_ - lo_) * t_;                                                            \
        if (dec.range == 0) { /* a symbol with count 0: corrupt; the reference runs dry and fails */ \
            dec.range = 1u << 31;                                                                \
            bad = true;                                                                          \
        }                                                                                        \
        dec.pull();                                                                              \
        (SYM) = c_;                                                                              \
    }

    if (full) {
        for (u32 i = 0; i < maxlen; i += 16) {
            u32 word[4] = {0, 0, 0, 0};
            dec.topup();
#pragma unroll
            for (u32 s = 0; s < 16; ++s) {
                u32 sym;
                RCX_STATIC_SYMBOL(sym);
                word[s >> 2] |= sym << (8 * (s & 3));
            }
            U4 o;
            o.x = word[0];
            o.y = word[1];
            o.z = word[2];
            o.w = word[3];
            *reinterpret_cast<U4*>(out + i) = o;
        }
    } else {
        for (u32 i = 0; i < maxlen; ++i) {
            if ((i & 15u) == 0) dec.topup();
            if (i < len) {
                u32 sym;
                RCX_STATIC_SYMBOL(sym);
                out[i] = (u8)sym;
                if (STREAM && short_at == 0xFFFFFFFFu && (bad || dec.taken() + (RCX_STATIC_HEADER - 3) > stream_len)) short_at = i;
            }
        }
    }
#undef RCX_STATIC_SYMBOL
    // cpprcoder.h:506-509: running out of input inside the renormalisation is a failure
    if (STREAM) {
        if (lane == 0) track[0] = live ? short_at : 0u;
    } else if (live && (bad || dec.taken() + (RCX_STATIC_HEADER - 3) > stream_len)) {
        rcx_flag(status, RCX_ST_CORRUPT, blk);
    }
}

// ===========================================================================
// Static decode, 4 lanes per block: rcx_dec_quad_k's machinery (rcx_oct.hpp) with less to do per
// symbol -- the table never changes.  The block's 256 cumulative counts cum[1..256] (cpprcoder.h:573-583)
// sit in LDS as 16 nodes of 16 entries in the same bank-conflict-free table groups; lane j keeps the
// upper bounds of its four nodes, cum[16(4j+1)] .. cum[16(4j+4)], in registers for the whole block.
// find() (cpprcoder.h:521-535) counts the entries cum[1..255] that are <= low/t; in the scaled domain
// (cum*t <= low: every product is <= total*t <= range < 2^32) that is round 1 = the node bounds that do not
// borrow in low - bound*t, round 2 = the same over the node's 16 entries, read as they are (they are already
// cumulative: no scan).  low - cum[c]*t is the unsigned minimum of the differences, and the new range
// (cum[c+1] - cum[c])*t is minimum - maximum (mod 2^32), as in the adaptive decoder.  No update.
// The reference normalises AFTER a symbol (cpprcoder.h:500-517); with range = 0xFFFFFFFF at the start that is
// the same as normalising before the next one, plus once after the last.
// A stream whose target runs past the table or that names a symbol of count 0 (damaged input) is marked in
// `redo` and decoded by rcx_dec_static_k, which reports it the way the reference fails.
// ===========================================================================
#define RCX_SQUAD_LDS_BYTES (4 * RCX_QUAD_GROUP_BYTES + RCX_QUAD_BLOCKS * RCX_QUAD_RING_BYTES)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rcx_dec_static_quad_k(const u8* __restrict__ comp, u64 comp_size, const u64* __restrict__ offsets,
                                                                    u64 nblocks, u32 block, u64 n, u8* __restrict__ dst,
                                                                    u32* status, u32* __restrict__ redo, u32 quads_used)
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

    const u32 group = 2u * (quad >> 3) + ((0x96u >> (quad & 7u)) & 1u), quarter = (quad & 7u) >> 1; // see rcx_dec_quad_k
    u8* mine = lds + group * RCX_QUAD_GROUP_BYTES + quarter * 64;
    U4* leaves = reinterpret_cast<U4*>(mine) + j;
    U4* parked = reinterpret_cast<U4*>(mine + 16 * 256);
    u32* block_ring = reinterpret_cast<u32*>(lds + 4 * RCX_QUAD_GROUP_BYTES + quad * RCX_QUAD_RING_BYTES);

    QuadInput in;
    u64 stream_len = 0;
    u32 U1 = 1, U2 = 2, U3 = 3, U4_ = 4, total = 4;
    if (live) {
        const u64 s0 = offsets[blk], s1 = offsets[blk + 1];
        stream_len = s1 - s0;
        const u8* s = comp + s0;
        // cpprcoder.h:474-493: at least the header, one more byte, then 5 bytes for the lead-in and low
        bool good = s1 >= s0 && s1 <= comp_size && stream_len >= RCX_STATIC_HEADER + 5; // (an offset table that points past the buffer is not followed)
        if (good) {
            const u32 declared = (u32)s[0] | ((u32)s[1] << 8) | ((u32)s[2] << 16) | ((u32)s[3] << 24);
            good = declared == len;
        }
        if (good) {
            // cpprcoder.h:585-602 + :573-583: lane j takes the counts of symbols 64j .. 64j+63
            const u8* cs = s + 4 + 128 * j;
            u32 mysum = 0;
            for (u32 i = 0; i < 64; ++i) mysum += (u32)cs[2 * i] | ((u32)cs[2 * i + 1] << 8);
            u32 run = rcx_quad_excl_scan(mysum, (j & 1u) ? ~0u : 0u, (j & 2u) ? ~0u : 0u);
            for (u32 i = 0; i < 64; ++i) {
                run += (u32)cs[2 * i] | ((u32)cs[2 * i + 1] << 8);
                // entry e = 64j + i + 1 = cum[e] lives in node (e-1)/16 at position (e-1)%16
                reinterpret_cast<u32*>(mine + (4 * j + (i >> 4)) * 256)[i & 15u] = run;
                if (i == 15) U1 = run;
                if (i == 31) U2 = run;
                if (i == 47) U3 = run;
            }
            U4_ = run;
            total = rcx_dpp<0xFF>(U4_); // quad_perm [3,3,3,3]: cum[256]
            good = total != 0;          // the reference would divide by zero
        }
        if (good) {
            // QuadInput::begin expects 4 size bytes + 4 bytes of low; the static stream has its lead-in byte in
            // between (low = bytes[1..4] after the header, cpprcoder.h:494-498): start it 3 bytes early
            in.begin(s + RCX_STATIC_HEADER - 3, comp + s1, block_ring, parked + 3);
            in.range = 0xFFFFFFFFu;
        } else {
            if (j == 0 && in_use) rcx_flag(status, RCX_ST_CORRUPT, blk);
            live = false;
            len = 0;
        }
    }
    if (!live) {
        in.idle(comp, block_ring, parked + 3);
        U1 = 1, U2 = 2, U3 = 3, U4_ = 4, total = 4;
        U4 v;
        v.x = v.y = v.z = v.w = 4;
        for (u32 q = 0; q < 16; ++q) leaves[q * 16] = v;
    }
    {
        U4 v; // the scratch row ("node 16", reached only by a target past the table)
        v.x = v.y = v.z = v.w = 0;
        leaves[16 * 16] = v;
    }
    const DivEntry k = rcx_make_div_entry(total);
    const u64 kadd = k.add;

    const u32 maxlen = rcx_wave_max(len);
    const bool full = __all(live && len == block) && (block % 16u == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0);
    u8* out = dst + at;
    const bool leader = live && in_use && j == 0;
    const u32 leaves_lds = (u32)reinterpret_cast<uintptr_t>(leaves);
    const u32 ring_lds = (u32)reinterpret_cast<uintptr_t>(block_ring);
    u32 worst_node = 0;            // 16 = a target past the table
    u32 least_range = 0xFFFFFFFFu; // 0 = a symbol of count 0

#define RCX_QP1 "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define RCX_QP2 "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    // One symbol (instruction sequences as in rcx_dec_quad_k: no compare result or DPP source is used before two
    // other instructions have been issued).  Every lane of the quad ORs the symbol into WORD at bit SHIFT.
    // (as in rcx_dec_quad_k, what the coder state does not need -- the symbol's byte, the watch on the smallest range -- is
    // made by the NEXT symbol behind its leaf read: HP = 1, PWORD / PSHIFT = the earlier symbol's word and bit position;
    // RCX_SQUAD_FINISH makes the last one)
    u32 p_nd_ = 0, p_nb_ = 0, p_rg_ = 0xFFFFFFFFu;
#define RCX_SQ_PREV_0
#define RCX_SQ_PREV_1 "\n\tv_lshl_add_u32 %[ps], %[pnd], 4, %[pnb]\n\t"                                             \
                      "v_min_u32 %[lr], %[lr], %[prg]\n\t"                                                           \
                      "v_lshl_or_b32 %[pword], %[ps], %[psh], %[pword]"
#define RCX_SQUAD_FINISH(WORD, SHIFT)                                                                      \
    {                                                                                                      \
        (WORD) |= ((p_nd_ << 4) + p_nb_) << (SHIFT);                                                       \
        least_range = least_range < p_rg_ ? least_range : p_rg_;                                           \
    }
#define RCX_SQUAD_SYMBOL(HP, PWORD, PSHIFT)                                                                \
    {                                                                                                      \
        const u32 k8_ = rcx_clz(in.range) & 0x18u; /* cpprcoder.h:511-516, for the previous symbol */      \
        in.low = (u32)((((u64)in.low << 32) | in.n4) << k8_ >> 32);                                        \
        in.range <<= k8_;                                                                                  \
        const u32 t_ = (u32)(((u64)in.range * k.mul + kadd) >> 32) >> (k.shift & 31u); /* :502 */          \
        const u32 a1_ = U1 * t_, a2_ = U2 * t_, a3_ = U3 * t_, a4_ = U4_ * t_;                             \
        u32 node_, rem_, ro_, la_, x1_, x2_, x3_, x4_;                                                     \
        u64 c1_, c2_, c3_, c4_;                                                                            \
        /* (ordered as in rcx_dec_quad_k: the node index first -- the leaf read waits for it --, the next symbol's stream  \
           bytes asked for before it, the remainder's steps across the quad and the stream bytes' extraction behind it) */ \
        asm volatile("v_sub_co_u32_e64 %[x1], %[c1], %[low], %[a1]\n\t"                                    \
                     "v_sub_co_u32_e64 %[x2], %[c2], %[low], %[a2]\n\t"                                    \
                     "v_sub_co_u32_e64 %[x3], %[c3], %[low], %[a3]\n\t"                                    \
                     "v_sub_co_u32_e64 %[x4], %[c4], %[low], %[a4]\n\t"                                    \
                     "v_subb_co_u32_e64 %[nd], %[c1], 4, 0, %[c1]\n\t"                                     \
                     "v_subb_co_u32_e64 %[nd], %[c2], %[nd], 0, %[c2]\n\t"                                 \
                     "v_subb_co_u32_e64 %[nd], %[c3], %[nd], 0, %[c3]\n\t"                                 \
                     "v_subb_co_u32_e64 %[nd], %[c4], %[nd], 0, %[c4]\n\t"                                 \
                     "v_add_u32 %[bp], %[bp], %[k8]\n\t"                                                   \
                     "v_bfe_u32 %[ro], %[bp], 5, 5\n\t"                                                    \
                     "v_lshl_add_u32 %[ro], %[ro], 2, %[rb]"                                                \
                     : [nd] "=&v"(node_), [ro] "=&v"(ro_), [bp] "+v"(in.bp8),                               \
                       [x1] "=&v"(x1_), [x2] "=&v"(x2_), [x3] "=&v"(x3_), [x4] "=&v"(x4_),                 \
                       [c1] "=&s"(c1_), [c2] "=&s"(c2_), [c3] "=&s"(c3_), [c4] "=&s"(c4_)                  \
                     : [low] "v"(in.low), [a1] "v"(a1_), [a2] "v"(a2_), [a3] "v"(a3_), [a4] "v"(a4_),      \
                       [k8] "v"(k8_), [rb] "v"(ring_lds));                                                 \
        {                                                                                                  \
            const RcxLdsU32* at_ = reinterpret_cast<const RcxLdsU32*>(ro_); /* for the next symbol */      \
            in.w0 = at_[0];                                                                                \
            in.w1 = at_[1];                                                                                \
        }                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        asm volatile("v_add_u32_dpp %[nd], %[nd], %[nd] " RCX_QP1                                          \
                     "v_min3_u32 %[rm], %[x1], %[x2], %[x3]\n\t"                                           \
                     "v_min3_u32 %[rm], %[rm], %[x4], %[low]\n\t"                                          \
                     "v_add_u32_dpp %[nd], %[nd], %[nd] " RCX_QP2                                          \
                     "v_lshl_add_u32 %[la], %[nd], 8, %[lvb]"                                              \
                     : [nd] "+v"(node_), [rm] "=&v"(rem_), [la] "=&v"(la_)                                  \
                     : [low] "v"(in.low), [x1] "v"(x1_), [x2] "v"(x2_), [x3] "v"(x3_), [x4] "v"(x4_),      \
                       [lvb] "v"(leaves_lds));                                                             \
        const RcxV4 l_ = *reinterpret_cast<const RcxLdsV4*>(la_);                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        asm volatile("v_min_u32_dpp %[rm], %[rm], %[rm] " RCX_QP1                                          \
                     "v_max_u32 %[wn], %[wn], %[nd]"                                                       \
                     : [rm] "+v"(rem_), [wn] "+v"(worst_node)                                               \
                     : [nd] "v"(node_));                                                                   \
        u32 ps_;                                                                                           \
        asm volatile("v_alignbit_b32 %[n4], %[w1], %[w0], %[bp]\n\t" /* the 4 bytes at bp8 ... */           \
                     "v_min_u32_dpp %[rm], %[rm], %[rm] " RCX_QP2                                          \
                     "v_perm_b32 %[n4], %[n4], %[n4], %[swap]"      /* ... first one on top */              \
                     RCX_SQ_PREV_##HP                                                                      \
                     : [rm] "+v"(rem_), [n4] "=&v"(in.n4), [ps] "=&v"(ps_), [lr] "+v"(least_range), [pword] "+v"(PWORD) \
                     : [w0] "v"(in.w0), [w1] "v"(in.w1), [bp] "v"(in.bp8), [swap] "s"(0x00010203u),        \
                       [pnd] "v"(p_nd_), [pnb] "v"(p_nb_), [prg] "v"(p_rg_), [psh] "n"(PSHIFT));           \
        const u32 q1_ = l_.x * t_, q2_ = l_.y * t_, q3_ = l_.z * t_, q4_ = l_.w * t_;                      \
        u32 lo_, rg_, nb_, hi_, y1_, y2_, y3_, y4_;                                                        \
        asm volatile("v_sub_co_u32_e64 %[y1], %[c1], %[low], %[q1]\n\t"                                    \
                     "v_sub_co_u32_e64 %[y2], %[c2], %[low], %[q2]\n\t"                                    \
                     "v_sub_co_u32_e64 %[y3], %[c3], %[low], %[q3]\n\t"                                    \
                     "v_sub_co_u32_e64 %[y4], %[c4], %[low], %[q4]\n\t"                                    \
                     "v_subb_co_u32_e64 %[nb], %[c1], 4, 0, %[c1]\n\t"                                     \
                     "v_min3_u32 %[lo], %[y1], %[y2], %[y3]\n\t"                                           \
                     "v_subb_co_u32_e64 %[nb], %[c2], %[nb], 0, %[c2]\n\t"                                 \
                     "v_max3_u32 %[hi], %[y1], %[y2], %[y3]\n\t"                                           \
                     "v_subb_co_u32_e64 %[nb], %[c3], %[nb], 0, %[c3]\n\t"                                 \
                     "v_min3_u32 %[lo], %[lo], %[y4], %[rem]\n\t"                                          \
                     "v_subb_co_u32_e64 %[nb], %[c4], %[nb], 0, %[c4]\n\t"                                 \
                     "v_max_u32 %[hi], %[hi], %[y4]\n\t"                                                   \
                     "v_min_u32_dpp %[lo], %[lo], %[lo] " RCX_QP1                                          \
                     "v_add_u32_dpp %[nb], %[nb], %[nb] " RCX_QP1                                          \
                     "v_max_u32_dpp %[hi], %[hi], %[hi] " RCX_QP1                                          \
                     "v_min_u32_dpp %[lo], %[lo], %[lo] " RCX_QP2                                          \
                     "v_add_u32_dpp %[nb], %[nb], %[nb] " RCX_QP2                                          \
                     "v_max_u32_dpp %[hi], %[hi], %[hi] " RCX_QP2                                          \
                     "v_sub_u32 %[rg], %[lo], %[hi]"                                                       \
                     : [lo] "=&v"(lo_), [rg] "=&v"(rg_), [nb] "=&v"(nb_), [hi] "=&v"(hi_), [y1] "=&v"(y1_), \
                       [y2] "=&v"(y2_), [y3] "=&v"(y3_), [y4] "=&v"(y4_), [c1] "=&s"(c1_), [c2] "=&s"(c2_),  \
                       [c3] "=&s"(c3_), [c4] "=&s"(c4_)                                                    \
                     : [low] "v"(in.low), [q1] "v"(q1_), [q2] "v"(q2_), [q3] "v"(q3_), [q4] "v"(q4_),      \
                       [rem] "v"(rem_));                                                                   \
        in.low = lo_;   /* :504 */                                                                         \
        in.range = rg_; /* :505 */                                                                         \
        p_nd_ = node_;  /* the symbol = node << 4 | nb, in all four lanes */                                \
        p_nb_ = nb_;                                                                                       \
        p_rg_ = rg_;                                                                                       \
    }

    if (full) {
        U4 o_last;
        o_last.x = o_last.y = o_last.z = o_last.w = 0;
        for (u32 i0 = 0; i0 < maxlen; i0 += 16) {
            in.topup();
            const u32 g = (i0 >> 4) & 3u;
            if (g == 0 && i0 != 0 && leader) { // see rcx_dec_quad_k: the stores follow the top-up
                U4* o4 = reinterpret_cast<U4*>(out + (i0 - 64));
                const U4 p0 = parked[0], p1 = parked[1], p2 = parked[2];
                o4[0] = p0;
                o4[1] = p1;
                o4[2] = p2;
                o4[3] = o_last;
            }
            u32 w0_ = 0, w1_ = 0, w2_ = 0, w3_ = 0;
            RCX_SQUAD_SYMBOL(0, w0_, 0) RCX_SQUAD_SYMBOL(1, w0_, 0) RCX_SQUAD_SYMBOL(1, w0_, 8) RCX_SQUAD_SYMBOL(1, w0_, 16)
            RCX_SQUAD_SYMBOL(1, w0_, 24) RCX_SQUAD_SYMBOL(1, w1_, 0) RCX_SQUAD_SYMBOL(1, w1_, 8) RCX_SQUAD_SYMBOL(1, w1_, 16)
            RCX_SQUAD_SYMBOL(1, w1_, 24) RCX_SQUAD_SYMBOL(1, w2_, 0) RCX_SQUAD_SYMBOL(1, w2_, 8) RCX_SQUAD_SYMBOL(1, w2_, 16)
            RCX_SQUAD_SYMBOL(1, w2_, 24) RCX_SQUAD_SYMBOL(1, w3_, 0) RCX_SQUAD_SYMBOL(1, w3_, 8) RCX_SQUAD_SYMBOL(1, w3_, 16)
            RCX_SQUAD_FINISH(w3_, 24)
            U4 o;
            o.x = w0_;
            o.y = w1_;
            o.z = w2_;
            o.w = w3_;
            if (g == 3) o_last = o;
            else parked[g] = o;
        }
        if (leader && maxlen != 0) {
            const u32 groups = ((maxlen - 1) >> 4 & 3u) + 1;
            U4* o4 = reinterpret_cast<U4*>(out + ((maxlen - 1) & ~63u));
            o4[0] = parked[0];
            if (groups > 1) o4[1] = parked[1];
            if (groups > 2) o4[2] = parked[2];
            if (groups > 3) o4[3] = o_last;
        }
    } else {
        for (u32 i = 0; i < maxlen; ++i) {
            if ((i & 15u) == 0) in.topup();
            if (i < len) { // the 4 lanes of a quad agree
                u32 sym = 0;
                RCX_SQUAD_SYMBOL(0, sym, 0);
                RCX_SQUAD_FINISH(sym, 0);
                if (leader) out[i] = (u8)sym;
            }
        }
    }
#undef RCX_SQUAD_SYMBOL
#undef RCX_SQUAD_FINISH
#undef RCX_SQ_PREV_0
#undef RCX_SQ_PREV_1
#undef RCX_QP1
#undef RCX_QP2
    // the normalisation after the last symbol (it decides whether the input was long enough, cpprcoder.h:506-509)
    in.bp8 += rcx_clz(in.range) & 0x18u;
    const bool marked = live && (worst_node >= 16u || least_range == 0);
    if (leader && !marked && in.taken() + (RCX_STATIC_HEADER - 3) > stream_len) rcx_flag(status, RCX_ST_CORRUPT, blk);
    if (leader) redo[blk] = marked ? 1u : 0u;
    else if (j == 0 && in_use && blk < nblocks) redo[blk] = 0;
}
