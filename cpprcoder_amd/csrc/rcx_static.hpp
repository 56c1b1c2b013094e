// rcx_static.hpp -- the static (two-pass) range coder RangeEncoder<T> on gfx950, one lane per block.
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

// ===========================================================================
// Static encode, pass 1 (scan + scatter are shared with the adaptive coder)
// ===========================================================================
__global__ __launch_bounds__(64) void rcx_enc_static_k(const u8* __restrict__ src, u64 n, u32 block, u64 nblocks,
                                                       u8* __restrict__ slots, u64 slot, u32* __restrict__ sizes, u32* status)
{
    __shared__ u32 lds[RCX_STATIC_LDS_DW];
    const u32 lane = threadIdx.x;
    const u64 blk = (u64)blockIdx.x * RCX_LANES + lane;
    const bool live = blk < nblocks;
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
    for (u32 i = easy; i < maxlen; ++i) {
        if (i < len) {
            const u32 b = in[i];
            if (tab.get(b) >= 0xFFFFu) { // :549-555: every non-zero count becomes (c >> 1) | 1
                for (u32 q = 0; q < 256; ++q) {
                    const u32 c = tab.get(q);
                    if (c > 0) tab.set(q, (c >> 1) | 1u);
                }
            }
            tab.set(b, tab.get(b) + 1);
        }
    }
    // (the second rescale of count(), cpprcoder.h:561-570, needs n > 2^24: unreachable for RCX_MAX_BLOCK)

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
// Static decode
// ===========================================================================
// STREAM = the single-stream entry point: one block whose symbol count n the host took from the header;
// track[0] = first symbol whose renormalisation ran out of input (cpprcoder.h:506-509), or 0xFFFFFFFF.
template <bool STREAM>
__global__ __launch_bounds__(64) void rcx_dec_static_k(const u8* __restrict__ comp, const u64* __restrict__ offsets, u64 nblocks,
                                                       u32 block, u64 n, u8* __restrict__ dst, u32* status, u32* track)
{
    __shared__ u32 lds[RCX_STATIC_LDS_DW + RCX_RING_DW * RCX_LANES];
    const u32 lane = threadIdx.x;
    const u64 blk = (u64)blockIdx.x * RCX_LANES + lane;
    bool live = blk < nblocks;
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
        bool good = s1 >= s0 && stream_len >= RCX_STATIC_HEADER + 5;
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
