// rcx_lane.hpp -- the per-block ("one lane per block") coder steps.
//
// Everything a lane does for one symbol lives here, written so that the same
// source compiles (a) as __device__ code inside the gfx950 kernels and (b) as
// plain C++ for tests/sim/lane_sim.cpp, a host-side lane simulator the CPU test
// suite uses to check this arithmetic against the oracle before any GPU run.
// The simulator is test tooling: librcx.so contains only the device build.
//
// Semantics restated from the reference (all u32 with wraparound):
//   encoder step   cpprcoder.h:702-713 + normalize :764-802 + finish :744-762
//   decoder step   cpprcoder.h:900-917 + normalize :926-940
//   model          cpprcoder.h:1094-1243 (counts start at 1, +1 per symbol;
//                  results depend only on cum(c) = sum_{i<c} f[i], so the table
//                  is kept as a radix-4 tree of plain sums instead of the
//                  reference's 16 chunk prefixes)
// The table halving (cpprcoder.h:1138) needs total >= 2^24 and cannot happen
// for block <= RCX_MAX_BLOCK.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RCX_DEV __device__ __forceinline__
#else
#define RCX_DEV inline
#define RCX_HOST_SIM 1
#endif

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t s32;

struct alignas(16) U4 {
    u32 x, y, z, w;
};

// ---------------------------------------------------------------------------
// Model layout.  Per block: a 4-level radix-4 tree of plain sums.
//   level 3: 1 group  (4 nodes of 64 symbols)      group 0
//   level 2: 4 groups (16 nodes of 16 symbols)     groups 1..4
//   level 1: 16 groups (64 nodes of 4 symbols)     groups 5..20
//   level 0: 64 groups (256 counts)                groups 21..84
// A group is 16 bytes; groups are lane-interleaved in LDS (group g of lane l at
// (g*RCX_LANES + l) * 16) so that a ds_read_b128 of ANY per-lane group index
// is bank-conflict free.
// ---------------------------------------------------------------------------
#define RCX_LANES 64
#define RCX_G_L3 0
#define RCX_G_L2 1
#define RCX_G_L1 5
#define RCX_G_L0 21
#define RCX_GROUPS 85
#define RCX_STAGE 64 /* divisor-table entries staged per refill */

// One divisor-table entry for total = 256 + index:
//   floor(n / total) == (u32)(((u64)n * mul + add) >> 32) >> shift   for all n < 2^32
// (N-bit multiply-add division: round-up magic when it fits 32 bits, otherwise the
//  round-down magic with add = mul; powers of two use mul = add = 2^32-1.)
struct alignas(16) DivEntry {
    u32 mul, add, shift, total;
};

#if defined(RCX_HOST_SIM)
RCX_DEV u32 rcx_clz(u32 x) { return (u32)__builtin_clz(x); }
RCX_DEV u32 rcx_mul24(u32 a, u32 b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
RCX_DEV u32 rcx_perm(u32 hi, u32 lo, u32 sel)
{
    u64 v = ((u64)hi << 32) | lo;
    u32 out = 0;
    for (int i = 0; i < 4; ++i) out |= (u32)((v >> (8 * ((sel >> (8 * i)) & 7))) & 0xFF) << (8 * i);
    return out;
}
RCX_DEV u32 rcx_bswap(u32 x) { return __builtin_bswap32(x); }
RCX_DEV void rcx_lds_inc(u32* p) { *p += 1; }
RCX_DEV float rcx_rcp(float x) { return 1.0f / x; }
// coverage counters of the host simulator: [0] carries that left the register window,
// [1] bytes touched by those, [2] decoder symbols on the off-table (corrupt) path
extern uint64_t rcx_sim_counters[4];
#define RCX_SIM_COUNT(i, n) (rcx_sim_counters[i] += (n))
#else
#define RCX_SIM_COUNT(i, n) ((void)0)
RCX_DEV u32 rcx_clz(u32 x) { return (u32)__builtin_clz(x); }
RCX_DEV u32 rcx_mul24(u32 a, u32 b) { return __umul24(a, b); }
RCX_DEV u32 rcx_perm(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
RCX_DEV u32 rcx_bswap(u32 x) { return __builtin_bswap32(x); }
RCX_DEV void rcx_lds_inc(u32* p) { (void)__hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
RCX_DEV float rcx_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#endif

RCX_DEV u32 rcx_div(u32 n, const DivEntry& k) { return (u32)(((u64)n * k.mul + k.add) >> 32) >> k.shift; }

// The tree of one lane.  `base` already includes the lane offset.
struct Tree {
    U4* base;
    RCX_DEV U4 group(u32 g) const { return base[g * RCX_LANES]; }
    RCX_DEV void bump(u32 g, u32 p) const { rcx_lds_inc(reinterpret_cast<u32*>(&base[g * RCX_LANES]) + p); }
    // cpprcoder.h:1094-1132: every count 1.
    RCX_DEV void reset() const
    {
        U4 v;
        v.x = v.y = v.z = v.w = 64;
        base[0] = v;
        v.x = v.y = v.z = v.w = 16;
        for (u32 g = RCX_G_L2; g < RCX_G_L1; ++g) base[g * RCX_LANES] = v;
        v.x = v.y = v.z = v.w = 4;
        for (u32 g = RCX_G_L1; g < RCX_G_L0; ++g) base[g * RCX_LANES] = v;
        v.x = v.y = v.z = v.w = 1;
        for (u32 g = RCX_G_L0; g < RCX_GROUPS; ++g) base[g * RCX_LANES] = v;
    }
    // cpprcoder.h:1134-1177 without the (unreachable) halving: +1 on the path to the leaf.
    RCX_DEV void update(u32 c) const
    {
        bump(RCX_G_L3, c >> 6);
        bump(RCX_G_L2 + (c >> 6), (c >> 4) & 3);
        bump(RCX_G_L1 + (c >> 4), (c >> 2) & 3);
        bump(RCX_G_L0 + (c >> 2), c & 3);
    }
};

// sum of the first p entries of a group, p in 0..3
RCX_DEV u32 rcx_pre4(const U4& g, u32 p)
{
    u32 r = (p > 0) ? g.x : 0;
    r += (p > 1) ? g.y : 0;
    r += (p > 2) ? g.z : 0;
    return r;
}
RCX_DEV u32 rcx_sel4(const U4& g, u32 p)
{
    u32 lo = (p & 1) ? g.y : g.x;
    u32 hi = (p & 1) ? g.w : g.z;
    return (p & 2) ? hi : lo;
}

// ---------------------------------------------------------------------------
// Encoder lane.
//
// Output bytes are produced eagerly: every renormalisation byte is appended at
// once and a carry is added into the bytes already produced, which yields the
// same stream as the reference's held-byte + pending-0xFF counter
// (cpprcoder.h:767-800).  The newest 1..7 bytes live in `acc` (big-endian
// number, newest byte lowest) so that a carry is one integer add; only when it
// runs out of the register does it walk back through memory (rare).
// ---------------------------------------------------------------------------
struct EncLane {
    u32 low, range;
    u32 acc_lo, acc_hi; // pending output bytes, acc_hi is only non-zero between append and flush
    u32 nacc8;          // 8 * number of bytes in acc; 8..32 between steps
    u32 pos;            // payload bytes already stored
    u32 cap;            // payload capacity of the slot (bytes)
    u32 overflow;       // slot too small: output is dropped from here on
    u8* payload;        // slot + 4
    // TRACK only: where the reference's delayed writer (held byte + pending run,
    // cpprcoder.h:767-800) stands, to find the symbol at which a bounded sink fills.
    u32 trk_written;    // payload bytes the reference has passed to writeByte so far
    u32 trk_pending;    // its carry_ counter
    u32 trk_cap;        // writeByte calls that succeed (sink capacity - 4 header bytes)
    u32 trk_fail_at;    // first symbol whose normalize hits the full sink, or 0xFFFFFFFF

    RCX_DEV void begin(u8* slot, u32 slot_bytes, u32 declared)
    {
        // cpprcoder.h:678-695: u32 LE size, then the coder state; the reference's first
        // emitted byte is its initial buffer_ = 0, here already sitting in acc.
        slot[0] = (u8)declared;
        slot[1] = (u8)(declared >> 8);
        slot[2] = (u8)(declared >> 16);
        slot[3] = (u8)(declared >> 24);
        payload = slot + 4;
        cap = slot_bytes - 4;
        low = 0;
        range = 0xFFFFFF00u;
        acc_lo = 0;
        acc_hi = 0;
        nacc8 = 8;
        pos = 0;
        overflow = 0;
        trk_written = 0;
        trk_pending = 0;
        trk_cap = 0xFFFFFFFFu;
        trk_fail_at = 0xFFFFFFFFu;
    }

    // a lane without a block: keeps the arithmetic well-defined, never stores
    RCX_DEV void idle()
    {
        payload = nullptr;
        cap = 0;
        low = 0;
        range = 0xFFFFFF00u;
        acc_lo = 0;
        acc_hi = 0;
        nacc8 = 8;
        pos = 0;
        overflow = 0;
        trk_written = 0;
        trk_pending = 0;
        trk_cap = 0xFFFFFFFFu;
        trk_fail_at = 0xFFFFFFFFu;
    }

    // carry ran through every byte held in acc: continue in memory (cpprcoder.h:767-781)
    RCX_DEV void carry_into_memory()
    {
        u32 p = pos;
        RCX_SIM_COUNT(0, 1);
        while (p > 0) {
            --p;
            RCX_SIM_COUNT(1, 1);
            u8 v = (u8)(payload[p] + 1);
            payload[p] = v;
            if (v != 0) break;
        }
    }

    RCX_DEV void store4(u32 word_le)
    {
        if (pos + 4 <= cap) *reinterpret_cast<u32*>(payload + pos) = word_le;
        else overflow = 1;
        pos += 4;
    }

    // TRACK: replay cpprcoder.h:767-800 on counters only.  `moved` is low after the add,
    // k8 the renormalisation shift of this symbol.
    RCX_DEV void track(u32 index, u32 carry, u32 moved, u32 k8)
    {
        if (trk_fail_at != 0xFFFFFFFFu) return;
        if (carry && trk_pending > 0) { // :769-780 held+1, then pending-1 zero bytes
            if (trk_written + trk_pending > trk_cap) { trk_fail_at = index; return; }
            trk_written += trk_pending;
            trk_pending = 0;
        }
        for (u32 s = 0; s < k8; s += 8) { // :783-800
            if (((moved << s) >> 24) != 0xFFu) {
                if (trk_written + 1 + trk_pending > trk_cap) { trk_fail_at = index; return; }
                trk_written += 1 + trk_pending;
                trk_pending = 0;
            } else {
                trk_pending += 1;
            }
        }
    }
    // TRUE when finish() (cpprcoder.h:744-755) would run into the full sink
    RCX_DEV bool track_flush_fails() const { return trk_written + 1 + trk_pending > trk_cap; }

    template <bool TRACK = false, class TreeT>
    RCX_DEV void step(const TreeT& tree, u32 c, const DivEntry& k, u32 index = 0)
    {
        const U4 g3 = tree.group(RCX_G_L3);
        const U4 g2 = tree.group(RCX_G_L2 + (c >> 6));
        const U4 g1 = tree.group(RCX_G_L1 + (c >> 4));
        const U4 g0 = tree.group(RCX_G_L0 + (c >> 2));
        const u32 cum = rcx_pre4(g3, c >> 6) + rcx_pre4(g2, (c >> 4) & 3) + rcx_pre4(g1, (c >> 2) & 3) + rcx_pre4(g0, c & 3);
        const u32 f = rcx_sel4(g0, c & 3);

        const u32 t = rcx_div(range, k);          // cpprcoder.h:703
        u32 moved = low + rcx_mul24(cum, t);      // :706  (cum*t <= range < 2^32, both factors < 2^24)
        const u32 carry = moved < low ? 1u : 0u;
        range = rcx_mul24(f, t);                  // :707

        // :767-781 carry into the bytes produced so far
        acc_lo += carry;
        const u32 wrapped = 2u << (nacc8 - 1);    // 2^(nacc8) truncated to 32 bits (0 when 4 bytes are held)
        if (carry && acc_lo == wrapped) {
            acc_lo = 0;
            carry_into_memory();
        }

        // :783-800 renormalise: k8/8 bytes leave through the top of low
        const u32 k8 = rcx_clz(range) & 0x18u;
        if (TRACK) track(index, carry, moved, k8);
        const u64 pair = (((u64)acc_lo << 32) | moved) << k8; // {acc_lo, low} shifted together
        acc_hi = (u32)((u64)acc_lo >> (32 - k8));              // k8 == 0 -> acc_lo >> 32 == 0
        acc_lo = (u32)(pair >> 32);
        low = (u32)pair;
        range <<= k8;
        nacc8 += k8;

        if (nacc8 >= 40) { // 5..7 bytes held: store the 4 oldest, keep 1..3
            const u32 keep8 = nacc8 - 32;
            const u32 r = keep8 >> 3;
            store4(rcx_perm(acc_hi, acc_lo, 0x00010203u + rcx_perm(0u, r, 0u) /* r in every byte */));
            acc_lo &= (1u << keep8) - 1u;
            acc_hi = 0;
            nacc8 = keep8;
        }
        tree.update(c); // :712
    }

    // cpprcoder.h:744-762: the held bytes, then low big-endian.  Returns the stream size.
    RCX_DEV u32 finish()
    {
        u32 n = nacc8 >> 3;
        for (u32 i = 0; i < n; ++i) {
            u8 b = (u8)(acc_lo >> (8 * (n - 1 - i)));
            if (pos < cap) payload[pos] = b;
            else overflow = 1;
            ++pos;
        }
        for (u32 i = 0; i < 4; ++i) {
            u8 b = (u8)(low >> (24 - 8 * i));
            if (pos < cap) payload[pos] = b;
            else overflow = 1;
            ++pos;
        }
        return pos + 4;
    }
};

// ---------------------------------------------------------------------------
// Decoder lane.
// ---------------------------------------------------------------------------
struct DecLane {
    u32 low, range;
    u64 win;            // upcoming stream bytes, left-aligned (next byte on top)
    u32 navail8;        // 8 * bytes in win
    const u8* next;     // next aligned dword to load
    const u8* end;      // one past the block's stream
    u64 taken;          // stream bytes consumed by normalize (for the truncation check)
    u32 short_at;       // TRACK only: first symbol whose normalize ran past the input, or 0xFFFFFFFF

    RCX_DEV u32 load_be(const u8* p) const { return rcx_bswap(*reinterpret_cast<const u32*>(p)); }

    // cpprcoder.h:877-896 + :859-870.  `s` points at the block's stream (any alignment),
    // which must be at least 8 bytes long.  Returns the declared size.
    RCX_DEV u32 begin(const u8* s, const u8* stream_end)
    {
        u32 declared = (u32)s[0] | ((u32)s[1] << 8) | ((u32)s[2] << 16) | ((u32)s[3] << 24);
        low = ((u32)s[4] << 24) | ((u32)s[5] << 16) | ((u32)s[6] << 8) | (u32)s[7];
        range = 0x00FFFFFFu;
        end = stream_end;
        const u8* body = s + 8;
        u32 skew = (u32)((uintptr_t)body & 3);
        next = body - skew;
        win = 0;
        navail8 = 0;
        taken = 8;
        short_at = 0xFFFFFFFFu;
        if (next < end) {
            win = (u64)(load_be(next) << (8 * skew)) << 32;
            navail8 = 32 - 8 * skew;
        } else {
            navail8 = 32; // past the end: zeros
        }
        next += 4;
        return declared;
    }

    RCX_DEV void refill()
    {
        u32 d = 0;
        if (next < end) d = load_be(next);
        next += 4;
        win |= (u64)d << (32 - navail8);
        navail8 += 32;
    }

    // Decodes one symbol.  `total` = 256 + symbols decoded so far.
    template <bool TRACK = false, class TreeT>
    RCX_DEV u32 step(const TreeT& tree, const DivEntry& k, bool last, u32 index = 0, u64 stream_len = 0)
    {
        if (navail8 <= 32) refill();
        // cpprcoder.h:926-940
        const u32 k8 = rcx_clz(range) & 0x18u;
        low = (u32)((((u64)low << 32) | (u32)(win >> 32)) << k8 >> 32);
        win <<= k8;
        navail8 -= k8;
        range <<= k8;
        taken += k8 >> 3;
        if (TRACK && taken > stream_len && short_at == 0xFFFFFFFFu) short_at = index; // cpprcoder.h:901-903

        const u32 total = k.total;
        const u32 t = rcx_div(range, k); // :904
        // :905 target = low / t.  A target >= total (corrupt input) falls through the
        // reference's find() with code 0 and count = total (cpprcoder.h:1220-1242).
        const bool off_table = low >= rcx_mul24(total, t);
        u32 q = (u32)((float)low * rcx_rcp((float)t));
        {
            u32 back = low - rcx_mul24(q, t);
            if ((s32)back < 0) q -= 1;
            else if (back >= t) q += 1;
        }

        u32 rem = q, cum = 0, c = 0, f;
        U4 g = tree.group(RCX_G_L3);
        u32 p;
#define RCX_DESCEND()                                                  \
    {                                                                  \
        const u32 a = g.x, b = a + g.y, d = b + g.z;                   \
        u32 base = 0;                                                  \
        p = 0;                                                         \
        if (rem >= a) { base = a; p = 1; }                             \
        if (rem >= b) { base = b; p = 2; }                             \
        if (rem >= d) { base = d; p = 3; }                             \
        rem -= base;                                                   \
        cum += base;                                                   \
        c = (c << 2) | p;                                              \
    }
        RCX_DESCEND();
        g = tree.group(RCX_G_L2 + c);
        RCX_DESCEND();
        g = tree.group(RCX_G_L1 + c);
        RCX_DESCEND();
        g = tree.group(RCX_G_L0 + c);
        RCX_DESCEND();
#undef RCX_DESCEND
        f = rcx_sel4(g, p);
        if (off_table) {
            RCX_SIM_COUNT(2, 1);
            c = 0;
            cum = total;
            f = tree.group(RCX_G_L0).x;
        }
        low -= rcx_mul24(cum, t);   // :906
        range = rcx_mul24(f, t);    // :907
        if (!last) tree.update(c);  // :912-916 the last symbol returns before update
        return c;
    }
};
