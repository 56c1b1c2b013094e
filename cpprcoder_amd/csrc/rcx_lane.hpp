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
// The table halving (cpprcoder.h:1138-1176) needs total >= 2^24, i.e. more than
// RCX_MAX_BLOCK = 2^24 - 256 symbols: the block kernels never see it (every lane's total
// is 256 + the symbol index and the divisors come from a table); a single stream longer
// than that goes through the *_long steps below: the lane keeps its own total, divides by
// it and halves (Tree::halve).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RCX_DEV __device__ __forceinline__
#define RCX_HD __host__ __device__ inline
#else
#define RCX_HD inline
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
// (g*RCX_LANES + l) * 16) so that a ds_read_b128 of ANY per-lane group index is bank-conflict
// free.  The ds_add_u32 updates then hit each bank four times (35 % of the encoder's LDS cycles
// are such conflicts); the alternative, dword planes (RCX_TREE_PLANAR=1: conflict-free adds, two
// ds_read2st64_b32 per group), measured SLOWER (5.80 vs 5.54 ms per GiB): the encoder's waves are
// bound by instructions issued, not by LDS cycles, and the planar layout costs one more per read.
// ---------------------------------------------------------------------------
#define RCX_LANES 64
#define RCX_G_L3 0
#define RCX_G_L2 1
#define RCX_G_L1 5
#define RCX_G_L0 21
#define RCX_GROUPS 85
#define RCX_STAGE 64 /* divisor-table entries staged per refill */
#define RCX_RING_DW 32 /* decoder: dwords of compressed stream buffered per block in LDS */
#define RCX_HALVE_AT (1u << 24) /* cpprcoder.h:1138: the table is halved when ++total reaches MINRANGE */

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
RCX_DEV bool rcx_any(bool p) { return p; }
RCX_DEV u32 rcx_funnel_shr(u32 hi, u32 lo, u32 sh) { return (u32)(((((u64)hi) << 32) | lo) >> (sh & 31u)); }
#define RCX_COLD inline
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
RCX_DEV void rcx_lds_add(u32* p, u32 v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
RCX_DEV float rcx_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
RCX_DEV bool rcx_any(bool p) { return __builtin_expect(__any(p) != 0, 0); } // wave-uniform: some lane has p
RCX_DEV u32 rcx_funnel_shr(u32 hi, u32 lo, u32 sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); } // ({hi,lo} >> sh) low half
#define RCX_COLD __device__ __attribute__((noinline, cold))
#endif

RCX_DEV u32 rcx_div(u32 n, const DivEntry& k) { return (u32)(((u64)n * k.mul + k.add) >> 32) >> k.shift; }

// The tree of one lane.  `col` already includes the lane offset.
#if !defined(RCX_TREE_PLANAR)
#define RCX_TREE_PLANAR 0 /* 1 = dword planes (conflict-free ds_add, two ds_read2st64_b32 per group); 0 = 16-byte groups */
#endif
struct Tree {
    u32* col;
#if RCX_TREE_PLANAR
    RCX_DEV u32* at(u32 g, u32 p) const { return col + (4 * g + p) * RCX_LANES; }
    RCX_DEV U4 group(u32 g) const
    {
        const u32* a = col + 4 * g * RCX_LANES;
        U4 v;
        v.x = a[0];
        v.y = a[RCX_LANES];
        v.z = a[2 * RCX_LANES];
        v.w = a[3 * RCX_LANES];
        return v;
    }
    RCX_DEV void store(u32 g, const U4& v) const
    {
        u32* a = col + 4 * g * RCX_LANES;
        a[0] = v.x;
        a[RCX_LANES] = v.y;
        a[2 * RCX_LANES] = v.z;
        a[3 * RCX_LANES] = v.w;
    }
#else
    // group g of lane l = the 16 bytes at dword (g * RCX_LANES + l) * 4: `col` = image + 4 * lane
    RCX_DEV u32* at(u32 g, u32 p) const { return col + 4 * g * RCX_LANES + p; }
    RCX_DEV U4 group(u32 g) const { return *reinterpret_cast<const U4*>(col + 4 * g * RCX_LANES); }
    RCX_DEV void store(u32 g, const U4& v) const { *reinterpret_cast<U4*>(col + 4 * g * RCX_LANES) = v; }
#endif
    RCX_DEV void bump(u32 g, u32 p) const { rcx_lds_inc(at(g, p)); }
    // same effect as bump() when the caller already holds the element's current value
    RCX_DEV void put(u32 g, u32 p, u32 value) const { *at(g, p) = value; }
    // cpprcoder.h:1094-1132: every count 1.
    RCX_DEV void reset() const
    {
        U4 v;
        v.x = v.y = v.z = v.w = 64;
        store(RCX_G_L3, v);
        v.x = v.y = v.z = v.w = 16;
        for (u32 g = RCX_G_L2; g < RCX_G_L1; ++g) store(g, v);
        v.x = v.y = v.z = v.w = 4;
        for (u32 g = RCX_G_L1; g < RCX_G_L0; ++g) store(g, v);
        v.x = v.y = v.z = v.w = 1;
        for (u32 g = RCX_G_L0; g < RCX_GROUPS; ++g) store(g, v);
    }
    // cpprcoder.h:1134-1177 without the halving: +1 on the path to the leaf.
    RCX_DEV void update(u32 c) const
    {
        bump(RCX_G_L3, c >> 6);
        bump(RCX_G_L2 + (c >> 6), (c >> 4) & 3);
        bump(RCX_G_L1 + (c >> 4), (c >> 2) & 3);
        bump(RCX_G_L0 + (c >> 2), c & 3);
    }
    // cpprcoder.h:1138-1176: every count becomes (f >> 1) | 1, the sums above them are rebuilt.
    // Returns the new total.  (Once per 2^23 symbols of a very long single stream.)
    RCX_DEV u32 halve() const
    {
        u32 total = 0;
        for (u32 q3 = 0; q3 < 4; ++q3) {          // level-3 node q3 = symbols 64*q3 ..
            U4 n2;
            u32 sum2[4];
            for (u32 q2 = 0; q2 < 4; ++q2) {      // level-2 node = 16 symbols
                U4 n1;
                u32 sum1[4];
                for (u32 q1 = 0; q1 < 4; ++q1) {  // level-1 node = 4 symbols = one leaf group
                    const u32 g = RCX_G_L0 + 16 * q3 + 4 * q2 + q1;
                    U4 v = group(g);
                    v.x = (v.x >> 1) | 1u;
                    v.y = (v.y >> 1) | 1u;
                    v.z = (v.z >> 1) | 1u;
                    v.w = (v.w >> 1) | 1u;
                    store(g, v);
                    sum1[q1] = v.x + v.y + v.z + v.w;
                }
                n1.x = sum1[0];
                n1.y = sum1[1];
                n1.z = sum1[2];
                n1.w = sum1[3];
                store(RCX_G_L1 + 4 * q3 + q2, n1);
                sum2[q2] = sum1[0] + sum1[1] + sum1[2] + sum1[3];
            }
            n2.x = sum2[0];
            n2.y = sum2[1];
            n2.z = sum2[2];
            n2.w = sum2[3];
            store(RCX_G_L2 + q3, n2);
            const u32 s3 = sum2[0] + sum2[1] + sum2[2] + sum2[3];
            put(RCX_G_L3, q3, s3);
            total += s3;
        }
        return total;
    }
};

// Sum of the first p entries of a group and the p-th entry, p in 0..3.  Written with 0/1 factors instead of
// compares: (p > 0) = min(p, 1), (p > 1) = p >> 1, (p > 2) = p & (p >> 1), and the counts are < 2^24, so each
// term is one v_mad_u32_u24 -- no compare whose mask the next instruction has to wait two slots for.
RCX_DEV u32 rcx_min1(u32 p)
{
#if defined(RCX_HOST_SIM)
    return p < 1u ? p : 1u;
#else
    u32 m; // opaque to the compiler, which would turn "x * min(p, 1)" back into compare + select
    asm("v_min_u32 %0, 1, %1" : "=v"(m) : "v"(p));
    return m;
#endif
}
RCX_DEV u32 rcx_pre4(const U4& g, u32 p)
{
    const u32 m1 = rcx_min1(p), m2 = p >> 1, m3 = p & m2;
    return rcx_mul24(g.x, m1) + rcx_mul24(g.y, m2) + rcx_mul24(g.z, m3);
}
RCX_DEV u32 rcx_sel4(const U4& g, u32 p)
{
    const u32 m1 = rcx_min1(p), m2 = p >> 1, m3 = p & m2;
    // (sum of the first p + 1) - (sum of the first p)
    return g.x + rcx_mul24(g.y, m1) + rcx_mul24(g.z, m2) + rcx_mul24(g.w, m3) - (rcx_mul24(g.x, m1) + rcx_mul24(g.y, m2) + rcx_mul24(g.z, m3));
}

// Add `extra` into the `count` payload bytes already stored, from the newest backwards
// (cpprcoder.h:767-781 when the carry leaves the bytes held in registers).  Rare: ~9e-5 per symbol.
RCX_DEV void rcx_carry_walk(u8* out, u32 count, u32 extra)
{
    if (count && extra) RCX_SIM_COUNT(0, 1);
    while (count > 0 && extra) {
        --count;
        RCX_SIM_COUNT(1, 1);
        u32 v = (u32)out[count] + extra;
        out[count] = (u8)v;
        extra = v >> 8;
    }
}
// out of line, by value: keeps the coder's registers out of memory on the hot path
RCX_COLD void rcx_carry_slow(u8* out, u32 count, u32 extra) { rcx_carry_walk(out, count, extra); }

// ---------------------------------------------------------------------------
// Encoder lane.
//
// Output bytes are produced eagerly: every renormalisation byte is appended at
// once and a carry is added into the bytes already produced, which yields the
// same stream as the reference's held-byte + pending-0xFF counter
// (cpprcoder.h:767-800).  The newest 1..7 bytes live in `acc` (big-endian
// number, newest byte lowest) so that a carry is one 64-bit add.  A carry that
// runs through every held byte leaves a bit just above them; it is noticed when
// the bytes are flushed and only then walks back through memory (about 9e-5 per
// symbol on random data).
// ---------------------------------------------------------------------------
struct EncLane {
    u32 low, range;
    u64 acc;            // pending output bytes in the low nacc8 bits (+ possibly a carry bit above them)
    u32 nacc8;          // 8 * number of bytes in acc; 8..32 between steps
    u32 pos;            // payload bytes already stored
    u32 cap;            // payload capacity of the slot (bytes, multiple of 4, >= 12)
    u32 overflow;       // slot too small: output is dropped from here on
    u8* base;           // wave-uniform base of the slots this wave writes
    u32 off;            // this lane's payload offset from base (slot offset + 4)
    bool leader;        // several lanes may run one block's coder in lock-step; only the leader stores
    // TRACK only: where the reference's delayed writer (held byte + pending run,
    // cpprcoder.h:767-800) stands, to find the symbol at which a bounded sink fills.
    u32 trk_written;    // payload bytes the reference has passed to writeByte so far
    u32 trk_pending;    // its carry_ counter
    u32 trk_cap;        // writeByte calls that succeed (sink capacity - 4 header bytes)
    u32 trk_fail_at;    // first symbol whose normalize hits the full sink, or 0xFFFFFFFF

    RCX_DEV u8* payload() const { return base + off; }

    RCX_DEV void reset_state()
    {
        low = 0;
        range = 0xFFFFFF00u;
        acc = 0; // the reference's first emitted byte is its initial buffer_ = 0: already "held" here
        nacc8 = 8;
        pos = 0;
        overflow = 0;
        leader = true;
        trk_written = 0;
        trk_pending = 0;
        trk_cap = 0xFFFFFFFFu;
        trk_fail_at = 0xFFFFFFFFu;
    }

    // cpprcoder.h:678-695: u32 LE size, then the coder state.  The slot is wave_base + slot_off.
    RCX_DEV void begin(u8* wave_base, u32 slot_off, u32 slot_bytes, u32 declared)
    {
        u8* slot = wave_base + slot_off;
        slot[0] = (u8)declared;
        slot[1] = (u8)(declared >> 8);
        slot[2] = (u8)(declared >> 16);
        slot[3] = (u8)(declared >> 24);
        base = wave_base;
        off = slot_off + 4;
        cap = (slot_bytes - 4) & ~3u;
        reset_state();
    }

    // a lane without a block: keeps the arithmetic well-defined, never stores
    RCX_DEV void idle(u8* wave_base)
    {
        base = wave_base;
        off = 0;
        cap = 0;
        reset_state();
        leader = false;
    }

    // `extra` carries ran through every byte held in acc: continue in memory (cpprcoder.h:767-781)
    RCX_DEV void carry_into_memory(u32 extra) { rcx_carry_walk(base + off, leader ? (pos < cap ? pos : 0u) : 0u, extra); }

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

    // One symbol with the model's answer already in hand: cum = sum of the counts below the
    // symbol, f = its count (cpprcoder.h:703-711).  Two halves so that they can also run in two
    // different waves: arith() is the interval arithmetic (state: low, range) and returns a
    // record {top 24 bits of the moved low | 8 x the bytes leaving, in bits 3..4 | carry}; emit() is the byte
    // writer (state: acc, nacc8, pos).
    // WIDE: full 32-bit multiplies (the static coder's total can be tiny, so t can exceed 24 bits).
    template <bool WIDE = false>
    RCX_DEV u32 arith(u32 cum, u32 f, const DivEntry& k)
    {
        return arith_t<WIDE>(cum, f, rcx_div(range, k)); // cpprcoder.h:703
    }
    // the same with t = range / total in hand
    template <bool WIDE = false>
    RCX_DEV u32 arith_t(u32 cum, u32 f, u32 t)
    {
        const u32 moved = low + (WIDE ? cum * t : rcx_mul24(cum, t)); // :706  (cum*t <= range < 2^32)
        const u32 carry = moved < low ? 1u : 0u;
        range = WIDE ? f * t : rcx_mul24(f, t);      // :707
        const u32 k8 = rcx_clz(range) & 0x18u;       // :783-800: k8/8 bytes leave through the top of low
        low = moved << k8;
        range <<= k8;
        return (moved & 0xFFFFFF00u) | k8 | carry; // k8 is 0, 8, 16 or 24: bits 3 and 4
    }
    // the same with the divisor's fields passed separately (the addend as the 64-bit pair the multiply-add takes)
    RCX_DEV u32 arith_q(u32 cum, u32 f, u32 mul, u32 shift, u64 add)
    {
        const u32 t = (u32)(((u64)range * mul + add) >> 32) >> (shift & 31u); // cpprcoder.h:703
        const u32 moved = low + rcx_mul24(cum, t);                            // :706
        const u32 carry = moved < low ? 1u : 0u;
        range = rcx_mul24(f, t);                                              // :707
        const u32 k8 = rcx_clz(range) & 0x18u;
        low = moved << k8;
        range <<= k8;
        return (moved & 0xFFFFFF00u) | k8 | carry; // k8 is 0, 8, 16 or 24: bits 3 and 4
    }
    RCX_DEV void emit(u32 rec)
    {
        const u32 k8 = rec & 0x18u;
        acc += rec & 1u;                             // :767-781 carry, resolved lazily (see flush)
        acc = (acc << k8) | (((u64)rec << k8) >> 32);
        nacc8 += k8;
        flush();
    }
    template <bool TRACK = false, bool WIDE = false>
    RCX_DEV void code(u32 cum, u32 f, const DivEntry& k, u32 index = 0)
    {
        const u32 rec = arith<WIDE>(cum, f, k);
        if (TRACK) track(index, rec & 1u, rec & 0xFFFFFF00u, rec & 0x18u);
        emit(rec);
    }

    // With 5..7 bytes held: store the 4 oldest, keep 1..3.  Some lane of a wave is in that state on
    // almost every symbol, so this runs for all lanes without a branch: everything is computed and
    // selected, the store is the only predicated piece, and the carry-ran-off-the-register case is
    // one wave-uniform, normally not taken test.  `pos` may run past `cap` (the store address is
    // clamped, so a too-small slot keeps overwriting its last word); finish() reports that.
    RCX_DEV void flush()
    {
        const bool due = nacc8 >= 40;
        const u32 keep8 = (nacc8 - 32) & 31u;                    // 8, 16 or 24 when due
        const u32 acc_lo = (u32)acc, acc_hi = (u32)(acc >> 32);
        const u32 word = rcx_funnel_shr(acc_hi, acc_lo, keep8);  // the 4 oldest bytes
        const u32 extra = due ? acc_hi >> keep8 : 0u;            // a carry that ran off the held bytes
        if (rcx_any(extra != 0)) rcx_carry_slow(base + off, leader ? (pos < cap ? pos : 0u) : 0u, extra);
        const u32 where = pos < cap - 4 ? pos : cap - 4;
        if (due && leader) *reinterpret_cast<u32*>(base + (off + where)) = rcx_bswap(word);
        const u32 kept = acc_lo & ((1u << keep8) - 1u);
        acc = due ? (u64)kept : acc;
        nacc8 -= due ? 32u : 0u;
        pos += due ? 4u : 0u;
    }
    // One lane per block: model query, code, model update.
    template <bool TRACK = false, class TreeT>
    RCX_DEV void step(const TreeT& tree, u32 c, const DivEntry& k, u32 index = 0)
    {
        const U4 g3 = tree.group(RCX_G_L3);
        const U4 g2 = tree.group(RCX_G_L2 + (c >> 6));
        const U4 g1 = tree.group(RCX_G_L1 + (c >> 4));
        const U4 g0 = tree.group(RCX_G_L0 + (c >> 2));
        const u32 cum = rcx_pre4(g3, c >> 6) + rcx_pre4(g2, (c >> 4) & 3) + rcx_pre4(g1, (c >> 2) & 3) + rcx_pre4(g0, c & 3);
        const u32 f = rcx_sel4(g0, c & 3);
        code<TRACK>(cum, f, k, index);
        tree.update(c); // :712
    }

    // The same for a stream of any length: the lane's own `total` (cpprcoder.h:1096) is the divisor, a true
    // division, and the table is halved when the total reaches 2^24 (cpprcoder.h:1138).
    template <bool TRACK = false, class TreeT>
    RCX_DEV void step_long(const TreeT& tree, u32 c, u32& total, u32 index = 0)
    {
        const U4 g3 = tree.group(RCX_G_L3);
        const U4 g2 = tree.group(RCX_G_L2 + (c >> 6));
        const U4 g1 = tree.group(RCX_G_L1 + (c >> 4));
        const U4 g0 = tree.group(RCX_G_L0 + (c >> 2));
        const u32 cum = rcx_pre4(g3, c >> 6) + rcx_pre4(g2, (c >> 4) & 3) + rcx_pre4(g1, (c >> 2) & 3) + rcx_pre4(g0, c & 3);
        const u32 f = rcx_sel4(g0, c & 3);
        const u32 rec = arith_t(cum, f, range / total); // :703
        if (TRACK) track(index, rec & 1u, rec & 0xFFFFFF00u, rec & 0x18u);
        emit(rec);
        tree.update(c); // :712
        total += 1;
        if (total >= RCX_HALVE_AT) total = tree.halve();
    }

    // cpprcoder.h:744-762: the held bytes, then low big-endian.  Returns the stream size.
    RCX_DEV u32 finish()
    {
        if (pos > cap) overflow = 1; // flush() clamps its stores instead of testing
        const u32 extra = (u32)(acc >> nacc8);
        if (extra) carry_into_memory(extra);
        const u32 n = nacc8 >> 3;
        u8* out = payload();
        for (u32 i = 0; i < n; ++i) {
            u8 b = (u8)(acc >> (8 * (n - 1 - i)));
            if (pos >= cap) overflow = 1;
            else if (leader) out[pos] = b;
            ++pos;
        }
        for (u32 i = 0; i < 4; ++i) {
            u8 b = (u8)(low >> (24 - 8 * i));
            if (pos >= cap) overflow = 1;
            else if (leader) out[pos] = b;
            ++pos;
        }
        return pos + 4;
    }
};

// ---------------------------------------------------------------------------
// Decoder lane.
//
// find() (cpprcoder.h:1220-1242) runs in the scaled domain: with t = range/total the
// reference's "cum(c) <= low/t < cum(c+1)" (cpprcoder.h:905) is "cum(c)*t <= low <
// cum(c+1)*t"; every product is <= total*t <= range < 2^32, so the second divide is not
// needed and low - cum(c)*t (cpprcoder.h:906) is what is left when the descent ends.
// ---------------------------------------------------------------------------
struct DecLane {
    u32 low, range;
    u64 win;            // upcoming stream bytes, left-aligned (next byte on top)
    u32 navail8;        // 8 * bytes in win
    u32 ahead;          // the dword after the window, taken from the ring one refill early; RAW (memory order)
    u32 peek;           // ring[rd], read one symbol early so that its LDS latency is never waited for
    // The compressed stream reaches the window through a per-lane ring of RCX_RING_DW dwords in LDS that
    // is topped up every 16 symbols with 16-byte global loads issued one top-up ahead.  (Loading the
    // window straight from global memory puts a vmcnt wait -- which also waits for the older output
    // stores -- into every symbol: measured 430 cycles per symbol.)
    u32* ring;          // this block's ring column: dword d lives at ring[(d % RCX_RING_DW) * stride]
    u32 stride;         // columns in the ring image (64 with one lane per block; lanes of one block share a column)
    u32 rd, wr;         // dwords taken from / written to the ring so far, counted from `origin`
    const u8* origin;   // 16-byte aligned address of ring dword 0
    U4 pend0, pend1, pend2, pend3; // pieces requested at the previous top-up, not yet in the ring
    u32 npend;
    const u8* body;     // first stream byte after the 8 header bytes
    const u8* end;      // one past the block's stream
    u32 short_at;       // TRACK only: first symbol whose normalize ran past the input, or 0xFFFFFFFF
#if defined(RCX_STAMP_DEC) /* diagnostic build only */
    unsigned long long stamp_sum[8];
#define RCX_DSTAMP(i)                                            \
    {                                                            \
        __builtin_amdgcn_sched_barrier(0);                       \
        unsigned long long now_;                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                       \
        stamp_sum[i] += now_ - stamp_last;                       \
        stamp_last = now_;                                       \
    }
    unsigned long long stamp_last;
#else
#define RCX_DSTAMP(i)
#endif

    // 16 aligned bytes; a piece that starts at or past the end of the stream is zeros (one that
    // merely straddles the end stays inside the page that holds the stream's last byte)
    RCX_DEV U4 load16(const u8* p) const
    {
        U4 z;
        z.x = z.y = z.z = z.w = 0;
        return p < end ? *reinterpret_cast<const U4*>(p) : z;
    }
    RCX_DEV void ring_put(const U4& piece)
    {
        u32* at = ring + (wr % RCX_RING_DW) * stride; // wr is a multiple of 4: the piece never wraps
        at[0] = piece.x;                               // (lanes sharing a column store identical values)
        at[stride] = piece.y;
        at[2 * stride] = piece.z;
        at[3 * stride] = piece.w;
        wr += 4;
    }
    RCX_DEV u32 ring_get()
    {
        const u32 v = ring[(rd % RCX_RING_DW) * stride];
        rd += 1;
        return v;
    }

    // cpprcoder.h:877-896 + :859-870.  `s` points at the block's stream (any alignment),
    // which must be at least 8 bytes long.  Returns the declared size.
    RCX_DEV u32 begin(const u8* s, const u8* stream_end, u32* ring_column, u32 ring_stride = RCX_LANES)
    {
        stride = ring_stride;
        u32 declared = (u32)s[0] | ((u32)s[1] << 8) | ((u32)s[2] << 16) | ((u32)s[3] << 24);
        low = ((u32)s[4] << 24) | ((u32)s[5] << 16) | ((u32)s[6] << 8) | (u32)s[7];
        range = 0x00FFFFFFu;
        end = stream_end;
        body = s + 8;
        ring = ring_column;
        origin = body - ((uintptr_t)body & 15);
        wr = 0;
        for (u32 r = 0; r < 6; ++r) ring_put(load16(origin + 16 * r)); // prologue: 24 dwords, synchronously
        npend = 0;
        rd = (u32)(body - origin) >> 2;
        const u32 skew = (u32)((uintptr_t)body & 3);
        win = (u64)(rcx_bswap(ring_get()) << (8 * skew)) << 32;
        navail8 = 32 - 8 * skew;
        ahead = ring_get();
        peek = ring[(rd % RCX_RING_DW) * stride];
        short_at = 0xFFFFFFFFu;
        return declared;
    }

    // a lane without a block
    RCX_DEV void idle(const u8* anywhere, u32* ring_column, u32 ring_stride = RCX_LANES)
    {
        stride = ring_stride;
        low = 0;
        range = 0x01000000u;
        win = 0;
        navail8 = 64;
        ahead = 0;
        peek = 0;
        ring = ring_column;
        rd = wr = 0;
        npend = 0;
        origin = body = end = anywhere;
        short_at = 0xFFFFFFFFu;
    }

    // Every 16 symbols (a lane takes at most 13 dwords in that time): move the pieces requested
    // last time into the ring, then request as many new ones as fit -- at most 4 = 16 dwords.
    RCX_DEV void topup()
    {
        if (npend > 0) ring_put(pend0);
        if (npend > 1) ring_put(pend1);
        if (npend > 2) ring_put(pend2);
        if (npend > 3) ring_put(pend3);
        npend = 0;
        u32 planned = wr;
        // slots [rd, wr) are unread; slot rd-1 is already in `ahead`
        if (planned + 4 - rd <= RCX_RING_DW) { pend0 = load16(origin + 4 * (size_t)planned); planned += 4; npend = 1; }
        if (planned + 4 - rd <= RCX_RING_DW) { pend1 = load16(origin + 4 * (size_t)planned); planned += 4; npend = 2; }
        if (planned + 4 - rd <= RCX_RING_DW) { pend2 = load16(origin + 4 * (size_t)planned); planned += 4; npend = 3; }
        if (planned + 4 - rd <= RCX_RING_DW) { pend3 = load16(origin + 4 * (size_t)planned); planned += 4; npend = 4; }
    }

    // stream bytes consumed so far, header included (for the truncation check, cpprcoder.h:901-903)
    RCX_DEV u64 taken() const { return 8 + (u64)(4 * (size_t)(rd - 1)) - (u64)(body - origin) - (navail8 >> 3); }

    // cpprcoder.h:926-940: shift in the bytes that bring range back above 2^24.  The window is
    // topped up from `ahead`, whose replacement is requested from the ring right away and not
    // needed before the next top-up (a few symbols later), so no latency is exposed here.
    RCX_DEV void pull()
    {
        // branch-free: some lane of a wave tops up on almost every symbol, and without a branch the
        // compiler can schedule across symbols (the ring is read unconditionally; the value is only kept
        // by the lanes that needed it)
        const bool need = navail8 <= 32;
        const u64 add = need ? (u64)rcx_bswap(ahead) << ((32 - navail8) & 63u) : 0;
        win |= add;
        navail8 += need ? 32u : 0u;
        ahead = need ? peek : ahead;
        rd += need ? 1u : 0u;
        peek = ring[(rd % RCX_RING_DW) * stride]; // for the next symbol
        const u32 k8 = rcx_clz(range) & 0x18u;
        low = (u32)((((u64)low << 32) | (u32)(win >> 32)) << k8 >> 32);
        win <<= k8;
        navail8 -= k8;
        range <<= k8;
    }

    // Decodes one symbol (one lane per block).  LONG (a stream of any length): k.total is the lane's own total, the
    // division is a true one, and the caller halves the table when the total reaches 2^24 (see EncLane::step_long).
    template <bool TRACK = false, bool LONG = false, class TreeT>
    RCX_DEV u32 step(const TreeT& tree, const DivEntry& k, u32 index = 0, u64 stream_len = 0)
    {
        // the top two tree levels sit at fixed addresses: ask for them before anything else
        const U4 g3 = tree.group(RCX_G_L3);
        const U4 q0 = tree.group(RCX_G_L2 + 0), q1 = tree.group(RCX_G_L2 + 1);
        const U4 q2 = tree.group(RCX_G_L2 + 2), q3 = tree.group(RCX_G_L2 + 3);
        RCX_DSTAMP(0); // since the end of the previous symbol (loop glue, divisor fetch, the five reads issued)
        pull();
        RCX_DSTAMP(1);
        if (TRACK && taken() > stream_len && short_at == 0xFFFFFFFFu) short_at = index; // cpprcoder.h:901-903

        const u32 t = LONG ? range / k.total : rcx_div(range, k); // :904
        const u32 top = rcx_mul24(k.total, t);
        u32 rem = low, c = 0, p, hit;
        U4 g = g3;
        // one level: which of the 4 children holds rem, what is left of rem below it, and the
        // child's own count (`hit`, needed to store count+1 back without an LDS atomic)
#define RCX_DESCEND()                                                          \
    {                                                                          \
        const u32 s2 = g.x + g.y, s3 = s2 + g.z;                               \
        const u32 a = rcx_mul24(g.x, t), b = rcx_mul24(s2, t), d = rcx_mul24(s3, t); \
        u32 base = 0;                                                          \
        p = 0;                                                                 \
        hit = g.x;                                                             \
        if (rem >= a) { base = a; p = 1; hit = g.y; }                          \
        if (rem >= b) { base = b; p = 2; hit = g.z; }                          \
        if (rem >= d) { base = d; p = 3; hit = g.w; }                          \
        rem -= base;                                                           \
        c = (c << 2) | p;                                                      \
    }
        RCX_DESCEND();
        const u32 c3 = c, hit3 = hit;
        {
            const U4 lo = (p & 1) ? q1 : q0, hi = (p & 1) ? q3 : q2;
            g = (p & 2) ? hi : lo;
        }
        RCX_DESCEND();
        const u32 c2 = c, hit2 = hit;
        RCX_DSTAMP(2); // divide + the two register-resident levels
        g = tree.group(RCX_G_L1 + c);
        RCX_DESCEND();
        const u32 c1 = c, hit1 = hit;
        RCX_DSTAMP(3); // level 1 (dependent LDS read)
        g = tree.group(RCX_G_L0 + c);
        RCX_DESCEND();
        RCX_DSTAMP(4); // level 0 (dependent LDS read)
#undef RCX_DESCEND
        u32 f = hit;
        // A target at or past total (corrupt input) falls through the reference's find() with
        // code 0 and count = total (cpprcoder.h:1220-1242).
        if (low >= top) {
            RCX_SIM_COUNT(2, 1);
            c = 0;
            rem = low - top;
            f = tree.group(RCX_G_L0).x;
        }
        const bool off_table = low >= top;
        low = rem;                // :906
        range = rcx_mul24(f, t);  // :907
        // :916 +1 on the path to the leaf (after the last symbol the table is never looked at again)
        if (off_table) {
            tree.update(0);
        } else {
            tree.put(RCX_G_L3, c3, hit3 + 1);
            tree.put(RCX_G_L2 + c3, c2 & 3, hit2 + 1);
            tree.put(RCX_G_L1 + c2, c1 & 3, hit1 + 1);
            tree.put(RCX_G_L0 + c1, c & 3, f + 1);
        }
        RCX_DSTAMP(5); // tail: f, low/range, the four ds_add
        return c;
    }
};

// ---------------------------------------------------------------------------
// One decoder step on plain state, for the resumable single-stream decoder (rcx_dstream_*): no input ring, the
// caller has already shifted the bytes in (cpprcoder.h:926-940).  t = range / total as a true division, find() in
// the scaled domain as DecLane::step, update with the halving (cpprcoder.h:900-917, :1134-1177).
// ---------------------------------------------------------------------------
template <class TreeT>
RCX_DEV u32 rcx_decode_plain(const TreeT& tree, u32& low, u32& range, u32& total)
{
    const u32 t = range / total;           // :904
    const u32 top = rcx_mul24(total, t);
    u32 rem = low, c = 0;
    u32 path_g[4], path_p[4], path_hit[4];
    u32 g = RCX_G_L3;
    for (u32 level = 0; level < 4; ++level) {
        const U4 v = tree.group(g);
        const u32 s2 = v.x + v.y, s3 = s2 + v.z;
        const u32 a = rcx_mul24(v.x, t), b = rcx_mul24(s2, t), d = rcx_mul24(s3, t);
        u32 base = 0, p = 0, hit = v.x;
        if (rem >= a) { base = a; p = 1; hit = v.y; }
        if (rem >= b) { base = b; p = 2; hit = v.z; }
        if (rem >= d) { base = d; p = 3; hit = v.w; }
        rem -= base;
        path_g[level] = g;
        path_p[level] = p;
        path_hit[level] = hit;
        c = (c << 2) | p;
        g = (level == 0 ? RCX_G_L2 : level == 1 ? RCX_G_L1 : RCX_G_L0) + c;
    }
    u32 f = path_hit[3];
    if (low >= top) { // a target at or past the total: find() falls through with code 0, count = total (:1220-1242)
        c = 0;
        rem = low - top;
        f = tree.group(RCX_G_L0).x;
        tree.update(0);
    } else {
        for (u32 level = 0; level < 4; ++level) tree.put(path_g[level], path_p[level], path_hit[level] + 1);
    }
    low = rem;               // :906
    range = rcx_mul24(f, t); // :907
    total += 1;              // :1138
    if (total >= RCX_HALVE_AT) total = tree.halve();
    return c;
}

