// lane_sim.cpp -- host-side lane simulator (TEST TOOLING, not part of librcx.so).
//
// Compiles cpprcoder_amd/csrc/rcx_lane.hpp -- the very code the gfx950 kernels run per
// lane -- with g++ and drives it block by block, so that the CPU test suite can check
// the device arithmetic (divisor table, eager carry, register window, tree model)
// against the oracle without a GPU.  Lane `lane` of a 64-lane LDS image is used so that
// the interleaved layout is exercised too.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../cpprcoder_amd/csrc/rcx_divtab.hpp"
#include "../../cpprcoder_amd/csrc/rcx_bwt_tie.hpp"

uint64_t rcx_sim_counters[4] = {0, 0, 0, 0};

// The replay as round 2 ran it: one lane, the reference's moves one at a time (blksort.h:168-363 on row classes).  Test
// tooling: the wave-wide replay of csrc/rcx_bwt_tie.hpp must leave EVERY row where this leaves it.
namespace scalar_tie
{
struct ScalarTieSort {
    uint16_t* rows;   // 32768 rows, initial order 0, 1, 2, ...
    const u8* word;   // the period
    u32 pmask;        // p - 1
    u32 depth;        // 32768

    u32 byte_at(u32 row, u32 d) const { return word[(row + d) & pmask]; }

    // blksort.h:183-211 over the full depth
    bool less(u32 a, u32 b) const
    {
        const u32 ca = a & pmask, cb = b & pmask;
        if (ca == cb) return false;
        for (u32 d = 0; d <= pmask; ++d) {
            const u32 x = word[(ca + d) & pmask], y = word[(cb + d) & pmask];
            if (x != y) return x < y;
        }
        return false; // not reached: two classes of a primitive period differ within it
    }

    bool one_class(const uint16_t* v, u32 size) const
    {
        const u32 c = v[0] & pmask;
        for (u32 i = 1; i < size; ++i)
            if ((v[i] & pmask) != c) return false;
        return true;
    }

    // blksort.h:168-181
    u32 pivot_row(const uint16_t* v, u32 size) const
    {
        const u32 q = size >> 2;
        const u32 a = byte_at(v[q], 0), b = byte_at(v[2 * q], 0), c = byte_at(v[3 * q], 0);
        if (a < b) return b < c ? v[2 * q] : (a < c ? v[3 * q] : v[q]);
        return a < c ? v[q] : (b < c ? v[3 * q] : v[2 * q]);
    }

    // blksort.h:225-235
    void insertion(uint16_t* v, u32 size) const
    {
        for (u32 i = 1; i < size; ++i) {
            const uint16_t x = v[i];
            s32 j = (s32)i - 1;
            while (j >= 0 && less(x, v[j])) {
                v[j + 1] = v[j];
                --j;
            }
            v[j + 1] = x;
        }
    }

    // blksort.h:237-279 (h is 1-based)
    void sift(uint16_t* h, s32 i, s32 n, uint16_t x) const
    {
        s32 j;
        while ((j = i << 1) <= n) {
            if (j < n && less(h[j], h[j + 1])) ++j;
            if (!less(x, h[j])) break;
            h[i] = h[j];
            i = j;
        }
        h[i] = x;
    }
    void heap(uint16_t* v, u32 size) const
    {
        uint16_t* h = v - 1;
        s32 n = (s32)size;
        for (s32 k = n >> 1; k >= 1; --k) sift(h, k, n, h[k]);
        while (n > 1) {
            const uint16_t x = h[n];
            h[n] = h[1];
            --n;
            sift(h, 1, n, x);
        }
    }

    static void swap_rows(uint16_t* v, s32 a, s32 b)
    {
        const uint16_t t = v[a];
        v[a] = v[b];
        v[b] = t;
    }

    struct Part {
        u32 off, size, d;
        s32 level;
    };

    // blksort.h:281-363 from sort(size, rows, depth): level 11.  `stack` holds RCX_TIE_STACK parts.
    // Returns false if the stack would overflow (cannot happen: see the bound above).
    bool run(Part* stack) const
    {
        u32 top = 0;
        Part cur = {0u, depth, 0u, 11};
        bool have = true;
        while (have || top > 0) {
            if (!have) cur = stack[--top];
            have = false;
            uint16_t* v = rows + cur.off;
            if (cur.level <= 0) { // blksort.h:284-287
                heap(v, cur.size);
                continue;
            }
            if (cur.d >= depth) continue; // blksort.h:288
            if (cur.size < 37) {          // blksort.h:289-292
                insertion(v, cur.size);
                continue;
            }
            if (one_class(v, cur.size)) continue; // nothing moves any more
            const u32 p = byte_at(pivot_row(v, cur.size), cur.d);
            const s32 last = (s32)cur.size - 1;
            s32 lo = 0, hi = last, eq_lo = 0, eq_hi = last;
            for (;;) { // blksort.h:301-326
                while (lo <= hi) {
                    const u32 c = byte_at(v[lo], cur.d);
                    if (p < c) break;
                    if (p == c) swap_rows(v, lo, eq_lo++);
                    ++lo;
                }
                while (lo <= hi) {
                    const u32 c = byte_at(v[hi], cur.d);
                    if (c < p) break;
                    if (p == c) swap_rows(v, hi, eq_hi--);
                    --hi;
                }
                if (hi < lo) break;
                swap_rows(v, lo, hi);
                ++lo;
                --hi;
            }
            const s32 below = lo - eq_lo, above = eq_hi - hi; // blksort.h:327-336
            const s32 r0 = eq_lo < below ? eq_lo : below;
            for (s32 i = 0; i < r0; ++i) swap_rows(v, i, hi - i);
            const s32 right_eq = last - eq_hi;
            const s32 r1 = right_eq < above ? right_eq : above;
            for (s32 i = 0; i < r1; ++i) swap_rows(v, lo + i, last - i);
            const s32 m0 = below, m1 = last - above + 1;
            // blksort.h:337-348: [0, m0) and [m1, size) one level down at the same byte, [m0, m1) at the next byte
            Part kids[3];
            u32 nk = 0;
            if (m0 - 1 > 0) kids[nk++] = Part{cur.off, (u32)m0, cur.d, cur.level - 1};
            if (m1 < last) kids[nk++] = Part{cur.off + (u32)m1, (u32)((s32)cur.size - m1), cur.d, cur.level - 1};
            if (m1 > m0) kids[nk++] = Part{cur.off + (u32)m0, (u32)(m1 - m0), cur.d + 1, cur.level};
            // the smallest next, the others pushed largest first
            for (u32 a = 0; a + 1 < nk; ++a)
                for (u32 b = a + 1; b < nk; ++b)
                    if (kids[b].size > kids[a].size) {
                        const Part t = kids[a];
                        kids[a] = kids[b];
                        kids[b] = t;
                    }
            if (nk == 0) continue;
            if (top + nk - 1 > RCX_TIE_STACK) return false;
            for (u32 a = 0; a + 1 < nk; ++a) stack[top++] = kids[a];
            cur = kids[nk - 1];
            have = true;
        }
        return true;
    }

    // the row where rotation 0 ended up
    u32 row_of_zero() const
    {
        for (u32 i = 0; i < depth; ++i)
            if (rows[i] == 0) return i;
        return 0;
    }
};

} // namespace scalar_tie

extern "C" {

void sim_counters(uint64_t* out, int reset)
{
    for (int i = 0; i < 4; ++i) {
        out[i] = rcx_sim_counters[i];
        if (reset) rcx_sim_counters[i] = 0;
    }
}

// exhaustive-ish check of the divisor entries: returns the first failing divisor or 0
uint32_t sim_check_divtab(uint32_t d_first, uint32_t d_last)
{
    for (uint32_t d = d_first; d <= d_last; ++d) {
        DivEntry e = rcx_make_div_entry(d);
        const uint64_t probes[] = {0, 1, d - 1, d, d + 1, 2ull * d - 1, 2ull * d, 0x00FFFFFFull, 0x01000000ull, 0x7FFFFFFFull,
                                   0x80000000ull, 0xFFFFFF00ull, 0xFFFFFFFEull, 0xFFFFFFFFull};
        for (uint64_t n : probes)
            if (rcx_div((u32)n, e) != (u32)n / d) return d;
        // multiples of d around the top of the range, where the magic is most stressed
        uint64_t q = 0xFFFFFFFFull / d;
        for (uint64_t k = 0; k < 64 && k <= q; ++k) {
            uint64_t m = (q - k) * d;
            for (int delta = -1; delta <= 1; ++delta) {
                uint64_t n = m + delta;
                if (n > 0xFFFFFFFFull) continue;
                if (rcx_div((u32)n, e) != (u32)(n / d)) return d;
            }
        }
        uint64_t x = 0x9E3779B97F4A7C15ull * d;
        for (int k = 0; k < 64; ++k) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            u32 n = (u32)(x >> 16);
            if (rcx_div(n, e) != n / d) return d;
        }
    }
    return 0;
}

int sim_encode_blocks(const uint8_t* src, uint64_t n, uint32_t block, uint8_t* slots, uint64_t slot, uint32_t* sizes, uint32_t lane)
{
    std::vector<U4> lds((RCX_GROUPS + 1) * RCX_LANES);
    std::vector<DivEntry> tab(block + 2 * RCX_STAGE);
    for (size_t i = 0; i < tab.size(); ++i) tab[i] = rcx_make_div_entry((u32)(256 + i));
    uint64_t nblocks = (n + block - 1) / block;
    int overflow = 0;
    for (uint64_t b = 0; b < nblocks; ++b) {
        uint64_t at = b * block;
        uint32_t len = (uint32_t)((n - at) < block ? (n - at) : block);
        Tree tree{reinterpret_cast<u32*>(lds.data()) + (RCX_TREE_PLANAR ? 1 : 4) * lane};
        tree.reset();
        EncLane e;
        e.begin(slots, (u32)(b * slot), (u32)slot, len);
        for (uint32_t i = 0; i < len; ++i) e.step(tree, src[at + i], tab[i]);
        sizes[b] = e.finish();
        overflow |= (int)e.overflow;
    }
    return overflow;
}

// One stream into a sink of `sink16` bytes (already rounded as MemoryStream does): replays the
// TRACK pass of rcx_stream_encode.  out[0] = failing symbol or 0xFFFFFFFF, out[1] = flush fails,
// out[2] = full stream size.
void sim_stream_encode_track(const uint8_t* src, uint32_t n, uint64_t sink16, uint8_t* slot, uint64_t slot_bytes, uint32_t* out)
{
    std::vector<U4> lds((RCX_GROUPS + 1) * RCX_LANES);
    Tree tree{reinterpret_cast<u32*>(lds.data()) + (RCX_TREE_PLANAR ? 1 : 4) * 3};
    tree.reset();
    EncLane e;
    e.begin(slot, 0, (u32)slot_bytes, n);
    e.trk_cap = (u32)(sink16 - 4);
    for (uint32_t i = 0; i < n; ++i) e.step<true>(tree, src[i], rcx_make_div_entry(256 + i), i);
    out[0] = e.trk_fail_at;
    out[1] = e.track_flush_fails() ? 1u : 0u;
    out[2] = e.finish();
}

// One stream decoded for `count` symbols; returns short_at.
uint32_t sim_stream_decode_track(const uint8_t* comp, uint64_t comp_size, uint32_t count, uint8_t* dst)
{
    std::vector<U4> lds((RCX_GROUPS + 1) * RCX_LANES);
    Tree tree{reinterpret_cast<u32*>(lds.data()) + (RCX_TREE_PLANAR ? 1 : 4) * 9};
    tree.reset();
    std::vector<uint8_t> pad(comp_size + 64);
    uint8_t* base = pad.data();
    while (((uintptr_t)base & 15) != 0) ++base;
    base += 16 + 5; // any alignment will do; 5 exercises the skewed start
    memcpy(base, comp, comp_size);
    std::vector<u32> ringmem(RCX_RING_DW * RCX_LANES);
    DecLane d;
    d.begin(base, base + comp_size, ringmem.data() + 9);
    for (uint32_t i = 0; i < count; ++i) {
        if ((i & 15u) == 0) d.topup();
        dst[i] = (uint8_t)d.step<true>(tree, rcx_make_div_entry(256 + i), i, comp_size);
    }
    return d.short_at;
}

// One stream of any length through the *_long steps (the lane's own total, true division, halving at 2^24:
// what rcx_stream_encode / rcx_stream_decode run past RCX_MAX_BLOCK symbols).  Returns the stream size.
uint32_t sim_stream_encode_long(const uint8_t* src, uint32_t n, uint8_t* slot, uint64_t slot_bytes)
{
    std::vector<U4> lds((RCX_GROUPS + 1) * RCX_LANES);
    Tree tree{reinterpret_cast<u32*>(lds.data()) + (RCX_TREE_PLANAR ? 1 : 4) * 11};
    tree.reset();
    EncLane e;
    e.begin(slot, 0, (u32)slot_bytes, n);
    u32 total = 256;
    for (uint32_t i = 0; i < n; ++i) e.step_long(tree, src[i], total, i);
    const uint32_t size = e.finish();
    return e.overflow ? 0 : size;
}

uint32_t sim_stream_decode_long(const uint8_t* comp, uint64_t comp_size, uint32_t count, uint8_t* dst)
{
    std::vector<U4> lds((RCX_GROUPS + 1) * RCX_LANES);
    Tree tree{reinterpret_cast<u32*>(lds.data()) + (RCX_TREE_PLANAR ? 1 : 4) * 12};
    tree.reset();
    std::vector<uint8_t> pad(comp_size + 64);
    uint8_t* base = pad.data();
    while (((uintptr_t)base & 15) != 0) ++base;
    base += 16 + 3;
    memcpy(base, comp, comp_size);
    std::vector<u32> ringmem(RCX_RING_DW * RCX_LANES);
    DecLane d;
    d.begin(base, base + comp_size, ringmem.data() + 12);
    DivEntry k;
    k.mul = k.add = k.shift = 0;
    k.total = 256;
    for (uint32_t i = 0; i < count; ++i) {
        if ((i & 15u) == 0) d.topup();
        dst[i] = (uint8_t)d.step<true, true>(tree, k, i, comp_size);
        k.total += 1;
        if (k.total >= RCX_HALVE_AT) k.total = tree.halve();
    }
    return d.short_at;
}

// comp/offsets as the encoder's compacted output.  Returns 0, or 1 + index of the first bad block.
uint64_t sim_decode_blocks(const uint8_t* comp, const uint64_t* offsets, uint64_t nblocks, uint32_t block, uint64_t n, uint8_t* dst, uint32_t lane)
{
    std::vector<U4> lds((RCX_GROUPS + 1) * RCX_LANES);
    std::vector<DivEntry> tab(block + 2 * RCX_STAGE);
    for (size_t i = 0; i < tab.size(); ++i) tab[i] = rcx_make_div_entry((u32)(256 + i));
    for (uint64_t b = 0; b < nblocks; ++b) {
        uint64_t at = b * block;
        uint32_t len = (uint32_t)((n - at) < block ? (n - at) : block);
        uint64_t s0 = offsets[b], s1 = offsets[b + 1];
        if (s1 < s0 || s1 - s0 < 9) return b + 1;
        Tree tree{reinterpret_cast<u32*>(lds.data()) + (RCX_TREE_PLANAR ? 1 : 4) * lane};
        tree.reset();
        DecLane d;
        // the device code loads aligned 16-byte pieces that may start before / end after the stream:
        // give it a padded private copy at the same alignment
        std::vector<uint8_t> pad(s1 - s0 + 64);
        uint32_t skew = (uint32_t)(s0 & 15);
        uint8_t* base = pad.data();
        while (((uintptr_t)base & 15) != 0) ++base;
        base += 16 + skew;
        memcpy(base, comp + s0, s1 - s0);
        std::vector<u32> ringmem(RCX_RING_DW * RCX_LANES);
        u32 declared = d.begin(base, base + (s1 - s0), ringmem.data() + lane);
        if (declared != len) return b + 1;
        for (uint32_t i = 0; i < len; ++i) {
            if ((i & 15u) == 0) d.topup();
            dst[at + i] = (uint8_t)d.step(tree, tab[i]);
        }
        if (d.taken() > s1 - s0) return b + 1;
    }
    return 0;
}

// The row index of a periodic 32 KiB block (period p, a power of two), by the replay of the reference's sort that the
// GPU runs for such blocks (rcx_bwt_tie.hpp).  -> the row, or 0xFFFFFFFF if the part stack overflowed.
// Both replays over a block of `depth` rows (a power of two, so that smaller cases run fast) -> 0 if every row ended up
// in the same place, 1 if not, 2 if a stack overflowed.
uint32_t sim_bwt_tie_compare(const uint8_t* word, uint32_t p, uint32_t depth)
{
    std::vector<uint16_t> a(depth), b(depth);
    for (uint32_t i = 0; i < depth; ++i) a[i] = b[i] = (uint16_t)i;
    RcxTieSort::Part sa[RCX_TIE_STACK];
    scalar_tie::ScalarTieSort::Part sb[RCX_TIE_STACK];
    RcxTieSort ta{a.data(), word, p - 1, depth};
    std::vector<uint16_t> rank(p), tmp(p), order(p), order2(p), hist(256 * 64), sums(64);
    ta.rank_classes(rank.data(), tmp.data(), order.data(), order2.data(), hist.data(), sums.data());
    for (uint32_t x = 0; x < p; ++x)      // the ranks must order the classes as the byte-by-byte less() does
        for (uint32_t y = x + 1; y < p && y < x + 40; ++y)
            if ((rank[x] < rank[y]) != ta.less(x, y) || (rank[y] < rank[x]) != ta.less(y, x)) return 3;
    ta.rank = rank.data();
    scalar_tie::ScalarTieSort tb{b.data(), word, p - 1, depth};
    if (!ta.run(sa) || !tb.run(sb)) return 2;
    return a == b ? 0 : 1;
}

uint32_t sim_bwt_tie_row(const uint8_t* block, uint32_t p)
{
    std::vector<uint16_t> rows(32768);
    for (uint32_t i = 0; i < 32768; ++i) rows[i] = (uint16_t)i;
    RcxTieSort::Part stack[RCX_TIE_STACK];
    RcxTieSort t{rows.data(), block, p - 1, 32768u};
    std::vector<uint16_t> rank(p), tmp(p), sa(p), sa2(p), hist(256 * 64), sums(64); // as rcx_bwt_tie_k does
    t.rank_classes(rank.data(), tmp.data(), sa.data(), sa2.data(), hist.data(), sums.data());
    t.rank = rank.data();
    if (!t.run(stack)) return 0xFFFFFFFFu;
    return t.row_of_zero();
}

} // extern "C"
