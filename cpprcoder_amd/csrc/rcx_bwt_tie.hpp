// rcx_bwt_tie.hpp -- the row index of a PERIODIC block, as the reference's sort leaves it.
//
// The transformed bytes of a block (blksort.h:511-518) do not depend on how ties between equal rotations are broken;
// the 16-bit row index behind them does.  Rotations tie only when the block is periodic (period p, a power of two
// below 32768: rotation r equals rotation r + p over the whole depth), and then the row the reference stores is
// wherever its unstable sort -- multikey quicksort with a median-of-three pivot on the FIRST byte, insertion sort below
// 37 rows, heapsort after 11 levels (blksort.h:168-363) -- happens to leave row 0 of the initial order 0, 1, 2, ...
// The block-sort kernel (rcx_bwt.hpp) sorts by prefix doubling and finds such blocks by their ties; this file replays
// the reference's moves for them, one lane per block, so that the row is the reference's bit for bit.
//
// What makes that affordable: a row is known by its class r mod p, and all comparisons see only the class.
//   * byte d of row r is word[(r + d) mod p], word = the first p bytes of the block;
//   * less(a, b) is false at once for rows of one class (the reference compares all 32768 bytes to find that out)
//     and decided within p bytes otherwise;
//   * a part whose rows are all of one class is left exactly as it is by every further partition pass (each row equals
//     the pivot byte and is swapped with itself, blksort.h:304-312) and by the insertion sort, so it is dropped.
// The three parts of a partition are independent of each other, so they are worked off from an explicit stack,
// smallest first (at most 2 log2(32768) entries), instead of by recursion.
//
// Compiles as __device__ code and, for tests/sim/lane_sim.cpp, as plain C++ (the CPU suite checks it against the oracle).
#pragma once
#include "rcx_lane.hpp"

#define RCX_TIE_STACK 64

struct RcxTieSort {
    uint16_t* rows;   // 32768 rows, initial order 0, 1, 2, ...
    const u8* word;   // the period
    u32 pmask;        // p - 1
    u32 depth;        // 32768

    RCX_HD u32 byte_at(u32 row, u32 d) const { return word[(row + d) & pmask]; }

    // blksort.h:183-211 over the full depth
    RCX_HD bool less(u32 a, u32 b) const
    {
        const u32 ca = a & pmask, cb = b & pmask;
        if (ca == cb) return false;
        for (u32 d = 0; d <= pmask; ++d) {
            const u32 x = word[(ca + d) & pmask], y = word[(cb + d) & pmask];
            if (x != y) return x < y;
        }
        return false; // not reached: two classes of a primitive period differ within it
    }

    RCX_HD bool one_class(const uint16_t* v, u32 size) const
    {
        const u32 c = v[0] & pmask;
        for (u32 i = 1; i < size; ++i)
            if ((v[i] & pmask) != c) return false;
        return true;
    }

    // blksort.h:168-181
    RCX_HD u32 pivot_row(const uint16_t* v, u32 size) const
    {
        const u32 q = size >> 2;
        const u32 a = byte_at(v[q], 0), b = byte_at(v[2 * q], 0), c = byte_at(v[3 * q], 0);
        if (a < b) return b < c ? v[2 * q] : (a < c ? v[3 * q] : v[q]);
        return a < c ? v[q] : (b < c ? v[3 * q] : v[2 * q]);
    }

    // blksort.h:225-235
    RCX_HD void insertion(uint16_t* v, u32 size) const
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
    RCX_HD void sift(uint16_t* h, s32 i, s32 n, uint16_t x) const
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
    RCX_HD void heap(uint16_t* v, u32 size) const
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

    static RCX_HD void swap_rows(uint16_t* v, s32 a, s32 b)
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
    RCX_HD bool run(Part* stack) const
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
    RCX_HD u32 row_of_zero() const
    {
        for (u32 i = 0; i < depth; ++i)
            if (rows[i] == 0) return i;
        return 0;
    }
};
