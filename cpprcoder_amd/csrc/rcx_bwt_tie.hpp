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
// One WAVE replays a block.  The moves are sequential by nature (every swap depends on the ones before), so all 64 lanes
// run the same scalar code on the same values -- a wave executes in lockstep and the rows live in LDS, so 64 equal
// stores to one address are one store -- and the lanes part only where a stretch of the work has no dependencies:
//   * a partition pass looks at 64 rows at a time (one byte compare per lane, two ballots): rows that only move a
//     pointer (smaller ones in the scan from the left, larger ones in the scan from the right) are passed 64 at a time,
//     and a run of rows EQUAL to the pivot byte -- each of which the reference swaps to the end of its scan's zone, one
//     after the other -- is placed in one step: the run goes to the front, the zone's other rows behind it, rotated by
//     the run's length (what the swaps add up to);
//   * the exchange of the equal rows into the middle (blksort.h:327-336) swaps disjoint pairs, 64 at a time;
//   * less() finds the first byte that differs 64 bytes at a time, one_class() checks 64 rows at a time.
// That bounds the bad case: a period that is one long run and a single other byte, (a^16383 b)^2, makes the reference's
// sort run 16383 passes over ~32768 rows each, two rows leaving per pass -- 4e8 dependent steps for one lane (measured
// in tens of seconds), 16383 x 512 wave steps here (tests/bwt_cases.py "a run then b").
//
// Compiles as __device__ code and, for tests/sim/lane_sim.cpp, as plain C++ with the 64 lanes as a loop (the CPU suite
// checks it against the rows the reference stored).
#pragma once
#include "rcx_lane.hpp"

#define RCX_TIE_STACK 64

// The wave: on the device the 64 threads of the workgroup, on the host a loop over 64 "lanes".
struct RcxTieWave {
    static constexpr u32 W = 64;
#if defined(__HIP_DEVICE_COMPILE__)
    template <class P>
    static __device__ u64 ballot(P pred) { return __ballot(pred(threadIdx.x) ? 1 : 0); }
    template <class F>
    static __device__ void each(F f) { f(threadIdx.x); }
    static __device__ void sync() { __syncthreads(); } // (one wave: orders its LDS accesses for the compiler)
#else
    template <class P>
    static u64 ballot(P pred)
    {
        u64 m = 0;
        for (u32 l = 0; l < W; ++l)
            if (pred(l)) m |= 1ull << l;
        return m;
    }
    template <class F>
    static void each(F f)
    {
        for (u32 l = 0; l < W; ++l) f(l);
    }
    static void sync() {}
#endif
    static RCX_HD u32 first(u64 m) { return (u32)__builtin_ctzll(m); }
};

struct RcxTieSort {
    uint16_t* rows;   // 32768 rows, initial order 0, 1, 2, ...
    const u8* word;   // the period
    u32 pmask;        // p - 1
    u32 depth;        // 32768
    const uint16_t* rank = nullptr; // rank[c] = place of class c among the p rotations of the word (rank_classes), or none

    RCX_HD u32 byte_at(u32 row, u32 d) const { return word[(row + d) & pmask]; }

    // blksort.h:183-211 over the full depth
    RCX_HD bool less(u32 a, u32 b) const
    {
        const u32 ca = a & pmask, cb = b & pmask;
        if (ca == cb) return false;
        if (rank) return rank[ca] < rank[cb]; // rows compare as their classes' rotations do (a primitive period: all distinct)
        for (u32 d0 = 0; d0 <= pmask; d0 += RcxTieWave::W) {
            const u64 differ = RcxTieWave::ballot([&](u32 l) { return d0 + l <= pmask && word[(ca + d0 + l) & pmask] != word[(cb + d0 + l) & pmask]; });
            if (differ) {
                const u32 d = d0 + RcxTieWave::first(differ);
                return word[(ca + d) & pmask] < word[(cb + d) & pmask];
            }
        }
        return false; // not reached: two classes of a primitive period differ within it
    }

    // rank[c] = the place of rotation c among the p rotations of `word`, by prefix doubling on the circle (the ranks of
    // the 2h-byte pieces from the ranks of the h-byte pieces, Manber-Myers) with the wave as the sorter: a round sorts
    // the p classes by (rank[c], rank[c + h]) with a stable counting pass per 8-bit digit -- lane l counts and places
    // the l-th stretch of the list, the counts scanned in (digit, lane) order -- and gives equal pairs equal new ranks.
    // The reference's less() (blksort.h:183-211) walks two rows byte by byte; with a word like a^16383 b nearly every
    // such walk is thousands of bytes long, and its heapsort fallback makes a million of them (tests/bwt_cases.py).
    // sa, sa2, tmp: p entries each; hist: 256 x 64 entries (it may share its space with tmp); sums: 64 entries.
    RCX_HD void rank_classes(uint16_t* rk, uint16_t* tmp, uint16_t* sa, uint16_t* sa2, uint16_t* hist, uint16_t* sums) const
    {
        const u32 p = pmask + 1;
        const u32 chunk = (p + RcxTieWave::W - 1) / RcxTieWave::W;
        RcxTieWave::sync();
        RcxTieWave::each([&](u32 l) {
            for (u32 i = l; i < p; i += RcxTieWave::W) rk[i] = word[i];
        });
        RcxTieWave::sync();
        u32 top = 255; // the largest rank in use
        for (u32 h = 1; h < p; h <<= 1) {
            RcxTieWave::each([&](u32 l) {
                for (u32 i = l; i < p; i += RcxTieWave::W) sa[i] = (uint16_t)i;
            });
            const u32 digits = top > 255 ? 2 : 1;
            for (u32 pass = 0; pass < 2 * digits; ++pass) { // the second halves' digits first (least significant)
                const bool second = pass < digits;
                const u32 shift = 8 * (pass % digits);
                auto digit = [&](u32 c) -> u32 { return ((u32)rk[second ? (c + h) & pmask : c] >> shift) & 255u; };
                RcxTieWave::sync();
                RcxTieWave::each([&](u32 l) {
                    for (u32 d = 0; d < 256; ++d) hist[d * RcxTieWave::W + l] = 0;
                });
                RcxTieWave::sync();
                RcxTieWave::each([&](u32 l) {
                    const u32 from = l * chunk, to = from + chunk < p ? from + chunk : p;
                    for (u32 k = from; k < to; ++k) hist[digit(sa[k]) * RcxTieWave::W + l] += 1;
                });
                RcxTieWave::sync();
                // exclusive running sums over the 256 x 64 counts as they lie: lane l sums entries [256 l, 256 l + 256), ...
                RcxTieWave::each([&](u32 l) {
                    u32 sum = 0;
                    for (u32 e = 256 * l; e < 256 * l + 256; ++e) sum += hist[e];
                    sums[l] = (uint16_t)sum;
                });
                RcxTieWave::sync();
                RcxTieWave::each([&](u32 l) { // ... starts from the sum of the lanes before it, ...
                    u32 before = 0;
                    for (u32 m = 0; m < l; ++m) before += sums[m];
                    u32 run = before;
                    for (u32 e = 256 * l; e < 256 * l + 256; ++e) {
                        const u32 c = hist[e];
                        hist[e] = (uint16_t)run;
                        run += c;
                    }
                });
                RcxTieWave::sync();
                RcxTieWave::each([&](u32 l) { // ... and places its stretch in order
                    const u32 from = l * chunk, to = from + chunk < p ? from + chunk : p;
                    for (u32 k = from; k < to; ++k) {
                        const uint16_t c = sa[k];
                        const u32 at = digit(c) * RcxTieWave::W + l;
                        const u32 where = hist[at];
                        hist[at] = (uint16_t)(where + 1);
                        sa2[where] = c;
                    }
                });
                RcxTieWave::sync();
                uint16_t* t = sa;
                sa = sa2;
                sa2 = t;
            }
            // new ranks: equal pairs share one
            auto differs = [&](u32 k) -> u32 { // sa[k] against sa[k - 1]
                const u32 c = sa[k], b = sa[k - 1];
                return (rk[c] != rk[b] || rk[(c + h) & pmask] != rk[(b + h) & pmask]) ? 1u : 0u;
            };
            RcxTieWave::each([&](u32 l) {
                const u32 from = l * chunk, to = from + chunk < p ? from + chunk : p;
                u32 changes = 0;
                for (u32 k = from > 0 ? from : 1; k < to; ++k) changes += differs(k);
                sums[l] = (uint16_t)changes;
            });
            RcxTieWave::sync();
            RcxTieWave::each([&](u32 l) {
                const u32 from = l * chunk, to = from + chunk < p ? from + chunk : p;
                u32 r = 0;
                for (u32 m = 0; m < l; ++m) r += sums[m];
                for (u32 k = from; k < to; ++k) {
                    if (k > 0) r += differs(k);
                    tmp[sa[k]] = (uint16_t)r;
                }
            });
            RcxTieWave::sync();
            u32 all = 0;
            for (u32 m = 0; m < RcxTieWave::W; ++m) all += sums[m];
            top = all;
            RcxTieWave::each([&](u32 l) {
                for (u32 i = l; i < p; i += RcxTieWave::W) rk[i] = tmp[i];
            });
            RcxTieWave::sync();
            if (top == p - 1) break; // every class has its own rank
        }
    }

    RCX_HD bool one_class(const uint16_t* v, u32 size) const
    {
        const u32 c = v[0] & pmask;
        for (u32 i0 = 1; i0 < size; i0 += RcxTieWave::W)
            if (RcxTieWave::ballot([&](u32 l) { return i0 + l < size && (v[i0 + l] & pmask) != c; })) return false;
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

    // One scan of the partition loop (blksort.h:302-323), DIR = +1 from the left, -1 from the right with every index
    // mirrored (j stands for row last - j): `a` is the scan's own pointer, `b` the other scan's (it stops past it), `ea`
    // the end of its zone of rows equal to the pivot byte.  Rows [ea, a) are the zone's other rows (smaller ones on
    // the left, larger ones on the right): a queue -- an equal row swaps with its front, which goes to the back.
    template <int DIR>
    RCX_HD void scan(uint16_t* v, s32 last, u32 d, u32 p, s32& a, s32 b, s32& ea) const
    {
        auto at = [&](s32 j) -> uint16_t& { return v[DIR > 0 ? j : last - j]; };
        while (a <= b) {
            const u32 cnt = (u32)(b - a + 1) < RcxTieWave::W ? (u32)(b - a + 1) : RcxTieWave::W;
            // the rows the scan stops at (larger than the pivot byte from the left, smaller from the right), the equal ones
            const u64 stops = RcxTieWave::ballot([&](u32 l) {
                if (l >= cnt) return false;
                const u32 c = byte_at(at(a + (s32)l), d);
                return DIR > 0 ? p < c : c < p;
            });
            u64 equal = RcxTieWave::ballot([&](u32 l) { return l < cnt && byte_at(at(a + (s32)l), d) == p; });
            const u32 stop = stops ? RcxTieWave::first(stops) : cnt;
            if (stop < 64) equal &= (1ull << stop) - 1;
            u32 pos = 0; // rows of the window dealt with
            while (equal) {
                const u32 s = RcxTieWave::first(equal);
                a += (s32)(s - pos); // rows that only move the pointer
                const u64 after = ~(equal >> s);
                const u32 run = after ? RcxTieWave::first(after) : 64 - s;
                const s32 m = a - ea; // the queue in front of the run
                if (m > 0) {
                    if (m <= (s32)RcxTieWave::W) {
                        rotate_in<DIR>(v, last, ea, (u32)m, run);
                    } else {
                        for (u32 i = 0; i < run; ++i) { // the reference's swaps, one by one
                            const uint16_t t = at(a + (s32)i);
                            at(a + (s32)i) = at(ea + (s32)i);
                            at(ea + (s32)i) = t;
                        }
                    }
                }
                a += (s32)run;
                ea += (s32)run;
                pos = s + run;
                equal &= run + s >= 64 ? 0ull : ~0ull << (run + s);
            }
            a += (s32)(stop - pos);
            if (stop < cnt) return; // the row at `a` stops the scan
        }
    }

    // `run` equal rows behind a queue of m <= 64 rows at [ea, ea + m): what the reference's `run` swaps add up to is the
    // run in front and the queue behind it, rotated by run mod m.
    template <int DIR>
    RCX_HD void rotate_in(uint16_t* v, s32 last, s32 ea, u32 m, u32 run) const
    {
        auto at = [&](s32 j) -> uint16_t& { return v[DIR > 0 ? j : last - j]; };
        const u32 k = run % m;
#if defined(__HIP_DEVICE_COMPILE__)
        const u32 l = threadIdx.x;
        const uint16_t e = l < run ? at(ea + (s32)(m + l)) : (uint16_t)0;
        const uint16_t q = l < m ? at(ea + (s32)l) : (uint16_t)0;
        RcxTieWave::sync();
        if (l < run) at(ea + (s32)l) = e;
        if (l < m) at(ea + (s32)(run + (l + m - k) % m)) = q;
        RcxTieWave::sync();
#else
        uint16_t e[RcxTieWave::W], q[RcxTieWave::W];
        for (u32 l = 0; l < run; ++l) e[l] = at(ea + (s32)(m + l));
        for (u32 l = 0; l < m; ++l) q[l] = at(ea + (s32)l);
        for (u32 l = 0; l < run; ++l) at(ea + (s32)l) = e[l];
        for (u32 l = 0; l < m; ++l) at(ea + (s32)(run + (l + m - k) % m)) = q[l];
#endif
    }

    // v[first + i] <-> v[second - i] for i < count; the two stretches do not overlap.
    RCX_HD void swap_ranges(uint16_t* v, s32 first, s32 second, s32 count) const
    {
        RcxTieWave::sync();
        for (s32 i0 = 0; i0 < count; i0 += (s32)RcxTieWave::W)
            RcxTieWave::each([&](u32 l) {
                const s32 i = i0 + (s32)l;
                if (i < count) swap_rows(v, first + i, second - i);
            });
        RcxTieWave::sync();
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
                // while (lo <= hi) { c = byte(v[lo]); if (p < c) break; if (p == c) swap(lo, eq_lo++); ++lo; }
                scan<+1>(v, last, cur.d, p, lo, hi, eq_lo);
                // while (lo <= hi) { c = byte(v[hi]); if (c < p) break; if (p == c) swap(hi, eq_hi--); --hi; } -- the same, mirrored
                s32 from_top = last - hi, to_top = last - lo, eq_top = last - eq_hi;
                scan<-1>(v, last, cur.d, p, from_top, to_top, eq_top);
                hi = last - from_top;
                eq_hi = last - eq_top;
                if (hi < lo) break;
                swap_rows(v, lo, hi);
                ++lo;
                --hi;
            }
            const s32 below = lo - eq_lo, above = eq_hi - hi; // blksort.h:327-336
            const s32 r0 = eq_lo < below ? eq_lo : below;
            swap_ranges(v, 0, hi, r0);          // v[i] <-> v[hi - i], i < r0: disjoint pairs (2 r0 <= lo)
            const s32 right_eq = last - eq_hi;
            const s32 r1 = right_eq < above ? right_eq : above;
            swap_ranges(v + lo, 0, last - lo, r1); // v[lo + i] <-> v[last - i], i < r1
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
