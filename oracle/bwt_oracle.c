/*
 * bwt_oracle.c -- CPU restatement of the reference's block sort (blksort.h): the Burrows-Wheeler transform of
 * 32 KiB blocks that the reference harness puts in front of its entropy coders (test/main.cpp:961-986).
 *
 * TEST INFRASTRUCTURE ONLY (see rc_oracle.h): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this; the product never links or calls it.
 *
 * Parity status: PINNED.  Checked byte for byte against the reference itself compiled from /root/reference/blksort.h
 * (oracle/_ref/libblksort_ref.so, oracle/Makefile) and against tests/golden/bwt.json, generated from that build by
 * tests/golden/make_golden_bwt.py.
 *
 * What has to be restated exactly, and why: the transformed bytes (the last column of the sorted rotations) do not
 * depend on how the sort breaks ties, but the 16-bit index stored behind them -- the row of the unrotated block --
 * does when the block is periodic (rotations that are equal over the whole depth).  The reference's sort is not
 * stable, so the row it reports for such a block is a property of its particular multikey quicksort / insertion sort
 * / heapsort (blksort.h:233-363) applied to the rows in their initial order 0,1,2,...  This file therefore restates
 * that sort move for move, on row numbers instead of (pointer, id) pairs: a row IS its start offset, the reference's
 * Item::id_ and Item::str_ - buffer are the same number (blksort.h:466-487).
 *
 * BLOCKSORT_MTF is 0 in the reference (blksort.h:54): the move-to-front stage is compiled out and is not restated.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BWT_BLOCK 32768u            /* blksort.h:82 */
#define BWT_ENCODED (BWT_BLOCK + 2) /* blksort.h:85 */

typedef struct {
    const uint8_t* text; /* the block twice over (blksort.h:462-463): row r reads text[r .. r + BWT_BLOCK) */
    uint16_t* row;       /* the rows being sorted */
} bwt_sort;

/* blksort.h:183-211: is row a strictly below row b over `depth` bytes? */
static int row_less(const bwt_sort* s, uint16_t a, uint16_t b, uint32_t depth)
{
    const uint8_t *x = s->text + a, *y = s->text + b;
    for (uint32_t d = 0; d < depth; ++d)
        if (x[d] != y[d]) return x[d] < y[d];
    return 0;
}

/* blksort.h:168-181: the pivot row -- median of the FIRST bytes of the rows at 1/4, 2/4, 3/4 */
static uint16_t pivot_row(const bwt_sort* s, const uint16_t* v, uint32_t size)
{
    const uint32_t q = size >> 2;
    const uint8_t a = s->text[v[q]], b = s->text[v[2 * q]], c = s->text[v[3 * q]];
    if (a < b) return b < c ? v[2 * q] : (a < c ? v[3 * q] : v[q]);
    return a < c ? v[q] : (b < c ? v[3 * q] : v[2 * q]);
}

/* blksort.h:225-235 */
static void insertion(const bwt_sort* s, uint16_t* v, uint32_t size, uint32_t depth)
{
    for (uint32_t i = 1; i < size; ++i) {
        const uint16_t x = v[i];
        int64_t j = (int64_t)i - 1;
        while (j >= 0 && row_less(s, x, v[j], depth)) {
            v[j + 1] = v[j];
            --j;
        }
        v[j + 1] = x;
    }
}

/* blksort.h:237-279: 1-based heap; the sift-down is the same in both halves */
static void sift(const bwt_sort* s, uint16_t* h /* 1-based */, int32_t i, int32_t n, uint16_t x, uint32_t depth)
{
    int32_t j;
    while ((j = i << 1) <= n) {
        if (j < n && row_less(s, h[j], h[j + 1], depth)) ++j;
        if (!row_less(s, x, h[j], depth)) break;
        h[i] = h[j];
        i = j;
    }
    h[i] = x;
}
static uint64_t heap_calls; /* test instrumentation: how often the sort fell back to heapsort */
uint64_t rco_bwt_heap_calls(void) { return heap_calls; }

static void heap(const bwt_sort* s, uint16_t* v, uint32_t size, uint32_t depth)
{
    ++heap_calls;
    uint16_t* h = v - 1;
    int32_t n = (int32_t)size;
    for (int32_t k = n >> 1; k >= 1; --k) sift(s, h, k, n, h[k], depth);
    while (n > 1) {
        const uint16_t x = h[n];
        h[n] = h[1];
        --n;
        sift(s, h, 1, n, x, depth);
    }
}

static void swap_rows(uint16_t* v, int32_t a, int32_t b)
{
    const uint16_t t = v[a];
    v[a] = v[b];
    v[b] = t;
}

/* blksort.h:281-350: three-way partition on byte d around the pivot's byte (equal keys parked at both ends, then
 * swapped to the middle), recursion on the two outer parts with one level less, the middle part goes on with the
 * next byte.  Fewer than 37 rows: insertion sort over the whole depth; no levels left: heapsort. */
static void multikey(const bwt_sort* s, uint16_t* v, uint32_t size, uint32_t d, uint32_t depth, int32_t level)
{
    if (level <= 0) {
        heap(s, v, size, depth);
        return;
    }
    while (d < depth) {
        if (size < 37) {
            insertion(s, v, size, depth);
            return;
        }
        const uint8_t p = s->text[(uint32_t)pivot_row(s, v, size) + d];
        const int32_t last = (int32_t)size - 1;
        int32_t lo = 0, hi = last, eq_lo = 0, eq_hi = last;
        for (;;) {
            while (lo <= hi) {
                const uint8_t c = s->text[(uint32_t)v[lo] + d];
                if (p < c) break;
                if (p == c) swap_rows(v, lo, eq_lo++);
                ++lo;
            }
            while (lo <= hi) {
                const uint8_t c = s->text[(uint32_t)v[hi] + d];
                if (c < p) break;
                if (p == c) swap_rows(v, hi, eq_hi--);
                --hi;
            }
            if (hi < lo) break;
            swap_rows(v, lo, hi);
            ++lo;
            --hi;
        }
        const int32_t left_eq = eq_lo, below = lo - eq_lo;
        const int32_t r0 = left_eq < below ? left_eq : below;
        for (int32_t i = 0; i < r0; ++i) swap_rows(v, i, hi - i);
        const int32_t right_eq = last - eq_hi, above = eq_hi - hi;
        const int32_t r1 = right_eq < above ? right_eq : above;
        for (int32_t i = 0; i < r1; ++i) swap_rows(v, lo + i, last - i);
        const int32_t m0 = below;              /* rows below the pivot byte: [0, m0) */
        const int32_t m1 = last - above + 1;   /* rows above it: [m1, size) */
        if (m0 - 1 > 0) multikey(s, v, (uint32_t)m0, d, depth, level - 1);
        if (m1 < last) multikey(s, v + m1, (uint32_t)((int32_t)size - m1), d, depth, level - 1);
        if (m1 <= m0) break;
        v += m0;
        size = (uint32_t)(m1 - m0);
        ++d;
    }
}

/* blksort.h:455-541 (BlkSort::encode_internal): out gets BWT_BLOCK transformed bytes + the row of the block, LE16 */
static void bwt_block_forward(const uint8_t* in, uint8_t* out, uint8_t* twice, uint16_t* rows)
{
    memcpy(twice, in, BWT_BLOCK);
    memcpy(twice + BWT_BLOCK, in, BWT_BLOCK);
    for (uint32_t i = 0; i < BWT_BLOCK; ++i) rows[i] = (uint16_t)i;
    bwt_sort s = {twice, rows};
    multikey(&s, rows, BWT_BLOCK, 0, BWT_BLOCK, 11); /* blksort.h:352-363: the level is fixed at 11 */
    uint16_t at = 0;
    for (uint32_t i = 0; i < BWT_BLOCK; ++i) {
        out[i] = twice[(uint32_t)rows[i] + BWT_BLOCK - 1];
        if (rows[i] == 0) at = (uint16_t)i;
    }
    out[BWT_BLOCK] = (uint8_t)(at & 0xFF); /* memcpy of a uint16_t on the reference's little-endian hosts (blksort.h:518) */
    out[BWT_BLOCK + 1] = (uint8_t)(at >> 8);
}

/* blksort.h:543-679 (BlkSort::decode_internal) with counting_sort (blksort.h:365-397): `next[r]` is where the
 * r-th smallest byte of the block sits, equal bytes in their order of appearance */
static void bwt_block_inverse(const uint8_t* in, uint8_t* out, uint16_t* next)
{
    uint32_t count[257];
    memset(count, 0, sizeof count);
    for (uint32_t i = 0; i < BWT_BLOCK; ++i) ++count[in[i]];
    for (uint32_t c = 1; c < 256; ++c) count[c] += count[c - 1]; /* inclusive running sums */
    for (int32_t i = (int32_t)BWT_BLOCK - 1; i >= 0; --i) next[--count[in[i]]] = (uint16_t)i;
    const uint32_t top = (uint32_t)in[BWT_BLOCK] | ((uint32_t)in[BWT_BLOCK + 1] << 8);
    /* a row number past the block makes the reference read outside its arrays (blksort.h:663): here it wraps */
    uint16_t p = next[top & (BWT_BLOCK - 1)];
    for (uint32_t i = 0; i < BWT_BLOCK; ++i) {
        out[i] = in[p];
        p = next[p];
    }
}

uint64_t rco_bwt_encode_bound(uint64_t n) /* blksort.h:426-431 */
{
    const uint64_t blocks = n / BWT_BLOCK;
    return blocks * BWT_ENCODED + (n - blocks * BWT_BLOCK);
}

uint64_t rco_bwt_decode_bound(uint64_t n) /* blksort.h:433-438: a bound, and as the reference computes it simply n */
{
    const uint64_t blocks = n / BWT_BLOCK;
    return blocks * BWT_BLOCK + (n - blocks * BWT_BLOCK);
}

uint64_t rco_bwt_decoded_size(uint64_t n) /* what decode() writes: its block count is n / 32770 (blksort.h:453-454) */
{
    const uint64_t blocks = n / BWT_ENCODED;
    return blocks * BWT_BLOCK + (n - blocks * BWT_ENCODED);
}

/* blksort.h:440-449: whole blocks are transformed, what is left over is copied.  0 = ok. */
int rco_bwt_encode(const uint8_t* src, uint64_t n, uint8_t* dst)
{
    uint8_t* twice = (uint8_t*)malloc(2 * BWT_BLOCK);
    uint16_t* rows = (uint16_t*)malloc(sizeof(uint16_t) * BWT_BLOCK);
    if (!twice || !rows) {
        free(twice);
        free(rows);
        return -1;
    }
    const uint64_t blocks = n / BWT_BLOCK;
    for (uint64_t b = 0; b < blocks; ++b) bwt_block_forward(src + b * BWT_BLOCK, dst + b * BWT_ENCODED, twice, rows);
    memcpy(dst + blocks * BWT_ENCODED, src + blocks * BWT_BLOCK, (size_t)(n - blocks * BWT_BLOCK));
    free(twice);
    free(rows);
    return 0;
}

/* blksort.h:451-462 */
int rco_bwt_decode(const uint8_t* src, uint64_t n, uint8_t* dst)
{
    uint16_t* next = (uint16_t*)malloc(sizeof(uint16_t) * BWT_BLOCK);
    if (!next) return -1;
    const uint64_t blocks = n / BWT_ENCODED;
    for (uint64_t b = 0; b < blocks; ++b) bwt_block_inverse(src + b * BWT_ENCODED, dst + b * BWT_BLOCK, next);
    memcpy(dst + blocks * BWT_BLOCK, src + blocks * BWT_ENCODED, (size_t)(n - blocks * BWT_ENCODED));
    free(next);
    return 0;
}
