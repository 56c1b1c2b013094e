/*
 * rans_oracle.c -- CPU restatement of the reference's rANS coders (cppans.h), SURVEY.md section 8(f)-4.
 *
 * TEST INFRASTRUCTURE ONLY (see rc_oracle.h): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this; the product (cpprcoder_amd/librcx.so) never links or calls it.
 *
 * Parity status: PINNED against the reference itself compiled from /root/reference/cppans.h
 * (oracle/_ref/libcppans_ref.so, oracle/Makefile `ref`) and the golden vectors made from that build
 * (tests/golden/rans.json, tests/golden/make_golden_rans.py).  The reference publishes no rANS numbers
 * (README.md has none), so its own compiled output is the only pin there is.
 *
 * Two stream formats, both [u32 LE n][257 x u32 LE scaled cumulative counts][payload]:
 *   rANS::encode / decode            cppans.h:497-563   one 32-bit state, 14-bit probabilities, byte renormalisation;
 *                                                       payload = [u32 LE final state][bytes ...]
 *   rANS::encode_simd / decode_simd  cppans.h:567-649   eight interleaved states (symbol i belongs to state i & 7),
 *                                                       12-bit probabilities, 16-bit renormalisation;
 *                                                       payload = [8 x u32 LE states][u16 LE words ...]
 * Both encoders run over the input backwards and write backwards from the END of the destination, so the stream is
 * the last `encoded size` bytes of dst (test/main.cpp:384-387).
 */
#include <stdint.h>
#include <string.h>

#define RANS_PROB_BITS 14u            /* cppans.h:27 */
#define RANS_BYTE_LOW (1u << 23)      /* cppans.h:29 */
#define RANS_WORD_LOW (1u << 16)      /* cppans.h:30 */
#define RANS_WORD_BITS 12u            /* cppans.h:31 */
#define RANS_HEADER (258u * 4u)       /* cppans.h:521, :598 */

/* cppans.h:102-128 */
static void rans_count(uint32_t* freqs, uint32_t size, const uint8_t* src)
{
    memset(freqs, 0, 256 * sizeof(uint32_t));
    for (uint32_t i = 0; i < size; ++i) ++freqs[src[i]];
}

/* cppans.h:130-136 */
static void rans_cumulative(uint32_t* cum, const uint32_t* freqs)
{
    cum[0] = 0;
    for (uint32_t i = 0; i < 256; ++i) cum[i + 1] = cum[i] + freqs[i];
}

/* cppans.h:138-178: scale the cumulative counts to target_total, then give every symbol that occurs but lost its
 * range one slot, stolen from the symbol with the smallest range > 1 (the first such in index order). */
static void rans_normalize(uint32_t* freqs, uint32_t* cum, uint64_t target_total)
{
    const uint32_t current_total = cum[256];
    for (uint32_t i = 1; i < 257; ++i) cum[i] = (uint32_t)((target_total * cum[i]) / current_total); /* :142 */
    for (uint32_t i = 0; i < 256; ++i) {
        if (freqs[i] && cum[i + 1] == cum[i]) { /* :145 (freqs still holds the raw counts) */
            uint32_t best_freq = ~0u;
            int32_t best_steal = -1;
            for (int32_t j = 0; j < 256; ++j) {
                const uint32_t freq = cum[j + 1] - cum[j];
                if (1 < freq && freq < best_freq) {
                    best_freq = freq;
                    best_steal = j;
                }
            }
            if ((uint32_t)best_steal < i) { /* :156 */
                for (int32_t j = best_steal + 1; j <= (int32_t)i; ++j) --cum[j];
            } else {
                for (int32_t j = (int32_t)i + 1; j <= best_steal; ++j) ++cum[j];
            }
        }
    }
    for (uint32_t i = 0; i < 256; ++i) freqs[i] = cum[i + 1] - cum[i]; /* :176 */
}

/* ---------------------------------------------------------------------------------------------------------
 * rANS::encode, cppans.h:497-530 (+ put :265-287, flush :289-299, EncSymbol init :180-250).
 * The reference divides with a precomputed reciprocal (Alverson); for every state the encoder can hold that IS the
 * exact quotient (and the freq = 1 special case is exactly x*M + start), so the restatement divides.
 * Returns the encoded size, 0 if dst is too small; the stream is dst[dst_size - size, dst_size).
 * ------------------------------------------------------------------------------------------------------- */
uint32_t rco_rans_encode(uint64_t dst_size, uint8_t* dst, uint32_t src_size, const uint8_t* src)
{
    uint32_t freqs[256], cum[257];
    if (src_size == 0 || dst_size < RANS_HEADER + 4) return 0;
    rans_count(freqs, src_size, src);
    rans_cumulative(cum, freqs);
    rans_normalize(freqs, cum, 1u << RANS_PROB_BITS);
    uint32_t x = RANS_BYTE_LOW; /* :260-263 */
    uint8_t* ptr = dst + dst_size;
    uint8_t* const floor_ = dst + RANS_HEADER + 4; /* the reference checks only at the end (:522); same result */
    for (uint32_t i = src_size; 0 < i; --i) {
        const uint8_t s = src[i - 1];
        const uint32_t freq = freqs[s], start = cum[s];
        const uint32_t x_max = ((RANS_BYTE_LOW >> RANS_PROB_BITS) << 8) * freq; /* :203 */
        while (x_max <= x) {                                                    /* :272-279 */
            if (ptr <= floor_) return 0;
            *--ptr = (uint8_t)(x & 0xFFu);
            x >>= 8;
        }
        x = ((x / freq) << RANS_PROB_BITS) + (x % freq) + start; /* :285-286 == :189 */
    }
    ptr -= 4; /* :289-299 */
    ptr[0] = (uint8_t)(x >> 0);
    ptr[1] = (uint8_t)(x >> 8);
    ptr[2] = (uint8_t)(x >> 16);
    ptr[3] = (uint8_t)(x >> 24);
    ptr -= RANS_HEADER; /* :521-527 */
    if (ptr < dst) return 0;
    memcpy(ptr, &src_size, 4);
    memcpy(ptr + 4, cum, 257 * 4);
    return (uint32_t)(dst + dst_size - ptr);
}

/* rANS::decode, cppans.h:532-564 (+ init_decode :303-310, get :313-316, advance :321-334).
 * Returns the payload bytes consumed (what the reference returns), 0 on refusal; *out_size = symbols written.
 * The reference trusts the table; a table that is not a scaled cumulative table is refused here (returns 0). */
uint32_t rco_rans_decode(uint32_t dst_size, uint8_t* dst, uint32_t src_size, const uint8_t* src, uint32_t* out_size)
{
    static _Thread_local uint8_t cum2sym[1u << RANS_PROB_BITS];
    uint32_t cum[257], original_size;
    if (out_size) *out_size = 0;
    if (src_size < RANS_HEADER + 4) return 0;
    memcpy(&original_size, src, 4);
    if (dst_size < original_size) return 0; /* :541 */
    memcpy(cum, src + 4, 257 * 4);
    if (cum[0] != 0 || cum[256] != (1u << RANS_PROB_BITS)) return 0;
    for (uint32_t s = 0; s < 256; ++s)
        if (cum[s + 1] < cum[s]) return 0;
    for (uint32_t s = 0; s < 256; ++s)
        for (uint32_t i = cum[s]; i < cum[s + 1]; ++i) cum2sym[i] = (uint8_t)s; /* :546-550 */
    const uint8_t* ptr = src + RANS_HEADER;
    const uint8_t* const end = src + src_size;
    uint32_t x = (uint32_t)ptr[0] | ((uint32_t)ptr[1] << 8) | ((uint32_t)ptr[2] << 16) | ((uint32_t)ptr[3] << 24);
    ptr += 4;
    const uint32_t mask = (1u << RANS_PROB_BITS) - 1;
    for (uint32_t i = 0; i < original_size; ++i) {
        const uint8_t s = cum2sym[x & mask];
        dst[i] = s;
        const uint32_t freq = cum[s + 1] - cum[s];
        x = freq * (x >> RANS_PROB_BITS) + (x & mask) - cum[s]; /* :326 */
        while (x < RANS_BYTE_LOW) {                              /* :328-332 */
            if (ptr >= end) return 0; /* (the reference would read past the stream) */
            x = (x << 8) | *ptr++;
        }
    }
    if (out_size) *out_size = original_size;
    return (uint32_t)(ptr - (src + RANS_HEADER));
}

/* ---------------------------------------------------------------------------------------------------------
 * rANS::encode_simd, cppans.h:567-607 (+ wordEncPut :353-364, wordEncFlush :367-373).
 * ------------------------------------------------------------------------------------------------------- */
uint32_t rco_rans8_encode(uint64_t dst_size, uint8_t* dst, uint32_t src_size, const uint8_t* src)
{
    uint32_t freqs[256], cum[257], rans[8];
    if (src_size == 0 || dst_size < RANS_HEADER + 32 || (dst_size & 1u)) return 0;
    rans_count(freqs, src_size, src);
    rans_cumulative(cum, freqs);
    rans_normalize(freqs, cum, 1u << RANS_WORD_BITS);
    for (uint32_t i = 0; i < 8; ++i) rans[i] = RANS_WORD_LOW; /* :585-588 */
    uint8_t* ptr = dst + dst_size;
    uint8_t* const floor_ = dst + RANS_HEADER + 32;
    for (uint32_t i = src_size; 0 < i; --i) {
        const uint8_t s = src[i - 1];
        const uint32_t freq = freqs[s], start = cum[s];
        uint32_t x = rans[(i - 1) & 7];
        if (((RANS_WORD_LOW >> RANS_WORD_BITS) << 16) * freq <= x) { /* :357 (u32 wrap for freq = 4096: 2^32 -> 0: always true) */
            if (ptr <= floor_) return 0;
            ptr -= 2;
            ptr[0] = (uint8_t)(x & 0xFFu); /* u16 store, little endian (x64) */
            ptr[1] = (uint8_t)((x >> 8) & 0xFFu);
            x >>= 16;
        }
        rans[(i - 1) & 7] = ((x / freq) << RANS_WORD_BITS) + (x % freq) + start; /* :363 */
    }
    for (uint32_t i = 8; 0 < i; --i) { /* :595-597 */
        const uint32_t x = rans[i - 1];
        ptr -= 4;
        ptr[0] = (uint8_t)(x >> 0);
        ptr[1] = (uint8_t)(x >> 8);
        ptr[2] = (uint8_t)(x >> 16);
        ptr[3] = (uint8_t)(x >> 24);
    }
    ptr -= RANS_HEADER;
    if (ptr < dst) return 0;
    memcpy(ptr, &src_size, 4);
    memcpy(ptr + 4, cum, 257 * 4);
    return (uint32_t)(dst + dst_size - ptr);
}

/* rANS::decode_simd, cppans.h:609-649 (+ initSymbols :342-351, simdDecSym :412-440, simdDecRenorm :443-488,
 * wordDecSym :384-392).  Groups of 8 symbols: all eight states step, then those below 2^16 take one word each in
 * state order; the last n mod 8 symbols step without renormalising (:643-647).  Returns n (what the reference
 * returns), 0 on refusal. */
uint32_t rco_rans8_decode(uint32_t dst_size, uint8_t* dst, uint32_t src_size, const uint8_t* src, uint32_t* out_size)
{
    static _Thread_local uint8_t slot2sym[1u << RANS_WORD_BITS];
    uint32_t cum[257], original_size, rans[8];
    if (out_size) *out_size = 0;
    if (src_size < RANS_HEADER + 32) return 0;
    memcpy(&original_size, src, 4);
    if (dst_size < original_size) return 0; /* :618 */
    memcpy(cum, src + 4, 257 * 4);
    if (cum[0] != 0 || cum[256] != (1u << RANS_WORD_BITS)) return 0;
    for (uint32_t s = 0; s < 256; ++s)
        if (cum[s + 1] < cum[s]) return 0;
    for (uint32_t s = 0; s < 256; ++s)
        for (uint32_t i = cum[s]; i < cum[s + 1]; ++i) slot2sym[i] = (uint8_t)s; /* :624-626 */
    const uint8_t* ptr = src + RANS_HEADER;
    const uint8_t* const end = src + src_size;
    for (uint32_t k = 0; k < 8; ++k, ptr += 4) /* :405-409 */
        rans[k] = (uint32_t)ptr[0] | ((uint32_t)ptr[1] << 8) | ((uint32_t)ptr[2] << 16) | ((uint32_t)ptr[3] << 24);
    const uint32_t mask = (1u << RANS_WORD_BITS) - 1;
    const uint32_t simd_size = original_size & ~7u;
    for (uint32_t i = 0; i < simd_size; i += 8) {
        for (uint32_t k = 0; k < 8; ++k) { /* :636-639 */
            const uint32_t x = rans[k], slot = x & mask;
            const uint8_t s = slot2sym[slot];
            dst[i + k] = s;
            rans[k] = (cum[s + 1] - cum[s]) * (x >> RANS_WORD_BITS) + (slot - cum[s]); /* freq * (x >> 12) + bias */
        }
        for (uint32_t k = 0; k < 8; ++k) { /* :640-641: states 0..3, then 4..7, one word each where needed */
            if (rans[k] < RANS_WORD_LOW) {
                if (ptr + 2 > end) return 0; /* (the reference would read past the stream) */
                rans[k] = (rans[k] << 16) | (uint32_t)ptr[0] | ((uint32_t)ptr[1] << 8);
                ptr += 2;
            }
        }
    }
    for (uint32_t i = simd_size; i < original_size; ++i) { /* :643-647 */
        const uint32_t x = rans[i & 7], slot = x & mask;
        const uint8_t s = slot2sym[slot];
        dst[i] = s;
        rans[i & 7] = (cum[s + 1] - cum[s]) * (x >> RANS_WORD_BITS) + (slot - cum[s]);
    }
    if (out_size) *out_size = original_size;
    return original_size;
}

/* cppans.h:492-495 */
uint64_t rco_rans_bound(uint32_t size) { return (uint64_t)size * 2 + RANS_HEADER; }

/* ---------------------------------------------------------------------------------------------------------
 * Many independent blocks, as rco_encode_block_range (rc_oracle.c): block b's stream -- what the reference returns
 * for that block alone, i.e. the last `size` bytes of its destination -- is stored at the START of slots + b*slot.
 * simd = 0: rANS::encode / decode, simd = 1: rANS::encode_simd / decode_simd.
 * ------------------------------------------------------------------------------------------------------- */
#include <stdlib.h>

int rco_rans_encode_block_range(const uint8_t* src, uint64_t n, uint32_t block, uint64_t first, uint64_t last,
                                uint8_t* slots, uint64_t slot, uint32_t* sizes, int simd)
{
    /* cppans.h:492-495 leaves no room for the eight flushed states of encode_simd when n < 16 (the reference then
     * writes in front of its destination): the blocks are coded into a destination 64 bytes larger than that */
    const uint64_t cap = rco_rans_bound(block) + 64;
    uint8_t* tmp = (uint8_t*)malloc(cap);
    int ok = tmp != NULL;
    for (uint64_t b = first; ok && b < last; ++b) {
        const uint64_t at = b * block;
        const uint32_t len = (uint32_t)((n - at < block) ? (n - at) : block);
        const uint64_t room = rco_rans_bound(len) + 64;
        const uint32_t size = simd ? rco_rans8_encode(room, tmp, len, src + at) : rco_rans_encode(room, tmp, len, src + at);
        if (size == 0 || size > slot) {
            ok = 0;
            sizes[b] = 0;
        } else {
            memcpy(slots + b * slot, tmp + room - size, size);
            sizes[b] = size;
        }
    }
    free(tmp);
    return ok;
}

int rco_rans_decode_block_range(const uint8_t* slots, uint64_t slot, const uint32_t* sizes, uint32_t block, uint64_t n,
                                uint64_t first, uint64_t last, uint8_t* dst, int simd)
{
    int ok = 1;
    for (uint64_t b = first; b < last; ++b) {
        const uint64_t at = b * block;
        const uint32_t len = (uint32_t)((n - at < block) ? (n - at) : block);
        uint32_t got = 0;
        const uint32_t r = simd ? rco_rans8_decode(len, dst + at, sizes[b], slots + b * slot, &got)
                                : rco_rans_decode(len, dst + at, sizes[b], slots + b * slot, &got);
        if (r == 0 || got != len) ok = 0;
    }
    return ok;
}
