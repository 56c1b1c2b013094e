/*
 * rc_oracle.c -- CPU restatement of the reference's range-coder hot path.
 * TEST INFRASTRUCTURE ONLY (see rc_oracle.h for the rules and the parity status).
 *
 * Written from the semantics of /root/reference/cpprcoder.h (cited per
 * function as file:line); no reference source is included or copied.
 */
#include "rc_oracle.h"

#include <stdlib.h>
#include <string.h>

#define RANGE_FLOOR 0x01000000u /* cpprcoder.h:631 MINRANGE */
#define ENC_RANGE0 0xFFFFFF00u  /* cpprcoder.h:630 encoder MAXRANGE */
#define DEC_RANGE0 0x00FFFFFFu  /* cpprcoder.h:813 decoder MAXRANGE */
#define TOP_GUARD 0xFF000000u   /* cpprcoder.h:784 (0xFFU << SHIFT) */

/* ================================================================== */
/* Sink (MemoryStream)                                                 */
/* ================================================================== */

static int32_t round16(int32_t v) { return (int32_t)(((uint32_t)v + 15u) & ~15u); }

/* cpprcoder.h:964-969: default stream has no storage at all. */
void rco_stream_init(rco_stream* s)
{
    s->capacity = 0;
    s->size = 0;
    s->buffer = NULL;
}

/* cpprcoder.h:971-978: non-positive request -> 16, else round up to 16. */
void rco_stream_init_cap(rco_stream* s, int32_t capacity)
{
    s->capacity = (capacity <= 0) ? 16 : round16(capacity);
    s->size = 0;
    s->buffer = (uint8_t*)malloc((size_t)s->capacity);
}

/* cpprcoder.h:980-983 */
void rco_stream_free(rco_stream* s)
{
    free(s->buffer);
    s->buffer = NULL;
}

/* cpprcoder.h:985-994: a reserve that does not shrink DISCARDS the contents
 * (free + malloc, no copy); only a strictly smaller request is a no-op. */
void rco_stream_reserve(rco_stream* s, int32_t capacity)
{
    capacity = round16(capacity);
    if (capacity < s->capacity) return;
    free(s->buffer);
    s->capacity = capacity;
    s->buffer = (uint8_t*)malloc((size_t)capacity);
}

/* cpprcoder.h:996-1003: only moves the cursor. */
void rco_stream_resize(rco_stream* s, int32_t size)
{
    if (s->capacity < size) rco_stream_reserve(s, size);
    s->size = size;
}

/* cpprcoder.h:1056-1077: 0 -> 1024, doubling below 16 KiB, then +16 KiB steps. */
static int stream_grow(rco_stream* s, int32_t need)
{
    int32_t cap = s->capacity;
    do {
        if (cap <= 0) cap = 1024;
        else if (cap < 4096 * 4) cap <<= 1;
        else cap += 4096 * 4;
    } while (cap < need);
    cap = round16(cap);
    uint8_t* fresh = (uint8_t*)malloc((size_t)cap);
    if (!fresh) return 0;
    if (s->capacity > 0) memcpy(fresh, s->buffer, (size_t)s->capacity);
    free(s->buffer);
    s->buffer = fresh;
    s->capacity = cap;
    return 1;
}

/* cpprcoder.h:1031-1045 */
int32_t rco_stream_write(rco_stream* s, int32_t size, const uint8_t* bytes)
{
    int32_t end = s->size + size;
    if (s->capacity < end && !stream_grow(s, end)) return -1;
    memcpy(s->buffer + s->size, bytes, (size_t)size);
    s->size = end;
    return size;
}

/* cpprcoder.h:1047-1054: never grows. */
int rco_stream_write_byte(rco_stream* s, uint8_t byte)
{
    if (s->capacity <= s->size) return 0;
    s->buffer[s->size++] = byte;
    return 1;
}

/* ================================================================== */
/* Model (AdaptiveFrequencyTable)                                      */
/* ================================================================== */

/* cpprcoder.h:1094-1132: all counts 1, chunk sums 16,32,...,256, total 256. */
void rco_model_init(rco_model* m)
{
    m->total = 256;
    for (int i = 0; i < 256; ++i) m->freq[i] = 1;
    for (int k = 0; k < 16; ++k) m->chunk_incl[k] = 16u * (uint32_t)(k + 1);
}

/* cpprcoder.h:1245-1261 */
static void model_rebuild_chunks(rco_model* m)
{
    uint32_t run = 0;
    for (int k = 0; k < 16; ++k) {
        for (int j = 0; j < 16; ++j) run += m->freq[16 * k + j];
        m->chunk_incl[k] = run;
    }
}

/* cpprcoder.h:1134-1177: +1; the halving test looks at the already
 * incremented total; halving keeps every count odd-or-one ((f>>1)|1). */
void rco_model_update(rco_model* m, uint8_t sym)
{
    m->freq[sym] += 1;
    m->total += 1;
    if (RANGE_FLOOR <= m->total) {
        uint32_t sum = 0;
        for (int i = 0; i < 256; ++i) {
            m->freq[i] = (m->freq[i] >> 1) | 1u;
            sum += m->freq[i];
        }
        m->total = sum;
        model_rebuild_chunks(m);
    } else {
        for (int k = sym >> 4; k < 16; ++k) m->chunk_incl[k] += 1;
    }
}

/* cpprcoder.h:1179-1187 */
uint32_t rco_model_cumulative(const rco_model* m, uint8_t sym)
{
    uint32_t k = sym >> 4;
    uint32_t acc = k ? m->chunk_incl[k - 1] : 0;
    for (uint32_t i = k << 4; i < sym; ++i) acc += m->freq[i];
    return acc;
}

/* cpprcoder.h:1220-1242 (the active, scalar branch).  A target at or past
 * total matches no chunk, the linear walk then runs off the table and the
 * reference is left with code=0, count=total; reproduced here. */
void rco_model_find(const rco_model* m, uint32_t target, uint32_t* count, uint8_t* code)
{
    uint32_t k = 0, acc = 0;
    for (uint32_t i = 0; i < 16; ++i) {
        if (target < m->chunk_incl[i]) {
            k = i;
            acc = k ? m->chunk_incl[k - 1] : 0;
            break;
        }
    }
    uint32_t first = k << 4;
    for (uint32_t i = first; i < 256; ++i) {
        uint32_t next = acc + m->freq[i];
        if (target < next) {
            *count = acc;
            *code = (uint8_t)i;
            return;
        }
        acc = next;
    }
    *count = acc;
    *code = (uint8_t)first;
}

/* ================================================================== */
/* Adaptive encoder                                                    */
/* ================================================================== */

/* cpprcoder.h:678-695 */
int rco_encoder_begin(rco_encoder* e, rco_stream* sink, uint32_t declared)
{
    rco_model_init(&e->model);
    e->sink = sink;
    e->range = ENC_RANGE0;
    e->declared = declared;
    e->consumed = 0;
    e->low = 0;
    e->held = 0;
    e->pending = 0;
    uint8_t hdr[4] = {(uint8_t)declared, (uint8_t)(declared >> 8), (uint8_t)(declared >> 16), (uint8_t)(declared >> 24)};
    return 0 < rco_stream_write(sink, 4, hdr);
}

/* cpprcoder.h:764-802 */
static int encoder_settle(rco_encoder* e, uint32_t low_before)
{
    if (e->low < low_before) { /* carry out of the 32-bit window */
        e->held = (uint8_t)(e->held + 1);
        if (e->pending > 0) {
            if (!rco_stream_write_byte(e->sink, e->held)) return 0;
            for (uint32_t i = 1; i < e->pending; ++i)
                if (!rco_stream_write_byte(e->sink, 0)) return 0;
            e->held = 0;
            e->pending = 0;
        }
    }
    while (e->range < RANGE_FLOOR) {
        if (e->low < TOP_GUARD) {
            if (!rco_stream_write_byte(e->sink, e->held)) return 0;
            for (uint32_t i = 0; i < e->pending; ++i)
                if (!rco_stream_write_byte(e->sink, 0xFF)) return 0;
            e->held = (uint8_t)(e->low >> 24);
            e->pending = 0;
        } else {
            e->pending += 1;
        }
        e->low <<= 8;
        e->range <<= 8;
    }
    return 1;
}

/* cpprcoder.h:744-762: result is ignored by the callers (:716, :738). */
static int encoder_flush(rco_encoder* e)
{
    if (!rco_stream_write_byte(e->sink, e->held)) return 0;
    for (uint32_t i = 0; i < e->pending; ++i)
        if (!rco_stream_write_byte(e->sink, 0xFF)) return 0;
    uint8_t tail[4] = {(uint8_t)(e->low >> 24), (uint8_t)(e->low >> 16), (uint8_t)(e->low >> 8), (uint8_t)e->low};
    return 0 < rco_stream_write(e->sink, 4, tail);
}

static int encoder_step(rco_encoder* e, uint8_t sym)
{
    uint32_t t = e->range / e->model.total;
    uint32_t before = e->low;
    e->low += rco_model_cumulative(&e->model, sym) * t;
    e->range = e->model.freq[sym] * t;
    return encoder_settle(e, before);
}

/* cpprcoder.h:697-720.  A sink that fills mid-symbol leaves low/range
 * advanced and the model not updated, exactly as the reference does. */
rco_result rco_encoder_put(rco_encoder* e, int32_t size, const uint8_t* bytes)
{
    rco_result r;
    for (int32_t i = 0; i < size; ++i) {
        if (!encoder_step(e, bytes[i])) {
            e->consumed += (uint32_t)i;
            r.status = RCO_PENDING;
            r.request_size = e->declared - e->consumed;
            return r;
        }
        rco_model_update(&e->model, bytes[i]);
    }
    e->consumed += (uint32_t)size;
    if (e->declared <= e->consumed) {
        encoder_flush(e);
        r.status = RCO_SUCCESS;
        r.request_size = 0;
        return r;
    }
    r.status = RCO_PENDING;
    r.request_size = e->declared - e->consumed;
    return r;
}

/* cpprcoder.h:722-742 (on a full sink it counts the byte as consumed). */
rco_result rco_encoder_put1(rco_encoder* e, uint8_t byte)
{
    rco_result r;
    if (!encoder_step(e, byte)) {
        e->consumed += 1;
        r.status = RCO_PENDING;
        r.request_size = e->declared - e->consumed;
        return r;
    }
    rco_model_update(&e->model, byte);
    e->consumed += 1;
    if (e->declared <= e->consumed) {
        encoder_flush(e);
        r.status = RCO_SUCCESS;
        r.request_size = 0;
        return r;
    }
    r.status = RCO_PENDING;
    r.request_size = e->declared - e->consumed;
    return r;
}

/* ================================================================== */
/* Adaptive decoder                                                    */
/* ================================================================== */

/* cpprcoder.h:859-870: range starts BELOW the floor so that the first
 * settle shifts out the encoder's lead-in byte. */
int rco_decoder_begin(rco_decoder* d, rco_stream* sink)
{
    rco_model_init(&d->model);
    d->sink = sink;
    d->declared = 0;
    d->produced = 0;
    d->state = 0;
    d->range = DEC_RANGE0;
    d->low = 0;
    d->last = 0;
    return 1;
}

/* cpprcoder.h:872-940 */
rco_result rco_decoder_feed(rco_decoder* d, int32_t size, const uint8_t* bytes)
{
    rco_result r;
    if (d->state == 0) {
        if (size < 8) {
            r.status = RCO_PENDING;
            r.request_size = 8;
            return r;
        }
        d->declared = (uint32_t)bytes[0] | ((uint32_t)bytes[1] << 8) | ((uint32_t)bytes[2] << 16) | ((uint32_t)bytes[3] << 24);
        d->low = ((uint32_t)bytes[4] << 24) | ((uint32_t)bytes[5] << 16) | ((uint32_t)bytes[6] << 8) | (uint32_t)bytes[7];
        bytes += 8;
        size -= 8;
        d->state = 1;
    } else if (d->state != 1) {
        r.status = RCO_ERROR;
        r.request_size = 0;
        return r;
    }
    for (;;) {
        while (d->range < RANGE_FLOOR) { /* :926-940 */
            if (size <= 0) {
                r.status = RCO_PENDING;
                r.request_size = d->declared - d->produced;
                return r;
            }
            d->range <<= 8;
            d->low = (d->low << 8) + *bytes++;
            --size;
        }
        uint32_t t = d->range / d->model.total;
        uint32_t below;
        rco_model_find(&d->model, d->low / t, &below, &d->last);
        d->low -= t * below;
        d->range = t * d->model.freq[d->last];
        if (!rco_stream_write_byte(d->sink, d->last)) {
            r.status = RCO_PENDING;
            r.request_size = d->declared - d->produced;
            return r;
        }
        d->produced += 1;
        if (d->declared <= d->produced) { /* n = 0 still emitted one byte */
            r.status = RCO_SUCCESS;
            r.request_size = 0;
            return r;
        }
        rco_model_update(&d->model, d->last);
    }
}

/* ================================================================== */
/* One-shot helpers                                                    */
/* ================================================================== */

rco_result rco_adaptive_encode(const uint8_t* src, uint32_t n, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size)
{
    rco_stream s;
    rco_encoder e;
    rco_result r;
    rco_stream_init_cap(&s, (int32_t)dst_cap);
    if (!rco_encoder_begin(&e, &s, n)) {
        r.status = RCO_ERROR;
        r.request_size = 0;
    } else {
        r = rco_encoder_put(&e, (int32_t)n, src);
    }
    if (out_size) *out_size = (uint64_t)s.size;
    if (dst) memcpy(dst, s.buffer, (size_t)((uint64_t)s.size < dst_cap ? (uint64_t)s.size : dst_cap));
    rco_stream_free(&s);
    return r;
}

rco_result rco_adaptive_decode(const uint8_t* comp, uint64_t comp_size, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size)
{
    rco_stream s;
    rco_decoder d;
    rco_stream_init_cap(&s, (int32_t)dst_cap);
    rco_decoder_begin(&d, &s);
    rco_result r = rco_decoder_feed(&d, (int32_t)comp_size, comp);
    if (out_size) *out_size = (uint64_t)s.size;
    if (dst) memcpy(dst, s.buffer, (size_t)((uint64_t)s.size < dst_cap ? (uint64_t)s.size : dst_cap));
    rco_stream_free(&s);
    return r;
}

/* ================================================================== */
/* Static (two-pass) coder: RangeEncoder<T>                            */
/* ================================================================== */

typedef struct {
    uint32_t cum[257];
} static_table;

/* cpprcoder.h:543-571: histogram with the order-dependent 16-bit squeeze,
 * then (only for n > 2^24) the shift that skips symbol 0. */
static void static_count(static_table* t, uint32_t n, const uint8_t* bytes)
{
    memset(t->cum, 0, sizeof(t->cum));
    for (uint32_t i = 0; i < n; ++i) {
        uint8_t b = bytes[i];
        if (0xFFFFu <= t->cum[b]) {
            for (int j = 0; j < 256; ++j)
                if (t->cum[j] > 0) t->cum[j] = (t->cum[j] >> 1) | 1u;
        }
        t->cum[b] += 1;
    }
    if (RANGE_FLOOR < n) {
        uint32_t sh = 0;
        while (RANGE_FLOOR < n) {
            n >>= 1;
            ++sh;
        }
        for (int i = 1; i < 256; ++i) t->cum[i] = t->cum[i] ? ((t->cum[i] >> sh) | 1u) : 0;
    }
}

/* cpprcoder.h:573-583: counts -> exclusive running sums, cum[256] = total. */
static void static_accumulate(static_table* t)
{
    uint32_t run = 0;
    for (int i = 0; i < 256; ++i) {
        uint32_t c = t->cum[i];
        t->cum[i] = run;
        run += c;
    }
    t->cum[256] = run;
}

/* cpprcoder.h:521-535 */
static uint8_t static_find(const static_table* t, uint32_t target)
{
    uint32_t lo = 0, hi = 255;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (t->cum[mid + 1] <= target) lo = mid + 1;
        else hi = mid;
    }
    return (uint8_t)lo;
}

/* cpprcoder.h:375-458 */
static int static_encode_stream(rco_stream* s, uint32_t n, const uint8_t* bytes)
{
    static_table t;
    static_count(&t, n, bytes);
    uint32_t range = 0xFFFFFFFFu, low = 0, pending = 0, held = 0;
    uint8_t hdr[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    if (rco_stream_write(s, 4, hdr) <= 0) return 0;
    /* :604-619 the 256 counts as native-endian (little-endian here) u16, 4 x 128 B */
    for (int i = 0; i < 256; i += 64) {
        uint8_t row[128];
        for (int j = 0; j < 64; ++j) {
            uint32_t c = t.cum[i + j] & 0xFFFFu;
            row[2 * j] = (uint8_t)c;
            row[2 * j + 1] = (uint8_t)(c >> 8);
        }
        if (rco_stream_write(s, 128, row) <= 0) return 0;
    }
    static_accumulate(&t);
    for (uint32_t i = 0; i < n; ++i) {
        uint8_t b = bytes[i];
        uint32_t step = range / t.cum[256];
        uint32_t moved = low + t.cum[b] * step;
        range = (t.cum[b + 1] - t.cum[b]) * step;
        if (moved < low) {
            ++held;
            for (; pending != 0; --pending) {
                if (!rco_stream_write_byte(s, (uint8_t)held)) return 0;
                held = 0;
            }
        }
        low = moved;
        while (range < RANGE_FLOOR) {
            if (low < TOP_GUARD) {
                if (!rco_stream_write_byte(s, (uint8_t)held)) return 0;
                for (; pending != 0; --pending)
                    if (!rco_stream_write_byte(s, 0xFF)) return 0;
                held = low >> 24;
            } else {
                ++pending;
            }
            low <<= 8;
            range <<= 8;
        }
    }
    uint8_t fill = 0xFF; /* :439-451 */
    if (0xFFFFFFFFu <= low) {
        ++held;
        fill = 0;
    }
    if (!rco_stream_write_byte(s, (uint8_t)held)) return 0;
    for (; pending != 0; --pending)
        if (!rco_stream_write_byte(s, fill)) return 0;
    uint8_t tail[4] = {(uint8_t)(low >> 24), (uint8_t)(low >> 16), (uint8_t)(low >> 8), (uint8_t)low};
    return 0 < rco_stream_write(s, 4, tail);
}

/* cpprcoder.h:460-519.  Deviation, for safety only: a corrupt table whose
 * total is 0, or a step of 0, divides by zero in the reference; here it
 * returns false. */
static int static_decode_stream(rco_stream* s, uint32_t size, const uint8_t* bytes)
{
    static_table t;
    uint32_t range = 0xFFFFFFFFu, low;
    if (size < 1) return 0;
    const uint8_t* end = bytes + size;
    if (size < 516) return 0;
    uint32_t n = (uint32_t)bytes[0] | ((uint32_t)bytes[1] << 8) | ((uint32_t)bytes[2] << 16) | ((uint32_t)bytes[3] << 24);
    if (n == 0) return 1;
    bytes += 4;
    for (int i = 0; i < 256; ++i) t.cum[i] = (uint32_t)bytes[2 * i] | ((uint32_t)bytes[2 * i + 1] << 8); /* :585-602 */
    t.cum[256] = 0;
    bytes += 512;
    if (!(bytes < end)) return 0;
    static_accumulate(&t);
    if ((uint32_t)(end - bytes) < 5) return 0;
    low = ((uint32_t)bytes[1] << 24) | ((uint32_t)bytes[2] << 16) | ((uint32_t)bytes[3] << 8) | (uint32_t)bytes[4];
    bytes += 5;
    if (t.cum[256] == 0) return 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t step = range / t.cum[256];
        if (step == 0) return 0;
        uint8_t c = static_find(&t, low / step);
        low -= t.cum[c] * step;
        range = (t.cum[c + 1] - t.cum[c]) * step;
        while (range < RANGE_FLOOR) {
            if (end <= bytes) return 0;
            range <<= 8;
            low = (low << 8) | *bytes++;
        }
        if (!rco_stream_write_byte(s, c)) return 0;
    }
    return 1;
}

int rco_static_encode(const uint8_t* src, uint32_t n, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size)
{
    rco_stream s;
    rco_stream_init_cap(&s, (int32_t)dst_cap);
    int ok = static_encode_stream(&s, n, src);
    if (out_size) *out_size = (uint64_t)s.size;
    if (dst) memcpy(dst, s.buffer, (size_t)((uint64_t)s.size < dst_cap ? (uint64_t)s.size : dst_cap));
    rco_stream_free(&s);
    return ok;
}

int rco_static_decode(const uint8_t* comp, uint32_t comp_size, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size)
{
    rco_stream s;
    rco_stream_init_cap(&s, (int32_t)dst_cap);
    int ok = static_decode_stream(&s, comp_size, comp);
    if (out_size) *out_size = (uint64_t)s.size;
    if (dst) memcpy(dst, s.buffer, (size_t)((uint64_t)s.size < dst_cap ? (uint64_t)s.size : dst_cap));
    rco_stream_free(&s);
    return ok;
}

/* ================================================================== */
/* Many independent blocks                                             */
/* ================================================================== */

uint64_t rco_block_count(uint64_t n, uint32_t block) { return block ? (n + block - 1) / block : 0; }

/* Room for any adaptive or static stream of one block, rounded to 16 so that
 * it is a valid MemoryStream capacity as given. */
uint64_t rco_block_bound(uint32_t block)
{
    uint64_t b = (uint64_t)block + block / 32 + 1024;
    return (b + 15) & ~(uint64_t)15;
}

int rco_encode_block_range(const uint8_t* src, uint64_t n, uint32_t block, uint64_t first, uint64_t last,
                           uint8_t* slots, uint64_t slot, uint32_t* sizes, int coder)
{
    rco_stream s;
    rco_stream_init_cap(&s, (int32_t)slot);
    int ok = 1;
    for (uint64_t b = first; b < last; ++b) {
        uint64_t at = b * block;
        uint32_t len = (uint32_t)((n - at < block) ? (n - at) : block);
        s.size = 0; /* fresh stream per block: test/main.cpp:321 */
        if (coder == 0) {
            rco_encoder e; /* fresh coder per block: test/main.cpp:325-330 */
            if (!rco_encoder_begin(&e, &s, len)) ok = 0;
            else if (rco_encoder_put(&e, (int32_t)len, src + at).status != RCO_SUCCESS) ok = 0;
        } else {
            if (!static_encode_stream(&s, len, src + at)) ok = 0;
        }
        if ((uint64_t)s.size > slot) ok = 0;
        else memcpy(slots + b * slot, s.buffer, (size_t)s.size);
        sizes[b] = (uint32_t)s.size;
    }
    rco_stream_free(&s);
    return ok;
}

int rco_decode_block_range(const uint8_t* slots, uint64_t slot, const uint32_t* sizes, uint32_t block, uint64_t n,
                           uint64_t first, uint64_t last, uint8_t* dst, int coder)
{
    rco_stream s;
    rco_stream_init_cap(&s, (int32_t)block);
    int ok = 1;
    for (uint64_t b = first; b < last; ++b) {
        uint64_t at = b * block;
        uint32_t len = (uint32_t)((n - at < block) ? (n - at) : block);
        s.size = 0;
        if (coder == 0) {
            rco_decoder d; /* test/main.cpp:339-344 */
            rco_decoder_begin(&d, &s);
            if (rco_decoder_feed(&d, (int32_t)sizes[b], slots + b * slot).status != RCO_SUCCESS) ok = 0;
        } else {
            if (!static_decode_stream(&s, sizes[b], slots + b * slot)) ok = 0;
        }
        uint32_t got = (uint32_t)s.size < len ? (uint32_t)s.size : len;
        memcpy(dst + at, s.buffer, got);
        if ((uint32_t)s.size != len && !(len == 0)) ok = 0;
    }
    rco_stream_free(&s);
    return ok;
}

uint64_t rco_fnv1a64(const uint8_t* p, uint64_t n)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint64_t i = 0; i < n; ++i) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

/* ================================================================== */
/* Probes (same shape as the ref_shim.cpp exports)                     */
/* ================================================================== */

rco_result rco_adaptive_encode_chunked(const uint8_t* src, uint32_t n, uint32_t piece, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size)
{
    rco_stream s;
    rco_encoder e;
    rco_result r = {RCO_ERROR, 0};
    rco_stream_init_cap(&s, (int32_t)dst_cap);
    if (rco_encoder_begin(&e, &s, n)) {
        r.status = RCO_SUCCESS;
        if (n == 0) r = rco_encoder_put(&e, 0, src);
        for (uint32_t at = 0; at < n;) {
            if (piece == 0) {
                r = rco_encoder_put1(&e, src[at]);
                at += 1;
            } else {
                uint32_t len = (n - at < piece) ? (n - at) : piece;
                r = rco_encoder_put(&e, (int32_t)len, src + at);
                at += len;
            }
            if (r.status == RCO_ERROR) break;
        }
    }
    if (out_size) *out_size = (uint64_t)s.size;
    if (dst) memcpy(dst, s.buffer, (size_t)((uint64_t)s.size < dst_cap ? (uint64_t)s.size : dst_cap));
    rco_stream_free(&s);
    return r;
}

/* The same, and how many bytes the sink held after initialize() (sizes[0]) and after every encode() call
 * (sizes[1..]): what a caller that looks at its sink between calls sees (cpprcoder.h:764-802: everything the coder has
 * made except its held byte and the 0xFF bytes pending behind it).  -> the number of entries written. */
uint32_t rco_adaptive_encode_trace(const uint8_t* src, uint32_t n, uint32_t piece, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size,
                                   uint32_t* sizes, uint32_t max_sizes, rco_result* last)
{
    rco_stream s;
    rco_encoder e;
    rco_result r = {RCO_ERROR, 0};
    uint32_t calls = 0;
    rco_stream_init_cap(&s, (int32_t)dst_cap);
    if (rco_encoder_begin(&e, &s, n)) {
        r.status = RCO_SUCCESS;
        if (calls < max_sizes) sizes[calls++] = (uint32_t)s.size;
        if (n == 0) {
            r = rco_encoder_put(&e, 0, src);
            if (calls < max_sizes) sizes[calls++] = (uint32_t)s.size;
        }
        for (uint32_t at = 0; at < n;) {
            uint32_t len = (n - at < piece) ? (n - at) : piece;
            r = rco_encoder_put(&e, (int32_t)len, src + at);
            at += len;
            if (calls < max_sizes) sizes[calls++] = (uint32_t)s.size;
            if (r.status == RCO_ERROR || (r.status == RCO_PENDING && at < n && r.request_size != n - at)) break; /* the sink filled */
        }
    }
    if (last) *last = r;
    if (out_size) *out_size = (uint64_t)s.size;
    if (dst) memcpy(dst, s.buffer, (size_t)((uint64_t)s.size < dst_cap ? (uint64_t)s.size : dst_cap));
    rco_stream_free(&s);
    return calls;
}

rco_result rco_adaptive_decode_chunked(const uint8_t* comp, uint64_t comp_size, uint32_t piece, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size)
{
    rco_stream s;
    rco_decoder d;
    rco_result r = {RCO_PENDING, 0};
    rco_stream_init_cap(&s, (int32_t)dst_cap);
    rco_decoder_begin(&d, &s);
    uint64_t at = 0;
    int first = 1;
    while (at < comp_size) {
        uint64_t len = comp_size - at < piece ? comp_size - at : piece;
        if (first && len < 8) len = comp_size - at < 8 ? comp_size - at : 8;
        first = 0;
        r = rco_decoder_feed(&d, (int32_t)len, comp + at);
        at += len;
        if (r.status != RCO_PENDING) break;
    }
    if (out_size) *out_size = (uint64_t)s.size;
    if (dst) memcpy(dst, s.buffer, (size_t)((uint64_t)s.size < dst_cap ? (uint64_t)s.size : dst_cap));
    rco_stream_free(&s);
    return r;
}

void rco_model_probe(const uint8_t* syms, uint64_t n, uint32_t* total, uint32_t* freq256, uint32_t* cum256,
                     const uint32_t* targets, uint32_t ntargets, uint32_t* found_count, uint8_t* found_code)
{
    rco_model m;
    rco_model_init(&m);
    for (uint64_t i = 0; i < n; ++i) rco_model_update(&m, syms[i]);
    *total = m.total;
    for (uint32_t c = 0; c < 256; ++c) {
        freq256[c] = m.freq[c];
        cum256[c] = rco_model_cumulative(&m, (uint8_t)c);
    }
    for (uint32_t k = 0; k < ntargets; ++k) rco_model_find(&m, targets[k], &found_count[k], &found_code[k]);
}

/* ops: 0 default init, 1 init(arg), 2 write(arg pattern bytes), 3 write_byte(arg), 4 reserve(arg), 5 resize(arg) */
int rco_stream_script(const int32_t* ops, int nops, int32_t* out)
{
    rco_stream s;
    int live = 0, w = 0;
    static uint8_t pattern[1 << 16];
    for (int i = 0; i < (1 << 16); ++i) pattern[i] = (uint8_t)(i * 7 + 1);
    for (int i = 0; i < nops; ++i) {
        int32_t op = ops[2 * i], arg = ops[2 * i + 1], ret = 0;
        switch (op) {
        case 0:
            if (live) rco_stream_free(&s);
            rco_stream_init(&s);
            live = 1;
            break;
        case 1:
            if (live) rco_stream_free(&s);
            rco_stream_init_cap(&s, arg);
            live = 1;
            break;
        case 2: ret = rco_stream_write(&s, arg, pattern); break;
        case 3: ret = rco_stream_write_byte(&s, (uint8_t)arg); break;
        case 4: rco_stream_reserve(&s, arg); break;
        case 5: rco_stream_resize(&s, arg); break;
        default: return -1;
        }
        out[w++] = ret;
        out[w++] = s.capacity;
        out[w++] = s.size;
    }
    if (live) rco_stream_free(&s);
    return w;
}
