/*
 * rc_oracle.h -- CPU restatement of the reference's range-coder hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / the timed CPU baseline.
 * The product path (cpprcoder_amd/librcx.so) never links or calls it.
 *
 * Parity status: PINNED.  This restatement is checked byte-for-byte against
 *   (1) the reference itself compiled from /root/reference/cpprcoder.h
 *       (oracle/_ref/libcpprcoder_ref.so, built by oracle/Makefile), and
 *   (2) the committed golden vectors in tests/golden/ that were generated
 *       from that reference build (tests/golden/make_golden.py), which also
 *       reproduce all 22 compressed sizes published in the reference's
 *       README.md:16-46.
 *
 * Every function cites the reference lines (cpprcoder.h unless noted) whose
 * behaviour it restates.  All arithmetic is u32 with wraparound.
 */
#ifndef RC_ORACLE_H_
#define RC_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status values: cpprcoder.h:112-117 */
enum { RCO_SUCCESS = 0, RCO_PENDING = 1, RCO_ERROR = -1 };

/* Result: cpprcoder.h:119-123 */
typedef struct { int32_t status; uint32_t request_size; } rco_result;

/* ------------------------------------------------------------------ */
/* Sink: restates MemoryStream (cpprcoder.h:185-247, 964-1077).        */
/* write() grows, write_byte() never grows.                            */
/* ------------------------------------------------------------------ */
typedef struct {
    int32_t capacity;
    int32_t size;
    uint8_t* buffer;
} rco_stream;

void rco_stream_init(rco_stream* s);                       /* :964-969  */
void rco_stream_init_cap(rco_stream* s, int32_t capacity); /* :971-978  */
void rco_stream_free(rco_stream* s);                       /* :980-983  */
void rco_stream_reserve(rco_stream* s, int32_t capacity);  /* :985-994  */
void rco_stream_resize(rco_stream* s, int32_t size);       /* :996-1003 */
int32_t rco_stream_write(rco_stream* s, int32_t size, const uint8_t* bytes); /* :1031-1045 */
int rco_stream_write_byte(rco_stream* s, uint8_t byte);    /* :1047-1054 */

/* ------------------------------------------------------------------ */
/* Model: restates AdaptiveFrequencyTable (:256-314, 1085-1261).       */
/* ------------------------------------------------------------------ */
typedef struct {
    uint32_t total;
    uint32_t chunk_incl[16]; /* inclusive running sums of 16-symbol chunks */
    uint32_t freq[256];
} rco_model;

void rco_model_init(rco_model* m);                          /* :1094-1132 */
void rco_model_update(rco_model* m, uint8_t sym);           /* :1134-1177 */
uint32_t rco_model_cumulative(const rco_model* m, uint8_t sym); /* :1179-1187 */
void rco_model_find(const rco_model* m, uint32_t target, uint32_t* count, uint8_t* code); /* :1220-1242 */

/* ------------------------------------------------------------------ */
/* Streaming adaptive coder objects (resumable, like the reference).   */
/* ------------------------------------------------------------------ */
typedef struct {
    rco_stream* sink;
    rco_model model;
    uint32_t declared, consumed;
    uint32_t range, low;
    uint8_t held;      /* reference: buffer_ */
    uint32_t pending;  /* reference: carry_  */
} rco_encoder;

typedef struct {
    rco_stream* sink;
    rco_model model;
    uint32_t declared, produced;
    int32_t state;
    uint32_t range, low;
    uint8_t last;
} rco_decoder;

int rco_encoder_begin(rco_encoder* e, rco_stream* sink, uint32_t declared);       /* :678-695 */
rco_result rco_encoder_put(rco_encoder* e, int32_t size, const uint8_t* bytes);   /* :697-720 */
rco_result rco_encoder_put1(rco_encoder* e, uint8_t byte);                        /* :722-742 */
int rco_decoder_begin(rco_decoder* d, rco_stream* sink);                          /* :859-870 */
rco_result rco_decoder_feed(rco_decoder* d, int32_t size, const uint8_t* bytes);  /* :872-924 */

/* ------------------------------------------------------------------ */
/* One-shot helpers over caller memory (what the parity tests call).   */
/* dst_cap plays the role of the MemoryStream capacity for write_byte. */
/* Returns the Result; *out_size = bytes in dst.                       */
/* ------------------------------------------------------------------ */
rco_result rco_adaptive_encode(const uint8_t* src, uint32_t n, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size);
rco_result rco_adaptive_decode(const uint8_t* comp, uint64_t comp_size, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size);

/* Static (two-pass) coder, RangeEncoder<T>::encode/decode (:375-519). 1 = true. */
int rco_static_encode(const uint8_t* src, uint32_t n, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size);
int rco_static_decode(const uint8_t* comp, uint32_t comp_size, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size);

/* ------------------------------------------------------------------ */
/* Many independent blocks: block b = src[b*block, min(n,(b+1)*block)),*/
/* each coded by a fresh coder exactly as test/main.cpp:325-330 does   */
/* for a whole file.  offsets has nblocks+1 entries.                   */
/* [first,last) lets several threads split the block range; streams    */
/* land in dst at b*slot (slot >= rco_block_bound(block)).             */
/* ------------------------------------------------------------------ */
uint64_t rco_block_count(uint64_t n, uint32_t block);
uint64_t rco_block_bound(uint32_t block);
int rco_encode_block_range(const uint8_t* src, uint64_t n, uint32_t block, uint64_t first, uint64_t last,
                           uint8_t* slots, uint64_t slot, uint32_t* sizes, int coder /*0 adaptive, 1 static*/);
int rco_decode_block_range(const uint8_t* slots, uint64_t slot, const uint32_t* sizes, uint32_t block, uint64_t n,
                           uint64_t first, uint64_t last, uint8_t* dst, int coder);

uint64_t rco_fnv1a64(const uint8_t* p, uint64_t n);

/* ------------------------------------------------------------------ */
/* Probes with the same shape as the ones ref_shim.cpp exports, so the  */
/* tests can run one script against both.                              */
/* ------------------------------------------------------------------ */
rco_result rco_adaptive_encode_chunked(const uint8_t* src, uint32_t n, uint32_t piece, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size);
uint32_t rco_adaptive_encode_trace(const uint8_t* src, uint32_t n, uint32_t piece, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size,
                                   uint32_t* sizes, uint32_t max_sizes, rco_result* last);
rco_result rco_adaptive_decode_chunked(const uint8_t* comp, uint64_t comp_size, uint32_t piece, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size);
void rco_model_probe(const uint8_t* syms, uint64_t n, uint32_t* total, uint32_t* freq256, uint32_t* cum256,
                     const uint32_t* targets, uint32_t ntargets, uint32_t* found_count, uint8_t* found_code);
int rco_stream_script(const int32_t* ops, int nops, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* RC_ORACLE_H_ */
