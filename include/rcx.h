/*
 * rcx.h -- C ABI of the MI355X-native many-block range coder.
 *
 * The reference (taqu/cpprcoder) is a header-only C++ template library with no
 * FFI of its own; its boundary is the class API in cpprcoder.h:
 *     bool   AdaptiveRangeEncoder<T>::initialize(T&, u32)      cpprcoder.h:636, 678-695
 *     Result AdaptiveRangeEncoder<T>::encode(s32, const u8*)   cpprcoder.h:637, 697-720
 *     bool   AdaptiveRangeDecoder<T>::initialize(T&)           cpprcoder.h:819, 859-870
 *     Result AdaptiveRangeDecoder<T>::decode(s32, const u8*)   cpprcoder.h:820, 872-924
 *     bool   RangeEncoder<T>::encode / decode                  cpprcoder.h:336-337, 375-519
 *     enum Status / struct Result                              cpprcoder.h:112-123
 * and it is driven per whole buffer by test/main.cpp:321-344.  This header is
 * the plain-C surface a binding (cgo, JNI, ctypes, N-API ...) or the C++ facade
 * include/cpprcoder_amd/cpprcoder.h sits on.  Every entry point names the
 * reference interface it replaces.
 *
 * Data model.  A buffer of n bytes is cut into blocks of `block` bytes (the last
 * one may be short).  Block b is coded by a fresh coder exactly as the reference
 * codes a whole file: the per-block stream is byte-identical to
 *     initialize(stream, len_b); encode(len_b, src + b*block);
 * i.e. [u32 LE len_b][0x00][payload...][u32 BE low].  The streams are stored
 * back to back ("compacted"); offsets[b] .. offsets[b+1] delimits block b and
 * offsets[nblocks] is the total size.
 *
 * All functions return an rcx status (0 = success, 1 = pending, negative = error)
 * and never throw, abort or print.  Pointers named d_* are device (HBM) pointers
 * of the context's GPU; all others are host pointers.  `stream` is a hipStream_t
 * passed as void* (NULL = the default stream).  The *_device calls only enqueue
 * work; block-level failures (slot overflow, corrupt input) are latched on the
 * device and read with rcx_ctx_sync_status().
 */
#ifndef RCX_H_
#define RCX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCX_VERSION 300 /* 0.3.0 */

/* Status codes.  0 / 1 / -1 are the reference's Status enum (cpprcoder.h:112-117). */
enum {
    RCX_OK = 0,          /* Status_Success */
    RCX_PENDING = 1,     /* Status_Pending: output sink full or input exhausted */
    RCX_ERROR = -1,      /* Status_Error */
    RCX_E_ARG = -2,      /* bad argument (null pointer, block size out of range ...) */
    RCX_E_CAPACITY = -3, /* destination too small / a block outgrew its scratch slot */
    RCX_E_CORRUPT = -4,  /* a block's stream is truncated or its header disagrees with the layout */
    RCX_E_HIP = -5,      /* HIP runtime error, or no usable MI355X */
    RCX_E_NOMEM = -6,    /* device or host allocation failed */
    RCX_E_COMM = -7      /* RCCL error */
};

/* Which coder a call uses.  With RCX_CODER_STATIC the single-stream entry points have the bool semantics of
 * RangeEncoder<T>::encode / decode (cpprcoder.h:336-337): RCX_OK = true, RCX_ERROR = false; encode hands back
 * the whole stream (the caller replays write / writeByte on its sink), decode the symbols written before a
 * failure; request_size is not used. */
enum {
    RCX_CODER_ADAPTIVE = 0, /* AdaptiveRangeEncoder/Decoder, cpprcoder.h:626-940 */
    RCX_CODER_STATIC = 1,   /* RangeEncoder (two-pass, 516-byte table header), cpprcoder.h:321-619 */
    /* The reference's rANS siblings (cppans.h).  A block's stream is what rANS::encode / encode_simd return for that
     * block alone (the last `size` bytes of their destination, test/main.cpp:384-387):
     * [u32 LE n][257 x u32 LE scaled cumulative counts][payload].  With these coders the single-stream entry points
     * have the semantics of the reference's static functions: encode hands back the stream, decode the n symbols;
     * RCX_ERROR where the reference returns 0. */
    RCX_CODER_RANS = 2,     /* rANS::encode / decode, cppans.h:497-563: one state, 14-bit probabilities, bytes */
    RCX_CODER_RANS8 = 3     /* rANS::encode_simd / decode_simd, cppans.h:567-649: 8 interleaved states, 12-bit, 16-bit words */
};

#define RCX_MIN_BLOCK 16u
/* Largest block of the many-block calls: the last size at which the table halving of cpprcoder.h:1138-1176 cannot
 * happen (the total starts at 256 and the halving comes when ++total reaches 2^24), so every block's i-th symbol
 * sees total = 256 + i.  Single streams (rcx_stream_*) may be longer, up to RCX_MAX_STREAM = the reference's
 * MAX_SIZE (cpprcoder.h:329): past RCX_MAX_BLOCK symbols the lane keeps its own total and halves the table. */
#define RCX_MAX_BLOCK ((1u << 24) - 256u)
#define RCX_MAX_STREAM 0x7FFFFFFFu
#define RCX_MAX_RANS_STREAM 0x7FFF0000u /* rANS single streams: 2n + 1096 must fit the u32 sizes of cppans.h:72-76 */

typedef struct rcx_ctx rcx_ctx;

int rcx_version(void);
const char* rcx_status_string(int status);

/* A context owns the device scratch (per-block slots, size table, divisor table)
 * of one GPU.  It is single-threaded; use one context per thread / stream. */
int rcx_ctx_create(int device, rcx_ctx** out);
void rcx_ctx_destroy(rcx_ctx* ctx);
/* Pre-allocate scratch for buffers up to n bytes at this block size (otherwise the
 * first call that needs more allocates, which is not allowed under graph capture). */
int rcx_ctx_reserve(rcx_ctx* ctx, uint64_t n, uint32_t block);
int rcx_ctx_reserve_for(rcx_ctx* ctx, int coder, uint64_t n, uint32_t block); /* the same for a given coder (rANS slots are larger) */
/* Wait for `stream`, then return and clear the latched device-side status:
 * RCX_OK, RCX_E_CAPACITY or RCX_E_CORRUPT; *first_bad_block (optional) gets the
 * lowest failing block index. */
int rcx_ctx_sync_status(rcx_ctx* ctx, void* stream, uint64_t* first_bad_block);

/* Geometry helpers (pure functions). */
uint64_t rcx_block_count(uint64_t n, uint32_t block);   /* ceil(n / block) */
uint64_t rcx_block_bound(uint32_t block);               /* bytes reserved for one block's stream (the range coders) */
uint64_t rcx_encode_bound(uint64_t n, uint32_t block);  /* safe dst_cap for the compacted streams (the range coders) */
/* The same per coder.  The rANS coders get the reference's own bound (rANS::calc_encoded_size, cppans.h:492-495:
 * 2n + 1032) + 64: encode_simd spends 2 bytes per symbol on a block of one repeated byte (the renormalisation test
 * of cppans.h:357 wraps to "always" for a frequency of 4096), and its eight flushed states need room even when n < 16. */
uint64_t rcx_block_bound_for(int coder, uint32_t block);
uint64_t rcx_encode_bound_for(int coder, uint64_t n, uint32_t block);

/*
 * Encode n bytes as independent blocks on the GPU.
 * Replaces, per block: AdaptiveRangeEncoder<T>::initialize + encode (cpprcoder.h:678-720),
 * RangeEncoder<T>::encode (cpprcoder.h:375-458) -- and the MemoryStream sink (cpprcoder.h:1031-1054) as the
 * output writer -- or rANS::encode / encode_simd (cppans.h:497-530, :567-607).
 *   d_src      n input bytes
 *   d_dst      compacted streams, capacity dst_cap (>= rcx_encode_bound_for(coder, n, block) is always enough)
 *   d_offsets  nblocks+1 u64, exclusive prefix of the per-block stream sizes (written)
 */
int rcx_encode_blocks_device(rcx_ctx* ctx, int coder, const void* d_src, uint64_t n, uint32_t block,
                             void* d_dst, uint64_t dst_cap, uint64_t* d_offsets, void* stream);

/*
 * Decode blocks produced by rcx_encode_blocks_device (or by the reference, block by block).
 * Replaces, per block: AdaptiveRangeDecoder<T>::initialize + decode (cpprcoder.h:859-924),
 * RangeEncoder<T>::decode (cpprcoder.h:460-519) or rANS::decode / decode_simd (cppans.h:532-564, :609-649).
 *   d_comp     the compacted streams, comp_size bytes (a block whose offsets point past comp_size is reported
 *              as RCX_E_CORRUPT and not read)
 *   d_offsets  nblocks+1 u64 as written by the encoder
 *   n          total decoded size; block b must declare min(block, n - b*block) bytes
 *   d_dst      n output bytes
 */
int rcx_decode_blocks_device(rcx_ctx* ctx, int coder, const void* d_comp, uint64_t comp_size,
                             const uint64_t* d_offsets, uint64_t nblocks, uint32_t block,
                             uint64_t n, void* d_dst, void* stream);

/* Host-buffer variants: copy in, run the device path, copy out, synchronise.
 * offsets may be NULL for encode when only the payload is wanted. */
int rcx_encode_blocks(rcx_ctx* ctx, int coder, const uint8_t* src, uint64_t n, uint32_t block,
                      uint8_t* dst, uint64_t dst_cap, uint64_t* dst_size, uint64_t* offsets);
int rcx_decode_blocks(rcx_ctx* ctx, int coder, const uint8_t* comp, uint64_t comp_size,
                      const uint64_t* offsets, uint64_t nblocks, uint32_t block,
                      uint8_t* dst, uint64_t dst_cap, uint64_t* dst_size);

/*
 * Single-stream calls with the reference's exact stream semantics, used by the
 * C++ facade: one stream of any size 0 .. RCX_MAX_STREAM, coded by one GPU lane.
 *   rcx_stream_encode == initialize(sink, n); encode(n, src)  into a sink that holds
 *       exactly sink_capacity bytes before writeByte fails (for a MemoryStream that is
 *       its capacity(), i.e. the constructor argument rounded up to 16, cpprcoder.h:975):
 *       returns RCX_OK with the whole stream in dst (*dst_size bytes: everything that fit through
 *       writeByte plus the final write(4), which grows the sink -- up to sink_capacity + 4), or
 *       RCX_PENDING with *request_size set as cpprcoder.h:708-711 does when the sink fills
 *       (dst then holds the max(sink_capacity, 4) bytes written so far).  dst holds dst_cap bytes;
 *       if the bytes to hand back do not fit, nothing is copied, *dst_size says how many there
 *       are and the call returns RCX_E_CAPACITY.  rcx_block_bound(n) is always enough.
 *   rcx_stream_decode == initialize(sink); decode(comp_size, comp): RCX_OK,
 *       RCX_PENDING (+ request_size) for short input or a full sink (one that accepts
 *       sink_capacity bytes), including the
 *       reference's quirk that a stream declaring 0 bytes yields one byte (cpprcoder.h:912).
 *       dst must have room for min(max(declared, 1), sink_capacity) bytes.
 */
int rcx_stream_encode(rcx_ctx* ctx, int coder, const uint8_t* src, uint32_t n,
                      uint8_t* dst, uint64_t dst_cap, uint64_t sink_capacity, uint64_t* dst_size, uint32_t* request_size);
int rcx_stream_decode(rcx_ctx* ctx, int coder, const uint8_t* comp, uint64_t comp_size,
                      uint8_t* dst, uint64_t sink_capacity, uint64_t* dst_size, uint32_t* request_size);

/*
 * The adaptive decoder as a resumable object: AdaptiveRangeDecoder<T>::initialize + decode(size, bytes) called piece by
 * piece (cpprcoder.h:859-924).  The decoder's state (low, range, the frequency table, how far it got) stays on the GPU
 * between calls, so every call costs what its new bytes allow, as with the reference's object.
 *   rcx_dstream_decode   feed `size` more bytes of the stream; the symbols that can be decoded now, at most dst_cap of
 *       them (the room the caller's sink has), go to dst and *produced_now says how many.  RCX_OK once max(declared, 1)
 *       symbols have been produced (cpprcoder.h:912), else RCX_PENDING with *request_size = declared - produced so
 *       far: the input ran dry (cpprcoder.h:901-903) or dst is full (call again with size 0 to go on -- unlike the
 *       reference, whose sink-full return has already swallowed a symbol, cpprcoder.h:909-911).  A first call with
 *       fewer than 8 bytes keeps nothing and asks for 8 (cpprcoder.h:877-880).
 *   Memory: the object keeps, on the device, the bytes it was fed and has not read yet; what it has read is dropped when its
 *   buffer grows, so it holds about twice the largest piece it was ever given, whatever the length of the stream.
 */
typedef struct rcx_dstream rcx_dstream;
int rcx_dstream_create(rcx_ctx* ctx, rcx_dstream** out);
void rcx_dstream_destroy(rcx_dstream* stream);
int rcx_dstream_decode(rcx_dstream* stream, const uint8_t* bytes, uint64_t size, uint8_t* dst, uint64_t dst_cap,
                       uint64_t* produced_now, uint32_t* request_size);

/*
 * The adaptive encoder as a resumable object: AdaptiveRangeEncoder<T>::initialize + encode(size, bytes) called piece by
 * piece (cpprcoder.h:678-720).  The reference writes to its sink as it goes: after every encode() call the sink holds the
 * header, and every payload byte except the one the coder still holds and the 0xFF bytes pending behind it
 * (cpprcoder.h:764-802); the last call adds those through writeByte() and the final low through write(4)
 * (cpprcoder.h:744-762).  This object gives a caller exactly those bytes call by call; its state stays on the GPU.
 *   rcx_estream_create   `declared` = the size passed to initialize() (the caller writes the 4 header bytes itself,
 *       cpprcoder.h:689-694)
 *   rcx_estream_encode   feed `size` more bytes (not more than are still expected: RCX_E_ARG, the reference asserts).
 *       dst receives what the reference passes to its sink during this call, *emitted_now bytes: payload bytes for
 *       writeByte() and, on the last call, *tail_bytes = 4 bytes behind them that go through write().  sink_room = how
 *       many bytes the caller's sink still takes through writeByte (UINT64_MAX: no limit).  Returns RCX_PENDING with
 *       *request_size = declared - fed so far while input is expected; RCX_OK when the stream is complete -- also when
 *       only finish() ran into the full sink (cpprcoder.h:716): dst then holds what fitted and *tail_bytes is 0.  If the
 *       sink fills while a symbol is coded: RCX_PENDING with *request_size as cpprcoder.h:708-711 sets it (declared -
 *       symbols coded before that one), dst holding the bytes that fitted; the object is then of no further use, as the
 *       reference's.  dst_cap >= 3 * size + 8 + the pending run is always enough; RCX_E_CAPACITY says how much is needed
 *       in *emitted_now (the call can be repeated after rcx_estream_rewind).
 *   rcx_estream_rewind   back to before the last rcx_estream_encode call: for a sink that only tells by failing how
 *       much room it had -- encode without a limit, hand the bytes on, and if the sink fails after k of them rewind and
 *       encode the same piece with sink_room = k to learn what the reference would have returned.
 */
typedef struct rcx_estream rcx_estream;
int rcx_estream_create(rcx_ctx* ctx, uint32_t declared, rcx_estream** out);
void rcx_estream_destroy(rcx_estream* stream);
int rcx_estream_encode(rcx_estream* stream, const uint8_t* bytes, uint64_t size, uint8_t* dst, uint64_t dst_cap, uint64_t sink_room,
                       uint64_t* emitted_now, uint32_t* tail_bytes, uint32_t* request_size);
int rcx_estream_rewind(rcx_estream* stream);

/*
 * The block sort (blksort.h): the Burrows-Wheeler transform of 32 KiB blocks the reference harness runs in front of
 * its entropy coders (test/main.cpp:961-986), and its inverse.  Replaces blksort::BlkSort:
 *   rcx_bwt_encode_bound   BlkSort::encodeBound (blksort.h:426-431): n + 2 bytes per whole block of 32768
 *   rcx_bwt_decode_bound   BlkSort::decodeBound (blksort.h:433-438): a bound, and as the reference computes it, n
 *   rcx_bwt_decoded_size   what BlkSort::decode writes for n encoded bytes (blksort.h:451-462: n / 32770 blocks of
 *                          32768 bytes and the rest as it is)
 *   rcx_bwt_encode(_device)   BlkSort::encode (blksort.h:440-449): every whole block becomes the last column of its
 *       sorted rotations + the row of the unrotated block as a little-endian u16; a shorter rest is copied.  Byte for
 *       byte the reference's output -- including the row it stores for a PERIODIC block, whose rotations tie: that row
 *       depends on the moves of the reference's unstable sort, which are replayed for such blocks.
 *   rcx_bwt_decode(_device)   BlkSort::decode (blksort.h:451-462, the source is not modified).  A stored row of 32768 or
 *       more, with which the reference reads outside its arrays, is RCX_E_CORRUPT (through rcx_ctx_sync_status for
 *       the device call).
 * Source and destination must not overlap (blocks are transformed concurrently).
 * The _device calls take device pointers of any alignment, enqueue on `stream` and do not synchronise; scratch is
 * grown on first use or ahead of time by rcx_bwt_reserve(ctx, n).  Capacities are checked before anything is enqueued
 * (RCX_E_CAPACITY); the host-buffer calls also report the size needed in *dst_size.
 * rcx_bwt_last_ties: how many blocks of the last forward call were periodic with a period above 1 and went through the
 * replay (diagnostic; synchronises).
 */
#define RCX_BWT_BLOCK_BYTES 32768u
#define RCX_BWT_ENCODED_BYTES 32770u
uint64_t rcx_bwt_encode_bound(uint64_t n);
uint64_t rcx_bwt_decode_bound(uint64_t n);
uint64_t rcx_bwt_decoded_size(uint64_t n);
int rcx_bwt_reserve(rcx_ctx* ctx, uint64_t n);
int rcx_bwt_encode_device(rcx_ctx* ctx, const void* d_src, uint64_t n, void* d_dst, uint64_t dst_cap, void* stream);
int rcx_bwt_decode_device(rcx_ctx* ctx, const void* d_src, uint64_t n, void* d_dst, uint64_t dst_cap, void* stream);
int rcx_bwt_encode(rcx_ctx* ctx, const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t dst_cap, uint64_t* dst_size);
int rcx_bwt_decode(rcx_ctx* ctx, const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t dst_cap, uint64_t* dst_size);
int rcx_bwt_last_ties(rcx_ctx* ctx, uint64_t* count);

/*
 * Multi-GPU (new; the reference has no multi-device code).  Blocks are independent, so n bytes are sharded over the
 * GPUs of a node as contiguous block ranges and coded with the calls above, no collective involved.  The one real
 * exchange of the path is putting the compressed segments of all GPUs -- and their block tables -- together on
 * every GPU ("allgatherv"; RCCL has no such call): one process per GPU, one rcx_comm per process, RCCL over xGMI.
 *   rcx_comm_unique_id   rank 0 makes an id (RCX_COMM_ID_BYTES bytes) and hands it to the other ranks out of band
 *   rcx_comm_create      collective over the nranks processes: rank `rank` joins on GPU `device`
 *   rcx_exchange_plan    pure function: exclusive prefix sums of what every rank brings (segment bytes, blocks)
 *   rcx_allgatherv_segments   collective, enqueued on `stream` except for ONE host synchronisation (the sizes):
 *       d_segment / d_offsets[nblocks+1]   this rank's compacted streams and their table, as rcx_encode_blocks_device
 *                                          wrote them (offsets[nblocks] is the segment size; ranks may differ in nblocks)
 *       d_concat (concat_cap bytes)        receives every rank's segment back to back, rank 0 first
 *       d_table (table_cap entries, or NULL)  receives the table of the concatenation: sum(nblocks)+1 offsets into d_concat
 *       seg_base_out / block_base_out      optional host arrays of nranks+1: where rank r's bytes / blocks start
 *     The capacities and whether a table is wanted travel with the sizes: every rank judges by the smallest room any rank
 *     has, so RCX_E_CAPACITY -- or RCX_E_ARG if only some ranks pass a table -- is returned by ALL ranks, before anything moves.
 *     (Two or more ranks over RCCL have not run on hardware yet: the 1-GPU box runs one rank, tests/test_gpu_comm.py; the
 *     logic for more is checked on gloo, tests/test_parallel_gloo.py.)
 */
#define RCX_COMM_ID_BYTES 128
typedef struct rcx_comm rcx_comm;
int rcx_comm_unique_id(void* id);
int rcx_comm_create(int device, const void* id, int nranks, int rank, rcx_comm** out);
void rcx_comm_destroy(rcx_comm* comm);
int rcx_comm_rank(const rcx_comm* comm);
int rcx_comm_size(const rcx_comm* comm);
int rcx_exchange_plan(const uint64_t* seg_bytes, const uint64_t* nblocks, int nranks, uint64_t* seg_base, uint64_t* block_base);
int rcx_allgatherv_segments(rcx_comm* comm, const void* d_segment, const uint64_t* d_offsets, uint64_t nblocks,
                            void* d_concat, uint64_t concat_cap, uint64_t* d_table, uint64_t table_cap,
                            uint64_t* seg_base_out, uint64_t* block_base_out, void* stream);

/* Per-kernel device time of the calls made since the last reset, in milliseconds,
 * measured with HIP events on the stream the kernels ran on (off by default). */
enum { RCX_T_ENCODE = 0, RCX_T_SCAN = 1, RCX_T_SCATTER = 2, RCX_T_DECODE = 3, RCX_T_BWT_FORWARD = 4, RCX_T_BWT_INVERSE = 5, RCX_T_COUNT = 6 };
/* How many of the first `nblocks` blocks of the LAST rcx_encode_blocks* / rcx_decode_blocks* call on this
 * context were handed to the one-lane kernels (the many-lane kernels mark, and do not finish, a block whose
 * carry runs through more output bytes than they keep in LDS, or whose stream asks for a symbol past the
 * table; results are the same either way).  Synchronises the device.  Diagnostic: 0 on ordinary data. */
int rcx_ctx_last_redo(rcx_ctx* ctx, uint64_t nblocks, uint64_t* count);

int rcx_ctx_set_timing(rcx_ctx* ctx, int enabled);
/* Synchronises the events; ms[RCX_T_COUNT] = summed durations, launches[RCX_T_COUNT] = launch counts. */
int rcx_ctx_get_timing(rcx_ctx* ctx, double* ms, uint64_t* launches, int reset);

#ifdef __cplusplus
}
#endif
#endif /* RCX_H_ */
