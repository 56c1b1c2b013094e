/*
 * cpprcoder.h (MI355X facade) -- the reference's class API on top of the rcx C ABI.
 *
 * A caller written against taqu/cpprcoder's cpprcoder.h (namespace cpprcoder,
 * MemoryStream, AdaptiveRangeEncoder<T>, AdaptiveRangeDecoder<T>, Status, Result) can
 * include this header instead and link librcx.so: the names, argument meaning, return
 * values and the bytes that reach the sink are the reference's
 *     Status / Result                      cpprcoder.h:112-123
 *     IStream<T> / MemoryStream            cpprcoder.h:130-247, 964-1077
 *     AdaptiveRangeEncoder<T>              cpprcoder.h:626-802
 *     AdaptiveRangeDecoder<T>              cpprcoder.h:809-940
 *     RangeEncoder<T> (static coder)       cpprcoder.h:321-619
 * but the coding itself runs on the GPU (rcx_stream_encode / rcx_stream_decode).
 *
 * What differs, by design:
 *   - The GPU codes whole streams.  encode() collects its pieces until the declared
 *     size has been seen (returning {Status_Pending, remaining} like the reference)
 *     and only then emits bytes into the sink; the sink's final contents, the return
 *     values, and the behaviour of a sink that fills up are the reference's.
 *   - decode() keeps its coder state on the GPU between calls (rcx_dstream), so a stream fed in pieces costs what
 *     the pieces cost; after a sink that filled up it keeps answering Status_Pending (the reference's decoder has
 *     lost a symbol at that point, cpprcoder.h:909-911, and is unusable as well).
 *   - One stream is one GPU lane: this facade is for drop-in compatibility.  Throughput
 *     comes from coding many blocks at once through rcx_encode_blocks_device (rcx.h),
 *     see BlockCoder below.
 *   - Streams may be as long as the reference's (MAX_SIZE = 0x7FFFFFFF); past 2^24 - 256 symbols the GPU lane
 *     halves its table as cpprcoder.h:1138-1176 does.
 */
#ifndef INC_CPPRCODER_AMD_FACADE_H_
#define INC_CPPRCODER_AMD_FACADE_H_

#include <cassert>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../rcx.h"

#ifndef CPPRCODER_ASSERT
#    define CPPRCODER_ASSERT(exp) assert(exp)
#endif

namespace cpprcoder
{
typedef int8_t s8;
typedef int16_t s16;
typedef int32_t s32;
typedef int64_t s64;
typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

enum Status
{
    Status_Success = RCX_OK,
    Status_Pending = RCX_PENDING,
    Status_Error = RCX_ERROR,
};

struct Result
{
    Status status_;
    u32 requestSize_;
};

// One rcx context per thread, created on first use on device RCX_DEVICE (default 0), destroyed with the thread.
inline rcx_ctx* facade_context()
{
    struct Holder {
        rcx_ctx* ctx = nullptr;
        ~Holder() { rcx_ctx_destroy(ctx); }
    };
    static thread_local Holder h;
    if (!h.ctx) {
        const char* dev = getenv("RCX_DEVICE");
        if (rcx_ctx_create(dev ? atoi(dev) : 0, &h.ctx) != RCX_OK) h.ctx = nullptr; // no CPU fallback: callers see Status_Error
    }
    return h.ctx;
}

//--- IStream (cpprcoder.h:130-166)
template<class T>
class IStream
{
public:
    s32 read(s32 size, u8* bytes) { return static_cast<T*>(this)->read(size, bytes); }
    bool readByte(u8& byte) { return static_cast<T*>(this)->readByte(byte); }
    s32 write(s32 size, const u8* bytes) { return static_cast<T*>(this)->write(size, bytes); }
    bool writeByte(u8 byte) { return static_cast<T*>(this)->writeByte(byte); }

protected:
    IStream() {}
    ~IStream() {}

private:
    IStream(const IStream&) = delete;
    IStream& operator=(const IStream&) = delete;
};

//--- MemoryStream (cpprcoder.h:185-247, 964-1077): write() grows, writeByte() never does
class MemoryStream : public IStream<MemoryStream>
{
public:
    MemoryStream() : capacity_(0), size_(0), buffer_(nullptr) {}
    explicit MemoryStream(s32 capacity) : capacity_(capacity), size_(0)
    {
        capacity_ = (capacity_ <= 0) ? 16 : static_cast<s32>((static_cast<u32>(capacity_) + 15U) & ~15U);
        buffer_ = static_cast<u8*>(malloc(static_cast<size_t>(capacity_)));
    }
    ~MemoryStream() { free(buffer_); }

    s32 capacity() const { return capacity_; }
    s32 size() const { return size_; }
    const u8* get() const { return buffer_; }
    const u8& operator[](s32 index) const { CPPRCODER_ASSERT(0 <= index && index < size_); return buffer_[index]; }
    u8& operator[](s32 index) { CPPRCODER_ASSERT(0 <= index && index < size_); return buffer_[index]; }

    void reserve(s32 capacity) // discards the contents unless the request is smaller (cpprcoder.h:985-994)
    {
        capacity = static_cast<s32>((static_cast<u32>(capacity) + 15U) & ~15U);
        if (capacity < capacity_) return;
        free(buffer_);
        capacity_ = capacity;
        buffer_ = static_cast<u8*>(malloc(static_cast<size_t>(capacity_)));
    }
    void resize(s32 size)
    {
        CPPRCODER_ASSERT(0 <= size);
        if (capacity_ < size) reserve(size);
        size_ = size;
    }
    s32 read(s32 size, u8* bytes)
    {
        s32 end = size_ + size;
        if (capacity_ < end) return -1;
        memcpy(bytes, buffer_ + size_, static_cast<size_t>(size));
        size_ = end;
        return size;
    }
    bool readByte(u8& byte)
    {
        if (capacity_ < size_ + 1) return false;
        byte = buffer_[size_++];
        return true;
    }
    s32 write(s32 size, const u8* bytes)
    {
        s32 end = size_ + size;
        if (capacity_ < end && !expand(end)) return -1;
        memcpy(buffer_ + size_, bytes, static_cast<size_t>(size));
        size_ = end;
        return size;
    }
    bool writeByte(u8 byte)
    {
        if (capacity_ <= size_) return false;
        buffer_[size_++] = byte;
        return true;
    }

private:
    MemoryStream(const MemoryStream&) = delete;
    MemoryStream& operator=(const MemoryStream&) = delete;

    bool expand(s32 size) // 0 -> 1024, doubling below 16 KiB, then +16 KiB steps (cpprcoder.h:1056-1077)
    {
        s32 cap = capacity_;
        do {
            if (cap <= 0) cap = 1024;
            else if (cap < 4096 * 4) cap <<= 1;
            else cap += 4096 * 4;
        } while (cap < size);
        cap = static_cast<s32>((static_cast<u32>(cap) + 15U) & ~15U);
        u8* fresh = static_cast<u8*>(malloc(static_cast<size_t>(cap)));
        if (!fresh) return false;
        if (0 < capacity_) memcpy(fresh, buffer_, static_cast<size_t>(capacity_));
        free(buffer_);
        buffer_ = fresh;
        capacity_ = cap;
        return true;
    }

    s32 capacity_;
    s32 size_;
    u8* buffer_;
};

//--- AdaptiveRangeEncoder (cpprcoder.h:626-802): the encoder's state lives on the GPU between calls (rcx_estream), and
//    every call passes to the sink what the reference passes to it during that call -- the payload bytes that have become
//    final (everything but the coder's held byte and the 0xFF bytes pending behind it, cpprcoder.h:764-802) through
//    writeByte(), the last call also the final low through write(4) (cpprcoder.h:744-762).
template<class T = MemoryStream>
class AdaptiveRangeEncoder
{
public:
    AdaptiveRangeEncoder() : stream_(nullptr), encoder_(nullptr), umcompressedSize_(0), inSize_(0), dead_(false) {}
    ~AdaptiveRangeEncoder() { rcx_estream_destroy(encoder_); }

    // cpprcoder.h:678-695: the header goes out at once through write()
    bool initialize(T& stream, u32 umcompressedSize)
    {
        stream_ = &stream;
        umcompressedSize_ = umcompressedSize;
        inSize_ = 0;
        dead_ = false;
        rcx_estream_destroy(encoder_);
        encoder_ = nullptr;
        u8 bytes[4] = {static_cast<u8>(umcompressedSize), static_cast<u8>(umcompressedSize >> 8),
                       static_cast<u8>(umcompressedSize >> 16), static_cast<u8>(umcompressedSize >> 24)};
        return 0 < stream_->write(4, bytes);
    }

    // cpprcoder.h:697-720
    Result encode(s32 size, const u8* bytes)
    {
        CPPRCODER_ASSERT((inSize_ + size) <= umcompressedSize_);
        if (dead_) return {Status_Pending, umcompressedSize_ - inSize_};
        rcx_ctx* ctx = facade_context();
        if (!ctx) return {Status_Error, 0};
        // The whole stream in one call (how the reference's harness drives it, test/main.cpp:330): the many-lane kernels
        // run one chain three to four times as fast as the resumable one-lane coder below.
        if (inSize_ == 0 && !encoder_ && static_cast<u32>(size) == umcompressedSize_) return whole(ctx, bytes);
        if (!encoder_ && rcx_estream_create(ctx, umcompressedSize_, &encoder_) != RCX_OK) return {Status_Error, 0};
        const uint64_t piece = static_cast<uint64_t>(size);
        if (out_.size() < 3 * piece + 4096) out_.resize(3 * piece + 4096);
        uint64_t got = 0;
        uint32_t tail = 0, req = 0;
        int st = rcx_estream_encode(encoder_, bytes, piece, out_.data(), out_.size(), UINT64_MAX, &got, &tail, &req);
        if (st == RCX_E_CAPACITY) { // a pending run longer than the guess: once more with the room it asked for
            out_.resize(got + 64);
            if (rcx_estream_rewind(encoder_) != RCX_OK) return {Status_Error, 0};
            st = rcx_estream_encode(encoder_, bytes, piece, out_.data(), out_.size(), UINT64_MAX, &got, &tail, &req);
        }
        if (st != RCX_OK && st != RCX_PENDING) return {Status_Error, 0};
        // Replay the reference's sink calls: payload byte by byte through writeByte (never grows) ...
        const uint64_t through_write_byte = got - tail;
        for (uint64_t i = 0; i < through_write_byte; ++i) {
            if (!stream_->writeByte(out_[i])) {
                // The sink took i bytes of this call.  What the reference returns then depends on where it was: inside a symbol
                // (Pending, the symbols before it counted, cpprcoder.h:708-711) or in finish() (Success, cpprcoder.h:716).  The
                // device knows: the same piece again, against a sink with room for exactly i bytes.
                dead_ = true; // the reference's coder is unusable from here on too
                if (rcx_estream_rewind(encoder_) != RCX_OK) return {Status_Error, 0};
                uint64_t got2 = 0;
                const int again = rcx_estream_encode(encoder_, bytes, piece, out_.data(), out_.size(), i, &got2, &tail, &req);
                if (again == RCX_PENDING) {
                    inSize_ = umcompressedSize_ - req;
                    return {Status_Pending, req};
                }
                if (again == RCX_OK) {
                    inSize_ = umcompressedSize_;
                    return {Status_Success, 0};
                }
                return {Status_Error, 0};
            }
        }
        // ... and the last four bytes through write (may grow): cpprcoder.h:756-761, result ignored as in the reference
        if (tail) stream_->write(static_cast<s32>(tail), out_.data() + through_write_byte);
        inSize_ += static_cast<u32>(size);
        if (st == RCX_OK) return {Status_Success, 0};
        return {Status_Pending, req};
    }

    // cpprcoder.h:722-742
    Result encode(u8 byte) { return encode(1, &byte); }

private:
    AdaptiveRangeEncoder(const AdaptiveRangeEncoder&) = delete;
    AdaptiveRangeEncoder& operator=(const AdaptiveRangeEncoder&) = delete;

    Result whole(rcx_ctx* ctx, const u8* bytes)
    {
        const u32 n = umcompressedSize_;
        std::vector<u8> out(static_cast<size_t>(rcx_block_bound(n < RCX_MIN_BLOCK ? RCX_MIN_BLOCK : n)) + 64);
        uint64_t size = 0;
        uint32_t req = 0;
        if (rcx_stream_encode(ctx, RCX_CODER_ADAPTIVE, bytes, n, out.data(), out.size(), out.size(), &size, &req) != RCX_OK)
            return {Status_Error, 0};
        // Replay the reference's sink calls: payload byte by byte through writeByte (never grows),
        // the last four bytes through write (may grow) -- cpprcoder.h:744-762, 783-800.
        const uint64_t payload_end = size - 4;
        for (uint64_t i = 4; i < payload_end; ++i) {
            if (!stream_->writeByte(out[i])) {
                dead_ = true; // the reference's coder is unusable from here on too
                // Which symbol was being coded when byte i did not fit?  Ask the device to replay the
                // reference's delayed writer against a sink that accepts exactly i bytes.
                uint64_t size2 = 0;
                int st = rcx_stream_encode(ctx, RCX_CODER_ADAPTIVE, bytes, n, out.data(), out.size(), i, &size2, &req);
                if (st == RCX_PENDING) {
                    inSize_ = n - req;
                    return {Status_Pending, req}; // cpprcoder.h:708-711
                }
                inSize_ = n;
                return {Status_Success, 0}; // only finish() ran into the full sink (cpprcoder.h:716)
            }
        }
        stream_->write(4, out.data() + payload_end); // result ignored, as in the reference
        inSize_ = n;
        return {Status_Success, 0};
    }

    T* stream_;
    rcx_estream* encoder_;
    u32 umcompressedSize_;
    u32 inSize_;
    bool dead_;
    std::vector<u8> out_;
};

//--- AdaptiveRangeDecoder (cpprcoder.h:809-940): the decoder's state lives on the GPU between calls (rcx_dstream)
template<class T = MemoryStream>
class AdaptiveRangeDecoder
{
public:
    AdaptiveRangeDecoder() : stream_(nullptr), decoder_(nullptr), outSize_(0), declared_(0), started_(false), dead_(false) {}
    ~AdaptiveRangeDecoder() { rcx_dstream_destroy(decoder_); }

    bool initialize(T& stream) // cpprcoder.h:859-870
    {
        stream_ = &stream;
        outSize_ = 0;
        declared_ = 0;
        started_ = false;
        dead_ = false;
        rcx_dstream_destroy(decoder_);
        decoder_ = nullptr;
        return true;
    }

    // cpprcoder.h:872-924: every call decodes what its bytes allow and writes it to the sink
    Result decode(s32 size, const u8* bytes)
    {
        if (dead_) return {Status_Pending, declared_ - outSize_}; // the reference's decoder is unusable after a full sink too
        if (!started_ && size < 8) return {Status_Pending, 8};    // State_Init needs 8 bytes in one call (:877-880)
        rcx_ctx* ctx = facade_context();
        if (!ctx) return {Status_Error, 0};
        if (!decoder_ && rcx_dstream_create(ctx, &decoder_) != RCX_OK) return {Status_Error, 0};
        if (!started_) {
            declared_ = static_cast<u32>(bytes[0]) | (static_cast<u32>(bytes[1]) << 8) | (static_cast<u32>(bytes[2]) << 16) |
                        (static_cast<u32>(bytes[3]) << 24);
            started_ = true;
        }
        if (chunk_.empty()) chunk_.resize(1u << 16);
        const u8* feed = bytes;
        uint64_t feed_size = static_cast<uint64_t>(size);
        for (;;) {
            uint64_t got = 0;
            uint32_t req = 0;
            const int st = rcx_dstream_decode(decoder_, feed, feed_size, chunk_.data(), chunk_.size(), &got, &req);
            feed = nullptr;
            feed_size = 0;
            if (st != RCX_OK && st != RCX_PENDING) return {Status_Error, 0};
            for (uint64_t i = 0; i < got; ++i) {
                if (!stream_->writeByte(chunk_[i])) { // cpprcoder.h:909-911
                    dead_ = true;
                    return {Status_Pending, declared_ - outSize_};
                }
                ++outSize_;
            }
            if (st == RCX_OK) return {Status_Success, 0};
            if (got < chunk_.size()) return {Status_Pending, req}; // the input ran dry (cpprcoder.h:901-903)
        }
    }

private:
    AdaptiveRangeDecoder(const AdaptiveRangeDecoder&) = delete;
    AdaptiveRangeDecoder& operator=(const AdaptiveRangeDecoder&) = delete;

    T* stream_;
    rcx_dstream* decoder_;
    u32 outSize_;
    u32 declared_;
    bool started_;
    bool dead_;
    std::vector<u8> chunk_;
};

//--- RangeEncoder: the static two-pass coder (cpprcoder.h:321-619); encode and decode are members of one class
template<class T = MemoryStream>
class RangeEncoder
{
public:
    static const u32 MAX_SIZE = 0x7FFFFFFFU;
    static const u32 FREQUENCY_SIZE = 256;
    static const u32 HeaderSize16 = sizeof(u32) + sizeof(u16) * FREQUENCY_SIZE;

    RangeEncoder() {}

    // cpprcoder.h:375-458: header and table through write(), every payload byte through writeByte(),
    // the last four bytes through write(); false as soon as the sink refuses one.
    bool encode(T& stream, u32 size, const u8* bytes)
    {
        CPPRCODER_ASSERT(size <= MAX_SIZE);
        rcx_ctx* ctx = facade_context();
        if (!ctx || RCX_MAX_STREAM < size) return false;
        std::vector<u8> out(static_cast<size_t>(rcx_block_bound(size < RCX_MIN_BLOCK ? RCX_MIN_BLOCK : size)) + 64);
        uint64_t total = 0;
        if (rcx_stream_encode(ctx, RCX_CODER_STATIC, bytes, size, out.data(), out.size(), out.size(), &total, nullptr) != RCX_OK) return false;
        if (stream.write(4, out.data()) <= 0) return false;
        for (u32 i = 0; i < 4; ++i) { // write16(): 4 x 64 counts (cpprcoder.h:604-619)
            if (stream.write(128, out.data() + 4 + 128 * i) <= 0) return false;
        }
        const uint64_t payload_end = total - 4;
        for (uint64_t i = HeaderSize16; i < payload_end; ++i) {
            if (!stream.writeByte(out[i])) return false;
        }
        return 0 < stream.write(4, out.data() + payload_end);
    }

    // cpprcoder.h:460-519
    bool decode(T& stream, u32 size, const u8* bytes)
    {
        CPPRCODER_ASSERT(size <= MAX_SIZE);
        rcx_ctx* ctx = facade_context();
        if (!ctx) return false;
        if (size < HeaderSize16) return false;
        const u32 declared = static_cast<u32>(bytes[0]) | (static_cast<u32>(bytes[1]) << 8) | (static_cast<u32>(bytes[2]) << 16) |
                             (static_cast<u32>(bytes[3]) << 24);
        if (declared == 0) return true;
        if (declared > MAX_SIZE) return false; // a damaged header: nothing the encoder could have written (cpprcoder.h:329)
        std::vector<u8> out;
        try {
            out.resize(static_cast<size_t>(declared) + 64);
        } catch (...) {
            return false; // no memory for what the header claims
        }
        uint64_t produced = 0;
        int st = rcx_stream_decode(ctx, RCX_CODER_STATIC, bytes, size, out.data(), declared, &produced, nullptr);
        if (st != RCX_OK && st != RCX_ERROR) return false;
        for (uint64_t i = 0; i < produced; ++i) {
            if (!stream.writeByte(out[i])) return false;
        }
        return st == RCX_OK;
    }

private:
    RangeEncoder(const RangeEncoder&) = delete;
    RangeEncoder& operator=(const RangeEncoder&) = delete;
};

//--- BlockCoder: the many-blocks entry point in the facade's vocabulary (new; the reference has
//    no counterpart).  Host buffers in, host buffers out; see rcx.h for the device-pointer calls.
class BlockCoder
{
public:
    explicit BlockCoder(u32 blockSize = 65536) : block_(blockSize) {}

    // dst is resized to the compacted streams; offsets gets nblocks+1 entries
    bool encode(std::vector<u8>& dst, std::vector<u64>& offsets, u64 size, const u8* bytes)
    {
        rcx_ctx* ctx = facade_context();
        if (!ctx) return false;
        const u64 nblocks = rcx_block_count(size, block_);
        dst.resize(static_cast<size_t>(rcx_encode_bound(size, block_)));
        offsets.resize(static_cast<size_t>(nblocks + 1));
        uint64_t total = 0;
        if (rcx_encode_blocks(ctx, RCX_CODER_ADAPTIVE, bytes, size, block_, dst.data(), dst.size(), &total, offsets.data()) != RCX_OK)
            return false;
        dst.resize(static_cast<size_t>(total));
        return true;
    }

    bool decode(std::vector<u8>& dst, const std::vector<u8>& src, const std::vector<u64>& offsets)
    {
        rcx_ctx* ctx = facade_context();
        if (!ctx || offsets.empty()) return false;
        const u64 nblocks = offsets.size() - 1;
        dst.resize(static_cast<size_t>(nblocks * block_));
        uint64_t total = 0;
        if (rcx_decode_blocks(ctx, RCX_CODER_ADAPTIVE, src.data(), src.size(), offsets.data(), nblocks, block_, dst.data(),
                              dst.size(), &total) != RCX_OK)
            return false;
        dst.resize(static_cast<size_t>(total));
        return true;
    }

private:
    u32 block_;
};

} // namespace cpprcoder
#endif // INC_CPPRCODER_AMD_FACADE_H_
