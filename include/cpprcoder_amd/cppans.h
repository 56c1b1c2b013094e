/*
 * cppans.h (MI355X facade) -- the reference's rANS class API on top of the rcx C ABI.
 *
 * A caller written against taqu/cpprcoder's cppans.h (namespace cppans, class rANS with the static functions
 * calc_encoded_size / encode / decode / encode_simd / decode_simd; cppans.h:23-80) can include this header instead and
 * link librcx.so: the signatures, the return values and the bytes are the reference's
 *     u64 rANS::calc_encoded_size(u32 size)                              cppans.h:492-495
 *     u32 rANS::encode(u32 dst_size, u8* dst, u32 src_size, const u8*)   cppans.h:497-530
 *     u32 rANS::decode(u32 dst_size, u8* dst, u32 src_size, const u8*)   cppans.h:532-564
 *     u32 rANS::encode_simd / decode_simd                                cppans.h:567-649
 * -- including that the encoders leave the stream in the LAST `return value` bytes of dst (test/main.cpp:384-387) --
 * but the coding runs on the GPU (rcx_stream_encode / rcx_stream_decode with RCX_CODER_RANS / RCX_CODER_RANS8).
 * One stream is one GPU block (8 lanes); throughput comes from coding many blocks at once through
 * rcx_encode_blocks_device (rcx.h).  Where the reference asserts (empty input) or would leave its arrays (a damaged
 * table, a payload that runs out), these return 0.
 */
#ifndef INC_CPPANS_AMD_FACADE_H_
#define INC_CPPANS_AMD_FACADE_H_

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../rcx.h"

namespace cppans
{
using s8 = int8_t;
using s16 = int16_t;
using s32 = int32_t;
using s64 = int64_t;
using u8 = uint8_t;
using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;

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
        if (rcx_ctx_create(dev ? atoi(dev) : 0, &h.ctx) != RCX_OK) h.ctx = nullptr; // no CPU fallback: callers see 0
    }
    return h.ctx;
}

class rANS
{
public:
    inline static constexpr u32 MaxSize = 0x7FFFFFFFUL;
    inline static constexpr u32 ProbBits = 14;
    inline static constexpr u32 ProbScale = 1 << ProbBits;
    inline static constexpr u32 rANSByteLowBounds = 1UL << 23;
    inline static constexpr u32 WordLowBounds = 1UL << 16;
    inline static constexpr u32 WordScaleBits = 12;
    inline static constexpr u32 WordM = 1 << WordScaleBits;

    using State = u32;

    static u64 calc_encoded_size(u32 size) { return static_cast<u64>(size) * 2 + sizeof(u32) * 258; } // cppans.h:492-495

    static u32 encode(u32 dst_size, u8* dst, u32 src_size, const u8* src) { return encode_with(RCX_CODER_RANS, dst_size, dst, src_size, src); }
    static u32 encode_simd(u32 dst_size, u8* dst, u32 src_size, const u8* src) { return encode_with(RCX_CODER_RANS8, dst_size, dst, src_size, src); }
    static u32 decode(u32 dst_size, u8* dst, u32 src_size, const u8* src) { return decode_with(RCX_CODER_RANS, dst_size, dst, src_size, src); }
    static u32 decode_simd(u32 dst_size, u8* dst, u32 src_size, const u8* src) { return decode_with(RCX_CODER_RANS8, dst_size, dst, src_size, src); }

private:
    rANS(const rANS&) = delete;
    rANS& operator=(const rANS&) = delete;

    static u32 encode_with(int coder, u32 dst_size, u8* dst, u32 src_size, const u8* src)
    {
        rcx_ctx* ctx = facade_context();
        if (!ctx || !dst || !src || src_size == 0) return 0;
        std::vector<u8> out(static_cast<size_t>(rcx_block_bound_for(coder, src_size < RCX_MIN_BLOCK ? RCX_MIN_BLOCK : src_size)));
        uint64_t size = 0;
        if (rcx_stream_encode(ctx, coder, src, src_size, out.data(), out.size(), out.size(), &size, nullptr) != RCX_OK) return 0;
        if (size > dst_size) return 0; // cppans.h:522-524 / :599-601: the header does not fit in front of the payload
        memcpy(dst + dst_size - size, out.data(), static_cast<size_t>(size)); // the stream is the END of dst
        return static_cast<u32>(size);
    }

    static u32 decode_with(int coder, u32 dst_size, u8* dst, u32 src_size, const u8* src)
    {
        rcx_ctx* ctx = facade_context();
        if (!ctx || !dst || !src || src_size < sizeof(u32) * 258) return 0;
        uint64_t produced = 0;
        uint32_t ret = 0;
        if (rcx_stream_decode(ctx, coder, src, src_size, dst, dst_size, &produced, &ret) != RCX_OK) return 0;
        return ret; // decode: payload bytes consumed (cppans.h:562); decode_simd: the symbol count (cppans.h:648)
    }
};
} // namespace cppans
#endif // INC_CPPANS_AMD_FACADE_H_
