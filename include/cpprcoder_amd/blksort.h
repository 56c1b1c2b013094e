/*
 * blksort.h (MI355X facade) -- the reference's block-sort class on top of the rcx C ABI.
 *
 * A caller written against taqu/cpprcoder's blksort.h (namespace blksort, class BlkSort; blksort.h:78-110) can include
 * this header instead and link librcx.so: the signatures, sizes and bytes are the reference's
 *     static uint32_t BlkSort::encodeBound(uint32_t size)                               blksort.h:426-431
 *     static uint32_t BlkSort::decodeBound(uint32_t size)                               blksort.h:433-438
 *     void BlkSort::encode(uint32_t size, uint8_t* dst, const uint8_t* data)            blksort.h:440-449
 *     void BlkSort::decode(uint32_t size, uint8_t* dst, uint8_t* data)                  blksort.h:451-462
 * (as run_blksort uses them, test/main.cpp:812-825) but the sorting runs on the GPU: all whole 32 KiB blocks of a call
 * at once, one workgroup per block (rcx_bwt_encode / rcx_bwt_decode).  The reference's functions return nothing; a
 * failure here (no GPU, a stored row index of 32768 or more in decode) leaves dst untouched and is reported by ok().
 * The move-to-front stage is compiled out in the reference (BLOCKSORT_MTF 0, blksort.h:54) and is not part of this.
 */
#ifndef INC_BLKSORT_AMD_FACADE_H_
#define INC_BLKSORT_AMD_FACADE_H_

#include <cstdint>
#include <cstdlib>

#include "../rcx.h"

namespace blksort
{
class BlkSort
{
public:
    inline static constexpr uint32_t Align = 16;
    inline static constexpr uint32_t BlockSize = 1024 * 32;
    inline static constexpr uint32_t BlockShift = 15;
    inline static constexpr uint32_t BlockMask = BlockSize - 1;
    inline static constexpr uint32_t EncodedSize = BlockSize + 2;

    BlkSort()
    {
        const char* dev = getenv("RCX_DEVICE");
        if (rcx_ctx_create(dev ? atoi(dev) : 0, &ctx_) != RCX_OK) ctx_ = nullptr; // no CPU fallback
        status_ = ctx_ ? RCX_OK : RCX_E_HIP;
    }
    ~BlkSort() { rcx_ctx_destroy(ctx_); }

    static uint32_t encodeBound(uint32_t size) { return static_cast<uint32_t>(rcx_bwt_encode_bound(size)); }
    static uint32_t decodeBound(uint32_t size) { return static_cast<uint32_t>(rcx_bwt_decode_bound(size)); }

    void encode(uint32_t size, uint8_t* dst, const uint8_t* data)
    {
        uint64_t written = 0;
        status_ = ctx_ ? rcx_bwt_encode(ctx_, data, size, dst, rcx_bwt_encode_bound(size), &written) : RCX_E_HIP;
    }
    void decode(uint32_t size, uint8_t* dst, uint8_t* data)
    {
        uint64_t written = 0;
        status_ = ctx_ ? rcx_bwt_decode(ctx_, data, size, dst, rcx_bwt_decoded_size(size), &written) : RCX_E_HIP;
    }

    // not in the reference: did the last call (or the constructor) succeed, and with what rcx status
    bool ok() const { return status_ == RCX_OK; }
    int status() const { return status_; }

private:
    BlkSort(const BlkSort&) = delete;
    BlkSort& operator=(const BlkSort&) = delete;

    rcx_ctx* ctx_ = nullptr;
    int status_ = RCX_OK;
};
} // namespace blksort
#endif // INC_BLKSORT_AMD_FACADE_H_
