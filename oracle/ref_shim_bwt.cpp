// ref_shim_bwt.cpp -- C entry points around the UNMODIFIED reference header blksort.h.
//
// TEST INFRASTRUCTURE ONLY (see ref_shim.cpp).  This translation unit contains no reference code: it includes the
// reference's blksort.h from where it lies (-DRCX_REFERENCE_BWT_HEADER="/root/reference/blksort.h", oracle/Makefile)
// with BLKSORT_IMPLEMENTATION defined, as the reference's own blksort.cpp:1-2 does, and is compiled into
// oracle/_ref/libblksort_ref.so, which is git-ignored.
//
// The call sequences follow the reference harness: test/main.cpp:812-825 (run_blksort).
#ifndef RCX_REFERENCE_BWT_HEADER
#error "compile with -DRCX_REFERENCE_BWT_HEADER=\"/path/to/blksort.h\""
#endif
#define BLKSORT_IMPLEMENTATION (1)
#include RCX_REFERENCE_BWT_HEADER

#include <cstdint>
#include <cstring>
#include <vector>

extern "C" {

// same shapes as rco_bwt_* (bwt_oracle.c); the reference's sizes are u32
uint64_t ref_bwt_encode_bound(uint64_t n) { return blksort::BlkSort::encodeBound(static_cast<uint32_t>(n)); }
uint64_t ref_bwt_decode_bound(uint64_t n) { return blksort::BlkSort::decodeBound(static_cast<uint32_t>(n)); }
// what BlkSort::decode writes for n encoded bytes: its block count is n / EncodedSize (blksort.h:453-454)
uint64_t ref_bwt_decoded_size(uint64_t n)
{
    const uint64_t blocks = n / blksort::BlkSort::EncodedSize;
    return blocks * blksort::BlkSort::BlockSize + (n - blocks * blksort::BlkSort::EncodedSize);
}

// dst: encodeBound(n) bytes
int ref_bwt_encode(const uint8_t* src, uint64_t n, uint8_t* dst)
{
    blksort::BlkSort coder;
    coder.encode(static_cast<uint32_t>(n), dst, src);
    return 0;
}

// dst: ref_bwt_decoded_size(n) bytes.  BlkSort::decode takes a non-const source (it would MTF-decode in place if
// BLOCKSORT_MTF were on, blksort.h:545-547): it gets a copy.
int ref_bwt_decode(const uint8_t* src, uint64_t n, uint8_t* dst)
{
    std::vector<uint8_t> copy(src, src + n);
    blksort::BlkSort coder;
    coder.decode(static_cast<uint32_t>(n), dst, copy.data());
    return 0;
}

} // extern "C"
