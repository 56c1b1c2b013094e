// ref_shim_ans.cpp -- C entry points around the UNMODIFIED reference header cppans.h.
//
// TEST INFRASTRUCTURE ONLY (see ref_shim.cpp).  This translation unit contains no reference code: it includes the
// reference's cppans.h from where it lies (-DRCX_REFERENCE_ANS_HEADER="/root/reference/cppans.h", oracle/Makefile)
// and is compiled into oracle/_ref/libcppans_ref.so, which is git-ignored.
//
// Build note: cppans.h does not compile with g++ as it stands -- its compiler detection tests `__gnuc__` (lower
// case, cppans.h:91), so under g++ the macro CPPANS_RESTRICT it uses at cppans.h:102 is never defined (clang++ gets
// past that and then rejects `static alignas(16) const` at cppans.h:445).  The recipe defines that one macro on the
// command line (-DCPPANS_RESTRICT=__restrict, what the header's own g++ branch would have set); no source is edited
// and nothing else is supplied.
//
// The call sequences follow the reference harness: test/main.cpp:384-387 (the stream is the LAST `size` bytes of the
// destination).  decode_simd reads up to 8 bytes past the stream (cppans.h:479-481): the shim hands it a padded copy.
#ifndef RCX_REFERENCE_ANS_HEADER
#error "compile with -DRCX_REFERENCE_ANS_HEADER=\"/path/to/cppans.h\""
#endif
#define CPPANS_IMPLEMENTATION
#include RCX_REFERENCE_ANS_HEADER

#include <cstdint>
#include <cstring>
#include <vector>

extern "C" {

// -> encoded size (0 = refused); the stream is copied to the START of `out` (capacity out_cap >= calc_encoded_size(n)).
uint32_t ref_rans_encode(const uint8_t* src, uint32_t n, uint8_t* out, uint64_t out_cap, int simd)
{
    // calc_encoded_size (cppans.h:492-495) leaves no room for the eight flushed states of encode_simd when n < 16 (the
    // reference then writes in front of its destination): 64 bytes more, which changes nothing for larger inputs
    const uint64_t cap = cppans::rANS::calc_encoded_size(n) + 64;
    if (n == 0 || out_cap + 64 < cap) return 0;
    std::vector<uint8_t> dst(cap);
    const uint32_t size = simd ? cppans::rANS::encode_simd(static_cast<uint32_t>(cap), dst.data(), n, src)
                               : cppans::rANS::encode(static_cast<uint32_t>(cap), dst.data(), n, src);
    if (size) memcpy(out, dst.data() + cap - size, size);
    return size;
}

// -> the reference's return value; dst gets dst_cap >= declared size bytes.
uint32_t ref_rans_decode(const uint8_t* comp, uint32_t comp_size, uint8_t* dst, uint32_t dst_cap, int simd)
{
    std::vector<uint8_t> padded(static_cast<size_t>(comp_size) + 64, 0);
    memcpy(padded.data(), comp, comp_size);
    return simd ? cppans::rANS::decode_simd(dst_cap, dst, comp_size, padded.data())
                : cppans::rANS::decode(dst_cap, dst, comp_size, padded.data());
}

uint64_t ref_rans_bound(uint32_t n) { return cppans::rANS::calc_encoded_size(n); }

} // extern "C"

extern "C" {

// Many independent blocks, same shape as rco_rans_encode_block_range (rans_oracle.c).
int ref_rans_encode_block_range(const uint8_t* src, uint64_t n, uint32_t block, uint64_t first, uint64_t last,
                                uint8_t* slots, uint64_t slot, uint32_t* sizes, int simd)
{
    int ok = 1;
    std::vector<uint8_t> tmp(cppans::rANS::calc_encoded_size(block) + 64);
    for (uint64_t b = first; b < last; ++b) {
        const uint64_t at = b * block;
        const uint32_t len = static_cast<uint32_t>((n - at < block) ? (n - at) : block);
        const uint32_t size = ref_rans_encode(src + at, len, tmp.data(), tmp.size(), simd);
        if (size == 0 || size > slot) {
            ok = 0;
            sizes[b] = 0;
        } else {
            memcpy(slots + b * slot, tmp.data(), size);
            sizes[b] = size;
        }
    }
    return ok;
}

int ref_rans_decode_block_range(const uint8_t* slots, uint64_t slot, const uint32_t* sizes, uint32_t block, uint64_t n,
                                uint64_t first, uint64_t last, uint8_t* dst, int simd)
{
    int ok = 1;
    for (uint64_t b = first; b < last; ++b) {
        const uint64_t at = b * block;
        const uint32_t len = static_cast<uint32_t>((n - at < block) ? (n - at) : block);
        if (ref_rans_decode(slots + b * slot, sizes[b], dst + at, len, simd) == 0) ok = 0;
    }
    return ok;
}

} // extern "C"
