// ref_shim.cpp -- C entry points around the UNMODIFIED reference header.
//
// TEST INFRASTRUCTURE ONLY.  This translation unit contains no reference
// code: it includes the reference's cpprcoder.h from where it lies
// (-DRCX_REFERENCE_HEADER="/root/reference/cpprcoder.h", see oracle/Makefile)
// and is compiled into oracle/_ref/libcpprcoder_ref.so, which is git-ignored.
// It exists so that (a) the restatement in rc_oracle.c can be validated
// against the real thing, (b) tests/golden/make_golden.py can produce golden
// vectors from the real thing, and (c) bench.py can time the reference's own
// CPU path ("cpu_baseline.kind": "reference") on the GPU box's host cores.
//
// The call sequences follow the reference harness: test/main.cpp:321-344
// (adaptive) and test/main.cpp:270-283 (static).
#ifndef RCX_REFERENCE_HEADER
#error "compile with -DRCX_REFERENCE_HEADER=\"/path/to/cpprcoder.h\""
#endif
#define CPPRCODER_IMPLEMENTATION
#include RCX_REFERENCE_HEADER

#include <cstdint>
#include <cstring>
#include <new>

using namespace cpprcoder;

namespace
{
void copy_out(const MemoryStream& s, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size)
{
    if (out_size) *out_size = static_cast<uint64_t>(s.size());
    if (dst && 0 < s.size()) {
        uint64_t n = static_cast<uint64_t>(s.size()) < dst_cap ? static_cast<uint64_t>(s.size()) : dst_cap;
        memcpy(dst, s.get(), n);
    }
}
} // namespace

extern "C" {

struct ref_result
{
    int32_t status;
    uint32_t request_size;
};

// One-shot adaptive encode into MemoryStream(dst_cap).
ref_result ref_adaptive_encode(const uint8_t* src, uint32_t n, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size)
{
    MemoryStream s(static_cast<s32>(dst_cap));
    AdaptiveRangeEncoder<> enc;
    ref_result r = {Status_Error, 0};
    if (enc.initialize(s, n)) {
        Result q = enc.encode(static_cast<s32>(n), src);
        r.status = q.status_;
        r.request_size = q.requestSize_;
    }
    copy_out(s, dst, dst_cap, out_size);
    return r;
}

// Same bytes fed in pieces of `piece` (piece == 0: one byte at a time through encode(u8)).
ref_result ref_adaptive_encode_chunked(const uint8_t* src, uint32_t n, uint32_t piece, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size)
{
    MemoryStream s(static_cast<s32>(dst_cap));
    AdaptiveRangeEncoder<> enc;
    ref_result r = {Status_Error, 0};
    if (enc.initialize(s, n)) {
        Result q = {Status_Success, 0};
        if (n == 0) q = enc.encode(0, src);
        for (uint32_t at = 0; at < n;) {
            if (piece == 0) {
                q = enc.encode(src[at]);
                at += 1;
            } else {
                uint32_t len = (n - at < piece) ? (n - at) : piece;
                q = enc.encode(static_cast<s32>(len), src + at);
                at += len;
            }
            if (q.status_ == Status_Error) break;
        }
        r.status = q.status_;
        r.request_size = q.requestSize_;
    }
    copy_out(s, dst, dst_cap, out_size);
    return r;
}

// The same, and the sink's size after initialize() and after every encode() call (see rco_adaptive_encode_trace).
uint32_t ref_adaptive_encode_trace(const uint8_t* src, uint32_t n, uint32_t piece, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size,
                                   uint32_t* sizes, uint32_t max_sizes, ref_result* last)
{
    MemoryStream s(static_cast<s32>(dst_cap));
    AdaptiveRangeEncoder<> enc;
    ref_result r = {Status_Error, 0};
    uint32_t calls = 0;
    if (enc.initialize(s, n)) {
        Result q = {Status_Success, 0};
        if (calls < max_sizes) sizes[calls++] = static_cast<uint32_t>(s.size());
        if (n == 0) {
            q = enc.encode(0, src);
            if (calls < max_sizes) sizes[calls++] = static_cast<uint32_t>(s.size());
        }
        for (uint32_t at = 0; at < n;) {
            uint32_t len = (n - at < piece) ? (n - at) : piece;
            q = enc.encode(static_cast<s32>(len), src + at);
            at += len;
            if (calls < max_sizes) sizes[calls++] = static_cast<uint32_t>(s.size());
            if (q.status_ == Status_Error || (q.status_ == Status_Pending && at < n && q.requestSize_ != n - at)) break; // the sink filled
        }
        r.status = q.status_;
        r.request_size = q.requestSize_;
    }
    if (last) *last = r;
    copy_out(s, dst, dst_cap, out_size);
    return calls;
}

ref_result ref_adaptive_decode(const uint8_t* comp, uint64_t comp_size, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size)
{
    MemoryStream s(static_cast<s32>(dst_cap));
    AdaptiveRangeDecoder<> dec;
    dec.initialize(s);
    Result q = dec.decode(static_cast<s32>(comp_size), comp);
    copy_out(s, dst, dst_cap, out_size);
    ref_result r = {q.status_, q.requestSize_};
    return r;
}

// Decode fed in pieces (first piece is at least 8 bytes, as State_Init demands).
ref_result ref_adaptive_decode_chunked(const uint8_t* comp, uint64_t comp_size, uint32_t piece, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size)
{
    MemoryStream s(static_cast<s32>(dst_cap));
    AdaptiveRangeDecoder<> dec;
    dec.initialize(s);
    Result q = {Status_Pending, 0};
    uint64_t at = 0;
    bool first = true;
    while (at < comp_size) {
        uint64_t len = comp_size - at < piece ? comp_size - at : piece;
        if (first && len < 8) len = comp_size - at < 8 ? comp_size - at : 8;
        first = false;
        q = dec.decode(static_cast<s32>(len), comp + at);
        at += len;
        if (q.status_ != Status_Pending) break;
    }
    copy_out(s, dst, dst_cap, out_size);
    ref_result r = {q.status_, q.requestSize_};
    return r;
}

int ref_static_encode(const uint8_t* src, uint32_t n, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size)
{
    MemoryStream s(static_cast<s32>(dst_cap));
    RangeEncoder<> enc;
    bool ok = enc.encode(s, n, src);
    copy_out(s, dst, dst_cap, out_size);
    return ok ? 1 : 0;
}

int ref_static_decode(const uint8_t* comp, uint32_t comp_size, uint8_t* dst, uint64_t dst_cap, uint64_t* out_size)
{
    MemoryStream s(static_cast<s32>(dst_cap));
    RangeEncoder<> enc;
    bool ok = enc.decode(s, comp_size, comp);
    copy_out(s, dst, dst_cap, out_size);
    return ok ? 1 : 0;
}

// Fresh coder + fresh stream per block, as test/main.cpp does per file.
int ref_encode_block_range(const uint8_t* src, uint64_t n, uint32_t block, uint64_t first, uint64_t last,
                           uint8_t* slots, uint64_t slot, uint32_t* sizes, int coder)
{
    int ok = 1;
    MemoryStream s(static_cast<s32>(slot));
    for (uint64_t b = first; b < last; ++b) {
        uint64_t at = b * block;
        uint32_t len = static_cast<uint32_t>((n - at < block) ? (n - at) : block);
        s.resize(0);
        if (coder == 0) {
            AdaptiveRangeEncoder<> enc;
            if (!enc.initialize(s, len)) ok = 0;
            else if (enc.encode(static_cast<s32>(len), src + at).status_ != Status_Success) ok = 0;
        } else {
            RangeEncoder<> enc;
            if (!enc.encode(s, len, src + at)) ok = 0;
        }
        if (static_cast<uint64_t>(s.size()) > slot) ok = 0;
        else memcpy(slots + b * slot, s.get(), static_cast<size_t>(s.size()));
        sizes[b] = static_cast<uint32_t>(s.size());
    }
    return ok;
}

int ref_decode_block_range(const uint8_t* slots, uint64_t slot, const uint32_t* sizes, uint32_t block, uint64_t n,
                           uint64_t first, uint64_t last, uint8_t* dst, int coder)
{
    int ok = 1;
    MemoryStream s(static_cast<s32>(block));
    for (uint64_t b = first; b < last; ++b) {
        uint64_t at = b * block;
        uint32_t len = static_cast<uint32_t>((n - at < block) ? (n - at) : block);
        s.resize(0);
        if (coder == 0) {
            AdaptiveRangeDecoder<> dec;
            dec.initialize(s);
            if (dec.decode(static_cast<s32>(sizes[b]), slots + b * slot).status_ != Status_Success) ok = 0;
        } else {
            RangeEncoder<> enc;
            if (!enc.decode(s, sizes[b], slots + b * slot)) ok = 0;
        }
        uint32_t got = static_cast<uint32_t>(s.size()) < len ? static_cast<uint32_t>(s.size()) : len;
        memcpy(dst + at, s.get(), got);
        if (static_cast<uint32_t>(s.size()) != len) ok = 0;
    }
    return ok;
}

// Model probe: feed `n` symbols through update(); report total, the 256
// counts, cumulative(c) for every c, and find() for each target.
void ref_model_probe(const uint8_t* syms, uint64_t n, uint32_t* total, uint32_t* freq256, uint32_t* cum256,
                     const uint32_t* targets, uint32_t ntargets, uint32_t* found_count, uint8_t* found_code)
{
    AdaptiveFrequencyTable t;
    t.initialize();
    for (uint64_t i = 0; i < n; ++i) t.update(syms[i]);
    *total = t.total();
    for (u32 c = 0; c < 256; ++c) {
        freq256[c] = t[c];
        cum256[c] = t.cumulative(static_cast<u8>(c));
    }
    for (uint32_t k = 0; k < ntargets; ++k) {
        u32 count = 0;
        u8 code = 0;
        t.find(count, code, targets[k]);
        found_count[k] = count;
        found_code[k] = code;
    }
}

// Sink probe: a little op script against MemoryStream; after every op the
// triple (return value, capacity, size) is appended to out.
//   0: default ctor   1: ctor(arg)   2: write(arg bytes of pattern)   3: writeByte(arg)
//   4: reserve(arg)   5: resize(arg)
// The stream is (re)constructed by op 0 / op 1, which must come first.
int ref_stream_script(const int32_t* ops, int nops, int32_t* out)
{
    alignas(MemoryStream) unsigned char store[sizeof(MemoryStream)];
    MemoryStream* s = nullptr;
    uint8_t pattern[1 << 16];
    for (int i = 0; i < (1 << 16); ++i) pattern[i] = static_cast<uint8_t>(i * 7 + 1);
    int w = 0;
    for (int i = 0; i < nops; ++i) {
        int32_t op = ops[2 * i], arg = ops[2 * i + 1], ret = 0;
        switch (op) {
        case 0:
            if (s) s->~MemoryStream();
            s = new (store) MemoryStream();
            break;
        case 1:
            if (s) s->~MemoryStream();
            s = new (store) MemoryStream(arg);
            break;
        case 2: ret = s->write(arg, pattern); break;
        case 3: ret = s->writeByte(static_cast<u8>(arg)) ? 1 : 0; break;
        case 4: s->reserve(arg); break;
        case 5: s->resize(arg); break;
        default: return -1;
        }
        out[w++] = ret;
        out[w++] = s->capacity();
        out[w++] = s->size();
    }
    if (s) s->~MemoryStream();
    return w;
}

} // extern "C"
