// facade_test.cpp -- drives include/cpprcoder_amd/cpprcoder.h the way the reference harness drives
// cpprcoder.h (test/main.cpp:321-344) and the way its disabled unit test does (test/main.cpp:1200-1238).
// Built and run by tests/test_gpu_facade.py on the GPU box; results go to stdout / an output file and
// are compared with the oracle there.
//
//   facade_test enc  <in> <out> <sink_capacity> <piece>   piece: 0 = one call, N = pieces of N bytes, -1 = encode(u8)
//   facade_test enct <in> <out> <sink_capacity> <piece>   as enc in pieces of N bytes; prints the sink's size after initialize()
//                                                         and after every encode() call (what a caller sees between calls)
//   facade_test dec  <in> <out> <sink_capacity> <piece>
//   facade_test senc|sdec <in> <out> <sink_capacity>       static RangeEncoder<>::encode / decode
//   facade_test blocks <in> <out> <block>                  BlockCoder round trip; out = compacted streams
//   facade_test renc|rdec <in> <out> <size> <simd>         cppans::rANS::encode(_simd) / decode(_simd), test/main.cpp:384-387
//   facade_test benc|bdec <in> <out> 0                     blksort::BlkSort::encode / decode, test/main.cpp:812-825
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <vector>

#include "cpprcoder_amd/blksort.h"
#include "cpprcoder_amd/cppans.h"
#include "cpprcoder_amd/cpprcoder.h"

static std::vector<cpprcoder::u8> slurp(const char* path)
{
    std::ifstream f(path, std::ios::binary);
    return std::vector<cpprcoder::u8>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static void dump(const char* path, const cpprcoder::u8* p, size_t n)
{
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(p), static_cast<std::streamsize>(n));
}

int main(int argc, char** argv)
{
    using namespace cpprcoder;
    if (argc < 5) return 2;
    std::vector<u8> in = slurp(argv[2]);
    if (!strcmp(argv[1], "enc")) {
        s32 cap = atoi(argv[4]);
        int piece = atoi(argv[5]);
        MemoryStream sink(cap);
        AdaptiveRangeEncoder<> enc;
        if (!enc.initialize(sink, static_cast<u32>(in.size()))) return 3;
        Result r = {Status_Success, 0};
        if (piece == 0 || in.empty()) {
            r = enc.encode(static_cast<s32>(in.size()), in.data());
        } else if (piece < 0) {
            for (size_t i = 0; i < in.size(); ++i) r = enc.encode(in[i]);
        } else {
            for (size_t at = 0; at < in.size(); at += piece) {
                size_t len = in.size() - at < static_cast<size_t>(piece) ? in.size() - at : piece;
                r = enc.encode(static_cast<s32>(len), in.data() + at);
            }
        }
        dump(argv[3], sink.get(), static_cast<size_t>(sink.size()));
        printf("%d %u %d %d\n", static_cast<int>(r.status_), r.requestSize_, sink.size(), sink.capacity());
        return 0;
    }
    if (!strcmp(argv[1], "enct")) {
        s32 cap = atoi(argv[4]);
        int piece = atoi(argv[5]);
        MemoryStream sink(cap);
        AdaptiveRangeEncoder<> enc;
        if (!enc.initialize(sink, static_cast<u32>(in.size()))) return 3;
        Result r = {Status_Success, 0};
        std::vector<s32> sizes(1, sink.size());
        if (in.empty()) {
            r = enc.encode(0, in.data());
            sizes.push_back(sink.size());
        }
        for (size_t at = 0; at < in.size();) {
            size_t len = in.size() - at < static_cast<size_t>(piece) ? in.size() - at : piece;
            r = enc.encode(static_cast<s32>(len), in.data() + at);
            at += len;
            sizes.push_back(sink.size());
            if (r.status_ == Status_Error || (r.status_ == Status_Pending && at < in.size() && r.requestSize_ != in.size() - at)) break; // the sink filled
        }
        dump(argv[3], sink.get(), static_cast<size_t>(sink.size()));
        printf("%d %u %d %d\n", static_cast<int>(r.status_), r.requestSize_, sink.size(), sink.capacity());
        for (size_t i = 0; i < sizes.size(); ++i) printf("%d ", sizes[i]);
        printf("\n");
        return 0;
    }
    if (!strcmp(argv[1], "dec")) {
        s32 cap = atoi(argv[4]);
        int piece = atoi(argv[5]);
        MemoryStream sink(cap);
        AdaptiveRangeDecoder<> dec;
        dec.initialize(sink);
        Result r = {Status_Pending, 0};
        if (piece <= 0) {
            r = dec.decode(static_cast<s32>(in.size()), in.data());
        } else {
            size_t at = 0;
            bool first = true;
            while (at < in.size()) {
                size_t len = in.size() - at < static_cast<size_t>(piece) ? in.size() - at : piece;
                if (first && len < 8) len = in.size() - at < 8 ? in.size() - at : 8;
                first = false;
                r = dec.decode(static_cast<s32>(len), in.data() + at);
                at += len;
                if (r.status_ != Status_Pending) break;
            }
        }
        dump(argv[3], sink.get(), static_cast<size_t>(sink.size()));
        printf("%d %u %d %d\n", static_cast<int>(r.status_), r.requestSize_, sink.size(), sink.capacity());
        return 0;
    }
    if (!strcmp(argv[1], "senc") || !strcmp(argv[1], "sdec")) { // static coder: test/main.cpp:270-283
        s32 cap = atoi(argv[4]);
        MemoryStream sink(cap);
        RangeEncoder<> coder;
        bool ok = !strcmp(argv[1], "senc") ? coder.encode(sink, static_cast<u32>(in.size()), in.data())
                                           : coder.decode(sink, static_cast<u32>(in.size()), in.data());
        dump(argv[3], sink.get(), static_cast<size_t>(sink.size()));
        printf("%d 0 %d %d\n", ok ? 1 : 0, sink.size(), sink.capacity());
        return 0;
    }
    if (!strcmp(argv[1], "renc")) { // the stream is the last `size` bytes of the destination (test/main.cpp:384-386)
        const bool simd = atoi(argv[5]) != 0;
        const cppans::u64 cap = atoi(argv[4]) > 0 ? static_cast<cppans::u64>(atoi(argv[4])) : cppans::rANS::calc_encoded_size(static_cast<u32>(in.size()));
        std::vector<u8> dst(static_cast<size_t>(cap));
        const u32 size = simd ? cppans::rANS::encode_simd(static_cast<u32>(cap), dst.data(), static_cast<u32>(in.size()), in.data())
                              : cppans::rANS::encode(static_cast<u32>(cap), dst.data(), static_cast<u32>(in.size()), in.data());
        dump(argv[3], dst.data() + cap - size, size);
        printf("%u %llu\n", size, static_cast<unsigned long long>(cap));
        return 0;
    }
    if (!strcmp(argv[1], "rdec")) {
        const bool simd = atoi(argv[5]) != 0;
        std::vector<u8> dst(static_cast<size_t>(atoi(argv[4])));
        const u32 ret = simd ? cppans::rANS::decode_simd(static_cast<u32>(dst.size()), dst.data(), static_cast<u32>(in.size()), in.data())
                             : cppans::rANS::decode(static_cast<u32>(dst.size()), dst.data(), static_cast<u32>(in.size()), in.data());
        dump(argv[3], dst.data(), dst.size());
        printf("%u %zu\n", ret, dst.size());
        return 0;
    }
    if (!strcmp(argv[1], "benc") || !strcmp(argv[1], "bdec")) { // run_blksort: test/main.cpp:812-825
        const bool forward = !strcmp(argv[1], "benc");
        const uint32_t size = static_cast<uint32_t>(in.size());
        // decode writes blocks * 32768 + rest bytes, blocks = size / 32770 (blksort.h:451-462)
        const uint32_t room = forward ? blksort::BlkSort::encodeBound(size) : static_cast<uint32_t>(rcx_bwt_decoded_size(size));
        std::vector<u8> dst(room);
        blksort::BlkSort blk;
        if (forward) blk.encode(size, dst.data(), in.data());
        else blk.decode(size, dst.data(), in.data());
        dump(argv[3], dst.data(), dst.size());
        printf("%d %u %u\n", blk.ok() ? 1 : 0, room, blksort::BlkSort::decodeBound(size));
        return 0;
    }
    if (!strcmp(argv[1], "blocks")) {
        u32 block = static_cast<u32>(atoi(argv[4]));
        BlockCoder coder(block);
        std::vector<u8> comp, back;
        std::vector<u64> offsets;
        if (!coder.encode(comp, offsets, in.size(), in.data())) return 4;
        if (!coder.decode(back, comp, offsets)) return 5;
        if (back != in) return 6;
        dump(argv[3], comp.data(), comp.size());
        printf("%zu %zu\n", comp.size(), offsets.size());
        return 0;
    }
    return 2;
}
