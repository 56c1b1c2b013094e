// rcx_api.hip -- the C ABI of include/rcx.h on top of the gfx950 kernels.
//
// Host-side restatement of the reference's driver code path: where
// test/main.cpp:321-344 constructs a MemoryStream and a coder per buffer and
// calls initialize/encode/decode, a caller here makes one rcx_ctx per GPU and
// calls rcx_encode_blocks_device / rcx_decode_blocks_device per buffer.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/rcx.h"
#include "rcx_divtab.hpp"
#include "rcx_kernels.hpp"

namespace
{

// RCX_DEBUG=1 (diagnostic): say which HIP call failed, on stderr; the status code stays the only thing a caller gets
#define HIP_TRY(expr)                                                                                              \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess) {                                                                                    \
            if (getenv("RCX_DEBUG")) fprintf(stderr, "rcx: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return RCX_E_HIP;                                                                                      \
        }                                                                                                          \
    } while (0)

struct EventPair {
    hipEvent_t a, b;
    int what;
};

// Every entry point begins here.  Since HIP 7 an error code returned by ANY earlier runtime call of this thread -- the
// caller's, another library's -- stays in the thread's "last error" until somebody reads it, and the launches below are
// checked by reading it: what was there before is not ours to report (found by a test that ran after another one had left
// an error behind: the first kernel launch of the next call "failed").
inline hipError_t rcx_enter_device(int device)
{
    (void)hipGetLastError();
    return hipSetDevice(device);
}

struct HostPipe; // rcx_host.hpp: streams, threads' staging and bookkeeping of the host-buffer entry points

} // namespace

struct rcx_ctx {
    int device = 0;
    int lanes_per_block = 0; // decode: 0 = default (4, the quad kernel), 8 = octet, 4 = quad, 1 = one lane per block (RCX_LANES_PER_BLOCK)
    int wide_wg = -1;        // decode workgroups: -1/1 = multi-wave (default), 0 = single-wave (RCX_WIDE_WG)
    int enc_variant = 3;     // encode: 0 = one wave per 64 blocks, 1 = octet, 2 = 4-wave model/coder split, 3 = 5-wave split (RCX_ENC_VARIANT)
    int enc_lanes = 0;       // blocks per multi-wave encode workgroup: 0 = from the block count, else 1..64 (RCX_ENC_LANES)
    int dec_quads = 0;       // blocks per quad-decoder wave: 0 = from the block count, else 1, 2, 4, 8, 16 (RCX_DEC_QUADS)
    int cus = 256;           // compute units of the device
    bool rans1_lds_set = false; // rcx_enc_rans1_k has been allowed its 128 KiB of dynamic LDS
    bool rans1w_lds_set = false; // the same for rcx_enc_rans1w_k
    bool rans_track = false; // the single-stream rANS decode wants the payload bytes consumed (status[2])
    // scratch
    u8* slots = nullptr;
    u64 slots_bytes = 0;
    u32* sizes = nullptr;
    u64 sizes_count = 0;
    u32* starts = nullptr;      // rANS: where each block's stream begins in its slot (the encoders write backwards)
    u64 starts_count = 0;
    u32* models = nullptr;      // one-state rANS: every block's scaled cumulative counts + coding table (rcx_rans_model_k)
    u64 models_bytes = 0;
    u32* redo = nullptr;        // decode: blocks the quad kernel leaves to the one-lane kernel (corrupt input only)
    u64 redo_count = 0;
    u32* ties = nullptr;        // block sort: [count, (block, period) ...] of the periodic blocks of the last forward call
    u64 ties_count = 0;
    bool bwt_lds_set = false;   // the block-sort kernels have been allowed their dynamic LDS
    bool bwt_atomic = false;    // their counting passes rank with ds_add_rtn_u32 (checked on this device) instead of ballots
    DivEntry* divtab = nullptr;
    u32 divtab_block = 0;
    u32* status = nullptr;      // device: [flags, first bad block, track0, track1]
    u32* status_host = nullptr; // pinned, same 4 words
    // staging for the host-pointer entry points
    u8* h_in = nullptr;
    u64 h_in_bytes = 0;
    u8* h_out = nullptr;
    u64 h_out_bytes = 0;
    u64* h_off = nullptr;
    u64 h_off_count = 0;
    HostPipe* pipe = nullptr;   // made by the first host-buffer call that is large enough to be cut into chunks
    // timing
    bool timing = false;
    std::vector<EventPair> pending;
    std::vector<EventPair> pool;
    double ms[RCX_T_COUNT] = {};
    uint64_t launches[RCX_T_COUNT] = {};
};

namespace
{

struct Timed {
    rcx_ctx* c;
    hipStream_t s;
    EventPair p;
    bool on;
    Timed(rcx_ctx* ctx, hipStream_t st, int what) : c(ctx), s(st), on(ctx->timing)
    {
        if (!on) return;
        if (!c->pool.empty()) {
            p = c->pool.back();
            c->pool.pop_back();
        } else if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) {
            on = false;
            return;
        }
        p.what = what;
        (void)hipEventRecord(p.a, s);
    }
    ~Timed()
    {
        if (!on) return;
        (void)hipEventRecord(p.b, s);
        c->pending.push_back(p);
    }
};

int grow(void** p, u64* have, u64 want)
{
    if (*have >= want) return RCX_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    if (hipMalloc(p, want) != hipSuccess) return RCX_E_NOMEM;
    *have = want;
    return RCX_OK;
}

// Entry i serves total = 256 + i (rcx_divtab.hpp); built on the device, 16 bytes per symbol of the largest block.
__global__ void rcx_divtab_k(DivEntry* __restrict__ tab, u64 entries)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < entries) tab[i] = rcx_make_div_entry((u32)(256 + i));
}

int ensure_divtab(rcx_ctx* c, u32 block)
{
    if (c->divtab && c->divtab_block >= block) return RCX_OK;
    // round up so that a sweep of block sizes builds the table once or twice
    u32 cover = 1u << 16;
    while (cover < block) cover <<= 1;
    const u64 entries = (u64)cover + 2 * RCX_STAGE;
    if (c->divtab) (void)hipFree(c->divtab);
    c->divtab = nullptr;
    c->divtab_block = 0;
    if (hipMalloc(reinterpret_cast<void**>(&c->divtab), entries * sizeof(DivEntry)) != hipSuccess) return RCX_E_NOMEM;
    hipLaunchKernelGGL(rcx_divtab_k, dim3((u32)((entries + 255) / 256)), dim3(256), 0, nullptr, c->divtab, entries);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return RCX_E_HIP;
    c->divtab_block = cover;
    return RCX_OK;
}

bool block_ok(uint32_t block) { return block >= RCX_MIN_BLOCK && block <= RCX_MAX_BLOCK; }

// Multi-wave workgroups (waves spread over the SIMDs of one CU) or single-wave ones (more waves per CU).
bool wide_workgroups(const rcx_ctx* c, u64 nblocks)
{
    (void)nblocks;
    return c->wide_wg != 0; // default: multi-wave (RCX_WIDE_WG=0 selects single-wave workgroups)
}

// Lanes per block for the adaptive decoder.  A wave-instruction costs its SIMD 4 cycles whatever it
// serves, so fewer lanes per block means less machine-wide work: with its waves placed one per SIMD
// (multi-wave workgroups) the quad kernel beats the octet kernel at every block count measured on 1 GiB
// (4 KiB ... 256 KiB blocks, profiles/r01s_decode_variants.jsonl).  The octet and one-lane kernels stay
// selectable.
int decode_lanes(const rcx_ctx* c, u64 nblocks)
{
    (void)nblocks;
    return c->lanes_per_block ? c->lanes_per_block : 4;
}

// Launch shape.  A wave-instruction costs its SIMD the same whatever its lanes do, and a block is one serial
// chain, so with few blocks the work is spread thin rather than packed: the multi-wave encoders carry
// `lanes` blocks per workgroup such that every CU has a workgroup before any carries 64, the quad decoders
// `quads` blocks per wave such that every SIMD has a wave before any carries 16.
u32 pow2_at_least(u64 x)
{
    u32 p = 1;
    while (p < x) p <<= 1;
    return p;
}
u32 encode_lanes(const rcx_ctx* c, u64 nblocks)
{
    if (c->enc_lanes) return (u32)c->enc_lanes;
    const u32 want = pow2_at_least((nblocks + c->cus - 1) / c->cus);
    return want > RCX_LANES ? RCX_LANES : want;
}
u32 decode_quads(const rcx_ctx* c, u64 nblocks)
{
    if (c->dec_quads) return (u32)c->dec_quads;
    const u64 simds = 4ull * c->cus;
    const u32 want = pow2_at_least((nblocks + simds - 1) / simds);
    return want > RCX_QUAD_BLOCKS ? RCX_QUAD_BLOCKS : want;
}

int ensure_redo(rcx_ctx* c, u64 nblocks)
{
    u64 bytes = c->redo_count * sizeof(u32);
    const int r = grow(reinterpret_cast<void**>(&c->redo), &bytes, (nblocks + 1) * sizeof(u32));
    c->redo_count = r == RCX_OK ? bytes / sizeof(u32) : 0;
    return r;
}

// Which part of the context's per-block scratch a set of launches uses, and in which launch shape.
struct ScratchRange {
    u64 first = 0;       // blocks into slots / sizes / starts / models / redo
    bool packed = false; // full workgroups and waves whatever the block count (chunks that share the machine)
};
struct ScratchView {
    u8* slots;
    u32* sizes;
    u32* starts;
    u32* models;
    u32* redo;
};
int encode_range(rcx_ctx* c, int coder, const void* d_src, u64 n, u32 block, void* d_dst, u64 dst_cap, u64* d_offsets, hipStream_t s, ScratchRange rg);
int decode_range(rcx_ctx* c, int coder, const void* d_comp, u64 comp_size, const u64* d_offsets, u64 nblocks, u32 block, u64 n, void* d_dst,
                 hipStream_t s, ScratchRange rg);

bool is_rans(int coder) { return coder == RCX_CODER_RANS || coder == RCX_CODER_RANS8; }
bool coder_ok(int coder) { return coder == RCX_CODER_ADAPTIVE || coder == RCX_CODER_STATIC || is_rans(coder); }

int reserve(rcx_ctx* c, u64 n, u32 block, int coder = RCX_CODER_ADAPTIVE)
{
    const u64 nblocks = rcx_block_count(n, block);
    const u64 slot = rcx_block_bound_for(coder, block);
    int r = is_rans(coder) ? RCX_OK : ensure_divtab(c, block);
    if (r != RCX_OK) return r;
    r = grow(reinterpret_cast<void**>(&c->slots), &c->slots_bytes, nblocks * slot + 256);
    if (r != RCX_OK) return r;
    u64 bytes = c->sizes_count * sizeof(u32);
    r = grow(reinterpret_cast<void**>(&c->sizes), &bytes, (nblocks + 1) * sizeof(u32));
    if (r != RCX_OK) return r;
    c->sizes_count = bytes / sizeof(u32);
    if (is_rans(coder)) {
        bytes = c->starts_count * sizeof(u32);
        r = grow(reinterpret_cast<void**>(&c->starts), &bytes, (nblocks + 1) * sizeof(u32));
        if (r != RCX_OK) return r;
        c->starts_count = bytes / sizeof(u32);
        if (coder == RCX_CODER_RANS) {
            r = grow(reinterpret_cast<void**>(&c->models), &c->models_bytes, nblocks * RCX_RANS_MODEL_DW * sizeof(u32));
            if (r != RCX_OK) return r;
        }
    }
    return ensure_redo(c, nblocks);
}

} // namespace

#include "rcx_host.hpp"

extern "C" {

int rcx_version(void) { return RCX_VERSION; }

const char* rcx_status_string(int status)
{
    switch (status) {
    case RCX_OK: return "success";
    case RCX_PENDING: return "pending";
    case RCX_ERROR: return "error";
    case RCX_E_ARG: return "bad argument";
    case RCX_E_CAPACITY: return "destination too small";
    case RCX_E_CORRUPT: return "corrupt or truncated block stream";
    case RCX_E_HIP: return "HIP runtime error";
    case RCX_E_NOMEM: return "out of memory";
    case RCX_E_COMM: return "RCCL error";
    default: return "unknown status";
    }
}

uint64_t rcx_block_count(uint64_t n, uint32_t block) { return block ? (n + block - 1) / block : 0; }

// Worst case of one adaptive block: n + (255/2)*log2(n)/8 (estimator regret) + the truncation loss of
// t = range/total, which is < log2(1 + 1/t) bits per symbol: < n/64 bytes while total <= 2^20 (t >= 16), and up to
// one bit per symbol as the total approaches 2^24 (t >= 1); plus the 9 framing bytes.
// The static coder needs 521 + n + slack.  Rounded to 16 so slots keep 16-byte alignment.
uint64_t rcx_block_bound(uint32_t block)
{
    uint64_t b = (uint64_t)block + block / 32 + 2048;
    if (block > (1u << 20)) b += block / 8;
    return (b + 15) & ~(uint64_t)15;
}

uint64_t rcx_encode_bound(uint64_t n, uint32_t block) { return rcx_block_count(n, block) * rcx_block_bound(block) + 16; }

uint64_t rcx_block_bound_for(int coder, uint32_t block)
{
    if (!is_rans(coder)) return rcx_block_bound(block);
    return (2 * (uint64_t)block + 1032 + 64 + 15) & ~(uint64_t)15; // cppans.h:492-495 + 64, see rcx.h
}

uint64_t rcx_encode_bound_for(int coder, uint64_t n, uint32_t block) { return rcx_block_count(n, block) * rcx_block_bound_for(coder, block) + 16; }

int rcx_ctx_create(int device, rcx_ctx** out)
{
    if (!out) return RCX_E_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return RCX_E_HIP; // no CPU fallback: fail loudly
    if (device < 0 || device >= count) return RCX_E_ARG;
    HIP_TRY(rcx_enter_device(device));
    rcx_ctx* c = new (std::nothrow) rcx_ctx();
    if (!c) return RCX_E_NOMEM;
    c->device = device;
#if defined(RCX_WITH_VARIANTS) // (the diagnostic build with csrc/variants/: the superseded kernels can be chosen too)
    const bool variants = true;
#else
    const bool variants = false;
#endif
    if (const char* v = getenv("RCX_LANES_PER_BLOCK")) c->lanes_per_block = (atoi(v) == 1 || atoi(v) == 4 || (variants && atoi(v) == 8)) ? atoi(v) : 0;
    if (const char* v = getenv("RCX_WIDE_WG")) c->wide_wg = atoi(v) ? 1 : 0;
    if (const char* v = getenv("RCX_ENC_VARIANT")) c->enc_variant = (atoi(v) == 0 || atoi(v) == 3 || (variants && (atoi(v) == 1 || atoi(v) == 2))) ? atoi(v) : 3;
    if (const char* v = getenv("RCX_ENC_LANES")) c->enc_lanes = atoi(v) >= 1 && atoi(v) <= 64 ? atoi(v) : 0;
    if (const char* v = getenv("RCX_DEC_QUADS")) { const int q = atoi(v); c->dec_quads = (q == 1 || q == 2 || q == 4 || q == 8 || q == 16) ? q : 0; }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->cus = cus;
    }
    if (hipMalloc(reinterpret_cast<void**>(&c->status), 4 * sizeof(u32)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&c->status_host), 4 * sizeof(u32), hipHostMallocDefault) != hipSuccess) {
        rcx_ctx_destroy(c);
        return RCX_E_NOMEM;
    }
    const u32 init[2] = {0u, 0xFFFFFFFFu};
    if (hipMemcpy(c->status, init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) {
        rcx_ctx_destroy(c);
        return RCX_E_HIP;
    }
    *out = c;
    return RCX_OK;
}

void rcx_ctx_destroy(rcx_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto& p : c->pending) c->pool.push_back(p);
    for (auto& p : c->pool) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    if (c->slots) (void)hipFree(c->slots);
    if (c->sizes) (void)hipFree(c->sizes);
    if (c->starts) (void)hipFree(c->starts);
    if (c->models) (void)hipFree(c->models);
    if (c->redo) (void)hipFree(c->redo);
    if (c->ties) (void)hipFree(c->ties);
    if (c->divtab) (void)hipFree(c->divtab);
    if (c->status) (void)hipFree(c->status);
    if (c->status_host) (void)hipHostFree(c->status_host);
    if (c->h_in) (void)hipFree(c->h_in);
    if (c->h_out) (void)hipFree(c->h_out);
    if (c->h_off) (void)hipFree(c->h_off);
    host_pipe_destroy(c->pipe);
    delete c;
}

int rcx_ctx_reserve(rcx_ctx* c, uint64_t n, uint32_t block)
{
    if (!c || !block_ok(block)) return RCX_E_ARG;
    HIP_TRY(rcx_enter_device(c->device));
    return reserve(c, n, block);
}

int rcx_ctx_reserve_for(rcx_ctx* c, int coder, uint64_t n, uint32_t block)
{
    if (!c || !block_ok(block) || !coder_ok(coder)) return RCX_E_ARG;
    HIP_TRY(rcx_enter_device(c->device));
    return reserve(c, n, block, coder);
}

int rcx_ctx_sync_status(rcx_ctx* c, void* stream, uint64_t* first_bad_block)
{
    if (!c) return RCX_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(rcx_enter_device(c->device));
    HIP_TRY(hipMemcpyAsync(c->status_host, c->status, 2 * sizeof(u32), hipMemcpyDeviceToHost, s));
    const u32 init[2] = {0u, 0xFFFFFFFFu};
    HIP_TRY(hipStreamSynchronize(s));
    const u32 flags = c->status_host[0], bad = c->status_host[1];
    if (flags) {
        HIP_TRY(hipMemcpyAsync(c->status, init, sizeof(init), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    if (first_bad_block) *first_bad_block = bad;
    if (flags & RCX_ST_CORRUPT) return RCX_E_CORRUPT;
    if (flags & RCX_ST_CAPACITY) return RCX_E_CAPACITY;
    return RCX_OK;
}

int rcx_encode_blocks_device(rcx_ctx* c, int coder, const void* d_src, uint64_t n, uint32_t block,
                             void* d_dst, uint64_t dst_cap, uint64_t* d_offsets, void* stream)
{
    if (!c || !block_ok(block) || !d_offsets || (n && (!d_src || !d_dst))) return RCX_E_ARG;
    if (!coder_ok(coder)) return RCX_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(rcx_enter_device(c->device));
    const u64 nblocks = rcx_block_count(n, block);
    if (nblocks == 0) return hipMemsetAsync(d_offsets, 0, sizeof(u64), s) == hipSuccess ? RCX_OK : RCX_E_HIP;
    if (nblocks > 0x7FFFFFFFull) return RCX_E_ARG; // grid.x limit with 8 blocks per workgroup to spare
    int r = reserve(c, n, block, coder);
    if (r != RCX_OK) return r;
    return encode_range(c, coder, d_src, n, block, d_dst, dst_cap, d_offsets, s, ScratchRange{});
}

} // extern "C"

namespace
{

// The encode launches for blocks whose scratch (slots, sizes, starts, models, redo) begins `rg.first` blocks into the
// context's arrays, which the caller has reserved.  The many-block call above is the whole range; the host-buffer
// pipeline (rcx_host.hpp) runs several chunks of one buffer at once, each on its own stream and its own part of the
// scratch, `packed` = every workgroup / wave carries its full load of blocks, so that chunks share the machine.
int encode_range(rcx_ctx* c, int coder, const void* d_src, u64 n, u32 block, void* d_dst, u64 dst_cap, u64* d_offsets, hipStream_t s,
                 ScratchRange rg)
{
    const u64 nblocks = rcx_block_count(n, block);
    const u64 slot = rcx_block_bound_for(coder, block);
    ScratchView v{c->slots + rg.first * slot, c->sizes + rg.first, c->starts ? c->starts + rg.first : nullptr,
                  c->models ? c->models + rg.first * RCX_RANS_MODEL_DW : nullptr, c->redo + rg.first};
    // Static coder: with fewer than 32768 blocks (two one-wave workgroups per CU) the three-wave kernel, which
    // spreads 64 blocks over three SIMDs, is faster (157 vs 112 GB/s at 16384 blocks); with more, the one-wave
    // kernel fills the machine by itself (202 vs 157 GB/s at 32768 blocks).
    const bool static3 = coder == RCX_CODER_STATIC && c->enc_variant >= 2 && nblocks < 32768;
    {
        Timed t(c, s, RCX_T_ENCODE);
        if (is_rans(coder)) { // cppans.h: a block is an octet of lanes, four 8-block waves per workgroup
            const u64 per_wg = 4 * RCX_RANS_BLOCKS;
            const u32 grid = (u32)((nblocks + per_wg - 1) / per_wg);
            if (coder == RCX_CODER_RANS8)
                hipLaunchKernelGGL(rcx_enc_rans_k<true>, dim3(grid), dim3(256), 0, s, static_cast<const u8*>(d_src), n, block, nblocks, v.slots,
                                   slot, v.sizes, v.starts, c->status);
            else {
                // one state per block = one chain per block: the model by octets, then the coding loop one lane per
                // block, `lanes` blocks per wave so that every SIMD has a wave before any wave carries 64
                hipLaunchKernelGGL(rcx_rans_model_k<14>, dim3(grid), dim3(256), 0, s, static_cast<const u8*>(d_src), n, block, nblocks, v.models);
                // The coding loop: two waves per 64 blocks (coder, writer), 2 KiB of table per block: one workgroup per CU.
                // RCX_RANS1_WAVES=1 (diagnostic): the one-wave kernel it replaced, 16 blocks per wave (RCX_RANS1_LANES),
                // four waves per workgroup.
                const char* one = getenv("RCX_RANS1_WAVES");
                if (!(one && atoi(one) == 1)) {
                    if (!c->rans1w_lds_set) { // more than the 64 KiB a kernel gets without asking
                        if (hipFuncSetAttribute(reinterpret_cast<const void*>(rcx_enc_rans1w_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RCX_R1W_LDS_BYTES) != hipSuccess)
                            return RCX_E_HIP;
                        c->rans1w_lds_set = true;
                    }
                    hipLaunchKernelGGL(rcx_enc_rans1w_k, dim3((u32)((nblocks + 63) / 64)), dim3(128), RCX_R1W_LDS_BYTES, s, static_cast<const u8*>(d_src), n, block,
                                       nblocks, static_cast<const u32*>(v.models), v.slots, slot, v.sizes, v.starts, c->status);
                } else {
                    u32 lanes = 16;
                    if (const char* v2 = getenv("RCX_RANS1_LANES")) { const int q = atoi(v2); if (q == 1 || q == 2 || q == 4 || q == 8 || q == 16) lanes = (u32)q; }
                    const u64 per_wg1 = (u64)lanes * RCX_RANS1_ENC_WAVES;
                    const u32 grid1 = (u32)((nblocks + per_wg1 - 1) / per_wg1);
                    u32 lds_bytes = lanes * RCX_RANS1_ENC_WAVES * 2048u;
                    // RCX_RANS1_ALONE=1 (diagnostic): more LDS than two workgroups have room for, so that thin workgroups are not
                    // stacked on one CU
                    if (getenv("RCX_RANS1_ALONE") && lds_bytes < 84u * 1024u) lds_bytes = 84u * 1024u;
                    u32 lanes_shift = 0;
                    while ((1u << lanes_shift) < lanes) ++lanes_shift;
                    if (!c->rans1_lds_set) {
                        if (hipFuncSetAttribute(reinterpret_cast<const void*>(rcx_enc_rans1_k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
                            return RCX_E_HIP;
                        c->rans1_lds_set = true;
                    }
                    hipLaunchKernelGGL(rcx_enc_rans1_k, dim3(grid1), dim3(64 * RCX_RANS1_ENC_WAVES), lds_bytes, s, static_cast<const u8*>(d_src), n, block,
                                       nblocks, static_cast<const u32*>(v.models), v.slots, slot, v.sizes, v.starts, c->status, lanes_shift);
                }
            }
        } else if (static3) {
            const u32 lanes = rg.packed ? RCX_LANES : encode_lanes(c, nblocks);
            const u32 grid = (u32)((nblocks + lanes - 1) / lanes);
            hipLaunchKernelGGL(rcx_enc_static3_k, dim3(grid), dim3(RCX_ST3_THREADS), 0, s, static_cast<const u8*>(d_src), n, block,
                               nblocks, v.slots, slot, v.sizes, c->status, v.redo, lanes);
        } else if (coder == RCX_CODER_STATIC) {
            const u32 grid = (u32)((nblocks + RCX_LANES - 1) / RCX_LANES);
            hipLaunchKernelGGL(rcx_enc_static_k, dim3(grid), dim3(64), 0, s, static_cast<const u8*>(d_src), n, block, nblocks,
                               v.slots, slot, v.sizes, c->status, static_cast<const u32*>(nullptr));
        } else if (c->enc_variant == 3) {
            const u32 lanes = rg.packed ? RCX_LANES : encode_lanes(c, nblocks);
            const u32 grid = (u32)((nblocks + lanes - 1) / lanes);
            hipLaunchKernelGGL(rcx_enc_mc5_k, dim3(grid), dim3(RCX_MC5_THREADS), 0, s, static_cast<const u8*>(d_src), n, block,
                               nblocks, v.slots, slot, v.sizes, c->divtab, c->status, v.redo, lanes);
#if defined(RCX_WITH_VARIANTS)
        } else if (c->enc_variant == 2) {
            const u32 grid = (u32)((nblocks + RCX_LANES - 1) / RCX_LANES);
            hipLaunchKernelGGL(rcx_enc_mc_k, dim3(grid), dim3(RCX_MC_THREADS), 0, s, static_cast<const u8*>(d_src), n, block, nblocks,
                               v.slots, slot, v.sizes, c->divtab, c->status);
        } else if (c->enc_variant == 1) {
            const u32 grid = (u32)((nblocks + RCX_OCT_BLOCKS - 1) / RCX_OCT_BLOCKS);
            hipLaunchKernelGGL(rcx_enc_oct_k, dim3(grid), dim3(64), 0, s, static_cast<const u8*>(d_src), n, block, nblocks,
                               v.slots, slot, v.sizes, c->divtab, c->status);
#endif
        } else {
            const u32 grid = (u32)((nblocks + RCX_LANES - 1) / RCX_LANES);
            hipLaunchKernelGGL(rcx_enc_adaptive_k<false>, dim3(grid), dim3(64), 0, s, static_cast<const u8*>(d_src), n, block,
                               nblocks, v.slots, slot, v.sizes, c->divtab, c->status, 0u, static_cast<u32*>(nullptr),
                               static_cast<const u32*>(nullptr));
        }
        // (the second passes are part of the encode time: on adversarial data they are not free)
        if (static3) { // the same for the static coder
            const u32 grid = (u32)((nblocks + RCX_LANES - 1) / RCX_LANES);
            hipLaunchKernelGGL(rcx_enc_static_k, dim3(grid), dim3(64), 0, s, static_cast<const u8*>(d_src), n, block, nblocks,
                               v.slots, slot, v.sizes, c->status, static_cast<const u32*>(v.redo));
        }
        if (coder == RCX_CODER_ADAPTIVE && c->enc_variant == 3) {
            // Blocks in which a carry ran through more output bytes than the five-wave kernel keeps in LDS were
            // marked, not finished: the one-wave kernel encodes them again.  Nothing is marked on ordinary data
            // and every wave of this launch returns at once.
            const u32 grid = (u32)((nblocks + RCX_LANES - 1) / RCX_LANES);
            hipLaunchKernelGGL(rcx_enc_adaptive_k<false>, dim3(grid), dim3(64), 0, s, static_cast<const u8*>(d_src), n, block,
                               nblocks, v.slots, slot, v.sizes, c->divtab, c->status, 0u, static_cast<u32*>(nullptr),
                               static_cast<const u32*>(v.redo));
        }
    }
    {
        Timed t(c, s, RCX_T_SCAN);
        hipLaunchKernelGGL(rcx_scan_sizes_k, dim3(1), dim3(1024), 0, s, v.sizes, nblocks, d_offsets, dst_cap, c->status);
    }
    {
        Timed t(c, s, RCX_T_SCATTER);
        hipLaunchKernelGGL(rcx_scatter_k, dim3((u32)nblocks), dim3(256), 0, s, v.slots, slot, v.sizes, d_offsets,
                           static_cast<u8*>(d_dst), dst_cap, is_rans(coder) ? static_cast<const u32*>(v.starts) : static_cast<const u32*>(nullptr));
    }
    {
        const hipError_t last = hipGetLastError();
        if (last != hipSuccess && getenv("RCX_DEBUG")) fprintf(stderr, "rcx: encode launches left %s\n", hipGetErrorString(last));
        return last == hipSuccess ? RCX_OK : RCX_E_HIP;
    }
}

} // namespace

extern "C" {

int rcx_decode_blocks_device(rcx_ctx* c, int coder, const void* d_comp, uint64_t comp_size,
                             const uint64_t* d_offsets, uint64_t nblocks, uint32_t block,
                             uint64_t n, void* d_dst, void* stream)
{
    if (!c || !block_ok(block) || (nblocks && (!d_comp || !d_offsets || !d_dst))) return RCX_E_ARG;
    if (!coder_ok(coder)) return RCX_E_ARG;
    if (nblocks != rcx_block_count(n, block)) return RCX_E_ARG;
    if (nblocks == 0) return RCX_OK;
    if (nblocks > 0x7FFFFFFFull) return RCX_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(rcx_enter_device(c->device));
    if (!is_rans(coder)) {
        int r = ensure_divtab(c, block);
        if (r != RCX_OK) return r;
        if ((r = ensure_redo(c, nblocks)) != RCX_OK) return r; // no-op after rcx_ctx_reserve
    }
    return decode_range(c, coder, d_comp, comp_size, d_offsets, nblocks, block, n, d_dst, s, ScratchRange{});
}

} // extern "C"

namespace
{

// The decode launches for `nblocks` blocks whose redo marks begin `rg.first` entries into the context's array (see
// encode_range); the divisor table and the redo array are in place.
int decode_range(rcx_ctx* c, int coder, const void* d_comp, u64 comp_size, const u64* d_offsets, u64 nblocks, u32 block, u64 n, void* d_dst,
                 hipStream_t s, ScratchRange rg)
{
    u32* const redo = c->redo ? c->redo + rg.first : nullptr;
    if (is_rans(coder)) {
        Timed t(c, s, RCX_T_DECODE);
        const u64 per_wg = 4 * RCX_RANS_BLOCKS;
        const u32 grid = (u32)((nblocks + per_wg - 1) / per_wg);
        if (coder == RCX_CODER_RANS8)
            hipLaunchKernelGGL(rcx_dec_rans8_k<4>, dim3(grid), dim3(256), 0, s, static_cast<const u8*>(d_comp), (u64)comp_size, d_offsets, nblocks,
                               block, n, static_cast<u8*>(d_dst), c->status);
        else {
            const u32 quads = rg.packed ? RCX_QUAD_BLOCKS : decode_quads(c, nblocks);
            const u64 per_wg1 = (u64)quads * RCX_QUAD_DEC_WAVES;
            const u32 grid1 = (u32)((nblocks + per_wg1 - 1) / per_wg1);
            hipLaunchKernelGGL(rcx_dec_rans1_quad_k<RCX_QUAD_DEC_WAVES>, dim3(grid1), dim3(64 * RCX_QUAD_DEC_WAVES), 0, s,
                               static_cast<const u8*>(d_comp), (u64)comp_size, d_offsets, nblocks, block, n, static_cast<u8*>(d_dst), c->status,
                               quads, c->rans_track ? c->status + 2 : static_cast<u32*>(nullptr));
        }
        return hipGetLastError() == hipSuccess ? RCX_OK : RCX_E_HIP;
    }
    const bool quad = coder == RCX_CODER_ADAPTIVE && decode_lanes(c, nblocks) == 4;
    const bool squad = coder == RCX_CODER_STATIC && decode_lanes(c, nblocks) != 1;
    {
        Timed t(c, s, RCX_T_DECODE);
        if (coder == RCX_CODER_STATIC && decode_lanes(c, nblocks) != 1) {
            const u32 quads = rg.packed ? RCX_QUAD_BLOCKS : decode_quads(c, nblocks);
            const u64 per_wg = (u64)quads * RCX_QUAD_DEC_WAVES;
            const u32 grid = (u32)((nblocks + per_wg - 1) / per_wg);
            hipLaunchKernelGGL(rcx_dec_static_quad_k<RCX_QUAD_DEC_WAVES>, dim3(grid), dim3(64 * RCX_QUAD_DEC_WAVES), 0, s,
                               static_cast<const u8*>(d_comp), (u64)comp_size, d_offsets, nblocks, block, n, static_cast<u8*>(d_dst), c->status,
                               redo, quads);
        } else if (coder == RCX_CODER_STATIC) {
            const u32 grid = (u32)((nblocks + RCX_LANES - 1) / RCX_LANES);
            hipLaunchKernelGGL(rcx_dec_static_k<false>, dim3(grid), dim3(64), 0, s, static_cast<const u8*>(d_comp), (u64)comp_size, d_offsets, nblocks,
                               block, n, static_cast<u8*>(d_dst), c->status, static_cast<u32*>(nullptr),
                               static_cast<const u32*>(nullptr));
        } else if (decode_lanes(c, nblocks) == 4) {
            const u32 quads = rg.packed ? RCX_QUAD_BLOCKS : decode_quads(c, nblocks);
            if (wide_workgroups(c, nblocks)) {
                const u64 per_wg = (u64)quads * RCX_QUAD_DEC_WAVES;
                const u32 grid = (u32)((nblocks + per_wg - 1) / per_wg);
                hipLaunchKernelGGL(rcx_dec_quad_k<RCX_QUAD_DEC_WAVES>, dim3(grid), dim3(64 * RCX_QUAD_DEC_WAVES), 0, s,
                                   static_cast<const u8*>(d_comp), (u64)comp_size, d_offsets, nblocks, block, n, static_cast<u8*>(d_dst), c->divtab,
                                   c->status, redo, quads);
            } else {
                const u32 grid = (u32)((nblocks + quads - 1) / quads);
                hipLaunchKernelGGL(rcx_dec_quad_k<1>, dim3(grid), dim3(64), 0, s, static_cast<const u8*>(d_comp), (u64)comp_size, d_offsets, nblocks,
                                   block, n, static_cast<u8*>(d_dst), c->divtab, c->status, redo, quads);
            }
#if defined(RCX_WITH_VARIANTS)
        } else if (decode_lanes(c, nblocks) == 8) {
            if (wide_workgroups(c, nblocks)) {
                const u64 per_wg = RCX_OCT_BLOCKS * RCX_OCT_DEC_WAVES;
                const u32 grid = (u32)((nblocks + per_wg - 1) / per_wg);
                hipLaunchKernelGGL(rcx_dec_oct_k<RCX_OCT_DEC_WAVES>, dim3(grid), dim3(64 * RCX_OCT_DEC_WAVES), 0, s,
                                   static_cast<const u8*>(d_comp), (u64)comp_size, d_offsets, nblocks, block, n, static_cast<u8*>(d_dst), c->divtab,
                                   c->status);
            } else {
                const u32 grid = (u32)((nblocks + RCX_OCT_BLOCKS - 1) / RCX_OCT_BLOCKS);
                hipLaunchKernelGGL(rcx_dec_oct_k<1>, dim3(grid), dim3(64), 0, s, static_cast<const u8*>(d_comp), (u64)comp_size, d_offsets, nblocks,
                                   block, n, static_cast<u8*>(d_dst), c->divtab, c->status);
            }
#endif
        } else {
            const u32 grid = (u32)((nblocks + RCX_LANES - 1) / RCX_LANES);
            hipLaunchKernelGGL(rcx_dec_adaptive_k<false>, dim3(grid), dim3(64), 0, s, static_cast<const u8*>(d_comp), (u64)comp_size, d_offsets,
                               nblocks, block, n, static_cast<u8*>(d_dst), c->divtab, c->status, static_cast<u32*>(nullptr),
                               static_cast<const u32*>(nullptr));
        }
        if (quad) {
            // Blocks whose stream asked for a symbol past the table (corrupt input) were marked, not decoded, by
            // the quad kernel: the one-lane kernel, which has the reference's fall-through for that case, decodes
            // them again.  On valid input nothing is marked and every wave of this launch returns at once.
            const u32 grid = (u32)((nblocks + RCX_LANES - 1) / RCX_LANES);
            hipLaunchKernelGGL(rcx_dec_adaptive_k<false>, dim3(grid), dim3(64), 0, s, static_cast<const u8*>(d_comp), (u64)comp_size, d_offsets,
                               nblocks, block, n, static_cast<u8*>(d_dst), c->divtab, c->status, static_cast<u32*>(nullptr),
                               static_cast<const u32*>(redo));
        }
        if (squad) { // the same for the static coder: a target past the table or a symbol of count 0
            const u32 grid = (u32)((nblocks + RCX_LANES - 1) / RCX_LANES);
            hipLaunchKernelGGL(rcx_dec_static_k<false>, dim3(grid), dim3(64), 0, s, static_cast<const u8*>(d_comp), (u64)comp_size, d_offsets, nblocks,
                               block, n, static_cast<u8*>(d_dst), c->status, static_cast<u32*>(nullptr), static_cast<const u32*>(redo));
        }
    }
    return hipGetLastError() == hipSuccess ? RCX_OK : RCX_E_HIP;
}

} // namespace

extern "C" {

int rcx_encode_blocks(rcx_ctx* c, int coder, const uint8_t* src, uint64_t n, uint32_t block,
                      uint8_t* dst, uint64_t dst_cap, uint64_t* dst_size, uint64_t* offsets)
{
    if (!c || !block_ok(block) || !dst_size || (n && (!src || !dst))) return RCX_E_ARG;
    HIP_TRY(rcx_enter_device(c->device));
    *dst_size = 0;
    if (!coder_ok(coder)) return RCX_E_ARG;
    const u64 nblocks = rcx_block_count(n, block);
    if (nblocks > 0x7FFFFFFFull) return RCX_E_ARG;
    const u64 bound = rcx_encode_bound_for(coder, n, block);
    const u64 cb = host_chunk_blocks(block, false, nblocks);
    const u64 chunks = (nblocks + cb - 1) / cb;
    int r = grow(reinterpret_cast<void**>(&c->h_in), &c->h_in_bytes, n + 64);
    if (r != RCX_OK) return r;
    r = grow(reinterpret_cast<void**>(&c->h_out), &c->h_out_bytes, bound + 64);
    if (r != RCX_OK) return r;
    u64 off_bytes = c->h_off_count * sizeof(u64);
    r = grow(reinterpret_cast<void**>(&c->h_off), &off_bytes, (nblocks + chunks + 1) * sizeof(u64));
    if (r != RCX_OK) return r;
    c->h_off_count = off_bytes / sizeof(u64);
    if (chunks < 2 || getenv("RCX_HOST_SERIAL")) { // one chunk: nothing to overlap
        if (n) HIP_TRY(hipMemcpy(c->h_in, src, n, hipMemcpyHostToDevice));
        r = rcx_encode_blocks_device(c, coder, c->h_in, n, block, c->h_out, bound, c->h_off, nullptr);
        if (r != RCX_OK) return r;
        r = rcx_ctx_sync_status(c, nullptr, nullptr);
        if (r != RCX_OK) return r;
        u64 total = 0;
        HIP_TRY(hipMemcpy(&total, c->h_off + nblocks, sizeof(u64), hipMemcpyDeviceToHost));
        *dst_size = total;
        if (offsets) HIP_TRY(hipMemcpy(offsets, c->h_off, (nblocks + 1) * sizeof(u64), hipMemcpyDeviceToHost));
        if (total > dst_cap) return RCX_E_CAPACITY;
        if (total) HIP_TRY(hipMemcpy(dst, c->h_out, total, hipMemcpyDeviceToHost));
        return RCX_OK;
    }
    // Chunks of `cb` blocks.  Chunk k is encoded into its own part of the device buffer (at its first block's slot
    // offset) with its own table, which comes back with it; where it goes in `dst` is known once the chunks before it
    // are: the drainers add the sizes up as the chunks finish.
    HostPipe* p = nullptr;
    if ((r = host_pipe_get(c, &p)) != RCX_OK) return r;
    if ((r = host_pipe_words(p, nblocks + chunks + 1)) != RCX_OK) return r;
    if ((r = reserve(c, n, block, coder)) != RCX_OK) return r;
    const u64 slot = rcx_block_bound_for(coder, block);
    u64 running = 0;
    bool fits = true;
    HostJob job;
    job.chunks = chunks;
    job.in = [&](u64 k) -> HostSpan {
        const u64 at = k * cb * block;
        return HostSpan{src + at, c->h_in + at, (n - at < cb * block) ? n - at : cb * block};
    };
    job.launch = [&](u64 k, hipStream_t s) -> int {
        const u64 b0 = k * cb, nb = (nblocks - b0 < cb) ? nblocks - b0 : cb;
        const u64 at = b0 * block, len = (n - at < cb * block) ? n - at : cb * block;
        const int e = encode_range(c, coder, c->h_in + at, len, block, c->h_out + b0 * slot, nb * slot, c->h_off + b0 + k, s, ScratchRange{b0, true});
        if (e != RCX_OK) return e;
        return hipMemcpyAsync(p->words + b0 + k, c->h_off + b0 + k, (nb + 1) * sizeof(u64), hipMemcpyDeviceToHost, s) == hipSuccess ? RCX_OK : RCX_E_HIP;
    };
    job.out = [&](u64 k, HostSpan* span) -> int {
        const u64 b0 = k * cb, nb = (nblocks - b0 < cb) ? nblocks - b0 : cb;
        const u64* rel = p->words + b0 + k;
        if (offsets)
            for (u64 i = 0; i < nb; ++i) offsets[b0 + i] = running + rel[i];
        const u64 bytes = rel[nb];
        if (bytes > nb * slot || running + bytes > dst_cap) fits = false; // (a slot overflow is latched on the device as well)
        *span = fits ? HostSpan{c->h_out + b0 * slot, dst + running, bytes} : HostSpan{};
        running += bytes;
        return RCX_OK;
    };
    job.work_streams = 2; // (measured: a third encode chunk in flight gains nothing and delays the copies back)
    job.caller_in = src;
    job.caller_out = dst;
    r = host_run(c, p, job);
    const int latched = rcx_ctx_sync_status(c, nullptr, nullptr);
    if (getenv("RCX_DEBUG") && (r != RCX_OK || latched != RCX_OK)) fprintf(stderr, "rcx: host encode: pipeline %d, latched status %d\n", r, latched);
    if (r != RCX_OK) return r;
    if (latched != RCX_OK) return latched;
    *dst_size = running;
    if (offsets) offsets[nblocks] = running;
    return fits ? RCX_OK : RCX_E_CAPACITY;
}

int rcx_decode_blocks(rcx_ctx* c, int coder, const uint8_t* comp, uint64_t comp_size,
                      const uint64_t* offsets, uint64_t nblocks, uint32_t block,
                      uint8_t* dst, uint64_t dst_cap, uint64_t* dst_size)
{
    if (!c || !block_ok(block) || !dst_size || (nblocks && (!comp || !offsets || !dst))) return RCX_E_ARG;
    if (!coder_ok(coder) || nblocks > 0x7FFFFFFFull) return RCX_E_ARG;
    HIP_TRY(rcx_enter_device(c->device));
    *dst_size = 0;
    if (nblocks == 0) return RCX_OK;
    if (offsets[nblocks] > comp_size || offsets[nblocks] < offsets[nblocks - 1] || offsets[nblocks] - offsets[nblocks - 1] < 4)
        return RCX_E_CORRUPT;
    // the last block's declared size fixes n (every earlier block is full)
    const uint8_t* lastp = comp + offsets[nblocks - 1];
    const u64 last_len = (u64)lastp[0] | ((u64)lastp[1] << 8) | ((u64)lastp[2] << 16) | ((u64)lastp[3] << 24);
    if (last_len == 0 || last_len > block) return RCX_E_CORRUPT;
    const u64 n = (nblocks - 1) * (u64)block + last_len;
    if (n > dst_cap) return RCX_E_CAPACITY;
    int r = grow(reinterpret_cast<void**>(&c->h_in), &c->h_in_bytes, comp_size + 64);
    if (r != RCX_OK) return r;
    r = grow(reinterpret_cast<void**>(&c->h_out), &c->h_out_bytes, n + 64);
    if (r != RCX_OK) return r;
    u64 off_bytes = c->h_off_count * sizeof(u64);
    r = grow(reinterpret_cast<void**>(&c->h_off), &off_bytes, (nblocks + 1) * sizeof(u64));
    if (r != RCX_OK) return r;
    c->h_off_count = off_bytes / sizeof(u64);
    const u64 cb = host_chunk_blocks(block, true, nblocks);
    const u64 chunks = (nblocks + cb - 1) / cb;
    if (chunks < 2 || getenv("RCX_HOST_SERIAL")) {
        HIP_TRY(hipMemcpy(c->h_in, comp, comp_size, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->h_off, offsets, (nblocks + 1) * sizeof(u64), hipMemcpyHostToDevice));
        r = rcx_decode_blocks_device(c, coder, c->h_in, comp_size, c->h_off, nblocks, block, n, c->h_out, nullptr);
        if (r != RCX_OK) return r;
        r = rcx_ctx_sync_status(c, nullptr, nullptr);
        if (r != RCX_OK) return r;
        HIP_TRY(hipMemcpy(dst, c->h_out, n, hipMemcpyDeviceToHost));
        *dst_size = n;
        return RCX_OK;
    }
    // Chunks of `cb` blocks: a chunk's streams are one stretch of `comp` (the table says which), and they keep their
    // place in the device copy, so the table goes over once and as it is.
    for (u64 b = 0; b < nblocks; ++b)
        if (offsets[b] > offsets[b + 1]) return RCX_E_CORRUPT; // (the host has the table: a chunk is offsets[b0] .. offsets[b1] of `comp`)
    HostPipe* p = nullptr;
    if ((r = host_pipe_get(c, &p)) != RCX_OK) return r;
    if (!is_rans(coder)) {
        if ((r = ensure_divtab(c, block)) != RCX_OK) return r;
        if ((r = ensure_redo(c, nblocks)) != RCX_OK) return r;
    }
    HIP_TRY(hipMemcpy(c->h_off, offsets, (nblocks + 1) * sizeof(u64), hipMemcpyHostToDevice));
    HostJob job;
    job.chunks = chunks;
    job.in = [&](u64 k) -> HostSpan {
        const u64 b0 = k * cb, b1 = (b0 + cb < nblocks) ? b0 + cb : nblocks;
        return HostSpan{comp + offsets[b0], c->h_in + offsets[b0], offsets[b1] - offsets[b0]};
    };
    job.launch = [&](u64 k, hipStream_t s) -> int {
        const u64 b0 = k * cb, nb = (nblocks - b0 < cb) ? nblocks - b0 : cb;
        const u64 at = b0 * block, len = (n - at < cb * block) ? n - at : cb * block;
        return decode_range(c, coder, c->h_in, comp_size, c->h_off + b0, nb, block, len, c->h_out + at, s, ScratchRange{b0, true});
    };
    job.out = [&](u64 k, HostSpan* span) -> int {
        const u64 at = k * cb * block;
        *span = HostSpan{c->h_out + at, dst + at, (n - at < cb * block) ? n - at : cb * block};
        return RCX_OK;
    };
    job.decode = true;
    job.caller_in = comp;
    job.caller_out = dst;
    r = host_run(c, p, job);
    const int latched = rcx_ctx_sync_status(c, nullptr, nullptr);
    if (r != RCX_OK) return r;
    if (latched != RCX_OK) return latched;
    *dst_size = n;
    return RCX_OK;
}


// ---------------------------------------------------------------------------
// Single streams with the reference's sink semantics (one block, lane 0 of one wave).
// ---------------------------------------------------------------------------

int rcx_stream_encode(rcx_ctx* c, int coder, const uint8_t* src, uint32_t n,
                      uint8_t* dst, uint64_t dst_cap, uint64_t sink_capacity, uint64_t* dst_size, uint32_t* request_size)
{
    if (!c || !dst || !dst_size || (n && !src)) return RCX_E_ARG;
    if (!coder_ok(coder) || n > RCX_MAX_STREAM) return RCX_E_ARG;
    HIP_TRY(rcx_enter_device(c->device));
    *dst_size = 0;
    if (request_size) *request_size = 0;
    if (is_rans(coder)) {
        // rANS::encode / encode_simd (cppans.h:497-530, :567-607): one block; RCX_ERROR where the reference returns 0
        // (it asserts 0 < src_size; its destination cannot be larger than u32 either)
        if (n == 0 || n > RCX_MAX_RANS_STREAM) return RCX_ERROR;
        const u32 rblock = n < RCX_MIN_BLOCK ? RCX_MIN_BLOCK : n;
        int rr = reserve(c, rblock, rblock, coder);
        if (rr != RCX_OK) return rr;
        rr = grow(reinterpret_cast<void**>(&c->h_in), &c->h_in_bytes, (u64)n + 64);
        if (rr != RCX_OK) return rr;
        HIP_TRY(hipMemcpy(c->h_in, src, n, hipMemcpyHostToDevice));
        const u64 rslot = rcx_block_bound_for(coder, rblock);
        if (coder == RCX_CODER_RANS8)
            hipLaunchKernelGGL(rcx_enc_rans_k<true>, dim3(1), dim3(256), 0, nullptr, c->h_in, (u64)n, rblock, (u64)1, c->slots, rslot, c->sizes,
                               c->starts, c->status);
        else
            hipLaunchKernelGGL(rcx_enc_rans_k<false>, dim3(1), dim3(256), 0, nullptr, c->h_in, (u64)n, rblock, (u64)1, c->slots, rslot, c->sizes,
                               c->starts, c->status);
        if (hipGetLastError() != hipSuccess) return RCX_E_HIP;
        rr = rcx_ctx_sync_status(c, nullptr, nullptr);
        if (rr != RCX_OK) return rr;
        u32 rsize = 0, rstart = 0;
        HIP_TRY(hipMemcpy(&rsize, c->sizes, sizeof(u32), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&rstart, c->starts, sizeof(u32), hipMemcpyDeviceToHost));
        *dst_size = rsize;
        if (rsize > sink_capacity || rsize > dst_cap) return RCX_E_CAPACITY;
        HIP_TRY(hipMemcpy(dst, c->slots + rstart, rsize, hipMemcpyDeviceToHost));
        return RCX_OK;
    }
    const u32 block = n < RCX_MIN_BLOCK ? RCX_MIN_BLOCK : n;
    const bool longer = n > RCX_MAX_BLOCK; // past the table halving of cpprcoder.h:1138: the lane divides by its own total
    const u64 slot = rcx_block_bound(block);
    int r = longer ? RCX_OK : ensure_divtab(c, block);
    if (r != RCX_OK) return r;
    r = grow(reinterpret_cast<void**>(&c->slots), &c->slots_bytes, slot + 256);
    if (r != RCX_OK) return r;
    u64 sizes_bytes = c->sizes_count * sizeof(u32);
    r = grow(reinterpret_cast<void**>(&c->sizes), &sizes_bytes, 2 * sizeof(u32));
    if (r != RCX_OK) return r;
    c->sizes_count = sizes_bytes / sizeof(u32);
    r = grow(reinterpret_cast<void**>(&c->h_in), &c->h_in_bytes, (u64)n + 64);
    if (r != RCX_OK) return r;
    if (n) HIP_TRY(hipMemcpy(c->h_in, src, n, hipMemcpyHostToDevice));
    if (coder == RCX_CODER_STATIC) {
        // RangeEncoder<T>::encode (cpprcoder.h:375-458) returns a bool; the caller (the facade) replays the
        // sink calls itself, so the whole stream is handed back: RCX_OK, or RCX_E_CAPACITY if dst is too small.
        if (c->enc_variant >= 2 && n >= RCX_MIN_BLOCK && n <= RCX_MAX_BLOCK) {
            // one chain runs faster through the three-wave encoder (table lookups / arithmetic / writer on three SIMDs)
            // than on a lone lane; the one-wave kernel behind it takes over if a carry outran the rings
            if ((r = ensure_redo(c, 1)) != RCX_OK) return r;
            hipLaunchKernelGGL(rcx_enc_static3_k, dim3(1), dim3(RCX_ST3_THREADS), 0, nullptr, c->h_in, (u64)n, block, (u64)1, c->slots, slot,
                               c->sizes, c->status, c->redo, 1u);
            hipLaunchKernelGGL(rcx_enc_static_k, dim3(1), dim3(64), 0, nullptr, c->h_in, (u64)n, block, (u64)1, c->slots, slot, c->sizes,
                               c->status, static_cast<const u32*>(c->redo));
        } else {
            hipLaunchKernelGGL(rcx_enc_static_k, dim3(1), dim3(64), 0, nullptr, c->h_in, (u64)n, block, (u64)1, c->slots, slot,
                               c->sizes, c->status, static_cast<const u32*>(nullptr));
        }
        if (hipGetLastError() != hipSuccess) return RCX_E_HIP;
        r = rcx_ctx_sync_status(c, nullptr, nullptr);
        if (r != RCX_OK) return r;
        u32 ssize = 0;
        HIP_TRY(hipMemcpy(&ssize, c->sizes, sizeof(u32), hipMemcpyDeviceToHost));
        *dst_size = ssize;
        if (ssize > sink_capacity || ssize > dst_cap) return RCX_E_CAPACITY;
        HIP_TRY(hipMemcpy(dst, c->slots, ssize, hipMemcpyDeviceToHost));
        return RCX_OK;
    }
    if (longer) {
        hipLaunchKernelGGL((rcx_enc_adaptive_k<false, true>), dim3(1), dim3(64), 0, nullptr, c->h_in, (u64)n, block, (u64)1, c->slots, slot,
                           c->sizes, c->divtab, c->status, 0u, static_cast<u32*>(nullptr), static_cast<const u32*>(nullptr));
    } else if (c->enc_variant == 3 && n >= RCX_MIN_BLOCK) {
        // One stream is one chain, and the five-wave encoder runs a chain about four times as fast as a lone lane does
        // (its model, arithmetic and writer are five instruction streams on four SIMDs): one block, one lane in use;
        // the one-wave kernel behind it takes over if a carry outran the rings (see rcx_encode_blocks_device).
        if ((r = ensure_redo(c, 1)) != RCX_OK) return r;
        hipLaunchKernelGGL(rcx_enc_mc5_k, dim3(1), dim3(RCX_MC5_THREADS), 0, nullptr, c->h_in, (u64)n, block, (u64)1, c->slots, slot, c->sizes,
                           c->divtab, c->status, c->redo, 1u);
        hipLaunchKernelGGL((rcx_enc_adaptive_k<false, false>), dim3(1), dim3(64), 0, nullptr, c->h_in, (u64)n, block, (u64)1, c->slots, slot,
                           c->sizes, c->divtab, c->status, 0u, static_cast<u32*>(nullptr), static_cast<const u32*>(c->redo));
    } else {
        hipLaunchKernelGGL((rcx_enc_adaptive_k<false, false>), dim3(1), dim3(64), 0, nullptr, c->h_in, (u64)n, block, (u64)1, c->slots, slot,
                           c->sizes, c->divtab, c->status, 0u, static_cast<u32*>(nullptr), static_cast<const u32*>(nullptr));
    }
    if (hipGetLastError() != hipSuccess) return RCX_E_HIP;
    r = rcx_ctx_sync_status(c, nullptr, nullptr);
    if (r != RCX_OK) return r;
    u32 size = 0;
    HIP_TRY(hipMemcpy(&size, c->sizes, sizeof(u32), hipMemcpyDeviceToHost));
    const u64 cap16 = sink_capacity < 4 ? 4 : sink_capacity; // the header went through the growing write()
    if ((u64)size - 4 <= cap16) { // every writeByte fits; the final write(4) grows the sink (cpprcoder.h:1031-1045)
        *dst_size = size;
        if (size > dst_cap) return RCX_E_CAPACITY;
        HIP_TRY(hipMemcpy(dst, c->slots, size, hipMemcpyDeviceToHost));
        return RCX_OK;
    }
    // The sink fills.  Second pass: replay the reference's delayed writer to find the symbol.
    *dst_size = cap16;
    if (cap16 > dst_cap) return RCX_E_CAPACITY;
    if (longer)
        hipLaunchKernelGGL((rcx_enc_adaptive_k<true, true>), dim3(1), dim3(64), 0, nullptr, c->h_in, (u64)n, block, (u64)1, c->slots, slot,
                           c->sizes, c->divtab, c->status, (u32)(cap16 - 4), c->status + 2, static_cast<const u32*>(nullptr));
    else
        hipLaunchKernelGGL((rcx_enc_adaptive_k<true, false>), dim3(1), dim3(64), 0, nullptr, c->h_in, (u64)n, block, (u64)1, c->slots, slot,
                           c->sizes, c->divtab, c->status, (u32)(cap16 - 4), c->status + 2, static_cast<const u32*>(nullptr));
    if (hipGetLastError() != hipSuccess) return RCX_E_HIP;
    HIP_TRY(hipMemcpy(c->status_host, c->status, 4 * sizeof(u32), hipMemcpyDeviceToHost));
    const u32 fail_at = c->status_host[2];
    HIP_TRY(hipMemcpy(dst, c->slots, cap16, hipMemcpyDeviceToHost)); // what was written before the sink filled
    if (fail_at != 0xFFFFFFFFu) { // cpprcoder.h:708-711
        if (request_size) *request_size = n - fail_at;
        return RCX_PENDING;
    }
    return RCX_OK; // only finish() failed and encode() ignores that (cpprcoder.h:716)
}

int rcx_stream_decode(rcx_ctx* c, int coder, const uint8_t* comp, uint64_t comp_size,
                      uint8_t* dst, uint64_t sink_capacity, uint64_t* dst_size, uint32_t* request_size)
{
    if (!c || !dst || !dst_size || (comp_size && !comp)) return RCX_E_ARG;
    if (!coder_ok(coder) || comp_size > 0xFFFFFFFFull) return RCX_E_ARG;
    HIP_TRY(rcx_enter_device(c->device));
    *dst_size = 0;
    if (request_size) *request_size = 0;
    if (is_rans(coder)) {
        // rANS::decode / decode_simd (cppans.h:532-564, :609-649): RCX_ERROR where the reference returns 0 (or would
        // leave its arrays: a header that is not a scaled cumulative table, a payload that runs out)
        if (comp_size < 1032 + 4) return RCX_ERROR;
        const u32 declared = (u32)comp[0] | ((u32)comp[1] << 8) | ((u32)comp[2] << 16) | ((u32)comp[3] << 24);
        if (declared > sink_capacity || declared == 0 || declared > RCX_MAX_RANS_STREAM) return RCX_ERROR; // :541, :618
        const u32 rblock = declared < RCX_MIN_BLOCK ? RCX_MIN_BLOCK : declared;
        int rr = grow(reinterpret_cast<void**>(&c->h_in), &c->h_in_bytes, comp_size + 64);
        if (rr != RCX_OK) return rr;
        rr = grow(reinterpret_cast<void**>(&c->h_out), &c->h_out_bytes, (u64)declared + 64);
        if (rr != RCX_OK) return rr;
        u64 roff_bytes = c->h_off_count * sizeof(u64);
        rr = grow(reinterpret_cast<void**>(&c->h_off), &roff_bytes, 2 * sizeof(u64));
        if (rr != RCX_OK) return rr;
        c->h_off_count = roff_bytes / sizeof(u64);
        const u64 roffs[2] = {0, comp_size};
        HIP_TRY(hipMemcpy(c->h_in, comp, comp_size, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->h_off, roffs, sizeof(roffs), hipMemcpyHostToDevice));
        c->rans_track = true;
        rr = rcx_decode_blocks_device(c, coder, c->h_in, comp_size, c->h_off, 1, rblock, declared, c->h_out, nullptr);
        c->rans_track = false;
        if (rr != RCX_OK) return rr;
        rr = rcx_ctx_sync_status(c, nullptr, nullptr);
        if (rr == RCX_E_CORRUPT) return RCX_ERROR;
        if (rr != RCX_OK) return rr;
        HIP_TRY(hipMemcpy(dst, c->h_out, declared, hipMemcpyDeviceToHost));
        *dst_size = declared;
        if (request_size) { // what the reference returns: payload bytes consumed (decode) / the symbol count (decode_simd)
            *request_size = declared;
            if (coder == RCX_CODER_RANS) {
                HIP_TRY(hipMemcpy(c->status_host, c->status, 4 * sizeof(u32), hipMemcpyDeviceToHost));
                *request_size = c->status_host[2];
            }
        }
        return RCX_OK;
    }
    if (coder == RCX_CODER_STATIC) {
        // RangeEncoder<T>::decode (cpprcoder.h:460-519): bool.  RCX_OK = true; RCX_ERROR = false, with the
        // symbols written before the failure in dst; a full sink is the caller's to notice (it replays writeByte).
        if (comp_size < 516) return RCX_ERROR;                       // :468-476
        const u32 declared = (u32)comp[0] | ((u32)comp[1] << 8) | ((u32)comp[2] << 16) | ((u32)comp[3] << 24);
        if (declared == 0) return RCX_OK;                            // :481-483
        if (comp_size < 516 + 1 || comp_size - 516 < 5) return RCX_ERROR; // :486-493
        const u64 count = declared < sink_capacity ? declared : sink_capacity;
        if (count > RCX_MAX_STREAM) return RCX_E_ARG;
        if (count == 0) return RCX_OK; // nothing fits: the caller's first writeByte fails
        const u32 block = count < RCX_MIN_BLOCK ? RCX_MIN_BLOCK : (u32)count;
        int r = grow(reinterpret_cast<void**>(&c->h_in), &c->h_in_bytes, comp_size + 64);
        if (r != RCX_OK) return r;
        r = grow(reinterpret_cast<void**>(&c->h_out), &c->h_out_bytes, count + 64);
        if (r != RCX_OK) return r;
        u64 off_bytes = c->h_off_count * sizeof(u64);
        r = grow(reinterpret_cast<void**>(&c->h_off), &off_bytes, 2 * sizeof(u64));
        if (r != RCX_OK) return r;
        c->h_off_count = off_bytes / sizeof(u64);
        const u64 offs[2] = {0, comp_size};
        HIP_TRY(hipMemcpy(c->h_in, comp, comp_size, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->h_off, offs, sizeof(offs), hipMemcpyHostToDevice));
        if (count == declared && count >= RCX_MIN_BLOCK && count <= RCX_MAX_BLOCK && decode_lanes(c, 1) != 1) {
            // the whole stream is wanted and the sink has room: the 4-lane decoder first; it only knows complete, valid
            // streams, and anything else it reports is redone by the exact one-lane decoder below
            if ((r = ensure_redo(c, 1)) != RCX_OK) return r;
            hipLaunchKernelGGL(rcx_dec_static_quad_k<RCX_QUAD_DEC_WAVES>, dim3(1), dim3(64 * RCX_QUAD_DEC_WAVES), 0, nullptr, c->h_in, (u64)comp_size,
                               c->h_off, (u64)1, block, count, c->h_out, c->status, c->redo, 1u);
            if (hipGetLastError() != hipSuccess) return RCX_E_HIP;
            u32 marked = 0;
            const int fast = rcx_ctx_sync_status(c, nullptr, nullptr); // (clears the latch)
            HIP_TRY(hipMemcpy(&marked, c->redo, sizeof(u32), hipMemcpyDeviceToHost));
            if (fast == RCX_OK && marked == 0) {
                HIP_TRY(hipMemcpy(dst, c->h_out, count, hipMemcpyDeviceToHost));
                *dst_size = count;
                return RCX_OK;
            }
        }
        hipLaunchKernelGGL(rcx_dec_static_k<true>, dim3(1), dim3(64), 0, nullptr, c->h_in, (u64)comp_size, c->h_off, (u64)1, block, count, c->h_out,
                           c->status, c->status + 2, static_cast<const u32*>(nullptr));
        if (hipGetLastError() != hipSuccess) return RCX_E_HIP;
        HIP_TRY(hipMemcpy(c->status_host, c->status, 4 * sizeof(u32), hipMemcpyDeviceToHost));
        const u32 short_at = c->status_host[2];
        const u64 produced = short_at < count ? short_at : count;
        if (produced) HIP_TRY(hipMemcpy(dst, c->h_out, produced, hipMemcpyDeviceToHost));
        *dst_size = produced;
        return short_at < count ? RCX_ERROR : RCX_OK;
    }
    if (comp_size < 8) { // cpprcoder.h:878-880
        if (request_size) *request_size = 8;
        return RCX_PENDING;
    }
    const u32 declared = (u32)comp[0] | ((u32)comp[1] << 8) | ((u32)comp[2] << 16) | ((u32)comp[3] << 24);
    const u64 cap16 = sink_capacity;
    const u64 want = declared ? declared : 1; // cpprcoder.h:912: the size test comes after the first writeByte
    const u64 count = want < cap16 ? want : cap16;
    if (count > RCX_MAX_STREAM) return RCX_E_ARG;
    if (count == 0) { // a sink that accepts nothing: the first writeByte fails (cpprcoder.h:909-911)
        if (request_size) *request_size = declared;
        return RCX_PENDING;
    }
    const u32 block = count < RCX_MIN_BLOCK ? RCX_MIN_BLOCK : (u32)count;
    const bool longer = count > RCX_MAX_BLOCK;
    int r = longer ? RCX_OK : ensure_divtab(c, block);
    if (r != RCX_OK) return r;
    r = grow(reinterpret_cast<void**>(&c->h_in), &c->h_in_bytes, comp_size + 64);
    if (r != RCX_OK) return r;
    r = grow(reinterpret_cast<void**>(&c->h_out), &c->h_out_bytes, count + 64);
    if (r != RCX_OK) return r;
    u64 off_bytes = c->h_off_count * sizeof(u64);
    r = grow(reinterpret_cast<void**>(&c->h_off), &off_bytes, 2 * sizeof(u64));
    if (r != RCX_OK) return r;
    c->h_off_count = off_bytes / sizeof(u64);
    const u64 offs[2] = {0, comp_size};
    HIP_TRY(hipMemcpy(c->h_in, comp, comp_size, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->h_off, offs, sizeof(offs), hipMemcpyHostToDevice));
    if (!longer && count == declared && count >= RCX_MIN_BLOCK && decode_lanes(c, 1) == 4) {
        // The whole stream is wanted and the sink has room: the 4-lane decoder runs the chain about three times as
        // fast as a lone lane.  It only knows complete, valid streams; anything else (input that runs dry: Pending,
        // cpprcoder.h:901-903; a target past the table) it reports, and the exact one-lane decoder below redoes it.
        if ((r = ensure_redo(c, 1)) != RCX_OK) return r;
        hipLaunchKernelGGL(rcx_dec_quad_k<RCX_QUAD_DEC_WAVES>, dim3(1), dim3(64 * RCX_QUAD_DEC_WAVES), 0, nullptr, c->h_in, (u64)comp_size, c->h_off,
                           (u64)1, block, count, c->h_out, c->divtab, c->status, c->redo, 1u);
        if (hipGetLastError() != hipSuccess) return RCX_E_HIP;
        u32 marked = 0;
        const int fast = rcx_ctx_sync_status(c, nullptr, nullptr); // (clears the latch)
        HIP_TRY(hipMemcpy(&marked, c->redo, sizeof(u32), hipMemcpyDeviceToHost));
        if (fast == RCX_OK && marked == 0) {
            HIP_TRY(hipMemcpy(dst, c->h_out, count, hipMemcpyDeviceToHost));
            *dst_size = count;
            return RCX_OK;
        }
    }
    if (longer)
        hipLaunchKernelGGL((rcx_dec_adaptive_k<true, true>), dim3(1), dim3(64), 0, nullptr, c->h_in, (u64)comp_size, c->h_off, (u64)1, block, count, c->h_out,
                           c->divtab, c->status, c->status + 2, static_cast<const u32*>(nullptr));
    else
        hipLaunchKernelGGL((rcx_dec_adaptive_k<true, false>), dim3(1), dim3(64), 0, nullptr, c->h_in, (u64)comp_size, c->h_off, (u64)1, block, count, c->h_out,
                           c->divtab, c->status, c->status + 2, static_cast<const u32*>(nullptr));
    if (hipGetLastError() != hipSuccess) return RCX_E_HIP;
    HIP_TRY(hipMemcpy(c->status_host, c->status, 4 * sizeof(u32), hipMemcpyDeviceToHost));
    const u32 short_at = c->status_host[2];
    u64 produced = count;
    int result = RCX_OK;
    if (short_at != 0xFFFFFFFFu && short_at < count) { // input ran dry first (cpprcoder.h:901-903)
        produced = short_at;
        result = RCX_PENDING;
    } else if (want > cap16) { // sink full (cpprcoder.h:909-911)
        result = RCX_PENDING;
    }
    if (result == RCX_PENDING && request_size) *request_size = declared - (u32)produced;
    if (produced) HIP_TRY(hipMemcpy(dst, c->h_out, produced, hipMemcpyDeviceToHost));
    *dst_size = produced;
    return result;
}

// ---------------------------------------------------------------------------
// The resumable single-stream decoder: AdaptiveRangeDecoder<T>::decode fed piece by piece (cpprcoder.h:872-924).
// ---------------------------------------------------------------------------
struct rcx_dstream {
    rcx_ctx* ctx = nullptr;
    RcxDState* state = nullptr; // device
    u8* in = nullptr;           // device: the bytes accepted so far that the decoder has not read yet (+ what it read since the last growth)
    u64 in_base = 0;            // where in the stream in[0] is
    u64 in_bytes = 0, in_cap = 0;
    u64 consumed = 0;           // how far the decoder has read (RcxDState::consumed)
    u8* out = nullptr;          // device: one launch's symbols
    u32* result = nullptr;      // device: {made, finished, declared, produced, consumed lo, consumed hi}
    u32* result_host = nullptr; // pinned
    bool finished = false;
    u32 declared = 0, produced = 0;
};
#define RCX_DSTREAM_CHUNK (1u << 20) /* symbols per launch */

int rcx_dstream_create(rcx_ctx* c, rcx_dstream** out)
{
    if (!c || !out) return RCX_E_ARG;
    *out = nullptr;
    HIP_TRY(rcx_enter_device(c->device));
    rcx_dstream* d = new (std::nothrow) rcx_dstream();
    if (!d) return RCX_E_NOMEM;
    d->ctx = c;
    if (hipMalloc(reinterpret_cast<void**>(&d->state), sizeof(RcxDState)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&d->out), RCX_DSTREAM_CHUNK) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&d->result), 6 * sizeof(u32)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&d->result_host), 6 * sizeof(u32), hipHostMallocDefault) != hipSuccess ||
        hipMemset(d->state, 0, sizeof(RcxDState)) != hipSuccess) {
        rcx_dstream_destroy(d);
        return RCX_E_NOMEM;
    }
    *out = d;
    return RCX_OK;
}

void rcx_dstream_destroy(rcx_dstream* d)
{
    if (!d) return;
    if (d->ctx) (void)hipSetDevice(d->ctx->device);
    if (d->state) (void)hipFree(d->state);
    if (d->in) (void)hipFree(d->in);
    if (d->out) (void)hipFree(d->out);
    if (d->result) (void)hipFree(d->result);
    if (d->result_host) (void)hipHostFree(d->result_host);
    delete d;
}

int rcx_dstream_decode(rcx_dstream* d, const uint8_t* bytes, uint64_t size, uint8_t* dst, uint64_t dst_cap,
                       uint64_t* produced_now, uint32_t* request_size)
{
    if (!d || !produced_now || (size && !bytes) || (dst_cap && !dst)) return RCX_E_ARG;
    *produced_now = 0;
    if (request_size) *request_size = 0;
    if (d->finished) return RCX_OK;
    HIP_TRY(rcx_enter_device(d->ctx->device));
    if (d->in_bytes == 0 && size < 8) { // cpprcoder.h:877-880: State_Init wants its 8 bytes in one call and keeps nothing
        if (request_size) *request_size = 8;
        return RCX_PENDING;
    }
    if (size) { // append
        if (d->in_bytes + size > d->in_cap) {
            // More room: the new buffer takes only what the decoder has not read yet (it reads all it can, so that is a few
            // bytes unless dst filled up first) -- the stream's past is dropped, and the decoder's memory stays at about twice
            // the largest piece it was ever fed, not the size of the stream.
            const u64 keep_from = d->consumed > d->in_base ? d->consumed - d->in_base : 0;
            const u64 keep = d->in_bytes - keep_from;
            u64 cap = d->in_cap ? d->in_cap : (1u << 16);
            while (cap < keep + size) cap *= 2;
            u8* bigger = nullptr;
            if (hipMalloc(reinterpret_cast<void**>(&bigger), cap) != hipSuccess) return RCX_E_NOMEM;
            if (keep && hipMemcpy(bigger, d->in + keep_from, keep, hipMemcpyDeviceToDevice) != hipSuccess) {
                (void)hipFree(bigger);
                return RCX_E_HIP;
            }
            if (d->in) (void)hipFree(d->in);
            d->in = bigger;
            d->in_cap = cap;
            d->in_base += keep_from;
            d->in_bytes = keep;
        }
        HIP_TRY(hipMemcpy(d->in + d->in_bytes, bytes, size, hipMemcpyHostToDevice));
        d->in_bytes += size;
    }
    u64 made_total = 0;
    for (;;) {
        const u64 room64 = dst_cap - made_total;
        const u32 room = room64 > RCX_DSTREAM_CHUNK ? RCX_DSTREAM_CHUNK : (u32)room64;
        // (the kernel counts from the start of the stream: in[0] is byte in_base of it, and it never looks before what it has read)
        hipLaunchKernelGGL(rcx_dec_resume_k, dim3(1), dim3(64), 0, nullptr, d->state, d->in - d->in_base, d->in_base + d->in_bytes, d->out, room, d->result);
        if (hipGetLastError() != hipSuccess) return RCX_E_HIP;
        HIP_TRY(hipMemcpy(d->result_host, d->result, 6 * sizeof(u32), hipMemcpyDeviceToHost));
        d->consumed = (u64)d->result_host[4] | ((u64)d->result_host[5] << 32);
        const u32 made = d->result_host[0];
        d->finished = d->result_host[1] != 0;
        d->declared = d->result_host[2];
        d->produced = d->result_host[3];
        if (made) HIP_TRY(hipMemcpy(dst + made_total, d->out, made, hipMemcpyDeviceToHost));
        made_total += made;
        if (d->finished || made < room || made_total == dst_cap) break; // done, input dry, or dst full
    }
    *produced_now = made_total;
    if (d->finished) return RCX_OK;
    if (request_size) *request_size = d->declared - d->produced; // cpprcoder.h:901-903 / :909-911
    return RCX_PENDING;
}

// ---------------------------------------------------------------------------
// The resumable single-stream encoder: AdaptiveRangeEncoder<T>::encode fed piece by piece (cpprcoder.h:697-720).
// ---------------------------------------------------------------------------
struct rcx_estream {
    rcx_ctx* ctx = nullptr;
    RcxEState* state = nullptr;   // device
    RcxEState* backup = nullptr;  // device: the state before the last call (rcx_estream_rewind)
    u8* slot = nullptr;           // device: the stream so far
    u64 slot_bytes = 0;
    u8* tail_backup = nullptr;    // device: the bytes of `slot` the last call could change
    u64 tail_cap = 0, tail_from = 0, tail_bytes = 0;
    u8* in = nullptr;             // device: the piece being fed
    u64 in_cap = 0;
    u32* result = nullptr;        // device
    u32* result_host = nullptr;   // pinned
    u32 declared = 0, consumed = 0;
    u64 written = 0;              // payload bytes handed on so far (the reference's writeByte count)
    u32 backup_consumed = 0;
    u64 backup_written = 0;
    bool have_backup = false, dead = false, finished = false;
    u64 pos_hint = 0;             // payload bytes in memory after the last call (how far a call can have changed things)
};

int rcx_estream_create(rcx_ctx* c, uint32_t declared, rcx_estream** out)
{
    if (!c || !out || declared > RCX_MAX_STREAM) return RCX_E_ARG;
    *out = nullptr;
    HIP_TRY(rcx_enter_device(c->device));
    rcx_estream* e = new (std::nothrow) rcx_estream();
    if (!e) return RCX_E_NOMEM;
    e->ctx = c;
    e->declared = declared;
    const u32 block = declared < RCX_MIN_BLOCK ? RCX_MIN_BLOCK : declared;
    e->slot_bytes = declared <= RCX_MAX_BLOCK ? rcx_block_bound(block) : (((u64)declared + declared / 8 + 4096 + 15) & ~(u64)15);
    RcxEState zero;
    memset(&zero, 0, sizeof(zero));
    zero.declared = declared;
    if (hipMalloc(reinterpret_cast<void**>(&e->state), sizeof(RcxEState)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&e->backup), sizeof(RcxEState)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&e->slot), e->slot_bytes + 64) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&e->result), 8 * sizeof(u32)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&e->result_host), 8 * sizeof(u32), hipHostMallocDefault) != hipSuccess ||
        hipMemcpy(e->state, &zero, sizeof(zero), hipMemcpyHostToDevice) != hipSuccess) {
        rcx_estream_destroy(e);
        return RCX_E_NOMEM;
    }
    *out = e;
    return RCX_OK;
}

void rcx_estream_destroy(rcx_estream* e)
{
    if (!e) return;
    if (e->ctx) (void)hipSetDevice(e->ctx->device);
    if (e->state) (void)hipFree(e->state);
    if (e->backup) (void)hipFree(e->backup);
    if (e->slot) (void)hipFree(e->slot);
    if (e->tail_backup) (void)hipFree(e->tail_backup);
    if (e->in) (void)hipFree(e->in);
    if (e->result) (void)hipFree(e->result);
    if (e->result_host) (void)hipHostFree(e->result_host);
    delete e;
}

int rcx_estream_encode(rcx_estream* e, const uint8_t* bytes, uint64_t size, uint8_t* dst, uint64_t dst_cap, uint64_t sink_room,
                       uint64_t* emitted_now, uint32_t* tail_bytes, uint32_t* request_size)
{
    if (!e || !emitted_now || (size && !bytes) || (dst_cap && !dst)) return RCX_E_ARG;
    *emitted_now = 0;
    if (tail_bytes) *tail_bytes = 0;
    if (request_size) *request_size = 0;
    if (e->finished) return RCX_OK;
    if (e->dead) { // the reference's coder is of no use after a full sink either (cpprcoder.h:708-711)
        if (request_size) *request_size = e->declared - e->consumed;
        return RCX_PENDING;
    }
    if (size > (u64)(e->declared - e->consumed)) return RCX_E_ARG; // CPPRCODER_ASSERT, cpprcoder.h:700
    HIP_TRY(rcx_enter_device(e->ctx->device));
    // what this call may change, kept for rcx_estream_rewind: the state, and the stream from the first byte the reference
    // has not written yet (a carry stops there) to a little past what is in memory
    {
        const u64 from = 4 + e->written, upto = 4 + e->pos_hint + 16 < e->slot_bytes ? 4 + e->pos_hint + 16 : e->slot_bytes;
        const u64 span = upto > from ? upto - from : 0;
        if (span > e->tail_cap) {
            if (e->tail_backup) (void)hipFree(e->tail_backup);
            e->tail_backup = nullptr;
            e->tail_cap = 0;
            u64 cap = 4096;
            while (cap < span) cap *= 2;
            if (hipMalloc(reinterpret_cast<void**>(&e->tail_backup), cap) != hipSuccess) return RCX_E_NOMEM;
            e->tail_cap = cap;
        }
        HIP_TRY(hipMemcpy(e->backup, e->state, sizeof(RcxEState), hipMemcpyDeviceToDevice));
        if (span) HIP_TRY(hipMemcpy(e->tail_backup, e->slot + from, span, hipMemcpyDeviceToDevice));
        e->tail_from = from;
        e->tail_bytes = span;
        e->backup_consumed = e->consumed;
        e->backup_written = e->written;
        e->have_backup = true;
    }
    if (size > e->in_cap) {
        if (e->in) (void)hipFree(e->in);
        e->in = nullptr;
        e->in_cap = 0;
        u64 cap = 1u << 16;
        while (cap < size) cap *= 2;
        if (hipMalloc(reinterpret_cast<void**>(&e->in), cap) != hipSuccess) return RCX_E_NOMEM;
        e->in_cap = cap;
    }
    if (size) HIP_TRY(hipMemcpy(e->in, bytes, size, hipMemcpyHostToDevice));
    const u32 room = sink_room > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)sink_room;
    hipLaunchKernelGGL(rcx_enc_resume_k, dim3(1), dim3(64), 0, nullptr, e->state, e->in, (u32)size, e->slot, (u32)(e->slot_bytes > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : e->slot_bytes), room,
                       e->result);
    if (hipGetLastError() != hipSuccess) return RCX_E_HIP;
    HIP_TRY(hipMemcpy(e->result_host, e->result, 8 * sizeof(u32), hipMemcpyDeviceToHost));
    const u32 written = e->result_host[0], fail_at = e->result_host[1], finished = e->result_host[2], stream_size = e->result_host[3],
              flush_fails = e->result_host[4], overflow = e->result_host[5];
    if (overflow) return RCX_E_CAPACITY; // (the slot is the bound of a stream of this size: cannot happen)
    u64 now = (u64)written - e->written; // payload bytes the reference passed to writeByte during this call
    int status = RCX_PENDING;
    u32 tail = 0;
    if (fail_at != 0xFFFFFFFFu) { // its sink filled inside symbol fail_at (cpprcoder.h:708-711)
        // it writes byte by byte until writeByte fails, so the sink is exactly full: also the part of the last group (held
        // byte + pending run) that still fitted, which the kernel's count of whole groups does not include
        now = sink_room;
        e->consumed = fail_at;
        e->dead = true;
    } else if (finished) { // cpprcoder.h:744-762: the held byte and the pending run through writeByte, low through write(4)
        const u64 through_write_byte = (u64)stream_size - 8 - e->written;
        if (flush_fails) {
            now = through_write_byte < sink_room ? through_write_byte : sink_room; // finish() gave up; encode() says Success (cpprcoder.h:716)
        } else {
            now = through_write_byte;
            tail = 4;
        }
        e->consumed = e->declared;
        e->finished = true;
        status = RCX_OK;
    } else {
        e->consumed += (u32)size;
    }
    *emitted_now = now + tail;
    if (now + tail > dst_cap) return RCX_E_CAPACITY;
    if (now) HIP_TRY(hipMemcpy(dst, e->slot + 4 + e->written, now, hipMemcpyDeviceToHost));
    if (tail) HIP_TRY(hipMemcpy(dst + now, e->slot + stream_size - 4, 4, hipMemcpyDeviceToHost));
    e->written += now;
    e->pos_hint = e->result_host[6]; // how far the stream reaches in memory: what the next call can change ends a little past it
    if (tail_bytes) *tail_bytes = tail;
    if (status == RCX_PENDING && request_size) *request_size = e->declared - e->consumed;
    return status;
}

// Back to before the last rcx_estream_encode call.  For a sink that only tells by failing how much room it has: encode with
// no limit, hand the bytes on, and if the sink fails after k of them rewind and encode the same piece with sink_room = k to
// learn which symbol the reference was coding then.
int rcx_estream_rewind(rcx_estream* e)
{
    if (!e || !e->have_backup) return RCX_E_ARG;
    HIP_TRY(rcx_enter_device(e->ctx->device));
    HIP_TRY(hipMemcpy(e->state, e->backup, sizeof(RcxEState), hipMemcpyDeviceToDevice));
    if (e->tail_bytes) HIP_TRY(hipMemcpy(e->slot + e->tail_from, e->tail_backup, e->tail_bytes, hipMemcpyDeviceToDevice));
    e->consumed = e->backup_consumed;
    e->written = e->backup_written;
    e->dead = false;
    e->finished = false;
    e->have_backup = false;
    return RCX_OK;
}

#if defined(RCX_STAMP_DEC)
int rcx_debug_dec_stamps(unsigned long long* out8)
{
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(rcx_dec_stamp_out), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
#if defined(RCX_BWT_STAMP)
int rcx_debug_bwt_stamps(unsigned long long* out16, int reset)
{
    static const unsigned long long zero[16] = {};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(rcx_bwt_stamp_out), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(rcx_bwt_stamp_out), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#endif
#if defined(RCX_STAMP)
int rcx_debug_stamps(unsigned long long* out16)
{
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(rcx_stamp_out), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

int rcx_ctx_last_redo(rcx_ctx* c, uint64_t nblocks, uint64_t* count)
{
    if (!c || !count) return RCX_E_ARG;
    *count = 0;
    if (nblocks == 0 || !c->redo) return RCX_OK;
    if (nblocks > c->redo_count) return RCX_E_ARG;
    HIP_TRY(rcx_enter_device(c->device));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<u32> host(nblocks);
    HIP_TRY(hipMemcpy(host.data(), c->redo, nblocks * sizeof(u32), hipMemcpyDeviceToHost));
    for (u64 i = 0; i < nblocks; ++i) *count += host[i] != 0;
    return RCX_OK;
}

// ---------------------------------------------------------------------------
// Block sort (blksort.h): whole 32 KiB blocks are transformed, what is left over is copied (blksort.h:440-462).
// ---------------------------------------------------------------------------

uint64_t rcx_bwt_encode_bound(uint64_t n)
{
    const u64 blocks = n / RCX_BWT_BLOCK;
    return blocks * RCX_BWT_ENCODED + (n - blocks * RCX_BWT_BLOCK);
}

uint64_t rcx_bwt_decode_bound(uint64_t n)
{
    const u64 blocks = n / RCX_BWT_BLOCK;
    return blocks * RCX_BWT_BLOCK + (n - blocks * RCX_BWT_BLOCK);
}

uint64_t rcx_bwt_decoded_size(uint64_t n)
{
    const u64 blocks = n / RCX_BWT_ENCODED;
    return blocks * RCX_BWT_BLOCK + (n - blocks * RCX_BWT_ENCODED);
}

int rcx_bwt_reserve(rcx_ctx* c, uint64_t n)
{
    if (!c) return RCX_E_ARG;
    HIP_TRY(rcx_enter_device(c->device));
    if (!c->bwt_lds_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&rcx_bwt_fwd_k<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RCX_BWT_FWD_LDS));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&rcx_bwt_fwd_k<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RCX_BWT_FWD_LDS));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&rcx_bwt_inv_k<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RCX_BWT_INV_LDS));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&rcx_bwt_inv_k<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RCX_BWT_INV_LDS));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&rcx_bwt_tie_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RCX_BWT_TIE_LDS));
        // The counting passes rank the keys of a batch with ballots (documented behaviour only).  RCX_BWT_MATCH=atomic asks
        // for one ds_add_rtn_u32 per key instead (5-25 % faster), which is only a stable rank if the LDS serves the lanes
        // of one instruction in ascending lane order -- the ISA manual does not say so, so it is opt-in, and even then only
        // taken if a short check on this device (rcx_bwt_lds_order_k, once per device and process) finds it to hold.
        const char* want = getenv("RCX_BWT_MATCH");
        c->bwt_atomic = false;
        if (want && !strcmp(want, "atomic")) {
            // (one answer per device and process: 0.3 ms the first time; a benign race if two threads ask at once)
            static int known[64]; // 0 = not asked, 1 = lane order holds, 2 = it does not
            int& answer = known[c->device & 63];
            if (answer == 0) {
                u32 bad = 1;
                u64 room = c->ties_count * sizeof(u32);
                const int rr = grow(reinterpret_cast<void**>(&c->ties), &room, 8 * sizeof(u32));
                if (rr != RCX_OK) return rr;
                c->ties_count = room / sizeof(u32);
                HIP_TRY(hipMemset(c->ties, 0, sizeof(u32)));
                hipLaunchKernelGGL(rcx_bwt_lds_order_k, dim3(4), dim3(1024), 0, nullptr, 512u, c->ties);
                HIP_TRY(hipMemcpy(&bad, c->ties, sizeof(u32), hipMemcpyDeviceToHost));
                answer = bad == 0 ? 1 : 2;
            }
            c->bwt_atomic = answer == 1;
        }
        c->bwt_lds_set = true;
    }
    const u64 blocks = n / RCX_BWT_BLOCK;
    u64 bytes = c->ties_count * sizeof(u32);
    const int r = grow(reinterpret_cast<void**>(&c->ties), &bytes, (RCX_BWT_TIES_HEAD + 2 * blocks + 2) * sizeof(u32));
    c->ties_count = r == RCX_OK ? bytes / sizeof(u32) : 0;
    return r;
}

int rcx_bwt_encode_device(rcx_ctx* c, const void* d_src, uint64_t n, void* d_dst, uint64_t dst_cap, void* stream)
{
    if (!c || (n && (!d_src || !d_dst))) return RCX_E_ARG;
    if (dst_cap < rcx_bwt_encode_bound(n)) return RCX_E_CAPACITY;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int r = rcx_bwt_reserve(c, n);
    if (r != RCX_OK) return r;
    const u64 blocks = n / RCX_BWT_BLOCK;
    const u8* src = static_cast<const u8*>(d_src);
    u8* dst = static_cast<u8*>(d_dst);
    if (blocks >> 32) return RCX_E_ARG; // (the block counters are 32 bits: 128 TiB)
    if (c->pipe) c->pipe->bwt_ties_valid = false;
    HIP_TRY(hipMemsetAsync(c->ties, 0, 2 * sizeof(u32), s)); // the tie count and the forward kernel's block counter
    if (blocks) {
        Timed t(c, s, RCX_T_BWT_FORWARD);
        const u32 grid = (u32)(blocks < (u64)c->cus ? blocks : (u64)c->cus); // one workgroup per CU, blocks off a counter
        if (c->bwt_atomic) hipLaunchKernelGGL(rcx_bwt_fwd_k<true>, dim3(grid), dim3(RCX_BWT_THREADS), RCX_BWT_FWD_LDS, s, src, blocks, dst, c->ties, c->status);
        else hipLaunchKernelGGL(rcx_bwt_fwd_k<false>, dim3(grid), dim3(RCX_BWT_THREADS), RCX_BWT_FWD_LDS, s, src, blocks, dst, c->ties, c->status);
        // periodic blocks (rotations that tie) get the row index the reference's sort would leave; usually none
        const u64 most = 2ull * (u64)c->cus;
        hipLaunchKernelGGL(rcx_bwt_tie_k, dim3((u32)(blocks < most ? blocks : most)), dim3(64), RCX_BWT_TIE_LDS, s, src, dst,
                           static_cast<const u32*>(c->ties), c->status);
    }
    const u64 rest = n - blocks * RCX_BWT_BLOCK;
    if (rest) HIP_TRY(hipMemcpyAsync(dst + blocks * RCX_BWT_ENCODED, src + blocks * RCX_BWT_BLOCK, rest, hipMemcpyDeviceToDevice, s));
    return hipGetLastError() == hipSuccess ? RCX_OK : RCX_E_HIP;
}

int rcx_bwt_decode_device(rcx_ctx* c, const void* d_src, uint64_t n, void* d_dst, uint64_t dst_cap, void* stream)
{
    if (!c || (n && (!d_src || !d_dst))) return RCX_E_ARG;
    if (dst_cap < rcx_bwt_decoded_size(n)) return RCX_E_CAPACITY;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int r = rcx_bwt_reserve(c, 0);
    if (r != RCX_OK) return r;
    const u64 blocks = n / RCX_BWT_ENCODED;
    if (blocks >> 32) return RCX_E_ARG;
    HIP_TRY(hipMemsetAsync(c->ties + 2, 0, sizeof(u32), s)); // the inverse kernel's block counter
    const u8* src = static_cast<const u8*>(d_src);
    u8* dst = static_cast<u8*>(d_dst);
    if (blocks) {
        Timed t(c, s, RCX_T_BWT_INVERSE);
        const u32 grid = (u32)(blocks < (u64)c->cus ? blocks : (u64)c->cus);
        if (c->bwt_atomic) hipLaunchKernelGGL(rcx_bwt_inv_k<true>, dim3(grid), dim3(RCX_BWT_THREADS), RCX_BWT_INV_LDS, s, src, blocks, dst, c->ties + 2, c->status);
        else hipLaunchKernelGGL(rcx_bwt_inv_k<false>, dim3(grid), dim3(RCX_BWT_THREADS), RCX_BWT_INV_LDS, s, src, blocks, dst, c->ties + 2, c->status);
    }
    const u64 rest = n - blocks * RCX_BWT_ENCODED;
    if (rest) HIP_TRY(hipMemcpyAsync(dst + blocks * RCX_BWT_BLOCK, src + blocks * RCX_BWT_ENCODED, rest, hipMemcpyDeviceToDevice, s));
    return hipGetLastError() == hipSuccess ? RCX_OK : RCX_E_HIP;
}

namespace
{
int bwt_host(rcx_ctx* c, bool forward, const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t dst_cap, uint64_t* dst_size)
{
    if (!c || !dst_size || (n && (!src || !dst))) return RCX_E_ARG;
    HIP_TRY(rcx_enter_device(c->device));
    const u64 out = forward ? rcx_bwt_encode_bound(n) : rcx_bwt_decoded_size(n);
    *dst_size = out;
    if (out > dst_cap) return RCX_E_CAPACITY;
    if (n == 0) return RCX_OK;
    int r = grow(reinterpret_cast<void**>(&c->h_in), &c->h_in_bytes, n + 64);
    if (r != RCX_OK) return r;
    r = grow(reinterpret_cast<void**>(&c->h_out), &c->h_out_bytes, out + 64);
    if (r != RCX_OK) return r;
    // Chunks of 1024 whole blocks (32 MiB); what is left over after the last whole block travels with the last chunk.
    // These kernels take blocks off a counter with one workgroup per CU, so a chunk's kernels take the chunk's share of
    // the time and follow each other on ONE stream (they also share the context's list of periodic blocks).
    const u64 unit_in = forward ? RCX_BWT_BLOCK : RCX_BWT_ENCODED, unit_out = forward ? RCX_BWT_ENCODED : RCX_BWT_BLOCK;
    const u64 blocks = n / unit_in, cb = 1024;
    const u64 chunks = (blocks + cb - 1) / cb;
    if (c->pipe) c->pipe->bwt_ties_valid = false;
    if (chunks < 2 || getenv("RCX_HOST_SERIAL")) {
        HIP_TRY(hipMemcpy(c->h_in, src, n, hipMemcpyHostToDevice));
        r = forward ? rcx_bwt_encode_device(c, c->h_in, n, c->h_out, out, nullptr) : rcx_bwt_decode_device(c, c->h_in, n, c->h_out, out, nullptr);
        if (r != RCX_OK) return r;
        r = rcx_ctx_sync_status(c, nullptr, nullptr);
        if (r != RCX_OK) return r;
        HIP_TRY(hipMemcpy(dst, c->h_out, out, hipMemcpyDeviceToHost));
        return RCX_OK;
    }
    HostPipe* p = nullptr;
    if ((r = host_pipe_get(c, &p)) != RCX_OK) return r;
    if ((r = host_pipe_words(p, chunks + 1)) != RCX_OK) return r;
    if ((r = rcx_bwt_reserve(c, forward ? n : 0)) != RCX_OK) return r;
    auto in_bytes = [&](u64 k) { return k + 1 < chunks ? cb * unit_in : n - k * cb * unit_in; };
    auto out_bytes = [&](u64 k) { return k + 1 < chunks ? cb * unit_out : out - k * cb * unit_out; };
    HostJob job;
    job.chunks = chunks;
    job.work_streams = 1;
    job.in = [&](u64 k) -> HostSpan { return HostSpan{src + k * cb * unit_in, c->h_in + k * cb * unit_in, in_bytes(k)}; };
    job.launch = [&](u64 k, hipStream_t s) -> int {
        const u8* from = c->h_in + k * cb * unit_in;
        u8* to = c->h_out + k * cb * unit_out;
        const int e = forward ? rcx_bwt_encode_device(c, from, in_bytes(k), to, out_bytes(k), s) : rcx_bwt_decode_device(c, from, in_bytes(k), to, out_bytes(k), s);
        if (e != RCX_OK || !forward) return e;
        // how many of the chunk's blocks were periodic (rcx_bwt_last_ties adds the chunks up)
        p->words[k] = 0;
        return hipMemcpyAsync(p->words + k, c->ties, sizeof(u32), hipMemcpyDeviceToHost, s) == hipSuccess ? RCX_OK : RCX_E_HIP;
    };
    job.out = [&](u64 k, HostSpan* span) -> int {
        *span = HostSpan{c->h_out + k * cb * unit_out, dst + k * cb * unit_out, out_bytes(k)};
        return RCX_OK;
    };
    job.caller_in = src;
    job.caller_out = dst;
    r = host_run(c, p, job);
    const int latched = rcx_ctx_sync_status(c, nullptr, nullptr);
    if (r != RCX_OK) return r;
    if (latched != RCX_OK) return latched;
    if (forward) {
        p->bwt_ties = 0;
        for (u64 k = 0; k < chunks; ++k) p->bwt_ties += p->words[k];
        p->bwt_ties_valid = true;
    }
    return RCX_OK;
}
} // namespace

int rcx_bwt_encode(rcx_ctx* c, const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t dst_cap, uint64_t* dst_size)
{
    return bwt_host(c, true, src, n, dst, dst_cap, dst_size);
}

int rcx_bwt_decode(rcx_ctx* c, const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t dst_cap, uint64_t* dst_size)
{
    return bwt_host(c, false, src, n, dst, dst_cap, dst_size);
}

int rcx_bwt_last_ties(rcx_ctx* c, uint64_t* count)
{
    if (!c || !count) return RCX_E_ARG;
    *count = 0;
    if (!c->ties) return RCX_OK;
    if (c->pipe && c->pipe->bwt_ties_valid) { // the last forward call was a host-buffer call made in chunks
        *count = c->pipe->bwt_ties;
        return RCX_OK;
    }
    HIP_TRY(rcx_enter_device(c->device));
    HIP_TRY(hipDeviceSynchronize());
    u32 v = 0;
    HIP_TRY(hipMemcpy(&v, c->ties, sizeof(u32), hipMemcpyDeviceToHost));
    *count = v;
    return RCX_OK;
}

int rcx_ctx_set_timing(rcx_ctx* c, int enabled)
{
    if (!c) return RCX_E_ARG;
    c->timing = enabled != 0;
    return RCX_OK;
}

int rcx_ctx_get_timing(rcx_ctx* c, double* ms, uint64_t* launches, int reset)
{
    if (!c) return RCX_E_ARG;
    HIP_TRY(rcx_enter_device(c->device));
    for (auto& p : c->pending) {
        float t = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) {
            c->ms[p.what] += t;
            c->launches[p.what] += 1;
        }
        c->pool.push_back(p);
    }
    c->pending.clear();
    for (int i = 0; i < RCX_T_COUNT; ++i) {
        if (ms) ms[i] = c->ms[i];
        if (launches) launches[i] = c->launches[i];
        if (reset) {
            c->ms[i] = 0;
            c->launches[i] = 0;
        }
    }
    return RCX_OK;
}

} // extern "C"
