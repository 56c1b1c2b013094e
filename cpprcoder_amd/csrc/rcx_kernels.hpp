// rcx_kernels.hpp -- gfx950 kernels of the many-block adaptive range coder.
//
//   rcx_enc_adaptive_k   pass 1: one lane per block, 64 blocks per wave, one wave per
//                        workgroup; every block's stream goes to its scratch slot and
//                        its size to sizes[].      (cpprcoder.h:678-802, 1094-1187)
//   rcx_scan_sizes_k     size prefix: exclusive scan of sizes[] -> offsets[] (u64)
//   rcx_scatter_k        pass 2: compacted scatter of the slots to dst + offsets[b]
//                        (the reference's MemoryStream is the sink, cpprcoder.h:1031-1054)
//   rcx_dec_adaptive_k   one lane per block decode.  (cpprcoder.h:859-940, 1189-1243)
//
// Roofline class: HBM-bound integer/byte work, no MFMA.  What actually bounds the two
// coder kernels is the serial dependency chain of one symbol (divide -> multiply ->
// renormalise -> table update) times the number of blocks in flight; see DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>

#include "rcx_lane.hpp"

#define RCX_ST_CAPACITY 1u
#define RCX_ST_CORRUPT 2u

// status[0] = OR of RCX_ST_* flags, status[1] = lowest failing block (saturated to u32)
__device__ __forceinline__ void rcx_flag(u32* status, u32 what, u64 blk)
{
    atomicOr(&status[0], what);
    atomicMin(&status[1], blk > 0xFFFFFFFEull ? 0xFFFFFFFEu : (u32)blk);
}

__device__ __forceinline__ u32 rcx_wave_max(u32 v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        u32 other = (u32)__shfl_xor((int)v, o, 64);
        v = v > other ? v : other;
    }
    return v;
}

__device__ __forceinline__ u32 rcx_byte_of(const U4& w, u32 j)
{
    const u32 word = (j < 4) ? w.x : (j < 8) ? w.y : (j < 12) ? w.z : w.w;
    return (word >> (8 * (j & 3))) & 0xFFu;
}

// LDS image of one wave: the 64 lane-interleaved trees, then 64 staged divisor entries.
#define RCX_LDS_U4 ((RCX_GROUPS + 1) * RCX_LANES)
// the decoder adds the 64 input rings (RCX_RING_DW dwords per lane, dword-interleaved)
#define RCX_DEC_LDS_U4 (RCX_LDS_U4 + RCX_RING_DW * RCX_LANES / 4)

// ===========================================================================
// Encode, pass 1
// ===========================================================================
// STREAM = the single-stream entry point's second pass: one block, and the lane also
// replays the reference's delayed writer to find where a bounded sink fills
// (track[0] = failing symbol or 0xFFFFFFFF, track[1] = 1 if only the final flush fails).
// LONG = a single stream of more than RCX_MAX_BLOCK symbols: no divisor table, the lane divides by its own total and
// halves the table at 2^24 (cpprcoder.h:1138-1176).
template <bool STREAM, bool LONG = false>
__global__ __launch_bounds__(64) void rcx_enc_adaptive_k(const u8* __restrict__ src, u64 n, u32 block, u64 nblocks,
                                                         u8* __restrict__ slots, u64 slot, u32* __restrict__ sizes,
                                                         const DivEntry* __restrict__ divtab, u32* status,
                                                         u32 sink_bytes, u32* track, const u32* __restrict__ only)
{
    __shared__ U4 lds[RCX_LDS_U4];
    const u32 lane = threadIdx.x;
    const u64 blk = (u64)blockIdx.x * RCX_LANES + lane;
    bool live = blk < nblocks;
    // second pass behind rcx_enc_mc5_k: only the blocks it marked (a carry through more output bytes than
    // it keeps in LDS: none on ordinary data)
    if (only) {
        live = live && only[blk] != 0;
        if (!__any(live)) return;
    }
    const u64 at = live ? blk * (u64)block : 0;
    const u32 len = live ? (u32)((n - at) < (u64)block ? (n - at) : (u64)block) : 0u;

    Tree tree{reinterpret_cast<u32*>(lds) + (RCX_TREE_PLANAR ? 1 : 4) * lane};
    tree.reset();
    DivEntry* stage = reinterpret_cast<DivEntry*>(lds + RCX_GROUPS * RCX_LANES);

    EncLane enc;
    u8* wave_slots = slots + (u64)blockIdx.x * RCX_LANES * slot; // wave-uniform; a lane's slot is a 32-bit offset from it
    if (live) enc.begin(wave_slots, lane * (u32)slot, (u32)slot, len);
    else enc.idle(wave_slots);
    if (STREAM) enc.trk_cap = sink_bytes;

    const u32 maxlen = rcx_wave_max(len);
    // fast path: every lane has a full block and 16-byte loads are aligned
    const bool full = !STREAM && __all(live && len == block) && (block % 16u == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    const u8* in = src + at;

    if (LONG) {
        u32 total = 256; // cpprcoder.h:1096
        for (u32 i = 0; i < maxlen; ++i)
            if (i < len) enc.template step_long<STREAM>(tree, in[i], total, i);
    } else {
    DivEntry ahead = divtab[lane];
    if (full) {
        U4 cur = *reinterpret_cast<const U4*>(in);
        for (u32 i0 = 0; i0 < maxlen; i0 += RCX_STAGE) {
            stage[lane] = ahead;
            ahead = divtab[i0 + RCX_STAGE + lane]; // table is padded by one stage
            const u32 jend = (maxlen - i0) < RCX_STAGE ? (maxlen - i0) : RCX_STAGE;
            for (u32 j0 = 0; j0 < jend; j0 += 16) {
                const u32 i = i0 + j0;
                U4 nxt = cur;
                if (i + 16 < maxlen) nxt = *reinterpret_cast<const U4*>(in + i + 16);
#pragma unroll
                for (u32 j = 0; j < 16; ++j) enc.step(tree, rcx_byte_of(cur, j), stage[j0 + j]);
                cur = nxt;
            }
        }
    } else {
        for (u32 i0 = 0; i0 < maxlen; i0 += RCX_STAGE) {
            stage[lane] = ahead;
            ahead = divtab[i0 + RCX_STAGE + lane];
            const u32 jend = (maxlen - i0) < RCX_STAGE ? (maxlen - i0) : RCX_STAGE;
            for (u32 j = 0; j < jend; ++j) {
                const u32 i = i0 + j;
                const DivEntry k = stage[j];
                if (i < len) enc.template step<STREAM>(tree, in[i], k, i);
            }
        }
    }
    }

    if (live) {
        if (STREAM) {
            track[0] = enc.trk_fail_at;
            track[1] = enc.track_flush_fails() ? 1u : 0u;
        }
        const u32 bytes = enc.finish();
        sizes[blk] = enc.overflow ? (u32)slot : bytes;
        if (enc.overflow) rcx_flag(status, RCX_ST_CAPACITY, blk);
    }
}

// ===========================================================================
// Size prefix: offsets[b] = sum_{i<b} sizes[i], offsets[nblocks] = total.
// One workgroup of 1024 threads; wave scans via DPP shuffles, 16 wave totals via LDS.
// ===========================================================================
__global__ __launch_bounds__(1024) void rcx_scan_sizes_k(const u32* __restrict__ sizes, u64 nblocks, u64* __restrict__ offsets,
                                                         u64 dst_cap, u32* status)
{
    __shared__ u64 wave_total[16];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 per = (nblocks + 1023) / 1024;
    const u64 first = (u64)tid * per;
    const u64 last = (first + per) < nblocks ? (first + per) : nblocks;
    u64 mine = 0;
    for (u64 b = first; b < last; ++b) mine += sizes[b];
    // inclusive wave scan
    u64 incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        u64 up = (u64)__shfl_up((unsigned long long)incl, o, 64);
        if ((int)lane >= o) incl += up;
    }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    u64 before = 0;
    for (u32 w = 0; w < wave; ++w) before += wave_total[w];
    u64 run = before + incl - mine;
    for (u64 b = first; b < last; ++b) {
        offsets[b] = run;
        run += sizes[b];
    }
    if (tid == 1023) {
        offsets[nblocks] = before + incl;
        if (before + incl > dst_cap) rcx_flag(status, RCX_ST_CAPACITY, nblocks);
    }
}

// ===========================================================================
// Encode, pass 2: slot b -> dst + offsets[b].  One workgroup per block; the
// destination is written in aligned 16-byte pieces, the (4-byte aligned) source
// words are byte-shifted into place.
// ===========================================================================
__global__ __launch_bounds__(256) void rcx_scatter_k(const u8* __restrict__ slots, u64 slot, const u32* __restrict__ sizes,
                                                     const u64* __restrict__ offsets, u8* __restrict__ dst, u64 dst_cap,
                                                     const u32* __restrict__ starts)
{
    const u64 blk = blockIdx.x;
    const u32 size = sizes[blk];
    const u64 off = offsets[blk];
    if (off + size > dst_cap) return; // flagged by the scan
    // the stream begins at the start of its slot (range coders) or wherever the backward-writing rANS encoders got to
    const u8* s = slots + blk * slot + (starts ? starts[blk] : 0u);
    u8* d = dst + off;
    const u32 tid = threadIdx.x;
    u32 head = (u32)((0 - reinterpret_cast<uintptr_t>(d)) & 15u);
    if (head > size) head = size;
    if (tid < head) d[tid] = s[tid];
    const u32 nvec = (size - head) >> 4;
    const uintptr_t from = reinterpret_cast<uintptr_t>(s) + head;
    const u32 sh = (u32)(from & 3u);
    const u32* w0p = reinterpret_cast<const u32*>(from & ~(uintptr_t)3);
    for (u32 v = tid; v < nvec; v += 256) {
        const u32* w = w0p + 4 * v;
        const u32 w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
        U4 out;
        out.x = __builtin_amdgcn_alignbyte(w1, w0, sh);
        out.y = __builtin_amdgcn_alignbyte(w2, w1, sh);
        out.z = __builtin_amdgcn_alignbyte(w3, w2, sh);
        out.w = __builtin_amdgcn_alignbyte(w4, w3, sh);
        *reinterpret_cast<U4*>(d + head + 16u * v) = out;
    }
    const u32 done = head + (nvec << 4);
    if (tid < size - done) d[done + tid] = s[done + tid];
}

// ===========================================================================
// Decode
// ===========================================================================
#if defined(RCX_STAMP_DEC)
static __device__ unsigned long long rcx_dec_stamp_out[8];
#endif
// `only` != nullptr: decode just the blocks with only[blk] != 0 (the others are left alone).
// STREAM = the single-stream entry point: one block whose symbol count n the host took from
// the header (max(declared,1) clipped to the sink); track[0] = first symbol whose normalize
// ran out of input, or 0xFFFFFFFF.
template <bool STREAM, bool LONG = false>
__global__ __launch_bounds__(64) void rcx_dec_adaptive_k(const u8* __restrict__ comp, u64 comp_size, const u64* __restrict__ offsets, u64 nblocks,
                                                         u32 block, u64 n, u8* __restrict__ dst,
                                                         const DivEntry* __restrict__ divtab, u32* status, u32* track,
                                                         const u32* __restrict__ only)
{
    __shared__ U4 lds[RCX_DEC_LDS_U4];
    const u32 lane = threadIdx.x;
    const u64 blk = (u64)blockIdx.x * RCX_LANES + lane;
    bool live = blk < nblocks;
    // second pass behind rcx_dec_quad_k: only the blocks it marked (none, on valid input)
    if (only) {
        live = live && only[blk] != 0;
        if (!__any(live)) return;
    }
    const u64 at = live ? blk * (u64)block : 0;
    u32 len = live ? (u32)((n - at) < (u64)block ? (n - at) : (u64)block) : 0u;

    Tree tree{reinterpret_cast<u32*>(lds) + (RCX_TREE_PLANAR ? 1 : 4) * lane};
    tree.reset();
    DivEntry* stage = reinterpret_cast<DivEntry*>(lds + RCX_GROUPS * RCX_LANES);
    u32* ring_col = reinterpret_cast<u32*>(lds + RCX_LDS_U4) + lane;

    DecLane dec;
#if defined(RCX_STAMP_DEC)
    for (int i_ = 0; i_ < 8; ++i_) dec.stamp_sum[i_] = 0;
    dec.stamp_last = __builtin_amdgcn_s_memtime();
#endif
    u64 stream_len = 0;
    if (live) {
        const u64 s0 = offsets[blk], s1 = offsets[blk + 1];
        stream_len = s1 - s0;
        if (s1 < s0 || s1 > comp_size || stream_len < (STREAM ? 8u : 9u)) { // cpprcoder.h:878: fewer than 8 bytes cannot even start
            rcx_flag(status, RCX_ST_CORRUPT, blk);
            live = false;
            len = 0;
        } else {
            const u32 declared = dec.begin(comp + s0, comp + s1, ring_col);
            if (!STREAM && declared != len) { // the layout says len; a header that disagrees is not ours
                rcx_flag(status, RCX_ST_CORRUPT, blk);
                live = false;
                len = 0;
            }
        }
    }
    if (!live) dec.idle(comp, ring_col);

    const u32 maxlen = rcx_wave_max(len);
    const bool full = !STREAM && __all(live && len == block) && (block % 16u == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0);
    u8* out = dst + at;

    if (LONG) {
        DivEntry k;
        k.mul = k.add = k.shift = 0;
        k.total = 256; // cpprcoder.h:1096
        for (u32 i = 0; i < maxlen; ++i) {
            if ((i & 15u) == 0) dec.topup();
            if (i < len) {
                out[i] = (u8)dec.template step<STREAM, true>(tree, k, i, stream_len);
                k.total += 1; // cpprcoder.h:1138 (the update itself was made by step)
                if (k.total >= RCX_HALVE_AT) k.total = tree.halve();
            }
        }
    } else {
    DivEntry ahead = divtab[lane];
    if (full) {
        for (u32 i0 = 0; i0 < maxlen; i0 += RCX_STAGE) {
            stage[lane] = ahead;
            ahead = divtab[i0 + RCX_STAGE + lane];
            const u32 jend = (maxlen - i0) < RCX_STAGE ? (maxlen - i0) : RCX_STAGE;
            for (u32 j0 = 0; j0 < jend; j0 += 16) {
                const u32 i = i0 + j0;
                u32 word[4] = {0, 0, 0, 0};
                dec.topup();
                DivEntry k_next = stage[j0]; // divisor of the next symbol: fetched one symbol early
#pragma unroll
                for (u32 j = 0; j < 16; ++j) {
                    const DivEntry k = k_next;
                    if (j + 1 < 16) k_next = stage[j0 + j + 1];
                    const u32 c = dec.step(tree, k);
                    word[j >> 2] |= c << (8 * (j & 3));
                }
                U4 o;
                o.x = word[0];
                o.y = word[1];
                o.z = word[2];
                o.w = word[3];
                *reinterpret_cast<U4*>(out + i) = o;
            }
        }
    } else {
        for (u32 i0 = 0; i0 < maxlen; i0 += RCX_STAGE) {
            stage[lane] = ahead;
            ahead = divtab[i0 + RCX_STAGE + lane];
            const u32 jend = (maxlen - i0) < RCX_STAGE ? (maxlen - i0) : RCX_STAGE;
            for (u32 j = 0; j < jend; ++j) {
                const u32 i = i0 + j;
                const DivEntry k = stage[j];
                if ((j & 15u) == 0) dec.topup();
                if (i < len) out[i] = (u8)dec.template step<STREAM>(tree, k, i, stream_len);
            }
        }
    }
    }
#if defined(RCX_STAMP_DEC)
    if (blockIdx.x == 7 && lane == 0)
        for (int i_ = 0; i_ < 8; ++i_) rcx_dec_stamp_out[i_] = dec.stamp_sum[i_];
#endif
    // the reference returns Status_Pending when normalize runs out of input (cpprcoder.h:901-903)
    if (STREAM) {
        if (live) track[0] = dec.short_at;
    } else if (live && dec.taken() > stream_len) {
        rcx_flag(status, RCX_ST_CORRUPT, blk);
    }
}

// ===========================================================================
// The resumable single-stream decoder (rcx_dstream_*, include/rcx.h): AdaptiveRangeDecoder<T>::decode called
// piece by piece (cpprcoder.h:872-924).  One lane; its whole state -- low, range, the model, how far it got --
// lives in `st` between launches, so every call decodes only what the new bytes allow (the reference does the
// same on its object).  A symbol is started only if the bytes its renormalisation needs have arrived (the
// reference stops in the middle of the renormalisation, :901-903, and resumes there: same bytes, same symbols).
// ===========================================================================
struct alignas(16) RcxDState {
    U4 tree[RCX_GROUPS];
    u32 low, range, total, started;
    u32 declared, produced; // produced counts towards max(declared, 1) (cpprcoder.h:912)
    u64 consumed;
};

__global__ __launch_bounds__(64) void rcx_dec_resume_k(RcxDState* __restrict__ st, const u8* __restrict__ in, u64 avail, u8* __restrict__ out,
                                                       u32 room, u32* __restrict__ result)
{
    __shared__ U4 lds[RCX_GROUPS * RCX_LANES];
    if (threadIdx.x != 0) return;
    Tree tree{reinterpret_cast<u32*>(lds)};
    u32 low = st->low, range = st->range, total = st->total, produced = st->produced, declared = st->declared;
    u64 consumed = st->consumed;
    if (!st->started) { // cpprcoder.h:859-870, :877-896: the caller made sure the first 8 bytes are here
        declared = (u32)in[0] | ((u32)in[1] << 8) | ((u32)in[2] << 16) | ((u32)in[3] << 24);
        low = ((u32)in[4] << 24) | ((u32)in[5] << 16) | ((u32)in[6] << 8) | (u32)in[7];
        range = 0x00FFFFFFu;
        total = 256;
        consumed = 8;
        produced = 0;
        tree.reset();
    } else {
        for (u32 g = 0; g < RCX_GROUPS; ++g) tree.store(g, st->tree[g]);
    }
    const u32 want = declared ? declared : 1u; // :912: the size test comes after the first writeByte
    u32 made = 0;
    while (produced < want && made < room) {
        const u32 k8 = rcx_clz(range) & 0x18u; // :926-940
        const u32 need = k8 >> 3;
        if (consumed + need > avail) break;     // input ran dry before this symbol (:901-903)
        for (u32 b = 0; b < need; ++b) low = (low << 8) | in[consumed + b];
        consumed += need;
        range <<= k8;
        out[made++] = (u8)rcx_decode_plain(tree, low, range, total);
        produced += 1;
    }
    st->low = low, st->range = range, st->total = total, st->started = 1;
    st->declared = declared, st->produced = produced, st->consumed = consumed;
    for (u32 g = 0; g < RCX_GROUPS; ++g) st->tree[g] = tree.group(g);
    result[0] = made;
    result[1] = produced >= want ? 1u : 0u; // finished
    result[2] = declared;
    result[3] = produced;
    result[4] = (u32)consumed; // how far into the stream the decoder has read: the host drops what lies before
    result[5] = (u32)(consumed >> 32);
}

// ===========================================================================
// The resumable single-stream encoder (rcx_estream_*, include/rcx.h): AdaptiveRangeEncoder<T>::encode called piece by
// piece (cpprcoder.h:697-720).  One lane; low, range, the model, the bytes it still holds and where the reference's
// delayed writer stands (held byte + pending 0xFF run, cpprcoder.h:764-802) live in `st` between launches.  After a
// launch the stream's payload in `slot` is complete up to the bytes the coder still holds in its register -- those
// are written behind it as they stand, without being given up -- so that the host can hand on exactly the bytes the
// reference has passed to writeByte by then: payload[written before, written now).  A byte the reference has written
// never changes (a carry stops at its held byte), so what the host copies is final.
// ===========================================================================
struct alignas(16) RcxEState {
    U4 tree[RCX_GROUPS];
    u64 acc;
    u32 low, range, total, started;
    u32 declared, consumed;      // symbols taken so far
    u32 nacc8, pos, overflow;
    u32 trk_written, trk_pending, trk_fail_at;
};

// result: {payload bytes the reference has written so far, symbol at which its sink filled or 0xFFFFFFFF, 1 if this
// launch finished the stream, stream size if finished, 1 if only finish() ran into the full sink, slot overflow,
// payload bytes in memory}
__global__ __launch_bounds__(64) void rcx_enc_resume_k(RcxEState* __restrict__ st, const u8* __restrict__ in, u32 count, u8* __restrict__ slot,
                                                       u32 slot_bytes, u32 sink_room, u32* __restrict__ result)
{
    __shared__ U4 lds[RCX_GROUPS * RCX_LANES];
    if (threadIdx.x != 0) return;
    Tree tree{reinterpret_cast<u32*>(lds)};
    EncLane enc;
    u32 total, consumed;
    const u32 declared = st->declared;
    if (!st->started) { // cpprcoder.h:678-695
        enc.begin(slot, 0, slot_bytes, declared);
        tree.reset();
        total = 256;
        consumed = 0;
    } else {
        enc.base = slot;
        enc.off = 4;
        enc.cap = (slot_bytes - 4) & ~3u;
        enc.leader = true;
        enc.low = st->low, enc.range = st->range, enc.acc = st->acc, enc.nacc8 = st->nacc8, enc.pos = st->pos, enc.overflow = st->overflow;
        enc.trk_written = st->trk_written, enc.trk_pending = st->trk_pending, enc.trk_fail_at = st->trk_fail_at;
        total = st->total;
        consumed = st->consumed;
        for (u32 g = 0; g < RCX_GROUPS; ++g) tree.store(g, st->tree[g]);
    }
    // the sink takes `sink_room` more bytes through writeByte from here on
    enc.trk_cap = sink_room > 0xFFFFFFFFu - enc.trk_written ? 0xFFFFFFFFu : enc.trk_written + sink_room;
    u32 i = 0;
    for (; i < count && enc.trk_fail_at == 0xFFFFFFFFu; ++i) enc.template step_long<true>(tree, in[i], total, consumed + i);
    u32 finished = 0, size = 0, flush_fails = 0;
    const bool failed = enc.trk_fail_at != 0xFFFFFFFFu;
    if (!failed) consumed += count;
    else consumed = enc.trk_fail_at; // the reference stops inside that symbol (cpprcoder.h:708-711): its state is of no use any more, nor is this one
    if (!failed && consumed >= declared) { // cpprcoder.h:714-717
        flush_fails = enc.track_flush_fails() ? 1u : 0u;
        size = enc.finish();
        finished = 1;
    } else {
        // what the register holds, behind what is in memory, as it stands (a carry that has run off it first)
        const u32 extra = (u32)(enc.acc >> enc.nacc8);
        if (extra) {
            enc.carry_into_memory(extra);
            enc.acc &= (1ull << enc.nacc8) - 1ull;
        }
        const u32 n = enc.nacc8 >> 3;
        u8* out = enc.payload();
        for (u32 k = 0; k < n; ++k)
            if (enc.pos + k < enc.cap) out[enc.pos + k] = (u8)(enc.acc >> (8 * (n - 1 - k)));
    }
    st->low = enc.low, st->range = enc.range, st->acc = enc.acc, st->nacc8 = enc.nacc8, st->pos = enc.pos, st->overflow = enc.overflow;
    st->trk_written = enc.trk_written, st->trk_pending = enc.trk_pending, st->trk_fail_at = enc.trk_fail_at;
    st->total = total, st->consumed = consumed, st->started = 1;
    for (u32 g = 0; g < RCX_GROUPS; ++g) st->tree[g] = tree.group(g);
    result[0] = enc.trk_written;
    result[1] = enc.trk_fail_at;
    result[2] = finished;
    result[3] = size;
    result[4] = flush_fails;
    result[5] = enc.overflow;
    result[6] = finished ? size - 4 : enc.pos + (enc.nacc8 >> 3); // payload bytes in memory now (the register's included)
}

#include "rcx_oct.hpp"
#if defined(RCX_WITH_VARIANTS) // superseded kernels, kept for comparison: only in the diagnostic build (build.py build_variants)
#include "variants/rcx_variants.hpp"
#endif
#include "rcx_static.hpp"
#include "rcx_rans.hpp"
#include "rcx_bwt.hpp"
