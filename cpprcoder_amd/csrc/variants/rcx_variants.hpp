// rcx_variants.hpp -- superseded kernels, kept selectable (RCX_ENC_VARIANT=1/2, RCX_LANES_PER_BLOCK=8) so that
// tests/test_gpu_parity.py::test_every_kernel_variant_is_bit_identical and the sweeps can compare them with the
// defaults in rcx_oct.hpp.  Included after rcx_oct.hpp by rcx_kernels.hpp.
//
// rcx_enc_oct_k / rcx_dec_oct_k -- why 8 lanes per block.  A block is one serial chain per symbol, so 1 GiB of 64 KiB blocks
// offers only 16384 chains.  With one lane per block that is 256 waves -- one per CU, three
// of four SIMDs idle, and every LDS/VMEM wait exposed (measured: 39 % / 58 % of the encode /
// decode wave cycles are waits).  With 8 lanes per block it is 2048 waves = 2 per SIMD: the
// partner wave covers the waits, and the 8 lanes split the model work:
//   * the table is a 2-level tree: 32 node sums (8 symbols each) + 256 counts;
//     lane j owns the 4 node sums 4j..4j+3 (one ds_read_b128 at a FIXED address) and keeps
//     B = the sum of all nodes before its group in a register;
//   * encode: cum(c) = [lane c>>5] B + nodes before node(c) in its group
//                    + [lanes < c&7] the leaf counts they read         -> one octet sum
//   * decode: find() is two rounds of compares in the scaled domain (threshold * t <= low):
//     round 1 finds the node among 32 (each lane tests its 4), round 2 the leaf among 8
//     (each lane tests 1) -- one dependent LDS read per symbol instead of four, no second
//     divide (cpprcoder.h:905) and low -= cum*t falls out of the descent.
// The coder arithmetic (low/range/carry/renormalise, cpprcoder.h:703-711, :764-802,
// :926-940) runs redundantly in all 8 lanes (SIMT makes that free); only lane 0 of the
// octet touches global memory.  Cross-lane sums are 3 DPP steps (quad_perm, quad_perm,
// row_half_mirror).  Results are bit-identical to the one-lane-per-block kernels.
#pragma once

#define RCX_OCT 8                 /* lanes per block */
#define RCX_OCT_BLOCKS 8          /* blocks per wave */
#define RCX_OCT_NODE_BYTES 128    /* 32 node sums */
#define RCX_OCT_BLOCK_BYTES 1152  /* + 256 counts */
#define RCX_OCT_LDS_BYTES (RCX_OCT_BLOCKS * RCX_OCT_BLOCK_BYTES + RCX_STAGE * 16)
#define RCX_OCT_DEC_LDS_BYTES (RCX_OCT_LDS_BYTES + RCX_RING_DW * RCX_LANES * 4)


struct OctModel {
    U4* nodes;   // this lane's group of 4 node sums (LDS)
    u32* leaves; // the block's 256 counts (LDS)
    u32 before;  // sum of the node sums of groups 0..j-1
    u32 j;

    __device__ __forceinline__ void reset(u8* lds_block, u32 lane_in_oct)
    {
        j = lane_in_oct;
        nodes = reinterpret_cast<U4*>(lds_block) + j;
        leaves = reinterpret_cast<u32*>(lds_block + RCX_OCT_NODE_BYTES);
        U4 v;
        v.x = v.y = v.z = v.w = 8; // cpprcoder.h:1094-1132: every count 1
        *nodes = v;
        U4 one;
        one.x = one.y = one.z = one.w = 1;
        U4* l4 = reinterpret_cast<U4*>(leaves);
#pragma unroll
        for (u32 q = 0; q < 8; ++q) l4[q * 8 + j] = one;
        before = 32u * j;
    }
    // cpprcoder.h:1134-1177 (+1; the halving cannot trigger below 2^24 symbols)
    __device__ __forceinline__ void update(u32 node, u32 leaf_lane)
    {
        const u32 grp = node >> 2;
        rcx_lds_add(&leaves[node * 8 + j], j == leaf_lane ? 1u : 0u);
        rcx_lds_add(reinterpret_cast<u32*>(nodes) + (node & 3), j == grp ? 1u : 0u);
        before += j > grp ? 1u : 0u;
    }
};

// ===========================================================================
// Encode, pass 1 (8 lanes per block, 8 blocks per wave, one wave per workgroup)
// ===========================================================================
__global__ __launch_bounds__(64) void rcx_enc_oct_k(const u8* __restrict__ src, u64 n, u32 block, u64 nblocks,
                                                    u8* __restrict__ slots, u64 slot, u32* __restrict__ sizes,
                                                    const DivEntry* __restrict__ divtab, u32* status)
{
    __shared__ __attribute__((aligned(16))) u8 lds[RCX_OCT_LDS_BYTES];
    const u32 lane = threadIdx.x;
    const u32 j = lane & 7u, oct = lane >> 3;
    const u64 blk = (u64)blockIdx.x * RCX_OCT_BLOCKS + oct;
    const bool live = blk < nblocks;
    const u64 at = live ? blk * (u64)block : 0;
    const u32 len = live ? (u32)((n - at) < (u64)block ? (n - at) : (u64)block) : 0u;

    OctModel model;
    model.reset(lds + oct * RCX_OCT_BLOCK_BYTES, j);
    DivEntry* stage = reinterpret_cast<DivEntry*>(lds + RCX_OCT_BLOCKS * RCX_OCT_BLOCK_BYTES);

    EncLane enc;
    u8* wave_slots = slots + (u64)blockIdx.x * RCX_OCT_BLOCKS * slot;
    if (live) {
        enc.begin(wave_slots, oct * (u32)slot, (u32)slot, len); // all 8 lanes write the same 4 header bytes
    } else {
        enc.idle(wave_slots);
    }
    enc.leader = live && j == 0;

    const u32 maxlen = rcx_wave_max(len);
    const bool full = __all(live && len == block) && (block % 16u == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    const u8* in = src + at;

#define RCX_OCT_ENC_SYMBOL(SYM, K)                                                        \
    {                                                                                     \
        const u32 c_ = (SYM);                                                             \
        const u32 node_ = c_ >> 3, grp_ = c_ >> 5, lp_ = c_ & 7u;                         \
        const U4 g_ = *model.nodes;                                                       \
        const u32 fj_ = model.leaves[node_ * 8 + j];                                      \
        u32 part_ = (j == grp_) ? model.before + rcx_pre4(g_, node_ & 3u) : 0u;          \
        part_ += (j < lp_) ? fj_ : 0u;                                                    \
        const u32 cum_ = rcx_oct_sum(part_);                                              \
        const u32 f_ = rcx_oct_sum(j == lp_ ? fj_ : 0u);                                  \
        enc.code(cum_, f_, (K));                                                          \
        model.update(node_, lp_);                                                         \
    }

    DivEntry ahead = divtab[lane];
    if (full) {
        U4 cur = *reinterpret_cast<const U4*>(in);
        for (u32 i0 = 0; i0 < maxlen; i0 += RCX_STAGE) {
            stage[lane] = ahead;
            ahead = divtab[i0 + RCX_STAGE + lane];
            const u32 jend = (maxlen - i0) < RCX_STAGE ? (maxlen - i0) : RCX_STAGE;
            for (u32 j0 = 0; j0 < jend; j0 += 16) {
                const u32 i = i0 + j0;
                U4 nxt = cur;
                if (i + 16 < maxlen) nxt = *reinterpret_cast<const U4*>(in + i + 16);
#pragma unroll
                for (u32 s = 0; s < 16; ++s) RCX_OCT_ENC_SYMBOL(rcx_byte_of(cur, s), stage[j0 + s]);
                cur = nxt;
            }
        }
    } else {
        for (u32 i0 = 0; i0 < maxlen; i0 += RCX_STAGE) {
            stage[lane] = ahead;
            ahead = divtab[i0 + RCX_STAGE + lane];
            const u32 jend = (maxlen - i0) < RCX_STAGE ? (maxlen - i0) : RCX_STAGE;
            for (u32 s = 0; s < jend; ++s) {
                const u32 i = i0 + s;
                const DivEntry k = stage[s];
                if (i < len) RCX_OCT_ENC_SYMBOL(in[i], k); // len is the same in all 8 lanes of an octet
            }
        }
    }
#undef RCX_OCT_ENC_SYMBOL

    const u32 bytes = enc.finish();
    if (enc.leader) {
        sizes[blk] = enc.overflow ? (u32)slot : bytes;
        if (enc.overflow) rcx_flag(status, RCX_ST_CAPACITY, blk);
    }
}

// ===========================================================================
// Decode (8 lanes per block)
// ===========================================================================
// A workgroup is WAVES independent waves (no barrier between them).  With few blocks, 8 waves per workgroup
// land two on each SIMD of one CU, which single-wave workgroups do not guarantee (measured: 20 % of the
// kernel time); with many blocks single-wave workgroups pack more waves onto a CU.
#define RCX_OCT_DEC_WAVES 8
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rcx_dec_oct_k(const u8* __restrict__ comp, u64 comp_size, const u64* __restrict__ offsets,
                                                            u64 nblocks, u32 block, u64 n, u8* __restrict__ dst,
                                                            const DivEntry* __restrict__ divtab, u32* status)
{
    __shared__ __attribute__((aligned(16))) u8 lds_all[WAVES * RCX_OCT_DEC_LDS_BYTES];
    const u32 lane = threadIdx.x & 63u;
    const u32 wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u8* lds = lds_all + wave_in_wg * RCX_OCT_DEC_LDS_BYTES;
    const u32 j = lane & 7u, oct = lane >> 3;
    const u64 blk = ((u64)blockIdx.x * WAVES + wave_in_wg) * RCX_OCT_BLOCKS + oct;
    bool live = blk < nblocks;
    const u64 at = live ? blk * (u64)block : 0;
    u32 len = live ? (u32)((n - at) < (u64)block ? (n - at) : (u64)block) : 0u;

    OctModel model;
    model.reset(lds + oct * RCX_OCT_BLOCK_BYTES, j);
    DivEntry* stage = reinterpret_cast<DivEntry*>(lds + RCX_OCT_BLOCKS * RCX_OCT_BLOCK_BYTES);
    // every lane keeps its own copy of the octet's input ring (8 identical columns: no cross-lane ordering needed)
    u32* ring_col = reinterpret_cast<u32*>(lds + RCX_OCT_LDS_BYTES) + lane;
    const u32 m1 = (j & 1u) ? ~0u : 0u, m2 = (j & 2u) ? ~0u : 0u, m4 = (j & 4u) ? ~0u : 0u;

    DecLane dec;
    u64 stream_len = 0;
    if (live) {
        const u64 s0 = offsets[blk], s1 = offsets[blk + 1];
        stream_len = s1 - s0;
        if (s1 < s0 || s1 > comp_size || stream_len < 9) {
            if (j == 0) rcx_flag(status, RCX_ST_CORRUPT, blk);
            live = false;
            len = 0;
        } else {
            const u32 declared = dec.begin(comp + s0, comp + s1, ring_col);
            if (declared != len) {
                if (j == 0) rcx_flag(status, RCX_ST_CORRUPT, blk);
                live = false;
                len = 0;
            }
        }
    }
    if (!live) dec.idle(comp, ring_col);

    const u32 maxlen = rcx_wave_max(len);
    const bool full = __all(live && len == block) && (block % 16u == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0);
    u8* out = dst + at;
    const bool leader = live && j == 0;

    // One symbol.  All comparisons are in the scaled domain: with t = range/total the reference's
    // "cum(c) <= low/t < cum(c+1)" (cpprcoder.h:905, :1220-1242) is "cum(c)*t <= low < cum(c+1)*t",
    // every product is <= total*t <= range < 2^32, and low - cum(c)*t (cpprcoder.h:906) is what
    // is left when the descent ends.
#define RCX_OCT_DEC_SYMBOL(K, SYM)                                                                         \
    {                                                                                                      \
        dec.pull();                                                                                        \
        const DivEntry k_ = (K);                                                                           \
        const u32 t_ = rcx_div(dec.range, k_);                                                             \
        const u32 top_ = rcx_mul24(k_.total, t_);                                                          \
        const U4 g_ = *model.nodes;                                                                        \
        /* round 1: which of the 32 nodes */                                                               \
        const u32 d_ = dec.low - rcx_mul24(model.before, t_);                                              \
        const u32 s2_ = g_.x + g_.y, s3_ = s2_ + g_.z, s4_ = s3_ + g_.w;                                   \
        const u32 a_ = rcx_mul24(g_.x, t_), b_ = rcx_mul24(s2_, t_), c_ = rcx_mul24(s3_, t_);              \
        const u32 e_ = rcx_mul24(s4_, t_);                                                                 \
        u32 p_ = 0, base_ = 0;                                                                             \
        if (d_ >= a_) { p_ = 1; base_ = a_; }                                                              \
        if (d_ >= b_) { p_ = 2; base_ = b_; }                                                              \
        if (d_ >= c_) { p_ = 3; base_ = c_; }                                                              \
        const bool own1_ = d_ < e_;                                                                        \
        const u32 node_ = rcx_oct_sum(own1_ ? 4u * j + p_ : 0u);                                           \
        const u32 rem_ = rcx_oct_sum(own1_ ? d_ - base_ : 0u);                                             \
        /* round 2: which of the node's 8 symbols */                                                       \
        const u32 fj_ = model.leaves[node_ * 8 + j];                                                       \
        const u32 ex_ = rcx_oct_excl_scan(fj_, m1, m2, m4);                                                \
        const u32 w_ = rcx_mul24(fj_, t_);                                                                 \
        const u32 d2_ = rem_ - rcx_mul24(ex_, t_);                                                         \
        const bool own2_ = d2_ < w_;                                                                       \
        u32 low_ = rcx_oct_sum(own2_ ? d2_ : 0u);                                                          \
        const u32 range_ = rcx_oct_sum(own2_ ? w_ : 0u);                                                   \
        const u32 lp_ = rcx_oct_sum(own2_ ? j : 0u);                                                       \
        /* target >= total: the reference's find() falls through to code 0 / count = total */              \
        if (dec.low >= top_) low_ = dec.low - top_;                                                        \
        dec.low = low_;                                                                                    \
        dec.range = range_;                                                                                \
        model.update(node_, lp_);                                                                          \
        (SYM) = node_ * 8 + lp_;                                                                           \
    }

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
                DivEntry k_next = stage[j0];
#pragma unroll
                for (u32 s = 0; s < 16; ++s) {
                    u32 sym;
                    const DivEntry kk = k_next;
                    if (s + 1 < 16) k_next = stage[j0 + s + 1];
                    RCX_OCT_DEC_SYMBOL(kk, sym);
                    word[s >> 2] |= sym << (8 * (s & 3));
                }
                if (leader) {
                    U4 o;
                    o.x = word[0];
                    o.y = word[1];
                    o.z = word[2];
                    o.w = word[3];
                    *reinterpret_cast<U4*>(out + i) = o;
                }
            }
        }
    } else {
        for (u32 i0 = 0; i0 < maxlen; i0 += RCX_STAGE) {
            stage[lane] = ahead;
            ahead = divtab[i0 + RCX_STAGE + lane];
            const u32 jend = (maxlen - i0) < RCX_STAGE ? (maxlen - i0) : RCX_STAGE;
            for (u32 s = 0; s < jend; ++s) {
                const u32 i = i0 + s;
                const DivEntry k = stage[s];
                if ((s & 15u) == 0) dec.topup();
                if (i < len) {
                    u32 sym;
                    RCX_OCT_DEC_SYMBOL(k, sym);
                    if (leader) out[i] = (u8)sym;
                }
            }
        }
    }
#undef RCX_OCT_DEC_SYMBOL
    if (leader && dec.taken() > stream_len) rcx_flag(status, RCX_ST_CORRUPT, blk);
}

// ===========================================================================
// Encode, pass 1, model/coder split ("MC"): the encoder's model side does not depend on the
// coder state -- cum(c_i) and f(c_i) are functions of the input prefix alone
// (cpprcoder.h:706-712) -- so one workgroup = 64 blocks runs as a 4-wave software pipeline:
//   wave 1  model, tree levels 3+2: partial cum, its ds_add updates
//   wave 2  model, tree level 1:    partial cum, its ds_add update
//   wave 3  model, leaf level:      partial cum + f, its ds_add update
//   wave 0  coder: divide / multiply / carry / renormalise / emit (cpprcoder.h:703-711, :764-802)
// The model waves run one 16-symbol chunk ahead and hand {cumA, cumB, cumC, f} over through a
// double-buffered LDS ring; one s_barrier per chunk.  Every instruction still serves 64 blocks
// (one lane per block), the four instruction streams run on the four SIMDs of the CU at once,
// and the time per symbol is the coder wave's alone.  Bytes are identical to the other kernels.
// ===========================================================================
template <bool FULL>
__device__ __forceinline__ void rcx_mc_pipeline(u32 wave, u32 lane, u32 len, u32 nchunks, const u8* in,
                                                const DivEntry* __restrict__ divtab, const Tree& tree, DivEntry* stage,
                                                U4* ring, EncLane& enc, DivEntry& ahead)
{
#if defined(RCX_STAMP)
    unsigned long long stamp_wait_ = 0;
    const unsigned long long stamp_begin_ = __builtin_amdgcn_s_memtime();
#endif
    // model waves: the input piece of the next chunk is loaded while this one is processed
    U4 piece_ahead;
    piece_ahead.x = piece_ahead.y = piece_ahead.z = piece_ahead.w = 0;
    if (FULL && wave != 0 && nchunks > 0) piece_ahead = *reinterpret_cast<const U4*>(in);
    for (u32 k = 0; k <= nchunks; ++k) {
        if (wave == 0) {
            // ---- coder: chunk k-1 ----
            if (k >= 1) {
                const u32 i0 = (k - 1) * RCX_MC_CHUNK;
                if ((i0 % RCX_STAGE) == 0) { // the next 64 divisors, loaded one stage ahead
                    stage[lane] = ahead;
                    ahead = divtab[i0 + RCX_STAGE + lane];
                }
                const U4* rs = ring + ((k - 1) & 1u) * (RCX_MC_CHUNK * RCX_LANES) + lane;
                const DivEntry* st = stage + (i0 % RCX_STAGE);
                // entry and divisor of the next symbol are fetched before the current one is coded,
                // so their LDS latency hides behind the coder arithmetic
                U4 e_next = rs[0];
                DivEntry k_next = st[0];
#pragma unroll
                for (u32 s = 0; s < RCX_MC_CHUNK; ++s) {
                    const U4 e = e_next;
                    const DivEntry kk = k_next;
                    if (s + 1 < RCX_MC_CHUNK) {
                        e_next = rs[(s + 1) * RCX_LANES];
                        k_next = st[s + 1];
                    }
                    if (FULL || i0 + s < len) enc.code(e.x + e.y + e.z, e.w, kk);
                }
            }
        } else if (k < nchunks) {
            // ---- model: chunk k ----
            const u32 i0 = k * RCX_MC_CHUNK;
            U4* ws = ring + (k & 1u) * (RCX_MC_CHUNK * RCX_LANES) + lane;
            U4 piece;
            if (FULL) {
                piece = piece_ahead;
                if (k + 1 < nchunks) piece_ahead = *reinterpret_cast<const U4*>(in + i0 + RCX_MC_CHUNK);
            } else {
                u32 w[4] = {0, 0, 0, 0};
                for (u32 s = 0; s < RCX_MC_CHUNK; ++s)
                    if (i0 + s < len) w[s >> 2] |= (u32)in[i0 + s] << (8 * (s & 3));
                piece.x = w[0];
                piece.y = w[1];
                piece.z = w[2];
                piece.w = w[3];
            }
            // Software pipeline, one symbol deep: the reads AND the ds_add updates of symbol s+1 are
            // issued before the sums of symbol s are formed.  LDS executes a wave's operations in
            // order, so the reads of s+1 still see the updates of s, and their latency hides
            // behind the arithmetic of s (the updates need only the symbol, not the read data).
            if (wave == 1) {
                u32 c = rcx_byte_of(piece, 0);
                bool on = FULL || i0 < len;
                U4 g3 = tree.group(RCX_G_L3), g2 = tree.group(RCX_G_L2 + (c >> 6));
                if (on) {
                    tree.bump(RCX_G_L3, c >> 6);
                    tree.bump(RCX_G_L2 + (c >> 6), (c >> 4) & 3);
                }
#pragma unroll
                for (u32 s = 0; s < RCX_MC_CHUNK; ++s) {
                    const u32 cc = c;
                    const bool onc = on;
                    const U4 h3 = g3, h2 = g2;
                    if (s + 1 < RCX_MC_CHUNK) {
                        c = rcx_byte_of(piece, s + 1);
                        on = FULL || i0 + s + 1 < len;
                        g3 = tree.group(RCX_G_L3);
                        g2 = tree.group(RCX_G_L2 + (c >> 6));
                        if (on) {
                            tree.bump(RCX_G_L3, c >> 6);
                            tree.bump(RCX_G_L2 + (c >> 6), (c >> 4) & 3);
                        }
                    }
                    if (onc) reinterpret_cast<u32*>(&ws[s * RCX_LANES])[0] = rcx_pre4(h3, cc >> 6) + rcx_pre4(h2, (cc >> 4) & 3);
                }
            } else if (wave == 2) {
                u32 c = rcx_byte_of(piece, 0);
                bool on = FULL || i0 < len;
                U4 g1 = tree.group(RCX_G_L1 + (c >> 4));
                if (on) tree.bump(RCX_G_L1 + (c >> 4), (c >> 2) & 3);
#pragma unroll
                for (u32 s = 0; s < RCX_MC_CHUNK; ++s) {
                    const u32 cc = c;
                    const bool onc = on;
                    const U4 h1 = g1;
                    if (s + 1 < RCX_MC_CHUNK) {
                        c = rcx_byte_of(piece, s + 1);
                        on = FULL || i0 + s + 1 < len;
                        g1 = tree.group(RCX_G_L1 + (c >> 4));
                        if (on) tree.bump(RCX_G_L1 + (c >> 4), (c >> 2) & 3);
                    }
                    if (onc) reinterpret_cast<u32*>(&ws[s * RCX_LANES])[1] = rcx_pre4(h1, (cc >> 2) & 3);
                }
            } else {
                u32 c = rcx_byte_of(piece, 0);
                bool on = FULL || i0 < len;
                U4 g0 = tree.group(RCX_G_L0 + (c >> 2));
                if (on) tree.bump(RCX_G_L0 + (c >> 2), c & 3);
#pragma unroll
                for (u32 s = 0; s < RCX_MC_CHUNK; ++s) {
                    const u32 cc = c;
                    const bool onc = on;
                    const U4 h0 = g0;
                    if (s + 1 < RCX_MC_CHUNK) {
                        c = rcx_byte_of(piece, s + 1);
                        on = FULL || i0 + s + 1 < len;
                        g0 = tree.group(RCX_G_L0 + (c >> 2));
                        if (on) tree.bump(RCX_G_L0 + (c >> 2), c & 3);
                    }
                    if (onc) {
                        u32* e = reinterpret_cast<u32*>(&ws[s * RCX_LANES]);
                        e[2] = rcx_pre4(h0, cc & 3);
                        e[3] = rcx_sel4(h0, cc & 3);
                    }
                }
            }
        }
#if defined(RCX_STAMP) /* diagnostic build only: where do the waves of workgroup 0 wait? */
        const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
        rcx_lds_barrier();
        const unsigned long long t1_ = __builtin_amdgcn_s_memtime();
        rcx_stamp_wait += t1_ - t0_;
#else
        rcx_lds_barrier();
#endif
    }
#if defined(RCX_STAMP)
    if (blockIdx.x == 7 && lane == 0) {
        rcx_stamp_out[wave * 2] = __builtin_amdgcn_s_memtime() - stamp_begin_;
        rcx_stamp_out[wave * 2 + 1] = stamp_wait_;
    }
#endif
}

__global__ __launch_bounds__(RCX_MC_THREADS) void rcx_enc_mc_k(const u8* __restrict__ src, u64 n, u32 block, u64 nblocks,
                                                              u8* __restrict__ slots, u64 slot, u32* __restrict__ sizes,
                                                              const DivEntry* __restrict__ divtab, u32* status)
{
    __shared__ U4 lds[RCX_MC_LDS_U4];
    const u32 lane = threadIdx.x & 63u;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u64 blk = (u64)blockIdx.x * RCX_LANES + lane;
    const bool live = blk < nblocks;
    const u64 at = live ? blk * (u64)block : 0;
    const u32 len = live ? (u32)((n - at) < (u64)block ? (n - at) : (u64)block) : 0u;

    Tree tree{reinterpret_cast<u32*>(lds) + (RCX_TREE_PLANAR ? 1 : 4) * lane};
    DivEntry* stage = reinterpret_cast<DivEntry*>(lds + RCX_GROUPS * RCX_LANES);
    U4* ring = lds + RCX_LDS_U4;

    const u32 maxlen = rcx_wave_max(len); // the four waves hold the same 64 blocks: same value in each
    const bool full = __all(live && len == block) && (block % 16u == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    const u8* in = src + at;
    const u32 nchunks = (maxlen + RCX_MC_CHUNK - 1) / RCX_MC_CHUNK;

    EncLane enc;
    DivEntry ahead;
    ahead.mul = ahead.add = ahead.shift = ahead.total = 0;
    U4 v;
    if (wave == 0) {
        u8* wave_slots = slots + (u64)blockIdx.x * RCX_LANES * slot;
        if (live) enc.begin(wave_slots, lane * (u32)slot, (u32)slot, len);
        else enc.idle(wave_slots);
        ahead = divtab[lane];
    } else if (wave == 1) { // cpprcoder.h:1094-1132: every count 1
        v.x = v.y = v.z = v.w = 64;
        tree.store(RCX_G_L3, v);
        v.x = v.y = v.z = v.w = 16;
        for (u32 g = RCX_G_L2; g < RCX_G_L1; ++g) tree.store(g, v);
    } else if (wave == 2) {
        v.x = v.y = v.z = v.w = 4;
        for (u32 g = RCX_G_L1; g < RCX_G_L0; ++g) tree.store(g, v);
    } else {
        v.x = v.y = v.z = v.w = 1;
        for (u32 g = RCX_G_L0; g < RCX_GROUPS; ++g) tree.store(g, v);
    }
    rcx_lds_barrier();

    if (full) rcx_mc_pipeline<true>(wave, lane, len, nchunks, in, divtab, tree, stage, ring, enc, ahead);
    else rcx_mc_pipeline<false>(wave, lane, len, nchunks, in, divtab, tree, stage, ring, enc, ahead);

    if (wave == 0 && live) {
        const u32 bytes = enc.finish();
        sizes[blk] = enc.overflow ? (u32)slot : bytes;
        if (enc.overflow) rcx_flag(status, RCX_ST_CAPACITY, blk);
    }
}

