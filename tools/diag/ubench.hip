// ubench.hip -- instruction-cost microbenchmarks for gfx950 (diagnostic tool, not part of librcx.so).
//
// One wave (optionally several per SIMD) runs REPS copies of a short instruction pattern between two
// s_memtime reads; the program prints shader-clock cycles per copy.  Used to decide between
// instruction sequences in the coder kernels (what does an s_nop cost a lone wave, is v_mad_u64_u32
// full rate, what is the LDS latency of a dependent ds_read_b128, what does a taken branch cost).
//
//   hipcc --offload-arch=gfx950 -O2 -o build/ubench tools/diag/ubench.hip && build/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define REPS 256
#define STR2(x) #x
#define STR(x) STR2(x)

#define BENCH_KERNEL(name, setup, body)                                                          \
    __global__ void name(unsigned long long* out, unsigned* sink)                                 \
    {                                                                                             \
        __shared__ unsigned lds[4096];                                                            \
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0;                          \
        __syncthreads();                                                                          \
        unsigned v0 = threadIdx.x, v1 = 3, v2 = 5, v3 = 7;                                        \
        unsigned long long t0, t1;                                                                \
        asm volatile(setup "\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t" \
                     ".rept " STR(REPS) "\n\t" body "\n\t.endr\n\t"                                \
                     "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)"  \
                     : [t0] "=&s"(t0), [t1] "=&s"(t1), [a] "+v"(v0), [b] "+v"(v1), [c] "+v"(v2), [d] "+v"(v3) \
                     :                                                                            \
                     : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "v40", "v41", "v42", \
                       "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "memory");                             \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                \
        sink[threadIdx.x] = v0 + v1 + v2 + v3 + lds[threadIdx.x];                                 \
    }

BENCH_KERNEL(k_add_dep, "", "v_add_u32 %[a], %[a], %[b]")
BENCH_KERNEL(k_add_indep, "", "v_add_u32 %[a], %[b], %[c]\n\tv_add_u32 %[d], %[b], %[c]")
BENCH_KERNEL(k_mul24, "", "v_mul_u32_u24 %[a], %[a], %[b]")
BENCH_KERNEL(k_mad64, "v_mov_b32 v40, 1\n\tv_mov_b32 v41, 0", "v_mad_u64_u32 v[40:41], s[20:21], %[a], %[b], v[40:41]")
BENCH_KERNEL(k_mulhi, "", "v_mul_hi_u32 %[a], %[a], %[b]")
BENCH_KERNEL(k_mullo, "", "v_mul_lo_u32 %[a], %[a], %[b]")
BENCH_KERNEL(k_lshl64, "v_mov_b32 v40, 1\n\tv_mov_b32 v41, 0", "v_lshlrev_b64 v[40:41], %[b], v[40:41]")
BENCH_KERNEL(k_add_nop0, "", "v_add_u32 %[a], %[a], %[b]\n\ts_nop 0")
BENCH_KERNEL(k_add_nop1, "", "v_add_u32 %[a], %[a], %[b]\n\ts_nop 1")
BENCH_KERNEL(k_add_nop3, "", "v_add_u32 %[a], %[a], %[b]\n\ts_nop 3")
BENCH_KERNEL(k_dpp_dep, "", "s_nop 1\n\tv_add_u32_dpp %[a], %[a], %[a] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
BENCH_KERNEL(k_dpp_fill, "", "v_add_u32 %[c], %[c], %[b]\n\tv_add_u32 %[d], %[d], %[b]\n\tv_add_u32_dpp %[a], %[a], %[a] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
BENCH_KERNEL(k_cmp_addc, "", "v_cmp_lt_u32 vcc, %[a], %[b]\n\ts_nop 1\n\tv_addc_co_u32 %[c], vcc, 0, %[c], vcc")
BENCH_KERNEL(k_cmp5_addc5, "",
             "v_cmp_lt_u32_e64 s[20:21], %[a], %[b]\n\tv_cmp_lt_u32_e64 s[22:23], %[a], %[c]\n\tv_cmp_lt_u32_e64 s[24:25], %[a], %[d]\n\t"
             "v_cmp_lt_u32_e64 s[26:27], %[b], %[c]\n\tv_cmp_lt_u32_e64 s[28:29], %[b], %[d]\n\t"
             "v_addc_co_u32_e64 v40, s[20:21], 0, v40, s[20:21]\n\tv_addc_co_u32_e64 v41, s[22:23], 0, v41, s[22:23]\n\t"
             "v_addc_co_u32_e64 v42, s[24:25], 0, v42, s[24:25]\n\tv_addc_co_u32_e64 v43, s[26:27], 0, v43, s[26:27]\n\t"
             "v_addc_co_u32_e64 v44, s[28:29], 0, v44, s[28:29]")
BENCH_KERNEL(k_min3, "", "v_min3_u32 %[a], %[a], %[b], %[c]")
BENCH_KERNEL(k_perm, "", "v_perm_b32 %[a], %[a], %[b], %[c]")
BENCH_KERNEL(k_alignbit, "", "v_alignbit_b32 %[a], %[a], %[b], %[c]")
BENCH_KERNEL(k_lds_b32, "v_mov_b32 v40, 0", "ds_read_b32 v40, v40\n\ts_waitcnt lgkmcnt(0)")
BENCH_KERNEL(k_lds_b64, "v_mov_b32 v40, 0", "ds_read_b64 v[40:41], v40\n\ts_waitcnt lgkmcnt(0)")
BENCH_KERNEL(k_lds_b128, "v_lshlrev_b32 v40, 4, %[a]\n\tv_mov_b32 v45, v40", "ds_read_b128 v[40:43], v45\n\ts_waitcnt lgkmcnt(0)\n\tv_or_b32 v45, v45, v40")
BENCH_KERNEL(k_lds_b128_same, "v_mov_b32 v40, 0\n\tv_mov_b32 v45, v40", "ds_read_b128 v[40:43], v45\n\ts_waitcnt lgkmcnt(0)\n\tv_or_b32 v45, v45, v40")
BENCH_KERNEL(k_lds_read2, "v_lshlrev_b32 v40, 2, %[a]", "ds_read2_b32 v[40:41], v40 offset1:16\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 v40, 0xfc, v40")
BENCH_KERNEL(k_lds_add, "v_lshlrev_b32 v40, 2, %[a]", "ds_add_u32 v40, %[b]")
BENCH_KERNEL(k_lds_add_read, "v_lshlrev_b32 v40, 4, %[a]\n\tv_mov_b32 v45, v40", "ds_add_u32 v45, %[b]\n\tds_read_b128 v[40:43], v45\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 v45, 0x3f0, v45")
BENCH_KERNEL(k_branch_taken, "", "s_branch 1f\n\tv_add_u32 %[a], %[a], %[b]\n1:\n\tv_add_u32 %[c], %[c], %[b]")
BENCH_KERNEL(k_branch_not, "s_mov_b64 vcc, 0", "s_cbranch_vccnz 1f\n\tv_add_u32 %[c], %[c], %[b]\n1:")
BENCH_KERNEL(k_cbranch_taken, "s_mov_b64 vcc, 0", "s_cbranch_vccz 1f\n\tv_add_u32 %[a], %[a], %[b]\n1:\n\tv_add_u32 %[c], %[c], %[b]")
BENCH_KERNEL(k_cvt_rcp, "", "v_cvt_f32_u32 %[a], %[a]\n\tv_rcp_f32 %[a], %[a]")
BENCH_KERNEL(k_ffbh, "", "v_ffbh_u32 %[a], %[a]")
BENCH_KERNEL(k_readlane, "", "v_readfirstlane_b32 s20, %[a]\n\tv_add_u32 %[a], s20, %[a]")
BENCH_KERNEL(k_salu, "", "s_add_u32 s20, s20, 1")
BENCH_KERNEL(k_valu_salu, "", "v_add_u32 %[a], %[a], %[b]\n\ts_add_u32 s20, s20, 1")


// LDS instructions inside a stream of vector instructions, results never waited for: what does ISSUING one cost a lone
// wave?  Eight dependent adds (k_t_none: 8 x 4.17) and one LDS operation per copy; the difference is the LDS operation.
#define V8 "v_add_u32 %[a], %[a], %[b]\n\tv_add_u32 %[a], %[a], %[b]\n\tv_add_u32 %[a], %[a], %[b]\n\tv_add_u32 %[a], %[a], %[b]\n\t" \
           "v_add_u32 %[a], %[a], %[b]\n\tv_add_u32 %[a], %[a], %[b]\n\tv_add_u32 %[a], %[a], %[b]\n\tv_add_u32 %[a], %[a], %[b]\n\t"
#define TSET "v_lshlrev_b32 v40, 4, %[c]\n\tv_and_b32 v40, 0x3ff0, v40\n\tv_mov_b32 v41, 1\n\tv_mov_b32 v42, 1\n\tv_mov_b32 v43, 1\n\tv_mov_b32 v44, 1\n\tv_mov_b32 v45, 1\n\tv_mov_b32 v46, 1\n\tv_mov_b32 v47, 1"
BENCH_KERNEL(k_t_none, TSET, V8)
BENCH_KERNEL(k_t_rd_b32, TSET, V8 "ds_read_b32 v48, v40")
BENCH_KERNEL(k_t_rd_b64, TSET, V8 "ds_read_b64 v[48:49], v40")
BENCH_KERNEL(k_t_rd_b128, TSET, V8 "ds_read_b128 v[48:51], v40")
BENCH_KERNEL(k_t_rd2_b32, TSET, V8 "ds_read2_b32 v[48:49], v40 offset1:1")
BENCH_KERNEL(k_t_rd2_b64, TSET, V8 "ds_read2_b64 v[48:51], v40 offset1:1")
BENCH_KERNEL(k_t_wr_b32, TSET, V8 "ds_write_b32 v40, v41")
BENCH_KERNEL(k_t_wr_b8, TSET, V8 "ds_write_b8 v40, v41")
BENCH_KERNEL(k_t_wr_b64, TSET, V8 "ds_write_b64 v40, v[42:43]")
BENCH_KERNEL(k_t_wr_b128, TSET, V8 "ds_write_b128 v40, v[44:47]")
BENCH_KERNEL(k_t_add, TSET, V8 "ds_add_u32 v40, v41")
BENCH_KERNEL(k_t_add_rd128, TSET, V8 "ds_add_u32 v40, v41\n\tds_read_b128 v[48:51], v40")
BENCH_KERNEL(k_t_wr128_rd128, TSET, V8 "ds_write_b128 v40, v[44:47]\n\tds_read_b128 v[48:51], v40")
BENCH_KERNEL(k_t_waitcnt, TSET, V8 "s_waitcnt lgkmcnt(0)")
BENCH_KERNEL(k_t_sdwa, TSET, V8 "v_cndmask_b32_sdwa %[d], %[d], %[b], vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_0")

// Vector memory instructions whose 64 lanes go to 64 different places (one 16-byte piece per lane, the lanes STRIDE bytes
// apart: 65536 = the coders' one-block-per-lane access, 16 = adjacent pieces): what does ONE of them cost a lone wave in a
// stream of vector instructions (eight dependent adds per copy), results never waited for?  `sink` must hold 4 MiB + 1 KiB.
#define VMEM_KERNEL(name, STRIDE, body)                                                           \
    __global__ void name(unsigned long long* out, unsigned* sink)                                 \
    {                                                                                             \
        unsigned v0 = threadIdx.x, v1 = 3;                                                        \
        unsigned long long ad = (unsigned long long)sink + (unsigned long long)(threadIdx.x & 63) * (STRIDE); \
        unsigned long long t0, t1;                                                                \
        asm volatile("v_mov_b32 v44, 1\n\tv_mov_b32 v45, 1\n\tv_mov_b32 v46, 1\n\tv_mov_b32 v47, 1\n\t"   \
                     "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t"     \
                     ".rept " STR(REPS) "\n\t" V8 body "\n\t.endr\n\t"                                \
                     "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)"       \
                     : [t0] "=&s"(t0), [t1] "=&s"(t1), [a] "+v"(v0), [b] "+v"(v1)                  \
                     : [ad] "v"(ad)                                                               \
                     : "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "memory");        \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                \
        sink[0] = v0;                                                                             \
    }
VMEM_KERNEL(k_v_ld_scatter, 65536, "global_load_dwordx4 v[48:51], %[ad], off")
VMEM_KERNEL(k_v_ld_contig, 16, "global_load_dwordx4 v[48:51], %[ad], off")
VMEM_KERNEL(k_v_st_scatter, 65536, "global_store_dwordx4 %[ad], v[44:47], off")
VMEM_KERNEL(k_v_st_contig, 16, "global_store_dwordx4 %[ad], v[44:47], off")
VMEM_KERNEL(k_v_st4_scatter, 65536, "global_store_dword %[ad], v44, off")

// ---------------------------------------------------------------------------------------------
// The quad decoder's LDS pattern: per "symbol" one broadcast ds_read2_b64, PRE dependent VALU ops, a
// ds_read_b128 whose address differs per quad (16 distinct 64-byte windows), a ds_read2_b32, GAP
// independent VALU ops, s_waitcnt lgkmcnt(1) + use, POST VALU ops, a ds_add.  Run with 4 waves per
// workgroup (one per SIMD, as in the kernel); cycles per copy minus 4 x (instructions) = exposed wait.
// ---------------------------------------------------------------------------------------------
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)
#define VADD "v_add_u32 %[a], %[a], %[b]\n\t"
#define VIND "v_add_u32 %[c], %[c], %[b]\n\t"
#define SYM_KERNEL(name, gap, leafread, atomic)                                                    \
    __global__ void name(unsigned long long* out, unsigned* sink)                                 \
    {                                                                                             \
        __shared__ unsigned lds[20480];                                                           \
        for (int i = threadIdx.x; i < 20480; i += blockDim.x) lds[i] = 0;                         \
        __syncthreads();                                                                          \
        const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 2, j = lane & 3; \
        unsigned v0 = threadIdx.x, v1 = 0, v2 = 5, v3 = 7;                                        \
        unsigned base = wave * 20480;                                                             \
        unsigned leaf = base + 512 + q * 1232 + ((q * 7 + wave * 3) & 15) * 64 + j * 16;          \
        unsigned ring = base + 512 + q * 1232 + 1088 + ((q * 5) & 31) * 4;                        \
        unsigned cnt = base + 512 + q * 1232 + (((q * 11) & 15) * 16 + j * 4 + (q & 3)) * 4;      \
        unsigned long long t0, t1;                                                                \
        asm volatile("v_mov_b32 v50, %[kb]\n\tv_mov_b32 v51, %[lf]\n\tv_mov_b32 v52, %[rg]\n\tv_mov_b32 v53, %[ct]\n\t" \
                     "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t" \
                     ".rept " STR(REPS) "\n\t"                                                     \
                     "ds_read2_b64 v[40:43], v50 offset1:1\n\t" REP16(VADD) REP4(VADD)            \
                     leafread "\n\t"                                                               \
                     "ds_read2_b32 v[48:49], v52 offset1:1\n\t" gap                               \
                     "s_waitcnt lgkmcnt(1)\n\tv_or_b32 %[a], %[a], v44\n\t" REP16(VADD) REP8(VADD) REP4(VADD) \
                     atomic REP4(VADD)                                                             \
                     "s_waitcnt lgkmcnt(1)\n\tv_or_b32 %[a], %[a], v48\n\tv_or_b32 %[a], %[a], v40\n\t" \
                     ".endr\n\t"                                                                   \
                     "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)"  \
                     : [t0] "=&s"(t0), [t1] "=&s"(t1), [a] "+v"(v0), [b] "+v"(v1), [c] "+v"(v2), [d] "+v"(v3) \
                     : [kb] "v"(base), [lf] "v"(leaf), [rg] "v"(ring), [ct] "v"(cnt)              \
                     : "vcc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", \
                       "v52", "v53", "memory");                                                   \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                \
        sink[threadIdx.x & 4095] = v0 + v1 + v2 + v3 + lds[threadIdx.x];                          \
    }
#define LEAF128 "ds_read_b128 v[44:47], v51"
#define LEAF64 "ds_read_b64 v[44:45], v51"
#define ATOM "ds_add_u32 v53, %[b]\n\t"
SYM_KERNEL(k_sym_gap0, "", LEAF128, ATOM)
SYM_KERNEL(k_sym_gap4, REP4(VIND), LEAF128, ATOM)
SYM_KERNEL(k_sym_gap8, REP8(VIND), LEAF128, ATOM)
SYM_KERNEL(k_sym_gap12, REP8(VIND) REP4(VIND), LEAF128, ATOM)
SYM_KERNEL(k_sym_gap16, REP16(VIND), LEAF128, ATOM)
SYM_KERNEL(k_sym_gap24, REP16(VIND) REP8(VIND), LEAF128, ATOM)
SYM_KERNEL(k_sym_gap32, REP16(VIND) REP16(VIND), LEAF128, ATOM)
SYM_KERNEL(k_sym_gap12_noatom, REP8(VIND) REP4(VIND), LEAF128, "")
SYM_KERNEL(k_sym_gap12_b64, REP8(VIND) REP4(VIND), LEAF64, ATOM)
SYM_KERNEL(k_sym_gap0_b64, "", LEAF64, ATOM)

// The quad decoder's three instruction sequences with fixed registers (timing only: the values are arbitrary).
#define QP1 " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define QP2 " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define BLOCK_A                                                                   \
    "v_sub_co_u32_e64 v40, s[20:21], %[a], %[b]\n\t"                              \
    "v_sub_co_u32_e64 v41, s[22:23], %[a], %[c]\n\t"                              \
    "v_sub_co_u32_e64 v42, s[24:25], %[a], %[d]\n\t"                              \
    "v_sub_co_u32_e64 v43, s[26:27], %[a], %[b]\n\t"                              \
    "v_subb_co_u32_e64 v44, s[20:21], 4, 0, s[20:21]\n\t"                         \
    "v_min3_u32 v45, v40, v41, v42\n\t"                                           \
    "v_subb_co_u32_e64 v44, s[22:23], v44, 0, s[22:23]\n\t"                       \
    "v_min3_u32 v45, v45, v43, %[a]\n\t"                                          \
    "v_subb_co_u32_e64 v44, s[24:25], v44, 0, s[24:25]\n\t"                       \
    "v_subb_co_u32_e64 v44, s[26:27], v44, 0, s[26:27]\n\t"                       \
    "v_min_u32_dpp v45, v45, v45" QP1                                             \
    "v_add_u32 v46, v46, %[b]\n\t"                                                \
    "v_add_u32_dpp v44, v44, v44" QP1                                             \
    "v_min_u32_dpp v45, v45, v45" QP2                                             \
    "v_bfe_u32 v47, v46, 5, 5\n\t"                                                \
    "v_add_u32_dpp v44, v44, v44" QP2                                             \
    "v_lshl_add_u32 %[a], v44, 8, v45\n\t"
#define BLOCK_U                                                                   \
    "v_cmp_lt_u32_e64 s[20:21], %[a], %[b]\n\t"                                   \
    "v_cmp_lt_u32_e64 s[22:23], %[a], %[c]\n\t"                                   \
    "v_cmp_lt_u32_e64 s[24:25], %[a], %[d]\n\t"                                   \
    "v_cmp_le_u32_e64 s[26:27], %[a], %[d]\n\t"                                   \
    "v_addc_co_u32_e64 v40, s[20:21], 0, v40, s[20:21]\n\t"                       \
    "v_addc_co_u32_e64 v41, s[22:23], 0, v41, s[22:23]\n\t"                       \
    "v_addc_co_u32_e64 v42, s[24:25], 0, v42, s[24:25]\n\t"                       \
    "v_addc_co_u32_e64 v43, s[26:27], 0, v43, s[26:27]\n\t"                       \
    "v_lshl_add_u32 %[a], %[a], 4, v40\n\t"
#define BLOCK_C                                                                   \
    "v_add_u32 v40, %[a], %[b]\n\t"                                               \
    "v_add_u32 v41, v40, %[c]\n\t"                                                \
    "v_add_u32 v42, v41, %[d]\n\t"                                                \
    "v_mul_u32_u24 v43, %[a], %[b]\n\t"                                           \
    "v_mul_u32_u24 v44, v40, %[b]\n\t"                                            \
    "v_add_u32_dpp v45, v42, v42" QP1                                             \
    "v_and_b32_dpp v46, v42, %[c]" QP1                                            \
    "v_mul_u32_u24 v47, v41, %[b]\n\t"                                            \
    "v_mul_u32_u24 v48, v42, %[b]\n\t"                                            \
    "v_and_b32_dpp v49, v45, %[d]" QP2                                            \
    "v_add_u32 v46, v46, v49\n\t"                                                 \
    "v_mul_u32_u24 v49, v46, %[b]\n\t"                                            \
    "v_sub_u32 v50, %[a], v49\n\t"                                                \
    "v_sub_co_u32_e64 v51, s[20:21], v50, v43\n\t"                                \
    "v_sub_co_u32_e64 v52, s[22:23], v50, v44\n\t"                                \
    "v_sub_co_u32_e64 v53, s[24:25], v50, v47\n\t"                                \
    "v_sub_co_u32_e64 v54, s[26:27], v50, v48\n\t"                                \
    "v_min3_u32 v55, v50, v51, v52\n\t"                                           \
    "v_max3_u32 v56, v51, v52, v53\n\t"                                           \
    "v_min_u32 v55, v55, v53\n\t"                                                 \
    "v_max_u32 v56, v56, v54\n\t"                                                 \
    "v_subb_co_u32_e64 v57, s[20:21], %[a], 0, s[20:21]\n\t"                      \
    "v_min_u32_dpp v55, v55, v55" QP1                                             \
    "v_subb_co_u32_e64 v57, s[22:23], v57, 0, s[22:23]\n\t"                       \
    "v_max_u32_dpp v56, v56, v56" QP1                                             \
    "v_subb_co_u32_e64 v57, s[24:25], v57, 0, s[24:25]\n\t"                       \
    "v_min_u32_dpp v55, v55, v55" QP2                                             \
    "v_cndmask_b32_e64 v58, 0, 1, s[26:27]\n\t"                                   \
    "v_max_u32_dpp v56, v56, v56" QP2                                             \
    "v_cndmask_b32_e64 v54, 0, v57, s[26:27]\n\t"                                 \
    "v_sub_u32 %[a], v55, v56\n\t"                                                \
    "v_lshl_or_b32 %[c], v54, 8, %[c]\n\t"                                        \
    "v_alignbit_b32 v59, %[b], %[d], v50\n\t"                                     \
    "v_perm_b32 v59, v59, v59, s28\n\t"                                           \
    "v_and_b32 v54, 3, v57\n\t"                                                   \
    "v_lshl_add_u32 v54, v54, 2, v55\n\t"
#define BK(name, body, clob)                                                                      \
    __global__ void name(unsigned long long* out, unsigned* sink)                                 \
    {                                                                                             \
        unsigned v0 = threadIdx.x, v1 = 3, v2 = 5, v3 = 7;                                        \
        unsigned long long t0, t1;                                                                \
        asm volatile("s_mov_b32 s28, 0x00010203\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t" \
                     ".rept " STR(REPS) "\n\t" body ".endr\n\t"                                    \
                     "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)"  \
                     : [t0] "=&s"(t0), [t1] "=&s"(t1), [a] "+v"(v0), [b] "+v"(v1), [c] "+v"(v2), [d] "+v"(v3) \
                     :                                                                            \
                     : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "v40", "v41", "v42", "v43", \
                       "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", \
                       "v58", "v59", "memory");                                                   \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                \
        sink[threadIdx.x & 4095] = v0 + v1 + v2 + v3;                                             \
    }
BK(k_blk_a, BLOCK_A, 0)
BK(k_blk_u, BLOCK_U, 0)
BK(k_blk_c, BLOCK_C, 0)
BK(k_blk_auc, BLOCK_A BLOCK_U BLOCK_C, 0)

// the same in a loop that fits the instruction cache: 16 copies x 16 iterations
#define BKL(name, body)                                                                           \
    __global__ void name(unsigned long long* out, unsigned* sink)                                 \
    {                                                                                             \
        unsigned v0 = threadIdx.x, v1 = 3, v2 = 5, v3 = 7;                                        \
        unsigned long long t0, t1;                                                                \
        asm volatile("s_mov_b32 s28, 0x00010203\n\ts_mov_b32 s29, 16\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t" \
                     "2:\n\t.rept 16\n\t" body ".endr\n\t"                                         \
                     "s_sub_u32 s29, s29, 1\n\ts_cmp_lg_u32 s29, 0\n\ts_cbranch_scc1 2b\n\t"       \
                     "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)"  \
                     : [t0] "=&s"(t0), [t1] "=&s"(t1), [a] "+v"(v0), [b] "+v"(v1), [c] "+v"(v2), [d] "+v"(v3) \
                     :                                                                            \
                     : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "v40", "v41", "v42", "v43", \
                       "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", \
                       "v58", "v59", "memory");                                                   \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                \
        sink[threadIdx.x & 4095] = v0 + v1 + v2 + v3;                                             \
    }
BKL(k_loop_auc, BLOCK_A BLOCK_U BLOCK_C)
BKL(k_loop_add62, REP16(VADD) REP16(VADD) REP16(VADD) REP8(VADD) REP4(VADD) "v_add_u32 %[a], %[a], %[b]\n\tv_add_u32 %[a], %[a], %[b]\n\t")
BKL(k_loop_min3_62, REP16("v_min3_u32 %[a], %[a], %[b], %[c]\n\t") REP16("v_min3_u32 %[a], %[a], %[b], %[c]\n\t") REP16("v_min3_u32 %[a], %[a], %[b], %[c]\n\t") REP8("v_min3_u32 %[a], %[a], %[b], %[c]\n\t") REP4("v_min3_u32 %[a], %[a], %[b], %[c]\n\t") "v_min3_u32 %[a], %[a], %[b], %[c]\n\tv_min3_u32 %[a], %[a], %[b], %[c]\n\t")

// instruction-size patterns (8 = v_min3_u32, VOP3, 8 bytes; 4 = v_add_u32 e32, 4 bytes; D = v_add_u32_dpp, 8 bytes)
#define I8 "v_min3_u32 %[a], %[a], %[b], %[c]\n\t"
#define I4 "v_add_u32 %[a], %[a], %[b]\n\t"
#define ID "v_add_u32_dpp %[c], %[d], %[d] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
BKL(k_pat_84, REP16(I8 I4) REP8(I8 I4) REP4(I8 I4) I8 I4 I8 I4 I8 I4)
BKL(k_pat_884, REP16(I8 I8 I4) REP4(I8 I8 I4) I8 I8)
BKL(k_pat_844, REP16(I8 I4 I4) REP4(I8 I4 I4) I8 I4)
BKL(k_pat_8884, REP8(I8 I8 I8 I4) REP4(I8 I8 I8 I4) I8 I8 I8 I4 I8 I8 I8 I4 I8 I8 I8 I4 I8 I8)
BKL(k_pat_D62, REP16(ID) REP16(ID) REP16(ID) REP8(ID) REP4(ID) ID ID)

// VGPR bank conflicts?  sources in one bank (v44, v48, v52) or in three (v45, v50, v55); with a 4-byte filler so that fetch is not the limit
BKL(k_bank_same3, REP16("v_min3_u32 v40, v44, v48, v52\n\t" I4) REP8("v_min3_u32 v40, v44, v48, v52\n\t" I4) REP4("v_min3_u32 v40, v44, v48, v52\n\t" I4) "v_min3_u32 v40, v44, v48, v52\n\t" I4 "v_min3_u32 v40, v44, v48, v52\n\t" I4 "v_min3_u32 v40, v44, v48, v52\n\t" I4)
BKL(k_bank_diff3, REP16("v_min3_u32 v40, v45, v50, v55\n\t" I4) REP8("v_min3_u32 v40, v45, v50, v55\n\t" I4) REP4("v_min3_u32 v40, v45, v50, v55\n\t" I4) "v_min3_u32 v40, v45, v50, v55\n\t" I4 "v_min3_u32 v40, v45, v50, v55\n\t" I4 "v_min3_u32 v40, v45, v50, v55\n\t" I4)
BKL(k_bank_same2, REP16("v_add_u32 v40, v44, v48\n\t") REP16("v_add_u32 v40, v44, v48\n\t") REP16("v_add_u32 v40, v44, v48\n\t") REP8("v_add_u32 v40, v44, v48\n\t") REP4("v_add_u32 v40, v44, v48\n\t") "v_add_u32 v40, v44, v48\n\tv_add_u32 v40, v44, v48\n\t")
BKL(k_mad64_same, REP16("v_mad_u64_u32 v[40:41], s[20:21], v44, v48, v[52:53]\n\t" I4) REP8("v_mad_u64_u32 v[40:41], s[20:21], v44, v48, v[52:53]\n\t" I4) REP4("v_mad_u64_u32 v[40:41], s[20:21], v44, v48, v[52:53]\n\t" I4) "v_mad_u64_u32 v[40:41], s[20:21], v44, v48, v[52:53]\n\t" I4 "v_mad_u64_u32 v[40:41], s[20:21], v44, v48, v[52:53]\n\t" I4 "v_mad_u64_u32 v[40:41], s[20:21], v44, v48, v[52:53]\n\t" I4)
BKL(k_subco_chain, REP16("v_sub_co_u32_e64 v40, s[20:21], v44, v45\n\tv_subb_co_u32_e64 v41, s[22:23], v46, 0, s[24:25]\n\t") REP8("v_sub_co_u32_e64 v40, s[20:21], v44, v45\n\tv_subb_co_u32_e64 v41, s[22:23], v46, 0, s[24:25]\n\t") REP4("v_sub_co_u32_e64 v40, s[20:21], v44, v45\n\tv_subb_co_u32_e64 v41, s[22:23], v46, 0, s[24:25]\n\t") "v_sub_co_u32_e64 v40, s[20:21], v44, v45\n\tv_subb_co_u32_e64 v41, s[22:23], v46, 0, s[24:25]\n\t" "v_sub_co_u32_e64 v40, s[20:21], v44, v45\n\tv_subb_co_u32_e64 v41, s[22:23], v46, 0, s[24:25]\n\t" "v_sub_co_u32_e64 v40, s[20:21], v44, v45\n\tv_subb_co_u32_e64 v41, s[22:23], v46, 0, s[24:25]\n\t")

struct Case {
    const char* name;
    void (*fn)(unsigned long long*, unsigned*);
    int per; // instructions of interest per copy, for the reader
};

int main(int argc, char** argv)
{
    const int waves = argc > 1 ? atoi(argv[1]) : 1; // waves per workgroup (1 = a lone wave on its SIMD)
    const int wgs = argc > 2 ? atoi(argv[2]) : 1;
    unsigned long long* out;
    unsigned* sink;
    hipMalloc(&out, 8);
    hipMalloc(&sink, (4 << 20) + 4096 * 4);
    const char* only = argc > 3 ? argv[3] : nullptr;
#define C(k, per) {#k, k, per}
    std::vector<Case> cases = {C(k_add_dep, 1), C(k_add_indep, 2), C(k_mul24, 1), C(k_mad64, 1), C(k_mulhi, 1), C(k_mullo, 1),
                               C(k_lshl64, 1), C(k_add_nop0, 2), C(k_add_nop1, 2), C(k_add_nop3, 2), C(k_dpp_dep, 2), C(k_dpp_fill, 3),
                               C(k_cmp_addc, 3), C(k_cmp5_addc5, 10), C(k_min3, 1), C(k_perm, 1), C(k_alignbit, 1), C(k_lds_b32, 1),
                               C(k_lds_b64, 1), C(k_lds_b128, 2), C(k_lds_b128_same, 2), C(k_lds_read2, 2), C(k_lds_add, 1), C(k_lds_add_read, 3),
                               C(k_branch_taken, 2), C(k_branch_not, 2), C(k_cbranch_taken, 2), C(k_cvt_rcp, 2), C(k_ffbh, 1),
                               C(k_readlane, 2), C(k_salu, 1), C(k_valu_salu, 2),
                               C(k_t_none, 8), C(k_t_rd_b32, 9), C(k_t_rd_b64, 9), C(k_t_rd_b128, 9), C(k_t_rd2_b32, 9), C(k_t_rd2_b64, 9), C(k_t_wr_b32, 9), C(k_t_wr_b8, 9),
                               C(k_t_wr_b64, 9), C(k_t_wr_b128, 9), C(k_t_add, 9), C(k_t_add_rd128, 10), C(k_t_wr128_rd128, 10), C(k_t_waitcnt, 9), C(k_t_sdwa, 9), C(k_v_ld_scatter, 9), C(k_v_ld_contig, 9), C(k_v_st_scatter, 9), C(k_v_st_contig, 9), C(k_v_st4_scatter, 9),
                               C(k_sym_gap0, 65), C(k_sym_gap4, 69), C(k_sym_gap8, 73), C(k_sym_gap12, 77), C(k_sym_gap16, 81), C(k_sym_gap24, 89),
                               C(k_sym_gap32, 97), C(k_sym_gap12_noatom, 76), C(k_sym_gap12_b64, 77), C(k_sym_gap0_b64, 65),
                               C(k_blk_a, 17), C(k_blk_u, 9), C(k_blk_c, 36), C(k_blk_auc, 62), C(k_loop_auc, 62), C(k_loop_add62, 62), C(k_loop_min3_62, 62),
                               C(k_pat_84, 62), C(k_pat_884, 62), C(k_pat_844, 62), C(k_pat_8884, 62), C(k_pat_D62, 62),
                               C(k_bank_same3, 62), C(k_bank_diff3, 62), C(k_bank_same2, 62), C(k_mad64_same, 62), C(k_subco_chain, 62)};
    printf("waves/workgroup %d, workgroups %d, %d copies per measurement\n", waves, wgs, REPS);
    for (auto& c : cases) {
        if (only && !strstr(c.name, only)) continue;
        unsigned long long best = ~0ull;
        for (int r = 0; r < 5; ++r) {
            hipLaunchKernelGGL(c.fn, dim3(wgs), dim3(64 * waves), 0, 0, out, sink);
            unsigned long long h = 0;
            hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
            if (h < best) best = h;
        }
        printf("%-18s %8.2f cycles per copy (%d instr)\n", c.name, (double)best / REPS, c.per);
    }
    return 0;
}
