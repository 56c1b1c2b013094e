// rcx_comm.hip -- the one exchange step of the path on RCCL: concatenating the per-GPU compressed segments (and
// their block tables) on every GPU of a node.  One process per GPU, one rcx_comm per process.
//
// The reference has no multi-device code; blocks are independent, so coding needs no
// collective.  RCCL has no allgatherv: after ONE fixed-size all-gather of {segment bytes, block count} every rank
// knows where each segment goes, and the payload moves as grouped point-to-point sends and receives straight into
// `concat + seg_base[peer]` -- no staging buffer, no padding -- which on xGMI (a full mesh of point-to-point links)
// is one direct transfer per link.  The block tables travel the same way, already shifted by their segment's base.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>
#include <vector>

#include "../../include/rcx.h"

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

struct rcx_comm {
    int device = 0;
    int nranks = 1;
    int rank = 0;
    ncclComm_t nccl = nullptr;
    u64* d_mine = nullptr;   // what this rank brings and has room for: RCX_COMM_WORDS words (rcx_pack_sizes_k)
    u64* d_every = nullptr;  // the same of every rank
    u64* h_every = nullptr;  // pinned copy
};

namespace
{
#define HIP_TRY(expr)                           \
    do {                                        \
        if ((expr) != hipSuccess) return RCX_E_HIP; \
    } while (0)
#define NCCL_TRY(expr)                           \
    do {                                         \
        if ((expr) != ncclSuccess) return RCX_E_COMM; \
    } while (0)

// What a rank tells the others: {segment bytes, block count, room in its concat buffer, room in its table, table wanted}.
// The capacities travel too, so that every rank judges the exchange by the SAME numbers (the smallest room anywhere) and
// all of them refuse together -- a rank that stopped alone would leave its peers waiting in their receives.
#define RCX_COMM_WORDS 5
__global__ void rcx_pack_sizes_k(const u64* __restrict__ offsets, u64 nblocks, u64 concat_cap, u64 table_cap, u64 has_table, u64* __restrict__ mine)
{
    mine[0] = offsets[nblocks]; // the encoder's exclusive prefix: offsets[nblocks] = segment bytes
    mine[1] = nblocks;
    mine[2] = concat_cap;
    mine[3] = table_cap;
    mine[4] = has_table;
}

// this rank's part of the global table: local offsets shifted by the segment's base; the last rank also writes
// the closing entry (= total bytes)
__global__ void rcx_shift_offsets_k(const u64* __restrict__ offsets, u64 nblocks, u64 seg_base, u64* __restrict__ part, u64 closing,
                                    int write_closing)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nblocks) part[i] = offsets[i] + seg_base;
    if (i == 0 && write_closing) part[nblocks] = closing;
}
} // namespace

extern "C" {

int rcx_exchange_plan(const uint64_t* seg_bytes, const uint64_t* nblocks, int nranks, uint64_t* seg_base, uint64_t* block_base)
{
    if (!seg_bytes || !nblocks || nranks <= 0 || !seg_base || !block_base) return RCX_E_ARG;
    seg_base[0] = 0;
    block_base[0] = 0;
    for (int r = 0; r < nranks; ++r) {
        seg_base[r + 1] = seg_base[r] + seg_bytes[r];
        block_base[r + 1] = block_base[r] + nblocks[r];
        if (seg_base[r + 1] < seg_base[r] || block_base[r + 1] < block_base[r]) return RCX_E_ARG; // wrapped
    }
    return RCX_OK;
}

int rcx_comm_unique_id(void* id)
{
    if (!id) return RCX_E_ARG;
    static_assert(RCX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "rcx.h and rccl.h disagree on the id size");
    ncclUniqueId u;
    NCCL_TRY(ncclGetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return RCX_OK;
}

int rcx_comm_create(int device, const void* id, int nranks, int rank, rcx_comm** out)
{
    if (!out) return RCX_E_ARG;
    *out = nullptr;
    if (!id || nranks <= 0 || rank < 0 || rank >= nranks) return RCX_E_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return RCX_E_HIP;
    if (device < 0 || device >= count) return RCX_E_ARG;
    HIP_TRY(hipSetDevice(device));
    rcx_comm* c = new (std::nothrow) rcx_comm();
    if (!c) return RCX_E_NOMEM;
    c->device = device;
    c->nranks = nranks;
    c->rank = rank;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    if (ncclCommInitRank(&c->nccl, nranks, u, rank) != ncclSuccess) {
        c->nccl = nullptr;
        rcx_comm_destroy(c);
        return RCX_E_COMM;
    }
    if (hipMalloc(reinterpret_cast<void**>(&c->d_mine), RCX_COMM_WORDS * sizeof(u64)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_every), RCX_COMM_WORDS * sizeof(u64) * nranks) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&c->h_every), RCX_COMM_WORDS * sizeof(u64) * nranks, hipHostMallocDefault) != hipSuccess) {
        rcx_comm_destroy(c);
        return RCX_E_NOMEM;
    }
    *out = c;
    return RCX_OK;
}

void rcx_comm_destroy(rcx_comm* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->nccl) (void)ncclCommDestroy(c->nccl);
    if (c->d_mine) (void)hipFree(c->d_mine);
    if (c->d_every) (void)hipFree(c->d_every);
    if (c->h_every) (void)hipHostFree(c->h_every);
    delete c;
}

int rcx_comm_rank(const rcx_comm* c) { return c ? c->rank : -1; }
int rcx_comm_size(const rcx_comm* c) { return c ? c->nranks : 0; }

int rcx_allgatherv_segments(rcx_comm* c, const void* d_segment, const uint64_t* d_offsets, uint64_t nblocks,
                            void* d_concat, uint64_t concat_cap, uint64_t* d_table, uint64_t table_cap,
                            uint64_t* seg_base_out, uint64_t* block_base_out, void* stream)
{
    if (!c || !d_offsets || !d_concat || (nblocks && !d_segment)) return RCX_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)hipGetLastError(); // (what an earlier call of this thread left behind is not this call's: see rcx_enter_device in rcx_api.hip)
    HIP_TRY(hipSetDevice(c->device));
    const int n = c->nranks;
    // 1. what does every rank bring?  (the one host synchronisation of the exchange: send and receive counts are
    //    host arguments of the point-to-point calls)
    hipLaunchKernelGGL(rcx_pack_sizes_k, dim3(1), dim3(1), 0, s, d_offsets, (u64)nblocks, (u64)concat_cap, (u64)table_cap, (u64)(d_table ? 1 : 0), c->d_mine);
    NCCL_TRY(ncclAllGather(c->d_mine, c->d_every, RCX_COMM_WORDS, ncclUint64, c->nccl, s));
    HIP_TRY(hipMemcpyAsync(c->h_every, c->d_every, RCX_COMM_WORDS * sizeof(u64) * n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    // (four small host arrays and the synchronisation above per call: nothing next to a 17 ms step)
    std::vector<u64> seg(n), blk(n), seg_base(n + 1), block_base(n + 1);
    u64 least_concat = ~0ull, least_table = ~0ull, tables = 0;
    for (int r = 0; r < n; ++r) {
        const u64* w = c->h_every + RCX_COMM_WORDS * r;
        seg[r] = w[0];
        blk[r] = w[1];
        if (w[2] < least_concat) least_concat = w[2];
        if (w[3] < least_table) least_table = w[3];
        tables += w[4] ? 1 : 0;
    }
    int st = rcx_exchange_plan(seg.data(), blk.data(), n, seg_base.data(), block_base.data());
    if (st != RCX_OK) return st;
    if (seg_base_out) memcpy(seg_base_out, seg_base.data(), sizeof(u64) * (n + 1));
    if (block_base_out) memcpy(block_base_out, block_base.data(), sizeof(u64) * (n + 1));
    // every rank sees the same plan AND the same capacities: they all stop here together, before anything moves
    if (tables != 0 && tables != (u64)n) return RCX_E_ARG; // some ranks want the table and some do not: unmatched messages
    if (seg_base[n] > least_concat) return RCX_E_CAPACITY;
    if (d_table && block_base[n] + 1 > least_table) return RCX_E_CAPACITY;
    // 2. own contribution in place
    u8* concat = static_cast<u8*>(d_concat);
    if (seg[c->rank]) HIP_TRY(hipMemcpyAsync(concat + seg_base[c->rank], d_segment, seg[c->rank], hipMemcpyDeviceToDevice, s));
    if (d_table) {
        const u32 grid = (u32)((nblocks + 255) / 256);
        hipLaunchKernelGGL(rcx_shift_offsets_k, dim3(grid ? grid : 1), dim3(256), 0, s, d_offsets, (u64)nblocks, seg_base[c->rank],
                           d_table + block_base[c->rank], seg_base[n], c->rank == n - 1 ? 1 : 0);
    }
    // 3. everybody else's, point to point, straight into place
    if (n > 1) {
        NCCL_TRY(ncclGroupStart());
        bool ok = true; // (a call that fails inside the group still closes it)
        for (int step = 1; step < n && ok; ++step) {
            const int to = (c->rank + step) % n, from = (c->rank - step + n) % n; // stagger the peers over the links
            if (seg[c->rank]) ok = ok && ncclSend(d_segment, seg[c->rank], ncclUint8, to, c->nccl, s) == ncclSuccess;
            if (seg[from]) ok = ok && ncclRecv(concat + seg_base[from], seg[from], ncclUint8, from, c->nccl, s) == ncclSuccess;
            if (d_table) {
                // the table parts as shifted by their owners; the last rank's part carries the closing entry
                const u64 mine = blk[c->rank] + (c->rank == n - 1 ? 1 : 0), theirs = blk[from] + (from == n - 1 ? 1 : 0);
                if (mine) ok = ok && ncclSend(d_table + block_base[c->rank], mine, ncclUint64, to, c->nccl, s) == ncclSuccess;
                if (theirs) ok = ok && ncclRecv(d_table + block_base[from], theirs, ncclUint64, from, c->nccl, s) == ncclSuccess;
            }
        }
        const bool closed = ncclGroupEnd() == ncclSuccess;
        if (!ok || !closed) return RCX_E_COMM;
    }
    return hipGetLastError() == hipSuccess ? RCX_OK : RCX_E_HIP;
}

} // extern "C"
