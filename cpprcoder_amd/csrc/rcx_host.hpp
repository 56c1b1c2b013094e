// rcx_host.hpp -- the host-buffer entry points of rcx.h as a pipeline (included by rcx_api.hip).
//
// The reference is driven with host memory on both sides (test/main.cpp:321-350: a file in a malloc'd buffer in, a
// MemoryStream out), so a drop-in caller of rcx_encode_blocks / rcx_decode_blocks / rcx_bwt_encode / rcx_bwt_decode
// sees link + kernels, not kernels.  One call is cut into chunks of whole blocks and three things run at once:
//   feeders   host threads that move chunk k+1 .. to the GPU, each on a copy stream of its own,
//   the caller's thread, which launches chunk k's kernels on one of a few work streams as soon as its bytes are
//             there (a chunk's launches are `packed`, rcx_api.hip: chunks share the machine, because a block is a
//             serial chain and a chunk's kernels take as long as a whole buffer's would),
//   drainers  host threads that bring chunk k-1's result back, all on one copy-back stream.
// The stages hand over on the HOST (a feeder waits for its own copy, the caller's thread for the feeders, a drainer
// for the chunk's last launch): no stream waits for another stream's event on the device.
//
// What shapes it (measured on the box; profiles/r03_host_pipeline.md has the timelines):
// * The runtime gives a process four hardware queues, one of them the null stream's; streams beyond that SHARE a queue
//   and their work runs in turn.  Host-to-device copies go through the DMA engines, but device-to-host copies are
//   kernels (__amd_rocclr_copyBuffer) that wait their turn in their stream's hardware queue.  So there are at most
//   three work streams and ONE stream for all copies back, made first; the feeders' streams only carry DMA copies.
// * Work streams that wait for an event of a copy stream (the first version) ran their chunks in turns, not side by
//   side; so does a copy-back stream that shares a queue with a work stream.
// * hipMemcpyAsync to PAGEABLE memory returns when the copy is done and keeps other threads' HIP calls waiting
//   meanwhile (the first chunk's copy back stalled the feeder for 7 ms of a 30 ms call); from pageable memory it costs
//   the feeder nothing comparable, and one feeder fills the link.  So by default pieces go in `direct` (the caller's
//   memory handed to hipMemcpyAsync, pinned by the runtime on the fly); an encoder's output comes back `staged` (into
//   one of two pinned slots per drainer, then memcpy by that drainer -- which is also what touches a fresh
//   destination's pages, a few threads at once instead of one page fault at a time inside the runtime); a decoder's
//   output, which only starts to come when the input is nearly in, comes back `direct` (three drainers' memcpy is
//   slower than the link, and there the copy back is what is left at the end).  `register` (hipHostRegister on the
//   caller's pieces first) is the third way; RCX_HOST_IN / RCX_HOST_OUT / RCX_HOST_MODE = direct | staged | register,
//   RCX_HOST_FEEDERS / RCX_HOST_DRAINERS / RCX_HOST_MOVERS, RCX_HOST_PIECE_MIB and RCX_HOST_WORK_STREAMS override
//   (tools/diag/host_sweep.sh; profiles/r03_host_rate.jsonl).
// * With W chunks' kernels in flight at most, a chunk must take the link at least 1/W as long as its kernels run
//   (host_chunk_blocks); a call then takes about  link time of the buffer + one chunk's kernels + one chunk's copy back.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <functional>
#include <mutex>
#include <thread>



namespace
{

enum { RCX_HOST_DIRECT = 0, RCX_HOST_STAGED = 1, RCX_HOST_REGISTER = 2 };
#define RCX_HOST_WORK_STREAMS 3     /* default */
#define RCX_HOST_MAX_WORK_STREAMS 4
#define RCX_HOST_MAX_MOVERS 8

struct HostPipe {
    hipStream_t work[RCX_HOST_MAX_WORK_STREAMS] = {};
    int work_streams = RCX_HOST_WORK_STREAMS;
    hipStream_t out_stream = nullptr;
    hipStream_t in_streams[RCX_HOST_MAX_MOVERS] = {};
    int in_mode = RCX_HOST_DIRECT, out_mode = RCX_HOST_STAGED;
    int out_mode_decode = RCX_HOST_DIRECT; // (the decoders' output: see the top of the file)
    int feeders = 1, drainers = 3; // host threads per direction
    u64 piece = 16ull << 20; // bytes per copy
    u8* pin = nullptr;       // staged: two slots of `piece` bytes per feeder, then two per drainer
    u64 pin_bytes = 0;
    u64* words = nullptr;    // pinned: what the caller's thread needs back from a chunk (its offsets, a count)
    u64 words_count = 0;
    u64 bwt_ties = 0;        // periodic blocks of the last host-buffer block sort (all chunks)
    bool bwt_ties_valid = false;
};

void host_pipe_destroy(HostPipe* p)
{
    if (!p) return;
    for (auto& s : p->in_streams)
        if (s) (void)hipStreamDestroy(s);
    if (p->out_stream) (void)hipStreamDestroy(p->out_stream);
    for (auto& s : p->work)
        if (s) (void)hipStreamDestroy(s);
    if (p->pin) (void)hipHostFree(p->pin);
    if (p->words) (void)hipHostFree(p->words);
    delete p;
}

int host_pipe_get(rcx_ctx* c, HostPipe** out)
{
    if (c->pipe) {
        *out = c->pipe;
        return RCX_OK;
    }
    HostPipe* p = new (std::nothrow) HostPipe();
    if (!p) return RCX_E_NOMEM;
    bool ok = true;
    if (const char* v = getenv("RCX_HOST_WORK_STREAMS")) {
        const int w = atoi(v);
        if (w >= 1 && w <= RCX_HOST_MAX_WORK_STREAMS) p->work_streams = w;
    }
    for (int w = 0; w < p->work_streams; ++w) ok = ok && hipStreamCreateWithFlags(&p->work[w], hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&p->out_stream, hipStreamNonBlocking) == hipSuccess;
    auto mode_of = [](const char* v, int otherwise) {
        if (!v) return otherwise;
        if (!strcmp(v, "staged")) return (int)RCX_HOST_STAGED;
        if (!strcmp(v, "register")) return (int)RCX_HOST_REGISTER;
        if (!strcmp(v, "direct")) return (int)RCX_HOST_DIRECT;
        return otherwise;
    };
    p->in_mode = mode_of(getenv("RCX_HOST_MODE"), p->in_mode);
    p->out_mode = mode_of(getenv("RCX_HOST_MODE"), p->out_mode);
    p->out_mode_decode = mode_of(getenv("RCX_HOST_MODE"), p->out_mode_decode);
    p->in_mode = mode_of(getenv("RCX_HOST_IN"), p->in_mode);
    p->out_mode = mode_of(getenv("RCX_HOST_OUT"), p->out_mode);
    p->out_mode_decode = mode_of(getenv("RCX_HOST_OUT"), p->out_mode_decode);
    auto count_of = [](const char* v, int otherwise) {
        const int m = v ? atoi(v) : 0;
        return m >= 1 && m <= RCX_HOST_MAX_MOVERS ? m : otherwise;
    };
    p->feeders = count_of(getenv("RCX_HOST_MOVERS"), p->feeders);
    p->drainers = count_of(getenv("RCX_HOST_MOVERS"), p->drainers);
    p->feeders = count_of(getenv("RCX_HOST_FEEDERS"), p->feeders);
    p->drainers = count_of(getenv("RCX_HOST_DRAINERS"), p->drainers);
    if (const char* v = getenv("RCX_HOST_PIECE_MIB")) {
        const int m = atoi(v);
        if (m >= 1 && m <= 256) p->piece = (u64)m << 20;
    }
    for (int t = 0; t < p->feeders; ++t) ok = ok && hipStreamCreateWithFlags(&p->in_streams[t], hipStreamNonBlocking) == hipSuccess;
    if (!ok) {
        if (getenv("RCX_DEBUG")) fprintf(stderr, "rcx_host: could not make the pipe's streams (last HIP error: %s)\n", hipGetErrorString(hipGetLastError()));
        host_pipe_destroy(p);
        return RCX_E_HIP;
    }
    c->pipe = p;
    *out = p;
    return RCX_OK;
}

int host_pipe_words(HostPipe* p, u64 count)
{
    if (p->words_count >= count) return RCX_OK;
    if (p->words) (void)hipHostFree(p->words);
    p->words = nullptr;
    p->words_count = 0;
    if (hipHostMalloc(reinterpret_cast<void**>(&p->words), count * sizeof(u64), hipHostMallocDefault) != hipSuccess) return RCX_E_NOMEM;
    p->words_count = count;
    return RCX_OK;
}

int host_pipe_pin(HostPipe* p)
{
    const bool out_staged = p->out_mode == RCX_HOST_STAGED || p->out_mode_decode == RCX_HOST_STAGED;
    const u64 want = ((p->in_mode == RCX_HOST_STAGED ? 2ull * p->feeders : 0) + (out_staged ? 2ull * p->drainers : 0)) * p->piece;
    if (p->pin_bytes >= want) return RCX_OK;
    if (p->pin) (void)hipHostFree(p->pin);
    p->pin = nullptr;
    p->pin_bytes = 0;
    if (hipHostMalloc(reinterpret_cast<void**>(&p->pin), want, hipHostMallocDefault) != hipSuccess) return RCX_E_NOMEM;
    p->pin_bytes = want;
    return RCX_OK;
}

// One piece of work for a mover: `bytes` from `from` to `to`, one of them the caller's memory.
struct HostSpan {
    const u8* from = nullptr;
    u8* to = nullptr;
    u64 bytes = 0;
};

// What a call is, chunk by chunk.  `in` and `launch` are called by the caller's thread, `out` by whichever drainer
// gets to chunk k first -- once per chunk, in chunk order, after the chunk's launches have finished.
struct HostJob {
    u64 chunks = 0;
    int work_streams = 0;                            // at most so many chunks' kernels at a time (0 = as many as the pipe has streams)
    std::function<HostSpan(u64)> in;                 // caller's bytes -> device
    std::function<int(u64, hipStream_t)> launch;     // the chunk's kernels (+ small copies into HostPipe::words)
    std::function<int(u64, HostSpan*)> out;          // device -> caller's bytes; may return an error
    bool decode = false;                             // the output is a decoder's (HostPipe::out_mode_decode)
    const void* caller_in = nullptr;                 // the caller's buffers (to see whether they are pinned already)
    const void* caller_out = nullptr;
};

struct HostRun {
    rcx_ctx* c;
    HostPipe* p;
    const HostJob* job;
    std::mutex m;
    std::condition_variable cv;
    int err = RCX_OK;
    std::vector<HostSpan> ins;       // per chunk
    std::vector<u64> in_first_piece; // prefix: global piece index of a chunk's first piece
    std::vector<u32> fed;            // pieces of chunk k whose copy has been enqueued
    u64 launched = 0;                // chunks whose launches have been enqueued and whose `done` event is recorded
    std::vector<hipEvent_t> done;
    std::vector<HostSpan> outs;
    u64 outs_ready = 0;
    std::vector<std::pair<void*, u64>> registered; // register mode: the caller's ranges pinned for this call
    int in_mode = RCX_HOST_DIRECT, out_mode = RCX_HOST_DIRECT; // this call's ways across the link (host_run decides)
    // RCX_HOST_TRACE=1: when each stage of each chunk happened, to stderr at the end of the call (diagnostic)
    bool trace = false;
    std::chrono::steady_clock::time_point t0;
    struct Mark { const char* what; int thread; u64 chunk; u64 piece; double ms; };
    std::vector<Mark> marks;
    void mark(const char* what, int thread, u64 chunk, u64 piece)
    {
        if (!trace) return;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::lock_guard<std::mutex> g(m);
        marks.push_back(Mark{what, thread, chunk, piece, ms});
    }

    void fail(int e, const char* where = "")
    {
        if (getenv("RCX_DEBUG")) fprintf(stderr, "rcx_host: stage failed with %d %s (last HIP error: %s)\n", e, where, hipGetErrorString(hipGetLastError()));
        std::lock_guard<std::mutex> g(m);
        if (err == RCX_OK) err = e;
        cv.notify_all();
    }
    bool failed()
    {
        std::lock_guard<std::mutex> g(m);
        return err != RCX_OK;
    }
};

u64 host_pieces(u64 bytes, u64 piece) { return bytes ? (bytes + piece - 1) / piece : 1; } // (an empty chunk still counts as one)

// Pins [ptr, ptr + bytes) of the caller's memory for this call (register mode).
bool host_register(HostRun* r, const void* ptr, u64 bytes)
{
    if (bytes == 0) return true;
    if (hipHostRegister(const_cast<void*>(ptr), bytes, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError(); // (already registered by the caller, or not registrable: the copy below still works)
        return true;
    }
    std::lock_guard<std::mutex> g(r->m);
    r->registered.emplace_back(const_cast<void*>(ptr), bytes);
    return true;
}

// A feeder: every movers-th piece of the input, in order.  A piece counts as fed once its copy has FINISHED (the
// feeder waits for its own stream), so the caller's thread launches a chunk's kernels without any wait on the device.
// Staged mode keeps two pinned slots: the memcpy of one piece runs while the other's copy is on the link.
void host_feeder(HostRun* r, int t)
{
    HostPipe* p = r->p;
    if (hipSetDevice(r->c->device) != hipSuccess) return r->fail(RCX_E_HIP);
    const bool staged = r->in_mode == RCX_HOST_STAGED;
    hipStream_t s = p->in_streams[t];
    u8* slot[2] = {staged ? p->pin + ((u64)t * 2 + 0) * p->piece : nullptr, staged ? p->pin + ((u64)t * 2 + 1) * p->piece : nullptr};
    u64 turn = 0;
    bool flying = false; // a copy of this feeder is on its way ...
    u64 flying_chunk = 0; // ... for this chunk
    auto landed = [&]() -> bool {
        if (!flying) return true;
        if (hipStreamSynchronize(s) != hipSuccess) return false;
        flying = false;
        r->mark("fed", t, flying_chunk, 0);
        std::lock_guard<std::mutex> g(r->m);
        r->fed[flying_chunk] += 1;
        r->cv.notify_all();
        return true;
    };
    bool ok = true;
    for (u64 k = 0; k < r->job->chunks && ok && !r->failed(); ++k) {
        const HostSpan& span = r->ins[k];
        const u64 pieces = host_pieces(span.bytes, p->piece);
        for (u64 j = 0; j < pieces && ok; ++j) {
            if ((r->in_first_piece[k] + j) % (u64)p->feeders != (u64)t) continue;
            const u64 at = j * p->piece;
            const u64 len = span.bytes - at < p->piece ? span.bytes - at : p->piece;
            r->mark("feed", t, k, j);
            if (staged && len) memcpy(slot[turn & 1], span.from + at, len); // (under the copy of the piece before)
            ok = landed();
            if (!ok) break;
            if (len) {
                if (r->in_mode == RCX_HOST_REGISTER) host_register(r, span.from + at, len);
                ok = hipMemcpyAsync(span.to + at, staged ? slot[turn & 1] : span.from + at, len, hipMemcpyHostToDevice, s) == hipSuccess;
                ++turn;
            }
            flying = true;
            flying_chunk = k;
        }
    }
    if (ok) ok = landed();
    if (!ok) r->fail(RCX_E_HIP, "in a feeder");
}

// A drainer: waits (on the host) for a chunk's last launch, then brings every movers-th piece of its output back.
// All drainers use the one copy-back stream (see the top of the file).
void host_drainer(HostRun* r, int t)
{
    HostPipe* p = r->p;
    if (hipSetDevice(r->c->device) != hipSuccess) return r->fail(RCX_E_HIP);
    const int out_mode = r->out_mode;
    const bool staged = out_mode == RCX_HOST_STAGED;
    hipStream_t s = p->out_stream;
    u8* const mine = staged ? p->pin + ((p->in_mode == RCX_HOST_STAGED ? 2ull * p->feeders : 0) + 2ull * t) * p->piece : nullptr;
    u8* slot[2] = {mine, staged ? mine + p->piece : nullptr};
    hipEvent_t there[2] = {nullptr, nullptr}; // staged: the piece is in its slot
    if (staged && (hipEventCreateWithFlags(&there[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&there[1], hipEventDisableTiming) != hipSuccess))
        return r->fail(RCX_E_HIP);
    HostSpan waiting; // staged: the piece on its way into a slot (from = the slot), to be copied on to the caller's memory
    int waiting_slot = 0;
    auto land = [&]() -> bool {
        if (!waiting.bytes) return true;
        if (hipEventSynchronize(there[waiting_slot]) != hipSuccess) return false;
        memcpy(waiting.to, waiting.from, waiting.bytes);
        waiting.bytes = 0;
        return true;
    };
    u64 turn = 0, piece_no = 0;
    bool ok = true;
    for (u64 k = 0; k < r->job->chunks && ok; ++k) {
        {
            std::unique_lock<std::mutex> g(r->m);
            r->cv.wait(g, [&] { return r->err != RCX_OK || r->launched > k; });
            if (r->err != RCX_OK) break;
        }
        if (hipEventSynchronize(r->done[k]) != hipSuccess) {
            r->fail(RCX_E_HIP, "waiting for a chunk's launches");
            break;
        }
        r->mark("done", t, k, 0);
        HostSpan span;
        {
            std::lock_guard<std::mutex> g(r->m);
            if (r->err != RCX_OK) break;
            if (r->outs_ready == k) { // the first drainer here describes the chunk's output (and does its bookkeeping)
                const int e = r->job->out(k, &r->outs[k]);
                if (e != RCX_OK) {
                    if (r->err == RCX_OK) r->err = e;
                    r->cv.notify_all();
                    break;
                }
                r->outs_ready = k + 1;
            }
            span = r->outs[k];
        }
        const u64 pieces = span.bytes ? host_pieces(span.bytes, p->piece) : 0;
        for (u64 j = 0; j < pieces && ok; ++j, ++piece_no) {
            if (piece_no % (u64)p->drainers != (u64)t) continue;
            const u64 at = j * p->piece;
            const u64 len = span.bytes - at < p->piece ? span.bytes - at : p->piece;
            r->mark("drain", t, k, j);
            if (staged) { // this piece sets off for one slot, then the piece before goes on from the other to the caller's memory
                const int into = (int)(turn++ & 1);
                ok = hipMemcpyAsync(slot[into], span.from + at, len, hipMemcpyDeviceToHost, s) == hipSuccess && hipEventRecord(there[into], s) == hipSuccess;
                if (ok) ok = land();
                waiting = HostSpan{slot[into], span.to + at, len};
                waiting_slot = into;
            } else {
                if (out_mode == RCX_HOST_REGISTER) host_register(r, span.to + at, len);
                ok = hipMemcpyAsync(span.to + at, span.from + at, len, hipMemcpyDeviceToHost, s) == hipSuccess;
            }
        }
    }
    if (ok) ok = land();
    if (ok && !staged) ok = hipStreamSynchronize(s) == hipSuccess;
    for (auto& e : there)
        if (e) (void)hipEventDestroy(e);
    if (!ok) r->fail(RCX_E_HIP, "in a drainer");
}

// Runs a job.  Returns the first error of any stage; the device is idle on return either way.
int host_run(rcx_ctx* c, HostPipe* p, const HostJob& job)
{
    HostRun r;
    r.c = c;
    r.p = p;
    r.job = &job;
    r.trace = getenv("RCX_HOST_TRACE") != nullptr;
    r.t0 = std::chrono::steady_clock::now();
    const u64 K = job.chunks;
    // Memory the caller has pinned itself (hipHostMalloc, hipHostRegister) needs neither staging nor pinning: the copies are
    // plain DMA and the calls return at once.
    auto pinned = [](const void* ptr) {
        if (!ptr) return false;
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, ptr) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        return a.type == hipMemoryTypeHost;
    };
    r.in_mode = p->in_mode;
    r.out_mode = job.decode ? p->out_mode_decode : p->out_mode;
    if (pinned(job.caller_in)) r.in_mode = RCX_HOST_DIRECT;
    if (pinned(job.caller_out)) r.out_mode = RCX_HOST_DIRECT;
    r.ins.resize(K);
    r.in_first_piece.resize(K + 1);
    r.fed.assign(K, 0);
    r.outs.resize(K);
    r.done.assign(K, nullptr);
    int rc = host_pipe_pin(p);
    if (rc != RCX_OK) return rc;
    if (r.trace) fprintf(stderr, "rcx_host modes: in %d out %d (0 direct, 1 staged, 2 register)\n", r.in_mode, r.out_mode);
    u64 pieces = 0;
    for (u64 k = 0; k < K; ++k) {
        r.ins[k] = job.in(k);
        r.in_first_piece[k] = pieces;
        pieces += host_pieces(r.ins[k].bytes, p->piece);
    }
    r.in_first_piece[K] = pieces;
    std::vector<hipEvent_t> began(r.trace ? K : 0, nullptr); // (trace: when the device started on a chunk's launches)
    for (u64 k = 0; k < K; ++k) {
        if (hipEventCreateWithFlags(&r.done[k], r.trace ? hipEventDefault : hipEventDisableTiming) != hipSuccess) {
            if (getenv("RCX_DEBUG")) fprintf(stderr, "rcx_host: could not make an event (last HIP error: %s)\n", hipGetErrorString(hipGetLastError()));
            rc = RCX_E_HIP;
        }
        if (r.trace && hipEventCreate(&began[k]) != hipSuccess) rc = RCX_E_HIP;
    }
    std::vector<std::thread> threads;
    if (rc == RCX_OK) {
        for (int t = 0; t < p->feeders; ++t) threads.emplace_back(host_feeder, &r, t);
        for (int t = 0; t < p->drainers; ++t) threads.emplace_back(host_drainer, &r, t);
        for (u64 k = 0; k < K; ++k) {
            const u32 need = (u32)(r.in_first_piece[k + 1] - r.in_first_piece[k]);
            {
                std::unique_lock<std::mutex> g(r.m);
                r.cv.wait(g, [&] { return r.err != RCX_OK || r.fed[k] == need; });
                if (r.err != RCX_OK) break;
            }
            const int W = job.work_streams && job.work_streams < p->work_streams ? job.work_streams : p->work_streams;
            hipStream_t s = p->work[k % (u64)W];
            r.mark("launch>", -1, k, 0);
            if (r.trace) (void)hipEventRecord(began[k], s);
            int e = job.launch(k, s);
            if (e == RCX_OK && hipEventRecord(r.done[k], s) != hipSuccess) e = RCX_E_HIP;
            r.mark("launch<", -1, k, 0);
            if (e != RCX_OK) {
                r.fail(e, "launching a chunk");
                break;
            }
            std::lock_guard<std::mutex> g(r.m);
            r.launched = k + 1;
            r.cv.notify_all();
        }
        for (auto& t : threads) t.join();
    }
    // whatever happened, nothing of this call is in flight when it returns
    for (int t = 0; t < p->feeders; ++t) (void)hipStreamSynchronize(p->in_streams[t]);
    for (int i = 0; i < p->work_streams; ++i) (void)hipStreamSynchronize(p->work[i]);

    (void)hipStreamSynchronize(p->out_stream);
    r.mark("idle", -1, K, 0);
    for (auto& reg : r.registered) (void)hipHostUnregister(reg.first);
    for (u64 k = 0; r.trace && k < K; ++k) {
        float at = 0.f, took = 0.f;
        if (hipEventElapsedTime(&at, began[0], began[k]) == hipSuccess && hipEventElapsedTime(&took, began[k], r.done[k]) == hipSuccess)
            fprintf(stderr, "rcx_host device: chunk %3llu began %8.3f ms after chunk 0 and took %8.3f ms\n", (unsigned long long)k, at, took);
    }
    for (auto& e : began)
        if (e) (void)hipEventDestroy(e);
    for (auto& mk : r.marks) fprintf(stderr, "rcx_host %8.3f ms  %-8s thread %2d chunk %3llu piece %3llu\n", mk.ms, mk.what, mk.thread, (unsigned long long)mk.chunk, (unsigned long long)mk.piece);
    for (u64 k = 0; k < K; ++k)
        if (r.done[k]) (void)hipEventDestroy(r.done[k]);
    if (rc != RCX_OK) return rc;
    return r.err;
}

// Blocks per chunk.  A chunk's kernels take as long as ONE block's chain whatever the chunk holds (85 ns a symbol
// for the encoders, 170 ns for the decoders), and two chunks' kernels run at a time, so chunks follow each other
// without a gap when a chunk takes the link at least half that long: blocks x block bytes / 57 GB/s >= block bytes x
// 85 (170) ns / 2 -- about 2400 blocks for the encoders and 4800 for the decoders whatever the block size (packed
// launches: 64 blocks per encode workgroup, 16 per decode wave).  A buffer is cut into equal chunks of about that many
// blocks; small blocks get chunks of at least 16 MiB (the fixed cost of a copy).  RCX_HOST_ENC_CHUNK / RCX_HOST_DEC_CHUNK
// (blocks) override.
u64 host_chunk_blocks(u32 block, bool decode, u64 nblocks)
{
    u64 cb = decode ? 4096 : 2048;
    if (const char* v = getenv(decode ? "RCX_HOST_DEC_CHUNK" : "RCX_HOST_ENC_CHUNK")) {
        const long long want = atoll(v);
        if (want >= 64) cb = (u64)want;
    }
    const u64 floor_blocks = ((16ull << 20) + block - 1) / block;
    if (cb < floor_blocks) cb = floor_blocks;
    while ((nblocks + cb - 1) / cb > 2048) cb *= 2; // (events and bookkeeping per chunk)
    const u64 chunks = (nblocks + cb - 1) / cb;     // equal chunks: no short straggler at the end
    if (chunks > 1) cb = (nblocks + chunks - 1) / chunks;
    return (cb + 63) & ~(u64)63;
}

} // namespace
