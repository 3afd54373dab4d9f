// Host <-> device transfers of the host-buffer entry points (the Writer / Reader handles, ssw_copy_*).
//
// What a single-image caller pays is the copy, not the device work (one 4K frame: 1.6 ms of kernels
// against 25 MB of 8-bit or 99.5 MB of f32 pixels each way), and a pageable hipMemcpy moves ~15 GB/s on
// this host.  This file replaces it with
//   * a direct DMA when the caller's buffer is pinned (ssw_host_alloc, hipHostMalloc, hipHostRegister),
//   * otherwise a ring of pinned staging buffers filled by a few host threads in 1 MiB pieces, the DMA of
//     an 8 MiB slice starting as soon as its pieces are in (upload), or the pieces of a slice being copied
//     out as soon as its DMA has landed (download) -- host copy and PCIe run concurrently.
// An upload returns when the caller's buffer may be reused, NOT when the device has the data: that is
// ordered on the stream it was given (the staging buffer is guarded by an event).
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

#include "ssw_host.hpp"

namespace ssw {
namespace host {

namespace {
constexpr size_t SUB = (size_t)1 << 20;          // piece one host thread copies at a time
// one DMA per slice of SUBS_PER_SLICE pieces (8 MiB by default -- 4 .. 8 MiB measured best for 25 MB frames, smaller
// slices pay per-call overhead, larger ones start the DMA late; SSW_COPY_SLICE_MB: 1 .. 64)
static const size_t SUBS_PER_SLICE = [] {
    const char* e = std::getenv("SSW_COPY_SLICE_MB");
    size_t v = e ? (size_t)std::atoi(e) : 8;
    return v < 1 ? (size_t)1 : (v > 64 ? (size_t)64 : v);
}();
#define SLICE (SUB * SUBS_PER_SLICE)
constexpr size_t PIECE = (size_t)64 << 20;       // size of one pinned staging buffer
constexpr int RING = 3;
constexpr size_t SMALL = (size_t)256 << 10;      // below this a plain pageable copy is as fast

inline void cpu_relax() { __builtin_ia32_pause(); }
}  // namespace

struct Transfer {
    struct Pinned {
        char* p = nullptr;
        hipEvent_t busy = nullptr;     // completes when the DMA that last used the buffer is done
        bool pending = false;
    };
    Pinned ring[RING];
    unsigned next = 0;
    std::vector<hipEvent_t> slice_ev;  // download: one per slice of a piece

    // worker pool: one job at a time, posted by the context's host thread, which takes part itself
    struct Job {
        char* dst = nullptr;
        const char* src = nullptr;
        size_t bytes = 0, n_sub = 0;
        std::atomic<size_t> next_sub{0};
        std::atomic<size_t> ready_sub{0};              // pieces whose source is valid (download: DMA landed)
        std::atomic<uint32_t> done[PIECE / SUB];       // pieces finished, per slice
        std::atomic<int> users{0};
    };
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv;
    Job* job = nullptr;
    uint64_t seq = 0;
    bool quit = false;
    int n_threads = -1;                // helpers besides the caller's thread; -1 = not decided yet

    double stats[SSW_TRANSFER_STAT_COUNT] = {0};

    static void copy_sub(Job& j, size_t s) {
        const size_t off = s * SUB, n = std::min(SUB, j.bytes - off);
        std::memcpy(j.dst + off, j.src + off, n);
        j.done[s / SUBS_PER_SLICE].fetch_add(1, std::memory_order_release);
    }
    void worker() {
        uint64_t seen = 0;
        for (;;) {
            Job* j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return quit || (job && seq != seen); });
                if (quit) return;
                seen = seq;
                j = job;
                j->users.fetch_add(1);
            }
            for (;;) {
                const size_t s = j->next_sub.fetch_add(1);
                if (s >= j->n_sub) break;
                while (s >= j->ready_sub.load(std::memory_order_acquire)) cpu_relax();
                copy_sub(*j, s);
            }
            j->users.fetch_sub(1);
        }
    }
    void ensure_workers() {
        if (n_threads < 0) {
            const char* e = std::getenv("SSW_COPY_THREADS");
            const int hw = (int)std::thread::hardware_concurrency();
            n_threads = e ? std::atoi(e) - 1 : std::min(3, std::max(hw / 2 - 1, 0));
            n_threads = std::max(0, std::min(n_threads, 31));
        }
        while ((int)workers.size() < n_threads) workers.emplace_back([this] { worker(); });
    }
    void post(Job& j) {
        ensure_workers();
        if (workers.empty()) return;
        {
            std::lock_guard<std::mutex> lk(m);
            job = &j;
            ++seq;
        }
        cv.notify_all();
    }
    void retire(Job& j) {
        {
            std::lock_guard<std::mutex> lk(m);
            job = nullptr;                              // late wakers find no job
        }
        while (j.users.load() != 0) cpu_relax();
    }
    void stop_workers() {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cv.notify_all();
        for (auto& t : workers) t.join();
        workers.clear();
        quit = false;
    }
};

namespace {

int get_transfer(ssw_ctx* ctx, Transfer** out) {
    if (!ctx->xfer) {
        ctx->xfer = new (std::nothrow) Transfer();
        if (!ctx->xfer) return SSW_ERR_OUT_OF_MEMORY;
    }
    *out = ctx->xfer;
    return SSW_OK;
}

int acquire(Transfer& t, Transfer::Pinned** out) {
    Transfer::Pinned& b = t.ring[t.next++ % RING];
    if (!b.p) {
        void* p = nullptr;
        const hipError_t e = hipHostMalloc(&p, PIECE, hipHostMallocDefault);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_last_error(std::string("hipHostMalloc(staging): ") + hipGetErrorString(e));
            return SSW_ERR_OUT_OF_MEMORY;
        }
        b.p = (char*)p;
        SSW_HIP_CHECK(hipEventCreateWithFlags(&b.busy, hipEventDisableTiming));
    }
    if (b.pending) { SSW_HIP_CHECK(hipEventSynchronize(b.busy)); b.pending = false; }
    *out = &b;
    return SSW_OK;
}

bool is_pinned(const void* p) {
    hipPointerAttribute_t a;
    std::memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

void transfer_destroy(ssw_ctx* ctx) {
    Transfer* t = ctx->xfer;
    if (!t) return;
    t->stop_workers();
    for (auto& b : t->ring) {
        if (b.pending) (void)hipEventSynchronize(b.busy);
        if (b.busy) (void)hipEventDestroy(b.busy);
        if (b.p) (void)hipHostFree(b.p);
    }
    for (auto& e : t->slice_ev) (void)hipEventDestroy(e);
    delete t;
    ctx->xfer = nullptr;
}

int transfer_set_threads(ssw_ctx* ctx, int threads) {
    Transfer* t = nullptr;
    SSW_TRY(get_transfer(ctx, &t));
    t->stop_workers();
    t->n_threads = threads <= 0 ? -1 : std::min(threads - 1, 31);
    return SSW_OK;
}

int transfer_stats(ssw_ctx* ctx, double* out, bool reset) {
    Transfer* t = nullptr;
    SSW_TRY(get_transfer(ctx, &t));
    for (int i = 0; i < SSW_TRANSFER_STAT_COUNT; ++i) {
        if (out) out[i] = t->stats[i];
        if (reset) t->stats[i] = 0;
    }
    return SSW_OK;
}

// host -> device on `st`.  Returns once `host_src` may be reused by the caller.
int upload(ssw_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return SSW_OK;
    Transfer* t = nullptr;
    SSW_TRY(get_transfer(ctx, &t));
    const double t0 = now_s();
    t->stats[SSW_TRANSFER_H2D_BYTES] += (double)bytes;
    if (bytes < SMALL || is_pinned(host_src)) {
        SSW_HIP_CHECK(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, st));
        SSW_HIP_CHECK(hipStreamSynchronize(st));        // the caller's buffer is the DMA source
        t->stats[SSW_TRANSFER_H2D_SECONDS] += now_s() - t0;
        if (bytes >= SMALL) t->stats[SSW_TRANSFER_DIRECT_BYTES] += (double)bytes;
        return SSW_OK;
    }
    for (size_t p0 = 0; p0 < bytes; p0 += PIECE) {
        const size_t pb = std::min(PIECE, bytes - p0);
        Transfer::Pinned* pin = nullptr;
        SSW_TRY(acquire(*t, &pin));
        Transfer::Job j;
        j.dst = pin->p; j.src = (const char*)host_src + p0; j.bytes = pb;
        j.n_sub = (pb + SUB - 1) / SUB;
        j.ready_sub.store(j.n_sub);
        const size_t n_slices = (pb + SLICE - 1) / SLICE;
        for (size_t s = 0; s < n_slices; ++s) j.done[s].store(0);
        t->post(j);
        size_t issued = 0;
        hipError_t err = hipSuccess;
        while (issued < n_slices) {
            const size_t subs = std::min(SUBS_PER_SLICE, j.n_sub - issued * SUBS_PER_SLICE);
            if (j.done[issued].load(std::memory_order_acquire) == subs) {
                const size_t off = issued * SLICE, n = std::min(SLICE, pb - off);
                if (err == hipSuccess)
                    err = hipMemcpyAsync((char*)dev_dst + p0 + off, pin->p + off, n, hipMemcpyHostToDevice, st);
                ++issued;
                continue;
            }
            const size_t s = j.next_sub.fetch_add(1);
            if (s < j.n_sub) Transfer::copy_sub(j, s);
            else cpu_relax();
        }
        t->retire(j);
        SSW_HIP_CHECK(err);
        SSW_HIP_CHECK(hipEventRecord(pin->busy, st));
        pin->pending = true;
    }
    t->stats[SSW_TRANSFER_H2D_SECONDS] += now_s() - t0;
    t->stats[SSW_TRANSFER_STAGED_BYTES] += (double)bytes;
    return SSW_OK;
}

int upload_nowait(ssw_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes, hipStream_t st, bool* async_out) {
    *async_out = false;
    if (bytes == 0) return SSW_OK;
    if (!is_pinned(host_src)) return upload(ctx, dev_dst, host_src, bytes, st);
    Transfer* t = nullptr;
    SSW_TRY(get_transfer(ctx, &t));
    t->stats[SSW_TRANSFER_H2D_BYTES] += (double)bytes;
    t->stats[SSW_TRANSFER_DIRECT_BYTES] += (double)bytes;
    SSW_HIP_CHECK(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, st));
    *async_out = true;
    return SSW_OK;
}

int download_nowait(ssw_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes, hipStream_t st, bool* async_out) {
    *async_out = false;
    if (bytes == 0) return SSW_OK;
    if (!is_pinned(host_dst)) return download(ctx, host_dst, dev_src, bytes, st);
    Transfer* t = nullptr;
    SSW_TRY(get_transfer(ctx, &t));
    t->stats[SSW_TRANSFER_D2H_BYTES] += (double)bytes;
    t->stats[SSW_TRANSFER_DIRECT_BYTES] += (double)bytes;
    SSW_HIP_CHECK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, st));
    *async_out = true;
    return SSW_OK;
}

// device -> host on `st`.  Returns when `host_dst` holds the data.
int download(ssw_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return SSW_OK;
    Transfer* t = nullptr;
    SSW_TRY(get_transfer(ctx, &t));
    const double t0 = now_s();
    t->stats[SSW_TRANSFER_D2H_BYTES] += (double)bytes;
    if (bytes < SMALL || is_pinned(host_dst)) {
        SSW_HIP_CHECK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, st));
        SSW_HIP_CHECK(hipStreamSynchronize(st));
        t->stats[SSW_TRANSFER_D2H_SECONDS] += now_s() - t0;
        if (bytes >= SMALL) t->stats[SSW_TRANSFER_DIRECT_BYTES] += (double)bytes;
        return SSW_OK;
    }
    const size_t SLICES = PIECE / SUB;                 // event slots per staging buffer (>= slices of a piece)
    while (t->slice_ev.size() < SLICES * RING) {
        hipEvent_t e = nullptr;
        SSW_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync));
        t->slice_ev.push_back(e);
    }
    // In rounds of RING pieces (one staging buffer each): all DMAs of a round are enqueued first, slice by
    // slice with an event behind each, then the pieces are copied out as their slices land.
    for (size_t r0 = 0; r0 < bytes; r0 += PIECE * RING) {
        const size_t rb = std::min(PIECE * RING, bytes - r0);
        const size_t n_pieces = (rb + PIECE - 1) / PIECE;
        Transfer::Pinned* pins[RING] = {nullptr, nullptr, nullptr};
        for (size_t p = 0; p < n_pieces; ++p) SSW_TRY(acquire(*t, &pins[p]));
        for (size_t p = 0; p < n_pieces; ++p) {
            const size_t p0 = r0 + p * PIECE, pb = std::min(PIECE, bytes - p0);
            for (size_t s = 0; s * SLICE < pb; ++s) {
                const size_t off = s * SLICE, n = std::min(SLICE, pb - off);
                SSW_HIP_CHECK(hipMemcpyAsync(pins[p]->p + off, (const char*)dev_src + p0 + off, n, hipMemcpyDeviceToHost, st));
                SSW_HIP_CHECK(hipEventRecord(t->slice_ev[p * SLICES + s], st));
            }
        }
        for (size_t p = 0; p < n_pieces; ++p) {
            const size_t p0 = r0 + p * PIECE, pb = std::min(PIECE, bytes - p0);
            const size_t n_slices = (pb + SLICE - 1) / SLICE;
            const hipEvent_t* ev = &t->slice_ev[p * SLICES];
            Transfer::Job j;
            j.dst = (char*)host_dst + p0; j.src = pins[p]->p; j.bytes = pb;
            j.n_sub = (pb + SUB - 1) / SUB;
            j.ready_sub.store(0);
            for (size_t s = 0; s < n_slices; ++s) j.done[s].store(0);
            // Sleep (blocking event) until the piece's first slice has landed: whatever the stream still has to run ahead
            // of the DMA -- a whole batch call, possibly -- is waited for here by one sleeping thread; the caller and the
            // copy threads used to spin through all of it.  On an error the stream is drained before the
            // staging buffers go back to the ring.
            {
                const hipError_t e0 = hipEventSynchronize(ev[0]);
                if (e0 != hipSuccess) {
                    (void)hipStreamSynchronize(st);
                    (void)hipGetLastError();
                    SSW_HIP_CHECK(e0);
                }
            }
            t->post(j);
            size_t synced = 0;
            constexpr size_t NONE = ~(size_t)0;
            size_t mine = NONE;
            bool exhausted = false;
            hipError_t err = hipSuccess;
            for (;;) {
                if (synced < n_slices) {
                    const hipError_t q = hipEventQuery(ev[synced]);
                    if (q == hipSuccess) {
                        ++synced;
                        j.ready_sub.store(std::min(synced * SUBS_PER_SLICE, j.n_sub), std::memory_order_release);
                    } else if (q != hipErrorNotReady) {
                        err = q;                            // drain the stream (no DMA may still be writing the staging
                        (void)hipStreamSynchronize(st);     // buffer), let the helpers run out, then report
                        synced = n_slices;
                        j.ready_sub.store(j.n_sub, std::memory_order_release);
                    }
                }
                if (mine == NONE && !exhausted) {
                    const size_t s = j.next_sub.fetch_add(1);
                    if (s < j.n_sub) mine = s; else exhausted = true;
                }
                if (mine != NONE && mine < j.ready_sub.load(std::memory_order_acquire)) {
                    Transfer::copy_sub(j, mine);
                    mine = NONE;
                } else if (exhausted && synced == n_slices) {
                    break;
                } else {
                    cpu_relax();
                }
            }
            t->retire(j);
            (void)hipGetLastError();
            SSW_HIP_CHECK(err);
        }
    }
    t->stats[SSW_TRANSFER_D2H_SECONDS] += now_s() - t0;
    t->stats[SSW_TRANSFER_STAGED_BYTES] += (double)bytes;
    return SSW_OK;
}

}  // namespace host
}  // namespace ssw
