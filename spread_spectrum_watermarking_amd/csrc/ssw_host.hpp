// Host-side helpers shared by ssw_lib.hip (C ABI, handles) and ssw_pipeline.hip (transform chains and the
// two-lane batch pipelines).
#pragma once

#include <functional>
#include <vector>

#include "ssw_internal.hpp"

namespace ssw {
namespace host {

// Makes the context's GPU current for the duration of one ABI call and restores the caller's device
// afterwards (a host thread may drive several contexts, or torch on another GPU).
struct DeviceGuard {
    int prev = -1, dev;
    explicit DeviceGuard(int d) : dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// DeviceGuard + "the context this thread is working for": dev_malloc() gives that context's pool of spare handle
// planes (up to SSW_PLANE_POOL_MB) back to the device and retries once before it reports out of memory.
struct CtxGuard {
    DeviceGuard dg;
    ssw_ctx* prev;
    explicit CtxGuard(ssw_ctx* ctx);
    ~CtxGuard();
};
// frees the spare planes of the context's pool; returns the bytes given back
size_t plane_pool_flush(ssw_ctx* ctx);

// Device allocation: a failing hipMalloc is reported as SSW_ERR_OUT_OF_MEMORY whatever code the runtime
// chose for it, and the runtime's sticky error is cleared so that the next call starts clean.  Before that the
// working context's plane pool is flushed and the allocation retried once.
int dev_malloc(void** p, size_t bytes);
#define SSW_ALLOC(pp, bytes) SSW_TRY(::ssw::host::dev_malloc((void**)(pp), (bytes)))
int grow(ssw_ctx::Buf& b, size_t bytes);
void release(ssw_ctx::Buf& b);
int grow_select(hipStream_t st, SelectWorkspace& s, size_t frames, size_t k);
void release_select(SelectWorkspace& s);

// transfer.hip: host buffers <-> device on `st`.  upload() returns once the caller's buffer may be reused
// (the data reaches the device in stream order), download() once the caller's buffer is complete.
int upload(ssw_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes, hipStream_t st);
int download(ssw_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes, hipStream_t st);
// The same for callers that keep every host buffer until they synchronise `st` themselves (the streaming entry
// points): a pinned buffer is only ENQUEUED (no wait); any other buffer takes the staged path above (upload: returns
// when staged; download: returns when complete).  *async_out: whether the transfer is still in flight on return.
int upload_nowait(ssw_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes, hipStream_t st, bool* async_out);
int download_nowait(ssw_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes, hipStream_t st, bool* async_out);
void transfer_destroy(ssw_ctx* ctx);
int transfer_set_threads(ssw_ctx* ctx, int threads);
int transfer_stats(ssw_ctx* ctx, double* out, bool reset);

// Stage timer: an event pair around a region on `st`, plus the work the region does (executed flop of the
// GEMM stages, algorithmic bytes of the HBM-bound ones) -- both only while timing is enabled.
struct StageTimer {
    ssw_ctx* ctx;
    int stage;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    bool armed = false;
    int alias = -1;              // a second stage that this region IS (a pass that runs as its one "main" launch): same time, no events of its own
    StageTimer(ssw_ctx* c, int s, hipStream_t stream, double work = 0.0, int alias_stage = -1);
    ~StageTimer();
    // algorithmic HBM bytes of a GEMM stage (operands in, results and the inverse's E / T2 / A1 exchange out and in):
    // the HBM-bound stages' bytes ARE their work and are counted by the constructor
    void traffic(double bytes);
    StageTimer(const StageTimer&) = delete;
    StageTimer& operator=(const StageTimer&) = delete;
};
int flush_timers(ssw_ctx* ctx);
// Anything that is enqueued outside a StageTimer -- an event wait or record, an unbracketed launch or copy -- ends the sharing
// of the last timer's end event (ssw_pipeline.hip: timer_event): the next stage records its own start behind it and is not
// billed for it (a lane's hop to the other stream is a hipStreamWaitEvent: shared, it counted the dependency stall as stage time).
inline void untimed_work(ssw_ctx* ctx) { if (ctx) ctx->tail_fresh = false; }

// kind 0 = dense N x N, 1 / 2 = even / odd half basis, 3 / 4 = the same, k-blocked (operand-ready GEMMs)
int get_basis(ssw_ctx* ctx, size_t n, bool inverse, bool f64, int kind, const void** out);

bool valid_method(int m);
bool valid_ordering(int o);
bool valid_precision(int p);
int check_config(const ssw_config* cfg);
size_t effective_chunk(const ssw_ctx* ctx, size_t w, size_t h, size_t n_frames);

// ---- chains of stages -------------------------------------------------------------------------------
// A chain is the ordered list of device stages one chunk goes through.  Every stage is either HBM-bound
// (pre-passes, selection, colour conversion, O(k) kernels) or a group of basis-GEMM launches; the batch
// pipelines put the two kinds on different streams, everything else runs a chain on one stream.
struct Stage {
    bool hbm;
    std::function<int(hipStream_t)> run;
    int tag = 0;        // two-lane scheduling hints: 1 = the GEMM launches of a ROW pass, 2 = the RGB pre-pass that opens a forward transform
};
typedef std::vector<Stage> Chain;

// dct2d::dct2_2d on n contiguous planes, `data` in place, `tmp` same-size scratch; `ws` supplies the operand
// planes.  `rgb` (optional; rows-first forward transforms the fused pre-pass accepts, see can_fuse_rgb):
// the frames `data` would have been converted from -- the first pass reads them directly and `data` is
// only written by the last pass; iq_i / iq_q receive the I and Q planes (both or neither).
struct Xform {
    int type, precision;
    size_t n, w, h;
    float* data;
    float* tmp;
    const void* rgb = nullptr;
    int rgb_u8 = 0;                 // SSW_PIX_* of `rgb`
    float* iq_i = nullptr;
    float* iq_q = nullptr;
    // inverse transforms (Writer::result): where the last pass may deliver RGB pixels instead of the Y plane
    // (iq_i / iq_q are then inputs); build_transform reports through `fused_rgb` whether it did
    void* rgb_out = nullptr;
    bool rgb_out_u8 = false;
    // intermediate plane of a deep forward transform in class-major column order (dct_pair_common.hpp): decided from
    // the frame's shape -- `full_h` when this Xform is a band of rows of a taller frame; `natural_order` forces the
    // natural order (the compact plane of the pruned transform has its own)
    size_t full_h = 0;
    bool natural_order = false;
};
int build_transform(ssw_ctx* ctx, ssw_ctx::Lane& ws, const Xform& x, Chain& ch, bool* fused_rgb = nullptr);
bool can_fuse_rgb(const ssw_ctx* ctx, bool f64, size_t w, size_t h, const float* y, const float* tmp, const void* rgb, int u8);
// rgb -> Y (+ I, Q) -> forward transform of Y into `y` (Writer::new / Reader::new_impl), fused where possible
int build_forward_from_rgb(ssw_ctx* ctx, ssw_ctx::Lane& ws, int precision, const void* rgb, int u8, size_t n, size_t w,
                           size_t h, float* y, float* i, float* q, float* tmp, Chain& ch);
// the same transform as bands + 1 chains: the row pass of a band of image rows (h / bands of them), and the column pass of the whole frame
bool can_split_forward_rows(const ssw_ctx* ctx, bool f64, size_t w, size_t h, size_t bands, const float* y, const float* tmp, const void* rgb, int u8);
int build_forward_rows_band(ssw_ctx* ctx, ssw_ctx::Lane& ws, int precision, const void* rgb, int u8, size_t w, size_t rows,
                            size_t frame_h, float* tmp, float* i, float* q, Chain& ch);
int build_forward_cols_after_rows(ssw_ctx* ctx, ssw_ctx::Lane& ws, int precision, size_t w, size_t h, float* tmp, float* y, Chain& ch);
int run_serial(Chain& ch, hipStream_t st);
// transform now, on the context's stream, with lane 0's workspace (handles, ssw_dct2d)
int dct2d_planes(ssw_ctx* ctx, int type, int precision, size_t n, size_t w, size_t h, float* data, float* tmp);

// first k entries of the reference's ordering for n planes (select.hip; full sort beyond its limit)
int topk(ssw_ctx* ctx, hipStream_t st, SelectWorkspace& sel, const float* coef, size_t n, size_t w, size_t h, int ordering,
         size_t k, uint32_t* idx);

int batch_embed_impl(ssw_ctx* ctx, const ssw_config* cfg, const void* dev_rgb, int u8_in, size_t n_frames, size_t w,
                     size_t h, const float* dev_marks, size_t k, void* dev_rgb_out, bool u8_out, float* dev_coef_out,
                     uint32_t* dev_indices_out);
int batch_extract_impl(ssw_ctx* ctx, const ssw_config* cfg, const void* dev_base_rgb, const void* dev_derived_rgb, int u8,
                       size_t n_frames, size_t w, size_t h, size_t k, float* dev_extracted, const float* dev_marks,
                       float* dev_sims);

// Reader::extract with a derived frame that is still RGB on the device (single-image handles): pruned transform +
// extraction enqueued on the context's stream; see ssw_pipeline.hip
int extract_single_pruned(ssw_ctx* ctx, int precision, const void* derived_rgb, int u8, size_t w, size_t h, const float* base_y,
                          const uint32_t* idx, size_t k, int method, float alpha, float* dev_out, uint32_t** dev_info,
                          bool* applicable);

}  // namespace host
}  // namespace ssw
