// C ABI of libssw_hip.so: context, workspace, stage timers and the handle types that mirror the
// reference's Writer / Reader / Tester.  The transform chains and the batch pipelines live in
// ssw_pipeline.hip.  See include/ssw.h for the reference file:line each entry point replaces.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

#include "ssw_host.hpp"

namespace ssw {
static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }
}  // namespace ssw

using namespace ssw;
using namespace ssw::host;

// ---- handles --------------------------------------------------------------------------------
struct ssw_writer {
    ssw_ctx* ctx;
    size_t w, h;
    ssw_config cfg;
    float *y = nullptr, *i = nullptr, *q = nullptr;   // device planes; y holds the coefficients
    // Writer::new fixes the ordering from the ORIGINAL coefficients (:314); it is materialised lazily, for
    // the longest mark seen so far.  embed() keeps a snapshot of the original plane so that a later,
    // longer embed() still ranks what Writer::new ranked (mark() consumes the writer and needs none).
    float* y0 = nullptr;
    uint32_t* idx = nullptr;
    size_t idx_k = 0;
    bool consumed = false;
    // host staging of embed(): lives as long as the handle, so the copies need no host synchronisation
    // (one block: the marks, 16-byte padded, then the offset and length tables -- one copy instead of three)
    std::vector<uint32_t> blob;
    bool staging_in_flight = false;
};
struct ssw_reader {
    ssw_ctx* ctx;
    size_t w, h;
    bool is_base;
    ssw_config cfg;
    float* y = nullptr;                               // coefficients (a derived reader: once something asked for them)
    uint32_t* idx = nullptr;                          // cached first idx_k indices
    size_t idx_k = 0;
    // A derived reader (Reader::derived, :469-471) keeps its frame on the device and transforms it when it is
    // used: in extract(), where the base reader's index list says which coefficients are read (:556-561), only the
    // frequency columns those need (the pruned transform of the batch path, bit-identical values); fully when
    // coefficients() is called or the pruned path does not apply.
    void* rgb = nullptr;
    int rgb_u8 = 0;                 // SSW_PIX_* of `rgb`
    hipEvent_t rgb_uploaded = nullptr;
};

namespace {
// handles and single-call entry points: lane 0's workspace, the context's stream
int topk0(ssw_ctx* ctx, const float* coef, size_t n, size_t w, size_t h, int ordering, size_t k, uint32_t* idx) {
    return topk(ctx, ctx->stream, ctx->lane[0].sel, coef, n, w, h, ordering, k, idx);
}

// ---- device planes of the handles: a size-keyed pool in the context ---------------------------
size_t plane_pool_cap() {
    static const size_t cap = [] {
        const char* e = std::getenv("SSW_PLANE_POOL_MB");
        return (e ? (size_t)std::atoll(e) : (size_t)4096) << 20;
    }();
    return cap;
}
int pool_get(ssw_ctx* ctx, size_t bytes, void** p) {
    bytes = (std::max<size_t>(bytes, 16) + 255) / 256 * 256;
    auto it = ctx->plane_pool.find(bytes);
    if (it != ctx->plane_pool.end()) {
        *p = it->second;
        ctx->plane_pool.erase(it);
        ctx->plane_pool_bytes -= bytes;
        return SSW_OK;
    }
    return dev_malloc(p, bytes);                                      // out of memory: flushes the pool and retries once
}
// Every use of a handle's plane was enqueued on the context's stream, and so is every later use by the next
// owner: no synchronisation.  (ssw_ctx_set_stream synchronises the stream it leaves.)
void pool_put(ssw_ctx* ctx, void* p, size_t bytes) {
    if (!p) return;
    bytes = (std::max<size_t>(bytes, 16) + 255) / 256 * 256;
    if (ctx->plane_pool_bytes + bytes > plane_pool_cap()) { (void)hipFree(p); return; }
    ctx->plane_pool.emplace(bytes, p);
    ctx->plane_pool_bytes += bytes;
}

// Derived readers' RGB staging (ssw_ctx::rgb_spares).  take: a spare of that size with its event -- the caller makes the copy
// stream wait on it; none: a pooled / fresh buffer and a new event, and the caller orders the copy behind the whole stream.
constexpr size_t RGB_SPARES_MAX = 4;
int rgb_spare_take(ssw_ctx* ctx, size_t bytes, void** p, hipEvent_t* ev, bool* fenced) {
    for (size_t i = 0; i < ctx->rgb_spares.size(); ++i)
        if (ctx->rgb_spares[i].bytes == bytes) {
            *p = ctx->rgb_spares[i].p; *ev = ctx->rgb_spares[i].released; *fenced = true;
            ctx->rgb_spares.erase(ctx->rgb_spares.begin() + (long)i);
            return SSW_OK;
        }
    *fenced = false;
    SSW_TRY(pool_get(ctx, bytes, p));
    if (hipEventCreateWithFlags(ev, hipEventDisableTiming) != hipSuccess) { pool_put(ctx, *p, bytes); *p = nullptr; return SSW_ERR_HIP; }
    return SSW_OK;
}
// the last kernel that reads the buffer is enqueued on the context's stream: mark the point and keep the pair
void rgb_spare_give(ssw_ctx* ctx, void* p, size_t bytes, hipEvent_t ev) {
    if (!p) { if (ev) (void)hipEventDestroy(ev); return; }
    if (ev && ctx->rgb_spares.size() < RGB_SPARES_MAX && hipEventRecord(ev, ctx->stream) == hipSuccess) {
        untimed_work(ctx);
        ctx->rgb_spares.push_back({p, bytes, ev});
        return;
    }
    if (ev) (void)hipEventDestroy(ev);
    pool_put(ctx, p, bytes);
}
// ---- frames in and out of the device ------------------------------------------------------------
int frame_stage_events(ssw_ctx::FrameStage& fs) {
    for (hipEvent_t& e : fs.uploaded)
        if (!e) SSW_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (!fs.consumed) SSW_HIP_CHECK(hipEventCreateWithFlags(&fs.consumed, hipEventDisableTiming));
    return SSW_OK;
}
// Host frame -> the next device staging buffer, on the copy stream (beside whatever the context's stream is
// still computing for an earlier handle); the context's stream waits for it.
int stage_frame_in(ssw_ctx* ctx, const void* host, size_t bytes, ssw_ctx::FrameStage** out) {
    ssw_ctx::FrameStage& fs = ctx->frame_stage[ctx->frame_stage_next++ & 1];
    SSW_TRY(frame_stage_events(fs));
    SSW_TRY(grow(fs.buf, bytes));
    if (fs.in_use) SSW_HIP_CHECK(hipStreamWaitEvent(ctx->copy_stream, fs.consumed, 0));
    SSW_TRY(upload(ctx, fs.buf.p, host, bytes, ctx->copy_stream));
    SSW_HIP_CHECK(hipEventRecord(fs.uploaded[0], ctx->copy_stream));
    SSW_HIP_CHECK(hipStreamWaitEvent(ctx->stream, fs.uploaded[0], 0));
    untimed_work(ctx);
    *out = &fs;
    return SSW_OK;
}
int stage_consumed(ssw_ctx* ctx, ssw_ctx::FrameStage& fs) {
    SSW_HIP_CHECK(hipEventRecord(fs.consumed, ctx->stream));
    untimed_work(ctx);
    fs.in_use = true;
    return SSW_OK;
}

// Writer::new / Reader::new_impl up to the coefficients (:308-313, :476-480): host frame (f32 or 8-bit) -> Y
// (+ I, Q) -> forward transform, through the same fused chain as the batch entry points (n = 1).  Enqueues only.
int forward_from_host(ssw_ctx* ctx, const void* host_rgb, int u8, size_t w, size_t h, int precision, float* y, float* i,
                      float* q) {
    const size_t plane = w * h;
    const size_t bytes = plane * 3 * pix_bytes(u8);
    SSW_TRY(grow(ctx->lane[0].plane[3], plane * sizeof(float)));
    float* tmp = (float*)ctx->lane[0].plane[3].p;
    {
        // Bands of image rows (two to four, `upload_bands`): the row pass of a band runs while the next ones are still
        // crossing PCIe (image rows are independent lines of a row pass: same values), then the column pass of the whole
        // frame.  Pinned frames: every copy is enqueued before the first kernel, the kernels wait on the bands' events,
        // and the host waits once, at the end, for the last copy (the caller's buffer is the DMA source) -- a wait per
        // band left the link idle for ~30 us between the bands.
        ssw_ctx::FrameStage& fs = ctx->frame_stage[ctx->frame_stage_next & 1];
        SSW_TRY(grow(fs.buf, bytes));
        const bool no_split = tuning(TUNE_BAND_SPLIT) == 0;                     // A/B switch (tuning.hip)
        long long nb = tuning(TUNE_UPLOAD_BANDS);
        nb = nb < 2 ? 2 : nb > ssw_ctx::FrameStage::MAX_BANDS ? ssw_ctx::FrameStage::MAX_BANDS : nb;
        while (nb > 2 && ((bytes / nb) % 16 != 0 || !can_split_forward_rows(ctx, precision == SSW_PRECISION_F64, w, h, (size_t)nb, y, tmp, fs.buf.p, u8))) --nb;
        if (!no_split && bytes >= ((size_t)8 << 20) &&
            can_split_forward_rows(ctx, precision == SSW_PRECISION_F64, w, h, (size_t)nb, y, tmp, fs.buf.p, u8)) {
            ++ctx->frame_stage_next;
            SSW_TRY(frame_stage_events(fs));
            if (fs.in_use) SSW_HIP_CHECK(hipStreamWaitEvent(ctx->copy_stream, fs.consumed, 0));
            const size_t hb = bytes / nb, hp = plane / nb, rows = h / nb;
            bool in_flight = false;
            int rc = SSW_OK;
            for (int band = 0; band < (int)nb && rc == SSW_OK; ++band) {
                bool async = false;
                rc = upload_nowait(ctx, (char*)fs.buf.p + band * hb, (const char*)host_rgb + band * hb, hb, ctx->copy_stream, &async);
                in_flight = in_flight || async;
                if (rc == SSW_OK && hipEventRecord(fs.uploaded[band], ctx->copy_stream) != hipSuccess) rc = SSW_ERR_HIP;
                if (rc == SSW_OK && hipStreamWaitEvent(ctx->stream, fs.uploaded[band], 0) != hipSuccess) rc = SSW_ERR_HIP;
                untimed_work(ctx);                  // the band's first stage starts its timer behind the wait for its upload
                if (rc != SSW_OK) break;
                Chain ch;
                rc = build_forward_rows_band(ctx, ctx->lane[0], precision, (char*)fs.buf.p + band * hb, u8, w, rows, h, tmp + band * hp,
                                             i ? i + band * hp : nullptr, q ? q + band * hp : nullptr, ch);
                if (rc == SSW_OK) rc = run_serial(ch, ctx->stream);
            }
            if (rc == SSW_OK) rc = stage_consumed(ctx, fs);
            if (rc == SSW_OK) {
                Chain ch;
                rc = build_forward_cols_after_rows(ctx, ctx->lane[0], precision, w, h, tmp, y, ch);
                if (rc == SSW_OK) rc = run_serial(ch, ctx->stream);
            }
            // the caller's buffer is the source of the copies still in flight: wait for them on every path out
            if (in_flight && hipStreamSynchronize(ctx->copy_stream) != hipSuccess && rc == SSW_OK) rc = SSW_ERR_HIP;
            return rc;
        }
    }
    ssw_ctx::FrameStage* fs = nullptr;
    SSW_TRY(stage_frame_in(ctx, host_rgb, bytes, &fs));
    Chain ch;
    SSW_TRY(build_forward_from_rgb(ctx, ctx->lane[0], precision, fs->buf.p, u8, 1, w, h, y, i, q, tmp, ch));
    SSW_TRY(run_serial(ch, ctx->stream));
    return stage_consumed(ctx, *fs);
}
}  // namespace

// ---- library / context ----------------------------------------------------------------------
extern "C" {

const char* ssw_version(void) { return "ssw-hip 0.1.0 (gfx950)"; }
int ssw_build_all_strategies(void) { return ssw::build_all_strategies() ? 1 : 0; }

int ssw_ctx_transform_plan(ssw_ctx* ctx, size_t n_frames, size_t w, size_t h, int dct_type, uint32_t* flags) {
    if (!ctx || !flags || w == 0 || h == 0) return SSW_ERR_BAD_ARG;
    const bool inverse = dct_type == SSW_DCT3;
    const bool pair = ctx->fold && ctx->fold_level >= 3 && w % 8 == 0 && h % 8 == 0 && w >= 16 && h >= 16 && ctx->split;
    uint32_t f = 0;
    if (pair) {
        f |= SSW_PLAN_PAIR_F64;
        const bool rows_first = w >= h;
        const bool deep_r = inverse ? ssw::dct_pair_can_deep_inv_rows(w) : ssw::dct_pair_can_deep_rows(w);
        const bool deep_c = ssw::dct_pair_can_deep_cols(h) && w % 4 == 0;
        if (deep_r) f |= SSW_PLAN_ROWS_DEEP;
        if (deep_c) f |= SSW_PLAN_COLS_DEEP;
        const bool l2r = deep_r && (inverse ? ssw::dct_pair_efold_inv(w) : ssw::dct_pair_efold(w));
        const bool cm = rows_first && deep_r && w % 4 == 0 && (deep_c || (ssw::dct_pair_can_semi_deep_cols(h) && ssw::dct_pair_prep_staged_cols_ok(w, true)));
        const bool l2c = deep_c && ssw::dct_pair_efold_cols(h, w, cm);
        if (l2r) f |= SSW_PLAN_ROWS_LEVEL2;
        if (l2c) f |= SSW_PLAN_COLS_LEVEL2;
        if (cm) f |= SSW_PLAN_CLASS_MAJOR;
        if (l2r && l2c && cm && (inverse ? ssw::dct_pair_can_fuse_inv_cols(n_frames, w, h) : ssw::dct_pair_can_fuse_cols(n_frames, w, h)))
            f |= SSW_PLAN_FUSED_COLS;
    }
    *flags = f;
    return SSW_OK;
}

const char* ssw_status_string(int s) {
    switch (s) {
    case SSW_OK: return "ok";
    case SSW_ERR_BAD_ARG: return "bad argument";
    case SSW_ERR_BAD_DIMS: return "bad dimensions";
    case SSW_ERR_LENGTH_MISMATCH: return "Derived coefficient length not equal to base coefficient length.";
    case SSW_ERR_K_TOO_LARGE: return "Desired extraction length exceeds available coefficients.";
    case SSW_ERR_NOT_BASE: return "reader was not created as a base reader";
    case SSW_ERR_UNSUPPORTED: return "unsupported (custom closures cannot run on the device)";
    case SSW_ERR_CONSUMED: return "writer already consumed";
    case SSW_ERR_HIP: return "HIP runtime error";
    case SSW_ERR_NO_DEVICE: return "no HIP device (this library has no CPU path)";
    case SSW_ERR_OUT_OF_MEMORY: return "out of device memory";
    default: return "unknown status";
    }
}

const char* ssw_last_error(void) { return g_last_error.c_str(); }

void ssw_config_default(ssw_config* cfg) {
    if (!cfg) return;
    cfg->ordering = SSW_ORDER_ENERGY;        // src/algorithm.rs:104-111
    cfg->method = SSW_OPTION2;
    cfg->alpha = 0.1f;
    cfg->precision = SSW_PRECISION_F64;      // canonical: bit-parity with the CPU path
}

int ssw_ctx_create(int device_id, ssw_ctx** out) {
    if (!out) return SSW_ERR_BAD_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        set_last_error("hipGetDeviceCount: no device");
        return SSW_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= count) return SSW_ERR_BAD_ARG;
    DeviceGuard g(device_id);                // the caller's current device is restored on return
    ssw_ctx* ctx = new (std::nothrow) ssw_ctx();
    if (!ctx) return SSW_ERR_OUT_OF_MEMORY;
    ctx->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; set_last_error(hipGetErrorString(e)); return SSW_ERR_HIP; }
    ctx->stream = ctx->own_stream;
    // second stream of the batch pipelines (ssw_pipeline.hip): the HBM-bound stages of one chunk run here
    // while the basis GEMMs of the other chunk in flight run on the context's stream.
    // (Measured: restricting this stream to 16 .. 128 CUs with a CU mask only slows the step down -- the
    // chunk's GEMMs wait for its pre-pass -- so it is a plain stream.  Measured again at level 2, where the pre-passes
    // are a third of a step, also with the complement mask on the GEMM stream: 64 CUs 8.8 / 9.8, 96 + complement 10.0,
    // 128 CUs 12.1 Gpix/s against 13.9.)
    // Stream priorities (either way round) change nothing either.
    e = hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { (void)hipStreamDestroy(ctx->own_stream); delete ctx; set_last_error(hipGetErrorString(e)); return SSW_ERR_HIP; }
    // frames of the single-image handles cross PCIe here, beside the kernels of the previous handle
    e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(ctx->aux_stream); (void)hipStreamDestroy(ctx->own_stream);
        delete ctx; set_last_error(hipGetErrorString(e)); return SSW_ERR_HIP;
    }
    // the selection's fallback counter lives as long as the context (allocated here, not lazily inside a batch call:
    // a first call under stream capture must not allocate)
    if (hipMalloc((void**)&ctx->select_fallbacks, sizeof(uint32_t)) != hipSuccess ||
        hipMemset(ctx->select_fallbacks, 0, sizeof(uint32_t)) != hipSuccess) {
        (void)hipGetLastError();
        if (ctx->select_fallbacks) (void)hipFree(ctx->select_fallbacks);
        (void)hipStreamDestroy(ctx->copy_stream); (void)hipStreamDestroy(ctx->aux_stream); (void)hipStreamDestroy(ctx->own_stream);
        delete ctx; set_last_error("hipMalloc: selection counter"); return SSW_ERR_OUT_OF_MEMORY;
    }
    const char* ov = std::getenv("SSW_OVERLAP");
    if (ov) ctx->overlap = std::atoi(ov) != 0;
    const char* pr = std::getenv("SSW_PRUNE");
    if (pr) ctx->prune = std::atoi(pr) != 0;
    const char* sp = std::getenv("SSW_ODD_SPLIT");
    if (sp) ctx->split = std::atoi(sp) != 0;
    *out = ctx;
    return SSW_OK;
}

int ssw_ctx_destroy(ssw_ctx* ctx) {
    if (!ctx) return SSW_OK;
    CtxGuard g(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->aux_stream) (void)hipStreamSynchronize(ctx->aux_stream);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    if (ctx->down_stream) (void)hipStreamSynchronize(ctx->down_stream);
    transfer_destroy(ctx);
    for (int s = 0; s < ssw_ctx::HostStream::NB; ++s) {
        release(ctx->hs.in[s]); release(ctx->hs.in2[s]); release(ctx->hs.out[s]);
        for (hipEvent_t e : {ctx->hs.up_done[s], ctx->hs.k_done[s], ctx->hs.down_done[s]}) if (e) (void)hipEventDestroy(e);
    }
    release(ctx->hs.marks); release(ctx->hs.ext); release(ctx->hs.sims);
    for (void* p : ctx->retired) (void)hipFree(p);
    ctx->retired.clear();
    (void)plane_pool_flush(ctx);
    for (auto& fs : ctx->frame_stage) {
        release(fs.buf);
        for (hipEvent_t e : fs.uploaded)
            if (e) (void)hipEventDestroy(e);
        if (fs.consumed) (void)hipEventDestroy(fs.consumed);
    }
    for (auto& kv : ctx->basis) (void)hipFree(kv.second);
    for (auto& ln : ctx->lane) {
        for (auto& b : ln.plane) release(b);
        for (auto& b : ln.operand) release(b);
        for (auto& b : ln.compact) release(b);
        release(ln.idx);
        release(ln.gathered);
        release(ln.prune_u32);
        release_select(ln.sel);
    }
    if (ctx->select_fallbacks) (void)hipFree(ctx->select_fallbacks);
    release(ctx->overflow);
    release(ctx->small);
    release(ctx->sort_scratch);
    release(ctx->resize_tmp);
    for (auto& kv : ctx->taps) { (void)hipFree(kv.second.left); (void)hipFree(kv.second.count); (void)hipFree(kv.second.weights); }
    for (auto& e : ctx->sync_events) (void)hipEventDestroy(e);
    for (auto& p : ctx->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto& e : ctx->free_events) (void)hipEventDestroy(e);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->down_stream) (void)hipStreamDestroy(ctx->down_stream);
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return SSW_OK;
}

int ssw_ctx_synchronize(ssw_ctx* ctx) {
    if (!ctx) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    SSW_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SSW_OK;
}

void* ssw_ctx_stream(ssw_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int ssw_ctx_set_stream(ssw_ctx* ctx, void* hip_stream) {
    if (!ctx) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    SSW_TRY(flush_timers(ctx));                                      // pending event pairs belong to the old stream
    SSW_HIP_CHECK(hipStreamSynchronize(ctx->stream));                // workspace in flight on the old stream
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return SSW_OK;
}

int ssw_ctx_wait_event(ssw_ctx* ctx, void* hip_event) {
    if (!ctx || !hip_event) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    SSW_HIP_CHECK(hipStreamWaitEvent(ctx->stream, (hipEvent_t)hip_event, 0));
    return SSW_OK;
}

int ssw_ctx_record_event(ssw_ctx* ctx, void* hip_event) {
    if (!ctx || !hip_event) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    SSW_HIP_CHECK(hipEventRecord((hipEvent_t)hip_event, ctx->stream));
    return SSW_OK;
}

int ssw_ctx_set_chunk_frames(ssw_ctx* ctx, size_t frames) {
    if (!ctx) return SSW_ERR_BAD_ARG;
    ctx->chunk_frames = frames;                    // 0 = automatic
    return SSW_OK;
}

size_t ssw_ctx_pass_frames(ssw_ctx* ctx, size_t n_frames, size_t w, size_t h) {
    return ctx ? effective_chunk(ctx, w, h, n_frames) : 0;
}

int ssw_ctx_set_dct_folding(ssw_ctx* ctx, int enable) {
    if (!ctx) return SSW_ERR_BAD_ARG;
    if (enable < 0 || enable > 6) return SSW_ERR_BAD_ARG;
    ctx->fold = enable != 0;
    ctx->fold_level = enable;                 // strategy levels: include/ssw.h
    return SSW_OK;
}

int ssw_ctx_enable_timing(ssw_ctx* ctx, int enable) {
    if (!ctx) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    SSW_TRY(flush_timers(ctx));
    ctx->timing = enable != 0;
    return SSW_OK;
}

int ssw_ctx_reset_timing(ssw_ctx* ctx) {
    if (!ctx) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    SSW_TRY(flush_timers(ctx));
    for (int s = 0; s < SSW_STAGE_COUNT; ++s) { ctx->stage_ms[s] = 0; ctx->stage_launches[s] = 0; ctx->stage_work[s] = 0; ctx->stage_bytes[s] = 0; }
    ctx->pruned_chunks = ctx->redone_chunks = ctx->pruned_columns = 0;
    ctx->select_frames = 0;
    // the finish kernels of either lane add to the counter: both lane streams are idle before it is zeroed (a caller's
    // stream set through ssw_ctx_set_stream IS ctx->stream)
    SSW_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->aux_stream) SSW_HIP_CHECK(hipStreamSynchronize(ctx->aux_stream));
    SSW_HIP_CHECK(hipMemset(ctx->select_fallbacks, 0, sizeof(uint32_t)));
    return SSW_OK;
}

int ssw_ctx_get_timing(ssw_ctx* ctx, double* ms, uint64_t* launches) {
    if (!ctx) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    SSW_TRY(flush_timers(ctx));
    for (int s = 0; s < SSW_STAGE_COUNT; ++s) {
        if (ms) ms[s] = ctx->stage_ms[s];
        if (launches) launches[s] = ctx->stage_launches[s];
    }
    return SSW_OK;
}

int ssw_ctx_get_work(ssw_ctx* ctx, double* work) {
    if (!ctx || !work) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    SSW_TRY(flush_timers(ctx));
    for (int s = 0; s < SSW_STAGE_COUNT; ++s) work[s] = ctx->stage_work[s];
    return SSW_OK;
}

int ssw_ctx_get_traffic(ssw_ctx* ctx, double* bytes) {
    if (!ctx || !bytes) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    SSW_TRY(flush_timers(ctx));
    for (int s = 0; s < SSW_STAGE_COUNT; ++s) bytes[s] = ctx->stage_bytes[s];
    return SSW_OK;
}

int ssw_ctx_set_overlap(ssw_ctx* ctx, int enable) {
    if (!ctx) return SSW_ERR_BAD_ARG;
    ctx->overlap = enable != 0;
    return SSW_OK;
}

int ssw_ctx_set_prune(ssw_ctx* ctx, int enable) {
    if (!ctx) return SSW_ERR_BAD_ARG;
    ctx->prune = enable != 0;
    return SSW_OK;
}

int ssw_ctx_set_odd_split(ssw_ctx* ctx, int enable) {
    if (!ctx) return SSW_ERR_BAD_ARG;
    ctx->split = enable != 0;
    return SSW_OK;
}

int ssw_ctx_get_prune_stats(ssw_ctx* ctx, uint64_t* stats) {
    if (!ctx || !stats) return SSW_ERR_BAD_ARG;
    stats[0] = ctx->pruned_chunks;
    stats[1] = ctx->redone_chunks;
    stats[2] = ctx->pruned_columns;
    return SSW_OK;
}

int ssw_ctx_get_select_stats(ssw_ctx* ctx, uint64_t* stats) {
    if (!ctx || !stats) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    stats[0] = ctx->select_frames;
    stats[1] = 0;
    if (ctx->select_fallbacks) {
        uint32_t v = 0;
        SSW_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->aux_stream) SSW_HIP_CHECK(hipStreamSynchronize(ctx->aux_stream));
        SSW_HIP_CHECK(hipMemcpy(&v, ctx->select_fallbacks, sizeof(v), hipMemcpyDeviceToHost));
        stats[1] = v;
    }
    return SSW_OK;
}

int ssw_dev_mem_info(ssw_ctx* ctx, size_t* free_bytes, size_t* total_bytes) {
    if (!ctx || !free_bytes || !total_bytes) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    SSW_HIP_CHECK(hipMemGetInfo(free_bytes, total_bytes));
    return SSW_OK;
}

int ssw_dev_alloc(ssw_ctx* ctx, size_t bytes, void** dev_ptr) {
    if (!ctx || !dev_ptr) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    SSW_ALLOC(dev_ptr, bytes);
    return SSW_OK;
}
int ssw_dev_free(ssw_ctx* ctx, void* dev_ptr) {
    if (!ctx) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    SSW_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (dev_ptr) SSW_HIP_CHECK(hipFree(dev_ptr));
    return SSW_OK;
}
int ssw_copy_to_dev(ssw_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes) {
    if (!ctx || (bytes && (!dev_dst || !host_src))) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    SSW_TRY(upload(ctx, dev_dst, host_src, bytes, ctx->stream));
    SSW_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SSW_OK;
}
int ssw_copy_to_host(ssw_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes) {
    if (!ctx || (bytes && (!host_dst || !dev_src))) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    return download(ctx, host_dst, dev_src, bytes, ctx->stream);
}

int ssw_host_alloc(ssw_ctx* ctx, size_t bytes, void** host_ptr) {
    if (!ctx || !host_ptr) return SSW_ERR_BAD_ARG;
    *host_ptr = nullptr;
    CtxGuard g(ctx);
    const hipError_t e = hipHostMalloc(host_ptr, bytes ? bytes : 16, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *host_ptr = nullptr;
        set_last_error(std::string("hipHostMalloc(") + std::to_string(bytes) + " bytes): " + hipGetErrorString(e));
        return SSW_ERR_OUT_OF_MEMORY;
    }
    return SSW_OK;
}
int ssw_host_free(ssw_ctx* ctx, void* host_ptr) {
    if (!ctx) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    if (host_ptr) SSW_HIP_CHECK(hipHostFree(host_ptr));
    return SSW_OK;
}
int ssw_ctx_set_copy_threads(ssw_ctx* ctx, int threads) {
    if (!ctx || threads < 0) return SSW_ERR_BAD_ARG;
    return transfer_set_threads(ctx, threads);
}
int ssw_ctx_get_transfer_stats(ssw_ctx* ctx, double* stats, int reset) {
    if (!ctx) return SSW_ERR_BAD_ARG;
    return transfer_stats(ctx, stats, reset != 0);
}

// ---- transforms -----------------------------------------------------------------------------
int ssw_rgb_to_yiq(ssw_ctx* ctx, const float* dev_rgb, size_t n_frames, size_t w, size_t h,
                   float* dev_y, float* dev_i, float* dev_q) {
    if (!ctx || !dev_rgb || !dev_y || ((dev_i == nullptr) != (dev_q == nullptr))) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    StageTimer t(ctx, SSW_STAGE_RGB_TO_YIQ, ctx->stream);
    return launch_rgb_to_yiq(ctx->stream, dev_rgb, n_frames * w * h, dev_y, dev_i, dev_q);
}

int ssw_yiq_to_rgb(ssw_ctx* ctx, const float* dev_y, const float* dev_i, const float* dev_q,
                   size_t n_frames, size_t w, size_t h, float* dev_rgb) {
    if (!ctx || !dev_rgb || !dev_y || !dev_i || !dev_q) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    StageTimer t(ctx, SSW_STAGE_YIQ_TO_RGB, ctx->stream);
    return launch_yiq_to_rgb(ctx->stream, dev_y, dev_i, dev_q, n_frames * w * h, dev_rgb);
}

int ssw_dct2d(ssw_ctx* ctx, int dct_type, int precision, size_t n_frames, size_t w, size_t h,
              float* dev_planes) {
    if (!ctx || !dev_planes) return SSW_ERR_BAD_ARG;
    if (dct_type < SSW_DCT2 || dct_type > SSW_DCT3 || !valid_precision(precision)) return SSW_ERR_BAD_ARG;
    if (w == 0 || h == 0) return SSW_ERR_BAD_DIMS;
    CtxGuard g(ctx);
    const size_t chunk = effective_chunk(ctx, w, h, n_frames);
    SSW_TRY(grow(ctx->lane[0].plane[3], chunk * w * h * sizeof(float)));
    for (size_t f0 = 0; f0 < n_frames; f0 += chunk) {
        const size_t n = std::min(chunk, n_frames - f0);
        SSW_TRY(dct2d_planes(ctx, dct_type, precision, n, w, h, dev_planes + f0 * w * h, (float*)ctx->lane[0].plane[3].p));
    }
    return SSW_OK;
}

int ssw_topk_indices(ssw_ctx* ctx, const float* dev_coef, size_t n_frames, size_t w, size_t h,
                     int ordering, size_t k, uint32_t* dev_indices) {
    if (!ctx || !dev_coef || !dev_indices) return SSW_ERR_BAD_ARG;
    if (ordering == SSW_ORDER_CUSTOM) return SSW_ERR_UNSUPPORTED;
    if (!valid_ordering(ordering)) return SSW_ERR_BAD_ARG;
    if (w == 0 || h == 0) return SSW_ERR_BAD_DIMS;
    if (k > w * h - 1) return SSW_ERR_K_TOO_LARGE;
    CtxGuard g(ctx);
    return topk0(ctx, dev_coef, n_frames, w, h, ordering, k, dev_indices);
}

int ssw_embed_coefficients(ssw_ctx* ctx, float* dev_coef, size_t n_frames, size_t plane_len,
                           const uint32_t* dev_indices, size_t k, int method, float alpha,
                           const float* dev_marks, size_t n_marks) {
    if (!ctx || !dev_coef || !dev_indices || !dev_marks) return SSW_ERR_BAD_ARG;
    if (method == SSW_METHOD_CUSTOM) return SSW_ERR_UNSUPPORTED;
    if (!valid_method(method)) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    StageTimer t(ctx, SSW_STAGE_EMBED, ctx->stream);
    return launch_embed(ctx->stream, dev_coef, n_frames, plane_len, dev_indices, k, dev_marks, nullptr,
                        nullptr, n_marks, k, k, method, alpha);
}

int ssw_extract_coefficients(ssw_ctx* ctx, const float* dev_base, const float* dev_derived,
                             size_t n_frames, size_t plane_len, const uint32_t* dev_indices,
                             size_t k, int method, float alpha, float* dev_out) {
    if (!ctx || !dev_base || !dev_derived || !dev_indices || !dev_out) return SSW_ERR_BAD_ARG;
    if (method == SSW_METHOD_CUSTOM) return SSW_ERR_UNSUPPORTED;
    if (!valid_method(method)) return SSW_ERR_BAD_ARG;
    if (k >= plane_len) return SSW_ERR_K_TOO_LARGE;                   // src/algorithm.rs:553-555
    CtxGuard g(ctx);
    StageTimer t(ctx, SSW_STAGE_EXTRACT, ctx->stream);
    return launch_extract(ctx->stream, dev_base, dev_derived, n_frames, plane_len, dev_indices, k, method,
                          alpha, dev_out);
}

int ssw_similarity_batch(ssw_ctx* ctx, const float* dev_extracted, const float* dev_marks,
                         size_t n_pairs, size_t k, float* dev_sims) {
    if (!ctx || !dev_extracted || !dev_marks || !dev_sims) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    StageTimer t(ctx, SSW_STAGE_SIMILARITY, ctx->stream);
    return launch_similarity(ctx->stream, dev_extracted, dev_marks, n_pairs, k, dev_sims);
}

int ssw_similarity_matrix(ssw_ctx* ctx, const float* dev_extracted, size_t n_extracted, const float* dev_marks,
                          size_t n_marks, size_t k, float* dev_sims) {
    if (!ctx || !dev_extracted || !dev_marks || !dev_sims) return SSW_ERR_BAD_ARG;
    if (n_extracted == 0 || n_marks == 0) return SSW_OK;
    CtxGuard g(ctx);
    SSW_TRY(grow(ctx->small, n_extracted * sizeof(float)));
    StageTimer t(ctx, SSW_STAGE_SIMILARITY, ctx->stream);
    SSW_TRY(launch_sim_den(ctx->stream, dev_extracted, n_extracted, k, (float*)ctx->small.p));
    SSW_TRY(launch_gemm_nt_f32(ctx->stream, dev_extracted, n_extracted, dev_marks, n_marks, k, dev_sims));
    return launch_sim_scale(ctx->stream, dev_sims, (const float*)ctx->small.p, n_extracted, n_marks);
}

// ---- whole path, batched (ssw_pipeline.hip) ------------------------------------------------
namespace {

int get_taps(ssw_ctx* ctx, size_t in_len, size_t out_len, DeviceTaps* out) {
    auto key = std::make_pair(in_len, out_len);
    auto it = ctx->taps.find(key);
    if (it != ctx->taps.end()) { *out = it->second; return SSW_OK; }
    ResizeTaps host;
    build_resize_taps(in_len, out_len, host);
    DeviceTaps d;
    d.max_taps = host.max_taps;
    for (int e = 0; e < 8; ++e) {                       // reach of 2^e consecutive outputs, maximum over all aligned groups
        const size_t g = (size_t)1 << e;
        uint32_t m = 0;
        for (size_t o = 0; o < out_len; o += g) {
            const size_t last = std::min(o + g, out_len) - 1;
            m = std::max(m, host.left[last] + host.count[last] - host.left[o]);
        }
        d.span[e] = m;
    }
    d.quad_uniform = out_len % 4 == 0;
    for (size_t o = 0; o < out_len && d.quad_uniform; ++o)
        d.quad_uniform = host.count[o] <= 5 && host.left[o] == host.left[o & ~(size_t)3] && host.count[o] == host.count[o & ~(size_t)3];
    // uploaded on the context's stream (the one the resize kernels run on); the host vectors die with this call
    auto put = [&](void** dev, const void* host, size_t bytes) -> int {
        SSW_ALLOC(dev, bytes);
        SSW_HIP_CHECK(hipMemcpyAsync(*dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
        return SSW_OK;
    };
    int rc = put((void**)&d.left, host.left.data(), out_len * sizeof(uint32_t));
    if (rc == SSW_OK) rc = put((void**)&d.count, host.count.data(), out_len * sizeof(uint32_t));
    if (rc == SSW_OK) rc = put((void**)&d.weights, host.weights.data(), host.weights.size() * sizeof(float));
    if (rc == SSW_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = SSW_ERR_HIP;
    if (rc != SSW_OK) {                                   // nothing of a half-built table stays behind
        (void)hipFree(d.left); (void)hipFree(d.count); (void)hipFree(d.weights);
        return rc;
    }
    ctx->taps[key] = d;
    *out = d;
    return SSW_OK;
}

}  // namespace

int ssw_batch_embed(ssw_ctx* ctx, const ssw_config* cfg, const float* dev_rgb, size_t n_frames,
                    size_t w, size_t h, const float* dev_marks, size_t k, float* dev_rgb_out,
                    float* dev_coef_out, uint32_t* dev_indices_out) {
    return batch_embed_impl(ctx, cfg, dev_rgb, false, n_frames, w, h, dev_marks, k, dev_rgb_out, false,
                            dev_coef_out, dev_indices_out);
}

int ssw_batch_extract(ssw_ctx* ctx, const ssw_config* cfg, const float* dev_base_rgb,
                      const float* dev_derived_rgb, size_t n_frames, size_t w, size_t h, size_t k,
                      float* dev_extracted, const float* dev_marks, float* dev_sims) {
    return batch_extract_impl(ctx, cfg, dev_base_rgb, dev_derived_rgb, false, n_frames, w, h, k, dev_extracted,
                              dev_marks, dev_sims);
}

int ssw_batch_embed_rgb8(ssw_ctx* ctx, const ssw_config* cfg, const uint8_t* dev_rgb, size_t n_frames,
                         size_t w, size_t h, const float* dev_marks, size_t k, uint8_t* dev_rgb_out) {
    return batch_embed_impl(ctx, cfg, dev_rgb, SSW_PIX_U8, n_frames, w, h, dev_marks, k, dev_rgb_out, true, nullptr, nullptr);
}

int ssw_batch_extract_rgb8(ssw_ctx* ctx, const ssw_config* cfg, const uint8_t* dev_base_rgb,
                           const uint8_t* dev_derived_rgb, size_t n_frames, size_t w, size_t h, size_t k,
                           float* dev_extracted, const float* dev_marks, float* dev_sims) {
    return batch_extract_impl(ctx, cfg, dev_base_rgb, dev_derived_rgb, SSW_PIX_U8, n_frames, w, h, k, dev_extracted,
                              dev_marks, dev_sims);
}

// ---- 16-bit boundary (`into_rgb32f()` of an Rgb16 image, v / 65535, fused into the first operand
// pre-pass like the 8-bit form; Writer::mark returns Rgb32F, src/algorithm.rs:355-379, so the marked frames leave as f32) ----
int ssw_batch_embed_rgb16(ssw_ctx* ctx, const ssw_config* cfg, const uint16_t* dev_rgb, size_t n_frames,
                          size_t w, size_t h, const float* dev_marks, size_t k, float* dev_rgb_out) {
    return batch_embed_impl(ctx, cfg, dev_rgb, SSW_PIX_U16, n_frames, w, h, dev_marks, k, dev_rgb_out, false, nullptr, nullptr);
}

int ssw_batch_extract_rgb16(ssw_ctx* ctx, const ssw_config* cfg, const uint16_t* dev_base_rgb,
                            const uint16_t* dev_derived_rgb, size_t n_frames, size_t w, size_t h, size_t k,
                            float* dev_extracted, const float* dev_marks, float* dev_sims) {
    return batch_extract_impl(ctx, cfg, dev_base_rgb, dev_derived_rgb, SSW_PIX_U16, n_frames, w, h, k, dev_extracted,
                              dev_marks, dev_sims);
}

int ssw_convert_rgb16_to_f32(ssw_ctx* ctx, const uint16_t* dev_in, size_t n_values, float* dev_out) {
    if (!ctx || (n_values && (!dev_in || !dev_out))) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    StageTimer t(ctx, SSW_STAGE_CONVERT, ctx->stream);
    return launch_u16_to_f32(ctx->stream, dev_in, n_values, dev_out);
}

int ssw_convert_f32_to_rgb16(ssw_ctx* ctx, const float* dev_in, size_t n_values, uint16_t* dev_out) {
    if (!ctx || (n_values && (!dev_in || !dev_out))) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    StageTimer t(ctx, SSW_STAGE_CONVERT, ctx->stream);
    return launch_f32_to_u16(ctx->stream, dev_in, n_values, dev_out);
}

// ---- 8-bit boundary and the resize attack ---------------------------------------------------------
int ssw_convert_rgb8_to_f32(ssw_ctx* ctx, const uint8_t* dev_in, size_t n_values, float* dev_out) {
    if (!ctx || (n_values && (!dev_in || !dev_out))) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    StageTimer t(ctx, SSW_STAGE_CONVERT, ctx->stream);
    return launch_u8_to_f32(ctx->stream, dev_in, n_values, dev_out);
}

int ssw_convert_f32_to_rgb8(ssw_ctx* ctx, const float* dev_in, size_t n_values, uint8_t* dev_out) {
    if (!ctx || (n_values && (!dev_in || !dev_out))) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    StageTimer t(ctx, SSW_STAGE_CONVERT, ctx->stream);
    return launch_f32_to_u8(ctx->stream, dev_in, n_values, dev_out);
}

int ssw_resize_rgb8(ssw_ctx* ctx, const uint8_t* dev_in, size_t n_frames, size_t w, size_t h, size_t nw,
                    size_t nh, uint8_t* dev_out) {
    if (!ctx || !dev_in || !dev_out) return SSW_ERR_BAD_ARG;
    if (w == 0 || h == 0 || nw == 0 || nh == 0) return SSW_ERR_BAD_DIMS;
    CtxGuard g(ctx);
    if (nw == w && nh == h) {                         // the crate copies when the size is unchanged
        SSW_HIP_CHECK(hipMemcpyAsync(dev_out, dev_in, n_frames * w * h * 3, hipMemcpyDeviceToDevice, ctx->stream));
        return SSW_OK;
    }
    DeviceTaps vt, ht;
    SSW_TRY(get_taps(ctx, h, nh, &vt));
    SSW_TRY(get_taps(ctx, w, nw, &ht));
    const size_t chunk = effective_chunk(ctx, w, h, n_frames);
    const size_t tmp_bytes = resize_tmp_bytes(dev_in, std::min(chunk, n_frames), w, h, nw, nh, vt, ht, dev_out);
    if (tmp_bytes) SSW_TRY(grow(ctx->resize_tmp, tmp_bytes));       // two-pass fallback only (unaligned rows)
    for (size_t f0 = 0; f0 < n_frames; f0 += chunk) {
        const size_t n = std::min(chunk, n_frames - f0);
        StageTimer t(ctx, SSW_STAGE_RESIZE, ctx->stream, (double)n * 3.0 * ((double)w * h + (double)nw * nh));
        SSW_TRY(launch_resize_rgb8(ctx->stream, dev_in + f0 * w * h * 3, n, w, h, nw, nh, vt, ht,
                                   (float*)ctx->resize_tmp.p, dev_out + f0 * nw * nh * 3));
    }
    return SSW_OK;
}

// ---- Writer ---------------------------------------------------------------------------------
static int writer_create_impl(ssw_ctx* ctx, const void* rgb_hwc, int u8, size_t w, size_t h, const ssw_config* cfg,
                              ssw_writer** out) {
    if (!ctx || !rgb_hwc || !out) return SSW_ERR_BAD_ARG;
    *out = nullptr;
    SSW_TRY(check_config(cfg));
    if (w == 0 || h == 0) return SSW_ERR_BAD_DIMS;
    CtxGuard g(ctx);
    const size_t plane = w * h;
    ssw_writer* wr = new (std::nothrow) ssw_writer();
    if (!wr) return SSW_ERR_OUT_OF_MEMORY;
    wr->ctx = ctx; wr->w = w; wr->h = h; wr->cfg = *cfg;
    auto fail = [&](int rc) { ssw_writer_destroy(wr); return rc; };
    if (pool_get(ctx, plane * 4, (void**)&wr->y) != SSW_OK || pool_get(ctx, plane * 4, (void**)&wr->i) != SSW_OK ||
        pool_get(ctx, plane * 4, (void**)&wr->q) != SSW_OK)
        return fail(SSW_ERR_OUT_OF_MEMORY);
    const int rc = forward_from_host(ctx, rgb_hwc, u8, w, h, cfg->precision, wr->y, wr->i, wr->q);   // :308-313
    if (rc != SSW_OK) return fail(rc);
    *out = wr;
    return SSW_OK;
}

int ssw_writer_create(ssw_ctx* ctx, const float* rgb_hwc, size_t w, size_t h,
                      const ssw_config* cfg, ssw_writer** out) {
    return writer_create_impl(ctx, rgb_hwc, false, w, h, cfg, out);
}

int ssw_writer_create_rgb8(ssw_ctx* ctx, const uint8_t* rgb_hwc, size_t w, size_t h,
                           const ssw_config* cfg, ssw_writer** out) {
    return writer_create_impl(ctx, rgb_hwc, SSW_PIX_U8, w, h, cfg, out);
}

int ssw_writer_create_rgb16(ssw_ctx* ctx, const uint16_t* rgb_hwc, size_t w, size_t h,
                            const ssw_config* cfg, ssw_writer** out) {
    return writer_create_impl(ctx, rgb_hwc, SSW_PIX_U16, w, h, cfg, out);
}

int ssw_writer_coefficients(ssw_writer* wr, float* out_plane) {
    if (!wr || !out_plane) return SSW_ERR_BAD_ARG;
    if (wr->consumed) return SSW_ERR_CONSUMED;
    return ssw_copy_to_host(wr->ctx, out_plane, wr->y, wr->w * wr->h * sizeof(float));
}

static int writer_embed_impl(ssw_writer* wr, const float* const* marks, const size_t* lens, size_t n_marks, bool keep_original) {
    if (!wr || (n_marks && (!marks || !lens))) return SSW_ERR_BAD_ARG;
    if (wr->consumed) return SSW_ERR_CONSUMED;
    ssw_ctx* ctx = wr->ctx;
    CtxGuard g(ctx);
    const size_t plane = wr->w * wr->h;
    if (n_marks == 0) return SSW_OK;
    for (size_t m = 0; m < n_marks; ++m)
        if (lens[m] && !marks[m]) return SSW_ERR_BAD_ARG;
    if (wr->staging_in_flight) {                                      // a second embed(): the first one's copies first
        SSW_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        wr->staging_in_flight = false;
    }
    // zip(indices, mark) truncates every mark at w*h-1 entries (:396, :402)
    size_t total = 0, max_len = 0;
    for (size_t m = 0; m < n_marks; ++m) {
        const size_t len = std::min(lens[m], plane - 1);
        total += len; max_len = std::max(max_len, len);
    }
    if (max_len == 0) return SSW_OK;
    const size_t bytes_marks = (total * 4 + 15) / 16 * 16, bytes_tab = (n_marks * 4 + 15) / 16 * 16;
    wr->blob.assign((bytes_marks + 2 * bytes_tab) / 4, 0u);
    uint32_t* offs = wr->blob.data() + bytes_marks / 4;
    uint32_t* lns = offs + bytes_tab / 4;
    size_t at = 0;
    for (size_t m = 0; m < n_marks; ++m) {
        const size_t len = std::min(lens[m], plane - 1);
        offs[m] = (uint32_t)at; lns[m] = (uint32_t)len;
        if (len) std::memcpy(wr->blob.data() + at, marks[m], len * sizeof(float));
        at += len;
    }
    SSW_TRY(grow(ctx->small, bytes_marks + 2 * bytes_tab));
    char* base = (char*)ctx->small.p;
    // the host vector belongs to the handle and stays until it is destroyed: no host synchronisation here
    SSW_HIP_CHECK(hipMemcpyAsync(base, wr->blob.data(), bytes_marks + 2 * bytes_tab, hipMemcpyHostToDevice, ctx->stream));
    wr->staging_in_flight = true;
    if (keep_original && !wr->y0) {
        SSW_TRY(pool_get(ctx, plane * sizeof(float), (void**)&wr->y0));
        SSW_HIP_CHECK(hipMemcpyAsync(wr->y0, wr->y, plane * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (max_len > wr->idx_k) {                                        // :314, from the original coefficients
        pool_put(ctx, wr->idx, wr->idx_k * sizeof(uint32_t));
        wr->idx = nullptr; wr->idx_k = 0;
        SSW_TRY(pool_get(ctx, max_len * sizeof(uint32_t), (void**)&wr->idx));
        SSW_TRY(topk0(ctx, wr->y0 ? wr->y0 : wr->y, 1, wr->w, wr->h, wr->cfg.ordering, max_len, wr->idx));
        wr->idx_k = max_len;
    }
    {
        StageTimer t(ctx, SSW_STAGE_EMBED, ctx->stream);
        SSW_TRY(launch_embed(ctx->stream, wr->y, 1, plane, wr->idx, max_len, (const float*)base,
                             (const uint32_t*)(base + bytes_marks), (const uint32_t*)(base + bytes_marks + bytes_tab),
                             n_marks, max_len, max_len, wr->cfg.method, wr->cfg.alpha));
    }
    return SSW_OK;
}

int ssw_writer_embed(ssw_writer* wr, const float* const* marks, const size_t* lens, size_t n_marks) {
    return writer_embed_impl(wr, marks, lens, n_marks, true);
}

// Writer::result (:361-379): inverse transform with the colour conversion (and into_rgb8) in the epilogue of its last
// pass where the shape allows, straight into a device staging buffer, then one download.
static int writer_result_impl(ssw_writer* wr, void* out_rgb_hwc, bool u8_out) {
    if (!wr || !out_rgb_hwc) return SSW_ERR_BAD_ARG;
    if (wr->consumed) return SSW_ERR_CONSUMED;
    ssw_ctx* ctx = wr->ctx;
    CtxGuard g(ctx);
    const size_t plane = wr->w * wr->h;
    const size_t out_bytes = plane * 3 * (u8_out ? 1 : sizeof(float));
    SSW_TRY(grow(ctx->lane[0].plane[3], plane * sizeof(float)));
    ssw_ctx::FrameStage& fs = ctx->frame_stage[ctx->frame_stage_next++ & 1];
    SSW_TRY(frame_stage_events(fs));
    SSW_TRY(grow(fs.buf, out_bytes));
    Xform inv{SSW_DCT3, wr->cfg.precision, 1, wr->w, wr->h, wr->y, (float*)ctx->lane[0].plane[3].p};      // :368-374
    inv.iq_i = wr->i; inv.iq_q = wr->q; inv.rgb_out = fs.buf.p; inv.rgb_out_u8 = u8_out;                 // + :377
    bool fused_rgb = false;
    Chain ch;
    SSW_TRY(build_transform(ctx, ctx->lane[0], inv, ch, &fused_rgb));
    SSW_TRY(run_serial(ch, ctx->stream));
    if (!fused_rgb) {
        StageTimer t(ctx, SSW_STAGE_YIQ_TO_RGB, ctx->stream);
        if (u8_out) SSW_TRY(launch_yiq_to_rgb8(ctx->stream, wr->y, wr->i, wr->q, plane, (uint8_t*)fs.buf.p));
        else        SSW_TRY(launch_yiq_to_rgb(ctx->stream, wr->y, wr->i, wr->q, plane, (float*)fs.buf.p));   // :377
    }
    const int rc = download(ctx, out_rgb_hwc, fs.buf.p, out_bytes, ctx->stream);
    SSW_TRY(stage_consumed(ctx, fs));
    SSW_TRY(rc);
    wr->staging_in_flight = false;                                    // download() waited for the stream
    wr->consumed = true;                                              // `result(self)` consumes
    return SSW_OK;
}

int ssw_writer_result(ssw_writer* wr, float* out_rgb_hwc) { return writer_result_impl(wr, out_rgb_hwc, false); }
int ssw_writer_result_rgb8(ssw_writer* wr, uint8_t* out_rgb_hwc) { return writer_result_impl(wr, out_rgb_hwc, true); }

int ssw_writer_mark(ssw_writer* wr, const float* const* marks, const size_t* lens, size_t n_marks,
                    float* out_rgb_hwc) {
    if (!out_rgb_hwc) return SSW_ERR_BAD_ARG;
    SSW_TRY(writer_embed_impl(wr, marks, lens, n_marks, false));      // :356 (consumed next: no snapshot needed)
    return writer_result_impl(wr, out_rgb_hwc, false);                // :357
}

int ssw_writer_mark_rgb8(ssw_writer* wr, const float* const* marks, const size_t* lens, size_t n_marks,
                         uint8_t* out_rgb_hwc) {
    if (!out_rgb_hwc) return SSW_ERR_BAD_ARG;
    SSW_TRY(writer_embed_impl(wr, marks, lens, n_marks, false));
    return writer_result_impl(wr, out_rgb_hwc, true);
}

int ssw_writer_destroy(ssw_writer* wr) {
    if (!wr) return SSW_OK;
    ssw_ctx* ctx = wr->ctx;
    CtxGuard g(ctx);
    if (wr->staging_in_flight) (void)hipStreamSynchronize(ctx->stream);     // host vectors of embed() die with the handle
    const size_t plane = wr->w * wr->h;
    pool_put(ctx, wr->y, plane * 4);
    pool_put(ctx, wr->i, plane * 4);
    pool_put(ctx, wr->q, plane * 4);
    pool_put(ctx, wr->y0, plane * 4);
    pool_put(ctx, wr->idx, wr->idx_k * sizeof(uint32_t));
    delete wr;
    return SSW_OK;
}

// ---- Reader ---------------------------------------------------------------------------------
static int reader_ensure_indices(ssw_reader* rd, size_t k);
static int reader_create_impl(ssw_ctx* ctx, const void* rgb_hwc, int u8, size_t w, size_t h, int is_base,
                              const ssw_config* cfg, ssw_reader** out) {
    if (!ctx || !rgb_hwc || !out) return SSW_ERR_BAD_ARG;
    *out = nullptr;
    ssw_config c;
    ssw_config_default(&c);
    if (cfg) c = *cfg;
    else if (is_base) return SSW_ERR_BAD_ARG;                         // config.unwrap(), :482
    SSW_TRY(check_config(&c));
    if (w == 0 || h == 0) return SSW_ERR_BAD_DIMS;
    CtxGuard g(ctx);
    const size_t plane = w * h;
    ssw_reader* rd = new (std::nothrow) ssw_reader();
    if (!rd) return SSW_ERR_OUT_OF_MEMORY;
    rd->ctx = ctx; rd->w = w; rd->h = h; rd->is_base = is_base != 0; rd->cfg = c;
    auto fail = [&](int rc) { ssw_reader_destroy(rd); return rc; };
    if (!is_base && ctx->prune) {
        // upload only; transformed on first use (see ssw_reader)
        const size_t bytes = plane * 3 * pix_bytes(u8);
        rd->rgb_u8 = u8;
        bool fenced = false;
        const int rt = rgb_spare_take(ctx, bytes, &rd->rgb, &rd->rgb_uploaded, &fenced);
        if (rt != SSW_OK) return fail(rt);
        // a spare carries the point of the stream at which its previous reader was done with it; any other buffer may have
        // been handed back by a handle whose last kernels are still queued on the context's stream: behind all of it
        if (!fenced && hipEventRecord(rd->rgb_uploaded, ctx->stream) != hipSuccess) return fail(SSW_ERR_HIP);
        if (hipStreamWaitEvent(ctx->copy_stream, rd->rgb_uploaded, 0) != hipSuccess) return fail(SSW_ERR_HIP);
        int rc = upload(ctx, rd->rgb, rgb_hwc, bytes, ctx->copy_stream);
        if (rc != SSW_OK) return fail(rc);
        if (hipEventRecord(rd->rgb_uploaded, ctx->copy_stream) != hipSuccess ||
            hipStreamWaitEvent(ctx->stream, rd->rgb_uploaded, 0) != hipSuccess) return fail(SSW_ERR_HIP);
        untimed_work(ctx);
        *out = rd;
        return SSW_OK;
    }
    if (pool_get(ctx, plane * 4, (void**)&rd->y) != SSW_OK) return fail(SSW_ERR_OUT_OF_MEMORY);
    const int rc = forward_from_host(ctx, rgb_hwc, u8, w, h, c.precision, rd->y, nullptr, nullptr);   // :476-480
    if (rc != SSW_OK) return fail(rc);
    // The ordering is the reader's (:493) but its length is only known at extract(): queue it now for the length the
    // context's last extraction used, so that it runs while the derived frame is still crossing PCIe instead of after it
    // (a list that turns out too short is redone at extract(); a longer one serves as it is).
    if (is_base && ctx->expected_k && ctx->expected_k <= plane - 1 && ctx->expected_k <= select_max_k() && tuning(TUNE_SPECULATE_K))
        (void)reader_ensure_indices(rd, ctx->expected_k);
    *out = rd;
    return SSW_OK;
}

// a derived reader's full transform, on demand (Reader::derived :475-480)
static int reader_ensure_coefficients(ssw_reader* rd) {
    if (rd->y) return SSW_OK;
    ssw_ctx* ctx = rd->ctx;
    const size_t plane = rd->w * rd->h;
    // rd->y is set only once the whole chain is enqueued: a failure below (out of memory in grow() is the realistic
    // one) leaves the reader as it was -- still RGB, a retry works -- instead of pointing at a recycled plane
    float* y = nullptr;
    SSW_TRY(pool_get(ctx, plane * 4, (void**)&y));
    auto run = [&]() -> int {
        SSW_TRY(grow(ctx->lane[0].plane[3], plane * sizeof(float)));
        Chain ch;
        SSW_TRY(build_forward_from_rgb(ctx, ctx->lane[0], rd->cfg.precision, rd->rgb, rd->rgb_u8, 1, rd->w, rd->h, y,
                                       nullptr, nullptr, (float*)ctx->lane[0].plane[3].p, ch));
        return run_serial(ch, ctx->stream);
    };
    const int rc = run();
    if (rc != SSW_OK) { pool_put(ctx, y, plane * 4); return rc; }
    rd->y = y;
    rgb_spare_give(ctx, rd->rgb, plane * 3 * pix_bytes(rd->rgb_u8), rd->rgb_uploaded);      // reuse is ordered behind this point of the stream
    rd->rgb = nullptr;
    rd->rgb_uploaded = nullptr;
    return SSW_OK;
}

int ssw_reader_create(ssw_ctx* ctx, const float* rgb_hwc, size_t w, size_t h, int is_base,
                      const ssw_config* cfg, ssw_reader** out) {
    return reader_create_impl(ctx, rgb_hwc, false, w, h, is_base, cfg, out);
}

int ssw_reader_create_rgb8(ssw_ctx* ctx, const uint8_t* rgb_hwc, size_t w, size_t h, int is_base,
                           const ssw_config* cfg, ssw_reader** out) {
    return reader_create_impl(ctx, rgb_hwc, SSW_PIX_U8, w, h, is_base, cfg, out);
}

int ssw_reader_create_rgb16(ssw_ctx* ctx, const uint16_t* rgb_hwc, size_t w, size_t h, int is_base,
                            const ssw_config* cfg, ssw_reader** out) {
    return reader_create_impl(ctx, rgb_hwc, SSW_PIX_U16, w, h, is_base, cfg, out);
}

int ssw_reader_coefficients(ssw_reader* rd, float* out_plane) {
    if (!rd || !out_plane) return SSW_ERR_BAD_ARG;
    CtxGuard g(rd->ctx);
    SSW_TRY(reader_ensure_coefficients(rd));
    return ssw_copy_to_host(rd->ctx, out_plane, rd->y, rd->w * rd->h * sizeof(float));
}

static int reader_ensure_indices(ssw_reader* rd, size_t k) {
    if (!rd->is_base) return SSW_ERR_NOT_BASE;                        // base.unwrap(), :507 / :530
    if (k > rd->w * rd->h - 1) return SSW_ERR_K_TOO_LARGE;
    if (k <= rd->idx_k) return SSW_OK;
    ssw_ctx* ctx = rd->ctx;
    pool_put(ctx, rd->idx, rd->idx_k * sizeof(uint32_t));
    rd->idx = nullptr; rd->idx_k = 0;
    uint32_t* idx = nullptr;
    SSW_TRY(pool_get(ctx, k * sizeof(uint32_t), (void**)&idx));
    const int rc = topk0(ctx, rd->y, 1, rd->w, rd->h, rd->cfg.ordering, k, idx);           // :493
    if (rc != SSW_OK) { pool_put(ctx, idx, k * sizeof(uint32_t)); return rc; }          // back under the size it was taken with
    rd->idx = idx;
    rd->idx_k = k;
    return SSW_OK;
}

int ssw_reader_indices(ssw_reader* rd, size_t k, uint64_t* out) {
    if (!rd || (k && !out)) return SSW_ERR_BAD_ARG;
    CtxGuard g(rd->ctx);
    if (k == 0) return rd->is_base ? SSW_OK : SSW_ERR_NOT_BASE;
    SSW_TRY(reader_ensure_indices(rd, k));
    ssw_ctx* ctx = rd->ctx;
    ctx->expected_k = k;
    SSW_TRY(grow(ctx->small, k * sizeof(uint64_t)));
    SSW_TRY(launch_widen_indices(ctx->stream, rd->idx, k, (uint64_t*)ctx->small.p));
    return ssw_copy_to_host(ctx, out, ctx->small.p, k * sizeof(uint64_t));
}

int ssw_reader_extract(ssw_reader* base, ssw_reader* derived, float* out, size_t k) {
    if (!base || !derived || (k && !out)) return SSW_ERR_BAD_ARG;
    if (!base->is_base) return SSW_ERR_NOT_BASE;                                          // :530
    if (base->ctx != derived->ctx) return SSW_ERR_BAD_ARG;
    if (derived->w * derived->h != base->w * base->h) return SSW_ERR_LENGTH_MISMATCH;     // :550-552
    const size_t plane = base->w * base->h;
    if (k >= plane) return SSW_ERR_K_TOO_LARGE;                                           // :553-555
    if (k == 0) return SSW_OK;
    ssw_ctx* ctx = base->ctx;
    CtxGuard g(ctx);
    SSW_TRY(reader_ensure_indices(base, k));
    ctx->expected_k = k;
    SSW_TRY(grow(ctx->small, k * sizeof(float)));
    if (!derived->y && derived->w == base->w && derived->h == base->h) {
        // the derived frame is still RGB: transform it only where the first k indices of the base reader read it
        bool pruned = false;
        uint32_t* info = nullptr;
        SSW_TRY(extract_single_pruned(ctx, derived->cfg.precision, derived->rgb, derived->rgb_u8, base->w, base->h, base->y,
                                      base->idx, k, base->cfg.method, base->cfg.alpha, (float*)ctx->small.p, &info, &pruned));
        if (pruned) {
            uint32_t overflow = 1;
            SSW_HIP_CHECK(hipMemcpyAsync(&overflow, info, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            SSW_TRY(ssw_copy_to_host(ctx, out, ctx->small.p, k * sizeof(float)));      // waits for the stream
            if (overflow == 0) return SSW_OK;
        }
    }
    SSW_TRY(reader_ensure_coefficients(derived));
    {
        StageTimer t(ctx, SSW_STAGE_EXTRACT, ctx->stream);
        // cached list may be longer than k: its first k entries are the first k of the order
        SSW_TRY(launch_extract(ctx->stream, base->y, derived->y, 1, plane, base->idx, k, base->cfg.method,
                               base->cfg.alpha, (float*)ctx->small.p));
    }
    return ssw_copy_to_host(ctx, out, ctx->small.p, k * sizeof(float));
}

int ssw_reader_destroy(ssw_reader* rd) {
    if (!rd) return SSW_OK;
    ssw_ctx* ctx = rd->ctx;
    CtxGuard g(ctx);
    pool_put(ctx, rd->y, rd->w * rd->h * 4);
    pool_put(ctx, rd->idx, rd->idx_k * sizeof(uint32_t));
    rgb_spare_give(ctx, rd->rgb, rd->w * rd->h * 3 * pix_bytes(rd->rgb_u8), rd->rgb_uploaded);
    delete rd;
    return SSW_OK;
}

// ---- Tester ---------------------------------------------------------------------------------
int ssw_similarity(ssw_ctx* ctx, const float* extracted, size_t n_extracted, const float* mark,
                   size_t n_mark, float* out_similarity) {
    if (!ctx || !out_similarity || (n_extracted && !extracted) || (n_mark && !mark)) return SSW_ERR_BAD_ARG;
    if (n_extracted != n_mark) return SSW_ERR_LENGTH_MISMATCH;                            // :697-700
    CtxGuard g(ctx);
    const size_t k = n_extracted;
    const size_t bytes = (k * 4 + 15) / 16 * 16;
    SSW_TRY(grow(ctx->small, 2 * bytes + 16));
    char* base = (char*)ctx->small.p;
    if (k) {
        SSW_HIP_CHECK(hipMemcpyAsync(base, extracted, k * 4, hipMemcpyHostToDevice, ctx->stream));
        SSW_HIP_CHECK(hipMemcpyAsync(base + bytes, mark, k * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    {
        StageTimer t(ctx, SSW_STAGE_SIMILARITY, ctx->stream);
        SSW_TRY(launch_similarity(ctx->stream, (const float*)base, (const float*)(base + bytes), 1, k,
                                  (float*)(base + 2 * bytes)));
    }
    return ssw_copy_to_host(ctx, out_similarity, base + 2 * bytes, sizeof(float));
}

// ---- synthetic frames -----------------------------------------------------------------------
int ssw_synth_frames(ssw_ctx* ctx, uint32_t seed, uint32_t first_frame, size_t n_frames, size_t w,
                     size_t h, float* dev_rgb) {
    if (!ctx || !dev_rgb) return SSW_ERR_BAD_ARG;
    CtxGuard g(ctx);
    return launch_synth(ctx->stream, seed, first_frame, n_frames, w, h, dev_rgb);
}

}  // extern "C"
