// Host-image STREAMING batch entry points: n frames that live on the host go through the batch pipelines in groups,
// with the upload of group g + 1, the kernels of group g and the download of group g - 1 in flight together.
//
// The reference's callers loop over images on the host -- examples/main.rs:271-278 (`watermark`: per file Writer::new ->
// mark -> into_rgb8 -> save) and :383-415 (`test`: per file Reader::base / derived -> extract -> similarity) -- and the
// single-image handles serve exactly that loop, one synchronous call per image: upload, 2 ms of single-frame launches,
// download, one after the other (2.4 Gpix/s from pinned 4K frames against 10 device-resident).  Here the same loop
// is ONE call: the frames of a group share the GEMM launches (>= 8 frames per pass: batch-sized tiles instead of
// single-frame ones), PCIe runs both ways beside them (two copy streams, two alternating device buffers per
// direction), and the caller's buffers -- pinned ones are the DMA source / target themselves, any other goes through
// the staging ring of transfer.hip -- belong to the library until the call returns.  Results are bit-identical to the
// handles (tests/test_stream_gpu.py): the groups run the same batch code as ssw_batch_*_rgb8.
#include "ssw_host.hpp"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace ssw {
namespace host {
namespace {

// frames per group: >= 8, or what fills ~2^26 pixels (1080p: 32)
size_t stream_group(size_t w, size_t h, size_t n_frames) {
    const char* e = std::getenv("SSW_STREAM_GROUP");                  // tests: force a group size (read per call)
    const size_t forced = e ? (size_t)std::atoll(e) : (size_t)0;
    size_t g = forced ? forced : std::max<size_t>(8, ((size_t)1 << 26) / std::max<size_t>(w * h, 1));
    return std::max<size_t>(1, std::min(g, n_frames));
}
// The groups of a call: first frame and size.  The pipeline's fill is the upload of the first group and its drain the
// download of the last one, neither overlapped with anything -- so the call ramps up (G/4, G/2, then groups of G) and,
// when `ramp_down` (embed: frames come back), down again; the short groups' less efficient launches run in the shadow
// of their neighbours' transfers.  SSW_STREAM_RAMP=0: equal groups.
struct Group { size_t f0, n; };
std::vector<Group> stream_groups(size_t w, size_t h, size_t n_frames, bool ramp_down) {
    const size_t G = stream_group(w, h, n_frames);
    const char* e = std::getenv("SSW_STREAM_RAMP");
    const bool ramp = !(e && std::atoi(e) == 0);
    std::vector<size_t> head, tail;
    if (ramp && G >= 4) {
        for (size_t s = G / 4; s < G; s *= 2) head.push_back(s);
        if (ramp_down) tail.assign(head.rbegin(), head.rend());
    }
    size_t edge = 0;
    for (size_t s : head) edge += s;
    for (size_t s : tail) edge += s;
    if (n_frames < edge + G) { head.clear(); tail.clear(); edge = 0; }      // too few frames to ramp
    std::vector<Group> out;
    size_t f = 0;
    for (size_t s : head) { out.push_back({f, s}); f += s; }
    size_t tail_sum = 0;
    for (size_t s : tail) tail_sum += s;
    while (f < n_frames - tail_sum) { const size_t n = std::min(G, n_frames - tail_sum - f); out.push_back({f, n}); f += n; }
    for (size_t s : tail) { out.push_back({f, s}); f += s; }
    return out;
}

constexpr int NB = ssw_ctx::HostStream::NB;      // device buffers per direction: uploads run up to NB - 1 groups ahead of the kernels

int stream_setup(ssw_ctx* ctx) {
    if (!ctx->down_stream) SSW_HIP_CHECK(hipStreamCreateWithFlags(&ctx->down_stream, hipStreamNonBlocking));
    for (int s = 0; s < NB; ++s)
        for (hipEvent_t* e : {&ctx->hs.up_done[s], &ctx->hs.k_done[s], &ctx->hs.down_done[s]})
            if (!*e) SSW_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    return SSW_OK;
}

// whatever happens, nothing of this call may still be in flight when it returns (the caller's buffers are DMA
// sources / targets): waits for the three streams
int stream_drain(ssw_ctx* ctx, int rc) {
    ctx->hs.active = false;
    const hipError_t a = hipStreamSynchronize(ctx->copy_stream), b = hipStreamSynchronize(ctx->stream), c = hipStreamSynchronize(ctx->down_stream);
    if (rc != SSW_OK) { (void)hipGetLastError(); return rc; }
    SSW_HIP_CHECK(a); SSW_HIP_CHECK(b); SSW_HIP_CHECK(c);
    return SSW_OK;
}

}  // namespace

int stream_embed_rgb8(ssw_ctx* ctx, const ssw_config* cfg, const uint8_t* const* host_frames, size_t n_frames, size_t w, size_t h,
                      const float* host_marks, size_t k, uint8_t* const* host_out) {
    if (!ctx || !cfg || (n_frames && (!host_frames || !host_out)) || (n_frames && k && !host_marks)) return SSW_ERR_BAD_ARG;
    SSW_TRY(check_config(cfg));
    if (w == 0 || h == 0) return SSW_ERR_BAD_DIMS;
    if (n_frames == 0) return SSW_OK;
    for (size_t f = 0; f < n_frames; ++f) if (!host_frames[f] || !host_out[f]) return SSW_ERR_BAD_ARG;
    CtxGuard guard(ctx);
    SSW_TRY(stream_setup(ctx));
    ssw_ctx::HostStream& hs = ctx->hs;
    struct Active { bool& a; explicit Active(bool& x) : a(x) { a = true; } ~Active() { a = false; } } active_guard(hs.active);
    const size_t fb = w * h * 3, G = stream_group(w, h, n_frames);
    const std::vector<Group> groups = stream_groups(w, h, n_frames, true);
    const size_t n_groups = groups.size();
    for (int s = 0; s < (int)std::min<size_t>(n_groups, NB); ++s) { SSW_TRY(grow(hs.in[s], G * fb)); SSW_TRY(grow(hs.out[s], G * fb)); }
    SSW_TRY(grow(hs.marks, std::max<size_t>(n_frames * k * sizeof(float), 16)));
    auto body = [&]() -> int {
        bool as = false;
        if (k) SSW_TRY(upload_nowait(ctx, hs.marks.p, host_marks, n_frames * k * sizeof(float), ctx->stream, &as));
        auto h2d = [&](size_t g) -> int {
            const int s = (int)(g % NB);
            const size_t f0 = groups[g].f0, n = groups[g].n;
            if (g >= (size_t)NB) SSW_HIP_CHECK(hipStreamWaitEvent(ctx->copy_stream, hs.k_done[s], 0));      // kernels of group g - NB read in[s]
            for (size_t j = 0; j < n; ++j) SSW_TRY(upload_nowait(ctx, (char*)hs.in[s].p + j * fb, host_frames[f0 + j], fb, ctx->copy_stream, &as));
            SSW_HIP_CHECK(hipEventRecord(hs.up_done[s], ctx->copy_stream));
            return SSW_OK;
        };
        auto d2h = [&](size_t g) -> int {
            const int s = (int)(g % NB);
            const size_t f0 = groups[g].f0, n = groups[g].n;
            SSW_HIP_CHECK(hipStreamWaitEvent(ctx->down_stream, hs.k_done[s], 0));
            for (size_t j = 0; j < n; ++j) SSW_TRY(download_nowait(ctx, host_out[f0 + j], (const char*)hs.out[s].p + j * fb, fb, ctx->down_stream, &as));
            SSW_HIP_CHECK(hipEventRecord(hs.down_done[s], ctx->down_stream));
            return SSW_OK;
        };
        for (size_t g = 0; g + 1 < (size_t)NB && g < n_groups; ++g) SSW_TRY(h2d(g));
        for (size_t g = 0; g < n_groups; ++g) {
            const int s = (int)(g % NB);
            const size_t f0 = groups[g].f0, n = groups[g].n;
            SSW_HIP_CHECK(hipStreamWaitEvent(ctx->stream, hs.up_done[s], 0));
            if (g >= (size_t)NB) SSW_HIP_CHECK(hipStreamWaitEvent(ctx->stream, hs.down_done[s], 0));   // download of group g - NB reads out[s]
            untimed_work(ctx);                 // the group's first stage timer starts behind these waits
            SSW_TRY(batch_embed_impl(ctx, cfg, hs.in[s].p, SSW_PIX_U8, n, w, h, (const float*)hs.marks.p + f0 * k, k, hs.out[s].p, true, nullptr, nullptr));
            SSW_HIP_CHECK(hipEventRecord(hs.k_done[s], ctx->stream));
            untimed_work(ctx);
            if (g + NB - 1 < n_groups) SSW_TRY(h2d(g + NB - 1));      // staged (pageable) frames: the host copies while group g computes
            // the download of group g - 1 is issued AFTER the kernels of group g: a pageable output buffer makes
            // download_nowait a blocking staged copy, and the device must have its next group queued before the host sleeps
            if (g >= 1) SSW_TRY(d2h(g - 1));
        }
        SSW_TRY(d2h(n_groups - 1));
        return SSW_OK;
    };
    return stream_drain(ctx, body());
}

int stream_extract_rgb8(ssw_ctx* ctx, const ssw_config* cfg, const uint8_t* const* host_base, const uint8_t* const* host_derived,
                        size_t n_frames, size_t w, size_t h, size_t k, float* host_extracted, const float* host_marks, float* host_sims) {
    if (!ctx || !cfg || (n_frames && (!host_base || !host_derived)) || (n_frames && k && !host_extracted)) return SSW_ERR_BAD_ARG;
    if ((host_marks == nullptr) != (host_sims == nullptr)) return SSW_ERR_BAD_ARG;
    SSW_TRY(check_config(cfg));
    if (w == 0 || h == 0) return SSW_ERR_BAD_DIMS;
    if (k >= w * h) return SSW_ERR_K_TOO_LARGE;                        // :553-555
    if (n_frames == 0) return SSW_OK;
    for (size_t f = 0; f < n_frames; ++f) if (!host_base[f] || !host_derived[f]) return SSW_ERR_BAD_ARG;
    CtxGuard guard(ctx);
    SSW_TRY(stream_setup(ctx));
    ssw_ctx::HostStream& hs = ctx->hs;
    struct Active { bool& a; explicit Active(bool& x) : a(x) { a = true; } ~Active() { a = false; } } active_guard(hs.active);
    const size_t fb = w * h * 3, G = stream_group(w, h, n_frames);
    const std::vector<Group> groups = stream_groups(w, h, n_frames, false);
    const size_t n_groups = groups.size();
    for (int s = 0; s < (int)std::min<size_t>(n_groups, NB); ++s) { SSW_TRY(grow(hs.in[s], G * fb)); SSW_TRY(grow(hs.in2[s], G * fb)); }
    SSW_TRY(grow(hs.ext, std::max<size_t>(n_frames * k * sizeof(float), 16)));
    SSW_TRY(grow(hs.sims, n_frames * sizeof(float)));
    if (host_marks) SSW_TRY(grow(hs.marks, std::max<size_t>(n_frames * k * sizeof(float), 16)));
    auto body = [&]() -> int {
        bool as = false;
        if (host_marks && k) SSW_TRY(upload_nowait(ctx, hs.marks.p, host_marks, n_frames * k * sizeof(float), ctx->stream, &as));
        auto h2d = [&](size_t g) -> int {
            const int s = (int)(g % NB);
            const size_t f0 = groups[g].f0, n = groups[g].n;
            if (g >= (size_t)NB) SSW_HIP_CHECK(hipStreamWaitEvent(ctx->copy_stream, hs.k_done[s], 0));
            for (size_t j = 0; j < n; ++j) {
                SSW_TRY(upload_nowait(ctx, (char*)hs.in[s].p + j * fb, host_base[f0 + j], fb, ctx->copy_stream, &as));
                SSW_TRY(upload_nowait(ctx, (char*)hs.in2[s].p + j * fb, host_derived[f0 + j], fb, ctx->copy_stream, &as));
            }
            SSW_HIP_CHECK(hipEventRecord(hs.up_done[s], ctx->copy_stream));
            return SSW_OK;
        };
        for (size_t g = 0; g + 1 < (size_t)NB && g < n_groups; ++g) SSW_TRY(h2d(g));
        for (size_t g = 0; g < n_groups; ++g) {
            const int s = (int)(g % NB);
            const size_t f0 = groups[g].f0, n = groups[g].n;
            // the following groups' frames are requested BEFORE this group's kernels: with pruning on, the batch call below
            // looks at its overflow flags once at its end (one host wait per group) -- PCIe keeps running through it
            if (g + NB - 1 < n_groups) SSW_TRY(h2d(g + NB - 1));
            SSW_HIP_CHECK(hipStreamWaitEvent(ctx->stream, hs.up_done[s], 0));
            untimed_work(ctx);
            SSW_TRY(batch_extract_impl(ctx, cfg, hs.in[s].p, hs.in2[s].p, SSW_PIX_U8, n, w, h, k, (float*)hs.ext.p + f0 * k,
                                       host_marks ? (const float*)hs.marks.p + f0 * k : nullptr, host_sims ? (float*)hs.sims.p + f0 : nullptr));
            SSW_HIP_CHECK(hipEventRecord(hs.k_done[s], ctx->stream));
            untimed_work(ctx);
        }
        SSW_HIP_CHECK(hipStreamWaitEvent(ctx->down_stream, hs.k_done[(n_groups - 1) % NB], 0));
        if (k) SSW_TRY(download_nowait(ctx, host_extracted, hs.ext.p, n_frames * k * sizeof(float), ctx->down_stream, &as));
        if (host_sims) SSW_TRY(download_nowait(ctx, host_sims, hs.sims.p, n_frames * sizeof(float), ctx->down_stream, &as));
        return SSW_OK;
    };
    return stream_drain(ctx, body());
}

}  // namespace host
}  // namespace ssw

extern "C" {

int ssw_batch_embed_host_rgb8(ssw_ctx* ctx, const ssw_config* cfg, const uint8_t* const* host_frames, size_t n_frames, size_t w, size_t h,
                              const float* host_marks, size_t k, uint8_t* const* host_out) {
    return ssw::host::stream_embed_rgb8(ctx, cfg, host_frames, n_frames, w, h, host_marks, k, host_out);
}

int ssw_batch_extract_host_rgb8(ssw_ctx* ctx, const ssw_config* cfg, const uint8_t* const* host_base, const uint8_t* const* host_derived,
                                size_t n_frames, size_t w, size_t h, size_t k, float* host_extracted, const float* host_marks, float* host_sims) {
    return ssw::host::stream_extract_rgb8(ctx, cfg, host_base, host_derived, n_frames, w, h, k, host_extracted, host_marks, host_sims);
}

}  // extern "C"
