// Orchestration of the reference's call stacks on the device: the 2-D transform as a chain of stages
// (operand pre-passes = HBM-bound, basis GEMMs = MFMA-bound), the pruned transform of derived frames, and
// the two-lane pipelines behind ssw_batch_embed / ssw_batch_extract.
//
// Two lanes, two streams.  A chunk's stages depend on each other in sequence, and they alternate between
// HBM-bound and MFMA-bound kernels; a single stream therefore leaves the matrix cores idle during ~13 % of
// a 4K step (25 % at full HD) and HBM idle for the rest.  The batch pipelines keep two chunks in flight,
// each with its own workspace ("lane"): every GEMM stage goes to the context's stream, every HBM-bound
// stage to `aux_stream`, stages are enqueued round-robin over the lanes, and a lane crosses from one
// stream to the other through an event.  GEMM launches never overlap each other (one stream), so their
// event timings stay meaningful; what overlaps is one lane's pre-pass / selection / colour conversion with
// the other lane's GEMMs.  Results are bit-identical to the serial order (same kernels, same chunks).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "ssw_host.hpp"
#include "dct_pair_split.hpp"

#include <array>
#include <cstdlib>

namespace ssw {
namespace host {

// ---- small helpers ------------------------------------------------------------------------------------
namespace { thread_local ssw_ctx* tl_ctx = nullptr; }
CtxGuard::CtxGuard(ssw_ctx* ctx) : dg(ctx->device), prev(tl_ctx) {
    tl_ctx = ctx;
    if (prev != ctx) { ctx->tail_event = nullptr; ctx->tail_fresh = false; }     // timer events are shared inside one call only
    // buffers retired by grow() during the previous call: nothing built then is still to be enqueued (hipFree waits for
    // work in flight); nested guards (prev == ctx) leave them to the outermost one
    if (prev != ctx && !ctx->retired.empty()) {
        for (void* p : ctx->retired) (void)hipFree(p);
        ctx->retired.clear();
    }
}
CtxGuard::~CtxGuard() { tl_ctx = prev; }

size_t plane_pool_flush(ssw_ctx* ctx) {
    size_t bytes = ctx->plane_pool_bytes;
    for (auto& kv : ctx->plane_pool) (void)hipFree(kv.second);
    ctx->plane_pool.clear();
    ctx->plane_pool_bytes = 0;
    for (auto& sp : ctx->rgb_spares) { (void)hipFree(sp.p); (void)hipEventDestroy(sp.released); bytes += sp.bytes; }
    ctx->rgb_spares.clear();
    return bytes;
}

// The device ring of the host-image streaming entry points (ssw_stream.hip: 3 slots x two buffers of a group of frames,
// ~1.2 GB at the default group size) stays allocated between calls; under memory pressure it goes back like the plane pool.
size_t host_stream_release(ssw_ctx* ctx) {
    ssw_ctx::HostStream& hs = ctx->hs;
    if (hs.active) return 0;
    size_t bytes = 0;
    for (int s = 0; s < ssw_ctx::HostStream::NB; ++s)
        for (ssw_ctx::Buf* b : {&hs.in[s], &hs.in2[s], &hs.out[s]})
            if (b->p) { bytes += b->bytes; (void)hipFree(b->p); b->p = nullptr; b->bytes = 0; }
    return bytes;
}

int dev_malloc(void** p, size_t bytes) {
    *p = nullptr;
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e == hipSuccess) return SSW_OK;
    (void)hipGetLastError();
    if (tl_ctx && (!tl_ctx->plane_pool.empty() || !tl_ctx->rgb_spares.empty())) {      // spare planes of destroyed handles: give them back first
        (void)plane_pool_flush(tl_ctx);
        e = hipMalloc(p, bytes ? bytes : 16);
        if (e == hipSuccess) return SSW_OK;
        (void)hipGetLastError();
    }
    if (tl_ctx && host_stream_release(tl_ctx)) {                   // then the idle streaming ring
        e = hipMalloc(p, bytes ? bytes : 16);
        if (e == hipSuccess) return SSW_OK;
        (void)hipGetLastError();
    }
    *p = nullptr;
    set_last_error(std::string("hipMalloc(") + std::to_string(bytes) + " bytes): " + hipGetErrorString(e));
    return SSW_ERR_OUT_OF_MEMORY;
}

int grow(ssw_ctx::Buf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return SSW_OK;
    // A buffer that grows in the middle of a call may already be captured by stages built earlier in the same chain (a
    // row pass that exchanges A1 through operand[2], then a column pass that asks for a larger operand[2]): the old
    // allocation is retired, not freed -- those stages keep working on it, the later ones use the new one -- and the next
    // entry into the library frees it (CtxGuard).  Without a context in scope: free now (hipFree waits for work in flight).
    if (b.p) {
        if (tl_ctx) tl_ctx->retired.push_back(b.p);
        else SSW_HIP_CHECK(hipFree(b.p));
        b.p = nullptr; b.bytes = 0;
    }
    SSW_ALLOC(&b.p, bytes);
    b.bytes = bytes ? bytes : 16;
    return SSW_OK;
}

void release(ssw_ctx::Buf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

void release_select(SelectWorkspace& s) {
    if (s.hist) (void)hipFree(s.hist);
    if (s.ctrl) (void)hipFree(s.ctrl);
    if (s.cand) (void)hipFree(s.cand);
    s = SelectWorkspace();
}

int grow_select(hipStream_t st, SelectWorkspace& s, size_t frames, size_t k) {
    const size_t want = select_cand_capacity(k);
    if (s.frames >= frames && s.cap >= want && s.hist) return SSW_OK;
    const size_t nf = std::max(frames, s.frames), cap = std::max(want, s.cap);
    release_select(s);
    SSW_ALLOC(&s.hist, nf * 2048 * sizeof(uint32_t));
    SSW_ALLOC(&s.ctrl, nf * 4 * sizeof(uint32_t));
    SSW_ALLOC(&s.cand, nf * cap * sizeof(uint64_t));
    SSW_HIP_CHECK(hipMemsetAsync(s.hist, 0, nf * 2048 * sizeof(uint32_t), st));   // see select.hip:
    SSW_HIP_CHECK(hipMemsetAsync(s.ctrl, 0, nf * 4 * sizeof(uint32_t), st));      // zero between uses
    untimed_work(tl_ctx);
    s.frames = nf;
    s.cap = cap;
    return SSW_OK;
}

// Stage timers share their boundary events (r5): a stage that starts right after another one ended on the same stream --
// nothing enqueued in between: the stages of a chain, a "main launch" timer nested at the start or the end of its pass --
// takes the event that is already in the stream instead of recording a second one.  An event record is a barrier in the
// queue (~3 us during which the next kernel cannot overlap the previous one's tail): a single-frame embed had 28 of them in
// 0.6 ms of kernels.  ctx->tail_event is the last event recorded by a timer, valid while ctx->tail_fresh: cleared by the next
// timer's own launches, by untimed_work() (ssw_host.hpp) -- called wherever something is enqueued outside a timer: the lanes'
// hops / done records / stagger waits, the pruned transform's table launches, basis generation, workspace memsets, copies,
// the upload / download hand-offs of the handle and streaming entry points -- and at every entry into the library.
static hipEvent_t timer_event(ssw_ctx* ctx, hipStream_t st) {
    if (ctx->tail_event && ctx->tail_fresh && ctx->tail_stream == st) return ctx->tail_event;
    hipEvent_t e = nullptr;
    if (!ctx->free_events.empty()) { e = ctx->free_events.back(); ctx->free_events.pop_back(); }
    else if (hipEventCreateWithFlags(&e, hipEventReleaseToDevice) != hipSuccess) return nullptr;
    if (hipEventRecord(e, st) != hipSuccess) { ctx->free_events.push_back(e); return nullptr; }
    ctx->tail_event = e; ctx->tail_stream = st; ctx->tail_fresh = true;
    return e;
}
StageTimer::StageTimer(ssw_ctx* c, int s, hipStream_t stream, double work, int alias_stage) : ctx(c), stage(s), st(stream), alias(alias_stage) {
    if (!ctx->timing) return;
    ctx->stage_work[stage] += work;
    if (alias >= 0) ctx->stage_work[alias] += work;
    const bool gemm = stage == SSW_STAGE_DCT_ROW || stage == SSW_STAGE_DCT_COL || stage == SSW_STAGE_DCT_ROW_MAIN || stage == SSW_STAGE_DCT_COL_MAIN;
    if (!gemm) ctx->stage_bytes[stage] += work;
    a = timer_event(ctx, st);
    ctx->tail_fresh = false;                  // the stage's launches follow: its end needs an event of its own
    armed = a != nullptr;
}
void StageTimer::traffic(double bytes) {
    if (ctx->timing) ctx->stage_bytes[stage] += bytes;
}
StageTimer::~StageTimer() {
    if (!ctx->timing || !armed) return;
    // (a nested timer that just ended left a fresh event: the outer stage ends at the same point)
    b = timer_event(ctx, st);
    if (b) ctx->pending.push_back({stage, a, b, alias});
}

int flush_timers(ssw_ctx* ctx) {
    std::vector<hipEvent_t> used;
    for (auto& p : ctx->pending) {
        float ms = 0.f;
        SSW_HIP_CHECK(hipEventSynchronize(p.b));
        SSW_HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
        ctx->stage_ms[p.stage] += ms;
        ctx->stage_launches[p.stage] += 1;
        if (p.alias >= 0) { ctx->stage_ms[p.alias] += ms; ctx->stage_launches[p.alias] += 1; }
        used.push_back(p.a);
        used.push_back(p.b);
    }
    ctx->pending.clear();
    // shared boundary events appear in two entries: each goes back to the pool once
    std::sort(used.begin(), used.end());
    used.erase(std::unique(used.begin(), used.end()), used.end());
    for (hipEvent_t e : used) ctx->free_events.push_back(e);
    ctx->tail_event = nullptr; ctx->tail_fresh = false;
    return SSW_OK;
}

// Bases are generated on the context's stream, the stream every GEMM launch uses.
int get_basis(ssw_ctx* ctx, size_t n, bool inverse, bool f64, int kind, const void** out) {
    auto key = std::make_tuple(n, inverse, f64, kind);
    auto it = ctx->basis.find(key);
    if (it != ctx->basis.end()) { *out = it->second; return SSW_OK; }
    void* p = nullptr;
    // kinds 5..8: split odd half bases (cosE, sinE, cosO, sinO), 9: the rotation table, 10: sinE for the launches (row 0 =
    // row n/8: class E's first and last pair share a slot) -- f64 only
    const int split_which = kind == 10 ? 4 : kind - 5;
    const size_t elems = kind == 0 ? n * dense_basis_kpad(n)
                       : kind == 9 ? n / 2
                       : kind >= 5 ? dct_pair_split_basis_rows(n, split_which) * dct_pair_split_kpad(n)
                       : kind >= 3 ? (n / 2) * dct_pair_kpad(f64, n) : (n / 2) * half_basis_kpad(n);
    if (kind >= 5 && !f64) return SSW_ERR_BAD_ARG;
    SSW_ALLOC(&p, std::max<size_t>(elems, 1) * (f64 ? sizeof(double) : sizeof(float)));
    int rc = kind == 9 ? launch_make_rot_table(ctx->stream, n, (double*)p)
             : kind >= 5 ? launch_make_split_basis_blocked(ctx->stream, n, inverse, split_which, (double*)p)
             : kind >= 3 ? launch_make_half_basis_blocked(ctx->stream, f64, n, inverse, kind - 3, p)
             : kind != 0 ? (f64 ? launch_make_half_basis_f64(ctx->stream, n, inverse, kind - 1, (double*)p)
                              : launch_make_half_basis_f32(ctx->stream, n, inverse, kind - 1, (float*)p))
             : f64     ? launch_make_basis_f64(ctx->stream, n, inverse, (double*)p)
                       : launch_make_basis_f32(ctx->stream, n, inverse, (float*)p);
    untimed_work(ctx);
    if (rc != SSW_OK) { (void)hipFree(p); return rc; }
    ctx->basis[key] = p;
    *out = p;
    return SSW_OK;
}

bool valid_method(int m) { return m == SSW_OPTION1 || m == SSW_OPTION2 || m == SSW_OPTION3; }
bool valid_ordering(int o) { return o == SSW_ORDER_ENERGY || o == SSW_ORDER_ENERGY_ORTHOGONAL || o == SSW_ORDER_LEGACY; }
bool valid_precision(int p) { return p == SSW_PRECISION_F32 || p == SSW_PRECISION_F64; }

int check_config(const ssw_config* cfg) {
    if (!cfg) return SSW_ERR_BAD_ARG;
    if (cfg->method == SSW_METHOD_CUSTOM || cfg->ordering == SSW_ORDER_CUSTOM) return SSW_ERR_UNSUPPORTED;
    if (!valid_method(cfg->method) || !valid_ordering(cfg->ordering) || !valid_precision(cfg->precision))
        return SSW_ERR_BAD_ARG;
    return SSW_OK;
}

// Frames per internal pass: the caller's setting, or (0 = automatic, the default) about 2^30 pixels -- 128 4K
// frames, 514 full-HD ones, 32 8K ones -- capped where the f64 operand planes of a pass would pass 4 GB (the
// operand-ready GEMMs walk them with 32-bit offsets).  The GEMM grids then run ~64 rounds of blocks: against
// 2^28 pixels (16 rounds, the r1 default) the tails and first-tile latencies weigh 2.8 % less at 4K, 1.8 % at
// full HD, 1.4 % at 8K.  Workspace: up to 44 B/px of a pass per lane (r5: the fused forward transform keeps row and column operands side by side), sized for 288 GB of HBM -- and
// clamped to half of what the device can give right now (free memory + what the lanes already hold), so that
// a smaller device or a co-tenant (torch's caching allocator) gets smaller passes instead of an
// SSW_ERR_OUT_OF_MEMORY.
namespace {
size_t lane_bytes_held(const ssw_ctx* ctx) {
    size_t held = 0;
    for (const auto& ln : ctx->lane) {
        for (const auto& b : ln.plane) held += b.bytes;
        for (const auto& b : ln.operand) held += b.bytes;
        for (const auto& b : ln.compact) held += b.bytes;
        held += ln.idx.bytes + ln.gathered.bytes + ln.prune_u32.bytes;
    }
    return held;
}
}  // namespace

size_t effective_chunk(const ssw_ctx* ctx, size_t w, size_t h, size_t n_frames) {
    size_t c = ctx->chunk_frames;
    if (c == 0) {
        const size_t px = std::max<size_t>(w * h, 1);
        c = std::max<size_t>(1, ((size_t)1 << 30) / px);
        const size_t per_frame = dct_pair_operand_elems(true, 1, w, h) * sizeof(double);
        if (per_frame) c = std::max<size_t>(1, std::min(c, (size_t)0xFFFFFFFFull / per_frame));
        if (c >= 16) c &= ~(size_t)7;              // whole groups of 8 frames (4K: 129 -> 128: the GEMM line tiles stay whole)
        c = std::min(c, std::max<size_t>(n_frames, 1));
        size_t free_b = 0, total_b = 0;
        DeviceGuard g(ctx->device);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            // half of what the device can give: free memory, what the lanes already hold, and the spare handle planes
            // (dev_malloc gives those back before it fails).  A clamp can turn a one-pass call into several passes,
            // i.e. into two lanes: budget for one lane first, then for the lane count the clamped pass size implies.
            const size_t budget = (free_b + lane_bytes_held(ctx) + ctx->plane_pool_bytes) / 2;
            size_t c1 = std::max<size_t>(1, std::min(c, budget / (44 * px)));
            if (ctx->overlap && n_frames > c1) c1 = std::max<size_t>(1, std::min(c1, budget / (2 * 44 * px)));
            c = c1;
        } else {
            (void)hipGetLastError();
        }
    }
    return std::min(c, std::max<size_t>(n_frames, 1));
}

// ---- the 2-D transform as a chain ---------------------------------------------------------------------
namespace {

int pair_gemm(hipStream_t st, bool f64, bool is_row, bool inverse, int kind, int sub, const void* x1, const void* x2,
              const void* y1, const void* y2, float* dst, void* tmpE, size_t n, size_t w, size_t h, Epilogue ep,
              const RgbSink* sink = nullptr, void* tmp_out = nullptr, bool class_major = false) {
    return f64 ? launch_dct_pair_gemm_f64(st, is_row, inverse, kind, sub, (const double*)x1, (const double*)x2, (const double*)y1,
                                          (const double*)y2, dst, (double*)tmpE, n, w, h, ep, sink, (double*)tmp_out, class_major)
               : launch_dct_pair_gemm_f32(st, is_row, inverse, kind, sub, (const float*)x1, (const float*)x2, (const float*)y1,
                                          (const float*)y2, dst, (float*)tmpE, n, w, h, ep, sink);
}

// executed flop of one launch of the operand-ready GEMM (two products of lines x pairs x K multiply-adds)
double pair_gemm_flop(bool is_row, int kind, int sub, size_t n, size_t w, size_t h) {
    const double lines = (double)(is_row ? n * h : n * w);
    const size_t leff = (is_row ? w : h) >> sub;
    if (kind == 3 || kind == 4) return 4.0 * lines * (double)(leff / 8) * (double)(leff / 8);      // class E: n/8 + 1 pairs in n/8 slots
    if (kind >= 5) return 4.0 * lines * (double)(leff / 16) * (double)(leff / 16);                  // level 2: sums of len/16 terms, len/16 pairs
    const double np = (double)(kind == 0 ? leff / 2 : leff / 4), k = (double)(kind == 1 ? leff / 4 : leff / 2);
    return 4.0 * lines * np * k;
}

// doubles of the lane's sixth operand buffer: the split planes of either pass, or the deep row pre-pass's ten planes
size_t split_scratch_elems(size_t n, size_t w, size_t h) {
    size_t e = dct_pair_split_elems(n, w, h);
    if (dct_pair_can_deep_rows(w) || dct_pair_can_deep_inv_rows(w)) e = std::max(e, dct_pair_deep_elems(n * h, w));
    if (dct_pair_can_deep_cols(h)) e = std::max(e, dct_pair_deep_elems(n * w, h));
    if (dct_pair_can_fuse_cols(n, w, h) || dct_pair_can_fuse_inv_cols(n, w, h))
        e = std::max(e, dct_pair_deep_elems(n * 16 * dct_pair_fused_units(h), w));                                      // unit-ordered, padded lines
    if (dct_pair_can_semi_deep_cols(h)) e = std::max(e, dct_pair_semi_deep_elems(n * w, h));
    return e;
}

// Passes of at most this many lines run the classes of a stage as ONE launch (a single frame's launches are too small alone).
// SSW_MERGE_MAX_LINES: A/B switch.
static size_t merge_max_lines() {
    return (size_t)tuning(TUNE_MERGE_MAX_LINES);
}
// ... and batch passes too when `merge_batch` is on: their classes then run class after class inside the one launch (cls_major)
static bool merge_classes(size_t lines) {
    return lines <= merge_max_lines() || tuning(TUNE_MERGE_BATCH) != 0;
}

// Can the column pre-pass of an `fh`-row plane read the class-major order a deep row pass leaves (dct_pair_common.hpp)?  The
// deep kernels all can; of the semi-deep ones (fh % 16 == 8: 1080 rows) only the LDS-staged forms.
static bool cols_read_class_major(size_t fh, size_t w) {
    // (dct_pair_can_split: the column pass reaches its deep / semi-deep branch only through the split -- fh >= 128 whatever the thresholds say)
    return dct_pair_can_fold2_cols(fh) && dct_pair_can_split(fh, false) &&
           (dct_pair_can_deep_cols(fh) || (dct_pair_can_semi_deep_cols(fh) && dct_pair_prep_staged_cols_ok(w, true)));
}

// One pass of the separable transform (src -> dst along rows or columns) appended to `ch`.
int build_pass_impl(ssw_ctx* ctx, ssw_ctx::Lane& ws, const Xform& x, bool first_pass, bool is_row, const float* src, float* dst,
                    Epilogue ep, Chain& ch, bool* fused_rgb);
int build_pass(ssw_ctx* ctx, ssw_ctx::Lane& ws, const Xform& x, bool first_pass, bool is_row, const float* src, float* dst,
               Epilogue ep, Chain& ch, bool* fused_rgb = nullptr) {
    const size_t first = ch.size();
    SSW_TRY(build_pass_impl(ctx, ws, x, first_pass, is_row, src, dst, ep, ch, fused_rgb));
    for (size_t i = first; i < ch.size(); ++i) {                  // hints for the two-lane scheduler (run_pipeline_impl)
        if (!ch[i].hbm && is_row) ch[i].tag = 1;
        if (ch[i].hbm && is_row && first_pass && x.type != SSW_DCT3 && x.rgb) ch[i].tag = 2;
    }
    return SSW_OK;
}
int build_pass_impl(ssw_ctx* ctx, ssw_ctx::Lane& ws, const Xform& x, bool first_pass, bool is_row, const float* src, float* dst,
                    Epilogue ep, Chain& ch, bool* fused_rgb) {
    const bool inverse = (x.type == SSW_DCT3);
    const bool f64 = (x.precision == SSW_PRECISION_F64);
    const int precision = x.precision;
    const size_t n = x.n, w = x.w, h = x.h;
    const size_t len = is_row ? w : h;
    const double px = (double)n * (double)w * (double)h;
    const double esz = f64 ? 8.0 : 4.0;
    const bool can_fold = ctx->fold && (is_row ? dct_rows_can_fold(w, src, dst) : dct_cols_can_fold(w, h, src, dst));
    const bool operand = can_fold && ctx->fold_level >= 3 && dct_pair_can_run(f64, n, w, h, src, dst);
    // (default build: the in-kernel folding of dct_folded*.hip is not compiled in -- what the pair path does not take runs dense)
    const bool fold = can_fold && (operand || build_all_strategies());
    const bool from_rgb = x.rgb && first_pass;
    if (from_rgb && !(operand && is_row && ctx->fold_level >= 4 && dct_pair_can_fold2(len)))
        return SSW_ERR_BAD_ARG;                                    // can_fuse_rgb() checks the same conditions
    const int st_pass = is_row ? SSW_STAGE_DCT_ROW : SSW_STAGE_DCT_COL;
    const int st_main = is_row ? SSW_STAGE_DCT_ROW_MAIN : SSW_STAGE_DCT_COL_MAIN;
    const void* rgb = x.rgb;
    const int rgb_u8 = x.rgb_u8;
    float *iq_i = x.iq_i, *iq_q = x.iq_q;
    // algorithmic bytes of the pre-pass: the f32 plane (or the RGB frame) in, the operand planes (one
    // element per pixel in the GEMM's precision, whatever the number of folding levels) and I / Q out
    const double prep_bytes = from_rgb ? px * (3.0 * (double)pix_bytes(rgb_u8) + (iq_i ? 8.0 : 0.0) + esz) : px * (4.0 + esz);
    const int st_prep = from_rgb ? SSW_STAGE_RGB_TO_YIQ : SSW_STAGE_DCT_PREP;
    // algorithmic bytes of the pass's GEMM launches (ssw_ctx_get_traffic): the operand planes in (one element per pixel), the
    // f32 result out -- or, in the last pass of Writer::result, I and Q in and the RGB frame out -- plus `xch` bytes per
    // pixel that the dependent launches of an inverse pass write and read back as doubles (A1: 1 + 1, T2: 2 + 2, E: 4 + 4)
    const bool sink_pass = inverse && !first_pass && !is_row && x.rgb_out && x.iq_i && x.iq_q;
    const double out_bpp = sink_pass ? 8.0 + 3.0 * (double)pix_bytes(x.rgb_out_u8) : 4.0;
    auto gemm_bytes = [=](double xch) { return px * (esz + xch + out_bpp); };
    if (operand) {
        const size_t bytes = dct_pair_operand_elems(f64, n, w, h) * (size_t)esz;
        const bool two = ctx->fold_level >= 4 && (is_row ? dct_pair_can_fold2(len) : dct_pair_can_fold2_cols(len));
        const void *b0 = nullptr, *b1 = nullptr;
        SSW_TRY(get_basis(ctx, len, inverse, f64, 3, &b0));        // k-blocked half bases
        SSW_TRY(get_basis(ctx, len, inverse, f64, 4, &b1));
        // a third level pays once the sums are long enough (4K: +1.6 %, 1080p: -3 %); level 6 forces it
        const bool three = two && !inverse && dct_pair_can_fold3(len) &&
                           (ctx->fold_level >= 6 || (ctx->fold_level == 5 && len >= 3072));
        // the odd half as a rotated pair of quarter-length cosine / sine transforms (f64): a quarter of its multiply-adds
        const bool split = two && f64 && ctx->split && dct_pair_can_split(len, is_row);
        const void *sb[4] = {nullptr, nullptr, nullptr, nullptr}, *rot = nullptr;
        double* sp = nullptr;
        const size_t lines = is_row ? n * h : n * w;
        const size_t sp_plane = lines * (split ? dct_pair_split_kpad(len) : 0);
        if (split) {
            for (int b = 0; b < 4; ++b) SSW_TRY(get_basis(ctx, len, inverse, true, b == 1 ? 10 : 5 + b, &sb[b]));      // 10: sinE for launches
            SSW_TRY(get_basis(ctx, len, false, true, 9, &rot));
            SSW_TRY(grow(ws.operand[5], split_scratch_elems(n, w, h) * sizeof(double)));
            sp = (double*)ws.operand[5].p;
        }
        // forward passes of 64-divisible (rows) / 16-divisible (columns) length: one pre-pass writes the operands of five
        // launches (D and SD split, SS folded a third time)
        const bool deep = split && !inverse && (is_row ? dct_pair_can_deep_rows(len) : dct_pair_can_deep_cols(len) && w % 4 == 0);
        if (deep) {
            const void *e0 = nullptr, *e1 = nullptr, *sb2[4], *rot2 = nullptr, *rot3 = nullptr, *h0 = nullptr, *h1 = nullptr;
            SSW_TRY(get_basis(ctx, len / 4, false, true, 3, &e0));
            SSW_TRY(get_basis(ctx, len / 4, false, true, 4, &e1));
            for (int b = 0; b < 4; ++b) SSW_TRY(get_basis(ctx, len / 2, false, true, b == 1 ? 10 : 5 + b, &sb2[b]));
            SSW_TRY(get_basis(ctx, len / 2, false, true, 9, &rot2));
            // Row passes of 1280 columns or more run at LEVEL 2 (r4b, dct_pair_efold): every operand of the full-length split
            // and of SS folds or rotates once more in the pre-pass (dct_pair_split.hpp, DeepPlanes), so that all eight
            // launches are sums of len/16 terms over len/16 pairs -- 2/3 of the level-1 pass's multiply-adds:
            //   class E (DCT-II of AS, DST-II of BD) folds exactly        -> kinds 5 / 6   16i +/- 1,  16i + 9 | 16i + 7
            //   class O (DCT-IV of AD, DST-IV of BS) rotates              -> kinds 7 / 8   16i +/- 5,  16i +/- 3
            //   R2 (DCT-IV) rotates, R1 (DCT-II) folds exactly            -> kind 9, kind 1 sub 2      16i +/- 4,  16i | 16i + 8
            // Shorter rows stay at level 1.  The "main" timer brackets ONE launch: kind 7 (level 1: class O of the full-length split).
            // Column passes of 720 rows or more do the same (r4c, dct_pair_efold_cols; the pre-pass holds a unit and its
            // mirror in one thread): K = H/16 = 135 at 4K -- such launches reach 50 TFLOP/s against 64 for K = 270, at half
            // the multiply-adds.
            const size_t fh0 = x.full_h ? x.full_h : h;
            const bool cm0 = !x.natural_order && w >= fh0 && w % 4 == 0 && dct_pair_can_deep_rows(w) && cols_read_class_major(fh0, w) &&
                             (is_row ? first_pass : !first_pass);
            const bool l2 = is_row ? dct_pair_efold(len) : dct_pair_efold_cols(len, w, cm0);
            if (l2) {
                SSW_TRY(get_basis(ctx, len / 4, false, true, 9, &rot3));
                SSW_TRY(get_basis(ctx, len / 8, false, true, 3, &h0));
                SSW_TRY(get_basis(ctx, len / 8, false, true, 4, &h1));
            }
            const void *sb0 = sb[0], *sb1 = sb[1], *sb2_ = sb[2], *sb3 = sb[3];
            const void *t0 = sb2[0], *t1 = sb2[1], *t2 = sb2[2], *t3 = sb2[3];
            // rows first and both passes deep: the row launches write class-major, the column pre-pass reads it back
            const size_t fh = x.full_h ? x.full_h : h;
            const bool cm = cm0;
            (void)fh;
            // r5: FUSED forward transform (dct_pair_can_fuse_cols: rows first, both passes at level 2, batches on 128-line
            // tiles).  The row pre-pass orders its lines (frame, unit of the column fold, line of the unit), the row launches'
            // epilogue (EPI_FWD_COLOP) rounds to f32 -- the store between the passes, src/dct2d.rs:152-168 -- applies the column
            // pre-pass's arithmetic to its accumulators and writes the sixteen column-operand planes; the column pass is its
            // eight launches only.  The f32 plane between the passes and the column pre-pass are gone: 4 + 4 + 8 B/px become 8.
            const bool fuse = l2 && cm && !x.full_h && dct_pair_can_fuse_cols(n, w, h);
            if (fuse) {
                const size_t k16h = dct_pair_split_kpad(h / 2), cplane = n * w * k16h;             // column operands: n w lines, K16(h) wide
                SSW_TRY(grow(ws.operand[0], 16 * cplane * sizeof(double)));
                double* cop = (double*)ws.operand[0].p;
                auto CP = [=](int j) { return (const double*)(cop + (size_t)j * cplane); };
                const double f_main = pair_gemm_flop(is_row, 7, 0, n, w, h), f_all = 8.0 * f_main;
                if (is_row) {
                    const void *crot1 = nullptr, *crot2 = nullptr, *crot3 = nullptr;
                    SSW_TRY(get_basis(ctx, h, false, true, 9, &crot1));
                    SSW_TRY(get_basis(ctx, h / 2, false, true, 9, &crot2));
                    SSW_TRY(get_basis(ctx, h / 4, false, true, 9, &crot3));
                    const size_t lpad = n * 16 * dct_pair_fused_units(h), p16r = lpad * dct_pair_split_kpad(len / 2);
                    auto P = [=](int j) { return (const double*)(sp + (size_t)j * p16r); };
                    const double pad = (double)lpad / (double)(n * h);                              // the padding units' share of the flop
                    ch.push_back({true, [=](hipStream_t st) -> int {
                        StageTimer t(ctx, st_prep, st, prep_bytes);
                        return launch_dct_pair_prep16_rows(st, from_rgb ? pix_src_kind(rgb_u8) : 0, from_rgb ? rgb : (const void*)src, n, w, h, sp,
                                                           (const double*)rot, (const double*)rot2, (const double*)rot3,
                                                           from_rgb ? iq_i : nullptr, from_rgb ? iq_q : nullptr, true);
                    }});
                    const PairClassDesc d[8] = {{1, 2, P(8), P(9), (const double*)h0, (const double*)h1},
                                                {9, 0, P(10), P(11), (const double*)t0, (const double*)t1},
                                                {3, 1, P(12), P(13), (const double*)t0, (const double*)t1},
                                                {4, 1, P(14), P(15), (const double*)t2, (const double*)t3},
                                                {5, 0, P(0), P(3), (const double*)t0, (const double*)t1},
                                                {6, 0, P(1), P(2), (const double*)t2, (const double*)t3},
                                                {8, 0, P(6), P(7), (const double*)t0, (const double*)t1},
                                                {7, 0, P(4), P(5), (const double*)t0, (const double*)t1}};
                    const FuseCols fc{FUSE_ROWS_COP, cop, (const double*)crot1, (const double*)crot2, (const double*)crot3};
                    const bool merge_r = merge_classes(lpad);       // a single frame: the eight classes in one launch
                    ch.push_back({false, [=](hipStream_t st) -> int {
                        StageTimer t(ctx, st_pass, st, f_all * pad, merge_r ? st_main : -1);
                        t.traffic(px * (esz + 8.0));                     // row operands in, column operands out
                        if (merge_r) {
                            return launch_dct_pair_gemm_multi_f64(st, true, false, 8, d, nullptr, nullptr, n, w, h, ep, nullptr, nullptr, true, &fc);
                        }
                        for (int c = 0; c < 7; ++c)
                            SSW_TRY(launch_dct_pair_gemm_multi_f64(st, true, false, 1, &d[c], nullptr, nullptr, n, w, h, ep, nullptr, nullptr, true, &fc));
                        StageTimer tm(ctx, st_main, st, f_main * pad);
                        return launch_dct_pair_gemm_multi_f64(st, true, false, 1, &d[7], nullptr, nullptr, n, w, h, ep, nullptr, nullptr, true, &fc);
                    }});
                    return SSW_OK;
                }
                const PairClassDesc d[8] = {{1, 2, CP(8), CP(9), (const double*)h0, (const double*)h1},
                                            {9, 0, CP(10), CP(11), (const double*)t0, (const double*)t1},
                                            {3, 1, CP(12), CP(13), (const double*)t0, (const double*)t1},
                                            {4, 1, CP(14), CP(15), (const double*)t2, (const double*)t3},
                                            {5, 0, CP(0), CP(3), (const double*)t0, (const double*)t1},
                                            {6, 0, CP(1), CP(2), (const double*)t2, (const double*)t3},
                                            {8, 0, CP(6), CP(7), (const double*)t0, (const double*)t1},
                                            {7, 0, CP(4), CP(5), (const double*)t0, (const double*)t1}};
                const FuseCols fc{FUSE_COLS};
                const bool merge_c = merge_classes(lines);
                ch.push_back({false, [=](hipStream_t st) -> int {
                    StageTimer t(ctx, st_pass, st, f_all, merge_c ? st_main : -1);
                    t.traffic(gemm_bytes(0.0));
                    if (merge_c) {
                        return launch_dct_pair_gemm_multi_f64(st, false, false, 8, d, dst, nullptr, n, w, h, ep, nullptr, nullptr, false, &fc);
                    }
                    for (int c = 0; c < 7; ++c)
                        SSW_TRY(launch_dct_pair_gemm_multi_f64(st, false, false, 1, &d[c], dst, nullptr, n, w, h, ep, nullptr, nullptr, false, &fc));
                    StageTimer tm(ctx, st_main, st, f_main);
                    return launch_dct_pair_gemm_multi_f64(st, false, false, 1, &d[7], dst, nullptr, n, w, h, ep, nullptr, nullptr, false, &fc);
                }});
                return SSW_OK;
            }
            const size_t p8 = lines * dct_pair_split_kpad(len), p16 = lines * dct_pair_split_kpad(len / 2);
            double* q = sp + 6 * p8;
            ch.push_back({true, [=](hipStream_t st) -> int {
                StageTimer t(ctx, st_prep, st, prep_bytes);
                if (!is_row) return launch_dct_pair_prep16_cols(st, src, n, w, h, sp, (const double*)rot, (const double*)rot2, cm, (const double*)rot3);
                return launch_dct_pair_prep16_rows(st, from_rgb ? pix_src_kind(rgb_u8) : 0, from_rgb ? rgb : (const void*)src, n, w, h, sp,
                                                   (const double*)rot, (const double*)rot2, (const double*)rot3,
                                                   from_rgb ? iq_i : nullptr, from_rgb ? iq_q : nullptr);
            }});
            // a single frame's launches are too small alone (class E of a 4K frame: 272 blocks for 512 slots): one launch
            // over all classes instead
            const bool merge = lines <= merge_max_lines();             // (merging the batch launches as well: measured, no difference)
            if (l2) {
                // the sixteen planes of launch_dct_pair_prep16_rows, K16 wide each, by number
                auto P = [=](int j) { return (const double*)(sp + (size_t)j * p16); };
                const double f_main = pair_gemm_flop(is_row, 7, 0, n, w, h), f_all = 8.0 * f_main;
                const PairClassDesc d[8] = {{1, 2, P(8), P(9), (const double*)h0, (const double*)h1},          // R1+ R1-
                                            {9, 0, P(10), P(11), (const double*)t0, (const double*)t1},        // R2 rotated
                                            {3, 1, P(12), P(13), (const double*)t0, (const double*)t1},        // AS2 BD2
                                            {4, 1, P(14), P(15), (const double*)t2, (const double*)t3},        // AD2 BS2
                                            {5, 0, P(0), P(3), (const double*)t0, (const double*)t1},          // AS+ BD-
                                            {6, 0, P(1), P(2), (const double*)t2, (const double*)t3},          // AS- BD+
                                            {8, 0, P(6), P(7), (const double*)t0, (const double*)t1},          // O rotated, "-"
                                            {7, 0, P(4), P(5), (const double*)t0, (const double*)t1}};         // O rotated, "+"
                ch.push_back({false, [=](hipStream_t st) -> int {
                    StageTimer t(ctx, st_pass, st, f_all, merge ? st_main : -1);
                    t.traffic(gemm_bytes(0.0));
                    const bool rcm = cm && is_row;
                    if (merge) {
                        return launch_dct_pair_gemm_multi_f64(st, is_row, false, 8, d, dst, nullptr, n, w, h, ep, nullptr, nullptr, rcm);
                    }
                    for (int c = 0; c < 7; ++c)
                        SSW_TRY(pair_gemm(st, true, is_row, false, d[c].kind, d[c].sub, d[c].x1, d[c].x2, d[c].y1, d[c].y2, dst, nullptr, n, w, h, ep,
                                          nullptr, nullptr, rcm));
                    StageTimer tm(ctx, st_main, st, f_main);
                    return pair_gemm(st, true, is_row, false, 7, 0, d[7].x1, d[7].x2, d[7].y1, d[7].y2, dst, nullptr, n, w, h, ep, nullptr, nullptr, rcm);
                }});
                return SSW_OK;
            }
            const double f_main = pair_gemm_flop(is_row, 4, 0, n, w, h);
            const double f_all = f_main + pair_gemm_flop(is_row, 3, 0, n, w, h) + pair_gemm_flop(is_row, 1, 1, n, w, h) +
                                 pair_gemm_flop(is_row, 3, 1, n, w, h) + pair_gemm_flop(is_row, 4, 1, n, w, h);
            ch.push_back({false, [=](hipStream_t st) -> int {
                StageTimer t(ctx, st_pass, st, f_all);
                t.traffic(gemm_bytes(0.0));
                const bool rcm = cm && is_row;
                if (merge) {
                    const PairClassDesc d[5] = {{1, 1, sp + 4 * p8, sp + 5 * p8, (const double*)e0, (const double*)e1},
                                                {3, 1, q, q + p16, (const double*)t0, (const double*)t1},
                                                {4, 1, q + 2 * p16, q + 3 * p16, (const double*)t2, (const double*)t3},
                                                {4, 0, sp + 2 * p8, sp + 3 * p8, (const double*)sb2_, (const double*)sb3},
                                                {3, 0, sp, sp + p8, (const double*)sb0, (const double*)sb1}};
                    StageTimer tm(ctx, st_main, st, f_all);
                    return launch_dct_pair_gemm_multi_f64(st, is_row, false, 5, d, dst, nullptr, n, w, h, ep, nullptr, nullptr, rcm);
                }
                SSW_TRY(pair_gemm(st, true, is_row, false, 1, 1, sp + 4 * p8, sp + 5 * p8, e0, e1, dst, nullptr, n, w, h, ep, nullptr, nullptr, rcm));
                SSW_TRY(pair_gemm(st, true, is_row, false, 3, 1, q, q + p16, t0, t1, dst, nullptr, n, w, h, ep, nullptr, nullptr, rcm));
                SSW_TRY(pair_gemm(st, true, is_row, false, 4, 1, q + 2 * p16, q + 3 * p16, t2, t3, dst, nullptr, n, w, h, ep, nullptr, nullptr, rcm));
                SSW_TRY(pair_gemm(st, true, is_row, false, 3, 0, sp, sp + p8, sb0, sb1, dst, nullptr, n, w, h, ep, nullptr, nullptr, rcm));
                StageTimer tm(ctx, st_main, st, f_main);
                return pair_gemm(st, true, is_row, false, 4, 0, sp + 2 * p8, sp + 3 * p8, sb2_, sb3, dst, nullptr, n, w, h, ep, nullptr, nullptr, rcm);
            }});
            return SSW_OK;
        }
        // the odd half of the full-length transform from the operand plane `odd`: one launch, or rotate + two
        auto odd_rotate = [=](hipStream_t st, const void* odd) -> int {
            return split ? launch_dct_pair_rotate(st, (const double*)odd, (const double*)rot, sp, lines, len) : SSW_OK;
        };
        auto odd_gemm = [=](hipStream_t st, const void* odd, const void* basis, void* tmpE, const RgbSink* sink) -> int {
            if (!split) return pair_gemm(st, f64, is_row, inverse, 2, 0, odd, odd, basis, (const char*)basis + (len / 4) * 64, dst, tmpE, n, w, h, ep, sink);
            SSW_TRY(pair_gemm(st, true, is_row, inverse, 3, 0, sp, sp + sp_plane, sb[0], sb[1], dst, tmpE, n, w, h, ep, sink));
            return pair_gemm(st, true, is_row, inverse, 4, 0, sp + 2 * sp_plane, sp + 3 * sp_plane, sb[2], sb[3], dst, tmpE, n, w, h, ep, sink);
        };
        const double f_odd = split ? pair_gemm_flop(is_row, 3, 0, n, w, h) + pair_gemm_flop(is_row, 4, 0, n, w, h)
                                   : pair_gemm_flop(is_row, 2, 0, n, w, h);
        // forward column passes of 8- but not 16-divisible length (1080 rows): D split and SS folded a third time in one
        // pre-pass, SD stays one launch (H/16 is not whole)
        const bool semi = split && !inverse && !is_row && dct_pair_can_semi_deep_cols(len) && w % 4 == 0;
        if (semi) {
            const void *e0 = nullptr, *e1 = nullptr, *h1 = nullptr;
            SSW_TRY(get_basis(ctx, len / 4, false, true, 3, &e0));
            SSW_TRY(get_basis(ctx, len / 4, false, true, 4, &e1));
            SSW_TRY(get_basis(ctx, len / 2, false, true, 4, &h1));
            const size_t p8 = lines * dct_pair_split_kpad(len);
            double* m = sp + 6 * p8;
            const void *sb0 = sb[0], *sb1 = sb[1], *sb2_ = sb[2], *sb3 = sb[3];
            // behind a deep row pass the plane arrives class-major (r4c: the staged pre-pass reads it like the deep one)
            const size_t fh = x.full_h ? x.full_h : h;
            const bool cm = !x.natural_order && w >= fh && dct_pair_can_deep_rows(w) && cols_read_class_major(fh, w) && !first_pass;
            ch.push_back({true, [=](hipStream_t st) -> int {
                StageTimer t(ctx, st_prep, st, prep_bytes);
                return launch_dct_pair_prep16_cols(st, src, n, w, h, sp, (const double*)rot, (const double*)rot, cm);
            }});
            const double f_main = pair_gemm_flop(is_row, 3, 0, n, w, h);
            const double f_all = f_main + pair_gemm_flop(is_row, 4, 0, n, w, h) + pair_gemm_flop(is_row, 1, 1, n, w, h) + pair_gemm_flop(is_row, 2, 1, n, w, h);
            const bool merge = lines <= merge_max_lines();
            ch.push_back({false, [=](hipStream_t st) -> int {
                StageTimer t(ctx, st_pass, st, f_all);
                t.traffic(gemm_bytes(0.0));
                if (merge) {      // the SD launch shares its image operand between its two products: another template instance
                    SSW_TRY(pair_gemm(st, true, is_row, false, 2, 1, m, m, h1, (const char*)h1 + (len / 8) * 64, dst, nullptr, n, w, h, ep));
                    const PairClassDesc d[3] = {{1, 1, sp + 4 * p8, sp + 5 * p8, (const double*)e0, (const double*)e1},
                                                {4, 0, sp + 2 * p8, sp + 3 * p8, (const double*)sb2_, (const double*)sb3},
                                                {3, 0, sp, sp + p8, (const double*)sb0, (const double*)sb1}};
                    StageTimer tm(ctx, st_main, st, f_all - pair_gemm_flop(is_row, 2, 1, n, w, h));
                    return launch_dct_pair_gemm_multi_f64(st, is_row, false, 3, d, dst, nullptr, n, w, h, ep);
                }
                SSW_TRY(pair_gemm(st, true, is_row, false, 1, 1, sp + 4 * p8, sp + 5 * p8, e0, e1, dst, nullptr, n, w, h, ep));
                SSW_TRY(pair_gemm(st, true, is_row, false, 2, 1, m, m, h1, (const char*)h1 + (len / 8) * 64, dst, nullptr, n, w, h, ep));
                SSW_TRY(pair_gemm(st, true, is_row, false, 4, 0, sp + 2 * p8, sp + 3 * p8, sb2_, sb3, dst, nullptr, n, w, h, ep));
                StageTimer tm(ctx, st_main, st, f_main);
                return pair_gemm(st, true, is_row, false, 3, 0, sp, sp + p8, sb0, sb1, dst, nullptr, n, w, h, ep);
            }});
            return SSW_OK;
        }
        // ... and the inverse column pass of such a length: c[8q] / c[8q+4] -> T2, the whole c[4q+2] part + T2 -> E, split odd
        // part + E -> output
        const bool semi_inv = split && inverse && !is_row && dct_pair_can_semi_deep_cols(len) && w % 4 == 0;
        if (semi_inv) {
            const void *e0 = nullptr, *e1 = nullptr, *h1 = nullptr;
            SSW_TRY(get_basis(ctx, len / 4, true, true, 3, &e0));
            SSW_TRY(get_basis(ctx, len / 4, true, true, 4, &e1));
            SSW_TRY(get_basis(ctx, len / 2, true, true, 4, &h1));
            SSW_TRY(grow(ws.operand[1], bytes));
            SSW_TRY(grow(ws.operand[4], bytes));
            void* T2 = ws.operand[1].p;
            void* TE = ws.operand[4].p;
            const size_t p8 = lines * dct_pair_split_kpad(len);
            double* m = sp + 6 * p8;
            const void *sb0 = sb[0], *sb1 = sb[1], *sb2_ = sb[2], *sb3 = sb[3];
            const bool cm = !x.natural_order && w >= h && dct_pair_can_deep_inv_rows(w) && cols_read_class_major(h, w) && !first_pass;
            ch.push_back({true, [=](hipStream_t st) -> int {
                StageTimer t(ctx, st_prep, st, prep_bytes);
                return launch_dct_pair_prep16_inv_cols(st, src, n, w, h, sp, (const double*)rot, (const double*)rot, cm);
            }});
            RgbSink sink;
            if (!first_pass && x.rgb_out && x.iq_i && x.iq_q) {
                sink.iq_i = x.iq_i; sink.iq_q = x.iq_q; sink.rgb = x.rgb_out; sink.u8 = x.rgb_out_u8;
                if (fused_rgb) *fused_rgb = true;
            }
            const bool with_sink = sink.rgb != nullptr;
            const double f_all = pair_gemm_flop(is_row, 3, 0, n, w, h) + pair_gemm_flop(is_row, 4, 0, n, w, h) + pair_gemm_flop(is_row, 1, 1, n, w, h) +
                                 pair_gemm_flop(is_row, 2, 1, n, w, h);
            ch.push_back({false, [=](hipStream_t st) -> int {
                StageTimer t(ctx, st_pass, st, f_all);
                t.traffic(gemm_bytes(12.0));
                SSW_TRY(pair_gemm(st, true, is_row, true, 1, 1, sp + 4 * p8, sp + 5 * p8, e0, e1, dst, T2, n, w, h, ep));
                SSW_TRY(pair_gemm(st, true, is_row, true, 2, 1, m, m, h1, (const char*)h1 + (len / 8) * 64, dst, T2, n, w, h, ep, nullptr, TE));
                if (lines <= merge_max_lines()) {
                    const PairClassDesc d0[2] = {{3, 0, sp, sp + p8, (const double*)sb0, (const double*)sb1},
                                                 {4, 0, sp + 2 * p8, sp + 3 * p8, (const double*)sb2_, (const double*)sb3}};
                    return launch_dct_pair_gemm_multi_f64(st, is_row, true, 2, d0, dst, (double*)TE, n, w, h, ep, with_sink ? &sink : nullptr);
                }
                SSW_TRY(pair_gemm(st, true, is_row, true, 3, 0, sp, sp + p8, sb0, sb1, dst, TE, n, w, h, ep, with_sink ? &sink : nullptr));
                return pair_gemm(st, true, is_row, true, 4, 0, sp + 2 * p8, sp + 3 * p8, sb2_, sb3, dst, TE, n, w, h, ep, with_sink ? &sink : nullptr);
            }});
            return SSW_OK;
        }
        // the inverse the same way: c[8q] / c[8q+4] -> T2, the split c[4q+2] part + T2 -> T (the even half E), then the
        // split odd part + T -> the output; one pre-pass for all five launches
        const bool deep_inv = split && inverse && (is_row ? dct_pair_can_deep_inv_rows(len) : dct_pair_can_deep_cols(len) && w % 4 == 0);
        if (deep_inv) {
            const void *e0 = nullptr, *e1 = nullptr, *sb2[4], *rot2 = nullptr, *rot3 = nullptr, *h0 = nullptr, *h1 = nullptr;
            SSW_TRY(get_basis(ctx, len / 4, true, true, 3, &e0));
            SSW_TRY(get_basis(ctx, len / 4, true, true, 4, &e1));
            for (int b = 0; b < 4; ++b) SSW_TRY(get_basis(ctx, len / 2, true, true, b == 1 ? 10 : 5 + b, &sb2[b]));
            SSW_TRY(get_basis(ctx, len / 2, false, true, 9, &rot2));
            // (fused inverse transform, r5: the row pass runs over unit-ordered lines padded to whole k-blocks of units)
            const size_t xlines = (is_row && dct_pair_can_fuse_inv_cols(n, w, h)) ? n * 16 * dct_pair_fused_units(h) : lines;
            SSW_TRY(grow(ws.operand[1], std::max<size_t>(bytes, xlines * (len / 4) * sizeof(double))));
            SSW_TRY(grow(ws.operand[4], std::max<size_t>(bytes, xlines * (len / 2) * sizeof(double))));
            // Row passes of 1280 columns or more (a multiple of 256) run at LEVEL 2 (r4c, dct_pair_efold_inv), the transpose of
            // the forward pass's: every launch sums len/16 coefficients --
            //   the quarter-length even part T2 = (its even half A1: kind 1 sub 2, folded) +/- (its odd half: R2 rotated, kind 9)
            //   the half-length odd part (kinds 3 / 4 sub 1) as at level 1:  E = T2 +/- .
            //   the odd part: class E folded (kinds 5 / 6), class O rotated (kinds 7 / 8):  x = E +/- .
            // 8/14 of the level-1 pass's multiply-adds.
            // Column passes of 720 rows or more likewise (dct_pair_efold_cols).
            const bool cm0 = !x.natural_order && w >= h && w % 4 == 0 && dct_pair_can_deep_inv_rows(w) && cols_read_class_major(h, w) &&
                             (is_row ? first_pass : !first_pass);
            const bool il2 = is_row ? dct_pair_efold_inv(len) : dct_pair_efold_cols(len, w, cm0);
            void* A1 = nullptr;
            if (il2) {
                SSW_TRY(get_basis(ctx, len / 4, false, true, 9, &rot3));
                SSW_TRY(get_basis(ctx, len / 8, true, true, 3, &h0));
                SSW_TRY(get_basis(ctx, len / 8, true, true, 4, &h1));
                SSW_TRY(grow(ws.operand[2], std::max<size_t>(bytes, xlines * (len / 8) * sizeof(double))));      // (>= what any other pass asks of it)
                A1 = ws.operand[2].p;             // the eighth-length even part, unrounded: len/8 doubles per line
            }
            void* T2 = ws.operand[1].p;       // quarter-length even half, unrounded
            void* TE = ws.operand[4].p;       // the even half E, unrounded
            const size_t p8 = lines * dct_pair_split_kpad(len), p16 = lines * dct_pair_split_kpad(len / 2);
            double* q = sp + 6 * p8;
            const void *sb0 = sb[0], *sb1 = sb[1], *sb2_ = sb[2], *sb3 = sb[3];
            const void *t0 = sb2[0], *t1 = sb2[1], *t2 = sb2[2], *t3 = sb2[3];
            // rows first and both passes deep: the row pass's split launches write (and exchange E) class-major
            const bool cm = cm0;
            const bool rcm = cm && is_row;
            RgbSink sink;
            if (!first_pass && !is_row && x.rgb_out && x.iq_i && x.iq_q) {
                sink.iq_i = x.iq_i; sink.iq_q = x.iq_q; sink.rgb = x.rgb_out; sink.u8 = x.rgb_out_u8;
                if (fused_rgb) *fused_rgb = true;
            }
            const bool with_sink = sink.rgb != nullptr;
            // r5: FUSED inverse transform (dct_pair_can_fuse_inv_cols), the mirror image of the forward one: the row pre-pass
            // orders its lines by unit of the inverse column fold, the four launches of the odd part (EPI_INV_O_COLOP) round
            // their results to f32 -- the store between the passes -- and write the column operands; the column pass is its
            // eight launches.  A1 / T2 / E are exchanged per line as before (their planes are indexed by operand line).
            const bool fuse_inv = il2 && cm && !x.full_h && dct_pair_can_fuse_inv_cols(n, w, h);
            if (fuse_inv) {
                const size_t k16h = dct_pair_split_kpad(h / 2), cplane = n * w * k16h;
                SSW_TRY(grow(ws.operand[0], 16 * cplane * sizeof(double)));
                double* cop = (double*)ws.operand[0].p;
                const double f_all = 8.0 * pair_gemm_flop(is_row, 7, 0, n, w, h);
                if (is_row) {
                    const void *crot1 = nullptr, *crot2 = nullptr, *crot3 = nullptr;
                    SSW_TRY(get_basis(ctx, h, false, true, 9, &crot1));
                    SSW_TRY(get_basis(ctx, h / 2, false, true, 9, &crot2));
                    SSW_TRY(get_basis(ctx, h / 4, false, true, 9, &crot3));
                    const size_t lpad = n * 16 * dct_pair_fused_units(h), p16r = lpad * dct_pair_split_kpad(len / 2);
                    double *A1p = (double*)A1, *T2p = (double*)T2, *TEp = (double*)TE;      // sized for the padded lines above
                    auto P = [=](int j) { return (const double*)(sp + (size_t)j * p16r); };
                    const double pad = (double)lpad / (double)(n * h);
                    ch.push_back({true, [=](hipStream_t st) -> int {
                        StageTimer t(ctx, st_prep, st, prep_bytes);
                        return launch_dct_pair_prep16_inv_rows(st, src, n, w, h, sp, (const double*)rot, (const double*)rot2, (const double*)rot3, true);
                    }});
                    const PairClassDesc da = {1, 2, P(8), P(9), (const double*)h0, (const double*)h1};
                    const PairClassDesc db = {9, 2, P(10), P(11), (const double*)t0, (const double*)t1};
                    const PairClassDesc d1[2] = {{3, 1, P(12), P(13), (const double*)t0, (const double*)t1},
                                                 {4, 1, P(14), P(15), (const double*)t2, (const double*)t3}};
                    const PairClassDesc d0[4] = {{5, 0, P(0), P(3), (const double*)t0, (const double*)t1},
                                                 {6, 0, P(1), P(2), (const double*)t2, (const double*)t3},
                                                 {7, 0, P(4), P(5), (const double*)t0, (const double*)t1},
                                                 {8, 0, P(6), P(7), (const double*)t0, (const double*)t1}};
                    const FuseCols fl{FUSE_ROWS_LINES};
                    const FuseCols fc{FUSE_ROWS_COP, cop, (const double*)crot1, (const double*)crot2, (const double*)crot3};
                    ch.push_back({false, [=](hipStream_t st) -> int {
                        StageTimer t(ctx, st_pass, st, f_all * pad);
                        t.traffic(px * (esz + 14.0 + 8.0));              // operands in, A1 / T2 / E out and in, column operands out
                        SSW_TRY(launch_dct_pair_gemm_multi_f64(st, true, true, 1, &da, dst, A1p, n, w, h, ep, nullptr, nullptr, false, &fl));
                        SSW_TRY(launch_dct_pair_gemm_multi_f64(st, true, true, 1, &db, dst, A1p, n, w, h, ep, nullptr, T2p, true, &fl));
                        for (int c = 0; c < 2; ++c)
                            SSW_TRY(launch_dct_pair_gemm_multi_f64(st, true, true, 1, &d1[c], dst, T2p, n, w, h, ep, nullptr, TEp, true, &fl));
                        for (int c = 0; c < 4; ++c)
                            SSW_TRY(launch_dct_pair_gemm_multi_f64(st, true, true, 1, &d0[c], nullptr, TEp, n, w, h, ep, nullptr, nullptr, true, &fc));
                        return SSW_OK;
                    }});
                    return SSW_OK;
                }
                auto CP = [=](int j) { return (const double*)(cop + (size_t)j * cplane); };
                const PairClassDesc da = {1, 2, CP(8), CP(9), (const double*)h0, (const double*)h1};
                const PairClassDesc db = {9, 2, CP(10), CP(11), (const double*)t0, (const double*)t1};
                const PairClassDesc d1[2] = {{3, 1, CP(12), CP(13), (const double*)t0, (const double*)t1},
                                             {4, 1, CP(14), CP(15), (const double*)t2, (const double*)t3}};
                const PairClassDesc d0[4] = {{5, 0, CP(0), CP(3), (const double*)t0, (const double*)t1},
                                             {6, 0, CP(1), CP(2), (const double*)t2, (const double*)t3},
                                             {7, 0, CP(4), CP(5), (const double*)t0, (const double*)t1},
                                             {8, 0, CP(6), CP(7), (const double*)t0, (const double*)t1}};
                const FuseCols fcc{FUSE_COLS};
                ch.push_back({false, [=](hipStream_t st) -> int {
                    StageTimer t(ctx, st_pass, st, f_all);
                    t.traffic(gemm_bytes(14.0));
                    SSW_TRY(launch_dct_pair_gemm_multi_f64(st, false, true, 1, &da, dst, (double*)A1, n, w, h, ep, nullptr, nullptr, false, &fcc));
                    SSW_TRY(launch_dct_pair_gemm_multi_f64(st, false, true, 1, &db, dst, (double*)A1, n, w, h, ep, nullptr, (double*)T2, false, &fcc));
                    for (int c = 0; c < 2; ++c)
                        SSW_TRY(launch_dct_pair_gemm_multi_f64(st, false, true, 1, &d1[c], dst, (double*)T2, n, w, h, ep, nullptr, (double*)TE, false, &fcc));
                    for (int c = 0; c < 4; ++c)
                        SSW_TRY(launch_dct_pair_gemm_multi_f64(st, false, true, 1, &d0[c], dst, (double*)TE, n, w, h, ep, with_sink ? &sink : nullptr, nullptr, false, &fcc));
                    return SSW_OK;
                }});
                return SSW_OK;
            }
            ch.push_back({true, [=](hipStream_t st) -> int {
                StageTimer t(ctx, st_prep, st, prep_bytes);
                if (!is_row) return launch_dct_pair_prep16_inv_cols(st, src, n, w, h, sp, (const double*)rot, (const double*)rot2, cm, (const double*)rot3);
                return launch_dct_pair_prep16_inv_rows(st, src, n, w, h, sp, (const double*)rot, (const double*)rot2, (const double*)rot3);
            }});
            if (il2) {
                auto P = [=](int j) { return (const double*)(sp + (size_t)j * p16); };      // the planes of prep16_inv_rows_l2_kernel
                const double f_all = 8.0 * pair_gemm_flop(is_row, 7, 0, n, w, h);
                const PairClassDesc da = {1, 2, P(8), P(9), (const double*)h0, (const double*)h1};             // c[16 s] | c[16 s + 8] -> A1
                const PairClassDesc db = {9, 2, P(10), P(11), (const double*)t0, (const double*)t1};           // R2 rotated + A1 -> T2
                const PairClassDesc d1[2] = {{3, 1, P(12), P(13), (const double*)t0, (const double*)t1},       // AS2 BD2 + T2 -> E
                                             {4, 1, P(14), P(15), (const double*)t2, (const double*)t3}};      // AD2 BS2
                const PairClassDesc d0[4] = {{5, 0, P(0), P(3), (const double*)t0, (const double*)t1},         // AS+ BD-  + E -> x
                                             {6, 0, P(1), P(2), (const double*)t2, (const double*)t3},         // AS- BD+
                                             {7, 0, P(4), P(5), (const double*)t0, (const double*)t1},         // O rotated, "+"
                                             {8, 0, P(6), P(7), (const double*)t0, (const double*)t1}};        // O rotated, "-"
                ch.push_back({false, [=](hipStream_t st) -> int {
                    StageTimer t(ctx, st_pass, st, f_all);
                    t.traffic(gemm_bytes(14.0));
                    SSW_TRY(launch_dct_pair_gemm_multi_f64(st, is_row, true, 1, &da, dst, (double*)A1, n, w, h, ep));
                    SSW_TRY(launch_dct_pair_gemm_multi_f64(st, is_row, true, 1, &db, dst, (double*)A1, n, w, h, ep, nullptr, (double*)T2, rcm));
                    if (merge_classes(lines)) {          // single frames (merge_batch: batches too): the classes of each dependent stage in one launch
                        SSW_TRY(launch_dct_pair_gemm_multi_f64(st, is_row, true, 2, d1, dst, (double*)T2, n, w, h, ep, nullptr, (double*)TE, rcm));
                        return launch_dct_pair_gemm_multi_f64(st, is_row, true, 4, d0, dst, (double*)TE, n, w, h, ep, with_sink ? &sink : nullptr, nullptr, rcm);
                    }
                    for (int c = 0; c < 2; ++c)
                        SSW_TRY(launch_dct_pair_gemm_multi_f64(st, is_row, true, 1, &d1[c], dst, (double*)T2, n, w, h, ep, nullptr, (double*)TE, rcm));
                    for (int c = 0; c < 4; ++c)
                        SSW_TRY(launch_dct_pair_gemm_multi_f64(st, is_row, true, 1, &d0[c], dst, (double*)TE, n, w, h, ep, with_sink ? &sink : nullptr, nullptr, rcm));
                    return SSW_OK;
                }});
                return SSW_OK;
            }
            const double f_all = pair_gemm_flop(is_row, 3, 0, n, w, h) + pair_gemm_flop(is_row, 4, 0, n, w, h) + pair_gemm_flop(is_row, 1, 1, n, w, h) +
                                 pair_gemm_flop(is_row, 3, 1, n, w, h) + pair_gemm_flop(is_row, 4, 1, n, w, h);
            ch.push_back({false, [=](hipStream_t st) -> int {
                StageTimer t(ctx, st_pass, st, f_all);
                t.traffic(gemm_bytes(12.0));
                SSW_TRY(pair_gemm(st, true, is_row, true, 1, 1, sp + 4 * p8, sp + 5 * p8, e0, e1, dst, T2, n, w, h, ep));
                if (lines <= merge_max_lines()) {          // single frames: the two classes of each dependent stage in one launch
                    const PairClassDesc d1[2] = {{3, 1, q, q + p16, (const double*)t0, (const double*)t1},
                                                 {4, 1, q + 2 * p16, q + 3 * p16, (const double*)t2, (const double*)t3}};
                    SSW_TRY(launch_dct_pair_gemm_multi_f64(st, is_row, true, 2, d1, dst, (double*)T2, n, w, h, ep, nullptr, (double*)TE, rcm));
                    const PairClassDesc d0[2] = {{3, 0, sp, sp + p8, (const double*)sb0, (const double*)sb1},
                                                 {4, 0, sp + 2 * p8, sp + 3 * p8, (const double*)sb2_, (const double*)sb3}};
                    return launch_dct_pair_gemm_multi_f64(st, is_row, true, 2, d0, dst, (double*)TE, n, w, h, ep, with_sink ? &sink : nullptr, nullptr, rcm);
                }
                SSW_TRY(pair_gemm(st, true, is_row, true, 3, 1, q, q + p16, t0, t1, dst, T2, n, w, h, ep, nullptr, TE, rcm));
                SSW_TRY(pair_gemm(st, true, is_row, true, 4, 1, q + 2 * p16, q + 3 * p16, t2, t3, dst, T2, n, w, h, ep, nullptr, TE, rcm));
                SSW_TRY(pair_gemm(st, true, is_row, true, 3, 0, sp, sp + p8, sb0, sb1, dst, TE, n, w, h, ep, with_sink ? &sink : nullptr, nullptr, rcm));
                return pair_gemm(st, true, is_row, true, 4, 0, sp + 2 * p8, sp + 3 * p8, sb2_, sb3, dst, TE, n, w, h, ep, with_sink ? &sink : nullptr, nullptr, rcm);
            }});
            return SSW_OK;
        }
        if (three) {
            // forward pass, three levels: x- (odd frequencies), S- (2 mod 4), (SSS, SS-) (0 and 4 mod 8); on a column
            // pass (8K: 4320 rows) the pre-pass transposes like the two-level one
            for (int b = 0; b < 4; ++b) SSW_TRY(grow(ws.operand[b], bytes));
            void* d1 = ws.operand[1].p;
            void* d2 = ws.operand[0].p;
            void* r1 = ws.operand[2].p;
            void* r2 = ws.operand[3].p;
            const void *h1 = nullptr, *e0 = nullptr, *e1 = nullptr;
            SSW_TRY(get_basis(ctx, len / 2, false, f64, 4, &h1));          // odd half basis of len/2
            SSW_TRY(get_basis(ctx, len / 4, false, f64, 3, &e0));          // half bases of len/4
            SSW_TRY(get_basis(ctx, len / 4, false, f64, 4, &e1));
            ch.push_back({true, [=](hipStream_t st) -> int {
                StageTimer t(ctx, st_prep, st, prep_bytes);
                if (!is_row) SSW_TRY(launch_dct_pair_prep8_cols(st, f64, src, n, w, h, r1, r2, d2, d1));
                else SSW_TRY(launch_dct_pair_prep8_rows(st, f64, from_rgb ? pix_src_kind(rgb_u8) : 0, from_rgb ? rgb : (const void*)src, n, w, h,
                                                        r1, r2, d2, d1, from_rgb ? iq_i : nullptr, from_rgb ? iq_q : nullptr));
                return odd_rotate(st, d1);
            }});
            const double f_main = f_odd;
            const double f_all = f_main + pair_gemm_flop(is_row, 1, 1, n, w, h) + pair_gemm_flop(is_row, 2, 1, n, w, h);
            ch.push_back({false, [=](hipStream_t st) -> int {
                StageTimer t(ctx, st_pass, st, f_all);
                t.traffic(gemm_bytes(0.0));
                SSW_TRY(pair_gemm(st, f64, is_row, inverse, 1, 1, r1, r2, e0, e1, dst, nullptr, n, w, h, ep));
                SSW_TRY(pair_gemm(st, f64, is_row, inverse, 2, 1, d2, d2, h1, (const char*)h1 + (len / 8) * 64, dst, nullptr, n, w, h, ep));
                StageTimer tm(ctx, st_main, st, f_main);
                return odd_gemm(st, d1, b1, nullptr, nullptr);
            }});
        } else if (!two) {
            for (int b = 0; b < 2; ++b) SSW_TRY(grow(ws.operand[b], bytes));
            void* x1 = ws.operand[0].p;
            void* x2 = ws.operand[1].p;
            ch.push_back({true, [=](hipStream_t st) -> int {
                StageTimer t(ctx, SSW_STAGE_DCT_PREP, st, prep_bytes);
                return launch_dct_pair_prep(st, f64, is_row, inverse, src, n, w, h, x1, x2);
            }});
            const double f_main = pair_gemm_flop(is_row, 0, 0, n, w, h);
            ch.push_back({false, [=](hipStream_t st) -> int {
                StageTimer t(ctx, st_pass, st, f_main);
                t.traffic(gemm_bytes(0.0));
                StageTimer tm(ctx, st_main, st, f_main);
                return pair_gemm(st, f64, is_row, inverse, 0, 0, x1, x2, b0, b1, dst, nullptr, n, w, h, ep);
            }});
        } else {
            for (int b = 1; b < (inverse ? 5 : 4); ++b) SSW_TRY(grow(ws.operand[b], bytes));
            void* x2 = ws.operand[1].p;       // D | O
            void* xx1 = ws.operand[2].p;      // SS | EE
            void* xx2 = ws.operand[3].p;      // SD | EO
            void* tmpE = ws.operand[4].p;     // inverse: the even half E, unrounded
            const void *q0 = nullptr, *q1 = nullptr;
            SSW_TRY(get_basis(ctx, len / 2, inverse, f64, 3, &q0));
            SSW_TRY(get_basis(ctx, len / 2, inverse, f64, 4, &q1));
            ch.push_back({true, [=](hipStream_t st) -> int {
                StageTimer t(ctx, st_prep, st, prep_bytes);
                if (from_rgb) SSW_TRY(launch_dct_pair_prep4_rows_rgb(st, f64, rgb_u8, rgb, n, w, h, xx1, xx2, x2, iq_i, iq_q));
                else SSW_TRY(launch_dct_pair_prep4(st, f64, is_row, inverse, src, n, w, h, xx1, xx2, x2));
                return odd_rotate(st, x2);
            }});
            const double f_main = f_odd;
            const double f_all = f_main + pair_gemm_flop(is_row, 1, 0, n, w, h);
            // Writer::result: the last pass of an inverse transform (a column pass) converts to RGB in its epilogue
            RgbSink sink;
            if (inverse && !first_pass && !is_row && x.rgb_out && x.iq_i && x.iq_q) {
                sink.iq_i = x.iq_i; sink.iq_q = x.iq_q; sink.rgb = x.rgb_out; sink.u8 = x.rgb_out_u8;
                if (fused_rgb) *fused_rgb = true;
            }
            const bool with_sink = sink.rgb != nullptr;
            ch.push_back({false, [=](hipStream_t st) -> int {
                StageTimer t(ctx, st_pass, st, f_all);
                t.traffic(gemm_bytes(inverse ? esz : 0.0));          // inverse: the even half E out and in
                // even half: a half-length transform of S (forward) / of the even coefficients (inverse), folded again
                SSW_TRY(pair_gemm(st, f64, is_row, inverse, 1, 0, xx1, xx2, q0, q1, dst, tmpE, n, w, h, ep));
                // odd half: full half-length sum, the odd basis split into two row blocks (second block:
                // len/4 lines further inside every k-block of the same plane = 64 bytes per line)
                StageTimer tm(ctx, st_main, st, f_main);
                return odd_gemm(st, x2, b1, tmpE, with_sink ? &sink : nullptr);
            }});
        }
        return SSW_OK;
    }
    const void *b0 = nullptr, *b1 = nullptr;
    if (fold) {
        SSW_TRY(get_basis(ctx, len, inverse, f64, 1, &b0));
        SSW_TRY(get_basis(ctx, len, inverse, f64, 2, &b1));
    } else {
        SSW_TRY(get_basis(ctx, len, inverse, f64, 0, &b0));
    }
    const double dense = is_row ? 2.0 * n * h * (double)w * w : 2.0 * n * w * (double)h * h;
    const double flop = fold ? 0.5 * dense : dense;
    if (is_row) {
        ch.push_back({false, [=](hipStream_t st) -> int {
            StageTimer t(ctx, SSW_STAGE_DCT_ROW, st, flop);
            t.traffic(px * 8.0);
            if (fold && f64) return launch_dct_rows_folded_f64(st, inverse, src, dst, n * h, w, (const double*)b0, (const double*)b1, ep);
            if (fold) return launch_dct_rows_folded_f32(st, inverse, src, dst, n * h, w, (const float*)b0, (const float*)b1, ep);
            return launch_dct_rows(st, precision, src, dst, n * h, w, b0, ep);
        }});
    } else {
        ch.push_back({false, [=](hipStream_t st) -> int {
            StageTimer t(ctx, SSW_STAGE_DCT_COL, st, flop);
            t.traffic(px * 8.0);
            if (fold && f64) return launch_dct_cols_folded_f64(st, inverse, src, dst, n, w, h, (const double*)b0, (const double*)b1, ep);
            if (fold) return launch_dct_cols_folded_f32(st, inverse, src, dst, n, w, h, (const float*)b0, (const float*)b1, ep);
            return launch_dct_cols(st, precision, src, dst, n, w, h, b0, ep);
        }});
    }
    return SSW_OK;
}

// The operand-ready GEMMs walk an operand plane with 32-bit scalar offsets: a call's planes must stay below
// 4 GB, so more frames than that are transformed in groups (frames are independent).
size_t operand_frame_limit(const ssw_ctx* ctx, bool f64, size_t w, size_t h) {
    if (!(ctx->fold && ctx->fold_level >= 3)) return ~(size_t)0;
    const size_t per_frame = dct_pair_operand_elems(f64, 1, w, h) * (f64 ? sizeof(double) : sizeof(float));
    const size_t m = per_frame ? 0xFFFFFFFFull / per_frame : ~(size_t)0;
    return m >= 1 ? m : ~(size_t)0;
}

}  // namespace

int build_transform(ssw_ctx* ctx, ssw_ctx::Lane& ws, const Xform& x, Chain& ch, bool* fused_rgb) {
    const bool f64 = (x.precision == SSW_PRECISION_F64);
    const size_t n = x.n, w = x.w, h = x.h;
    if (n == 0) return SSW_OK;
    const size_t max_frames = operand_frame_limit(ctx, f64, w, h);
    if (n > max_frames) {
        for (size_t f0 = 0; f0 < n; f0 += max_frames) {
            Xform s = x;
            s.n = std::min(max_frames, n - f0);
            s.data = x.data + f0 * w * h;
            s.tmp = x.tmp + f0 * w * h;
            if (x.rgb) s.rgb = static_cast<const char*>(x.rgb) + f0 * w * h * 3 * pix_bytes(x.rgb_u8);
            if (x.iq_i) s.iq_i = x.iq_i + f0 * w * h;
            if (x.iq_q) s.iq_q = x.iq_q + f0 * w * h;
            if (x.rgb_out) s.rgb_out = static_cast<char*>(x.rgb_out) + f0 * w * h * 3 * (x.rgb_out_u8 ? 1 : sizeof(float));
            SSW_TRY(build_transform(ctx, ws, s, ch, fused_rgb));
        }
        return SSW_OK;
    }
    const bool rows_first = (w >= h);                                  // src/dct2d.rs:93-98
    Epilogue plain{1.f, 1.f};
    auto ortho = [&](size_t len) {                                      // src/dct2d.rs:154-155
        Epilogue e{std::sqrt(1.0f / (4.0f * (float)len)), std::sqrt(1.0f / (2.0f * (float)len))};
        return e;
    };
    Epilogue last = plain;
    if (x.type == SSW_DCT3) last.first = last.base = (float)4 / (float)(w * h);           // :213-217
    for (int pass = 0; pass < 2; ++pass) {
        const bool is_row = (pass == 0) ? rows_first : !rows_first;
        const float* src = (pass == 0) ? x.data : x.tmp;
        float* dst = (pass == 0) ? x.tmp : x.data;
        const Epilogue ep = (x.type == SSW_DCT2_ORTHOGONAL) ? ortho(is_row ? w : h) : (pass == 1 ? last : plain);
        SSW_TRY(build_pass(ctx, ws, x, pass == 0, is_row, src, dst, ep, ch, fused_rgb));
    }
    return SSW_OK;
}

bool can_fuse_rgb(const ssw_ctx* ctx, bool f64, size_t w, size_t h, const float* y, const float* tmp, const void* rgb, int u8) {
    return ctx->fold && ctx->fold_level >= 4 && dct_pair_can_run(f64, 1, w, h, y, tmp) && dct_pair_can_prep_from_rgb(w, h, rgb, u8);
}

// Writer::new / Reader::base / Reader::derived: rgb -> Y (+ I, Q) -> forward 2-D DCT of Y into `y`.
// Where the default GEMM strategy applies (rows first, two folding levels on the row axis) the colour
// conversion is fused into the first operand pre-pass and the f32 Y plane is never materialised.
int build_forward_from_rgb(ssw_ctx* ctx, ssw_ctx::Lane& ws, int precision, const void* rgb, int u8, size_t n, size_t w,
                           size_t h, float* y, float* i, float* q, float* tmp, Chain& ch) {
    const bool f64 = precision == SSW_PRECISION_F64;
    Xform x{SSW_DCT2, precision, n, w, h, y, tmp};
    if (can_fuse_rgb(ctx, f64, w, h, y, tmp, rgb, u8)) {
        x.rgb = rgb; x.rgb_u8 = u8; x.iq_i = i; x.iq_q = q;
        return build_transform(ctx, ws, x, ch);
    }
    const size_t npix = n * w * h;
    const double bytes = (double)npix * (3.0 * (double)pix_bytes(u8) + (i ? 12.0 : 4.0));
    ch.push_back({true, [=](hipStream_t st) -> int {
        StageTimer t(ctx, SSW_STAGE_RGB_TO_YIQ, st, bytes);
        if (u8 == SSW_PIX_U8) return launch_rgb8_to_yiq(st, static_cast<const uint8_t*>(rgb), npix, y, i, q);
        if (u8 == SSW_PIX_U16) return launch_rgb16_to_yiq(st, static_cast<const uint16_t*>(rgb), npix, y, i, q);
        return launch_rgb_to_yiq(st, static_cast<const float*>(rgb), npix, y, i, q);
    }});
    return build_transform(ctx, ws, x, ch);
}

// The two passes of a rows-first forward transform from RGB as separate chains (single-image handles: the row pass
// of the top half of a frame runs while the bottom half is still crossing PCIe -- image rows are independent lines
// of a row pass, so any band of rows gives the values the whole frame gives).  `rows` consecutive image rows starting
// at `rgb` -> the same rows of the intermediate plane `tmp` (+ I, Q); then the column pass tmp -> y on the whole frame.
bool can_split_forward_rows(const ssw_ctx* ctx, bool f64, size_t w, size_t h, size_t bands, const float* y, const float* tmp, const void* rgb, int u8) {
    return bands >= 2 && w >= h && h % 16 == 0 && h % bands == 0 && can_fuse_rgb(ctx, f64, w, h, y, tmp, rgb, u8) &&
           can_fuse_rgb(ctx, f64, w, h / bands, y, tmp, rgb, u8);
}
int build_forward_rows_band(ssw_ctx* ctx, ssw_ctx::Lane& ws, int precision, const void* rgb, int u8, size_t w, size_t rows,
                            size_t frame_h, float* tmp, float* i, float* q, Chain& ch) {
    Xform x{SSW_DCT2, precision, 1, w, rows, tmp /* never read or written by this pass */, tmp};
    x.rgb = rgb; x.rgb_u8 = u8; x.iq_i = i; x.iq_q = q;
    x.full_h = frame_h;
    return build_pass(ctx, ws, x, true, true, tmp, tmp, Epilogue{1.f, 1.f}, ch);
}
int build_forward_cols_after_rows(ssw_ctx* ctx, ssw_ctx::Lane& ws, int precision, size_t w, size_t h, float* tmp, float* y, Chain& ch) {
    Xform x{SSW_DCT2, precision, 1, w, h, y, tmp};
    x.full_h = h;            // the row pass ran band by band through the f32 plane: the column pass reads it back (no fused operands)
    return build_pass(ctx, ws, x, false, false, tmp, y, Epilogue{1.f, 1.f}, ch);
}

int run_serial(Chain& ch, hipStream_t st) {
    for (auto& s : ch) SSW_TRY(s.run(st));
    return SSW_OK;
}

int dct2d_planes(ssw_ctx* ctx, int type, int precision, size_t n, size_t w, size_t h, float* data, float* tmp) {
    Chain ch;
    Xform x{type, precision, n, w, h, data, tmp};
    SSW_TRY(build_transform(ctx, ctx->lane[0], x, ch));
    return run_serial(ch, ctx->stream);
}

int topk(ssw_ctx* ctx, hipStream_t st, SelectWorkspace& sel, const float* coef, size_t n, size_t w, size_t h, int ordering,
         size_t k, uint32_t* idx) {
    const double bytes = 4.0 * (double)n * (double)w * (double)h;
    if (k > select_max_k()) {
        // Beyond the in-LDS top-k limit (16384 entries; BASELINE marks are 1000 and 10000 long): the full order of
        // every plane (sort_full.hip: a batched radix sort over all frames of the call), first k entries kept.  Off
        // the hot path: only Reader::indices() with a large k and marks that long reach it.
        size_t sb = 0;
        SSW_TRY(full_sort_scratch_bytes(w * h, n, &sb));
        SSW_TRY(grow(ctx->sort_scratch, sb));
        StageTimer t(ctx, SSW_STAGE_SELECT, st, bytes);
        return launch_full_sort(st, coef, n, w, h, ordering, ctx->sort_scratch.p, ctx->sort_scratch.bytes, idx, k);
    }
    SSW_TRY(grow_select(st, sel, n, k));
    sel.fallbacks = ctx->select_fallbacks;
    ctx->select_frames += n;
    StageTimer t(ctx, SSW_STAGE_SELECT, st, bytes);
    return launch_topk(st, coef, n, w, h, ordering, k, sel, idx);
}

// ---- two-lane pipeline ----------------------------------------------------------------------------------
namespace {

int next_sync_event(ssw_ctx* ctx, hipEvent_t* out) {
    constexpr size_t RING = 256;
    if (ctx->sync_events.size() < RING) {
        hipEvent_t e = nullptr;
        SSW_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->sync_events.push_back(e);
        *out = e;
        return SSW_OK;
    }
    *out = ctx->sync_events[ctx->sync_next % RING];
    ctx->sync_next++;
    return SSW_OK;
}

// A lane's stage has just been enqueued on ln.cur: remember that point (an event recorded NOW -- recorded
// later it would also cover what the other lane has enqueued on the same stream in the meantime, and the
// lanes would serialise each other).
int mark_done(ssw_ctx* ctx, ssw_ctx::Lane& ln) {
    SSW_TRY(next_sync_event(ctx, &ln.done));
    SSW_HIP_CHECK(hipEventRecord(ln.done, ln.cur));
    untimed_work(ctx);
    return SSW_OK;
}

// continue the lane's chain on stream `to`: the lane's last stage happens before
int hop(ssw_ctx* ctx, ssw_ctx::Lane& ln, hipStream_t to) {
    if (ln.cur == to) return SSW_OK;
    if (ln.done) SSW_HIP_CHECK(hipStreamWaitEvent(to, ln.done, 0));
    untimed_work(ctx);                             // the next stage's timer starts behind the wait, not in front of it
    ln.cur = to;
    return SSW_OK;
}

bool pipeline_uses_two_lanes(const ssw_ctx* ctx, size_t n_chunks) {
    // from two passes on (r3: with the pre-passes a quarter of a pass, the second lane's pre-passes fill the gaps between
    // the first lane's GEMM launches: +2.1 % at 2 x 128 4K frames; r2 measured nothing below three passes)
    return ctx->overlap && n_chunks >= 2 && ctx->aux_stream != nullptr;
}

// Runs build(chunk, lane, chain) for every chunk and enqueues the chains: one lane on one stream when
// overlap is off, else two lanes round-robin with GEMM stages on ctx->stream and HBM-bound stages on
// ctx->aux_stream.  On return the context's stream is ordered after everything that was enqueued.
int run_pipeline_impl(ssw_ctx* ctx, size_t n_chunks, bool two, const std::function<int(size_t, ssw_ctx::Lane&, Chain&)>& build) {
    hipStream_t G = ctx->stream, H = two ? ctx->aux_stream : ctx->stream;
    const int n_lanes = two ? 2 : 1;
    if (two) {                                     // the caller's earlier work on the context's stream comes first
        hipEvent_t ev = nullptr;
        SSW_TRY(next_sync_event(ctx, &ev));
        SSW_HIP_CHECK(hipEventRecord(ev, G));
        SSW_HIP_CHECK(hipStreamWaitEvent(H, ev, 0));
        untimed_work(ctx);
    }
    Chain chain[ssw_ctx::MAX_LANES];
    size_t at[ssw_ctx::MAX_LANES] = {0, 0};
    bool active[ssw_ctx::MAX_LANES] = {false, false};
    size_t next = 0;
    auto start = [&](int l) -> int {
        active[l] = false;
        while (next < n_chunks) {
            chain[l].clear();
            at[l] = 0;
            SSW_TRY(build(next++, ctx->lane[l], chain[l]));
            if (!chain[l].empty()) { active[l] = true; break; }
        }
        return SSW_OK;
    };
    for (int l = 0; l < n_lanes; ++l) {
        ctx->lane[l].cur = H;
        ctx->lane[l].done = nullptr;
        SSW_TRY(start(l));
    }
    // `lane_stagger` (r5 experiment): the second lane starts one stage late, and an RGB pre-pass waits until the other lane's
    // row pass is through -- it then runs beside that lane's COLUMN launches (MFMA-bound, 12 B/px) instead of its row
    // launches (16 B/px and an epilogue that is HBM-bound itself)
    const bool stagger = two && tuning(TUNE_LANE_STAGGER) != 0;
    bool held_back = stagger;
    while (active[0] || active[1]) {
        for (int l = 0; l < n_lanes; ++l) {
            if (!active[l]) continue;
            if (l == 1 && held_back) { held_back = false; continue; }
            Stage& s = chain[l][at[l]];
            SSW_TRY(hop(ctx, ctx->lane[l], s.hbm ? H : G));
            if (stagger && s.tag == 2) {
                const int o = 1 - l;
                if (active[o] && at[o] > 0 && chain[o][at[o] - 1].tag == 1 && ctx->lane[o].done)
                    SSW_HIP_CHECK(hipStreamWaitEvent(ctx->lane[l].cur, ctx->lane[o].done, 0));
                untimed_work(ctx);
            }
            SSW_TRY(s.run(ctx->lane[l].cur));
            if (two) SSW_TRY(mark_done(ctx, ctx->lane[l]));
            if (++at[l] == chain[l].size()) SSW_TRY(start(l));
        }
    }
    for (int l = 0; l < n_lanes; ++l) SSW_TRY(hop(ctx, ctx->lane[l], G));
    return SSW_OK;
}

int run_pipeline(ssw_ctx* ctx, size_t n_chunks, const std::function<int(size_t, ssw_ctx::Lane&, Chain&)>& build) {
    if (n_chunks == 0) return SSW_OK;
    // two lanes pay from three chunks on (with two, each lane would run a single chunk: measured equal to one lane)
    const bool two = pipeline_uses_two_lanes(ctx, n_chunks);
    int rc = run_pipeline_impl(ctx, n_chunks, two, build);
    if (rc == SSW_ERR_OUT_OF_MEMORY && !ctx->retired.empty()) {
        // Workspaces that grew during this call left their old allocations retired (grow(): stages already built may hold
        // them), so the call's peak was old + new.  Everything enqueued so far is complete after the waits below and nothing
        // built is still to run: give the retired buffers back and run the chunks again (same kernels, same outputs).
        if (two) (void)hipStreamSynchronize(ctx->aux_stream);
        (void)hipStreamSynchronize(ctx->stream);
        for (void* p : ctx->retired) (void)hipFree(p);
        ctx->retired.clear();
        untimed_work(ctx);
        rc = run_pipeline_impl(ctx, n_chunks, two, build);
    }
    if (rc != SSW_OK && two) {
        // a failing stage (out of memory inside grow(), most likely) must not leave the second stream running
        // behind the caller's back: the lanes' buffers are reused by the next call on the context's stream
        (void)hipStreamSynchronize(ctx->aux_stream);
        (void)hipStreamSynchronize(ctx->stream);
    }
    return rc;
}

// ---- pruned transform of the derived frames (prune.hip) ---------------------------------------------------
struct PruneSetup {
    bool on = false;
    PrunePlan plan;
    int levels = 0;                 // folding levels of the forward row pass: 2 or 3
    bool split = false;             // odd frequencies through the split odd half (two classes instead of one)
    bool deep = false;              // deep row pre-pass: frequencies 2 mod 4 split as well, 0 / 4 mod 8 from the third level
};

// capacity of the compact plane in frequency columns: the index lists of natural spectra use ~3 sqrt(k)
// distinct columns (measured: 80..110 for k = 1000 at full HD and 4K); 8 sqrt(k), split over the classes
// in proportion to their share of all frequencies, leaves a wide margin, and a chunk that does not fit is
// redone with the full transform.
size_t prune_capacity(size_t k) {
    size_t c = (size_t)std::ceil(8.0 * std::sqrt((double)k));
    c = (c + 31) / 32 * 32;
    return std::max<size_t>(c, 64);
}

PruneSetup make_prune_setup(const ssw_ctx* ctx, bool f64, size_t n, size_t w, size_t h, size_t k, const float* y, const float* tmp,
                            const void* rgb, int u8) {
    PruneSetup ps;
    if (!ctx->prune || k == 0) return ps;
    if (!can_fuse_rgb(ctx, f64, w, h, y, tmp, rgb, u8)) return ps;       // rows first, >= two folding levels on the rows
    if (!dct_pair_can_run(f64, n, w, h, y, tmp)) return ps;               // the chunk's planes within the 4 GB walk
    const size_t cap = prune_capacity(k);
    if (cap * 4 > w) return ps;                                           // not worth it: full transform
    if (!dct_pair_can_run(f64, n, cap, h, y, tmp)) return ps;
    const bool three = dct_pair_can_fold3(w) && (ctx->fold_level >= 6 || (ctx->fold_level == 5 && w >= 3072));
    ps.levels = three ? 3 : 2;
    ps.plan.W = (unsigned)w;
    ps.plan.cap_total = (unsigned)cap;
    const unsigned c = (unsigned)cap;
    ps.split = f64 && ctx->split && dct_pair_can_split(w, true);
    unsigned nc = 0, off = 0;
    auto add = [&](unsigned mod, unsigned rem, unsigned cc, unsigned rem2 = PRUNE_NO_REM, unsigned radd = 0) {
        ps.plan.c[nc] = {mod, rem, cc, off, rem2, radd};
        off += cc;
        ++nc;
    };
    ps.deep = ps.split && dct_pair_can_deep_rows(w);
    if (ps.deep && dct_pair_efold(w)) {
        // level 2 (build_pass): nine classes of sums of w/16 terms; v = 16i +/- r -> row i (= (v + r) / 16) of the bases of
        // E even, v = 16i + 9 | 16i + 7 -> row i of E odd; the same two for the half-length split; R1 folded: 16i, 16i + 8
        add(16, 1, c / 8, 15, 1);         // AS+ BD-
        add(16, 9, c / 8, 7, 0);          // AS- BD+
        add(16, 5, c / 8, 11, 5);         // O rotated "+"
        add(16, 3, c / 8, 13, 3);         // O rotated "-"
        add(16, 2, c / 8, 14, 2);         // AS2 BD2
        add(16, 10, c / 8, 6, 0);         // AD2 BS2
        add(16, 4, c / 8, 12, 4);         // R2 rotated
        add(16, 0, c / 16);               // R1+
        add(16, 8, c / 16);               // R1-
        ps.plan.n_classes = nc;
        ps.on = true;
        return ps;
    }
    if (ps.split) {                       // odd v = 8i +/- 1 -> class E row i (= (v + 1) / 8), v = 8i + 5 | 8i + 3 -> class O row i
        add(8, 1, c / 4, 7, 1);
        add(8, 5, c / 4, 3, 0);
    } else {
        add(2, 1, c / 2);
    }
    if (ps.deep) {                        // v = 2 (8i +/- 1) -> class E' row i (= (v + 2) / 16), v = 2 (8i + 5) | 2 (8i + 3) -> class O' row i
        add(16, 2, c / 8, 14, 2);
        add(16, 10, c / 8, 6, 0);
        add(8, 0, c / 8);
        add(8, 4, c / 8);
    } else if (three) {
        add(4, 2, c / 4);
        add(8, 0, c / 8);
        add(8, 4, c / 8);
    } else {
        add(4, 0, c / 4);
        add(4, 2, c / 4);
    }
    ps.plan.n_classes = nc;
    ps.on = true;
    return ps;
}

// derived rgb frames -> compact coefficient plane ws.compact[1] [n][h][cap_total] holding, for every
// frequency column the chunk's index lists use, the column the full transform would produce
int build_pruned_derived(ssw_ctx* ctx, ssw_ctx::Lane& ws, int precision, const void* rgb, int u8, size_t n, size_t w,
                         size_t h, size_t k, const uint32_t* idx, const PruneSetup& ps, uint32_t* info, Chain& ch) {
    const bool f64 = precision == SSW_PRECISION_F64;
    const size_t esz = f64 ? 8 : 4;
    const PrunePlan plan = ps.plan;
    const size_t cap = plan.cap_total;
    const size_t bytes = dct_pair_operand_elems(f64, n, w, h) * esz;
    if (!ps.deep) for (int b = 0; b < 4; ++b) SSW_TRY(grow(ws.operand[b], bytes));      // the deep pre-pass writes into operand[5] only
    for (int b = 0; b < 2; ++b) SSW_TRY(grow(ws.compact[b], n * h * cap * sizeof(float)));
    SSW_TRY(grow(ws.prune_u32, (2 * w + cap + 64) * sizeof(uint32_t)));
    uint32_t* flag = (uint32_t*)ws.prune_u32.p;
    uint32_t* pos = flag + w;
    uint32_t* rows = pos + w;
    // class -> image operand plane(s), cached basis plane(s), padded / true sum length
    struct ClassSrc { const void* x; const void* basis; size_t src_rows, kp, ktrue; const void* x2 = nullptr; const void* basis2 = nullptr; };
    ClassSrc cs[9];
    int pn1[9] = {0}, pn2[9] = {0};          // level 2: the classes' operand planes by number (the fused kernel's A-fragments)
    bool level2 = false;
    unsigned ci = 0;
    const size_t lines = n * h;
    const void *rot = nullptr, *rot2 = nullptr, *rot3 = nullptr;
    double* sp = nullptr;
    if (ps.deep && dct_pair_efold(w)) {          // level 2: the plan of make_prune_setup, planes by number like build_pass
        const void *sb2[4], *h0 = nullptr, *h1 = nullptr;
        for (int b = 0; b < 4; ++b) SSW_TRY(get_basis(ctx, w / 2, false, true, 5 + b, &sb2[b]));
        SSW_TRY(get_basis(ctx, w, false, true, 9, &rot));
        SSW_TRY(get_basis(ctx, w / 2, false, true, 9, &rot2));
        SSW_TRY(get_basis(ctx, w / 4, false, true, 9, &rot3));
        SSW_TRY(get_basis(ctx, w / 8, false, true, 3, &h0));
        SSW_TRY(get_basis(ctx, w / 8, false, true, 4, &h1));
        SSW_TRY(grow(ws.operand[5], split_scratch_elems(n, w, h) * sizeof(double)));
        sp = (double*)ws.operand[5].p;
        const size_t kp16 = dct_pair_split_kpad(w / 2), p16 = lines * kp16;
        const size_t re = dct_pair_split_basis_rows(w / 2, 0), ro = dct_pair_split_basis_rows(w / 2, 2);
        auto P = [=](int j) { return (const void*)(sp + (size_t)j * p16); };
        cs[ci++] = {P(0), sb2[0], re, kp16, w / 16, P(3), sb2[1]};
        cs[ci++] = {P(1), sb2[2], ro, kp16, w / 16, P(2), sb2[3]};
        cs[ci++] = {P(4), sb2[0], re, kp16, w / 16, P(5), sb2[1]};
        cs[ci++] = {P(6), sb2[0], re, kp16, w / 16, P(7), sb2[1]};
        cs[ci++] = {P(12), sb2[0], re, kp16, w / 16, P(13), sb2[1]};
        cs[ci++] = {P(14), sb2[2], ro, kp16, w / 16, P(15), sb2[3]};
        cs[ci++] = {P(10), sb2[0], re, kp16, w / 16, P(11), sb2[1]};
        cs[ci++] = {P(8), h0, w / 16, kp16, w / 16};
        cs[ci++] = {P(9), h1, w / 16, kp16, w / 16};
        static const int planes_l2[9][2] = {{0, 3}, {1, 2}, {4, 5}, {6, 7}, {12, 13}, {14, 15}, {10, 11}, {8, -1}, {9, -1}};
        for (int c = 0; c < 9; ++c) { pn1[c] = planes_l2[c][0]; pn2[c] = planes_l2[c][1]; }
        level2 = true;
    } else if (ps.split) {
        const void* sb[4];
        for (int b = 0; b < 4; ++b) SSW_TRY(get_basis(ctx, w, false, true, 5 + b, &sb[b]));
        SSW_TRY(get_basis(ctx, w, false, true, 9, &rot));
        const size_t kp8 = dct_pair_split_kpad(w), plane = lines * kp8;
        SSW_TRY(grow(ws.operand[5], split_scratch_elems(n, w, h) * sizeof(double)));
        sp = (double*)ws.operand[5].p;
        const void *sb2[4] = {nullptr, nullptr, nullptr, nullptr}, *e0 = nullptr, *e1 = nullptr;
        const size_t kp16 = dct_pair_split_kpad(w / 2), p16 = lines * kp16;
        double* q = sp + 6 * plane;
        if (ps.deep) for (int b = 0; b < 4; ++b) SSW_TRY(get_basis(ctx, w / 2, false, true, 5 + b, &sb2[b]));
        cs[ci++] = {sp, sb[0], dct_pair_split_basis_rows(w, 0), kp8, w / 8, sp + plane, sb[1]};              // AS x cosE, BD x sinE
        cs[ci++] = {sp + 2 * plane, sb[2], dct_pair_split_basis_rows(w, 2), kp8, w / 8, sp + 3 * plane, sb[3]};  // AD x cosO, BS x sinO
        if (ps.deep) {
            SSW_TRY(get_basis(ctx, w / 2, false, true, 9, &rot2));
            SSW_TRY(get_basis(ctx, w / 4, false, true, 3, &e0));
            SSW_TRY(get_basis(ctx, w / 4, false, true, 4, &e1));
            cs[ci++] = {q, sb2[0], dct_pair_split_basis_rows(w / 2, 0), kp16, w / 16, q + p16, sb2[1]};              // AS2 x cosE', BD2 x sinE'
            cs[ci++] = {q + 2 * p16, sb2[2], dct_pair_split_basis_rows(w / 2, 2), kp16, w / 16, q + 3 * p16, sb2[3]}; // AD2 x cosO', BS2 x sinO'
            cs[ci++] = {sp + 4 * plane, e0, w / 8, kp8, w / 8};                                                      // R1 (SSS): 0 mod 8
            cs[ci++] = {sp + 5 * plane, e1, w / 8, kp8, w / 8};                                                      // R2 (SS-): 4 mod 8
        }
    } else {
        const void* b1 = nullptr;
        SSW_TRY(get_basis(ctx, w, false, f64, 4, &b1));
        cs[ci++] = {ws.operand[1].p, b1, w / 2, dct_pair_kpad(f64, w), w / 2};       // x- | D : odd
    }
    if (ps.deep) {
    } else if (ps.levels == 3) {
        const void *h1 = nullptr, *e0 = nullptr, *e1 = nullptr;
        SSW_TRY(get_basis(ctx, w / 2, false, f64, 4, &h1));
        SSW_TRY(get_basis(ctx, w / 4, false, f64, 3, &e0));
        SSW_TRY(get_basis(ctx, w / 4, false, f64, 4, &e1));
        cs[ci++] = {ws.operand[0].p, h1, w / 4, dct_pair_kpad(f64, w / 2), w / 4};      // S-  : 2 mod 4
        cs[ci++] = {ws.operand[2].p, e0, w / 8, dct_pair_kpad(f64, w / 4), w / 8};      // SSS : 0 mod 8
        cs[ci++] = {ws.operand[3].p, e1, w / 8, dct_pair_kpad(f64, w / 4), w / 8};      // SS- : 4 mod 8
    } else {
        const void *q0 = nullptr, *q1 = nullptr;
        SSW_TRY(get_basis(ctx, w / 2, false, f64, 3, &q0));
        SSW_TRY(get_basis(ctx, w / 2, false, f64, 4, &q1));
        cs[ci++] = {ws.operand[2].p, q0, w / 4, dct_pair_kpad(f64, w / 2), w / 4};      // SS : 0 mod 4
        cs[ci++] = {ws.operand[3].p, q1, w / 4, dct_pair_kpad(f64, w / 2), w / 4};      // SD : 2 mod 4
    }
    if (ci != plan.n_classes) return SSW_ERR_BAD_ARG;
    size_t goff[9], goff2[9], gtotal = 0;
    for (unsigned c = 0; c < plan.n_classes; ++c) {
        const size_t cap16 = (plan.c[c].cap + 15) / 16 * 16;               // (whole tiles of 16 rows: the fused pass's fragment order)
        goff[c] = gtotal; gtotal += cs[c].kp * cap16 * esz;
        goff2[c] = gtotal; if (cs[c].x2) gtotal += cs[c].kp * cap16 * esz;
    }
    SSW_TRY(grow(ws.gathered, gtotal));
    char* gathered = (char*)ws.gathered.p;
    float* t_compact = (float*)ws.compact[0].p;
    void *o0 = ws.operand[0].p, *o1 = ws.operand[1].p, *o2 = ws.operand[2].p, *o3 = ws.operand[3].p;
    const int levels = ps.levels;
    const bool deep = ps.deep;
    const double px = (double)n * (double)w * (double)h;
    const double prep_bytes = px * (3.0 * (double)pix_bytes(u8) + (double)esz);
    double flop = 0.0;
    for (unsigned c = 0; c < plan.n_classes; ++c) flop += (cs[c].x2 ? 4.0 : 2.0) * (double)lines * plan.c[c].cap * (double)cs[c].ktrue;
    auto gather_jobs = [=](bool frag) {
        PruneGatherJobs jobs;
        jobs.n = 0;
        for (unsigned c = 0; c < plan.n_classes; ++c) {
            const unsigned kblocks = (unsigned)(cs[c].kp / (64 / esz));
            jobs.j[jobs.n++] = {rows + plan.c[c].off, (const char*)cs[c].basis, gathered + goff[c], plan.c[c].cap, (unsigned)cs[c].src_rows, kblocks, 0u, false, frag};
            if (cs[c].x2) jobs.j[jobs.n++] = {rows + plan.c[c].off, (const char*)cs[c].basis2, gathered + goff2[c], plan.c[c].cap, (unsigned)cs[c].src_rows, kblocks, 0u, true, frag};
        }
        return jobs;
    };
    // r5: marks of up to 1024 entries at level 2 -- the whole row pass in one kernel (dct_pair_derived.hip): no operand planes
    if (level2 && f64) {
        DerivedFusedClass fc[9];
        for (unsigned c = 0; c < plan.n_classes; ++c)
            fc[c] = {(const double*)(gathered + goff[c]), cs[c].x2 ? (const double*)(gathered + goff2[c]) : nullptr, (unsigned)pn1[c],
                     (unsigned)(pn2[c] < 0 ? 0 : pn2[c]), plan.c[c].cap, plan.c[c].off, cs[c].x2 != nullptr};
        // (a single frame is 135 blocks of 16 lines for 256 CUs: the merged launches below are 35 us faster there)
        if (lines > (size_t)tuning(TUNE_MERGE_MAX_LINES) && dct_pair_derived_fused_ok(w, plan.n_classes, fc)) {
            const unsigned ncl = plan.n_classes;
            std::array<DerivedFusedClass, 9> fca;
            for (unsigned c = 0; c < 9; ++c) fca[c] = fc[c < ncl ? c : 0];
            ch.push_back({true, [=](hipStream_t st) -> int {
                untimed_work(ctx);
                SSW_TRY(launch_prune_build(st, idx, n, k, plan, flag, rows, pos, info));
                return launch_prune_gather_bases(st, gather_jobs(true));
            }});
            const double in_bytes = px * 3.0 * (double)pix_bytes(u8);
            // (timed with the RGB pre-passes: an HBM-bound kernel -- frames in, compact plane out -- whose 0.1e12 flop ride along;
            // bench.py's GEMM family stays "every pair_gemm_f64_kernel launch")
            ch.push_back({false, [=](hipStream_t st) -> int {
                StageTimer t(ctx, SSW_STAGE_RGB_TO_YIQ, st, in_bytes + (double)lines * (double)cap * 4.0);
                return launch_dct_pair_derived_fused(st, pix_src_kind(u8), rgb, lines, w, (const double*)rot, (const double*)rot2, (const double*)rot3,
                                                     ncl, fca.data(), t_compact, (unsigned)cap);
            }});
            Xform xc{SSW_DCT2, precision, n, cap, h, (float*)ws.compact[1].p, t_compact};
            xc.natural_order = true;
            return build_pass(ctx, ws, xc, false, false, t_compact, (float*)ws.compact[1].p, Epilogue{1.f, 1.f}, ch);
        }
    }
    // the set of columns, then Reader::derived's colour conversion + operand pre-pass (same kernels as the full path)
    ch.push_back({true, [=](hipStream_t st) -> int {
        SSW_TRY(launch_prune_build(st, idx, n, k, plan, flag, rows, pos, info));
        untimed_work(ctx);
        StageTimer t(ctx, SSW_STAGE_RGB_TO_YIQ, st, prep_bytes);
        if (deep) return launch_dct_pair_prep16_rows(st, pix_src_kind(u8), rgb, n, w, h, sp, (const double*)rot, (const double*)rot2,
                                                     (const double*)rot3, nullptr, nullptr);
        if (levels == 3) SSW_TRY(launch_dct_pair_prep8_rows(st, f64, pix_src_kind(u8), rgb, n, w, h, o2, o3, o0, o1, nullptr, nullptr));
        else SSW_TRY(launch_dct_pair_prep4_rows_rgb(st, f64, u8, rgb, n, w, h, o2, o3, o1, nullptr, nullptr));
        return sp ? launch_dct_pair_rotate(st, (const double*)o1, (const double*)rot, sp, lines, w) : SSW_OK;
    }});
    ch.back().tag = 2;
    ch.push_back({false, [=](hipStream_t st) -> int {
        SSW_TRY(launch_prune_gather_bases(st, gather_jobs(false)));
        untimed_work(ctx);
        StageTimer t(ctx, SSW_STAGE_DCT_ROW, st, flop);
        t.traffic(px * (double)esz + (double)lines * (double)cap * 4.0);      // every operand plane once in, the compact plane out
        if (f64 && lines <= (size_t)tuning(TUNE_MERGE_MAX_LINES)) {          // a single frame: the classes side by side in one launch per kind
            PairSubsetClass sc[9];
            for (unsigned c = 0; c < plan.n_classes; ++c)
                sc[c] = {(const double*)cs[c].x, (const double*)cs[c].x2, (const double*)(gathered + goff[c]),
                         cs[c].x2 ? (const double*)(gathered + goff2[c]) : nullptr, plan.c[c].cap, (unsigned)cs[c].kp, plan.c[c].off};
            return launch_dct_pair_gemm_rows_subset_merged_f64(st, sc, plan.n_classes, t_compact, (unsigned)cap, lines);
        }
        for (unsigned c = 0; c < plan.n_classes; ++c) {
            if (cs[c].x2) SSW_TRY(launch_dct_pair_gemm_rows_subset_split_f64(st, (const double*)cs[c].x, (const double*)cs[c].x2, (const double*)(gathered + goff[c]),
                                                                             (const double*)(gathered + goff2[c]), plan.c[c].cap, (unsigned)cs[c].kp, t_compact,
                                                                             (unsigned)cap, plan.c[c].off, lines));
            else if (f64) SSW_TRY(launch_dct_pair_gemm_rows_subset_f64(st, (const double*)cs[c].x, (const double*)(gathered + goff[c]), plan.c[c].cap,
                                                                       (unsigned)cs[c].kp, t_compact, (unsigned)cap, plan.c[c].off, lines));
            else     SSW_TRY(launch_dct_pair_gemm_rows_subset_f32(st, (const float*)cs[c].x, (const float*)(gathered + goff[c]), plan.c[c].cap,
                                                                  (unsigned)cs[c].kp, t_compact, (unsigned)cap, plan.c[c].off, lines));
        }
        return SSW_OK;
    }});
    // column pass on the compact plane: the second pass of the same transform, `cap` columns wide
    Xform xc{SSW_DCT2, precision, n, cap, h, (float*)ws.compact[1].p, t_compact};
    xc.natural_order = true;                 // the compact plane's columns are the gathered frequencies, in the plan's order
    return build_pass(ctx, ws, xc, false, false, t_compact, (float*)ws.compact[1].p, Epilogue{1.f, 1.f}, ch);
}

}  // namespace

// ---- Writer::new + Writer::mark, batched ------------------------------------------------------------------
int batch_embed_impl(ssw_ctx* ctx, const ssw_config* cfg, const void* dev_rgb, int u8_in, size_t n_frames, size_t w,
                     size_t h, const float* dev_marks, size_t k, void* dev_rgb_out, bool u8_out, float* dev_coef_out,
                     uint32_t* dev_indices_out) {
    if (!ctx || !dev_rgb || !dev_marks || !dev_rgb_out) return SSW_ERR_BAD_ARG;
    SSW_TRY(check_config(cfg));
    if (w == 0 || h == 0) return SSW_ERR_BAD_DIMS;
    const size_t plane = w * h;
    const size_t k_eff = std::min(k, plane - 1);                       // zip() truncation, :396: a longer mark is cut silently
    CtxGuard g(ctx);
    const size_t chunk = effective_chunk(ctx, w, h, n_frames);
    const size_t n_chunks = (n_frames + chunk - 1) / chunk;
    const ssw_config c = *cfg;
    const size_t in_px = 3 * pix_bytes(u8_in), out_px = u8_out ? 3 : 3 * sizeof(float);
    auto build = [&](size_t ci, ssw_ctx::Lane& ws, Chain& ch) -> int {
        const size_t f0 = ci * chunk, n = std::min(chunk, n_frames - f0);
        for (int p = 0; p < 4; ++p) SSW_TRY(grow(ws.plane[p], chunk * plane * sizeof(float)));
        SSW_TRY(grow(ws.idx, chunk * std::max<size_t>(k_eff, 1) * sizeof(uint32_t)));
        float* y = (float*)ws.plane[0].p;
        float* pi = (float*)ws.plane[1].p;
        float* pq = (float*)ws.plane[2].p;
        float* tmp = (float*)ws.plane[3].p;
        const char* rgb = static_cast<const char*>(dev_rgb) + f0 * plane * in_px;
        char* out = static_cast<char*>(dev_rgb_out) + f0 * plane * out_px;
        uint32_t* idx = dev_indices_out ? dev_indices_out + f0 * k_eff : (uint32_t*)ws.idx.p;
        float* coef_out = dev_coef_out ? dev_coef_out + f0 * plane : nullptr;
        const float* marks = dev_marks + f0 * k;
        SelectWorkspace* sel = &ws.sel;
        SSW_TRY(build_forward_from_rgb(ctx, ws, c.precision, rgb, u8_in, n, w, h, y, pi, pq, tmp, ch));   // Writer::new :308-313
        ch.push_back({true, [=](hipStream_t st) -> int {
            if (coef_out) { SSW_HIP_CHECK(hipMemcpyAsync(coef_out, y, n * plane * sizeof(float), hipMemcpyDeviceToDevice, st)); untimed_work(ctx); }
            if (k_eff == 0) return SSW_OK;
            SSW_TRY(topk(ctx, st, *sel, y, n, w, h, c.ordering, k_eff, idx));                              // :314 (first k only)
            StageTimer t(ctx, SSW_STAGE_EMBED, st);                                                       // :356
            return launch_embed(st, y, n, plane, idx, k_eff, marks, nullptr, nullptr, 1, k_eff, k, c.method, c.alpha);
        }});
        Xform inv{SSW_DCT3, c.precision, n, w, h, y, tmp};                                                // :368-374
        inv.iq_i = pi; inv.iq_q = pq; inv.rgb_out = out; inv.rgb_out_u8 = u8_out;                        // + :377 in the last pass
        bool fused_rgb = false;
        SSW_TRY(build_transform(ctx, ws, inv, ch, &fused_rgb));
        if (fused_rgb) return SSW_OK;
        const double out_bytes = (double)n * plane * (12.0 + (u8_out ? 3.0 : 12.0));
        ch.push_back({true, [=](hipStream_t st) -> int {
            StageTimer t(ctx, SSW_STAGE_YIQ_TO_RGB, st, out_bytes);                                       // :377 (+ into_rgb8)
            if (u8_out) return launch_yiq_to_rgb8(st, y, pi, pq, n * plane, reinterpret_cast<uint8_t*>(out));
            return launch_yiq_to_rgb(st, y, pi, pq, n * plane, reinterpret_cast<float*>(out));
        }});
        return SSW_OK;
    };
    return run_pipeline(ctx, n_chunks, build);
}

// ---- Reader::base + Reader::derived + extract (+ Tester::similarity), batched --------------------------------
int batch_extract_impl(ssw_ctx* ctx, const ssw_config* cfg, const void* dev_base_rgb, const void* dev_derived_rgb, int u8,
                       size_t n_frames, size_t w, size_t h, size_t k, float* dev_extracted, const float* dev_marks,
                       float* dev_sims) {
    if (!ctx || !dev_base_rgb || !dev_derived_rgb || !dev_extracted) return SSW_ERR_BAD_ARG;
    if ((dev_marks == nullptr) != (dev_sims == nullptr)) return SSW_ERR_BAD_ARG;
    SSW_TRY(check_config(cfg));
    if (w == 0 || h == 0) return SSW_ERR_BAD_DIMS;
    const size_t plane = w * h;
    if (k >= plane) return SSW_ERR_K_TOO_LARGE;                        // :553-555
    if (n_frames == 0) return SSW_OK;
    CtxGuard g(ctx);
    const size_t chunk = effective_chunk(ctx, w, h, n_frames);
    const size_t n_chunks = (n_frames + chunk - 1) / chunk;
    const ssw_config c = *cfg;
    const bool f64 = c.precision == SSW_PRECISION_F64;
    const size_t px_bytes = 3 * pix_bytes(u8);
    // planes of lane 0 decide the (alignment-dependent) strategy for all lanes: hipMalloc'd, always 256-byte aligned
    // (lane 1 only when the call will run two lanes: three chunks or more)
    for (int l = 0; l < (pipeline_uses_two_lanes(ctx, n_chunks) ? 2 : 1); ++l)
        for (int p : {0, 2}) SSW_TRY(grow(ctx->lane[l].plane[p], chunk * plane * sizeof(float)));
    // The pruned path ends with one look at the overflow flags on the host; a stream that is being captured into
    // a graph cannot be waited for, so such a call takes the full transform (enqueue-only, no host round trip).
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &capture) != hipSuccess) { (void)hipGetLastError(); capture = hipStreamCaptureStatusNone; }
    const bool allow_prune = capture == hipStreamCaptureStatusNone;
    const PruneSetup ps = !allow_prune ? PruneSetup() : make_prune_setup(ctx, f64, std::min(chunk, n_frames), w, h, k, (const float*)ctx->lane[0].plane[0].p,
                                           (const float*)ctx->lane[0].plane[2].p, dev_derived_rgb, u8);
    if (ps.on) SSW_TRY(grow(ctx->overflow, n_chunks * SSW_PRUNE_INFO * sizeof(uint32_t)));
    uint32_t* overflow = (uint32_t*)ctx->overflow.p;

    // extract + similarity of one chunk from full coefficient planes (the un-pruned path and the redo)
    auto full_derived = [&](ssw_ctx::Lane& ws, size_t f0, size_t n, float* yb, float* tmp, uint32_t* idx, Chain& ch) -> int {
        SSW_TRY(grow(ws.plane[1], chunk * plane * sizeof(float)));
        float* yd = (float*)ws.plane[1].p;
        const char* drgb = static_cast<const char*>(dev_derived_rgb) + f0 * plane * px_bytes;
        SSW_TRY(build_forward_from_rgb(ctx, ws, c.precision, drgb, u8, n, w, h, yd, nullptr, nullptr, tmp, ch));   // Reader::derived
        float* ext = dev_extracted + f0 * k;
        const float* marks = dev_marks ? dev_marks + f0 * k : nullptr;
        float* sims = dev_sims ? dev_sims + f0 : nullptr;
        ch.push_back({true, [=](hipStream_t st) -> int {
            if (k > 0) {
                StageTimer t(ctx, SSW_STAGE_EXTRACT, st);                           // :529-539
                SSW_TRY(launch_extract(st, yb, yd, n, plane, idx, k, c.method, c.alpha, ext));
            }
            if (marks) {
                StageTimer t(ctx, SSW_STAGE_SIMILARITY, st);                        // :696-714
                SSW_TRY(launch_similarity(st, ext, marks, n, k, sims));
            }
            return SSW_OK;
        }});
        return SSW_OK;
    };
    auto build_chunk = [&](size_t ci, ssw_ctx::Lane& ws, Chain& ch, bool pruned) -> int {
        const size_t f0 = ci * chunk, n = std::min(chunk, n_frames - f0);
        for (int p : {0, 2}) SSW_TRY(grow(ws.plane[p], chunk * plane * sizeof(float)));
        SSW_TRY(grow(ws.idx, chunk * std::max<size_t>(k, 1) * sizeof(uint32_t)));
        float* yb = (float*)ws.plane[0].p;
        float* tmp = (float*)ws.plane[2].p;
        uint32_t* idx = (uint32_t*)ws.idx.p;
        const char* brgb = static_cast<const char*>(dev_base_rgb) + f0 * plane * px_bytes;
        SelectWorkspace* sel = &ws.sel;
        // Reader::base (:474-480): only the Y plane is ever used by a reader
        SSW_TRY(build_forward_from_rgb(ctx, ws, c.precision, brgb, u8, n, w, h, yb, nullptr, nullptr, tmp, ch));
        if (k > 0)
            ch.push_back({true, [=](hipStream_t st) -> int { return topk(ctx, st, *sel, yb, n, w, h, c.ordering, k, idx); }});   // :493
        if (!pruned) return full_derived(ws, f0, n, yb, tmp, idx, ch);
        const char* drgb = static_cast<const char*>(dev_derived_rgb) + f0 * plane * px_bytes;
        SSW_TRY(build_pruned_derived(ctx, ws, c.precision, drgb, u8, n, w, h, k, idx, ps, overflow + ci * SSW_PRUNE_INFO, ch));
        const float* compact = (const float*)ws.compact[1].p;
        const uint32_t* pos = (const uint32_t*)ws.prune_u32.p + w;
        const size_t cap = ps.plan.cap_total;
        float* ext = dev_extracted + f0 * k;
        const float* marks = dev_marks ? dev_marks + f0 * k : nullptr;
        float* sims = dev_sims ? dev_sims + f0 : nullptr;
        ch.push_back({true, [=](hipStream_t st) -> int {
            {
                StageTimer t(ctx, SSW_STAGE_EXTRACT, st);                           // :529-539, derived values from the compact plane
                SSW_TRY(launch_extract_pruned(st, yb, compact, n, w, h, cap, pos, idx, k, c.method, c.alpha, ext));
            }
            if (marks) {
                StageTimer t(ctx, SSW_STAGE_SIMILARITY, st);                        // :696-714
                SSW_TRY(launch_similarity(st, ext, marks, n, k, sims));
            }
            return SSW_OK;
        }});
        return SSW_OK;
    };
    SSW_TRY(run_pipeline(ctx, n_chunks, [&](size_t ci, ssw_ctx::Lane& ws, Chain& ch) { return build_chunk(ci, ws, ch, ps.on); }));
    if (!ps.on) return SSW_OK;
    // Chunks whose column set did not fit the compact plane are redone with the full transform.  This is
    // the one place a batch call waits for the device.
    std::vector<uint32_t> info(n_chunks * SSW_PRUNE_INFO);
    SSW_HIP_CHECK(hipMemcpyAsync(info.data(), overflow, info.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    SSW_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (size_t ci = 0; ci < n_chunks; ++ci) {
        ctx->pruned_chunks++;
        for (unsigned q = 0; q < ps.plan.n_classes; ++q) ctx->pruned_columns += info[ci * SSW_PRUNE_INFO + 1 + q];
        if (info[ci * SSW_PRUNE_INFO] == 0) continue;
        ctx->redone_chunks++;
        Chain ch;
        SSW_TRY(build_chunk(ci, ctx->lane[0], ch, false));
        SSW_TRY(run_serial(ch, ctx->stream));
    }
    return SSW_OK;
}

// ---- Reader::extract against ONE derived frame that has not been transformed yet (single-image handles) ------
// Reader::derived of the reference transforms the whole frame because it cannot know which coefficients will be
// read (:469-480); a derived handle therefore only uploads its frame, and the transform happens here, when the base
// reader's index list is known: the pruned transform of the batch path with n = 1 (same kernels, bit-identical
// values).  Enqueues on the context's stream: the k extracted values into dev_out, the overflow flag into
// dev_info[0] (non-zero: the columns did not fit, the caller must transform fully).  *applicable = false (nothing
// enqueued) when the shape or the settings do not take the pruned path.
int extract_single_pruned(ssw_ctx* ctx, int precision, const void* derived_rgb, int u8, size_t w, size_t h, const float* base_y,
                          const uint32_t* idx, size_t k, int method, float alpha, float* dev_out, uint32_t** dev_info,
                          bool* applicable) {
    *applicable = false;
    const size_t plane = w * h;
    ssw_ctx::Lane& ws = ctx->lane[0];
    for (int p : {0, 2}) SSW_TRY(grow(ws.plane[p], plane * sizeof(float)));
    const PruneSetup ps = make_prune_setup(ctx, precision == SSW_PRECISION_F64, 1, w, h, k, (const float*)ws.plane[0].p,
                                           (const float*)ws.plane[2].p, derived_rgb, u8);
    if (!ps.on) return SSW_OK;
    SSW_TRY(grow(ctx->overflow, SSW_PRUNE_INFO * sizeof(uint32_t)));
    uint32_t* info = (uint32_t*)ctx->overflow.p;
    Chain ch;
    SSW_TRY(build_pruned_derived(ctx, ws, precision, derived_rgb, u8, 1, w, h, k, idx, ps, info, ch));
    SSW_TRY(run_serial(ch, ctx->stream));
    {
        StageTimer t(ctx, SSW_STAGE_EXTRACT, ctx->stream);
        SSW_TRY(launch_extract_pruned(ctx->stream, base_y, (const float*)ws.compact[1].p, 1, w, h, ps.plan.cap_total,
                                      (const uint32_t*)ws.prune_u32.p + w, idx, k, method, alpha, dev_out));
    }
    *dev_info = info;
    *applicable = true;
    return SSW_OK;
}

}  // namespace host
}  // namespace ssw
