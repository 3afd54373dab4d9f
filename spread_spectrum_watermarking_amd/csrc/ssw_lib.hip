// C ABI of libssw_hip.so: context, workspace, stage timers and the orchestration of the
// reference's call stacks (Writer::new / mark, Reader::base / derived / extract, Tester).
// See include/ssw.h for the reference file:line each entry point replaces.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

#include "ssw_internal.hpp"

namespace ssw {
static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }
}  // namespace ssw

using namespace ssw;

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
};
struct ssw_reader {
    ssw_ctx* ctx;
    size_t w, h;
    bool is_base;
    ssw_config cfg;
    float* y = nullptr;                               // coefficients
    uint32_t* idx = nullptr;                          // cached first idx_k indices
    size_t idx_k = 0;
};

// ---- small helpers --------------------------------------------------------------------------
namespace {

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

// Device allocation: a failing hipMalloc is reported as SSW_ERR_OUT_OF_MEMORY whatever code the runtime
// chose for it, and the runtime's sticky error is cleared so that the next call starts clean.
int dev_malloc(void** p, size_t bytes) {
    *p = nullptr;
    const hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e == hipSuccess) return SSW_OK;
    (void)hipGetLastError();
    *p = nullptr;
    set_last_error(std::string("hipMalloc(") + std::to_string(bytes) + " bytes): " + hipGetErrorString(e));
    return SSW_ERR_OUT_OF_MEMORY;
}
#define SSW_ALLOC(pp, bytes) SSW_TRY(dev_malloc((void**)(pp), (bytes)))

int grow(ssw_ctx::Buf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return SSW_OK;
    if (b.p) { SSW_HIP_CHECK(hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    SSW_ALLOC(&b.p, bytes);
    b.bytes = bytes ? bytes : 16;
    return SSW_OK;
}

int grow_select(ssw_ctx* ctx, size_t frames, size_t k) {
    SelectWorkspace& s = ctx->sel;
    const size_t want = select_cand_capacity(k);
    if (s.frames >= frames && s.cap >= want && s.hist) return SSW_OK;
    const size_t nf = std::max(frames, s.frames), cap = std::max(want, s.cap);
    if (s.hist) (void)hipFree(s.hist);
    if (s.ctrl) (void)hipFree(s.ctrl);
    if (s.cand) (void)hipFree(s.cand);
    s = SelectWorkspace();
    SSW_ALLOC(&s.hist, nf * 2048 * sizeof(uint32_t));
    SSW_ALLOC(&s.ctrl, nf * 4 * sizeof(uint32_t));
    SSW_ALLOC(&s.cand, nf * cap * sizeof(uint64_t));
    SSW_HIP_CHECK(hipMemsetAsync(s.hist, 0, nf * 2048 * sizeof(uint32_t), ctx->stream));   // see select.hip:
    SSW_HIP_CHECK(hipMemsetAsync(s.ctrl, 0, nf * 4 * sizeof(uint32_t), ctx->stream));      // zero between uses
    s.frames = nf;
    s.cap = cap;
    return SSW_OK;
}

// Stage timer: records an event pair around a region on the context's stream.
struct StageTimer {
    ssw_ctx* ctx;
    int stage;
    hipEvent_t a = nullptr, b = nullptr;
    StageTimer(ssw_ctx* c, int s) : ctx(c), stage(s) {
        if (!ctx->timing) return;
        auto get = [&]() {
            hipEvent_t e = nullptr;
            if (!ctx->free_events.empty()) { e = ctx->free_events.back(); ctx->free_events.pop_back(); }
            else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
            return e;
        };
        a = get(); b = get();
        if (a) (void)hipEventRecord(a, ctx->stream);
    }
    ~StageTimer() {
        if (!ctx->timing || !a || !b) return;
        (void)hipEventRecord(b, ctx->stream);
        ctx->pending.push_back({stage, a, b});
    }
};

int flush_timers(ssw_ctx* ctx) {
    for (auto& p : ctx->pending) {
        float ms = 0.f;
        SSW_HIP_CHECK(hipEventSynchronize(p.b));
        SSW_HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
        ctx->stage_ms[p.stage] += ms;
        ctx->stage_launches[p.stage] += 1;
        ctx->free_events.push_back(p.a);
        ctx->free_events.push_back(p.b);
    }
    ctx->pending.clear();
    return SSW_OK;
}

int get_basis(ssw_ctx* ctx, size_t n, bool inverse, bool f64, int kind, const void** out) {
    auto key = std::make_tuple(n, inverse, f64, kind);
    auto it = ctx->basis.find(key);
    if (it != ctx->basis.end()) { *out = it->second; return SSW_OK; }
    void* p = nullptr;
    const size_t elems = kind == 0 ? n * dense_basis_kpad(n) : kind >= 3 ? (n / 2) * dct_pair_kpad(f64, n) : (n / 2) * half_basis_kpad(n);
    SSW_ALLOC(&p, std::max<size_t>(elems, 1) * (f64 ? sizeof(double) : sizeof(float)));
    int rc = kind >= 3 ? launch_make_half_basis_blocked(ctx->stream, f64, n, inverse, kind - 3, p)
             : kind != 0 ? (f64 ? launch_make_half_basis_f64(ctx->stream, n, inverse, kind - 1, (double*)p)
                              : launch_make_half_basis_f32(ctx->stream, n, inverse, kind - 1, (float*)p))
             : f64     ? launch_make_basis_f64(ctx->stream, n, inverse, (double*)p)
                       : launch_make_basis_f32(ctx->stream, n, inverse, (float*)p);
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

// Frames per internal pass: the caller's setting, or (0 = automatic, the default) about 2^28 pixels -- 32 4K
// frames, 129 full-HD ones: the GEMM grids then run ~16 rounds of blocks (a 16-frame pass of 1080p frames
// only 2.1, 11 % slower) for 36 B/px of workspace (4K: 9.6 GB).
size_t effective_chunk(const ssw_ctx* ctx, size_t w, size_t h, size_t n_frames) {
    size_t c = ctx->chunk_frames;
    if (c == 0) c = std::max<size_t>(1, ((size_t)1 << 28) / std::max<size_t>(w * h, 1));
    return std::min(c, std::max<size_t>(n_frames, 1));
}

// dct2d::dct2_2d on n contiguous planes, `data` in place, `tmp` same size scratch.
// `rgb` (optional; forward transforms only, see forward_from_rgb below): the frames `data` would have been
// converted from -- the first pass then reads them directly and `data` is only written by the last pass.
int dct2d_planes(ssw_ctx* ctx, int type, int precision, size_t n, size_t w, size_t h, float* data, float* tmp,
                 const void* rgb = nullptr, bool rgb_u8 = false, float* iq_i = nullptr, float* iq_q = nullptr) {
    const bool inverse = (type == SSW_DCT3);
    const bool f64 = (precision == SSW_PRECISION_F64);
    const bool rows_first = (w >= h);                                  // src/dct2d.rs:93-98
    // The operand-ready GEMMs walk an operand plane with 32-bit scalar offsets: keep each call's planes
    // below 4 GB by transforming the frames in groups (frames are independent).
    if (ctx->fold && ctx->fold_level >= 3 && n > 1) {
        const size_t per_frame = dct_pair_operand_elems(f64, 1, w, h) * (f64 ? sizeof(double) : sizeof(float));
        const size_t max_frames = per_frame ? 0xFFFFFFFFull / per_frame : 0;
        if (max_frames >= 1 && n > max_frames) {
            for (size_t f0 = 0; f0 < n; f0 += max_frames)
                SSW_TRY(dct2d_planes(ctx, type, precision, std::min(max_frames, n - f0), w, h, data + f0 * w * h, tmp + f0 * w * h,
                                     rgb ? static_cast<const char*>(rgb) + f0 * w * h * 3 * (rgb_u8 ? 1 : sizeof(float)) : nullptr,
                                     rgb_u8, iq_i ? iq_i + f0 * w * h : nullptr, iq_q ? iq_q + f0 * w * h : nullptr));
            return SSW_OK;
        }
    }
    Epilogue plain{1.f, 1.f};
    auto ortho = [&](size_t len) {                                      // src/dct2d.rs:154-155
        Epilogue e{std::sqrt(1.0f / (4.0f * (float)len)), std::sqrt(1.0f / (2.0f * (float)len))};
        return e;
    };
    Epilogue last = plain;
    if (type == SSW_DCT3) last.first = last.base = (float)4 / (float)(w * h);           // :213-217
    for (int pass = 0; pass < 2; ++pass) {
        const bool is_row = (pass == 0) ? rows_first : !rows_first;
        const float* src = (pass == 0) ? data : tmp;
        float* dst = (pass == 0) ? tmp : data;
        Epilogue ep = (type == SSW_DCT2_ORTHOGONAL) ? ortho(is_row ? w : h) : (pass == 1 ? last : plain);
        const bool fold = ctx->fold && (is_row ? dct_rows_can_fold(w, src, dst) : dct_cols_can_fold(w, h, src, dst));
        const size_t len = is_row ? w : h;
        const void *b0 = nullptr, *b1 = nullptr;
        const bool operand = fold && ctx->fold_level >= 3 && dct_pair_can_run(f64, n, w, h, src, dst);
        if (operand) {
            SSW_TRY(get_basis(ctx, len, inverse, f64, 3, &b0));        // k-blocked half bases
            SSW_TRY(get_basis(ctx, len, inverse, f64, 4, &b1));
        } else if (fold) {
            SSW_TRY(get_basis(ctx, len, inverse, f64, 1, &b0));
            SSW_TRY(get_basis(ctx, len, inverse, f64, 2, &b1));
        } else {
            SSW_TRY(get_basis(ctx, len, inverse, f64, 0, &b0));
        }
        if (rgb && pass == 0 && !(operand && is_row && ctx->fold_level >= 4 && dct_pair_can_fold2(len)))
            return SSW_ERR_BAD_ARG;                                    // forward_from_rgb() checks the same conditions
        if (operand) {
            const size_t esz = f64 ? sizeof(double) : sizeof(float);
            const size_t bytes = dct_pair_operand_elems(f64, n, w, h) * esz;
            const bool two = ctx->fold_level >= 4 && (is_row ? dct_pair_can_fold2(len) : dct_pair_can_fold2_cols(len));
            const int st_pass = is_row ? SSW_STAGE_DCT_ROW : SSW_STAGE_DCT_COL;
            const int st_main = is_row ? SSW_STAGE_DCT_ROW_MAIN : SSW_STAGE_DCT_COL_MAIN;
            auto gemm = [&](int kind, const void* x1, const void* x2, const void* y1, const void* y2, void* tmpE, int sub = 0) {
                return f64 ? launch_dct_pair_gemm_f64(ctx->stream, is_row, inverse, kind, sub, (const double*)x1, (const double*)x2,
                                                      (const double*)y1, (const double*)y2, dst, (double*)tmpE, n, w, h, ep)
                           : launch_dct_pair_gemm_f32(ctx->stream, is_row, inverse, kind, sub, (const float*)x1, (const float*)x2,
                                                      (const float*)y1, (const float*)y2, dst, (float*)tmpE, n, w, h, ep);
            };
            // a third level pays once the sums are long enough (4K: +1.6 %, 1080p: -3 %); level 6 forces it
            const bool three = two && !inverse && is_row && dct_pair_can_fold3(len) &&
                               (ctx->fold_level >= 6 || (ctx->fold_level == 5 && len >= 3072));
            if (three) {
                // forward row pass, three levels: x- (odd frequencies), S- (2 mod 4), (SSS, SS-) (0 and 4 mod 8)
                for (int b = 0; b < 4; ++b) SSW_TRY(grow(ctx->operand[b], bytes));
                void* d1 = ctx->operand[1].p;
                void* d2 = ctx->operand[0].p;
                void* r1 = ctx->operand[2].p;
                void* r2 = ctx->operand[3].p;
                const void *h1 = nullptr, *e0 = nullptr, *e1 = nullptr;
                SSW_TRY(get_basis(ctx, len / 2, false, f64, 4, &h1));          // odd half basis of len/2
                SSW_TRY(get_basis(ctx, len / 4, false, f64, 3, &e0));          // half bases of len/4
                SSW_TRY(get_basis(ctx, len / 4, false, f64, 4, &e1));
                {
                    StageTimer t(ctx, rgb && pass == 0 ? SSW_STAGE_RGB_TO_YIQ : SSW_STAGE_DCT_PREP);
                    const bool from_rgb = rgb && pass == 0;
                    SSW_TRY(launch_dct_pair_prep8_rows(ctx->stream, f64, from_rgb ? (rgb_u8 ? 2 : 1) : 0, from_rgb ? rgb : (const void*)src,
                                                       n, w, h, r1, r2, d2, d1, from_rgb ? iq_i : nullptr, from_rgb ? iq_q : nullptr));
                }
                StageTimer t(ctx, st_pass);
                SSW_TRY(gemm(1, r1, r2, e0, e1, nullptr, 1));
                SSW_TRY(gemm(2, d2, d2, h1, (const char*)h1 + (len / 8) * 64, nullptr, 1));
                StageTimer tm(ctx, st_main);
                SSW_TRY(gemm(2, d1, d1, b1, (const char*)b1 + (len / 4) * 64, nullptr, 0));
            } else if (!two) {
                for (int b = 0; b < 2; ++b) SSW_TRY(grow(ctx->operand[b], bytes));
                void* x1 = ctx->operand[0].p;
                void* x2 = ctx->operand[1].p;
                {
                    StageTimer t(ctx, SSW_STAGE_DCT_PREP);
                    SSW_TRY(launch_dct_pair_prep(ctx->stream, f64, is_row, inverse, src, n, w, h, x1, x2));
                }
                StageTimer t(ctx, st_pass);
                StageTimer tm(ctx, st_main);
                SSW_TRY(gemm(0, x1, x2, b0, b1, nullptr));
            } else {
                for (int b = 1; b < (inverse ? 5 : 4); ++b) SSW_TRY(grow(ctx->operand[b], bytes));
                void* x2 = ctx->operand[1].p;       // D | O
                void* xx1 = ctx->operand[2].p;      // SS | EE
                void* xx2 = ctx->operand[3].p;      // SD | EO
                void* tmpE = ctx->operand[4].p;     // inverse: the even half E, unrounded
                const void *q0 = nullptr, *q1 = nullptr;
                SSW_TRY(get_basis(ctx, len / 2, inverse, f64, 3, &q0));
                SSW_TRY(get_basis(ctx, len / 2, inverse, f64, 4, &q1));
                if (rgb && pass == 0) {                      // eligibility was checked by forward_from_rgb()
                    StageTimer t(ctx, SSW_STAGE_RGB_TO_YIQ);
                    SSW_TRY(launch_dct_pair_prep4_rows_rgb(ctx->stream, f64, rgb_u8, rgb, n, w, h, xx1, xx2, x2, iq_i, iq_q));
                } else {
                    StageTimer t(ctx, SSW_STAGE_DCT_PREP);
                    SSW_TRY(launch_dct_pair_prep4(ctx->stream, f64, is_row, inverse, src, n, w, h, xx1, xx2, x2));
                }
                StageTimer t(ctx, st_pass);
                // even half: a half-length transform of S (forward) / of the even coefficients (inverse), folded again
                SSW_TRY(gemm(1, xx1, xx2, q0, q1, tmpE));
                // odd half: full half-length sum, the odd basis split into two row blocks (second block:
                // len/4 lines further inside every k-block of the same plane = 64 bytes per line)
                const char* bo2 = (const char*)b1 + (len / 4) * 64;
                StageTimer tm(ctx, st_main);
                SSW_TRY(gemm(2, x2, x2, b1, bo2, tmpE));
            }
        } else if (is_row) {
            StageTimer t(ctx, SSW_STAGE_DCT_ROW);
            if (fold && f64) SSW_TRY(launch_dct_rows_folded_f64(ctx->stream, inverse, src, dst, n * h, w, (const double*)b0, (const double*)b1, ep));
            else if (fold)   SSW_TRY(launch_dct_rows_folded_f32(ctx->stream, inverse, src, dst, n * h, w, (const float*)b0, (const float*)b1, ep));
            else             SSW_TRY(launch_dct_rows(ctx->stream, precision, src, dst, n * h, w, b0, ep));
        } else {
            StageTimer t(ctx, SSW_STAGE_DCT_COL);
            if (fold && f64) SSW_TRY(launch_dct_cols_folded_f64(ctx->stream, inverse, src, dst, n, w, h, (const double*)b0, (const double*)b1, ep));
            else if (fold)   SSW_TRY(launch_dct_cols_folded_f32(ctx->stream, inverse, src, dst, n, w, h, (const float*)b0, (const float*)b1, ep));
            else             SSW_TRY(launch_dct_cols(ctx->stream, precision, src, dst, n, w, h, b0, ep));
        }
    }
    return SSW_OK;
}

int topk(ssw_ctx* ctx, const float* coef, size_t n, size_t w, size_t h, int ordering, size_t k, uint32_t* idx) {
    if (k > select_max_k()) {
        // beyond the in-LDS top-k limit: full device sort of each plane, first k entries kept
        size_t bytes = 0;
        SSW_TRY(full_sort_scratch_bytes(w * h, &bytes));
        SSW_TRY(grow(ctx->sort_scratch, bytes));
        StageTimer t(ctx, SSW_STAGE_SELECT);
        for (size_t f = 0; f < n; ++f)
            SSW_TRY(launch_full_sort(ctx->stream, coef + f * w * h, w, h, ordering, ctx->sort_scratch.p,
                                     ctx->sort_scratch.bytes, idx + f * k, k));
        return SSW_OK;
    }
    SSW_TRY(grow_select(ctx, n, k));
    StageTimer t(ctx, SSW_STAGE_SELECT);
    return launch_topk(ctx->stream, coef, n, w, h, ordering, k, ctx->sel, idx);
}

}  // namespace

// ---- library / context ----------------------------------------------------------------------
extern "C" {

const char* ssw_version(void) { return "ssw-hip 0.1.0 (gfx950)"; }

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
    SSW_HIP_CHECK(hipSetDevice(device_id));
    ssw_ctx* ctx = new (std::nothrow) ssw_ctx();
    if (!ctx) return SSW_ERR_OUT_OF_MEMORY;
    ctx->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; set_last_error(hipGetErrorString(e)); return SSW_ERR_HIP; }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return SSW_OK;
}

int ssw_ctx_destroy(ssw_ctx* ctx) {
    if (!ctx) return SSW_OK;
    DeviceGuard g(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->basis) (void)hipFree(kv.second);
    for (auto& b : ctx->plane) if (b.p) (void)hipFree(b.p);
    for (auto& b : ctx->operand) if (b.p) (void)hipFree(b.p);
    if (ctx->idx.p) (void)hipFree(ctx->idx.p);
    if (ctx->small.p) (void)hipFree(ctx->small.p);
    if (ctx->sort_scratch.p) (void)hipFree(ctx->sort_scratch.p);
    if (ctx->resize_tmp.p) (void)hipFree(ctx->resize_tmp.p);
    for (auto& kv : ctx->taps) { (void)hipFree(kv.second.left); (void)hipFree(kv.second.count); (void)hipFree(kv.second.weights); }
    if (ctx->sel.hist) (void)hipFree(ctx->sel.hist);
    if (ctx->sel.ctrl) (void)hipFree(ctx->sel.ctrl);
    if (ctx->sel.cand) (void)hipFree(ctx->sel.cand);
    for (auto& p : ctx->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto& e : ctx->free_events) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return SSW_OK;
}

int ssw_ctx_synchronize(ssw_ctx* ctx) {
    if (!ctx) return SSW_ERR_BAD_ARG;
    DeviceGuard g(ctx->device);
    SSW_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SSW_OK;
}

void* ssw_ctx_stream(ssw_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int ssw_ctx_set_stream(ssw_ctx* ctx, void* hip_stream) {
    if (!ctx) return SSW_ERR_BAD_ARG;
    DeviceGuard g(ctx->device);
    SSW_TRY(flush_timers(ctx));                                      // pending event pairs belong to the old stream
    SSW_HIP_CHECK(hipStreamSynchronize(ctx->stream));                // workspace in flight on the old stream
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return SSW_OK;
}

int ssw_ctx_wait_event(ssw_ctx* ctx, void* hip_event) {
    if (!ctx || !hip_event) return SSW_ERR_BAD_ARG;
    DeviceGuard g(ctx->device);
    SSW_HIP_CHECK(hipStreamWaitEvent(ctx->stream, (hipEvent_t)hip_event, 0));
    return SSW_OK;
}

int ssw_ctx_record_event(ssw_ctx* ctx, void* hip_event) {
    if (!ctx || !hip_event) return SSW_ERR_BAD_ARG;
    DeviceGuard g(ctx->device);
    SSW_HIP_CHECK(hipEventRecord((hipEvent_t)hip_event, ctx->stream));
    return SSW_OK;
}

int ssw_ctx_set_chunk_frames(ssw_ctx* ctx, size_t frames) {
    if (!ctx) return SSW_ERR_BAD_ARG;
    ctx->chunk_frames = frames;                    // 0 = automatic
    return SSW_OK;
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
    DeviceGuard g(ctx->device);
    SSW_TRY(flush_timers(ctx));
    ctx->timing = enable != 0;
    return SSW_OK;
}

int ssw_ctx_reset_timing(ssw_ctx* ctx) {
    if (!ctx) return SSW_ERR_BAD_ARG;
    DeviceGuard g(ctx->device);
    SSW_TRY(flush_timers(ctx));
    for (int s = 0; s < SSW_STAGE_COUNT; ++s) { ctx->stage_ms[s] = 0; ctx->stage_launches[s] = 0; }
    return SSW_OK;
}

int ssw_ctx_get_timing(ssw_ctx* ctx, double* ms, uint64_t* launches) {
    if (!ctx) return SSW_ERR_BAD_ARG;
    DeviceGuard g(ctx->device);
    SSW_TRY(flush_timers(ctx));
    for (int s = 0; s < SSW_STAGE_COUNT; ++s) {
        if (ms) ms[s] = ctx->stage_ms[s];
        if (launches) launches[s] = ctx->stage_launches[s];
    }
    return SSW_OK;
}

int ssw_dev_alloc(ssw_ctx* ctx, size_t bytes, void** dev_ptr) {
    if (!ctx || !dev_ptr) return SSW_ERR_BAD_ARG;
    DeviceGuard g(ctx->device);
    SSW_ALLOC(dev_ptr, bytes);
    return SSW_OK;
}
int ssw_dev_free(ssw_ctx* ctx, void* dev_ptr) {
    if (!ctx) return SSW_ERR_BAD_ARG;
    DeviceGuard g(ctx->device);
    SSW_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (dev_ptr) SSW_HIP_CHECK(hipFree(dev_ptr));
    return SSW_OK;
}
int ssw_copy_to_dev(ssw_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes) {
    if (!ctx || (bytes && (!dev_dst || !host_src))) return SSW_ERR_BAD_ARG;
    DeviceGuard g(ctx->device);
    SSW_HIP_CHECK(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    SSW_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SSW_OK;
}
int ssw_copy_to_host(ssw_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes) {
    if (!ctx || (bytes && (!host_dst || !dev_src))) return SSW_ERR_BAD_ARG;
    DeviceGuard g(ctx->device);
    SSW_HIP_CHECK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SSW_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SSW_OK;
}

// ---- transforms -----------------------------------------------------------------------------
int ssw_rgb_to_yiq(ssw_ctx* ctx, const float* dev_rgb, size_t n_frames, size_t w, size_t h,
                   float* dev_y, float* dev_i, float* dev_q) {
    if (!ctx || !dev_rgb || !dev_y || ((dev_i == nullptr) != (dev_q == nullptr))) return SSW_ERR_BAD_ARG;
    DeviceGuard g(ctx->device);
    StageTimer t(ctx, SSW_STAGE_RGB_TO_YIQ);
    return launch_rgb_to_yiq(ctx->stream, dev_rgb, n_frames * w * h, dev_y, dev_i, dev_q);
}

int ssw_yiq_to_rgb(ssw_ctx* ctx, const float* dev_y, const float* dev_i, const float* dev_q,
                   size_t n_frames, size_t w, size_t h, float* dev_rgb) {
    if (!ctx || !dev_rgb || !dev_y || !dev_i || !dev_q) return SSW_ERR_BAD_ARG;
    DeviceGuard g(ctx->device);
    StageTimer t(ctx, SSW_STAGE_YIQ_TO_RGB);
    return launch_yiq_to_rgb(ctx->stream, dev_y, dev_i, dev_q, n_frames * w * h, dev_rgb);
}

int ssw_dct2d(ssw_ctx* ctx, int dct_type, int precision, size_t n_frames, size_t w, size_t h,
              float* dev_planes) {
    if (!ctx || !dev_planes) return SSW_ERR_BAD_ARG;
    if (dct_type < SSW_DCT2 || dct_type > SSW_DCT3 || !valid_precision(precision)) return SSW_ERR_BAD_ARG;
    if (w == 0 || h == 0) return SSW_ERR_BAD_DIMS;
    DeviceGuard g(ctx->device);
    const size_t chunk = effective_chunk(ctx, w, h, n_frames);
    SSW_TRY(grow(ctx->plane[3], chunk * w * h * sizeof(float)));
    for (size_t f0 = 0; f0 < n_frames; f0 += chunk) {
        const size_t n = std::min(chunk, n_frames - f0);
        SSW_TRY(dct2d_planes(ctx, dct_type, precision, n, w, h, dev_planes + f0 * w * h, (float*)ctx->plane[3].p));
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
    DeviceGuard g(ctx->device);
    return topk(ctx, dev_coef, n_frames, w, h, ordering, k, dev_indices);
}

int ssw_embed_coefficients(ssw_ctx* ctx, float* dev_coef, size_t n_frames, size_t plane_len,
                           const uint32_t* dev_indices, size_t k, int method, float alpha,
                           const float* dev_marks, size_t n_marks) {
    if (!ctx || !dev_coef || !dev_indices || !dev_marks) return SSW_ERR_BAD_ARG;
    if (method == SSW_METHOD_CUSTOM) return SSW_ERR_UNSUPPORTED;
    if (!valid_method(method)) return SSW_ERR_BAD_ARG;
    DeviceGuard g(ctx->device);
    StageTimer t(ctx, SSW_STAGE_EMBED);
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
    DeviceGuard g(ctx->device);
    StageTimer t(ctx, SSW_STAGE_EXTRACT);
    return launch_extract(ctx->stream, dev_base, dev_derived, n_frames, plane_len, dev_indices, k, method,
                          alpha, dev_out);
}

int ssw_similarity_batch(ssw_ctx* ctx, const float* dev_extracted, const float* dev_marks,
                         size_t n_pairs, size_t k, float* dev_sims) {
    if (!ctx || !dev_extracted || !dev_marks || !dev_sims) return SSW_ERR_BAD_ARG;
    DeviceGuard g(ctx->device);
    StageTimer t(ctx, SSW_STAGE_SIMILARITY);
    return launch_similarity(ctx->stream, dev_extracted, dev_marks, n_pairs, k, dev_sims);
}

int ssw_similarity_matrix(ssw_ctx* ctx, const float* dev_extracted, size_t n_extracted, const float* dev_marks,
                          size_t n_marks, size_t k, float* dev_sims) {
    if (!ctx || !dev_extracted || !dev_marks || !dev_sims) return SSW_ERR_BAD_ARG;
    if (n_extracted == 0 || n_marks == 0) return SSW_OK;
    DeviceGuard g(ctx->device);
    SSW_TRY(grow(ctx->small, n_extracted * sizeof(float)));
    StageTimer t(ctx, SSW_STAGE_SIMILARITY);
    SSW_TRY(launch_sim_den(ctx->stream, dev_extracted, n_extracted, k, (float*)ctx->small.p));
    SSW_TRY(launch_gemm_nt_f32(ctx->stream, dev_extracted, n_extracted, dev_marks, n_marks, k, dev_sims));
    return launch_sim_scale(ctx->stream, dev_sims, (const float*)ctx->small.p, n_extracted, n_marks);
}

// ---- whole path, batched --------------------------------------------------------------------
namespace {

int rgb_in_to_yiq(ssw_ctx* ctx, const void* rgb, bool u8, size_t first_px, size_t npix, float* y, float* i, float* q) {
    StageTimer t(ctx, SSW_STAGE_RGB_TO_YIQ);
    if (u8) return launch_rgb8_to_yiq(ctx->stream, static_cast<const uint8_t*>(rgb) + first_px * 3, npix, y, i, q);
    return launch_rgb_to_yiq(ctx->stream, static_cast<const float*>(rgb) + first_px * 3, npix, y, i, q);
}

// Writer::new / Reader::base / Reader::derived: rgb -> Y (+ I, Q) -> forward 2-D DCT of Y into `y`.
// Where the default GEMM strategy applies (rows first, two folding levels on the row axis) the colour
// conversion is fused into the first operand pre-pass and the f32 Y plane is never materialised.
int forward_from_rgb(ssw_ctx* ctx, int precision, const void* rgb, bool u8, size_t first_px, size_t n, size_t w, size_t h,
                     float* y, float* i, float* q, float* tmp) {
    const bool f64 = precision == SSW_PRECISION_F64;
    const char* src = static_cast<const char*>(rgb) + first_px * 3 * (u8 ? 1 : sizeof(float));
    if (ctx->fold && ctx->fold_level >= 4 && dct_pair_can_run(f64, 1, w, h, y, tmp) && dct_pair_can_prep_from_rgb(w, h, src, u8))
        return dct2d_planes(ctx, SSW_DCT2, precision, n, w, h, y, tmp, src, u8, i, q);
    SSW_TRY(rgb_in_to_yiq(ctx, rgb, u8, first_px, n * w * h, y, i, q));
    return dct2d_planes(ctx, SSW_DCT2, precision, n, w, h, y, tmp);
}

int batch_embed_impl(ssw_ctx* ctx, const ssw_config* cfg, const void* dev_rgb, bool u8_in, size_t n_frames,
                     size_t w, size_t h, const float* dev_marks, size_t k, void* dev_rgb_out, bool u8_out,
                     float* dev_coef_out, uint32_t* dev_indices_out) {
    if (!ctx || !dev_rgb || !dev_marks || !dev_rgb_out) return SSW_ERR_BAD_ARG;
    SSW_TRY(check_config(cfg));
    if (w == 0 || h == 0) return SSW_ERR_BAD_DIMS;
    const size_t plane = w * h;
    const size_t k_eff = std::min(k, plane - 1);                       // zip() truncation, :396: a longer mark is cut silently
    DeviceGuard g(ctx->device);
    const size_t chunk = effective_chunk(ctx, w, h, n_frames);
    for (int p = 0; p < 4; ++p) SSW_TRY(grow(ctx->plane[p], chunk * plane * sizeof(float)));
    SSW_TRY(grow(ctx->idx, chunk * std::max<size_t>(k_eff, 1) * sizeof(uint32_t)));
    float* y = (float*)ctx->plane[0].p;
    float* pi = (float*)ctx->plane[1].p;
    float* pq = (float*)ctx->plane[2].p;
    float* tmp = (float*)ctx->plane[3].p;
    for (size_t f0 = 0; f0 < n_frames; f0 += chunk) {
        const size_t n = std::min(chunk, n_frames - f0);
        SSW_TRY(forward_from_rgb(ctx, cfg->precision, dev_rgb, u8_in, f0 * plane, n, w, h, y, pi, pq, tmp));   // Writer::new :308-313
        if (dev_coef_out)
            SSW_HIP_CHECK(hipMemcpyAsync(dev_coef_out + f0 * plane, y, n * plane * sizeof(float),
                                         hipMemcpyDeviceToDevice, ctx->stream));
        uint32_t* idx = dev_indices_out ? dev_indices_out + f0 * k_eff : (uint32_t*)ctx->idx.p;
        if (k_eff > 0) {
            SSW_TRY(topk(ctx, y, n, w, h, cfg->ordering, k_eff, idx));                  // :314 (first k only)
            StageTimer t(ctx, SSW_STAGE_EMBED);                                         // :356
            SSW_TRY(launch_embed(ctx->stream, y, n, plane, idx, k_eff, dev_marks + f0 * k, nullptr, nullptr,
                                 1, k_eff, k, cfg->method, cfg->alpha));
        }
        SSW_TRY(dct2d_planes(ctx, SSW_DCT3, cfg->precision, n, w, h, y, tmp));           // :368-374
        {
            StageTimer t(ctx, SSW_STAGE_YIQ_TO_RGB);                                    // :377 (+ into_rgb8)
            if (u8_out) SSW_TRY(launch_yiq_to_rgb8(ctx->stream, y, pi, pq, n * plane, static_cast<uint8_t*>(dev_rgb_out) + f0 * plane * 3));
            else        SSW_TRY(launch_yiq_to_rgb(ctx->stream, y, pi, pq, n * plane, static_cast<float*>(dev_rgb_out) + f0 * plane * 3));
        }
    }
    return SSW_OK;
}

int batch_extract_impl(ssw_ctx* ctx, const ssw_config* cfg, const void* dev_base_rgb, const void* dev_derived_rgb,
                       bool u8, size_t n_frames, size_t w, size_t h, size_t k, float* dev_extracted,
                       const float* dev_marks, float* dev_sims) {
    if (!ctx || !dev_base_rgb || !dev_derived_rgb || !dev_extracted) return SSW_ERR_BAD_ARG;
    if ((dev_marks == nullptr) != (dev_sims == nullptr)) return SSW_ERR_BAD_ARG;
    SSW_TRY(check_config(cfg));
    if (w == 0 || h == 0) return SSW_ERR_BAD_DIMS;
    const size_t plane = w * h;
    if (k >= plane) return SSW_ERR_K_TOO_LARGE;                        // :553-555
    DeviceGuard g(ctx->device);
    const size_t chunk = effective_chunk(ctx, w, h, n_frames);
    for (int p = 0; p < 3; ++p) SSW_TRY(grow(ctx->plane[p], chunk * plane * sizeof(float)));
    SSW_TRY(grow(ctx->idx, chunk * std::max<size_t>(k, 1) * sizeof(uint32_t)));
    float* yb = (float*)ctx->plane[0].p;
    float* yd = (float*)ctx->plane[1].p;
    float* tmp = (float*)ctx->plane[2].p;
    uint32_t* idx = (uint32_t*)ctx->idx.p;
    for (size_t f0 = 0; f0 < n_frames; f0 += chunk) {
        const size_t n = std::min(chunk, n_frames - f0);
        // Reader::base (:474-480): only the Y plane is ever used by a reader
        SSW_TRY(forward_from_rgb(ctx, cfg->precision, dev_base_rgb, u8, f0 * plane, n, w, h, yb, nullptr, nullptr, tmp));
        if (k > 0) SSW_TRY(topk(ctx, yb, n, w, h, cfg->ordering, k, idx));      // :493
        SSW_TRY(forward_from_rgb(ctx, cfg->precision, dev_derived_rgb, u8, f0 * plane, n, w, h, yd, nullptr, nullptr, tmp));   // Reader::derived
        if (k > 0) {
            StageTimer t(ctx, SSW_STAGE_EXTRACT);                               // :529-539
            SSW_TRY(launch_extract(ctx->stream, yb, yd, n, plane, idx, k, cfg->method, cfg->alpha,
                                   dev_extracted + f0 * k));
        }
        if (dev_marks) {
            StageTimer t(ctx, SSW_STAGE_SIMILARITY);                            // :696-714
            SSW_TRY(launch_similarity(ctx->stream, dev_extracted + f0 * k, dev_marks + f0 * k, n, k, dev_sims + f0));
        }
    }
    return SSW_OK;
}

int get_taps(ssw_ctx* ctx, size_t in_len, size_t out_len, DeviceTaps* out) {
    auto key = std::make_pair(in_len, out_len);
    auto it = ctx->taps.find(key);
    if (it != ctx->taps.end()) { *out = it->second; return SSW_OK; }
    ResizeTaps host;
    build_resize_taps(in_len, out_len, host);
    DeviceTaps d;
    d.max_taps = host.max_taps;
    SSW_ALLOC(&d.left, out_len * sizeof(uint32_t));
    SSW_ALLOC(&d.count, out_len * sizeof(uint32_t));
    SSW_ALLOC(&d.weights, host.weights.size() * sizeof(float));
    SSW_HIP_CHECK(hipMemcpy(d.left, host.left.data(), out_len * sizeof(uint32_t), hipMemcpyHostToDevice));
    SSW_HIP_CHECK(hipMemcpy(d.count, host.count.data(), out_len * sizeof(uint32_t), hipMemcpyHostToDevice));
    SSW_HIP_CHECK(hipMemcpy(d.weights, host.weights.data(), host.weights.size() * sizeof(float), hipMemcpyHostToDevice));
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
    return batch_embed_impl(ctx, cfg, dev_rgb, true, n_frames, w, h, dev_marks, k, dev_rgb_out, true, nullptr, nullptr);
}

int ssw_batch_extract_rgb8(ssw_ctx* ctx, const ssw_config* cfg, const uint8_t* dev_base_rgb,
                           const uint8_t* dev_derived_rgb, size_t n_frames, size_t w, size_t h, size_t k,
                           float* dev_extracted, const float* dev_marks, float* dev_sims) {
    return batch_extract_impl(ctx, cfg, dev_base_rgb, dev_derived_rgb, true, n_frames, w, h, k, dev_extracted,
                              dev_marks, dev_sims);
}

// ---- 8-bit boundary and the resize attack ---------------------------------------------------------
int ssw_convert_rgb8_to_f32(ssw_ctx* ctx, const uint8_t* dev_in, size_t n_values, float* dev_out) {
    if (!ctx || (n_values && (!dev_in || !dev_out))) return SSW_ERR_BAD_ARG;
    DeviceGuard g(ctx->device);
    StageTimer t(ctx, SSW_STAGE_CONVERT);
    return launch_u8_to_f32(ctx->stream, dev_in, n_values, dev_out);
}

int ssw_convert_f32_to_rgb8(ssw_ctx* ctx, const float* dev_in, size_t n_values, uint8_t* dev_out) {
    if (!ctx || (n_values && (!dev_in || !dev_out))) return SSW_ERR_BAD_ARG;
    DeviceGuard g(ctx->device);
    StageTimer t(ctx, SSW_STAGE_CONVERT);
    return launch_f32_to_u8(ctx->stream, dev_in, n_values, dev_out);
}

int ssw_resize_rgb8(ssw_ctx* ctx, const uint8_t* dev_in, size_t n_frames, size_t w, size_t h, size_t nw,
                    size_t nh, uint8_t* dev_out) {
    if (!ctx || !dev_in || !dev_out) return SSW_ERR_BAD_ARG;
    if (w == 0 || h == 0 || nw == 0 || nh == 0) return SSW_ERR_BAD_DIMS;
    DeviceGuard g(ctx->device);
    if (nw == w && nh == h) {                         // the crate copies when the size is unchanged
        SSW_HIP_CHECK(hipMemcpyAsync(dev_out, dev_in, n_frames * w * h * 3, hipMemcpyDeviceToDevice, ctx->stream));
        return SSW_OK;
    }
    DeviceTaps vt, ht;
    SSW_TRY(get_taps(ctx, h, nh, &vt));
    SSW_TRY(get_taps(ctx, w, nw, &ht));
    const size_t chunk = effective_chunk(ctx, w, h, n_frames);
    SSW_TRY(grow(ctx->resize_tmp, chunk * nh * w * 3 * sizeof(float)));
    for (size_t f0 = 0; f0 < n_frames; f0 += chunk) {
        const size_t n = std::min(chunk, n_frames - f0);
        StageTimer t(ctx, SSW_STAGE_RESIZE);
        SSW_TRY(launch_resize_rgb8(ctx->stream, dev_in + f0 * w * h * 3, n, w, h, nw, nh, vt, ht,
                                   (float*)ctx->resize_tmp.p, dev_out + f0 * nw * nh * 3));
    }
    return SSW_OK;
}

// ---- Writer ---------------------------------------------------------------------------------
int ssw_writer_create(ssw_ctx* ctx, const float* rgb_hwc, size_t w, size_t h,
                      const ssw_config* cfg, ssw_writer** out) {
    if (!ctx || !rgb_hwc || !out) return SSW_ERR_BAD_ARG;
    *out = nullptr;
    SSW_TRY(check_config(cfg));
    if (w == 0 || h == 0) return SSW_ERR_BAD_DIMS;
    DeviceGuard g(ctx->device);
    const size_t plane = w * h;
    ssw_writer* wr = new (std::nothrow) ssw_writer();
    if (!wr) return SSW_ERR_OUT_OF_MEMORY;
    wr->ctx = ctx; wr->w = w; wr->h = h; wr->cfg = *cfg;
    auto fail = [&](int rc) { ssw_writer_destroy(wr); return rc; };
    if (dev_malloc((void**)&wr->y, plane * 4) != SSW_OK || dev_malloc((void**)&wr->i, plane * 4) != SSW_OK ||
        dev_malloc((void**)&wr->q, plane * 4) != SSW_OK)
        return fail(SSW_ERR_OUT_OF_MEMORY);
    int rc = grow(ctx->plane[3], std::max(plane * 3, plane) * sizeof(float));
    if (rc != SSW_OK) return fail(rc);
    float* stage = (float*)ctx->plane[3].p;                          // rgb staging, then DCT scratch
    if (hipMemcpyAsync(stage, rgb_hwc, plane * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return fail(SSW_ERR_HIP);
    {
        StageTimer t(ctx, SSW_STAGE_RGB_TO_YIQ);
        rc = launch_rgb_to_yiq(ctx->stream, stage, plane, wr->y, wr->i, wr->q);          // :308
    }
    if (rc != SSW_OK) return fail(rc);
    rc = dct2d_planes(ctx, SSW_DCT2, cfg->precision, 1, w, h, wr->y, stage);             // :313
    if (rc != SSW_OK) return fail(rc);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(SSW_ERR_HIP);
    *out = wr;
    return SSW_OK;
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
    DeviceGuard g(ctx->device);
    const size_t plane = wr->w * wr->h;
    if (n_marks == 0) return SSW_OK;
    // zip(indices, mark) truncates every mark at w*h-1 entries (:396, :402)
    std::vector<uint32_t> offs(n_marks), lns(n_marks);
    size_t total = 0, max_len = 0;
    for (size_t m = 0; m < n_marks; ++m) {
        if (lens[m] && !marks[m]) return SSW_ERR_BAD_ARG;
        const size_t len = std::min(lens[m], plane - 1);
        offs[m] = (uint32_t)total; lns[m] = (uint32_t)len;
        total += len; max_len = std::max(max_len, len);
    }
    if (max_len == 0) return SSW_OK;
    std::vector<float> packed(total);
    for (size_t m = 0; m < n_marks; ++m)
        if (lns[m]) std::memcpy(packed.data() + offs[m], marks[m], lns[m] * sizeof(float));
    const size_t bytes_marks = (total * 4 + 15) / 16 * 16, bytes_tab = (n_marks * 4 + 15) / 16 * 16;
    SSW_TRY(grow(ctx->small, bytes_marks + 2 * bytes_tab));
    char* base = (char*)ctx->small.p;
    SSW_HIP_CHECK(hipMemcpyAsync(base, packed.data(), total * 4, hipMemcpyHostToDevice, ctx->stream));
    SSW_HIP_CHECK(hipMemcpyAsync(base + bytes_marks, offs.data(), n_marks * 4, hipMemcpyHostToDevice, ctx->stream));
    SSW_HIP_CHECK(hipMemcpyAsync(base + bytes_marks + bytes_tab, lns.data(), n_marks * 4, hipMemcpyHostToDevice, ctx->stream));
    if (keep_original && !wr->y0) {
        SSW_ALLOC(&wr->y0, plane * sizeof(float));
        SSW_HIP_CHECK(hipMemcpyAsync(wr->y0, wr->y, plane * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (max_len > wr->idx_k) {                                        // :314, from the original coefficients
        if (wr->idx) { SSW_HIP_CHECK(hipStreamSynchronize(ctx->stream)); SSW_HIP_CHECK(hipFree(wr->idx)); wr->idx = nullptr; wr->idx_k = 0; }
        SSW_ALLOC(&wr->idx, max_len * sizeof(uint32_t));
        SSW_TRY(topk(ctx, wr->y0 ? wr->y0 : wr->y, 1, wr->w, wr->h, wr->cfg.ordering, max_len, wr->idx));
        wr->idx_k = max_len;
    }
    {
        StageTimer t(ctx, SSW_STAGE_EMBED);
        SSW_TRY(launch_embed(ctx->stream, wr->y, 1, plane, wr->idx, max_len, (const float*)base,
                             (const uint32_t*)(base + bytes_marks), (const uint32_t*)(base + bytes_marks + bytes_tab),
                             n_marks, max_len, max_len, wr->cfg.method, wr->cfg.alpha));
    }
    SSW_HIP_CHECK(hipStreamSynchronize(ctx->stream));                 // host staging vectors die here
    return SSW_OK;
}

int ssw_writer_embed(ssw_writer* wr, const float* const* marks, const size_t* lens, size_t n_marks) {
    return writer_embed_impl(wr, marks, lens, n_marks, true);
}

int ssw_writer_result(ssw_writer* wr, float* out_rgb_hwc) {
    if (!wr || !out_rgb_hwc) return SSW_ERR_BAD_ARG;
    if (wr->consumed) return SSW_ERR_CONSUMED;
    ssw_ctx* ctx = wr->ctx;
    DeviceGuard g(ctx->device);
    const size_t plane = wr->w * wr->h;
    SSW_TRY(grow(ctx->plane[3], plane * 3 * sizeof(float)));
    float* stage = (float*)ctx->plane[3].p;
    SSW_TRY(dct2d_planes(ctx, SSW_DCT3, wr->cfg.precision, 1, wr->w, wr->h, wr->y, stage));      // :368-374
    {
        StageTimer t(ctx, SSW_STAGE_YIQ_TO_RGB);
        SSW_TRY(launch_yiq_to_rgb(ctx->stream, wr->y, wr->i, wr->q, plane, stage));              // :377
    }
    SSW_HIP_CHECK(hipMemcpyAsync(out_rgb_hwc, stage, plane * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SSW_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    wr->consumed = true;                                              // `result(self)` consumes
    return SSW_OK;
}

int ssw_writer_mark(ssw_writer* wr, const float* const* marks, const size_t* lens, size_t n_marks,
                    float* out_rgb_hwc) {
    SSW_TRY(writer_embed_impl(wr, marks, lens, n_marks, false));      // :356 (consumed next: no snapshot needed)
    return ssw_writer_result(wr, out_rgb_hwc);                        // :357
}

int ssw_writer_destroy(ssw_writer* wr) {
    if (!wr) return SSW_OK;
    DeviceGuard g(wr->ctx->device);
    (void)hipStreamSynchronize(wr->ctx->stream);
    if (wr->y) (void)hipFree(wr->y);
    if (wr->i) (void)hipFree(wr->i);
    if (wr->q) (void)hipFree(wr->q);
    if (wr->y0) (void)hipFree(wr->y0);
    if (wr->idx) (void)hipFree(wr->idx);
    delete wr;
    return SSW_OK;
}

// ---- Reader ---------------------------------------------------------------------------------
int ssw_reader_create(ssw_ctx* ctx, const float* rgb_hwc, size_t w, size_t h, int is_base,
                      const ssw_config* cfg, ssw_reader** out) {
    if (!ctx || !rgb_hwc || !out) return SSW_ERR_BAD_ARG;
    *out = nullptr;
    ssw_config c;
    ssw_config_default(&c);
    if (cfg) c = *cfg;
    else if (is_base) return SSW_ERR_BAD_ARG;                         // config.unwrap(), :482
    SSW_TRY(check_config(&c));
    if (w == 0 || h == 0) return SSW_ERR_BAD_DIMS;
    DeviceGuard g(ctx->device);
    const size_t plane = w * h;
    ssw_reader* rd = new (std::nothrow) ssw_reader();
    if (!rd) return SSW_ERR_OUT_OF_MEMORY;
    rd->ctx = ctx; rd->w = w; rd->h = h; rd->is_base = is_base != 0; rd->cfg = c;
    auto fail = [&](int rc) { ssw_reader_destroy(rd); return rc; };
    if (dev_malloc((void**)&rd->y, plane * 4) != SSW_OK) return fail(SSW_ERR_OUT_OF_MEMORY);
    int rc = grow(ctx->plane[3], plane * 3 * sizeof(float));
    if (rc != SSW_OK) return fail(rc);
    float* stage = (float*)ctx->plane[3].p;
    if (hipMemcpyAsync(stage, rgb_hwc, plane * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return fail(SSW_ERR_HIP);
    {
        StageTimer t(ctx, SSW_STAGE_RGB_TO_YIQ);
        rc = launch_rgb_to_yiq(ctx->stream, stage, plane, rd->y, nullptr, nullptr);      // :476
    }
    if (rc != SSW_OK) return fail(rc);
    rc = dct2d_planes(ctx, SSW_DCT2, c.precision, 1, w, h, rd->y, stage);                // :480
    if (rc != SSW_OK) return fail(rc);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(SSW_ERR_HIP);
    *out = rd;
    return SSW_OK;
}

int ssw_reader_coefficients(ssw_reader* rd, float* out_plane) {
    if (!rd || !out_plane) return SSW_ERR_BAD_ARG;
    return ssw_copy_to_host(rd->ctx, out_plane, rd->y, rd->w * rd->h * sizeof(float));
}

static int reader_ensure_indices(ssw_reader* rd, size_t k) {
    if (!rd->is_base) return SSW_ERR_NOT_BASE;                        // base.unwrap(), :507 / :530
    if (k > rd->w * rd->h - 1) return SSW_ERR_K_TOO_LARGE;
    if (k <= rd->idx_k) return SSW_OK;
    ssw_ctx* ctx = rd->ctx;
    if (rd->idx) { SSW_HIP_CHECK(hipFree(rd->idx)); rd->idx = nullptr; rd->idx_k = 0; }
    SSW_ALLOC(&rd->idx, k * sizeof(uint32_t));
    SSW_TRY(topk(ctx, rd->y, 1, rd->w, rd->h, rd->cfg.ordering, k, rd->idx));           // :493
    rd->idx_k = k;
    return SSW_OK;
}

int ssw_reader_indices(ssw_reader* rd, size_t k, uint64_t* out) {
    if (!rd || (k && !out)) return SSW_ERR_BAD_ARG;
    DeviceGuard g(rd->ctx->device);
    if (k == 0) return rd->is_base ? SSW_OK : SSW_ERR_NOT_BASE;
    SSW_TRY(reader_ensure_indices(rd, k));
    ssw_ctx* ctx = rd->ctx;
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
    DeviceGuard g(ctx->device);
    SSW_TRY(reader_ensure_indices(base, k));
    SSW_TRY(grow(ctx->small, k * sizeof(float)));
    {
        StageTimer t(ctx, SSW_STAGE_EXTRACT);
        // cached list may be longer than k: its first k entries are the first k of the order
        SSW_TRY(launch_extract(ctx->stream, base->y, derived->y, 1, plane, base->idx, k, base->cfg.method,
                               base->cfg.alpha, (float*)ctx->small.p));
    }
    return ssw_copy_to_host(ctx, out, ctx->small.p, k * sizeof(float));
}

int ssw_reader_destroy(ssw_reader* rd) {
    if (!rd) return SSW_OK;
    DeviceGuard g(rd->ctx->device);
    (void)hipStreamSynchronize(rd->ctx->stream);
    if (rd->y) (void)hipFree(rd->y);
    if (rd->idx) (void)hipFree(rd->idx);
    delete rd;
    return SSW_OK;
}

// ---- Tester ---------------------------------------------------------------------------------
int ssw_similarity(ssw_ctx* ctx, const float* extracted, size_t n_extracted, const float* mark,
                   size_t n_mark, float* out_similarity) {
    if (!ctx || !out_similarity || (n_extracted && !extracted) || (n_mark && !mark)) return SSW_ERR_BAD_ARG;
    if (n_extracted != n_mark) return SSW_ERR_LENGTH_MISMATCH;                            // :697-700
    DeviceGuard g(ctx->device);
    const size_t k = n_extracted;
    const size_t bytes = (k * 4 + 15) / 16 * 16;
    SSW_TRY(grow(ctx->small, 2 * bytes + 16));
    char* base = (char*)ctx->small.p;
    if (k) {
        SSW_HIP_CHECK(hipMemcpyAsync(base, extracted, k * 4, hipMemcpyHostToDevice, ctx->stream));
        SSW_HIP_CHECK(hipMemcpyAsync(base + bytes, mark, k * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    {
        StageTimer t(ctx, SSW_STAGE_SIMILARITY);
        SSW_TRY(launch_similarity(ctx->stream, (const float*)base, (const float*)(base + bytes), 1, k,
                                  (float*)(base + 2 * bytes)));
    }
    return ssw_copy_to_host(ctx, out_similarity, base + 2 * bytes, sizeof(float));
}

// ---- synthetic frames -----------------------------------------------------------------------
int ssw_synth_frames(ssw_ctx* ctx, uint32_t seed, uint32_t first_frame, size_t n_frames, size_t w,
                     size_t h, float* dev_rgb) {
    if (!ctx || !dev_rgb) return SSW_ERR_BAD_ARG;
    DeviceGuard g(ctx->device);
    return launch_synth(ctx->stream, seed, first_frame, n_frames, w, h, dev_rgb);
}

}  // extern "C"
