/*
 * ssw_oracle.c -- CPU restatement of the reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- see ssw_oracle.h for the rules and the pinning
 * status.  Plain C99, built with `gcc -O2 -ffp-contract=off` so that every
 * `*` and `+` rounds separately, exactly as the Rust reference does (rustc
 * never contracts to FMA).
 */
#include "ssw_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- 1-D transform kernels (stand-in for rustdct, see fft_impl.inc) ------ */
#define REAL double
#define SUF f64
#include "fft_impl.inc"
#undef REAL
#undef SUF
#define REAL float
#define SUF f32
#include "fft_impl.inc"
#undef REAL
#undef SUF

/* ---- yiq.rs ---------------------------------------------------------------
 * Matrices: src/yiq.rs:157-159 and :163-165.  Product order src/yiq.rs:131-136:
 * (m0*v0 + m1*v1) + m2*v2, each op rounded to f32.                           */
static const float RGB2YIQ[3][3] = {
    {0.30f, 0.59f, 0.11f}, {0.60f, -0.28f, -0.32f}, {0.21f, -0.52f, 0.31f}};
static const float YIQ2RGB[3][3] = {
    {1.0f, 0.948262f, 0.624013f}, {1.0f, -0.276066f, -0.639810f}, {1.0f, -1.105450f, 1.729860f}};

static inline float row3(const float m[3], float a, float b, float c) {
    return m[0] * a + m[1] * b + m[2] * c;
}
/* f32::clamp semantics (src/yiq.rs:139-147): NaN passes through */
static inline float clampf(float x, float lo, float hi) {
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}

void sswo_rgb_to_yiq(const float* rgb, size_t npix, float* y, float* i, float* q) {
    for (size_t p = 0; p < npix; ++p) {                     /* src/yiq.rs:181-183 */
        const float r = rgb[3 * p], g = rgb[3 * p + 1], b = rgb[3 * p + 2];
        y[p] = row3(RGB2YIQ[0], r, g, b);
        i[p] = row3(RGB2YIQ[1], r, g, b);
        q[p] = row3(RGB2YIQ[2], r, g, b);
    }
}

void sswo_yiq_to_rgb(const float* y, const float* i, const float* q, size_t npix, float* rgb) {
    for (size_t p = 0; p < npix; ++p) {                     /* src/yiq.rs:191-194 */
        rgb[3 * p + 0] = clampf(row3(YIQ2RGB[0], y[p], i[p], q[p]), 0.0f, 1.0f);
        rgb[3 * p + 1] = clampf(row3(YIQ2RGB[1], y[p], i[p], q[p]), 0.0f, 1.0f);
        rgb[3 * p + 2] = clampf(row3(YIQ2RGB[2], y[p], i[p], q[p]), 0.0f, 1.0f);
    }
}

/* ---- dct2d.rs ------------------------------------------------------------ */

/* cos(pi * num / den) with the argument reduced exactly in integers */
static double cos_pi_frac(uint64_t num, uint64_t den) {
    const double pi = 3.14159265358979323846;
    num %= 2 * den;                                    /* period 2 */
    double sign = 1.0;
    if (num > den) num = 2 * den - num;                /* cos(2pi - x) = cos x */
    if (2 * num > den) { num = den - num; sign = -1.0; }   /* cos(pi - x) = -cos x */
    return sign * cos(pi * (double)num / (double)den);
}

typedef struct {
    int backend; size_t n;
    plan_f64* p64; plan_f32* p32;
    double* in64; double* out64;
    float* out32;
} kernel1d;

static void kernel1d_init(kernel1d* k, int backend, size_t n) {
    memset(k, 0, sizeof(*k));
    k->backend = backend; k->n = n;
    if (backend == SSWO_BACKEND_F64) k->p64 = plan_new_f64(n);
    if (backend == SSWO_BACKEND_F32) k->p32 = plan_new_f32(n);
    k->in64 = (double*)malloc(sizeof(double) * (n ? n : 1));
    k->out64 = (double*)malloc(sizeof(double) * (n ? n : 1));
    k->out32 = (float*)malloc(sizeof(float) * (n ? n : 1));
}
static void kernel1d_free(kernel1d* k) {
    plan_free_f64(k->p64); plan_free_f32(k->p32);
    free(k->in64); free(k->out64); free(k->out32);
}

/* tmp[0..n) <- rustdct-convention DCT-II or DCT-III of tmp, result in f32
   (what `process_dct2/3_with_scratch` leaves in `tmp`, src/dct2d.rs:141-146). */
static void kernel1d_run(kernel1d* k, int dct3, float* tmp) {
    const size_t n = k->n;
    if (k->backend == SSWO_BACKEND_F32) {
        if (dct3) dct3_1d_f32(k->p32, tmp, k->out32); else dct2_1d_f32(k->p32, tmp, k->out32);
        memcpy(tmp, k->out32, sizeof(float) * n);
        return;
    }
    for (size_t i = 0; i < n; ++i) k->in64[i] = (double)tmp[i];
    if (k->backend == SSWO_BACKEND_F64) {
        if (dct3) dct3_1d_f64(k->p64, k->in64, k->out64); else dct2_1d_f64(k->p64, k->in64, k->out64);
    } else {                                            /* naive O(n^2), definition itself */
        for (size_t kk = 0; kk < n; ++kk) {
            double s = 0.0;
            if (!dct3) {
                for (size_t j = 0; j < n; ++j)
                    s += k->in64[j] * cos_pi_frac((uint64_t)kk * (2 * j + 1), 2 * n);
            } else {
                s = 0.5 * k->in64[0];
                for (size_t j = 1; j < n; ++j)
                    s += k->in64[j] * cos_pi_frac((uint64_t)j * (2 * kk + 1), 2 * n);
            }
            k->out64[kk] = s;
        }
    }
    for (size_t i = 0; i < n; ++i) tmp[i] = (float)k->out64[i];
}

int sswo_dct1d(int type, int backend, size_t n, float* data) {
    if (n == 0 || (type != SSWO_DCT2 && type != SSWO_DCT3) || backend < 0 || backend > 2) return -1;
    kernel1d k; kernel1d_init(&k, backend, n);
    kernel1d_run(&k, type == SSWO_DCT3, data);
    kernel1d_free(&k);
    return 0;
}

int sswo_dct2d(int type, int backend, size_t w, size_t h, float* data) {
    if (w == 0 || h == 0 || type < 0 || type > 2 || backend < 0 || backend > 2) return -1;
    /* src/dct2d.rs:93-98: larger dimension first */
    const int first_is_row = (w >= h);
    /* src/dct2d.rs:107-111 */
    const float scaling = (type == SSWO_DCT3) ? 0.5f : 2.0f;
    const int dct3 = (type == SSWO_DCT3);
    for (int pass = 0; pass < 2; ++pass) {
        const int is_row = (pass == 0) ? first_is_row : !first_is_row;
        const size_t length = is_row ? w : h;                /* src/dct2d.rs:114-118 */
        kernel1d k; kernel1d_init(&k, backend, length);
        float* tmp = (float*)malloc(sizeof(float) * length);
        /* src/dct2d.rs:154-155 / :190-191 */
        const float s0 = sqrtf(1.0f / (4.0f * (float)length));
        const float sn = sqrtf(1.0f / (2.0f * (float)length));
        const size_t lines = is_row ? h : w;
        const size_t stride = is_row ? 1 : w;
        for (size_t l = 0; l < lines; ++l) {
            float* base = is_row ? data + l * w : data + l;
            for (size_t j = 0; j < length; ++j) tmp[j] = base[j * stride];   /* :132-136 / :174-178 */
            kernel1d_run(&k, dct3, tmp);
            if (type == SSWO_DCT2_ORTHOGONAL) {                              /* :153-162 / :189-198 */
                for (size_t j = 0; j < length; ++j)
                    base[j * stride] = ((j == 0 ? s0 : sn) * scaling) * tmp[j];
            } else {                                                          /* :164-167 / :200-203 */
                for (size_t j = 0; j < length; ++j) base[j * stride] = scaling * tmp[j];
            }
        }
        free(tmp);
        kernel1d_free(&k);
    }
    if (type == SSWO_DCT3) {                                                  /* :213-217 */
        const float corr = (float)4 / (float)(w * h);
        for (size_t j = 0; j < w * h; ++j) data[j] = data[j] * corr;
    }
    return 0;
}

/* ---- algorithm.rs: ordering ---------------------------------------------- */

/* f32::total_cmp as an integer key: larger int32 <=> Greater */
static inline int32_t total_key(float v) {
    int32_t b; memcpy(&b, &v, 4);
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
    return b;
}

/* src/algorithm.rs:240-267, evaluated in f32 exactly as written there */
static inline float ortho_scaling(size_t index, float value, size_t width, size_t height) {
    const float s_k0_w = sqrtf(1.0f / (4.0f * (float)width));
    const float s_k0_h = sqrtf(1.0f / (4.0f * (float)height));
    const float s_w = sqrtf(1.0f / (2.0f * (float)width));
    const float s_h = sqrtf(1.0f / (2.0f * (float)height));
    const int first_row = index < width;
    const int first_column = (index % width) == 0;
    float scaling = 1.0f;
    if (first_row) scaling *= s_k0_w; else scaling *= s_w;
    if (first_column) scaling *= s_k0_h; else scaling *= s_h;
    return scaling * value;
}

int32_t sswo_order_key(int ordering, size_t w, size_t h, size_t index, float value) {
    switch (ordering) {
    case SSWO_ORDER_ENERGY:                                /* :214-221 */
        return total_key(value * value);
    case SSWO_ORDER_ENERGY_ORTHOGONAL: {                   /* :178-182 */
        const float s = ortho_scaling(index, value, w, h);
        return total_key(s * s);
    }
    default: {                                             /* Legacy :183-187, :225-232 */
        return total_key(ortho_scaling(index, value, w, h));
    }
    }
}

typedef struct { const float* coef; int ordering; size_t w, h; } cmp_ctx;

/* "a sorts before b"?  sort_by(|a, b| f(b, a)) (src/algorithm.rs:205): a precedes b
   iff key(b) < key(a); equal keys keep their original (index-ascending) order.   */
static int key_greater(const cmp_ctx* c, uint32_t ia, uint32_t ib) {
    return sswo_order_key(c->ordering, c->w, c->h, ia, c->coef[ia]) >
           sswo_order_key(c->ordering, c->w, c->h, ib, c->coef[ib]);
}

/* stable bottom-up merge sort of an index array, comparator evaluated on the fly
   (like the boxed comparator of the reference) */
static void merge_sort_idx(uint32_t* a, size_t n, const cmp_ctx* c) {
    if (n < 2) return;
    uint32_t* tmp = (uint32_t*)malloc(sizeof(uint32_t) * n);
    uint32_t* src = a; uint32_t* dst = tmp;
    for (size_t width = 1; width < n; width *= 2) {
        for (size_t lo = 0; lo < n; lo += 2 * width) {
            size_t mid = lo + width < n ? lo + width : n;
            size_t hi = lo + 2 * width < n ? lo + 2 * width : n;
            size_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) {
                /* take right only if strictly greater: stability */
                if (key_greater(c, src[j], src[i])) dst[o++] = src[j++]; else dst[o++] = src[i++];
            }
            while (i < mid) dst[o++] = src[i++];
            while (j < hi) dst[o++] = src[j++];
        }
        uint32_t* t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, sizeof(uint32_t) * n);
    free(tmp);
}

static int cmp_i32_desc(const void* a, const void* b) {
    int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return (x < y) - (x > y);
}

size_t sswo_indices(const float* coef, size_t n, int ordering, size_t w, size_t h,
                    size_t k, uint64_t* out) {
    if (n < 2) return 0;
    const size_t m = n - 1;                                 /* DC skipped, :204 */
    if (k > m) k = m;
    if (k == 0) return 0;
    cmp_ctx c = {coef, ordering, w, h};
    if (k == m) {                                           /* the reference's full sort */
        uint32_t* idx = (uint32_t*)malloc(sizeof(uint32_t) * m);
        for (size_t i = 0; i < m; ++i) idx[i] = (uint32_t)(i + 1);
        merge_sort_idx(idx, m, &c);
        for (size_t i = 0; i < m; ++i) out[i] = idx[i];
        free(idx);
        return m;
    }
    /* partial: threshold = k-th largest key, then the same stable order on the survivors */
    int32_t* keys = (int32_t*)malloc(sizeof(int32_t) * m);
    for (size_t i = 0; i < m; ++i) keys[i] = sswo_order_key(ordering, w, h, i + 1, coef[i + 1]);
    int32_t* sorted = (int32_t*)malloc(sizeof(int32_t) * m);
    memcpy(sorted, keys, sizeof(int32_t) * m);
    qsort(sorted, m, sizeof(int32_t), cmp_i32_desc);
    const int32_t thr = sorted[k - 1];
    size_t n_gt = 0;
    while (n_gt < m && sorted[n_gt] > thr) ++n_gt;
    free(sorted);
    size_t need_eq = k - n_gt;
    uint32_t* idx = (uint32_t*)malloc(sizeof(uint32_t) * k);
    size_t o = 0;
    for (size_t i = 0; i < m && o < k; ++i) {
        if (keys[i] > thr) idx[o++] = (uint32_t)(i + 1);
        else if (keys[i] == thr && need_eq) { idx[o++] = (uint32_t)(i + 1); --need_eq; }
    }
    free(keys);
    merge_sort_idx(idx, o, &c);
    for (size_t i = 0; i < o; ++i) out[i] = idx[i];
    free(idx);
    return o;
}

/* ---- algorithm.rs: embed / extract / similarity --------------------------- */

static inline float insert_fn(int method, float alpha, float original, float mark) {
    switch (method) {
    case SSWO_OPTION1: return original + alpha * mark;                  /* :414-416 */
    case SSWO_OPTION2: return original * (1.0f + alpha * mark);         /* :420-424 */
    default:           return original * expf(alpha * mark);            /* :428-432 */
    }
}
static inline float extract_fn(int method, float alpha, float base, float derived) {
    switch (method) {
    case SSWO_OPTION1: return (derived - base) / alpha;                 /* :566-572 */
    case SSWO_OPTION2: return (derived - base) / (base * alpha);        /* :576-583 */
    default:           return logf(derived / base) / alpha;             /* :587-593 */
    }
}

void sswo_embed(float* coef, size_t n, const uint64_t* indices, size_t n_indices,
                int method, float alpha,
                const float* const* marks, const size_t* mark_lens, size_t n_marks) {
    if (n_marks == 1) {                                                 /* :394-398 */
        const size_t len = mark_lens[0] < n_indices ? mark_lens[0] : n_indices;   /* zip */
        for (size_t i = 0; i < len; ++i) {
            const size_t j = (size_t)indices[i];
            coef[j] = insert_fn(method, alpha, coef[j], marks[0][i]);
        }
    } else {                                                            /* :399-408 */
        float* orig = (float*)malloc(sizeof(float) * (n ? n : 1));
        memcpy(orig, coef, sizeof(float) * n);
        for (size_t m = 0; m < n_marks; ++m) {
            const size_t len = mark_lens[m] < n_indices ? mark_lens[m] : n_indices;
            for (size_t i = 0; i < len; ++i) {
                const size_t j = (size_t)indices[i];
                const float updated = insert_fn(method, alpha, orig[j], marks[m][i]);
                const float change = updated - orig[j];
                coef[j] += change;
            }
        }
        free(orig);
    }
}

int sswo_extract(const float* base, size_t n_base, const float* derived, size_t n_derived,
                 const uint64_t* indices, int method, float alpha, float* out, size_t k) {
    if (n_derived != n_base) return 1;                                  /* :550-552 */
    if (k >= n_base) return 2;                                          /* :553-555 */
    for (size_t i = 0; i < k; ++i) {                                    /* :556-561 */
        const size_t j = (size_t)indices[i];
        out[i] = extract_fn(method, alpha, base[j], derived[j]);
    }
    return 0;
}

float sswo_similarity(const float* extracted, const float* mark, size_t k) {
    float nominator = 0.0f, denominator = 0.0f;                         /* :702-703 */
    for (size_t i = 0; i < k; ++i) {                                    /* :704-711 */
        nominator += extracted[i] * mark[i];
        denominator += extracted[i] * extracted[i];
    }
    return nominator / sqrtf(denominator);                              /* :712 */
}

/* ---- synthetic frames (bench plumbing; twin of the device generator) ----- */

static inline uint32_t h32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
static inline float u01(uint32_t h) { return (float)(h >> 8) * 0x1p-24f; }
static inline float lattice(uint32_t base, uint32_t ix, uint32_t iy) {
    return u01(h32(base ^ (ix * 0x9E3779B1U + iy * 0x85EBCA77U)));
}

void sswo_synth_frame(uint32_t seed, uint32_t frame, size_t w, size_t h, float* rgb) {
    const float norm = 1.0f / 2.004375f;     /* 1 / (sum_{o<7} 2^-o + 0.02) */
    const uint32_t fbase = h32(seed * 0x9E3779B1U + frame);
    for (size_t y = 0; y < h; ++y) {
        for (size_t x = 0; x < w; ++x) {
            for (uint32_t ch = 0; ch < 3; ++ch) {
                float acc = 0.0f;
                for (uint32_t o = 0; o < 7; ++o) {
                    const uint32_t base = h32(fbase ^ (ch * 0x632BE5ABU + o * 0x2545F491U + 1U));
                    const uint32_t shift = 8 - o, cell = 256U >> o;
                    const uint32_t ix = (uint32_t)x >> shift, iy = (uint32_t)y >> shift;
                    const float inv = 1.0f / (float)cell;                 /* power of two: exact */
                    const float fx = (float)((uint32_t)x & (cell - 1)) * inv;
                    const float fy = (float)((uint32_t)y & (cell - 1)) * inv;
                    const float v00 = lattice(base, ix, iy), v10 = lattice(base, ix + 1, iy);
                    const float v01 = lattice(base, ix, iy + 1), v11 = lattice(base, ix + 1, iy + 1);
                    const float top = v00 + fx * (v10 - v00);
                    const float bot = v01 + fx * (v11 - v01);
                    const float val = top + fy * (bot - top);
                    const float amp = 1.0f / (float)(1U << o);
                    acc = acc + amp * val;
                }
                const uint32_t wbase = h32(fbase ^ (ch * 0x632BE5ABU + 0x7F4A7C15U));
                const float white = lattice(wbase, (uint32_t)x, (uint32_t)y);
                acc = acc + 0.02f * white;
                rgb[3 * (y * w + x) + ch] = acc * norm;
            }
        }
    }
}

/* ---- whole-frame helpers for the cpu_baseline leg ------------------------ */

static void forward_y(const float* rgb, size_t w, size_t h, int backend,
                      float* y, float* i, float* q) {
    sswo_rgb_to_yiq(rgb, w * h, y, i, q);                 /* algorithm.rs:308 / :476 */
    sswo_dct2d(SSWO_DCT2, backend, w, h, y);              /* :313 / :480 */
}

void sswo_embed_frame(const float* rgb, size_t w, size_t h, int backend, int ordering,
                      int method, float alpha, const float* mark, size_t k,
                      int full_sort, float* out_rgb) {
    const size_t n = w * h;
    float* y = (float*)malloc(sizeof(float) * n);
    float* i = (float*)malloc(sizeof(float) * n);
    float* q = (float*)malloc(sizeof(float) * n);
    forward_y(rgb, w, h, backend, y, i, q);
    const size_t want = full_sort ? n - 1 : (k < n - 1 ? k : n - 1);
    uint64_t* idx = (uint64_t*)malloc(sizeof(uint64_t) * (want ? want : 1));
    const size_t got = sswo_indices(y, n, ordering, w, h, want, idx);   /* :314 */
    const float* marks[1] = {mark}; const size_t lens[1] = {k};
    sswo_embed(y, n, idx, got, method, alpha, marks, lens, 1);          /* :356 */
    sswo_dct2d(SSWO_DCT3, backend, w, h, y);                            /* :368-374 */
    sswo_yiq_to_rgb(y, i, q, n, out_rgb);                               /* :377 */
    free(idx); free(y); free(i); free(q);
}

float sswo_extract_frame(const float* base_rgb, const float* derived_rgb, size_t w, size_t h,
                         int backend, int ordering, int method, float alpha,
                         const float* mark, size_t k, int full_sort, float* extracted) {
    const size_t n = w * h;
    float* yb = (float*)malloc(sizeof(float) * n);
    float* yd = (float*)malloc(sizeof(float) * n);
    float* i = (float*)malloc(sizeof(float) * n);
    float* q = (float*)malloc(sizeof(float) * n);
    forward_y(base_rgb, w, h, backend, yb, i, q);                       /* Reader::base */
    const size_t want = full_sort ? n - 1 : (k < n - 1 ? k : n - 1);
    uint64_t* idx = (uint64_t*)malloc(sizeof(uint64_t) * (want ? want : 1));
    sswo_indices(yb, n, ordering, w, h, want, idx);                     /* :493 */
    forward_y(derived_rgb, w, h, backend, yd, i, q);                    /* Reader::derived */
    sswo_extract(yb, n, yd, n, idx, method, alpha, extracted, k);       /* :529-539 */
    const float sim = sswo_similarity(extracted, mark, k);              /* :696-714 */
    free(idx); free(yb); free(yd); free(i); free(q);
    return sim;
}

/* ---- 8-bit boundary and the resize attack (third-party `image 0.24.3`, NOT in the reference tree) --
 * Semantics restated from the published behaviour of the crate; "parity unpinned" beyond the
 * reference's own statistical asserts (tests/attack_resize.rs:65-66: sim > 9.5, published 9.85).
 *   into_rgb32f for 8-bit input : v / 255                      (call sites src/algorithm.rs:308, :476)
 *   into_rgb8 from Rgb32F       : round(clamp(v, 0, 1) * 255)  (tests/single_simple.rs:28)
 *   imageops::resize(CatmullRom): vertical pass into f32, then horizontal pass, clamp + round to
 *                                 u8 (tests/attack_resize.rs:17-36); cubic B = 0, C = 1/2, support 2,
 *                                 support and kernel argument scaled by max(1, in/out).          */
void sswo_u8_to_f32(const uint8_t* in, size_t n, float* out) {
    for (size_t i = 0; i < n; ++i) out[i] = (float)in[i] / 255.0f;
}
void sswo_f32_to_u8(const float* in, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        const float v = clampf(in[i], 0.0f, 1.0f) * 255.0f;
        out[i] = (uint8_t)roundf(v);
    }
}

/* 16-bit boundary (`image 0.24.3`, like the 8-bit forms):
 *   into_rgb32f for 16-bit input : v / 65535                      (call sites src/algorithm.rs:308, :476)
 *   into_rgb16 from Rgb32F       : round(clamp(v, 0, 1) * 65535)                                       */
void sswo_u16_to_f32(const uint16_t* in, size_t n, float* out) {
    for (size_t i = 0; i < n; ++i) out[i] = (float)in[i] / 65535.0f;
}
void sswo_f32_to_u16(const float* in, size_t n, uint16_t* out) {
    for (size_t i = 0; i < n; ++i) {
        const float v = clampf(in[i], 0.0f, 1.0f) * 65535.0f;
        out[i] = (uint16_t)roundf(v);
    }
}

static float catmullrom_kernel(float x) {          /* bc_cubic_spline(x, b = 0, c = 0.5) */
    const float b = 0.0f, c = 0.5f;
    const float a = fabsf(x);
    float k;
    if (a < 1.0f)
        k = (12.0f - 9.0f * b - 6.0f * c) * (a * a * a) + (-18.0f + 12.0f * b + 6.0f * c) * (a * a) + (6.0f - 2.0f * b);
    else if (a < 2.0f)
        k = (-b - 6.0f * c) * (a * a * a) + (6.0f * b + 30.0f * c) * (a * a) + (-12.0f * b - 48.0f * c) * a + (8.0f * b + 24.0f * c);
    else
        k = 0.0f;
    return k / 6.0f;
}

/* Filter taps of one output line: first input index and normalised weights (<= max_taps). */
size_t sswo_resize_taps(size_t in_len, size_t out_len, size_t out_idx, uint32_t* left_out, float* ws, size_t max_taps) {
    const float ratio = (float)in_len / (float)out_len;
    const float sratio = ratio < 1.0f ? 1.0f : ratio;
    const float src_support = 2.0f * sratio;
    float inputc = ((float)out_idx + 0.5f) * ratio;
    int64_t left = (int64_t)floorf(inputc - src_support);
    if (left < 0) left = 0;
    if (left > (int64_t)in_len - 1) left = (int64_t)in_len - 1;
    int64_t right = (int64_t)ceilf(inputc + src_support);
    if (right < left + 1) right = left + 1;
    if (right > (int64_t)in_len) right = (int64_t)in_len;
    inputc = inputc - 0.5f;
    size_t n = 0;
    float sum = 0.0f;
    for (int64_t i = left; i < right && n < max_taps; ++i) {
        const float w = catmullrom_kernel(((float)i - inputc) / sratio);
        ws[n++] = w;
        sum += w;
    }
    for (size_t i = 0; i < n; ++i) ws[i] /= sum;
    *left_out = (uint32_t)left;
    return n;
}

void sswo_resize_rgb8(const uint8_t* in, size_t w, size_t h, size_t nw, size_t nh, uint8_t* out) {
    if (nw == w && nh == h) { memcpy(out, in, w * h * 3); return; }
    enum { MAXT = 4096 };
    float* tmp = (float*)malloc(sizeof(float) * w * nh * 3);
    float* ws = (float*)malloc(sizeof(float) * MAXT);
    for (size_t oy = 0; oy < nh; ++oy) {                       /* vertical_sample */
        uint32_t left; const size_t nt = sswo_resize_taps(h, nh, oy, &left, ws, MAXT);
        for (size_t x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c) {
                float t = 0.0f;
                for (size_t i = 0; i < nt; ++i) t += (float)in[((left + i) * w + x) * 3 + c] * ws[i];
                tmp[(oy * w + x) * 3 + c] = t;
            }
    }
    for (size_t ox = 0; ox < nw; ++ox) {                       /* horizontal_sample */
        uint32_t left; const size_t nt = sswo_resize_taps(w, nw, ox, &left, ws, MAXT);
        for (size_t y = 0; y < nh; ++y)
            for (int c = 0; c < 3; ++c) {
                float t = 0.0f;
                for (size_t i = 0; i < nt; ++i) t += tmp[(y * w + left + i) * 3 + c] * ws[i];
                out[(y * nw + ox) * 3 + c] = (uint8_t)roundf(clampf(t, 0.0f, 255.0f));
            }
    }
    free(ws); free(tmp);
}
