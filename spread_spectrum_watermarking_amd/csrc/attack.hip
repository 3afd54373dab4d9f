// 8-bit image boundary and the resize attack of the reference's robustness tests.
//
// None of this arithmetic is in the reference tree: it lives in the third-party crate
// `image 0.24.3` (Cargo.lock), reached from
//   into_rgb32f()                      /root/reference/src/algorithm.rs:308, :476   (u8 -> v / 255)
//   into_rgb8()                        /root/reference/tests/single_simple.rs:28    (round(clamp(v,0,1)*255))
//   imageops::resize(.., CatmullRom)   /root/reference/tests/attack_resize.rs:17-36
// and is restated from the crate's published behaviour (vertical pass into f32, horizontal pass,
// clamp + round to u8; cubic B = 0, C = 1/2, support 2 scaled by max(1, in/out)).  The CPU oracle
// carries the same restatement and the two agree bit for bit; against the real crate the parity is
// "unpinned" (only the similarity thresholds of the reference's tests pin it).
//
// All kernels are HBM-bound streaming kernels; u8 frames cut the boundary traffic from 12 to 3 B/px.
#include <atomic>
#include <cmath>
#include <vector>

#include "ssw_internal.hpp"

namespace ssw {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline float clampf(float x, float lo, float hi) {
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}
__device__ inline float dot3a(float m0, float m1, float m2, float a, float b, float c) { return m0 * a + m1 * b + m2 * c; }

__global__ void u8_to_f32_kernel(const uint8_t* __restrict__ in, size_t n, float* __restrict__ out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (float)in[i] / 255.0f;
}
__global__ void f32_to_u8_kernel(const float* __restrict__ in, size_t n, uint8_t* __restrict__ out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (uint8_t)roundf(clampf(in[i], 0.0f, 1.0f) * 255.0f);
}

// fused u8 RGB -> Y (+ I, Q): into_rgb32f (v/255) followed by yiq.rs:177-186, 3 B/px read
template <bool WITH_IQ>
__global__ __launch_bounds__(256) void rgb8_to_yiq_kernel(const uint8_t* __restrict__ rgb, size_t npix,
                                                          float* __restrict__ y, float* __restrict__ ip,
                                                          float* __restrict__ qp) {
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
        const float r = (float)rgb[3 * p] / 255.0f, g = (float)rgb[3 * p + 1] / 255.0f, b = (float)rgb[3 * p + 2] / 255.0f;
        y[p] = dot3a(0.30f, 0.59f, 0.11f, r, g, b);
        if (WITH_IQ) {
            ip[p] = dot3a(0.60f, -0.28f, -0.32f, r, g, b);
            qp[p] = dot3a(0.21f, -0.52f, 0.31f, r, g, b);
        }
    }
}
// fused Y, I, Q -> u8 RGB: yiq.rs:187-197 (clamp to [0,1]) followed by into_rgb8
__global__ __launch_bounds__(256) void yiq_to_rgb8_kernel(const float* __restrict__ y, const float* __restrict__ ip,
                                                          const float* __restrict__ qp, size_t npix,
                                                          uint8_t* __restrict__ rgb) {
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
        const float yy = y[p], ii = ip[p], qq = qp[p];
        const float r = clampf(dot3a(1.0f, 0.948262f, 0.624013f, yy, ii, qq), 0.0f, 1.0f);
        const float g = clampf(dot3a(1.0f, -0.276066f, -0.639810f, yy, ii, qq), 0.0f, 1.0f);
        const float b = clampf(dot3a(1.0f, -1.105450f, 1.729860f, yy, ii, qq), 0.0f, 1.0f);
        rgb[3 * p + 0] = (uint8_t)roundf(clampf(r, 0.0f, 1.0f) * 255.0f);
        rgb[3 * p + 1] = (uint8_t)roundf(clampf(g, 0.0f, 1.0f) * 255.0f);
        rgb[3 * p + 2] = (uint8_t)roundf(clampf(b, 0.0f, 1.0f) * 255.0f);
    }
}

static inline unsigned sgrid(size_t items) {
    size_t b = (items + 255) / 256;
    if (b < 1) b = 1;
    if (b > 4096) b = 4096;
    return (unsigned)b;
}

// 16-bit boundary: into_rgb32f of an Rgb16 image (v / 65535, src/algorithm.rs:308, :476) and into_rgb16 of an Rgb32F one
// (round(clamp(v, 0, 1) * 65535), `image 0.24.3` like the 8-bit forms)
__global__ void u16_to_f32_kernel(const uint16_t* __restrict__ in, size_t n, float* __restrict__ out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (float)in[i] / 65535.0f;
}
__global__ void f32_to_u16_kernel(const float* __restrict__ in, size_t n, uint16_t* __restrict__ out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = in[i];
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);                // f32::clamp: NaN passes through and saturates to 0 below
        out[i] = (uint16_t)roundf(v * 65535.0f);
    }
}
template <bool WITH_IQ>
__global__ __launch_bounds__(256) void rgb16_to_yiq_kernel(const uint16_t* __restrict__ rgb, size_t npix,
                                                           float* __restrict__ y, float* __restrict__ ip, float* __restrict__ qp) {
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
        const float r = (float)rgb[3 * p] / 65535.0f, g = (float)rgb[3 * p + 1] / 65535.0f, b = (float)rgb[3 * p + 2] / 65535.0f;
        y[p] = dot3a(0.30f, 0.59f, 0.11f, r, g, b);                 // yiq.rs:131-136, :177-186
        if (WITH_IQ) {
            ip[p] = dot3a(0.60f, -0.28f, -0.32f, r, g, b);
            qp[p] = dot3a(0.21f, -0.52f, 0.31f, r, g, b);
        }
    }
}

int launch_u8_to_f32(hipStream_t st, const uint8_t* in, size_t n, float* out) {
    if (!n) return SSW_OK;
    u8_to_f32_kernel<<<sgrid(n), 256, 0, st>>>(in, n, out);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}
int launch_f32_to_u8(hipStream_t st, const float* in, size_t n, uint8_t* out) {
    if (!n) return SSW_OK;
    f32_to_u8_kernel<<<sgrid(n), 256, 0, st>>>(in, n, out);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}
int launch_rgb8_to_yiq(hipStream_t st, const uint8_t* rgb, size_t npix, float* y, float* i, float* q) {
    if (!npix) return SSW_OK;
    if (i && q) rgb8_to_yiq_kernel<true><<<sgrid(npix), 256, 0, st>>>(rgb, npix, y, i, q);
    else        rgb8_to_yiq_kernel<false><<<sgrid(npix), 256, 0, st>>>(rgb, npix, y, nullptr, nullptr);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}
int launch_u16_to_f32(hipStream_t st, const uint16_t* in, size_t n, float* out) {
    if (n == 0) return SSW_OK;
    u16_to_f32_kernel<<<sgrid(n), 256, 0, st>>>(in, n, out);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}
int launch_f32_to_u16(hipStream_t st, const float* in, size_t n, uint16_t* out) {
    if (n == 0) return SSW_OK;
    f32_to_u16_kernel<<<sgrid(n), 256, 0, st>>>(in, n, out);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}
int launch_rgb16_to_yiq(hipStream_t st, const uint16_t* rgb, size_t npix, float* y, float* i, float* q) {
    if (npix == 0) return SSW_OK;
    if (i && q) rgb16_to_yiq_kernel<true><<<sgrid(npix), 256, 0, st>>>(rgb, npix, y, i, q);
    else        rgb16_to_yiq_kernel<false><<<sgrid(npix), 256, 0, st>>>(rgb, npix, y, nullptr, nullptr);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}
int launch_yiq_to_rgb8(hipStream_t st, const float* y, const float* i, const float* q, size_t npix, uint8_t* rgb) {
    if (!npix) return SSW_OK;
    yiq_to_rgb8_kernel<<<sgrid(npix), 256, 0, st>>>(y, i, q, npix, rgb);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// ---------------------------------------------------------------------------------------------
// CatmullRom resize.  Tap tables are built on the host in f32 exactly as the crate does
// (this file is compiled with -ffp-contract=off for host and device).
// ---------------------------------------------------------------------------------------------
static float catmullrom_kernel_host(float x) {           // bc_cubic_spline(x, b = 0, c = 0.5)
    const float b = 0.0f, c = 0.5f;
    const float a = std::fabs(x);
    float k;
    if (a < 1.0f)
        k = (12.0f - 9.0f * b - 6.0f * c) * (a * a * a) + (-18.0f + 12.0f * b + 6.0f * c) * (a * a) + (6.0f - 2.0f * b);
    else if (a < 2.0f)
        k = (-b - 6.0f * c) * (a * a * a) + (6.0f * b + 30.0f * c) * (a * a) + (-12.0f * b - 48.0f * c) * a + (8.0f * b + 24.0f * c);
    else
        k = 0.0f;
    return k / 6.0f;
}

void build_resize_taps(size_t in_len, size_t out_len, ResizeTaps& t) {
    const float ratio = (float)in_len / (float)out_len;
    const float sratio = ratio < 1.0f ? 1.0f : ratio;
    const float src_support = 2.0f * sratio;
    t.max_taps = (uint32_t)(2.0f * src_support) + 3;
    t.left.assign(out_len, 0);
    t.count.assign(out_len, 0);
    t.weights.assign(out_len * t.max_taps, 0.0f);
    for (size_t o = 0; o < out_len; ++o) {
        float inputc = ((float)o + 0.5f) * ratio;
        long long left = (long long)std::floor(inputc - src_support);
        if (left < 0) left = 0;
        if (left > (long long)in_len - 1) left = (long long)in_len - 1;
        long long right = (long long)std::ceil(inputc + src_support);
        if (right < left + 1) right = left + 1;
        if (right > (long long)in_len) right = (long long)in_len;
        inputc = inputc - 0.5f;
        uint32_t n = 0;
        volatile float sum = 0.0f;
        float* w = &t.weights[o * t.max_taps];
        for (long long i = left; i < right && n < t.max_taps; ++i) {
            const float v = catmullrom_kernel_host(((float)i - inputc) / sratio);
            w[n++] = v;
            sum = sum + v;
        }
        for (uint32_t i = 0; i < n; ++i) w[i] = w[i] / sum;
        t.left[o] = (uint32_t)left;
        t.count[o] = n;
    }
}

// vertical pass: tmp[f][oy][e] = sum_i in[f][left+i][e] * w[i], e = x*3 + c (f32, sequential sum).
// VEC = 4: four consecutive bytes per thread (one 32-bit load per tap) when the row length allows.
template <int VEC>
__global__ __launch_bounds__(256) void resize_vertical_kernel(const uint8_t* __restrict__ in, unsigned row_elems,
                                                              unsigned h, unsigned nh, const uint32_t* __restrict__ left,
                                                              const uint32_t* __restrict__ count,
                                                              const float* __restrict__ weights, unsigned max_taps,
                                                              float* __restrict__ tmp) {
    const unsigned oy = blockIdx.y, f = blockIdx.z;
    const unsigned l = left[oy], n = count[oy];
    const float* __restrict__ w = weights + (size_t)oy * max_taps;
    const uint8_t* __restrict__ src = in + ((size_t)f * h + l) * row_elems;
    float* __restrict__ dst = tmp + ((size_t)f * nh + oy) * row_elems;
    for (unsigned e = (blockIdx.x * blockDim.x + threadIdx.x) * VEC; e < row_elems; e += gridDim.x * blockDim.x * VEC) {
        if (VEC == 4) {
            float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
            for (unsigned i = 0; i < n; ++i) {
                const uint32_t v = *reinterpret_cast<const uint32_t*>(src + (size_t)i * row_elems + e);
                const float wi = w[i];
                t0 += (float)(v & 0xFF) * wi;
                t1 += (float)((v >> 8) & 0xFF) * wi;
                t2 += (float)((v >> 16) & 0xFF) * wi;
                t3 += (float)(v >> 24) * wi;
            }
            *reinterpret_cast<f32x4*>(dst + e) = (f32x4){t0, t1, t2, t3};
        } else {
            float t = 0.0f;
            for (unsigned i = 0; i < n; ++i) t += (float)src[(size_t)i * row_elems + e] * w[i];
            dst[e] = t;
        }
    }
}

// horizontal pass: out[row][ox][c] = round(clamp(sum_i tmp[row][left+i][c] * w[i], 0, 255)).
// One thread per output pixel (3 channels): the tap table entry is read once per pixel.
__global__ __launch_bounds__(256) void resize_horizontal_kernel(const float* __restrict__ tmp, unsigned w_in,
                                                                unsigned nw, unsigned rows,
                                                                const uint32_t* __restrict__ left,
                                                                const uint32_t* __restrict__ count,
                                                                const float* __restrict__ weights, unsigned max_taps,
                                                                uint8_t* __restrict__ out) {
    const unsigned ox = blockIdx.x * blockDim.x + threadIdx.x;
    if (ox >= nw) return;
    const unsigned l = left[ox], n = count[ox];
    const float* __restrict__ w = weights + (size_t)ox * max_taps;
    for (unsigned row = blockIdx.y; row < rows; row += gridDim.y) {
        const float* __restrict__ src = tmp + ((size_t)row * w_in + l) * 3;
        float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
        for (unsigned i = 0; i < n; ++i) {
            const float wi = w[i];
            t0 += src[3 * i + 0] * wi;
            t1 += src[3 * i + 1] * wi;
            t2 += src[3 * i + 2] * wi;
        }
        uint8_t* o = out + ((size_t)row * nw + ox) * 3;
        o[0] = (uint8_t)roundf(clampf(t0, 0.0f, 255.0f));
        o[1] = (uint8_t)roundf(clampf(t1, 0.0f, 255.0f));
        o[2] = (uint8_t)roundf(clampf(t2, 0.0f, 255.0f));
    }
}

// ---------------------------------------------------------------------------------------------
// Fused resize: one block = one tile of OYB x OXB output pixels of one frame.
//   1. the input pixels the tile's taps reach (rows [r0, r1) x bytes [a0, b1)) are loaded once into LDS,
//   2. vertical pass  LDS u8 -> LDS f32 strip (OYB rows), sequential f32 sum per element like the crate,
//   3. horizontal pass LDS f32 -> u8 pixels (clamp + round half away from zero), staged in LDS,
//   4. the tile's rows are written with 32-bit stores.
// The f32 intermediate of the two-pass kernels above (12 B per intermediate pixel written and re-read
// through HBM / L2, with byte-strided accesses on the narrow side) never leaves the CU.  Arithmetic and
// summation order are those of the two-pass kernels: bit-identical results.
// Needs 4-byte aligned rows (w * 3 % 4 == 0, nw * 3 % 4 == 0, OXB % 4 == 0).
// ---------------------------------------------------------------------------------------------
struct ResizeTile {
    unsigned oyb, oxb;            // output tile (powers of two, oxb >= 4)
    unsigned pitch;               // elements (bytes of s_in, floats of s_v) per LDS row, multiple of 4
    unsigned in_rows;             // LDS rows of the input tile
    unsigned tiles_x, tiles_y;
    unsigned oxb_log2;
};

typedef float rz_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4r __attribute__((ext_vector_type(4)));

// round(clamp(t, 0, 255)) with halves away from zero, as an integer -- exactly: scaling by 2^16 is exact, the
// conversion truncates, and floor(c * 2^16) still tells whether the fraction reaches 1/2
__device__ inline uint32_t resize_to_u8(float t) {
    const float c = __builtin_amdgcn_fmed3f(t, 0.0f, 255.0f);     // = clamp for the finite sums this sees
    return ((uint32_t)(c * 65536.0f) + 0x8000u) >> 16;
}

// vertical pass of one block: pieces of NW 32-bit words (4 NW bytes) per thread
template <int NW>
__device__ inline void resize_vertical_pieces(const unsigned char* s_in, float* s_v, const float* s_wv, const uint32_t* s_lv,
                                              const uint32_t* s_cv, unsigned r0, unsigned noy, unsigned pieces, unsigned vmax,
                                              unsigned pitch, unsigned tid) {
    typedef unsigned int uvec __attribute__((ext_vector_type(NW)));
    for (unsigned it = tid; it < noy * pieces; it += 256) {
        const unsigned j = it / pieces, ck = it - j * pieces;
        const unsigned n = s_cv[j];
        const unsigned char* col = s_in + (s_lv[j] - r0) * pitch + 4 * NW * ck;
        const float* wv = s_wv + j * vmax;
        rz_f32x2 t[2 * NW];
#pragma unroll
        for (int u = 0; u < 2 * NW; ++u) t[u] = (rz_f32x2){0.0f, 0.0f};
#pragma unroll 2
        for (unsigned i = 0; i < n; ++i) {
            const uvec v = *reinterpret_cast<const uvec*>(col + i * pitch);
            const float wi = wv[i];
            const rz_f32x2 ww = {wi, wi};
#pragma unroll
            for (int u = 0; u < NW; ++u) {
                const rz_f32x2 a = {(float)(v[u] & 0xFF), (float)((v[u] >> 8) & 0xFF)};
                const rz_f32x2 b = {(float)((v[u] >> 16) & 0xFF), (float)(v[u] >> 24)};
                t[2 * u] += a * ww;
                t[2 * u + 1] += b * ww;
            }
        }
        f32x4* o = reinterpret_cast<f32x4*>(s_v + j * pitch + 4 * NW * ck);
#pragma unroll
        for (int u = 0; u < NW; ++u) o[u] = (f32x4){t[2 * u][0], t[2 * u][1], t[2 * u + 1][0], t[2 * u + 1][1]};
    }
}

// HMODE (horizontal pass): 0 = one lane per (output pixel, channel) -- down-scaling along x, few outputs, long taps;
// 1 = one thread per aligned quad of output pixels that share their taps' positions (left, count <= 5: integer
// up-scaling) -- the <= 15 strip values are loaded once and feed 12 output bytes; 2 = one thread per output pixel.
template <int HMODE>
__global__ __launch_bounds__(256) void resize_fused_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                           unsigned w, unsigned h, unsigned nw, unsigned nh,
                                                           const uint32_t* __restrict__ vleft, const uint32_t* __restrict__ vcount,
                                                           const float* __restrict__ vweights, unsigned vmax,
                                                           const uint32_t* __restrict__ hleft, const uint32_t* __restrict__ hcount,
                                                           const float* __restrict__ hweights, unsigned hmax, ResizeTile tl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: s_v f32 [oyb][pitch] | s_wh f32 [hmax][oxb] (tap-major) | s_wv f32 [oyb][vmax] | meta u32 [2 oyb + 2 oxb] |
    //         s_in u8 [in_rows][pitch];  pitch is a multiple of 16: 16-byte LDS accesses throughout
    float* s_v = reinterpret_cast<float*>(smem);
    float* s_wh = s_v + (size_t)tl.oyb * tl.pitch;
    float* s_wv = s_wh + (size_t)tl.oxb * hmax;
    uint32_t* s_lv = reinterpret_cast<uint32_t*>(s_wv + (size_t)tl.oyb * vmax);
    uint32_t* s_cv = s_lv + tl.oyb;
    uint32_t* s_lh = s_cv + tl.oyb;
    uint32_t* s_ch = s_lh + tl.oxb;
    // (offset arithmetic on `smem` itself: a pointer rebuilt from an integer loses its LDS address space and every
    //  access through it becomes a flat load)
    const unsigned in_off = ((tl.oyb * tl.pitch + tl.oxb * hmax + tl.oyb * vmax + 2 * tl.oyb + 2 * tl.oxb) * 4 + 15) & ~15u;
    unsigned char* s_in = smem + in_off;
    unsigned char* s_out = s_in;                       // reused after the vertical pass: [oyb][oxb * 3]

    const unsigned tid = threadIdx.x, lane = tid & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned tx = blockIdx.x % tl.tiles_x, ty = blockIdx.x / tl.tiles_x, f = blockIdx.y;
    const unsigned oy0 = ty * tl.oyb, ox0 = tx * tl.oxb;
    const unsigned noy = nh - oy0 < tl.oyb ? nh - oy0 : tl.oyb;
    const unsigned nox = nw - ox0 < tl.oxb ? nw - ox0 : tl.oxb;

    // The tile's reach: left / right bounds grow with the output index, so it is first left .. last right
    // (block-uniform scalar loads; everything below is addressed from them).
    const unsigned r0 = vleft[oy0], r1 = vleft[oy0 + noy - 1] + vcount[oy0 + noy - 1];
    const unsigned b0 = hleft[ox0] * 3, b1 = (hleft[ox0 + nox - 1] + hcount[ox0 + nox - 1]) * 3;
    const unsigned a0 = b0 & ~3u;                                   // rows are 4-byte aligned: aligned word loads
    const unsigned words = (b1 - a0 + 3) / 4;
    const unsigned nrows = r1 - r0;
    {   // 1. tap tables and input tile -> LDS.  A block lives for a handful of memory latencies, so every global
        //    load that does not depend on another is issued before the first LDS write: the tables (<= 4 + 4 + 4
        //    values per thread) and the first six 16-byte pieces of the tile together, then the rest of the tile.
        float hw[4], vw[4];
        uint32_t meta[4] = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned i = tid + 256 * u;
            hw[u] = i < nox * hmax ? hweights[(size_t)ox0 * hmax + i] : 0.0f;
            vw[u] = i < noy * vmax ? vweights[(size_t)oy0 * vmax + i] : 0.0f;
        }
        if (tid < noy) { meta[0] = vleft[oy0 + tid]; meta[1] = vcount[oy0 + tid]; }
        if (tid < nox) { meta[2] = hleft[ox0 + tid]; meta[3] = hcount[ox0 + tid]; }
        const uint8_t* src = in + ((size_t)f * h + r0) * w * 3 + a0;
        const size_t row_bytes = (size_t)w * 3;
        const unsigned chunks = (words + 3) / 4;
        const unsigned avail = w * 3 - a0;                              // bytes from a0 to the end of the image row
        const unsigned total = nrows * chunks;
        constexpr int NB = 6;
        for (unsigned base = 0; base < total || base == 0; base += 256 * NB) {
            u32x4r v[NB];
            unsigned dst[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const unsigned it = base + u * 256 + tid;
                dst[u] = 0xFFFFFFFFu;
                if (it < total) {
                    const unsigned r = it / chunks, ck = it - r * chunks;
                    const uint8_t* p = src + r * row_bytes + 16 * ck;
                    dst[u] = r * tl.pitch + 16 * ck;
                    if (16 * ck + 16 <= avail) {
                        v[u] = *reinterpret_cast<const u32x4r*>(p);
                    } else {                                            // last piece of an image row: stay inside the row
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[u][e] = (16 * ck + 4 * e + 4 <= avail) ? *reinterpret_cast<const uint32_t*>(p + 4 * e) : 0u;
                    }
                }
            }
            if (base == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned i = tid + 256 * u;
                    if (i < nox * hmax) { const unsigned x = i / hmax, tp = i - x * hmax; s_wh[tp * tl.oxb + x] = hw[u]; }   // tap-major
                    if (i < noy * vmax) s_wv[i] = vw[u];
                }
                if (tid < noy) { s_lv[tid] = meta[0]; s_cv[tid] = meta[1]; }
                if (tid < nox) { s_lh[tid] = meta[2]; s_ch[tid] = meta[3]; }
            }
#pragma unroll
            for (int u = 0; u < NB; ++u)
                if (dst[u] != 0xFFFFFFFFu) *reinterpret_cast<u32x4r*>(s_in + dst[u]) = v[u];
        }
    }
    __syncthreads();
    // 2. vertical pass (vertical_sample of the crate): t = sum_i (float)in[left + i][e] * w[i], i ascending, mul and
    //    add rounded separately.  One thread = 16 (or, when that leaves half the block idle, 8) consecutive bytes of
    //    one output row: that many independent sums per LDS read.
    {
        // whole LDS rows, slack included: every strip element the horizontal pass may touch is then a finite sum
        const unsigned chunks16 = tl.pitch / 16;
        if (noy * chunks16 >= 192) resize_vertical_pieces<4>(s_in, s_v, s_wv, s_lv, s_cv, r0, noy, chunks16, vmax, tl.pitch, tid);
        else                       resize_vertical_pieces<2>(s_in, s_v, s_wv, s_lv, s_cv, r0, noy, tl.pitch / 8, vmax, tl.pitch, tid);
    }
    __syncthreads();
    // 3. horizontal pass (horizontal_sample): t = sum_i strip[left + i][c] * w[i]; clamp, round; staged in LDS
    const unsigned out_pitch = tl.oxb * 3;
    if (HMODE == 0) {
        for (unsigned j = wave; j < noy; j += 4)
            for (unsigned e = lane; e < nox * 3; e += 64) {
                const unsigned x = e / 3, c = e - 3 * x;
                const unsigned n = s_ch[x];
                const float* src = s_v + j * tl.pitch + (s_lh[x] * 3 - a0) + c;
                const float* wh = s_wh + x;
                float t = 0.0f;
#pragma unroll 4
                for (unsigned i = 0; i < n; ++i) t += src[3 * i] * wh[i * tl.oxb];
                s_out[j * out_pitch + e] = (unsigned char)resize_to_u8(t);
            }
    } else if (HMODE == 1) {
        const unsigned nq_log2 = tl.oxb_log2 - 2, nq = nox / 4;
        for (unsigned it = tid; it < (noy << nq_log2); it += 256) {
            const unsigned j = it >> nq_log2, q = it & ((1u << nq_log2) - 1);
            if (q >= nq) continue;
            const unsigned x0 = 4 * q;
            const float* src = s_v + j * tl.pitch + (s_lh[x0] * 3 - a0);
            // Five taps are read whatever n is: the tap tables are zero-padded beyond n (hmax >= 5 rows exist), and the
            // strip beyond the reach holds finite values (sums of bytes times weights), so 0 * value adds nothing.
            float p[5][3];
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int c = 0; c < 3; ++c) p[i][c] = src[3 * i + c];
            f32x4 wq[5];                                           // tap i of the quad's four outputs: one 16-byte read
#pragma unroll
            for (int i = 0; i < 5; ++i) wq[i] = *reinterpret_cast<const f32x4*>(s_wh + i * tl.oxb + x0);
            uint32_t by[12];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                rz_f32x2 t01 = {0.0f, 0.0f};
                float t2 = 0.0f;
#pragma unroll
                for (int i = 0; i < 5; ++i) {      // taps beyond n: 0 * 0 added to the sum leaves it unchanged
                    const float wi = wq[i][k];
                    t01 += (rz_f32x2){p[i][0], p[i][1]} * (rz_f32x2){wi, wi};
                    t2 += p[i][2] * wi;
                }
                by[3 * k + 0] = resize_to_u8(t01[0]);
                by[3 * k + 1] = resize_to_u8(t01[1]);
                by[3 * k + 2] = resize_to_u8(t2);
            }
            uint32_t* o = reinterpret_cast<uint32_t*>(s_out + j * out_pitch + 12 * q);
#pragma unroll
            for (int u = 0; u < 3; ++u) o[u] = by[4 * u] | (by[4 * u + 1] << 8) | (by[4 * u + 2] << 16) | (by[4 * u + 3] << 24);
        }
    } else {
        const unsigned nx_log2 = tl.oxb_log2;
        for (unsigned it = tid; it < (noy << nx_log2); it += 256) {
            const unsigned j = it >> nx_log2, x = it & ((1u << nx_log2) - 1);
            if (x >= nox) continue;
            const unsigned n = s_ch[x];
            const float* src = s_v + j * tl.pitch + (s_lh[x] * 3 - a0);
            const float* wh = s_wh + x;
            float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
#pragma unroll 4
            for (unsigned i = 0; i < n; ++i) {
                const float wi = wh[i * tl.oxb];
                t0 += src[3 * i + 0] * wi;
                t1 += src[3 * i + 1] * wi;
                t2 += src[3 * i + 2] * wi;
            }
            unsigned char* o = s_out + j * out_pitch + x * 3;
            o[0] = (unsigned char)resize_to_u8(t0);
            o[1] = (unsigned char)resize_to_u8(t1);
            o[2] = (unsigned char)resize_to_u8(t2);
        }
    }
    __syncthreads();
    {   // 4. tile rows -> global: 16-byte stores when the tile's rows start on 16-byte boundaries, else 4-byte ones
        const unsigned obytes = nox * 3;
        uint8_t* dst = out + (((size_t)f * nh + oy0) * nw + ox0) * 3;
        const size_t orow = (size_t)nw * 3;
        if ((obytes & 15) == 0 && (orow & 15) == 0 && ((ox0 * 3) & 15) == 0 && (out_pitch & 15) == 0) {
            const unsigned per_row = obytes / 16;
            for (unsigned it = tid; it < noy * per_row; it += 256) {
                const unsigned j = it / per_row, c = it - j * per_row;
                *reinterpret_cast<u32x4r*>(dst + j * orow + 16 * c) = *reinterpret_cast<const u32x4r*>(s_out + j * out_pitch + 16 * c);
            }
        } else {
            const unsigned owords = obytes / 4;
            for (unsigned j = wave; j < noy; j += 4)
                for (unsigned wd = lane; wd < owords; wd += 64)
                    *reinterpret_cast<uint32_t*>(dst + j * orow + 4 * wd) = *reinterpret_cast<const uint32_t*>(s_out + j * out_pitch + 4 * wd);
        }
    }
}

// tile shape: the cheapest one (input bytes loaded + vertical taps per output pixel) whose LDS footprint lets two
// blocks share a CU; spans = reach (in input samples) of 1, 2, 4, ... 128 consecutive outputs (DeviceTaps::span)
static bool pick_resize_tile(const DeviceTaps& vt, const DeviceTaps& ht, size_t w, size_t h, size_t nw, size_t nh, ResizeTile* out,
                             size_t* lds_bytes) {
    double best = 1e300;
    bool found = false;
    const double vtaps = (double)(vt.span[0] ? vt.span[0] : 1);
    for (int ey = 0; ey < 8; ++ey)
        for (int ex = 2; ex < 8; ++ex) {                                // OXB >= 4 (a multiple of 4)
            const unsigned oyb = 1u << ey, oxb = 1u << ex;
            if (oyb > 2 * nh || oxb > 2 * nw) continue;
            const unsigned rows = vt.span[ey], px = ht.span[ex];
            if (!rows || !px) continue;
            const unsigned pitch = (px * 3 + 3 + 6 + 15) / 16 * 16;     // + up to 3 bytes of alignment slack + 2 pixels the quad path may read past the reach; 16-byte LDS accesses
            const size_t out_tile = (size_t)oyb * oxb * 3;
            const size_t in_tile = (size_t)rows * pitch;
            const size_t lds = (size_t)oyb * pitch * 4 + ((size_t)oxb * ht.max_taps + (size_t)oyb * vt.max_taps) * 4 +
                               (2 * (size_t)oyb + 2 * (size_t)oxb) * 4 + (in_tile > out_tile ? in_tile : out_tile) + 16;
            if ((size_t)oxb * ht.max_taps > 1024 || (size_t)oyb * vt.max_taps > 1024) continue;   // tap tables: <= 4 values per thread
            if (lds > 78 * 1024) continue;                               // two blocks per CU (160 KB of LDS)
            const double cost = ((double)rows * pitch + (double)oyb * pitch * vtaps) / ((double)oyb * oxb);
            if (cost < best) {
                best = cost;
                found = true;
                *out = ResizeTile{oyb, oxb, pitch, rows, (unsigned)((nw + oxb - 1) / oxb), (unsigned)((nh + oyb - 1) / oyb), (unsigned)ex};
                *lds_bytes = lds;
            }
        }
    (void)w; (void)h;
    return found;
}

static bool resize_can_fuse(const uint8_t* in, size_t n_frames, size_t w, size_t h, size_t nw, size_t nh, const DeviceTaps& vt,
                            const DeviceTaps& ht, const uint8_t* out, ResizeTile* tl, size_t* lds) {
    const bool rows_aligned = (w * 3) % 4 == 0 && (nw * 3) % 4 == 0 &&
                              ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 3) == 0;
    return rows_aligned && n_frames <= 65535 && pick_resize_tile(vt, ht, w, h, nw, nh, tl, lds);
}

// f32 intermediate the two-pass fallback needs (0 when the fused kernel takes the call)
size_t resize_tmp_bytes(const uint8_t* in, size_t n_frames, size_t w, size_t h, size_t nw, size_t nh, const DeviceTaps& vt,
                        const DeviceTaps& ht, const uint8_t* out) {
    ResizeTile tl;
    size_t lds = 0;
    return resize_can_fuse(in, n_frames, w, h, nw, nh, vt, ht, out, &tl, &lds) ? 0 : n_frames * nh * w * 3 * sizeof(float);
}

int launch_resize_rgb8(hipStream_t st, const uint8_t* in, size_t n_frames, size_t w, size_t h, size_t nw, size_t nh,
                       const DeviceTaps& vt, const DeviceTaps& ht, float* tmp, uint8_t* out) {
    if (!n_frames) return SSW_OK;
    const unsigned row_elems = (unsigned)(w * 3);
    ResizeTile tl;
    size_t lds = 0;
    if (resize_can_fuse(in, n_frames, w, h, nw, nh, vt, ht, out, &tl, &lds)) {
        const dim3 grid(tl.tiles_x * tl.tiles_y, (unsigned)n_frames);
        {   // tiles above 64 KB of dynamic LDS need the per-device function attribute: set it once per device
            static std::atomic<bool> attr_set[64];           // two host threads, two contexts: no plain bools
            int dev = 0;
            SSW_HIP_CHECK(hipGetDevice(&dev));
            if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
                SSW_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(resize_fused_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
                SSW_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(resize_fused_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
                SSW_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(resize_fused_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
                if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
            }
        }
#define SSW_RESIZE_FUSED(MODE) resize_fused_kernel<MODE><<<grid, 256, lds, st>>>(in, out, (unsigned)w, (unsigned)h, (unsigned)nw, (unsigned)nh, \
        vt.left, vt.count, vt.weights, vt.max_taps, ht.left, ht.count, ht.weights, ht.max_taps, tl)
        if (nw < w) SSW_RESIZE_FUSED(0);
        else if (ht.quad_uniform) SSW_RESIZE_FUSED(1);
        else SSW_RESIZE_FUSED(2);
#undef SSW_RESIZE_FUSED
        SSW_HIP_CHECK(hipGetLastError());
        return SSW_OK;
    }
    const bool vec = (row_elems % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(tmp) & 15) == 0);
    if (vec) {
        dim3 gv((row_elems / 4 + 255) / 256, (unsigned)nh, (unsigned)n_frames);
        resize_vertical_kernel<4><<<gv, 256, 0, st>>>(in, row_elems, (unsigned)h, (unsigned)nh, vt.left, vt.count,
                                                      vt.weights, vt.max_taps, tmp);
    } else {
        dim3 gv((row_elems + 255) / 256, (unsigned)nh, (unsigned)n_frames);
        resize_vertical_kernel<1><<<gv, 256, 0, st>>>(in, row_elems, (unsigned)h, (unsigned)nh, vt.left, vt.count,
                                                      vt.weights, vt.max_taps, tmp);
    }
    SSW_HIP_CHECK(hipGetLastError());
    const unsigned rows = (unsigned)(n_frames * nh);
    dim3 gh((unsigned)((nw + 255) / 256), rows < 16384 ? rows : 16384);
    resize_horizontal_kernel<<<gh, 256, 0, st>>>(tmp, (unsigned)w, (unsigned)nw, rows, ht.left, ht.count, ht.weights,
                                                 ht.max_taps, out);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

}  // namespace ssw
