// RGB <-> YIQ conversion and the synthetic frame generator.  HBM-bound streaming kernels.
//
// rgb_to_yiq : From<&Rgb32FImage> for YIQ32FImage   (/root/reference/src/yiq.rs:177-186)
// yiq_to_rgb : From<&YIQ32FImage> for Rgb32FImage   (/root/reference/src/yiq.rs:187-197)
// Arithmetic follows Matrix3x3::product / product_clamp (src/yiq.rs:131-147) exactly:
// (m0*v0 + m1*v1) + m2*v2 with every op rounded to f32 (this file is built with
// -ffp-contract=off, so no FMA is formed), constants from src/yiq.rs:157-159 / :163-165.
//
// Layout: interleaved RGB [pixel][3] f32 (12 B/px) <-> three planar f32 images (4 B/px each).
// Each thread handles 4 consecutive pixels: 3 x 16-B loads/stores on the interleaved side,
// 1 x 16-B per plane on the planar side.
#include "ssw_internal.hpp"

namespace ssw {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline float dot3(float m0, float m1, float m2, float a, float b, float c) {
    return m0 * a + m1 * b + m2 * c;
}
// f32::clamp (src/yiq.rs:139-147): NaN passes through
__device__ inline float clamp01(float x) {
    if (x < 0.0f) return 0.0f;
    if (x > 1.0f) return 1.0f;
    return x;
}

__device__ inline void rgb2yiq_px(float r, float g, float b, float& y, float& i, float& q) {
    y = dot3(0.30f, 0.59f, 0.11f, r, g, b);
    i = dot3(0.60f, -0.28f, -0.32f, r, g, b);
    q = dot3(0.21f, -0.52f, 0.31f, r, g, b);
}
__device__ inline void yiq2rgb_px(float y, float i, float q, float& r, float& g, float& b) {
    r = clamp01(dot3(1.0f, 0.948262f, 0.624013f, y, i, q));
    g = clamp01(dot3(1.0f, -0.276066f, -0.639810f, y, i, q));
    b = clamp01(dot3(1.0f, -1.105450f, 1.729860f, y, i, q));
}

template <bool WITH_IQ>
__global__ __launch_bounds__(256) void rgb_to_yiq_kernel(const float* __restrict__ rgb, size_t npix,
                                                         float* __restrict__ y, float* __restrict__ ip,
                                                         float* __restrict__ qp) {
    const size_t nquad = npix / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < nquad; t += stride) {
        const f32x4* src = reinterpret_cast<const f32x4*>(rgb + 12 * t);
        const f32x4 v0 = src[0], v1 = src[1], v2 = src[2];
        float yy[4], ii[4], qq[4];
        rgb2yiq_px(v0[0], v0[1], v0[2], yy[0], ii[0], qq[0]);
        rgb2yiq_px(v0[3], v1[0], v1[1], yy[1], ii[1], qq[1]);
        rgb2yiq_px(v1[2], v1[3], v2[0], yy[2], ii[2], qq[2]);
        rgb2yiq_px(v2[1], v2[2], v2[3], yy[3], ii[3], qq[3]);
        *reinterpret_cast<f32x4*>(y + 4 * t) = (f32x4){yy[0], yy[1], yy[2], yy[3]};
        if (WITH_IQ) {
            *reinterpret_cast<f32x4*>(ip + 4 * t) = (f32x4){ii[0], ii[1], ii[2], ii[3]};
            *reinterpret_cast<f32x4*>(qp + 4 * t) = (f32x4){qq[0], qq[1], qq[2], qq[3]};
        }
    }
    // ragged tail (npix % 4 pixels)
    const size_t tail0 = nquad * 4;
    const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (g < npix - tail0) {
        const size_t p = tail0 + g;
        float yy, ii, qq;
        rgb2yiq_px(rgb[3 * p], rgb[3 * p + 1], rgb[3 * p + 2], yy, ii, qq);
        y[p] = yy;
        if (WITH_IQ) { ip[p] = ii; qp[p] = qq; }
    }
}

__global__ __launch_bounds__(256) void yiq_to_rgb_kernel(const float* __restrict__ y,
                                                         const float* __restrict__ ip,
                                                         const float* __restrict__ qp, size_t npix,
                                                         float* __restrict__ rgb) {
    const size_t nquad = npix / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < nquad; t += stride) {
        const f32x4 yy = *reinterpret_cast<const f32x4*>(y + 4 * t);
        const f32x4 ii = *reinterpret_cast<const f32x4*>(ip + 4 * t);
        const f32x4 qq = *reinterpret_cast<const f32x4*>(qp + 4 * t);
        float r[4], g[4], b[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) yiq2rgb_px(yy[e], ii[e], qq[e], r[e], g[e], b[e]);
        f32x4* dst = reinterpret_cast<f32x4*>(rgb + 12 * t);
        dst[0] = (f32x4){r[0], g[0], b[0], r[1]};
        dst[1] = (f32x4){g[1], b[1], r[2], g[2]};
        dst[2] = (f32x4){b[2], r[3], g[3], b[3]};
    }
    const size_t tail0 = nquad * 4;
    const size_t gidx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (gidx < npix - tail0) {
        const size_t p = tail0 + gidx;
        float r, g, b;
        yiq2rgb_px(y[p], ip[p], qp[p], r, g, b);
        rgb[3 * p] = r; rgb[3 * p + 1] = g; rgb[3 * p + 2] = b;
    }
}

static inline unsigned stream_grid(size_t work_items) {
    // memory-bound: cap at 256 CUs x 8 blocks and grid-stride the rest
    size_t blocks = (work_items + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    return (unsigned)blocks;
}
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// scalar fallbacks for buffers that are not 16-byte aligned (never the case for hipMalloc'd frames)
__global__ void rgb_to_yiq_scalar_kernel(const float* rgb, size_t npix, float* y, float* ip, float* qp) {
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
        float yy, ii, qq;
        rgb2yiq_px(rgb[3 * p], rgb[3 * p + 1], rgb[3 * p + 2], yy, ii, qq);
        y[p] = yy;
        if (ip) { ip[p] = ii; qp[p] = qq; }
    }
}
__global__ void yiq_to_rgb_scalar_kernel(const float* y, const float* ip, const float* qp, size_t npix, float* rgb) {
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
        float r, g, b;
        yiq2rgb_px(y[p], ip[p], qp[p], r, g, b);
        rgb[3 * p] = r; rgb[3 * p + 1] = g; rgb[3 * p + 2] = b;
    }
}

int launch_rgb_to_yiq(hipStream_t st, const float* rgb, size_t npix, float* y, float* i, float* q) {
    if (npix == 0) return SSW_OK;
    const bool with_iq = (i != nullptr) && (q != nullptr);
    const bool al = aligned16(rgb) && aligned16(y) && (!with_iq || (aligned16(i) && aligned16(q)));
    if (!al) {
        rgb_to_yiq_scalar_kernel<<<stream_grid(npix), 256, 0, st>>>(rgb, npix, y, with_iq ? i : nullptr, q);
    } else if (with_iq) {
        rgb_to_yiq_kernel<true><<<stream_grid(npix / 4 + 4), 256, 0, st>>>(rgb, npix, y, i, q);
    } else {
        rgb_to_yiq_kernel<false><<<stream_grid(npix / 4 + 4), 256, 0, st>>>(rgb, npix, y, nullptr, nullptr);
    }
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

int launch_yiq_to_rgb(hipStream_t st, const float* y, const float* i, const float* q, size_t npix,
                      float* rgb) {
    if (npix == 0) return SSW_OK;
    const bool al = aligned16(rgb) && aligned16(y) && aligned16(i) && aligned16(q);
    if (al) yiq_to_rgb_kernel<<<stream_grid(npix / 4 + 4), 256, 0, st>>>(y, i, q, npix, rgb);
    else    yiq_to_rgb_scalar_kernel<<<stream_grid(npix), 256, 0, st>>>(y, i, q, npix, rgb);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// ---------------------------------------------------------------------------------------------
// Synthetic frames: bit-identical twin of the test oracle's frame generator (integer hash ->
// exact float ops only, contraction off on both sides).  7 octaves of bilinear value noise
// (cell 256..4 px, amplitude 2^-o) + 2 % white noise, per channel.
// ---------------------------------------------------------------------------------------------
__device__ inline uint32_t h32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ inline float u01(uint32_t h) { return (float)(h >> 8) * 0x1p-24f; }
__device__ inline float lattice(uint32_t base, uint32_t ix, uint32_t iy) {
    return u01(h32(base ^ (ix * 0x9E3779B1U + iy * 0x85EBCA77U)));
}

__global__ __launch_bounds__(256) void synth_kernel(uint32_t seed, uint32_t first_frame, size_t n_frames,
                                                    unsigned w, unsigned h, float* __restrict__ rgb) {
    const float norm = 1.0f / 2.004375f;
    const size_t per_frame = (size_t)w * h * 3;
    const size_t total = per_frame * n_frames;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const uint32_t frame = (uint32_t)(e / per_frame);
        const size_t in_frame = e % per_frame;
        const uint32_t ch = (uint32_t)(in_frame % 3);
        const size_t pix = in_frame / 3;
        const uint32_t x = (uint32_t)(pix % w), y = (uint32_t)(pix / w);
        const uint32_t fbase = h32(seed * 0x9E3779B1U + (first_frame + frame));
        float acc = 0.0f;
#pragma unroll
        for (uint32_t o = 0; o < 7; ++o) {
            const uint32_t base = h32(fbase ^ (ch * 0x632BE5ABU + o * 0x2545F491U + 1U));
            const uint32_t shift = 8 - o, cell = 256U >> o;
            const uint32_t ix = x >> shift, iy = y >> shift;
            const float inv = 1.0f / (float)cell;
            const float fx = (float)(x & (cell - 1)) * inv;
            const float fy = (float)(y & (cell - 1)) * inv;
            const float v00 = lattice(base, ix, iy), v10 = lattice(base, ix + 1, iy);
            const float v01 = lattice(base, ix, iy + 1), v11 = lattice(base, ix + 1, iy + 1);
            const float top = v00 + fx * (v10 - v00);
            const float bot = v01 + fx * (v11 - v01);
            const float val = top + fy * (bot - top);
            const float amp = 1.0f / (float)(1U << o);
            acc = acc + amp * val;
        }
        const uint32_t wbase = h32(fbase ^ (ch * 0x632BE5ABU + 0x7F4A7C15U));
        acc = acc + 0.02f * lattice(wbase, x, y);
        rgb[e] = acc * norm;
    }
}

int launch_synth(hipStream_t st, uint32_t seed, uint32_t first_frame, size_t n_frames, size_t w,
                 size_t h, float* rgb) {
    if (n_frames == 0 || w == 0 || h == 0) return SSW_OK;
    synth_kernel<<<4096, 256, 0, st>>>(seed, first_frame, n_frames, (unsigned)w, (unsigned)h, rgb);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

}  // namespace ssw
