// rgb -> Y (I, Q) of four consecutive pixels of an interleaved RGB row (yiq.rs:177-186, the same arithmetic as color.hip /
// attack.hip): shared by the operand pre-passes that read the frames themselves (dct_pair_prep.hip, dct_pair_prep_light.hip,
// dct_pair_derived.hip).  In two steps -- the loads, and the arithmetic on what they returned -- so that a kernel can keep the
// next tile's pixels in flight while it works on this one.
#pragma once
#include "dct_pair_common.hpp"

namespace ssw {

__device__ inline float prep_dot3(float m0, float m1, float m2, float a, float b, float c) { return m0 * a + m1 * b + m2 * c; }

// the twelve samples of 4 consecutive pixels as they lie in memory: FMT SSW_PIX_F32: 3 x 16 bytes, U16: 3 x 8, U8: 3 x 4
template <int FMT> struct RawQuad { u32x4 w[3]; };
template <> struct RawQuad<SSW_PIX_U16> { u32x2 w[3]; };
template <> struct RawQuad<SSW_PIX_U8> { uint32_t w[3]; };

// 4 consecutive pixels starting at pixel x of a row (x % 4 == 0)
template <int FMT>
__device__ inline void load_raw4(const void* row_base, unsigned x, RawQuad<FMT>& q) {
    if (FMT == SSW_PIX_U16) {
        const u32x2* src = reinterpret_cast<const u32x2*>(static_cast<const uint16_t*>(row_base) + 3 * (size_t)x);
        auto& w = reinterpret_cast<RawQuad<SSW_PIX_U16>&>(q).w;
        w[0] = src[0]; w[1] = src[1]; w[2] = src[2];
    } else if (FMT == SSW_PIX_F32) {
        const u32x4* src = reinterpret_cast<const u32x4*>(static_cast<const float*>(row_base) + 3 * (size_t)x);
        auto& w = reinterpret_cast<RawQuad<SSW_PIX_F32>&>(q).w;
        w[0] = src[0]; w[1] = src[1]; w[2] = src[2];
    } else {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(row_base) + 3 * (size_t)x);
        auto& w = reinterpret_cast<RawQuad<SSW_PIX_U8>&>(q).w;
        w[0] = src[0]; w[1] = src[1]; w[2] = src[2];
    }
}

// Y (and I, Q) of the four pixels; into_rgb32f: f32 as it is, v / 255, v / 65535
template <int FMT, bool WITH_IQ>
__device__ inline void yiq_of_raw4(const RawQuad<FMT>& q, f32x4& y, f32x4& iv, f32x4& qv) {
    float r[4], g[4], b[4];
    if (FMT == SSW_PIX_U16) {
        const auto& w = reinterpret_cast<const RawQuad<SSW_PIX_U16>&>(q).w;
        const uint32_t wd[6] = {w[0][0], w[0][1], w[1][0], w[1][1], w[2][0], w[2][1]};
        float v[12];
#pragma unroll
        for (int e = 0; e < 6; ++e) {                               // into_rgb32f: v / 65535
            v[2 * e] = (float)(wd[e] & 0xFFFFu) / 65535.0f;
            v[2 * e + 1] = (float)(wd[e] >> 16) / 65535.0f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { r[e] = v[3 * e]; g[e] = v[3 * e + 1]; b[e] = v[3 * e + 2]; }
    } else if (FMT == SSW_PIX_F32) {
        const auto& w = reinterpret_cast<const RawQuad<SSW_PIX_F32>&>(q).w;
        float v[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) v[e] = __uint_as_float(w[e >> 2][e & 3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { r[e] = v[3 * e]; g[e] = v[3 * e + 1]; b[e] = v[3 * e + 2]; }
    } else {
        const auto& w = reinterpret_cast<const RawQuad<SSW_PIX_U8>&>(q).w;
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
        const uint8_t by[12] = {(uint8_t)w0, (uint8_t)(w0 >> 8), (uint8_t)(w0 >> 16), (uint8_t)(w0 >> 24),
                                (uint8_t)w1, (uint8_t)(w1 >> 8), (uint8_t)(w1 >> 16), (uint8_t)(w1 >> 24),
                                (uint8_t)w2, (uint8_t)(w2 >> 8), (uint8_t)(w2 >> 16), (uint8_t)(w2 >> 24)};
#pragma unroll
        for (int e = 0; e < 4; ++e) {                               // into_rgb32f: v / 255
            r[e] = (float)by[3 * e] / 255.0f; g[e] = (float)by[3 * e + 1] / 255.0f; b[e] = (float)by[3 * e + 2] / 255.0f;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        y[e] = prep_dot3(0.30f, 0.59f, 0.11f, r[e], g[e], b[e]);
        if (WITH_IQ) {
            iv[e] = prep_dot3(0.60f, -0.28f, -0.32f, r[e], g[e], b[e]);
            qv[e] = prep_dot3(0.21f, -0.52f, 0.31f, r[e], g[e], b[e]);
        }
    }
}

// both steps at once
template <int FMT, bool WITH_IQ>
__device__ inline void load_yiq4(const void* row_base, unsigned x, f32x4& y, f32x4& iv, f32x4& qv) {
    RawQuad<FMT> q;
    load_raw4<FMT>(row_base, x, q);
    yiq_of_raw4<FMT, WITH_IQ>(q, y, iv, qv);
}

}  // namespace ssw
