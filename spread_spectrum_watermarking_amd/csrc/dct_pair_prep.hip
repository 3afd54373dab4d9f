// HBM-bound pre-passes of the operand-ready GEMMs (dct_pair_f64.hip / dct_pair_f32.hip): they write
// the image operand of a pass once, already folded (forward) or split (inverse), in the precision the
// MFMA consumes (T = double: every sum exact; T = float: one rounding per sum) and in the k-blocked
// layout (dct_pair_common.hpp).  Also the half bases in that layout.
#include "dct_pair_split.hpp"
#include "dct_pair_colops.hpp"
#include "dct_pair_yiq_load.hpp"

#include <cstdlib>

namespace ssw {

// ---------------------------------------------------------------------------------------------
// Half bases in the k-blocked layout: [Kp / 8][n / 2][8], same values as make_half_basis_f64_kernel.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void make_half_basis_blocked_kernel(size_t n, bool inverse, int parity, size_t kpad, T* out) {
    const size_t nh = n / 2, total = nh * kpad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t o = i / kpad, s = i % kpad;
        double v = 0.0;
        if (s < nh) {
            const size_t freq = inverse ? 2 * s + parity : 2 * o + parity;
            const size_t pos = inverse ? o : s;
            unsigned long long a = (unsigned long long)freq * (2ull * pos + 1ull);
            a %= 4ull * n;
            const double c = cospi((double)a / (double)(2ull * n));
            v = !inverse ? 2.0 * c : (freq == 0 ? 0.25 : 0.5 * c);
        }
        out[blk_index<T>(o, (unsigned)s, nh)] = (T)v;
    }
}

size_t dct_pair_kpad(bool f64, size_t n) { return f64 ? pair_kpad<double>(n) : pair_kpad<float>(n); }

int launch_make_half_basis_blocked(hipStream_t st, bool f64, size_t n, bool inverse, int parity, void* out) {
    const size_t kp = dct_pair_kpad(f64, n), total = (n / 2) * kp;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (f64) make_half_basis_blocked_kernel<double><<<blocks ? blocks : 1, 256, 0, st>>>(n, inverse, parity, kp, (double*)out);
    else     make_half_basis_blocked_kernel<float><<<blocks ? blocks : 1, 256, 0, st>>>(n, inverse, parity, kp, (float*)out);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// ---------------------------------------------------------------------------------------------
// Split odd half.  The odd frequencies of a length-N transform are a DCT-IV of the M = N/2 differences d[n]:
//   X[k] = sum_{n<M} d[n] cos(pi (2k+1)(2n+1) / (4M)),  k < M           (frequency 2k+1; inverse: the same sum with the
//                                                                         roles of k and n exchanged -- the matrix is symmetric)
// Pairing n with M-1-n and splitting the angle of an even k = 2j into pi j (2n+1)/M + psi_n, psi_n = pi (2n+1)/(4M):
//   a[n] =  d[n] cos psi_n + d[M-1-n] sin psi_n,   b[n] = -d[n] sin psi_n + d[M-1-n] cos psi_n        (n < M/2: a rotation)
//   A[j] = sum_n a[n] cos(pi j (2n+1)/M),  B[j] = sum_n b[n] sin(pi j (2n+1)/M)                       (j = 0 .. M/2)
//   X[2j] = A[j] + B[j],   X[2j-1] = A[j] - B[j]
// A is a DCT-II and B a DST-II of length M/2; both fold once more (n <-> M/2-1-n, exact additions):
//   j = 2i   :  A = sum_{n<M/4} (a[n] + a[M/2-1-n]) cos(..),   B = sum (b[n] - b[M/2-1-n]) sin(..)    operands AS, BD
//   j = 2i+1 :  A = sum_{n<M/4} (a[n] - a[M/2-1-n]) cos(..),   B = sum (b[n] + b[M/2-1-n]) sin(..)    operands AD, BS
// i.e. two GEMM launches with K = M/4 = N/8 and N/8 (+1) output pairs instead of one with K = N/2 and N/4 pairs:
// a quarter of the multiply-adds.  The only inexact step before the MFMA sums is the rotation (two f64 products and
// one sum per operand element, relative error 2^-52 of |d|): the folded operands of the other launches stay exact.
// Bases, k-blocked like the half bases: rows i of scale * cos / sin(2 pi j (2n+1) / N), j = 2i (E, N/8 + 1 rows) or
// 2i + 1 (O, N/8 rows), n < N/8; scale 2 forward, 1/2 inverse as in make_half_basis_blocked_kernel.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void make_split_basis_blocked_kernel(size_t n, bool inverse, int which /*0 cosE, 1 sinE, 2 cosO, 3 sinO, 4 sinE with row 0 := row n/8*/, size_t kpad,
                                                size_t rows, T* out) {
    const size_t ktrue = n / 8, total = rows * kpad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t o = i / kpad, s = i % kpad;
        double v = 0.0;
        if (s < ktrue) {
            const unsigned long long j = (which & 2) ? 2ull * o + 1ull : (which == 4 && o == 0) ? 2ull * (n / 8) : 2ull * o;
            const unsigned long long a = (j * (2ull * s + 1ull)) % (unsigned long long)n;       // angle 2 pi a / n
            const double arg = (double)(2ull * a) / (double)n;
            const double c = ((which & 1) || which == 4) ? sinpi(arg) : cospi(arg);
            v = (inverse ? 0.5 : 2.0) * c;
        }
        out[blk_index<T>(o, (unsigned)s, rows)] = (T)v;
    }
}
// rotation table of a length-n axis: [0 .. n/4) cos psi, [n/4 .. n/2) sin psi, psi = pi (2 m + 1) / (2 n)
__global__ void make_rot_table_kernel(size_t n, double* out) {
    const size_t q = n / 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
        const double arg = (double)(2ull * i + 1ull) / (double)(2ull * n);
        out[i] = cospi(arg);
        out[q + i] = sinpi(arg);
    }
}

// first-level split through pair_rotate_kernel: any axis whose quarter length folds (rows and columns alike)
// tile width of the class-major plane orders of a line of length len (dct_pair_common.hpp); SSW_CLASS_TILE=0: one tile
// (the r3 order)
// Forward row passes of 1280 columns or more run at LEVEL 2 (r4b / r4c; ssw_pipeline.hip build_pass): every operand folds or
// rotates once more and all eight launches sum len/16 terms (240 at 4K: 79 % of peak against 84 % for len/8, at 2/3 of the
// multiply-adds).  Measured a gain from 1280 x 720 up (smaller: not measured) -- once the plane between the passes is
// class-major; in the natural order every launch writes 4-byte pieces 64 bytes apart (1080p before r4c: such launches took
// 1.0 ms for 16 GFLOP).  SSW_EFOLD_MIN: A/B switch (minimum length).
bool dct_pair_efold(size_t len) {
    const size_t mn = (size_t)tuning(TUNE_EFOLD_MIN);
    return dct_pair_can_deep_rows(len) && len >= mn;
}
// Inverse row passes of 1280 columns or more run at level 2 as well (r4c): the odd part's classes and the
// quarter-length even part fold / rotate once more (dct_pair_prep_staged.hip, prep16_inv_rows_l2_kernel).
// SSW_EFOLD_INV_MIN: A/B switch (minimum length).
bool dct_pair_efold_inv(size_t len) {
    const size_t mn = (size_t)tuning(TUNE_EFOLD_INV_MIN);
    return dct_pair_can_deep_inv_rows(len) && dct_pair_prep_staged_rows_ok() && len >= mn;      // (len % 128 == 0: can_deep_inv_rows)
}
// Column passes of 720 rows or more (a multiple of 16) run at level 2 in both directions (r4c; the staged pre-passes only):
// launches of K = H/16 = 135 at 4K run at 50 TFLOP/s against 64 for K = 270, but do half the multiply-adds (measured a gain
// from 1280 x 720 up).  SSW_EFOLD_COLS_MIN: A/B switch.
bool dct_pair_efold_cols(size_t h, size_t w, bool class_major) {
    const size_t mn = (size_t)tuning(TUNE_EFOLD_COLS_MIN);
    return dct_pair_can_deep_cols(h) && dct_pair_prep_staged_cols_ok(w, class_major) && h >= mn;
}
unsigned dct_pair_class_tile(size_t len) {
    const bool one_tile = tuning(TUNE_CLASS_TILE) == 0;
    return one_tile ? (unsigned)len : class_tile((unsigned)len);
}
bool dct_pair_can_split(size_t len, bool is_row) { (void)is_row; return len % 8 == 0 && len >= 128; }
size_t dct_pair_split_kpad(size_t len) { return pair_kpad<double>(len / 4); }
// doubles in the four split planes of a pass over n frames (the larger of the row and the column pass)
size_t dct_pair_split_elems(size_t n_frames, size_t w, size_t h) {
    const size_t a = n_frames * h * dct_pair_split_kpad(w), b = n_frames * w * dct_pair_split_kpad(h);
    return 4 * (a > b ? a : b);
}
size_t dct_pair_split_basis_rows(size_t len, int which) { return (which & 2) ? len / 8 : len / 8 + 1; }      // which 4 like 1

int launch_make_split_basis_blocked(hipStream_t st, size_t n, bool inverse, int which, double* out) {
    const size_t kp = dct_pair_split_kpad(n), rows = dct_pair_split_basis_rows(n, which), total = rows * kp;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    make_split_basis_blocked_kernel<double><<<blocks ? blocks : 1, 256, 0, st>>>(n, inverse, which, kp, rows, out);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}
int launch_make_rot_table(hipStream_t st, size_t n, double* out) {
    make_rot_table_kernel<<<(unsigned)((n / 4 + 255) / 256 ? (n / 4 + 255) / 256 : 1), 256, 0, st>>>(n, out);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// Rotation + fold of an odd operand plane P ([Kp / 8][L][8], positions n < M = len/2) into the four split planes
// ([K8 / 8][L][8], K8 = kpad(len/4), zero padded): one thread = one line x 8 consecutive e (one 64-byte piece of each
// output plane); consecutive threads take consecutive lines, so every piece access of a wave is one contiguous run.
// ALIGNED: M/2 is a multiple of 8 and the mirrored positions are whole pieces too (16-byte loads); otherwise the
// mirrored elements are fetched one by one.
template <bool ALIGNED>
__global__ __launch_bounds__(256) void pair_rotate_kernel(const double* __restrict__ P, const double* __restrict__ rot,
                                                         double* __restrict__ AS, double* __restrict__ BD,
                                                         double* __restrict__ AD, double* __restrict__ BS,
                                                         unsigned L, unsigned M, unsigned K8, unsigned line_blocks) {
    const unsigned line = (blockIdx.x % line_blocks) * 256 + threadIdx.x;
    const unsigned e0 = (blockIdx.x / line_blocks) * 8;
    if (line >= L || e0 >= K8) return;
    const unsigned Mh = M / 2, Mq = M / 4;
    double as[8], bd[8], ad[8], bs[8];
    auto ld = [&](unsigned pos) { return P[blk_index<double>(line, pos, L)]; };
    double d0[8], d1[8], d2[8], d3[8];          // d[e], d[M/2-1-e], d[M/2+e], d[M-1-e]
    if (ALIGNED && e0 + 8 <= Mq) {
        const double* p0 = P + blk_index<double>(line, e0, L);
        const double* p1 = P + blk_index<double>(line, Mh - 8 - e0, L);
        const double* p2 = P + blk_index<double>(line, Mh + e0, L);
        const double* p3 = P + blk_index<double>(line, M - 8 - e0, L);
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const f64x2 v0 = *reinterpret_cast<const f64x2*>(p0 + i), v1 = *reinterpret_cast<const f64x2*>(p1 + i);
            const f64x2 v2 = *reinterpret_cast<const f64x2*>(p2 + i), v3 = *reinterpret_cast<const f64x2*>(p3 + i);
            d0[i] = v0[0]; d0[i + 1] = v0[1];
            d1[7 - i] = v1[0]; d1[6 - i] = v1[1];
            d2[i] = v2[0]; d2[i + 1] = v2[1];
            d3[7 - i] = v3[0]; d3[6 - i] = v3[1];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned e = e0 + i;
            const bool ok = e < Mq;
            d0[i] = ok ? ld(e) : 0.0;
            d1[i] = ok ? ld(Mh - 1 - e) : 0.0;
            d2[i] = ok ? ld(Mh + e) : 0.0;
            d3[i] = ok ? ld(M - 1 - e) : 0.0;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned e = e0 + i;
        const unsigned ec = e < Mq ? e : 0, em = Mh - 1 - ec;          // n = e and its mirror n' = M/2 - 1 - e
        const double c = rot[ec], s = rot[Mh + ec], cm = rot[em], sm = rot[Mh + em];
        const double a = d0[i] * c + d3[i] * s, b = d3[i] * c - d0[i] * s;                 // n: partner M-1-n
        const double am = d1[i] * cm + d2[i] * sm, bm = d2[i] * cm - d1[i] * sm;           // n': partner M-1-n' = M/2 + e
        const bool ok = e < Mq;
        as[i] = ok ? a + am : 0.0;
        ad[i] = ok ? a - am : 0.0;
        bs[i] = ok ? b + bm : 0.0;
        bd[i] = ok ? b - bm : 0.0;
    }
    const size_t at = blk_index<double>(line, e0, L);
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        *reinterpret_cast<f64x2*>(AS + at + i) = (f64x2){as[i], as[i + 1]};
        *reinterpret_cast<f64x2*>(BD + at + i) = (f64x2){bd[i], bd[i + 1]};
        *reinterpret_cast<f64x2*>(AD + at + i) = (f64x2){ad[i], ad[i + 1]};
        *reinterpret_cast<f64x2*>(BS + at + i) = (f64x2){bs[i], bs[i + 1]};
    }
}

// P: the odd operand plane of a length-`len` axis (kpad(len) wide, `lines` lines) -> sp: four consecutive planes
// AS | BD | AD | BS of lines * K8 doubles each
int launch_dct_pair_rotate(hipStream_t st, const double* p, const double* rot, double* sp, size_t lines, size_t len) {
    if (lines == 0) return SSW_OK;
    if (lines > 0xFFFFFFFFull || len % 8 != 0) return SSW_ERR_BAD_DIMS;
    const unsigned L = (unsigned)lines, M = (unsigned)(len / 2), K8 = (unsigned)dct_pair_split_kpad(len);
    const unsigned line_blocks = (L + 255) / 256;
    const unsigned long long nblk = (unsigned long long)line_blocks * (K8 / 8);
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    const size_t plane = (size_t)L * K8;
    if ((M / 2) % 8 == 0) pair_rotate_kernel<true><<<(unsigned)nblk, 256, 0, st>>>(p, rot, sp, sp + plane, sp + 2 * plane, sp + 3 * plane, L, M, K8, line_blocks);
    else                  pair_rotate_kernel<false><<<(unsigned)nblk, 256, 0, st>>>(p, rot, sp, sp + plane, sp + 2 * plane, sp + 3 * plane, L, M, K8, line_blocks);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// ---------------------------------------------------------------------------------------------
// Pre-passes (HBM-bound): f32 plane -> the f64 operand planes of the pass, k-blocked.
// One folding level:   forward  O1 = S, O2 = D;   inverse  O1 = E (even coefficients), O2 = O (odd)
// ---------------------------------------------------------------------------------------------
// Row pass: line = image row, k along the row.  Block = 32 lines x 32 k; thread = 4 consecutive k of
// one line: 128-byte read runs per line, 512-byte write runs per k-block.
template <typename T, bool INVERSE>
__global__ __launch_bounds__(256) void pair_prep_rows_kernel(const float* __restrict__ X, T* __restrict__ O1,
                                                            T* __restrict__ O2, unsigned rows, unsigned W, unsigned Kp,
                                                            unsigned tiles_k) {
    const unsigned Nh = W / 2;
    const unsigned s = (blockIdx.x % tiles_k) * 32 + (threadIdx.x & 7) * 4;
    const unsigned row = (blockIdx.x / tiles_k) * 32 + (threadIdx.x >> 3);
    if (row >= rows || s >= Kp) return;
    vec4_t<T> a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    if (s < Nh) {                                                 // Nh % 4 == 0
        const float* x = X + (size_t)row * W;
        if (!INVERSE) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(x + s);
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (W - 4 - s));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = (T)u[e] + (T)v[3 - e];
                b[e] = (T)u[e] - (T)v[3 - e];
            }
        } else {
            const f32x4 u = *reinterpret_cast<const f32x4*>(x + 2 * s);
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + 2 * s + 4);
            a = (vec4_t<T>){(T)u[0], (T)u[2], (T)v[0], (T)v[2]};
            b = (vec4_t<T>){(T)u[1], (T)u[3], (T)v[1], (T)v[3]};
        }
    }
    *reinterpret_cast<vec4_t<T>*>(O1 + blk_index<T>(row, s, rows)) = a;
    *reinterpret_cast<vec4_t<T>*>(O2 + blk_index<T>(row, s, rows)) = b;
}

// Column pass: line = (frame, column), k along the image rows: fold / split + transpose through LDS.
// Block tile: 32 k x 64 columns; written as 4 KB runs (64 lines x one 64-byte k-block piece).
template <typename T, bool INVERSE>
__global__ __launch_bounds__(256) void pair_prep_cols_kernel(const float* __restrict__ IN, T* __restrict__ O1,
                                                            T* __restrict__ O2, unsigned W, unsigned H, unsigned Kp,
                                                            unsigned n_frames, unsigned tiles_k, unsigned tiles_c) {
    __shared__ T s1[64][33];
    __shared__ T s2[64][33];
    const unsigned Hh = H / 2;
    const unsigned z = blockIdx.x / (tiles_k * tiles_c);
    const unsigned tt = blockIdx.x % (tiles_k * tiles_c);
    const unsigned k0 = (tt % tiles_k) * 32, c0 = (tt / tiles_k) * 64;
    const float* __restrict__ P = IN + (size_t)z * H * W;
    const unsigned tid = threadIdx.x;
    {
        const unsigned kr = tid >> 4, cq = (tid & 15) * 4;       // 16 k-rows per sweep, 4 columns per thread
        unsigned c = c0 + cq;
        c = c + 4 <= W ? c : W - 4;                               // W % 4 == 0; duplicates are never written out
#pragma unroll
        for (int sw = 0; sw < 2; ++sw) {
            const unsigned kl = kr + 16 * sw, k = k0 + kl;
            f32x4 u = {0.f, 0.f, 0.f, 0.f}, v = {0.f, 0.f, 0.f, 0.f};
            if (k < Hh) {
                const unsigned ra = INVERSE ? 2 * k : k, rb = INVERSE ? 2 * k + 1 : H - 1 - k;
                u = *reinterpret_cast<const f32x4*>(P + (size_t)ra * W + c);
                v = *reinterpret_cast<const f32x4*>(P + (size_t)rb * W + c);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1[cq + e][kl] = INVERSE ? (T)u[e] : (T)u[e] + (T)v[e];
                s2[cq + e][kl] = INVERSE ? (T)v[e] : (T)u[e] - (T)v[e];
            }
        }
    }
    __syncthreads();
    {
        const unsigned cl = tid & 63, kq = (tid >> 6) * 8;        // one 64-byte k-block piece of one column per thread
        const unsigned c = c0 + cl;
        if (c < W && k0 + kq < Kp) {                              // Kp % 8 == 0
            const size_t at = blk_index<T>((size_t)z * W + c, k0 + kq, (size_t)n_frames * W);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                *reinterpret_cast<vec2_t<T>*>(O1 + at + e) = (vec2_t<T>){s1[cl][kq + e], s1[cl][kq + e + 1]};
                *reinterpret_cast<vec2_t<T>*>(O2 + at + e) = (vec2_t<T>){s2[cl][kq + e], s2[cl][kq + e + 1]};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Two-level pre-passes: f32 plane -> (SS, SD, D) forward / (EE, EO, O) inverse in one sweep
// (12 B/px of HBM traffic; a separate second-level pass over S would make it 20).
//   forward, q < n/4:  S[q] = x[q] + x[n-1-q],  S' = x[n/2-1-q] + x[n/2+q];  SS = S + S',  SD = S - S'
//                      D[q] = x[q] - x[n-1-q],  D[n/2-1-q] = x[n/2-1-q] - x[n/2+q]
//   inverse, q < n/4:  EE[q] = c[4q],  EO[q] = c[4q+2],  O[2q] = c[4q+1],  O[2q+1] = c[4q+3]
// Q1, Q2: kq = half_basis_kpad(n/2) wide; P: kp = half_basis_kpad(n) wide; k-blocked, zero padded.
// ---------------------------------------------------------------------------------------------
template <typename T, bool INVERSE>
__global__ __launch_bounds__(256) void pair_prep4_rows_kernel(const float* __restrict__ X, T* __restrict__ Q1,
                                                             T* __restrict__ Q2, T* __restrict__ P,
                                                             unsigned rows, unsigned W, unsigned Kq, unsigned Kp,
                                                             unsigned tiles_q) {
    const unsigned Nh = W / 2, Nq = W / 4;
    const unsigned q = (blockIdx.x % tiles_q) * 32 + (threadIdx.x & 7) * 4;
    const unsigned row = (blockIdx.x / tiles_q) * 32 + (threadIdx.x >> 3);
    if (row >= rows || q >= Kq) return;
    vec4_t<T> a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
    if (q < Nq) {                                                 // Nq % 4 == 0
        const float* x = X + (size_t)row * W;
        if (!INVERSE) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(x + q);
            const f32x4 b = *reinterpret_cast<const f32x4*>(x + (Nh - 4 - q));
            const f32x4 c = *reinterpret_cast<const f32x4*>(x + (Nh + q));
            const f32x4 d = *reinterpret_cast<const f32x4*>(x + (W - 4 - q));
            vec4_t<T> dn, dm;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const T s1 = (T)a[e] + (T)d[3 - e], s2 = (T)b[3 - e] + (T)c[e];
                a1[e] = s1 + s2;
                a2[e] = s1 - s2;
                dn[e] = (T)a[e] - (T)d[3 - e];
                dm[3 - e] = (T)b[3 - e] - (T)c[e];
            }
            *reinterpret_cast<vec4_t<T>*>(P + blk_index<T>(row, q, rows)) = dn;
            *reinterpret_cast<vec4_t<T>*>(P + blk_index<T>(row, Nh - 4 - q, rows)) = dm;
        } else {
            f32x4 c[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) c[e] = *reinterpret_cast<const f32x4*>(x + 4 * (q + e));
#pragma unroll
            for (int e = 0; e < 4; ++e) { a1[e] = (T)c[e][0]; a2[e] = (T)c[e][2]; }
            T* o = P + blk_index<T>(row, 2 * q, rows);           // 2 q is a multiple of 8: one whole k-block piece
            *reinterpret_cast<vec4_t<T>*>(o) = (vec4_t<T>){(T)c[0][1], (T)c[0][3], (T)c[1][1], (T)c[1][3]};
            *reinterpret_cast<vec4_t<T>*>(o + 4) = (vec4_t<T>){(T)c[2][1], (T)c[2][3], (T)c[3][1], (T)c[3][3]};
        }
    }
    *reinterpret_cast<vec4_t<T>*>(Q1 + blk_index<T>(row, q, rows)) = a1;
    *reinterpret_cast<vec4_t<T>*>(Q2 + blk_index<T>(row, q, rows)) = a2;
    if (q == 0)
        for (unsigned z = Nh; z < Kp; z += 4) *reinterpret_cast<vec4_t<T>*>(P + blk_index<T>(row, z, rows)) = (vec4_t<T>){0, 0, 0, 0};
}

// Column pass: lines = (frame, column); block tile 32 q x 32 columns, transposed through LDS and
// written as 2 KB runs (32 lines x one 64-byte k-block piece).
template <typename T, bool INVERSE>
__global__ __launch_bounds__(256) void pair_prep4_cols_kernel(const float* __restrict__ IN, T* __restrict__ Q1,
                                                             T* __restrict__ Q2, T* __restrict__ P,
                                                             unsigned W, unsigned H, unsigned Kq, unsigned Kp,
                                                             unsigned n_frames, unsigned tiles_q, unsigned tiles_c) {
    __shared__ T sA[32][33], sB[32][33], sC[32][33], sD[32][33];
    const unsigned Hh = H / 2, Hq = H / 4;
    const unsigned z = blockIdx.x / (tiles_q * tiles_c);
    const unsigned tt = blockIdx.x % (tiles_q * tiles_c);
    const unsigned q0 = (tt % tiles_q) * 32, c0 = (tt / tiles_q) * 32;
    const float* __restrict__ Pz = IN + (size_t)z * H * W;
    const unsigned tid = threadIdx.x;
    {
        const unsigned qr = tid >> 3, cq = (tid & 7) * 4, q = q0 + qr;
        unsigned c = c0 + cq;
        c = c + 4 <= W ? c : W - 4;                               // W % 4 == 0; duplicates are never written out
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a, cc = a, d = a;
        if (q < Hq) {
            const unsigned ra = INVERSE ? 4 * q : q, rb = INVERSE ? 4 * q + 2 : Hh - 1 - q;
            const unsigned rc = INVERSE ? 4 * q + 1 : Hh + q, rd = INVERSE ? 4 * q + 3 : H - 1 - q;
            a = *reinterpret_cast<const f32x4*>(Pz + (size_t)ra * W + c);
            b = *reinterpret_cast<const f32x4*>(Pz + (size_t)rb * W + c);
            cc = *reinterpret_cast<const f32x4*>(Pz + (size_t)rc * W + c);
            d = *reinterpret_cast<const f32x4*>(Pz + (size_t)rd * W + c);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (!INVERSE) {
                const T s1 = (T)a[e] + (T)d[e], s2 = (T)b[e] + (T)cc[e];
                sA[cq + e][qr] = s1 + s2;
                sB[cq + e][qr] = s1 - s2;
                sC[cq + e][qr] = (T)a[e] - (T)d[e];        // D[q]
                sD[cq + e][qr] = (T)b[e] - (T)cc[e];       // D[H/2-1-q]
            } else {
                sA[cq + e][qr] = (T)a[e];                        // EE[q]
                sB[cq + e][qr] = (T)b[e];                        // EO[q]
                sC[cq + e][qr] = (T)cc[e];                       // O[2q]
                sD[cq + e][qr] = (T)d[e];                        // O[2q+1]
            }
        }
    }
    __syncthreads();
    {
        const unsigned cl = tid & 31, kq = (tid >> 5) * 4;         // 4 consecutive q of one column per thread
        const unsigned c = c0 + cl, q = q0 + kq;
        if (c < W && q < Kq) {
            const size_t line = (size_t)z * W + c, lines = (size_t)n_frames * W;
            *reinterpret_cast<vec4_t<T>*>(Q1 + blk_index<T>(line, q, lines)) = (vec4_t<T>){sA[cl][kq], sA[cl][kq + 1], sA[cl][kq + 2], sA[cl][kq + 3]};
            *reinterpret_cast<vec4_t<T>*>(Q2 + blk_index<T>(line, q, lines)) = (vec4_t<T>){sB[cl][kq], sB[cl][kq + 1], sB[cl][kq + 2], sB[cl][kq + 3]};
            if (q < Hq && q + 4 > Hq) {                             // H/4 not a multiple of 4: the last quad is partial
                for (unsigned e = 0; q + e < Hq; ++e) {
                    if (!INVERSE) {
                        P[blk_index<T>(line, q + e, lines)] = sC[cl][kq + e];
                        P[blk_index<T>(line, Hh - 1 - (q + e), lines)] = sD[cl][kq + e];
                    } else {
                        P[blk_index<T>(line, 2 * (q + e), lines)] = sC[cl][kq + e];
                        P[blk_index<T>(line, 2 * (q + e) + 1, lines)] = sD[cl][kq + e];
                    }
                }
            } else if (q < Hq) {
                if (!INVERSE) {
                    *reinterpret_cast<vec4_t<T>*>(P + blk_index<T>(line, q, lines)) = (vec4_t<T>){sC[cl][kq], sC[cl][kq + 1], sC[cl][kq + 2], sC[cl][kq + 3]};
                    *reinterpret_cast<vec4_t<T>*>(P + blk_index<T>(line, Hh - 4 - q, lines)) = (vec4_t<T>){sD[cl][kq + 3], sD[cl][kq + 2], sD[cl][kq + 1], sD[cl][kq]};
                } else {
                    T* o = P + blk_index<T>(line, 2 * q, lines);
                    *reinterpret_cast<vec4_t<T>*>(o) = (vec4_t<T>){sC[cl][kq], sD[cl][kq], sC[cl][kq + 1], sD[cl][kq + 1]};
                    *reinterpret_cast<vec4_t<T>*>(o + 4) = (vec4_t<T>){sC[cl][kq + 2], sD[cl][kq + 2], sC[cl][kq + 3], sD[cl][kq + 3]};
                }
            }
            if (q == 0)
                for (unsigned zz = Hh; zz < Kp; zz += 4) *reinterpret_cast<vec4_t<T>*>(P + blk_index<T>(line, zz, lines)) = (vec4_t<T>){0, 0, 0, 0};
        }
    }
}
// ---------------------------------------------------------------------------------------------
// First pass of a forward transform straight from the RGB frame: rgb -> Y (yiq.rs:177-186, the same
// arithmetic as color.hip / attack.hip) fused with the two-level row pre-pass above, so the Y plane is
// never written and re-read as f32 (8 B/px less HBM traffic per transform); I and Q planes are written
// for the writer, not for readers.
// ---------------------------------------------------------------------------------------------
// (prep_dot3 / load_yiq4: dct_pair_yiq_load.hpp, shared with dct_pair_prep_light.hip)

template <typename T, int FMT, bool WITH_IQ>
__global__ __launch_bounds__(256) void pair_prep4_rows_rgb_kernel(const void* __restrict__ RGB, T* __restrict__ Q1,
                                                                 T* __restrict__ Q2, T* __restrict__ P,
                                                                 float* __restrict__ IP, float* __restrict__ QP,
                                                                 unsigned rows, unsigned W, unsigned Kq, unsigned Kp,
                                                                 unsigned tiles_q) {
    const unsigned Nh = W / 2, Nq = W / 4;
    const unsigned q = (blockIdx.x % tiles_q) * 32 + (threadIdx.x & 7) * 4;
    const unsigned row = (blockIdx.x / tiles_q) * 32 + (threadIdx.x >> 3);
    if (row >= rows || q >= Kq) return;
    vec4_t<T> a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
    if (q < Nq) {                                                 // Nq % 4 == 0
        const void* base = static_cast<const char*>(RGB) + (size_t)row * W * 3 * (FMT == SSW_PIX_U8 ? 1 : FMT == SSW_PIX_U16 ? 2 : 4);
        const unsigned pos[4] = {q, Nh - 4 - q, Nh + q, W - 4 - q};
        f32x4 y[4], iv, qv;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            load_yiq4<FMT, WITH_IQ>(base, pos[u], y[u], iv, qv);
            if (WITH_IQ) {
                *reinterpret_cast<f32x4*>(IP + (size_t)row * W + pos[u]) = iv;
                *reinterpret_cast<f32x4*>(QP + (size_t)row * W + pos[u]) = qv;
            }
        }
        const f32x4 a = y[0], b = y[1], c = y[2], d = y[3];
        vec4_t<T> dn, dm;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const T s1 = (T)a[e] + (T)d[3 - e], s2 = (T)b[3 - e] + (T)c[e];
            a1[e] = s1 + s2;
            a2[e] = s1 - s2;
            dn[e] = (T)a[e] - (T)d[3 - e];
            dm[3 - e] = (T)b[3 - e] - (T)c[e];
        }
        *reinterpret_cast<vec4_t<T>*>(P + blk_index<T>(row, q, rows)) = dn;
        *reinterpret_cast<vec4_t<T>*>(P + blk_index<T>(row, Nh - 4 - q, rows)) = dm;
    }
    *reinterpret_cast<vec4_t<T>*>(Q1 + blk_index<T>(row, q, rows)) = a1;
    *reinterpret_cast<vec4_t<T>*>(Q2 + blk_index<T>(row, q, rows)) = a2;
    if (q == 0)
        for (unsigned z = Nh; z < Kp; z += 4) *reinterpret_cast<vec4_t<T>*>(P + blk_index<T>(row, z, rows)) = (vec4_t<T>){0, 0, 0, 0};
}

// ---------------------------------------------------------------------------------------------
// Three folding levels on a forward row pass (n % 32 == 0): the even part of the even part folds once
// more.  With S = fold(x) (length n/2), SS = fold(S) (n/4), SSS = fold(SS) (n/8):
//   R1 = SSS, R2 = SS - mirror (-> frequencies 0, 4 mod 8; width kpad(n/4))
//   M  = S - mirror            (-> frequencies 2 mod 4;    width kpad(n/2))
//   P  = x - mirror            (-> odd frequencies;        width kpad(n))
// One thread = one quad e < n/8 of one line: it reads the 8 quads of x that meet in it.  The source is
// an f32 plane or the interleaved RGB frame (Y formed on the fly, I / Q written for the writer).
// ---------------------------------------------------------------------------------------------
template <typename T, int SRC /*0 plane, 1 rgb f32, 2 rgb u8*/, bool WITH_IQ>
__global__ __launch_bounds__(256) void pair_prep8_rows_kernel(const void* __restrict__ SRCP, T* __restrict__ R1,
                                                             T* __restrict__ R2, T* __restrict__ M, T* __restrict__ P,
                                                             float* __restrict__ IP, float* __restrict__ QP,
                                                             unsigned rows, unsigned W, unsigned K8, unsigned Kq, unsigned Kp,
                                                             unsigned tiles_e) {
    const unsigned Nh = W / 2, Nq = W / 4, Ne = W / 8;
    const unsigned e0 = (blockIdx.x % tiles_e) * 32 + (threadIdx.x & 7) * 4;
    const unsigned row = (blockIdx.x / tiles_e) * 32 + (threadIdx.x >> 3);
    if (row >= rows || e0 >= K8) return;
    vec4_t<T> r1 = {0, 0, 0, 0}, r2 = {0, 0, 0, 0};
    if (e0 < Ne) {                                                // Ne % 4 == 0
        // quads of x, ascending positions; quad u mirrors quad 7 - u
        const unsigned pos[8] = {e0, Nq - 4 - e0, Nq + e0, Nh - 4 - e0, Nh + e0, 3 * Nq - 4 - e0, 3 * Nq + e0, W - 4 - e0};
        f32x4 x[8];
        if (SRC == 0) {
            const float* xr = static_cast<const float*>(SRCP) + (size_t)row * W;
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const f32x4*>(xr + pos[u]);
        } else {
            const void* base = SRC == 3 ? static_cast<const void*>(static_cast<const uint16_t*>(SRCP) + (size_t)row * W * 3)
                           : SRC == 2 ? static_cast<const void*>(static_cast<const uint8_t*>(SRCP) + (size_t)row * W * 3)
                                        : static_cast<const void*>(static_cast<const float*>(SRCP) + (size_t)row * W * 3);
            f32x4 iv, qv;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                load_yiq4<SRC - 1, WITH_IQ>(base, pos[u], x[u], iv, qv);
                if (WITH_IQ) {
                    *reinterpret_cast<f32x4*>(IP + (size_t)row * W + pos[u]) = iv;
                    *reinterpret_cast<f32x4*>(QP + (size_t)row * W + pos[u]) = qv;
                }
            }
        }
        // level 1: S and D1 at the positions of quads 0..3 (their mirrors are quads 7..4, reversed)
        vec4_t<T> S[4], D1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                S[u][i] = (T)x[u][i] + (T)x[7 - u][3 - i];
                D1[u][i] = (T)x[u][i] - (T)x[7 - u][3 - i];
            }
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<vec4_t<T>*>(P + blk_index<T>(row, pos[u], rows)) = D1[u];
        // level 2 on S (length n/2): quad 0 mirrors quad 3, quad 1 mirrors quad 2
        vec4_t<T> SS[2], D2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                SS[u][i] = S[u][i] + S[3 - u][3 - i];
                D2[u][i] = S[u][i] - S[3 - u][3 - i];
            }
#pragma unroll
        for (int u = 0; u < 2; ++u) *reinterpret_cast<vec4_t<T>*>(M + blk_index<T>(row, pos[u], rows)) = D2[u];
        // level 3 on SS (length n/4): quad 0 mirrors quad 1
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r1[i] = SS[0][i] + SS[1][3 - i];
            r2[i] = SS[0][i] - SS[1][3 - i];
        }
    }
    *reinterpret_cast<vec4_t<T>*>(R1 + blk_index<T>(row, e0, rows)) = r1;
    *reinterpret_cast<vec4_t<T>*>(R2 + blk_index<T>(row, e0, rows)) = r2;
    if (e0 == 0) {
        for (unsigned z = Nh; z < Kp; z += 4) *reinterpret_cast<vec4_t<T>*>(P + blk_index<T>(row, z, rows)) = (vec4_t<T>){0, 0, 0, 0};
        for (unsigned z = Nq; z < Kq; z += 4) *reinterpret_cast<vec4_t<T>*>(M + blk_index<T>(row, z, rows)) = (vec4_t<T>){0, 0, 0, 0};
    }
}

// ---------------------------------------------------------------------------------------------
// "Deep" forward pre-pass of a row pass (n % 64 == 0): everything the five GEMM launches of the pass consume, from one
// sweep over the source (see "Split odd half" above):
//   level 1   S = x + mirror, D = x - mirror                          (length n/2)
//   D   -> rotation + fold -> AS, BD, AD, BS        (n/8 each)        odd frequencies          (launches E, O)
//   level 2   SS = S + mirror, SD = S - mirror                        (length n/4)
//   SD  -> rotation + fold -> AS2, BD2, AD2, BS2    (n/16 each)       frequencies 2 mod 4      (launches E', O')
//   level 3   R1 = SS + mirror, R2 = SS - mirror    (n/8 each)        frequencies 0 and 4 mod 8 (launch R)
// One thread = 4 consecutive e < n/16 of one line: the 16 quads of x that meet in them (quad u mirrors quad 15 - u).
// Planes are k-blocked, K8 = kpad(n/4) (>= n/8) and K16 = kpad(n/8) (>= n/16) wide, zero padded.
// ---------------------------------------------------------------------------------------------
template <typename T, int SRC /*0 plane, 1 rgb f32, 2 rgb u8*/, bool WITH_IQ>
__global__ __launch_bounds__(256) void pair_prep16_rows_kernel(const void* __restrict__ SRCP, DeepPlanes dp,
                                                              const double* __restrict__ rot1, const double* __restrict__ rot2,
                                                              const double* __restrict__ rot3,
                                                              float* __restrict__ IP, float* __restrict__ QP,
                                                              unsigned rows, unsigned W, unsigned K8, unsigned K16, unsigned tiles_e, unsigned efold,
                                                              unsigned unit_h, unsigned unit_hup) {
    const unsigned Nh = W / 2, Nq = W / 4, N8 = W / 8, N16 = W / 16;
    const unsigned e0 = (blockIdx.x % tiles_e) * 32 + (threadIdx.x & 7) * 4;
    // operand line of this thread, and the image row it holds.  Natural order: the same number.  r5, fused forward
    // transform (unit_h = H != 0, level 2 only): the lines of a frame are ordered (unit of the COLUMN fold, line of the unit)
    // -- 16 * unit_hup lines per frame, unit_hup = H/16 rounded up to whole k-blocks of 8 -- so that a 16-line MFMA tile
    // of the row GEMM holds the sixteen rows that meet in one unit of the column pre-pass (dct_pair_colops.hpp).  A block's
    // 32 lines are then 32 rows of sixteen regions of the frame: the reads are 384-byte runs per row either way, the
    // stores stay 2 KB runs.  `rows` counts operand lines (the planes' line stride).
    const unsigned line = (blockIdx.x / tiles_e) * 32 + (threadIdx.x >> 3);
    if (line >= rows || e0 >= K16) return;
    unsigned row = line;
    bool pad_line = false;
    if (unit_h) {
        const unsigned lpf = 16 * unit_hup, z = line / lpf, rem = line - z * lpf;
        pad_line = (rem >> 4) >= unit_h / 16;
        row = z * unit_h + (pad_line ? 0u : col_unit_row(rem >> 4, rem & 15u, unit_h));
    }
    T* AD = static_cast<T*>(dp.ad); T* BS = static_cast<T*>(dp.bs);
    T* R1 = static_cast<T*>(dp.r1); T* R2 = static_cast<T*>(dp.r2);
    T* AS2 = static_cast<T*>(dp.as2); T* BD2 = static_cast<T*>(dp.bd2); T* AD2 = static_cast<T*>(dp.ad2); T* BS2 = static_cast<T*>(dp.bs2);
    auto put = [&](T* plane, unsigned k, const vec4_t<T>& v) { *reinterpret_cast<vec4_t<T>*>(plane + blk_index<T>(line, k, rows)) = v; };
    const vec4_t<T> zero = {0, 0, 0, 0};
    if (e0 >= N16 || pad_line) {                                  // padding of the n/16-wide planes; lines of the padding units
        put(AS2, e0, zero); put(BD2, e0, zero); put(AD2, e0, zero); put(BS2, e0, zero);
        if (efold) {
            void* const l2[12] = {dp.asp, dp.asm_, dp.bdp, dp.bdm, dp.oap, dp.obp, dp.oam, dp.obm, dp.r1p, dp.r1m, dp.r2a, dp.r2b};
#pragma unroll
            for (int a = 0; a < 12; ++a) put(static_cast<T*>(l2[a]), e0, zero);
        }
        return;
    }
    // quads of x, ascending positions inside a quad
    const unsigned pos[8] = {e0, N8 - 4 - e0, N8 + e0, Nq - 4 - e0, Nq + e0, 3 * N8 - 4 - e0, 3 * N8 + e0, Nh - 4 - e0};
    f32x4 x[16];
    if (SRC == 0) {
        const float* xr = static_cast<const float*>(SRCP) + (size_t)row * W;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = *reinterpret_cast<const f32x4*>(xr + pos[u]);
            x[15 - u] = *reinterpret_cast<const f32x4*>(xr + (W - 4 - pos[u]));
        }
    } else {
        const void* base = SRC == 3 ? static_cast<const void*>(static_cast<const uint16_t*>(SRCP) + (size_t)row * W * 3)
                           : SRC == 2 ? static_cast<const void*>(static_cast<const uint8_t*>(SRCP) + (size_t)row * W * 3)
                                    : static_cast<const void*>(static_cast<const float*>(SRCP) + (size_t)row * W * 3);
        f32x4 iv, qv;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const unsigned p = u < 8 ? pos[u] : W - 4 - pos[15 - u];
            load_yiq4<SRC - 1, WITH_IQ>(base, p, x[u], iv, qv);
            if (WITH_IQ) {
                *reinterpret_cast<f32x4*>(IP + (size_t)row * W + p) = iv;
                *reinterpret_cast<f32x4*>(QP + (size_t)row * W + p) = qv;
            }
        }
    }
    // level 1 at the positions of quads 0..7
    vec4_t<T> S[8], D[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            S[u][i] = (T)x[u][i] + (T)x[15 - u][3 - i];
            D[u][i] = (T)x[u][i] - (T)x[15 - u][3 - i];
        }
    // D (DCT-IV input of length n/2): the unit at e0 and its mirror unit at n/8 - 4 - e0.  AS and BD (class E: a DCT-II
    // and a DST-II of length n/8) fold once more with their mirrors -- element e0 + i meets element n/8 - 1 - (e0 + i), which
    // is element 3 - i of the mirror unit: exact additions -- into AS+ AS- BD+ BD- of length n/16 (launches E even / odd)
    {
        vec4_t<T> as, bd, ad, bs, asm_, bdm_, adm_, bsm_;
        split_unit<T>(D[0], D[3], D[4], D[7], rot1, e0, Nq, as, bd, ad, bs);
        split_unit<T>(D[1], D[2], D[5], D[6], rot1, N8 - 4 - e0, Nq, asm_, bdm_, adm_, bsm_);
        if (efold) {
            vec4_t<T> p, m, q, r;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                p[i] = as[i] + asm_[3 - i]; m[i] = as[i] - asm_[3 - i];
                q[i] = bd[i] + bdm_[3 - i]; r[i] = bd[i] - bdm_[3 - i];
            }
            put(static_cast<T*>(dp.asp), e0, p); put(static_cast<T*>(dp.asm_), e0, m);
            put(static_cast<T*>(dp.bdp), e0, q); put(static_cast<T*>(dp.bdm), e0, r);
            // Class O: X[8i+5] = T(i) + U(i), X[8i+3] = T(i) - U(i) with T the DCT-IV of AD and U the DST-IV of BS, length
            // L = n/8; U(i) = (-1)^i DCT-IV(reversed BS)(i).  One more rotation of the pairs (n, L-1-n), angle pi (2n+1) / (4L)
            // (the table of a length-n/4 axis), turns a DCT-IV of length L into a DCT-II of a and a DST-II of b, length L/2:
            //   a = d[n] cos + d[L-1-n] sin,  b = d[L-1-n] cos - d[n] sin;   even outputs A[j] + B[j], odd ones A[j] - B[j].
            // With (a, b) of AD plus / minus (a, b) of the reversed BS the two signs of (-1)^i sort themselves into two
            // launches of the class-E shape on the bases of E even: P -> 16j + 5, 16j - 5;  M -> 16j + 3, 16j - 3.
            const f64x4 c3 = *reinterpret_cast<const f64x4*>(rot3 + e0), s3 = *reinterpret_cast<const f64x4*>(rot3 + N16 + e0);
            vec4_t<T> oap, obp, oam, obm;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const T cc = (T)c3[i], ss = (T)s3[i];
                const T au = ad[i] * cc + adm_[3 - i] * ss, bu = adm_[3 - i] * cc - ad[i] * ss;
                const T av = bsm_[3 - i] * cc + bs[i] * ss, bv = bs[i] * cc - bsm_[3 - i] * ss;
                oap[i] = au + av; obp[i] = bu + bv;
                oam[i] = au - av; obm[i] = bu - bv;
            }
            put(static_cast<T*>(dp.oap), e0, oap); put(static_cast<T*>(dp.obp), e0, obp);
            put(static_cast<T*>(dp.oam), e0, oam); put(static_cast<T*>(dp.obm), e0, obm);
        } else {
            put(static_cast<T*>(dp.as), e0, as); put(static_cast<T*>(dp.bd), e0, bd);
            put(static_cast<T*>(dp.as), N8 - 4 - e0, asm_); put(static_cast<T*>(dp.bd), N8 - 4 - e0, bdm_);
            put(AD, e0, ad); put(BS, e0, bs);
            put(AD, N8 - 4 - e0, adm_); put(BS, N8 - 4 - e0, bsm_);
        }
    }
    // level 2 on S (length n/2): quad u mirrors quad 7 - u
    vec4_t<T> SS[4], SD[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            SS[u][i] = S[u][i] + S[7 - u][3 - i];
            SD[u][i] = S[u][i] - S[7 - u][3 - i];
        }
    // SD (DCT-IV input of length n/4): one unit at e0
    {
        vec4_t<T> as, bd, ad, bs;
        split_unit<T>(SD[0], SD[1], SD[2], SD[3], rot2, e0, N8, as, bd, ad, bs);
        put(AS2, e0, as); put(BD2, e0, bd); put(AD2, e0, ad); put(BS2, e0, bs);
    }
    // level 3 on SS (length n/4): quad 0 mirrors quad 3, quad 1 mirrors quad 2
    vec4_t<T> r1q[2], r2q[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r1q[u][i] = SS[u][i] + SS[3 - u][3 - i];
            r2q[u][i] = SS[u][i] - SS[3 - u][3 - i];
        }
    if (efold) {
        // R1 (a DCT-II input of length n/8) folds with its mirror (exact); R2 (a DCT-IV input) rotates like class O above
        const f64x4 c3 = *reinterpret_cast<const f64x4*>(rot3 + e0), s3 = *reinterpret_cast<const f64x4*>(rot3 + N16 + e0);
        vec4_t<T> p, m, a, b;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            p[i] = r1q[0][i] + r1q[1][3 - i];
            m[i] = r1q[0][i] - r1q[1][3 - i];
            const T cc = (T)c3[i], ss = (T)s3[i];
            a[i] = r2q[0][i] * cc + r2q[1][3 - i] * ss;
            b[i] = r2q[1][3 - i] * cc - r2q[0][i] * ss;
        }
        put(static_cast<T*>(dp.r1p), e0, p); put(static_cast<T*>(dp.r1m), e0, m);
        put(static_cast<T*>(dp.r2a), e0, a); put(static_cast<T*>(dp.r2b), e0, b);
        return;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        put(R1, pos[u], r1q[u]);
        put(R2, pos[u], r2q[u]);
    }
    if (e0 == 0)
        for (unsigned z = N8; z < K8; z += 4) {
            put(AD, z, zero); put(BS, z, zero); put(R1, z, zero); put(R2, z, zero);
            put(static_cast<T*>(dp.as), z, zero); put(static_cast<T*>(dp.bd), z, zero);
        }
}

// ---------------------------------------------------------------------------------------------
// Three folding levels on a forward COLUMN pass (H % 32 == 0; pays from ~3000 rows: 8K frames): the same
// folds as pair_prep8_rows_kernel, per column, transposed through LDS.  Lines = (frame, column); one block =
// 32 eighth-indices e x 32 columns; per (e, column) the 8 rows that meet in e:
//   rows  e, Hq-1-e, Hq+e, Hh-1-e, Hh+e, 3Hq-1-e, 3Hq+e, H-1-e      (Hq = H/4, Hh = H/2; row u mirrors row 7-u)
//   S[u] = x[u] + x[7-u], P(pos u) = x[u] - x[7-u]   (u < 4: positions e, Hq-1-e, Hq+e, Hh-1-e of the half axis)
//   SS[0] = S[0] + S[3], SS[1] = S[1] + S[2];  M(e) = S[0] - S[3], M(Hq-1-e) = S[1] - S[2]
//   R1(e) = SS[0] + SS[1], R2(e) = SS[0] - SS[1]
// R1, R2: K8 = kpad(H/4) wide; M: Kq = kpad(H/2); P: Kp = kpad(H); k-blocked, zero padded.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pair_prep8_cols_kernel(const float* __restrict__ IN, T* __restrict__ R1,
                                                             T* __restrict__ R2, T* __restrict__ M, T* __restrict__ P,
                                                             unsigned W, unsigned H, unsigned K8, unsigned Kq, unsigned Kp,
                                                             unsigned n_frames, unsigned tiles_e, unsigned tiles_c) {
    __shared__ T s[4][32][33];            // first R1, R2, M(e), M(Hq-1-e); then the four positions of P (34 KB in f64)
    const unsigned Hh = H / 2, Hq = H / 4, He = H / 8;
    const unsigned z = blockIdx.x / (tiles_e * tiles_c);
    const unsigned tt = blockIdx.x % (tiles_e * tiles_c);
    const unsigned e0 = (tt % tiles_e) * 32, c0 = (tt / tiles_e) * 32;
    const float* __restrict__ Pz = IN + (size_t)z * H * W;
    const unsigned tid = threadIdx.x;
    const unsigned er = tid >> 3, cq = (tid & 7) * 4;              // load side: one e, 4 columns
    const unsigned cl = tid & 31, kq = (tid >> 5) * 4;             // store side: one column, 4 consecutive e
    const unsigned cw = c0 + cl, ew = e0 + kq;
    const bool wr = cw < W && ew < K8;
    const size_t line = (size_t)z * W + cw, lines = (size_t)n_frames * W;
    auto fwd = [&](int a) { return (vec4_t<T>){s[a][cl][kq], s[a][cl][kq + 1], s[a][cl][kq + 2], s[a][cl][kq + 3]}; };
    auto rev = [&](int a) { return (vec4_t<T>){s[a][cl][kq + 3], s[a][cl][kq + 2], s[a][cl][kq + 1], s[a][cl][kq]}; };
    T d1[4][4];                                                     // P values of this thread's (e, 4 columns), kept for the second round
    {
        const unsigned e = e0 + er;
        unsigned c = c0 + cq;
        c = c + 4 <= W ? c : W - 4;                               // W % 4 == 0; duplicates are never written out
        f32x4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (e < He) {
            const unsigned rows[8] = {e, Hq - 1 - e, Hq + e, Hh - 1 - e, Hh + e, 3 * Hq - 1 - e, 3 * Hq + e, H - 1 - e};
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const f32x4*>(Pz + (size_t)rows[u] * W + c);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            T S[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                S[u] = (T)x[u][i] + (T)x[7 - u][i];
                d1[u][i] = (T)x[u][i] - (T)x[7 - u][i];
            }
            const T ss0 = S[0] + S[3], ss1 = S[1] + S[2];
            s[0][cq + i][er] = ss0 + ss1;
            s[1][cq + i][er] = ss0 - ss1;
            s[2][cq + i][er] = S[0] - S[3];
            s[3][cq + i][er] = S[1] - S[2];
        }
    }
    __syncthreads();
    if (wr) {
        *reinterpret_cast<vec4_t<T>*>(R1 + blk_index<T>(line, ew, lines)) = fwd(0);      // zero beyond He (x was zero)
        *reinterpret_cast<vec4_t<T>*>(R2 + blk_index<T>(line, ew, lines)) = fwd(1);
        if (ew < He) {                                              // He % 4 == 0: whole quads
            *reinterpret_cast<vec4_t<T>*>(M + blk_index<T>(line, ew, lines)) = fwd(2);
            *reinterpret_cast<vec4_t<T>*>(M + blk_index<T>(line, Hq - 4 - ew, lines)) = rev(3);
        }
        if (ew == 0)
            for (unsigned zz = Hq; zz < Kq; zz += 4) *reinterpret_cast<vec4_t<T>*>(M + blk_index<T>(line, zz, lines)) = (vec4_t<T>){0, 0, 0, 0};
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) s[u][cq + i][er] = d1[u][i];
    __syncthreads();
    if (wr) {
        if (ew < He) {
            *reinterpret_cast<vec4_t<T>*>(P + blk_index<T>(line, ew, lines)) = fwd(0);
            *reinterpret_cast<vec4_t<T>*>(P + blk_index<T>(line, Hq - 4 - ew, lines)) = rev(1);
            *reinterpret_cast<vec4_t<T>*>(P + blk_index<T>(line, Hq + ew, lines)) = fwd(2);
            *reinterpret_cast<vec4_t<T>*>(P + blk_index<T>(line, Hh - 4 - ew, lines)) = rev(3);
        }
        if (ew == 0)
            for (unsigned zz = Hh; zz < Kp; zz += 4) *reinterpret_cast<vec4_t<T>*>(P + blk_index<T>(line, zz, lines)) = (vec4_t<T>){0, 0, 0, 0};
    }
}

// ---------------------------------------------------------------------------------------------
// The deep forward pre-pass of a COLUMN pass (H % 16 == 0): the same planes as pair_prep16_rows_kernel, lines =
// (frame, column), transposed through LDS.  One block = 32 units e < H/16 x 32 columns; per (e, column) the 16 rows
// that meet in e (row u mirrors row 15 - u).  Four LDS rounds of four planes each (34 KB):
//   0: AS BD AD BS at e            1: the same planes at the mirror unit H/8 - 1 - e
//   2: AS2 BD2 AD2 BS2 at e        3: R1 R2 at e and at H/8 - 1 - e
// ---------------------------------------------------------------------------------------------
// v[0 .. nvalid) -> plane positions k0 .. (ascending); one 32-byte store when the run is a whole aligned quad
template <typename T>
__device__ inline void store_run(T* __restrict__ plane, size_t line, size_t lines, unsigned k0, const T (&v)[4], unsigned nvalid) {
    if (nvalid == 4 && (k0 & 3u) == 0) {
        *reinterpret_cast<vec4_t<T>*>(plane + blk_index<T>(line, k0, lines)) = (vec4_t<T>){v[0], v[1], v[2], v[3]};
    } else {
        for (unsigned j = 0; j < nvalid; ++j) plane[blk_index<T>(line, k0 + j, lines)] = v[j];
    }
}

// SPLIT_SD = false ("semi-deep", H % 8 == 0 but not % 16, e.g. 1080 rows): H/16 is not whole, so SD stays one DCT-IV input
// plane (`dp.as2`, kpad(H/2) wide: the r2 launch of the frequencies 2 mod 4) and the units run to ceil(H/16) -- the
// middle unit is its own mirror and stores its values twice.
template <typename T, bool SPLIT_SD>
__global__ __launch_bounds__(256) void pair_prep16_cols_kernel(const float* __restrict__ IN, DeepPlanes dp,
                                                              const double* __restrict__ rot1, const double* __restrict__ rot2,
                                                              unsigned W, unsigned H, unsigned K8, unsigned K16,
                                                              unsigned n_frames, unsigned tiles_e, unsigned tiles_c, unsigned class_major, unsigned ctile, unsigned efold) {
    __shared__ T s[4][32][33];
    const unsigned Hh = H / 2, Hq = H / 4, H8 = H / 8, H16 = SPLIT_SD ? H / 16 : (H / 8 + 1) / 2;      // H16: units
    const unsigned z = blockIdx.x / (tiles_e * tiles_c);
    const unsigned tt = blockIdx.x % (tiles_e * tiles_c);
    const unsigned e0 = (tt % tiles_e) * 32, tile = tt / tiles_e;
    const float* __restrict__ Pz = IN + (size_t)z * H * W;
    const unsigned tid = threadIdx.x;
    const unsigned er = tid >> 3, cq = (tid & 7) * 4;              // load side: one e, 4 columns
    const unsigned cl = tid & 31, kq = (tid >> 5) * 4;             // store side: one column, 4 consecutive e
    // class_major == 2 (W % 256 == 0, long lines): the block takes the ten runs of memory columns -- one per launch class,
    // 32 or 16 wide -- that hold the natural columns [256 tile, 256 tile + 256): its stores then fill whole runs of
    // operand lines (8K: both pre-passes of a forward transform 7.1 -> 6.5 ms per 32 frames; at 4K the plain mapping is faster)
    const unsigned nsub = 1u;
    for (unsigned sub = 0; sub < nsub; ++sub) {
    if (sub) __syncthreads();
    const unsigned wd = 32u;
    const unsigned c0 = tile * 32;
    const bool col_ok = cl < wd && c0 + cl < W;
    const unsigned cw = c0 + cl, ew = e0 + kq;
    // memory column cw of the intermediate plane holds frequency natural(cw) of the row pass (class-major order): the
    // operand line -- and with it the output column of the column GEMMs -- is the natural one
    const unsigned cn = (class_major && col_ok) ? ForwardClassLayout{W, ctile, efold != 0}.natural(cw) : cw;
    const size_t line = (size_t)z * W + cn, lines = (size_t)n_frames * W;
    T* planes8[6] = {static_cast<T*>(dp.as), static_cast<T*>(dp.bd), static_cast<T*>(dp.ad), static_cast<T*>(dp.bs),
                     static_cast<T*>(dp.r1), static_cast<T*>(dp.r2)};
    T* planes16[4] = {static_cast<T*>(dp.as2), static_cast<T*>(dp.bd2), static_cast<T*>(dp.ad2), static_cast<T*>(dp.bs2)};
    const unsigned e = e0 + er;
    const bool unit_ok = e < H16;
    const unsigned ec = unit_ok ? e : 0;                           // table indices stay in range
    f32x4 x[16];
    {
        unsigned c = c0 + cq;
        c = c + 4 <= W ? c : W - 4;                               // W % 4 == 0; duplicates are never written out
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (unit_ok) {
            const unsigned rp[8] = {e, H8 - 1 - e, H8 + e, Hq - 1 - e, Hq + e, 3 * H8 - 1 - e, 3 * H8 + e, Hh - 1 - e};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x[u] = *reinterpret_cast<const f32x4*>(Pz + (size_t)rp[u] * W + c);
                x[15 - u] = *reinterpret_cast<const f32x4*>(Pz + (size_t)(H - 1 - rp[u]) * W + c);
            }
        }
    }
    const unsigned nvalid = !col_ok ? 0u : (ew >= H16 ? 0u : (H16 - ew < 4 ? H16 - ew : 4u));      // valid units of the store quad
    auto gather = [&](int a, T (&v)[4], bool reverse) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = s[a][cl][kq + j];
        if (reverse) {                                              // first nvalid entries, reversed
            T w[4] = {0, 0, 0, 0};
            for (unsigned j = 0; j < nvalid; ++j) w[j] = v[nvalid - 1 - j];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = w[j];
        }
    };
#pragma unroll 1
    for (int round = 0; round < 4; ++round) {
        if (round) __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            T D[8], S[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                S[u] = (T)x[u][i] + (T)x[15 - u][i];
                D[u] = (T)x[u][i] - (T)x[15 - u][i];
            }
            T o[4] = {0, 0, 0, 0};
            if (round == 0) {
                split_one<T>(D[0], D[3], D[4], D[7], rot1, ec, Hq, o[0], o[1], o[2], o[3]);
            } else if (round == 1) {
                split_one<T>(D[1], D[2], D[5], D[6], rot1, H8 - 1 - ec, Hq, o[0], o[1], o[2], o[3]);
            } else {
                T SS[4], SD[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { SS[u] = S[u] + S[7 - u]; SD[u] = S[u] - S[7 - u]; }
                if (round == 2) {
                    if (SPLIT_SD) split_one<T>(SD[0], SD[1], SD[2], SD[3], rot2, ec, H8, o[0], o[1], o[2], o[3]);
                    else { o[0] = SD[0]; o[1] = SD[1]; o[2] = SD[2]; o[3] = SD[3]; }      // SD at e, H/8-1-e, H/8+e, H/4-1-e
                } else {
                    o[0] = SS[0] + SS[3]; o[1] = SS[0] - SS[3];      // R1, R2 at e
                    o[2] = SS[1] + SS[2]; o[3] = SS[1] - SS[2];      // R1, R2 at H/8 - 1 - e
                }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) s[a][cq + i][er] = unit_ok ? o[a] : (T)0;
        }
        __syncthreads();
        if (col_ok && ew < (SPLIT_SD ? K16 : ((H16 + 3) & ~3u))) {
            T v[4];
            if (round == 0) {
#pragma unroll
                for (int a = 0; a < 4; ++a) { gather(a, v, false); store_run<T>(planes8[a], line, lines, ew, v, nvalid); }
            } else if (round == 1) {
#pragma unroll
                for (int a = 0; a < 4; ++a) { gather(a, v, true); if (nvalid) store_run<T>(planes8[a], line, lines, H8 - ew - nvalid, v, nvalid); }
            } else if (round == 2) {
                if (SPLIT_SD) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) { gather(a, v, false); store_run<T>(planes16[a], line, lines, ew, v, 4u); }   // zeros beyond H/16
                } else {
                    T* M = planes16[0];
                    gather(0, v, false); store_run<T>(M, line, lines, ew, v, nvalid);
                    gather(2, v, false); store_run<T>(M, line, lines, H8 + ew, v, nvalid);
                    gather(1, v, true); if (nvalid) store_run<T>(M, line, lines, H8 - ew - nvalid, v, nvalid);
                    gather(3, v, true); if (nvalid) store_run<T>(M, line, lines, Hq - ew - nvalid, v, nvalid);
                    if (ew == 0) for (unsigned k = Hq; k < K16; ++k) M[blk_index<T>(line, k, lines)] = (T)0;      // K16: kpad(H/2) here
                }
            } else {
                gather(0, v, false); store_run<T>(planes8[4], line, lines, ew, v, nvalid);
                gather(1, v, false); store_run<T>(planes8[5], line, lines, ew, v, nvalid);
                gather(2, v, true); if (nvalid) store_run<T>(planes8[4], line, lines, H8 - ew - nvalid, v, nvalid);
                gather(3, v, true); if (nvalid) store_run<T>(planes8[5], line, lines, H8 - ew - nvalid, v, nvalid);
                if (ew == 0)
                    for (unsigned k = H8; k < K8; ++k)
#pragma unroll
                        for (int a = 0; a < 6; ++a) planes8[a][blk_index<T>(line, k, lines)] = (T)0;
            }
        }
    }
    }
}

// ---------------------------------------------------------------------------------------------
// Deep INVERSE pre-passes.  The inverse transform of coefficients c[0 .. n) combines, per axis,
//   x[m]  = T[m] +/- o[m]      T = half-length inverse of the even coefficients, o = DCT-IV of the odd ones (n/2 each)
//   T[m]  = T2[m] +/- o2[m]    T2 = quarter-length inverse of c[4q], o2 = DCT-IV of c[4q+2]                (n/4 each)
//   T2[m] = EEE-part +/- EEO-part: c[8q] and c[8q+4] against the half bases of length n/4                   (n/8 each)
// and both DCT-IVs are split like the forward ones: the operands are
//   AS BD AD BS  (n/8 each)   from k -> c[2k+1]  (rotation partners k, n/2-1-k; fold partners n/4-1-k, n/4+k)
//   AS2 .. BS2   (n/16 each)  from q -> c[4q+2]
//   R1 = c[8q], R2 = c[8q+4]  (n/8 each)
// in the DeepPlanes order of the forward pre-passes.
// Row pass (n % 128 == 0): one thread = the eight 16-element regions of a line that close under those pairings,
//   g = R, n/4-16-R, n/4+R, n/2-16-R, n/2+R, 3n/4-16-R, 3n/4+R, n-16-R   (R = 16 t): 128 coefficients in, 128 doubles out.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pair_prep16_inv_rows_kernel(const float* __restrict__ X, DeepPlanes dp,
                                                                  const double* __restrict__ rot1, const double* __restrict__ rot2,
                                                                  unsigned rows, unsigned W, unsigned K8, unsigned K16, unsigned tp_line /*threads per line, power of 2*/) {
    const unsigned t = threadIdx.x & (tp_line - 1);
    const unsigned row = blockIdx.x * (256 / tp_line) + threadIdx.x / tp_line;
    const unsigned R = 16 * t;
    const unsigned Nh = W / 2, Nq = W / 4, N8 = W / 8, N16 = W / 16;
    if (row >= rows || R >= N8) return;
    T* P8[6] = {static_cast<T*>(dp.as), static_cast<T*>(dp.bd), static_cast<T*>(dp.ad), static_cast<T*>(dp.bs), static_cast<T*>(dp.r1), static_cast<T*>(dp.r2)};
    T* P16[4] = {static_cast<T*>(dp.as2), static_cast<T*>(dp.bd2), static_cast<T*>(dp.ad2), static_cast<T*>(dp.bs2)};
    auto put4 = [&](T* plane, unsigned k, const vec4_t<T>& v) { *reinterpret_cast<vec4_t<T>*>(plane + blk_index<T>(row, k, rows)) = v; };
    auto put2 = [&](T* plane, unsigned k, T a, T b) { *reinterpret_cast<vec2_t<T>*>(plane + blk_index<T>(row, k, rows)) = (vec2_t<T>){a, b}; };
    const unsigned g[8] = {R, Nq - 16 - R, Nq + R, Nh - 16 - R, Nh + R, 3 * Nq - 16 - R, 3 * Nq + R, W - 16 - R};
    const float* xr = X + (size_t)row * W;
    f32x4 c[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) c[j][q] = *reinterpret_cast<const f32x4*>(xr + g[j] + 4 * q);
    // odd coefficients of region j as two ascending quads of k: element u = 2 t + 1 -> quad t / 4
    auto odd = [&](int j, int half) { return (vec4_t<T>){(T)c[j][2 * half][1], (T)c[j][2 * half][3], (T)c[j][2 * half + 1][1], (T)c[j][2 * half + 1][3]}; };
    // unit A: k = e0 .. e0+7 (e0 = R/2) from regions 0, 3, 4, 7; its mirror unit n/8-8-e0 .. from regions 1, 2, 5, 6
    const unsigned e0 = R / 2, m0u = N8 - 8 - e0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        vec4_t<T> as, bd, ad, bs;
        split_unit<T>(odd(0, half), odd(3, 1 - half), odd(4, half), odd(7, 1 - half), rot1, e0 + 4 * half, Nq, as, bd, ad, bs);
        put4(P8[0], e0 + 4 * half, as); put4(P8[1], e0 + 4 * half, bd); put4(P8[2], e0 + 4 * half, ad); put4(P8[3], e0 + 4 * half, bs);
        split_unit<T>(odd(1, half), odd(2, 1 - half), odd(5, half), odd(6, 1 - half), rot1, m0u + 4 * half, Nq, as, bd, ad, bs);
        put4(P8[0], m0u + 4 * half, as); put4(P8[1], m0u + 4 * half, bd); put4(P8[2], m0u + 4 * half, ad); put4(P8[3], m0u + 4 * half, bs);
    }
    // c[4q+2]: element 2 of every quad of a region, q ascending
    auto mid = [&](int j) { return (vec4_t<T>){(T)c[j][0][2], (T)c[j][1][2], (T)c[j][2][2], (T)c[j][3][2]}; };
    {
        const unsigned f0 = R / 4, f1 = N16 - 4 - f0;
        vec4_t<T> as, bd, ad, bs;
        split_unit<T>(mid(0), mid(3), mid(4), mid(7), rot2, f0, N8, as, bd, ad, bs);
        put4(P16[0], f0, as); put4(P16[1], f0, bd); put4(P16[2], f0, ad); put4(P16[3], f0, bs);
        split_unit<T>(mid(1), mid(2), mid(5), mid(6), rot2, f1, N8, as, bd, ad, bs);
        put4(P16[0], f1, as); put4(P16[1], f1, bd); put4(P16[2], f1, ad); put4(P16[3], f1, bs);
    }
    // c[8q] and c[8q+4]: elements 0 of quads 0, 2 and of quads 1, 3
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        put2(P8[4], g[j] / 8, (T)c[j][0][0], (T)c[j][2][0]);
        put2(P8[5], g[j] / 8, (T)c[j][1][0], (T)c[j][3][0]);
    }
    if (t == 0) {
        const vec4_t<T> zero = {0, 0, 0, 0};
        for (unsigned z = N8; z < K8; z += 4)
#pragma unroll
            for (int a = 0; a < 6; ++a) put4(P8[a], z, zero);
        for (unsigned z = N16; z < K16; z += 4)
#pragma unroll
            for (int a = 0; a < 4; ++a) put4(P16[a], z, zero);
    }
}

// Column pass (H % 16 == 0): lines = (frame, column).  One block = 32 columns x one group G of 8 units: thread (column,
// t) serves the D-split pair (e = 8G + t, H/8 - 1 - e), the level-2 unit e' = 8G + t and the level-3 pairs q = 16G + 2t,
// +1 -- 16 coefficients in (coalesced across the columns), 16 doubles out, transposed through LDS so that every store
// is a whole 64-byte piece of one line where the positions allow it.
// SPLIT_MID = false ("semi-deep", H % 8 == 0 but not % 16): c[4q+2] stays one DCT-IV input plane (`dp.as2`, kpad(H/2)
// wide) and the units run to ceil(H/16); the middle unit is its own mirror and stores its values twice.
template <typename T, bool SPLIT_MID>
__global__ __launch_bounds__(256) void pair_prep16_inv_cols_kernel(const float* __restrict__ IN, DeepPlanes dp,
                                                                  const double* __restrict__ rot1, const double* __restrict__ rot2,
                                                                  unsigned W, unsigned H, unsigned K8, unsigned K16,
                                                                  unsigned n_frames, unsigned groups, unsigned tiles_c, unsigned class_major, unsigned ctile) {
    __shared__ T s[16][32][9];
    const unsigned Hh = H / 2, Hq = H / 4, H8 = H / 8, H16 = SPLIT_MID ? H / 16 : (H / 8 + 1) / 2;      // H16: units
    const unsigned z = blockIdx.x / (groups * tiles_c);
    const unsigned tt = blockIdx.x % (groups * tiles_c);
    const unsigned G = tt % groups, c0 = (tt / groups) * 32;
    const float* __restrict__ Pz = IN + (size_t)z * H * W;
    const unsigned cl = threadIdx.x & 31, t = threadIdx.x >> 5;
    // class-major input: lane cl takes NATURAL column c0 + cl, i.e. memory column inverse_class_pos(c0 + cl) -- the loads
    // of a row are four 32-byte runs (one per residue class) and the stores below stay runs of consecutive operand lines
    const unsigned coln = c0 + cl < W ? c0 + cl : W - 1;
    const unsigned col = class_major ? inverse_class_pos(coln, W, ctile, class_major == 2) : coln;      // 2: the level-2 order
    auto ld = [&](unsigned r) { return (T)Pz[(size_t)r * W + col]; };
    T* P8[6] = {static_cast<T*>(dp.as), static_cast<T*>(dp.bd), static_cast<T*>(dp.ad), static_cast<T*>(dp.bs), static_cast<T*>(dp.r1), static_cast<T*>(dp.r2)};
    T* P16[4] = {static_cast<T*>(dp.as2), static_cast<T*>(dp.bd2), static_cast<T*>(dp.ad2), static_cast<T*>(dp.bs2)};
    const unsigned e = 8 * G + t;                                  // D-split pair index (< H/16) and level-2 unit (< H/16)
    const bool ok = e < H16;
    const unsigned ec = ok ? e : 0, em = H8 - 1 - ec;
    T o[16];
    {   // odd coefficients: k -> row 2k+1; unit at e: k = e, H/4-1-e, H/4+e, H/2-1-e; mirror unit at em
        const T d0 = ld(2 * ec + 1), d1 = ld(Hh - 1 - 2 * ec), d2 = ld(Hh + 2 * ec + 1), d3 = ld(H - 1 - 2 * ec);
        split_one<T>(d0, d1, d2, d3, rot1, ec, Hq, o[0], o[1], o[2], o[3]);
        const T m0v = ld(2 * em + 1), m1 = ld(Hh - 1 - 2 * em), m2 = ld(Hh + 2 * em + 1), m3 = ld(H - 1 - 2 * em);
        split_one<T>(m0v, m1, m2, m3, rot1, em, Hq, o[4], o[5], o[6], o[7]);
        // c[4q+2]: q = e, H/8-1-e, H/8+e, H/4-1-e
        const T q0 = ld(4 * ec + 2), q1 = ld(Hh - 2 - 4 * ec), q2 = ld(Hh + 4 * ec + 2), q3 = ld(H - 2 - 4 * ec);
        if (SPLIT_MID) split_one<T>(q0, q1, q2, q3, rot2, ec, H8, o[8], o[9], o[10], o[11]);
        else { o[8] = q0; o[9] = q1; o[10] = q2; o[11] = q3; }          // c[4q+2] itself at q = e, H/8-1-e, H/8+e, H/4-1-e
        // c[8q], c[8q+4] for q = 2e, 2e+1 (< H/8; with an odd H/8 the last unit has one)
        const bool q2ok = 2 * ec + 1 < H8;
        o[12] = ld(16 * ec); o[13] = q2ok ? ld(16 * ec + 8) : (T)0; o[14] = ld(16 * ec + 4); o[15] = q2ok ? ld(16 * ec + 12) : (T)0;
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) s[v][cl][t] = ok ? o[v] : (T)0;
    __syncthreads();
    if (c0 + cl >= W) return;
    const size_t line = (size_t)z * W + c0 + cl, lines = (size_t)n_frames * W;
    const unsigned nv = 8 * G >= H16 ? 0u : (H16 - 8 * G < 8 ? H16 - 8 * G : 8u);          // valid units of the group
    const unsigned h = t;                                          // store side: thread h of a column takes value type h
    {   // AS BD AD BS at e = 8G .. (h < 4) or at the mirror units H/8 - 1 - e, ascending from H/8 - 8G - nv (h >= 4)
        T* plane = P8[h & 3];
        if (h < 4) {
            if (nv == 8) {
                T* o8 = plane + blk_index<T>(line, 8 * G, lines);
                *reinterpret_cast<vec4_t<T>*>(o8) = (vec4_t<T>){s[h][cl][0], s[h][cl][1], s[h][cl][2], s[h][cl][3]};
                *reinterpret_cast<vec4_t<T>*>(o8 + 4) = (vec4_t<T>){s[h][cl][4], s[h][cl][5], s[h][cl][6], s[h][cl][7]};
            } else {
                for (unsigned j = 0; j < nv; ++j) plane[blk_index<T>(line, 8 * G + j, lines)] = s[h][cl][j];
            }
        } else {
            const unsigned k0 = H8 - 8 * G - nv;
            if (nv == 8 && (k0 & 7u) == 0) {
                T* o8 = plane + blk_index<T>(line, k0, lines);
                *reinterpret_cast<vec4_t<T>*>(o8) = (vec4_t<T>){s[h][cl][7], s[h][cl][6], s[h][cl][5], s[h][cl][4]};
                *reinterpret_cast<vec4_t<T>*>(o8 + 4) = (vec4_t<T>){s[h][cl][3], s[h][cl][2], s[h][cl][1], s[h][cl][0]};
            } else {
                for (unsigned j = 0; j < nv; ++j) plane[blk_index<T>(line, k0 + j, lines)] = s[h][cl][nv - 1 - j];
            }
        }
    }
    if (SPLIT_MID) {   // AS2 .. BS2 at e' = 8G + 4 (h / 4) .. +3: zeros beyond H/16 (the planes are K16 wide)
        T* plane = P16[h & 3];
        const unsigned k0 = 8 * G + 4 * (h >> 2), j0 = 4 * (h >> 2);
        if (k0 < K16) *reinterpret_cast<vec4_t<T>*>(plane + blk_index<T>(line, k0, lines)) = (vec4_t<T>){s[8 + (h & 3)][cl][j0], s[8 + (h & 3)][cl][j0 + 1], s[8 + (h & 3)][cl][j0 + 2], s[8 + (h & 3)][cl][j0 + 3]};
    } else if (h < 4) {   // the c[4q+2] plane: thread h takes the run of value type 8 + h (ascending for h = 0, 2; mirrored for 1, 3)
        T* plane = P16[0];
        const unsigned start = h == 0 ? 8 * G : h == 2 ? H8 + 8 * G : (h == 1 ? H8 : Hq) - 8 * G - nv;
        for (unsigned j = 0; j < nv; ++j) plane[blk_index<T>(line, start + j, lines)] = s[8 + h][cl][(h & 1) ? nv - 1 - j : j];
        if (G == 0 && h == 0) for (unsigned k = Hq; k < K16; ++k) plane[blk_index<T>(line, k, lines)] = (T)0;      // K16: kpad(H/2) here
    }
    {   // R1 = c[8q], R2 = c[8q+4] at q = 16G + 2h, +1 (units beyond H/16 wrote zeros: padding up to K8 where 16G < K8)
        const unsigned q0 = 16 * G + 2 * h;
        if (q0 < K8) {
            *reinterpret_cast<vec2_t<T>*>(P8[4] + blk_index<T>(line, q0, lines)) = (vec2_t<T>){s[12][cl][h], s[13][cl][h]};
            *reinterpret_cast<vec2_t<T>*>(P8[5] + blk_index<T>(line, q0, lines)) = (vec2_t<T>){s[14][cl][h], s[15][cl][h]};
        }
    }
    if (G == 0 && h == 0)                                           // AS .. BS beyond H/8
        for (unsigned k = H8; k < K8; ++k)
#pragma unroll
            for (int a = 0; a < 4; ++a) P8[a][blk_index<T>(line, k, lines)] = (T)0;
}

// ---------------------------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------------------------
size_t dct_pair_operand_elems(bool f64, size_t n_frames, size_t w, size_t h) {
    const size_t a = n_frames * h * dct_pair_kpad(f64, w), b = n_frames * w * dct_pair_kpad(f64, h);
    return a > b ? a : b;
}

bool dct_pair_can_run(bool f64, size_t n_frames, size_t w, size_t h, const float* in, const float* out) {
    // an operand plane must stay below 4 GB (32-bit scalar offsets walk its k-blocks)
    if (!f64 && !build_all_strategies()) return false;          // the f32 twin (dct_pair_f32.hip) is part of the diagnostic build only
    return dct_rows_can_fold(w, in, out) && dct_cols_can_fold(w, h, in, out) && w % 8 == 0 && h % 8 == 0 &&
           dct_pair_operand_elems(f64, n_frames, w, h) * (f64 ? 8 : 4) <= 0xFFFFFFFFull;
}
// second level along an axis of length len: quarter length a multiple of 4 (row passes read quads of a
// line; the transposing column pre-pass only needs an even quarter), at least one k-step pair
bool dct_pair_can_fold2(size_t len) { return len % 16 == 0 && len >= 64; }
bool dct_pair_can_fold2_cols(size_t len) { return len % 8 == 0 && len >= 64; }

template <typename T>
static int prep_impl(hipStream_t st, bool is_row, bool inverse, const float* in, size_t n_frames, size_t w, size_t h,
                     T* o1, T* o2) {
    if (n_frames == 0) return SSW_OK;
    if (w > 0xFFFFFFull || h > 0xFFFFFFull || n_frames > 0xFFFFFFull) return SSW_ERR_BAD_DIMS;
    if (is_row) {
        const unsigned Kp = (unsigned)pair_kpad<T>(w), tiles_k = (Kp + 31) / 32;
        const size_t rows = n_frames * h;
        const unsigned long long nblk = (unsigned long long)((rows + 31) / 32) * tiles_k;
        if (rows > 0xFFFFFFFFull || nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
        if (inverse) pair_prep_rows_kernel<T, true><<<(unsigned)nblk, 256, 0, st>>>(in, o1, o2, (unsigned)rows, (unsigned)w, Kp, tiles_k);
        else         pair_prep_rows_kernel<T, false><<<(unsigned)nblk, 256, 0, st>>>(in, o1, o2, (unsigned)rows, (unsigned)w, Kp, tiles_k);
    } else {
        const unsigned Kp = (unsigned)pair_kpad<T>(h);
        const unsigned tiles_k = (Kp + 31) / 32, tiles_c = (unsigned)((w + 63) / 64);
        const unsigned long long nblk = (unsigned long long)tiles_k * tiles_c * n_frames;
        if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
        if (inverse) pair_prep_cols_kernel<T, true><<<(unsigned)nblk, 256, 0, st>>>(in, o1, o2, (unsigned)w, (unsigned)h, Kp, (unsigned)n_frames, tiles_k, tiles_c);
        else         pair_prep_cols_kernel<T, false><<<(unsigned)nblk, 256, 0, st>>>(in, o1, o2, (unsigned)w, (unsigned)h, Kp, (unsigned)n_frames, tiles_k, tiles_c);
    }
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

template <typename T>
static int prep4_impl(hipStream_t st, bool is_row, bool inverse, const float* in, size_t n_frames, size_t w, size_t h,
                      T* q1, T* q2, T* p) {
    if (n_frames == 0) return SSW_OK;
    if (w > 0xFFFFFFull || h > 0xFFFFFFull || n_frames > 0xFFFFFFull) return SSW_ERR_BAD_DIMS;
    const size_t len = is_row ? w : h;
    const unsigned Kp = (unsigned)pair_kpad<T>(len), Kq = (unsigned)pair_kpad<T>(len / 2);
    const unsigned tiles_q = (Kq + 31) / 32;
    if (is_row) {
        const size_t rows = n_frames * h;
        const unsigned long long nblk = (unsigned long long)((rows + 31) / 32) * tiles_q;
        if (rows > 0xFFFFFFFFull || nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
        if (inverse) pair_prep4_rows_kernel<T, true><<<(unsigned)nblk, 256, 0, st>>>(in, q1, q2, p, (unsigned)rows, (unsigned)w, Kq, Kp, tiles_q);
        else         pair_prep4_rows_kernel<T, false><<<(unsigned)nblk, 256, 0, st>>>(in, q1, q2, p, (unsigned)rows, (unsigned)w, Kq, Kp, tiles_q);
    } else {
        const unsigned tiles_c = (unsigned)((w + 31) / 32);
        const unsigned long long nblk = (unsigned long long)tiles_q * tiles_c * n_frames;
        if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
        if (inverse) pair_prep4_cols_kernel<T, true><<<(unsigned)nblk, 256, 0, st>>>(in, q1, q2, p, (unsigned)w, (unsigned)h, Kq, Kp, (unsigned)n_frames, tiles_q, tiles_c);
        else         pair_prep4_cols_kernel<T, false><<<(unsigned)nblk, 256, 0, st>>>(in, q1, q2, p, (unsigned)w, (unsigned)h, Kq, Kp, (unsigned)n_frames, tiles_q, tiles_c);
    }
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

int launch_dct_pair_prep(hipStream_t st, bool f64, bool is_row, bool inverse, const float* in, size_t n_frames, size_t w,
                         size_t h, void* o1, void* o2) {
    return f64 ? prep_impl<double>(st, is_row, inverse, in, n_frames, w, h, (double*)o1, (double*)o2)
               : prep_impl<float>(st, is_row, inverse, in, n_frames, w, h, (float*)o1, (float*)o2);
}

int launch_dct_pair_prep4(hipStream_t st, bool f64, bool is_row, bool inverse, const float* in, size_t n_frames, size_t w,
                          size_t h, void* q1, void* q2, void* p) {
    return f64 ? prep4_impl<double>(st, is_row, inverse, in, n_frames, w, h, (double*)q1, (double*)q2, (double*)p)
               : prep4_impl<float>(st, is_row, inverse, in, n_frames, w, h, (float*)q1, (float*)q2, (float*)p);
}

template <typename T>
static int prep4_rgb_impl(hipStream_t st, int u8, const void* rgb, size_t n_frames, size_t w, size_t h,
                          T* q1, T* q2, T* p, float* ip, float* qp) {
    if (n_frames == 0) return SSW_OK;
    if (w > 0xFFFFFFull || h > 0xFFFFFFull) return SSW_ERR_BAD_DIMS;
    const unsigned Kp = (unsigned)pair_kpad<T>(w), Kq = (unsigned)pair_kpad<T>(w / 2), tiles_q = (Kq + 31) / 32;
    const size_t rows = n_frames * h;
    const unsigned long long nblk = (unsigned long long)((rows + 31) / 32) * tiles_q;
    if (rows > 0xFFFFFFFFull || nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    const bool iq = ip && qp;
#define SSW_PREP_RGB(U8V, IQV) pair_prep4_rows_rgb_kernel<T, U8V, IQV><<<(unsigned)nblk, 256, 0, st>>>( \
        rgb, q1, q2, p, ip, qp, (unsigned)rows, (unsigned)w, Kq, Kp, tiles_q)
    if (u8 == SSW_PIX_U8)       { if (iq) SSW_PREP_RGB(SSW_PIX_U8, true); else SSW_PREP_RGB(SSW_PIX_U8, false); }
    else if (u8 == SSW_PIX_U16) { if (iq) SSW_PREP_RGB(SSW_PIX_U16, true); else SSW_PREP_RGB(SSW_PIX_U16, false); }
    else                        { if (iq) SSW_PREP_RGB(SSW_PIX_F32, true); else SSW_PREP_RGB(SSW_PIX_F32, false); }
#undef SSW_PREP_RGB
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// rows-first forward transform with two folding levels on the row axis: the first pre-pass straight
// from the interleaved RGB frames (u8 or f32); ip / qp (both or neither) receive the I and Q planes.
bool dct_pair_can_prep_from_rgb(size_t w, size_t h, const void* rgb, int u8) {
    return w >= h && dct_pair_can_fold2(w) && (reinterpret_cast<uintptr_t>(rgb) & pix_align_mask(u8)) == 0;
}
int launch_dct_pair_prep4_rows_rgb(hipStream_t st, bool f64, int u8, const void* rgb, size_t n_frames, size_t w, size_t h,
                                   void* q1, void* q2, void* p, float* ip, float* qp) {
    return f64 ? prep4_rgb_impl<double>(st, u8, rgb, n_frames, w, h, (double*)q1, (double*)q2, (double*)p, ip, qp)
               : prep4_rgb_impl<float>(st, u8, rgb, n_frames, w, h, (float*)q1, (float*)q2, (float*)p, ip, qp);
}

template <typename T>
static int prep8_impl(hipStream_t st, int src_kind, const void* src, size_t n_frames, size_t w, size_t h,
                      T* r1, T* r2, T* m, T* p, float* ip, float* qp) {
    if (n_frames == 0) return SSW_OK;
    if (w > 0xFFFFFFull || h > 0xFFFFFFull) return SSW_ERR_BAD_DIMS;
    const unsigned Kp = (unsigned)pair_kpad<T>(w), Kq = (unsigned)pair_kpad<T>(w / 2), K8 = (unsigned)pair_kpad<T>(w / 4);
    const unsigned tiles_e = (K8 + 31) / 32;
    const size_t rows = n_frames * h;
    const unsigned long long nblk = (unsigned long long)((rows + 31) / 32) * tiles_e;
    if (rows > 0xFFFFFFFFull || nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    const bool iq = ip && qp;
#define SSW_PREP8(SRCV, IQV) pair_prep8_rows_kernel<T, SRCV, IQV><<<(unsigned)nblk, 256, 0, st>>>( \
        src, r1, r2, m, p, ip, qp, (unsigned)rows, (unsigned)w, K8, Kq, Kp, tiles_e)
    if (src_kind == 0) SSW_PREP8(0, false);
    else if (src_kind == 1) { if (iq) SSW_PREP8(1, true); else SSW_PREP8(1, false); }
    else if (src_kind == 2) { if (iq) SSW_PREP8(2, true); else SSW_PREP8(2, false); }
    else                    { if (iq) SSW_PREP8(3, true); else SSW_PREP8(3, false); }
#undef SSW_PREP8
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// third folding level along an axis of length len (forward row passes only)
bool dct_pair_can_fold3(size_t len) { return len % 32 == 0 && len >= 128; }
// src_kind: 0 = f32 plane, 1 = interleaved RGB f32, 2 = interleaved RGB u8 (ip / qp: I, Q planes out or null)
int launch_dct_pair_prep8_rows(hipStream_t st, bool f64, int src_kind, const void* src, size_t n_frames, size_t w, size_t h,
                               void* r1, void* r2, void* m, void* p, float* ip, float* qp) {
    return f64 ? prep8_impl<double>(st, src_kind, src, n_frames, w, h, (double*)r1, (double*)r2, (double*)m, (double*)p, ip, qp)
               : prep8_impl<float>(st, src_kind, src, n_frames, w, h, (float*)r1, (float*)r2, (float*)m, (float*)p, ip, qp);
}

template <typename T>
static int prep8_cols_impl(hipStream_t st, const float* in, size_t n_frames, size_t w, size_t h, T* r1, T* r2, T* m, T* p) {
    if (n_frames == 0) return SSW_OK;
    if (w > 0xFFFFFFull || h > 0xFFFFFFull || n_frames > 0xFFFFFFull) return SSW_ERR_BAD_DIMS;
    const unsigned Kp = (unsigned)pair_kpad<T>(h), Kq = (unsigned)pair_kpad<T>(h / 2), K8 = (unsigned)pair_kpad<T>(h / 4);
    const unsigned tiles_e = (K8 + 31) / 32, tiles_c = (unsigned)((w + 31) / 32);
    const unsigned long long nblk = (unsigned long long)tiles_e * tiles_c * n_frames;
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    pair_prep8_cols_kernel<T><<<(unsigned)nblk, 256, 0, st>>>(in, r1, r2, m, p, (unsigned)w, (unsigned)h, K8, Kq, Kp, (unsigned)n_frames, tiles_e, tiles_c);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// three levels on a forward column pass (H % 32 == 0): (SSS, SS-) [kpad(h/4) wide], S- [kpad(h/2)], x- [kpad(h)]
int launch_dct_pair_prep8_cols(hipStream_t st, bool f64, const float* in, size_t n_frames, size_t w, size_t h,
                               void* r1, void* r2, void* m, void* p) {
    return f64 ? prep8_cols_impl<double>(st, in, n_frames, w, h, (double*)r1, (double*)r2, (double*)m, (double*)p)
               : prep8_cols_impl<float>(st, in, n_frames, w, h, (float*)r1, (float*)r2, (float*)m, (float*)p);
}

// deep forward row pre-pass: src_kind 0 = f32 plane, 1 / 2 = interleaved RGB f32 / u8 (ip / qp: I, Q planes out or null);
// base: 6 planes of lines * K8 doubles (AS BD AD BS R1 R2) followed by 4 planes of lines * K16 (AS2 BD2 AD2 BS2)
bool dct_pair_can_deep_rows(size_t len) { const size_t mn = (size_t)tuning(TUNE_DEEP_MIN_ROWS); return len % 64 == 0 && len >= mn; }
// 6 planes K8 wide + 4 K16 wide, or (forward row passes at level 2) 16 planes K16 wide
size_t dct_pair_deep_elems(size_t lines, size_t len) {
    const size_t k8 = dct_pair_split_kpad(len), k16 = dct_pair_split_kpad(len / 2);
    return lines * (6 * k8 + 4 * k16 > 16 * k16 ? 6 * k8 + 4 * k16 : 16 * k16);
}
int launch_dct_pair_prep16_rows(hipStream_t st, int src_kind, const void* src, size_t n_frames, size_t w, size_t h, double* base,
                                const double* rot1, const double* rot2, const double* rot3, float* ip, float* qp, bool unit_order) {
    if (n_frames == 0) return SSW_OK;
    if (w > 0xFFFFFFull || h > 0xFFFFFFull || !dct_pair_can_deep_rows(w)) return SSW_ERR_BAD_DIMS;
    const unsigned K8 = (unsigned)dct_pair_split_kpad(w), K16 = (unsigned)dct_pair_split_kpad(w / 2);
    const unsigned tiles_e = (K16 + 31) / 32;
    if (unit_order && (h % 16 != 0 || !dct_pair_efold(w))) return SSW_ERR_BAD_ARG;
    const unsigned unit_hup = unit_order ? (unsigned)dct_pair_fused_units(h) : 0u;
    const size_t rows = unit_order ? n_frames * 16 * unit_hup : n_frames * h;       // operand lines
    const unsigned long long nblk = (unsigned long long)((rows + 31) / 32) * tiles_e;
    if (rows > 0xFFFFFFFFull || nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    DeepPlanes dp;
    double* p = base;
    const size_t p8 = rows * K8, p16 = rows * K16;
    const unsigned efold = dct_pair_efold(w) ? 1u : 0u;
    if (efold) {     // level 2: sixteen planes K16 wide, in this order (build_pass and the pruned pass index them by number)
        void** const l2[16] = {&dp.asp, &dp.asm_, &dp.bdp, &dp.bdm, &dp.oap, &dp.obp, &dp.oam, &dp.obm,
                               &dp.r1p, &dp.r1m, &dp.r2a, &dp.r2b, &dp.as2, &dp.bd2, &dp.ad2, &dp.bs2};
        for (int j = 0; j < 16; ++j) *l2[j] = p + (size_t)j * p16;
        dp.as = dp.bd = dp.ad = dp.bs = dp.r1 = dp.r2 = nullptr;
        if (!rot3) return SSW_ERR_BAD_ARG;
    } else {
        dp.as = p; dp.bd = p + p8; dp.ad = p + 2 * p8; dp.bs = p + 3 * p8; dp.r1 = p + 4 * p8; dp.r2 = p + 5 * p8;
        p += 6 * p8;
        dp.as2 = p; dp.bd2 = p + p16; dp.ad2 = p + 2 * p16; dp.bs2 = p + 3 * p16;
    }
    if (efold && dct_pair_prep_light_ok(w, rows))          // r5: the form that runs beside the GEMMs of the other lane (dct_pair_prep_light.hip)
        return launch_dct_pair_prep16_rows_light(st, src_kind, src, dp, rot1, rot2, rot3, ip, qp, rows, w, K16, unit_order ? (unsigned)h : 0u, unit_hup);
    const bool iq = ip && qp;
#define SSW_PREP16(SRCV, IQV) pair_prep16_rows_kernel<double, SRCV, IQV><<<(unsigned)nblk, 256, 0, st>>>( \
        src, dp, rot1, rot2, rot3, ip, qp, (unsigned)rows, (unsigned)w, K8, K16, tiles_e, efold, unit_order ? (unsigned)h : 0u, unit_hup)
    if (src_kind == 0) SSW_PREP16(0, false);
    else if (src_kind == 1) { if (iq) SSW_PREP16(1, true); else SSW_PREP16(1, false); }
    else if (src_kind == 2) { if (iq) SSW_PREP16(2, true); else SSW_PREP16(2, false); }
    else                    { if (iq) SSW_PREP16(3, true); else SSW_PREP16(3, false); }
#undef SSW_PREP16
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// deep forward column pre-pass (H % 16 == 0): same plane order as the row version, lines = n_frames * w
bool dct_pair_can_deep_cols(size_t len) { const size_t mn = (size_t)tuning(TUNE_DEEP_MIN_COLS); return len % 16 == 0 && len >= mn; }
// semi-deep: H % 8 == 0 but not % 16 (1080 rows): D split, SS folded a third time, SD left whole
bool dct_pair_can_semi_deep_cols(size_t len) { const size_t mn = (size_t)tuning(TUNE_DEEP_MIN_COLS); return len % 8 == 0 && len % 16 != 0 && len >= mn; }
size_t dct_pair_semi_deep_elems(size_t lines, size_t len) { return lines * (6 * dct_pair_split_kpad(len) + pair_kpad<double>(len / 2)); }
int launch_dct_pair_prep16_cols(hipStream_t st, const float* in, size_t n_frames, size_t w, size_t h, double* base,
                                const double* rot1, const double* rot2, bool class_major, const double* rot3) {
    if (n_frames == 0) return SSW_OK;
    const bool semi = dct_pair_can_semi_deep_cols(h);
    if (w > 0xFFFFFFull || h > 0xFFFFFFull || n_frames > 0xFFFFFFull || !(dct_pair_can_deep_cols(h) || semi) || w % 4 != 0) return SSW_ERR_BAD_DIMS;
    if (semi && class_major && !dct_pair_prep_staged_cols_ok(w, true)) return SSW_ERR_BAD_ARG;      // the r3 semi-deep kernels read the natural order only
    const unsigned K8 = (unsigned)dct_pair_split_kpad(h);
    const unsigned K16 = semi ? (unsigned)pair_kpad<double>(h / 2) : (unsigned)dct_pair_split_kpad(h / 2);      // semi: width of the SD plane
    if (dct_pair_efold_cols(h, w, class_major))
        return launch_prep16_cols_l2(st, in, n_frames, w, h, base, rot1, rot2, rot3, class_major, class_major && dct_pair_efold(w), K16);
    if (dct_pair_prep_staged_cols_ok(w, class_major))
        return launch_prep16_cols_staged(st, in, n_frames, w, h, base, rot1, rot2, class_major, semi, K8, K16, dct_pair_efold(w));
    const unsigned units = semi ? (unsigned)(((h / 8 + 1) / 2 + 3) & ~(size_t)3) : K16;
    const unsigned ctile = dct_pair_class_tile(w);
    const unsigned cm = !class_major ? 0u : 1u;      // (the r3 per-class-run mode, 2, knew the ten-class order; lines of such lengths take the staged kernels now)
    const unsigned tiles_e = (units + 31) / 32, tiles_c = cm == 2 ? (unsigned)(w / 256) : (unsigned)((w + 31) / 32);
    const unsigned long long nblk = (unsigned long long)tiles_e * tiles_c * n_frames;
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    const size_t lines = n_frames * w;
    DeepPlanes dp;
    double* p = base;
    const size_t p8 = lines * K8, p16 = lines * K16;
    dp.as = p; dp.bd = p + p8; dp.ad = p + 2 * p8; dp.bs = p + 3 * p8; dp.r1 = p + 4 * p8; dp.r2 = p + 5 * p8;
    p += 6 * p8;
    dp.as2 = p; dp.bd2 = p + p16; dp.ad2 = p + 2 * p16; dp.bs2 = p + 3 * p16;      // semi: as2 = the SD plane, the others unused
    if (semi) pair_prep16_cols_kernel<double, false><<<(unsigned)nblk, 256, 0, st>>>(in, dp, rot1, rot2, (unsigned)w, (unsigned)h, K8, K16, (unsigned)n_frames, tiles_e, tiles_c, 0u, ctile, 0u);
    else      pair_prep16_cols_kernel<double, true><<<(unsigned)nblk, 256, 0, st>>>(in, dp, rot1, rot2, (unsigned)w, (unsigned)h, K8, K16, (unsigned)n_frames, tiles_e, tiles_c, cm, ctile, dct_pair_efold(w) ? 1u : 0u);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// deep inverse pre-passes: same plane order (AS BD AD BS R1 R2 | AS2 BD2 AD2 BS2) with R1 = c[8q], R2 = c[8q+4]
bool dct_pair_can_deep_inv_rows(size_t len) { const size_t mn = (size_t)tuning(TUNE_DEEP_MIN_ROWS); return len % 128 == 0 && len >= mn && len <= 128 * 256; }
static DeepPlanes deep_planes(double* base, size_t lines, size_t len) {
    const size_t p8 = lines * dct_pair_split_kpad(len), p16 = lines * dct_pair_split_kpad(len / 2);
    DeepPlanes dp;
    double* p = base;
    dp.as = p; dp.bd = p + p8; dp.ad = p + 2 * p8; dp.bs = p + 3 * p8; dp.r1 = p + 4 * p8; dp.r2 = p + 5 * p8;
    p += 6 * p8;
    dp.as2 = p; dp.bd2 = p + p16; dp.ad2 = p + 2 * p16; dp.bs2 = p + 3 * p16;
    return dp;
}
int launch_dct_pair_prep16_inv_rows(hipStream_t st, const float* in, size_t n_frames, size_t w, size_t h, double* base,
                                    const double* rot1, const double* rot2, const double* rot3, bool unit_order) {
    if (n_frames == 0) return SSW_OK;
    if (!dct_pair_can_deep_inv_rows(w) || h > 0xFFFFFFull) return SSW_ERR_BAD_DIMS;
    if (unit_order && (h % 16 != 0 || !dct_pair_efold_inv(w))) return SSW_ERR_BAD_ARG;
    const unsigned unit_hup = unit_order ? (unsigned)dct_pair_fused_units(h) : 0u;
    const size_t rows = unit_order ? n_frames * 16 * unit_hup : n_frames * h;       // operand lines
    if (rows > 0xFFFFFFFFull) return SSW_ERR_BAD_DIMS;
    if (dct_pair_efold_inv(w) && dct_pair_inv_prep_light_ok(w, rows))          // r5 A/B: whole rows through LDS, one lane per unit (dct_pair_prep_light.hip)
        return launch_prep16_inv_rows_light(st, in, rows, w, base, rot1, rot2, rot3, (unsigned)dct_pair_split_kpad(w / 2), unit_order ? (unsigned)h : 0u, unit_hup);
    if (dct_pair_efold_inv(w))
        return launch_prep16_inv_rows_l2(st, in, rows, w, base, rot1, rot2, rot3, (unsigned)dct_pair_split_kpad(w / 2), unit_order ? (unsigned)h : 0u, unit_hup);
    if (dct_pair_prep_staged_rows_ok())
        return launch_prep16_inv_rows_staged(st, in, rows, w, base, rot1, rot2, (unsigned)dct_pair_split_kpad(w), (unsigned)dct_pair_split_kpad(w / 2));
    unsigned tp = 1;
    while (tp < w / 128) tp <<= 1;                                  // threads per line, <= 256
    const unsigned lpb = 256 / tp;
    const unsigned long long nblk = (rows + lpb - 1) / lpb;
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    pair_prep16_inv_rows_kernel<double><<<(unsigned)nblk, 256, 0, st>>>(in, deep_planes(base, rows, w), rot1, rot2, (unsigned)rows, (unsigned)w,
                                                                       (unsigned)dct_pair_split_kpad(w), (unsigned)dct_pair_split_kpad(w / 2), tp);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}
int launch_dct_pair_prep16_inv_cols(hipStream_t st, const float* in, size_t n_frames, size_t w, size_t h, double* base,
                                    const double* rot1, const double* rot2, bool class_major, const double* rot3) {
    if (n_frames == 0) return SSW_OK;
    const bool semi = dct_pair_can_semi_deep_cols(h);
    if (w > 0xFFFFFFull || h > 0xFFFFFFull || n_frames > 0xFFFFFFull || !(dct_pair_can_deep_cols(h) || semi)) return SSW_ERR_BAD_DIMS;
    if (semi && class_major && !dct_pair_prep_staged_cols_ok(w, true)) return SSW_ERR_BAD_ARG;      // the r3 semi-deep kernels read the natural order only
    const unsigned K8 = (unsigned)dct_pair_split_kpad(h);
    const unsigned K16 = semi ? (unsigned)pair_kpad<double>(h / 2) : (unsigned)dct_pair_split_kpad(h / 2);      // semi: width of the c[4q+2] plane
    if (dct_pair_efold_cols(h, w, class_major))
        return launch_prep16_inv_cols_l2(st, in, n_frames, w, h, base, rot1, rot2, rot3, class_major, class_major && dct_pair_efold_inv(w), K16);
    if (dct_pair_prep_staged_cols_ok(w, class_major))
        return launch_prep16_inv_cols_staged(st, in, n_frames, w, h, base, rot1, rot2, class_major, semi, K8, K16, class_major && dct_pair_efold_inv(w));
    // groups of 8 units: K16 / 8 covers the padding of the n/16-wide planes; semi: the units (and the R planes' padding up to K8)
    const unsigned groups = semi ? (unsigned)((K8 / 2 + 7) / 8) : K16 / 8, tiles_c = (unsigned)((w + 31) / 32);
    const unsigned long long nblk = (unsigned long long)groups * tiles_c * n_frames;
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    const size_t lines = n_frames * w;
    DeepPlanes dp;
    double* p = base;
    const size_t p8 = lines * K8, p16 = lines * K16;
    dp.as = p; dp.bd = p + p8; dp.ad = p + 2 * p8; dp.bs = p + 3 * p8; dp.r1 = p + 4 * p8; dp.r2 = p + 5 * p8;
    p += 6 * p8;
    dp.as2 = p; dp.bd2 = p + p16; dp.ad2 = p + 2 * p16; dp.bs2 = p + 3 * p16;      // semi: as2 = the c[4q+2] plane, the others unused
    if (semi) pair_prep16_inv_cols_kernel<double, false><<<(unsigned)nblk, 256, 0, st>>>(in, dp, rot1, rot2, (unsigned)w, (unsigned)h, K8, K16, (unsigned)n_frames, groups, tiles_c, 0u, (unsigned)w);
    else      pair_prep16_inv_cols_kernel<double, true><<<(unsigned)nblk, 256, 0, st>>>(in, dp, rot1, rot2, (unsigned)w, (unsigned)h, K8, K16, (unsigned)n_frames, groups, tiles_c, class_major ? (dct_pair_efold_inv(w) ? 2u : 1u) : 0u, dct_pair_class_tile(w));
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

}  // namespace ssw
