// Even/odd-folded basis GEMMs in f64 ("canonical" precision): v_mfma_f64_16x16x4_f64, f64 half
// bases, the image operand widened from f32 after the LDS read.  Same decomposition as
// dct_folded.hip (see its header), but nothing is rounded before the final result:
//   forward:  (double)x[n] +/- (double)x[N-1-n] is exact, products and sums are f64
//   inverse:  P + Q and P - Q are formed in f64 and rounded once to f32
// so the output is still the correctly rounded transform almost everywhere (what the oracle's f64
// backend returns), at half the multiply-adds of the dense f64 kernels in dct.hip.
//
// Block: 256 threads = 4 waves as 2 x 2; block tile 128 rows x 64 pairs (rows) or 64 pairs x 128
// columns (cols), k-step 16; each wave 16 MFMA 16x16 tiles (8 for acc1, 8 for acc2) = 128
// accumulator registers; LDS double-buffered, one barrier per k-step; 2 blocks per CU.
// Lane l: li = l & 15 (row/col inside a 16x16 tile), lq = l >> 4 (k = 4 lq + j at MFMA step j).
#include "dct_common.hpp"

namespace ssw {

constexpr int DT = 256;
constexpr int DBK = 16;
constexpr int DLU = DBK + 4;            // f32 image tile rows (k-contiguous): 20 floats
constexpr int DLB = DBK + 2;            // f64 basis tile rows: 18 doubles (144 B, 16-B aligned)
constexpr int DLN = 128 + 4;            // f32 image tile rows (n-contiguous)

__global__ void make_half_basis_f64_kernel(size_t n, bool inverse, int parity, size_t kpad, double* out) {
    const size_t nh = n / 2, total = nh * kpad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t o = i / kpad, s = i % kpad;
        if (s >= nh) { out[i] = 0.0; continue; }
        const size_t freq = inverse ? 2 * s + parity : 2 * o + parity;
        const size_t pos = inverse ? o : s;
        unsigned long long a = (unsigned long long)freq * (2ull * pos + 1ull);
        a %= 4ull * n;
        const double c = cospi((double)a / (double)(2ull * n));
        out[i] = !inverse ? 2.0 * c : (freq == 0 ? 0.25 : 0.5 * c);
    }
}

int launch_make_half_basis_f64(hipStream_t st, size_t n, bool inverse, int parity, double* out) {
    const size_t total = (n / 2) * half_basis_kpad(n);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    make_half_basis_f64_kernel<<<blocks ? blocks : 1, 256, 0, st>>>(n, inverse, parity, half_basis_kpad(n), out);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

__device__ inline void load4d(const double* __restrict__ p, double v[4]) {
    const f64x4 t = *reinterpret_cast<const f64x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}

// ---------------------------------------------------------------------------------------------
// Row pass.  X: M x W, OUT: M x W.
// ---------------------------------------------------------------------------------------------
template <bool INVERSE>
__global__ __launch_bounds__(DT, 2) void dct_rows_folded_f64_kernel(
    const float* __restrict__ X, const double* __restrict__ B1g, const double* __restrict__ B2g,
    float* __restrict__ OUT, unsigned M, unsigned W, unsigned Kp, unsigned tiles_m, unsigned tiles_n,
    Epilogue ep) {
    // forward: sU = x[k0 .. k0+15], sV = the mirrored block x[W-16-k0 .. W-1-k0] (natural order)
    // inverse: sU | sV = x[2 k0 .. 2 k0 + 31] (even/odd interleaved as in memory)
    __shared__ __attribute__((aligned(16))) float sU[2][128 * DLU];
    __shared__ __attribute__((aligned(16))) float sV[2][128 * DLU];
    __shared__ __attribute__((aligned(16))) double sB1[2][64 * DLB];
    __shared__ __attribute__((aligned(16))) double sB2[2][64 * DLB];

    const unsigned Nh = W / 2;
    unsigned tm, tn;
    tile_of_block(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
    const unsigned m0 = tm * 128, p0 = tn * 64;
    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63, wave = tid >> 6;
    const unsigned wm = (wave >> 1) * 64, wn = (wave & 1) * 32;
    const unsigned li = lane & 15, lq = lane >> 4;

    const unsigned srow = tid >> 2, sq = tid & 3;                 // staging: row = tid/4 (+64p), quad = tid%4
    const float* a_rows[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        unsigned r = m0 + srow + 64 * p; r = r < M ? r : M - 1;
        a_rows[p] = X + (size_t)r * W;
    }
    unsigned rb = p0 + srow; rb = rb < Nh ? rb : Nh - 1;
    const double* b1_row = B1g + (size_t)rb * Kp;
    const double* b2_row = B2g + (size_t)rb * Kp;

    f64x4 acc1[4][2], acc2[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { acc1[i][j] = (f64x4){0, 0, 0, 0}; acc2[i][j] = (f64x4){0, 0, 0, 0}; }

    f32x4 ru[2], rv[2];
    double rb1[4], rb2[4];
    auto gload = [&](unsigned t) {
        const unsigned k = t * DBK + 4 * sq;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (!INVERSE) {
                ru[p] = *reinterpret_cast<const f32x4*>(a_rows[p] + k);
                rv[p] = *reinterpret_cast<const f32x4*>(a_rows[p] + (W - 4 - k));
            } else {
                const unsigned kc = k < Nh - 4 ? k : Nh - 4;
                ru[p] = *reinterpret_cast<const f32x4*>(a_rows[p] + 2 * kc);
                rv[p] = *reinterpret_cast<const f32x4*>(a_rows[p] + 2 * kc + 4);
            }
        }
        load4d(b1_row + k, rb1);
        load4d(b2_row + k, rb2);
    };
    auto lstore = [&](unsigned buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const unsigned row = srow + 64 * p;
            if (!INVERSE) {
                *reinterpret_cast<f32x4*>(&sU[buf][row * DLU + 4 * sq]) = ru[p];
                *reinterpret_cast<f32x4*>(&sV[buf][row * DLU + 12 - 4 * sq]) = rv[p];
            } else {
                float* base = (sq < 2 ? sU[buf] : sV[buf]) + row * DLU + 8 * (sq & 1);
                *reinterpret_cast<f32x4*>(base) = ru[p];
                *reinterpret_cast<f32x4*>(base + 4) = rv[p];
            }
        }
        double* d1 = &sB1[buf][srow * DLB + 4 * sq];
        double* d2 = &sB2[buf][srow * DLB + 4 * sq];
#pragma unroll
        for (int e = 0; e < 4; ++e) { d1[e] = rb1[e]; d2[e] = rb2[e]; }
    };

    const unsigned nk = Kp / DBK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (unsigned t = 0; t < nk; ++t) {
        const unsigned cur = t & 1;
        if (t + 1 < nk) gload(t + 1);
        double b1[2][4], b2[2][4];
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            load4d(&sB1[cur][(wn + 16 * jn + li) * DLB + 4 * lq], b1[jn]);
            load4d(&sB2[cur][(wn + 16 * jn + li) * DLB + 4 * lq], b2[jn]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned row = wm + 16 * i + li;
            f32x4 u, v;
            if (!INVERSE) {
                u = *reinterpret_cast<const f32x4*>(&sU[cur][row * DLU + 4 * lq]);
                v = *reinterpret_cast<const f32x4*>(&sV[cur][row * DLU + 4 * (3 - lq)]);
            } else {
                const float* base = (lq < 2 ? sU[cur] : sV[cur]) + row * DLU + 8 * (lq & 1);
                u = *reinterpret_cast<const f32x4*>(base);
                v = *reinterpret_cast<const f32x4*>(base + 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double a1, a2;
                if (!INVERSE) {            // x[k] +/- x[W-1-k], exact in f64
                    a1 = (double)u[j] + (double)v[3 - j];
                    a2 = (double)u[j] - (double)v[3 - j];
                } else {                   // even / odd input of frequency pair 4 lq + j
                    const float e = j < 2 ? u[2 * j] : v[2 * j - 4];
                    const float o = j < 2 ? u[2 * j + 1] : v[2 * j - 3];
                    a1 = (double)e;
                    a2 = (double)o;
                }
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    acc1[i][jn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1[jn][j], acc1[i][jn], 0, 0, 0);
                    acc2[i][jn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2[jn][j], acc2[i][jn], 0, 0, 0);
                }
            }
        }
        if (t + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }

    // D map of 16x16x4 f64: col = lane & 15, row = (lane >> 4) + 4 reg
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
        const unsigned pair = p0 + wn + 16 * jn + li;
        if (pair >= Nh) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned row = m0 + wm + 16 * i + lq + 4 * r;
                if (row >= M) continue;
                float* o = OUT + (size_t)row * W;
                if (!INVERSE) {
                    const f32x2 v = {apply_epilogue(ep, (float)acc1[i][jn][r], 2 * pair),
                                     apply_epilogue(ep, (float)acc2[i][jn][r], 2 * pair + 1)};
                    *reinterpret_cast<f32x2*>(o + 2 * pair) = v;
                } else {
                    o[pair] = apply_epilogue(ep, (float)(acc1[i][jn][r] + acc2[i][jn][r]), pair);
                    o[W - 1 - pair] = apply_epilogue(ep, (float)(acc1[i][jn][r] - acc2[i][jn][r]), W - 1 - pair);
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// Column pass.  One frame per z: IN, OUT: H x W.
// ---------------------------------------------------------------------------------------------
template <bool INVERSE>
__global__ __launch_bounds__(DT, 2) void dct_cols_folded_f64_kernel(
    const double* __restrict__ A1g, const double* __restrict__ A2g, const float* __restrict__ INz,
    float* __restrict__ OUTz, unsigned H, unsigned W, unsigned Kp, unsigned tiles_m, unsigned tiles_n,
    unsigned tiles_per_frame, Epilogue ep) {
    // forward: sU = rows k0+kk, sV = rows H-1-(k0+kk);  inverse: sU = rows 2(k0+kk), sV = rows 2(k0+kk)+1
    __shared__ __attribute__((aligned(16))) double sA1[2][64 * DLB];
    __shared__ __attribute__((aligned(16))) double sA2[2][64 * DLB];
    __shared__ __attribute__((aligned(16))) float sU[2][DBK * DLN];
    __shared__ __attribute__((aligned(16))) float sV[2][DBK * DLN];

    const unsigned Hh = H / 2;
    const unsigned z = blockIdx.x / tiles_per_frame;
    unsigned tm, tn;
    tile_of_block(blockIdx.x % tiles_per_frame, tiles_per_frame, tiles_m, tiles_n, tm, tn);
    const unsigned p0 = tm * 64, n0 = tn * 128;
    const float* __restrict__ IN = INz + (size_t)z * H * W;
    float* __restrict__ OUT = OUTz + (size_t)z * H * W;

    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63, wave = tid >> 6;
    const unsigned wm = (wave >> 1) * 32, wn = (wave & 1) * 64;
    const unsigned li = lane & 15, lq = lane >> 4;

    const unsigned srow = tid >> 2, sq = tid & 3;                 // basis: row = tid/4, quad of doubles
    const unsigned bk = tid >> 5, bn = (tid & 31) * 4;            // image: k row = bk + 8p, n quad
    unsigned ra = p0 + srow; ra = ra < Hh ? ra : Hh - 1;
    const double* a1_row = A1g + (size_t)ra * Kp;
    const double* a2_row = A2g + (size_t)ra * Kp;
    const unsigned ncol = (n0 + bn) < W ? (n0 + bn) : W - 4;

    f64x4 acc1[2][4], acc2[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc1[i][j] = (f64x4){0, 0, 0, 0}; acc2[i][j] = (f64x4){0, 0, 0, 0}; }

    double ra1[4], ra2[4];
    f32x4 ru[2], rv[2];
    auto gload = [&](unsigned t) {
        load4d(a1_row + t * DBK + 4 * sq, ra1);
        load4d(a2_row + t * DBK + 4 * sq, ra2);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            unsigned kk = t * DBK + bk + 8 * p;
            kk = kk < Hh ? kk : Hh - 1;
            if (!INVERSE) {
                ru[p] = *reinterpret_cast<const f32x4*>(IN + (size_t)kk * W + ncol);
                rv[p] = *reinterpret_cast<const f32x4*>(IN + (size_t)(H - 1 - kk) * W + ncol);
            } else {
                ru[p] = *reinterpret_cast<const f32x4*>(IN + (size_t)(2 * kk) * W + ncol);
                rv[p] = *reinterpret_cast<const f32x4*>(IN + (size_t)(2 * kk + 1) * W + ncol);
            }
        }
    };
    auto lstore = [&](unsigned buf) {
        double* d1 = &sA1[buf][srow * DLB + 4 * sq];
        double* d2 = &sA2[buf][srow * DLB + 4 * sq];
#pragma unroll
        for (int e = 0; e < 4; ++e) { d1[e] = ra1[e]; d2[e] = ra2[e]; }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            *reinterpret_cast<f32x4*>(&sU[buf][(bk + 8 * p) * DLN + bn]) = ru[p];
            *reinterpret_cast<f32x4*>(&sV[buf][(bk + 8 * p) * DLN + bn]) = rv[p];
        }
    };

    const unsigned nk = Kp / DBK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (unsigned t = 0; t < nk; ++t) {
        const unsigned cur = t & 1;
        if (t + 1 < nk) gload(t + 1);
        double a1[2][4], a2[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            load4d(&sA1[cur][(wm + 16 * i + li) * DLB + 4 * lq], a1[i]);
            load4d(&sA2[cur][(wm + 16 * i + li) * DLB + 4 * lq], a2[i]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned krow = 4 * lq + j;
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) {
                const double u = (double)sU[cur][krow * DLN + wn + 16 * jn + li];
                const double v = (double)sV[cur][krow * DLN + wn + 16 * jn + li];
                const double b1 = INVERSE ? u : u + v;
                const double b2 = INVERSE ? v : u - v;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc1[i][jn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[i][j], b1, acc1[i][jn], 0, 0, 0);
                    acc2[i][jn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[i][j], b2, acc2[i][jn], 0, 0, 0);
                }
            }
        }
        if (t + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
        const unsigned col = n0 + wn + 16 * jn + li;
        if (col >= W) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned pair = p0 + wm + 16 * i + lq + 4 * r;
                if (pair >= Hh) continue;
                if (!INVERSE) {
                    OUT[(size_t)(2 * pair) * W + col] = apply_epilogue(ep, (float)acc1[i][jn][r], 2 * pair);
                    OUT[(size_t)(2 * pair + 1) * W + col] = apply_epilogue(ep, (float)acc2[i][jn][r], 2 * pair + 1);
                } else {
                    OUT[(size_t)pair * W + col] = apply_epilogue(ep, (float)(acc1[i][jn][r] + acc2[i][jn][r]), pair);
                    OUT[(size_t)(H - 1 - pair) * W + col] =
                        apply_epilogue(ep, (float)(acc1[i][jn][r] - acc2[i][jn][r]), H - 1 - pair);
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------------------------
int launch_dct_rows_folded_f64(hipStream_t st, bool inverse, const float* in, float* out, size_t rows,
                               size_t w, const double* b1, const double* b2, Epilogue ep) {
    if (rows == 0) return SSW_OK;
    if (rows > 0xFFFFFFFFull || w > 0xFFFFFFull) return SSW_ERR_BAD_DIMS;
    const unsigned M = (unsigned)rows, W = (unsigned)w, Nh = W / 2, Kp = (unsigned)half_basis_kpad(w);
    const unsigned tiles_m = (M + 127) / 128, tiles_n = (Nh + 63) / 64;
    const unsigned long long nblk = (unsigned long long)tiles_m * tiles_n;
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    if (inverse) dct_rows_folded_f64_kernel<true><<<(unsigned)nblk, DT, 0, st>>>(in, b1, b2, out, M, W, Kp, tiles_m, tiles_n, ep);
    else         dct_rows_folded_f64_kernel<false><<<(unsigned)nblk, DT, 0, st>>>(in, b1, b2, out, M, W, Kp, tiles_m, tiles_n, ep);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

int launch_dct_cols_folded_f64(hipStream_t st, bool inverse, const float* in, float* out, size_t n_frames,
                               size_t w, size_t h, const double* a1, const double* a2, Epilogue ep) {
    if (n_frames == 0) return SSW_OK;
    if (w > 0xFFFFFFull || h > 0xFFFFFFull) return SSW_ERR_BAD_DIMS;
    const unsigned H = (unsigned)h, W = (unsigned)w, Hh = H / 2, Kp = (unsigned)half_basis_kpad(h);
    const unsigned tiles_m = (Hh + 63) / 64, tiles_n = (W + 127) / 128;
    const unsigned tiles_per_frame = tiles_m * tiles_n;
    const unsigned long long nblk = (unsigned long long)tiles_per_frame * n_frames;
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    if (inverse) dct_cols_folded_f64_kernel<true><<<(unsigned)nblk, DT, 0, st>>>(a1, a2, in, out, H, W, Kp, tiles_m, tiles_n, tiles_per_frame, ep);
    else         dct_cols_folded_f64_kernel<false><<<(unsigned)nblk, DT, 0, st>>>(a1, a2, in, out, H, W, Kp, tiles_m, tiles_n, tiles_per_frame, ep);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

}  // namespace ssw
