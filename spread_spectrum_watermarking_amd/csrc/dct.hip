// Full-frame separable 2-D DCT-II / DCT-III as two basis GEMMs on the matrix cores.
//
// Replaces the bodies of dct2d::dct2_2d's row and column loops
// (/root/reference/src/dct2d.rs:129-170 and :172-206), i.e. the `h` + `w` calls into
// rustdct's 1-D kernels, by
//     row pass    T[r][v] = sum_c  Y[r][c] * B_W[v][c]        (NT GEMM, batched over all rows)
//     column pass C[u][v] = sum_r  B_H[u][r] * T[r][v]        (NN GEMM, one per frame)
// with the reference's scaling folded into the basis (exact: powers of two) and its
// rounding points kept (f32 store between the passes; separate 4/(W*H) multiply).
//
// f32 path: v_mfma_f32_32x32x2_f32 -- bit-for-bit an fmaf chain over k, 64 FLOP/clk/SIMD,
//           157.3 TFLOP/s chip peak.  Block tile 128x128x32, 4 waves as 2x2, each wave 2x2 MFMA
//           tiles (64 accumulator VGPRs), LDS double-buffered, one barrier per k-step.
// f64 path: v_mfma_f64_16x16x4_f64 with an f64 basis, result rounded once to f32
//           ("canonical": the correctly rounded transform almost everywhere).
#include "dct_common.hpp"

namespace ssw {

// ---------------------------------------------------------------------------------------------
// Basis generation (device, f64 cospi with exact integer argument reduction).
// ---------------------------------------------------------------------------------------------
__device__ inline double basis_value(size_t out_idx, size_t sum_idx, size_t n, bool inverse) {
    // forward: 2 cos(pi * k (2 j + 1) / 2N), k = out, j = sum
    // inverse: j == 0 ? 1/4 : cos(pi * j (2 k + 1) / 2N) / 2,  k = out, j = sum
    unsigned long long a = inverse ? (unsigned long long)sum_idx * (2ull * out_idx + 1ull)
                                   : (unsigned long long)out_idx * (2ull * sum_idx + 1ull);
    a %= 4ull * n;                                  // cos(pi a / 2N) has period 4N in a
    const double c = cospi((double)a / (double)(2ull * n));
    if (!inverse) return 2.0 * c;
    return sum_idx == 0 ? 0.25 : 0.5 * c;
}

// Basis rows are zero-padded to a multiple of 32 along the sum axis (row stride = dense_basis_kpad):
// the aligned GEMM main loops then run without tail predication (operand loads past the end of the
// sum axis are clamped to valid addresses and meet a zero basis entry).
size_t dense_basis_kpad(size_t n) { return (n + 31) / 32 * 32; }

__global__ void make_basis_f32_kernel(size_t n, size_t kpad, bool inverse, float* out) {
    const size_t total = n * kpad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x)
        out[i] = (i % kpad) < n ? (float)basis_value(i / kpad, i % kpad, n, inverse) : 0.0f;
}
__global__ void make_basis_f64_kernel(size_t n, size_t kpad, bool inverse, double* out) {
    const size_t total = n * kpad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x)
        out[i] = (i % kpad) < n ? basis_value(i / kpad, i % kpad, n, inverse) : 0.0;
}

int launch_make_basis_f32(hipStream_t st, size_t n, bool inverse, float* out) {
    const size_t total = n * dense_basis_kpad(n);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    make_basis_f32_kernel<<<blocks, 256, 0, st>>>(n, dense_basis_kpad(n), inverse, out);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}
int launch_make_basis_f64(hipStream_t st, size_t n, bool inverse, double* out) {
    const size_t total = n * dense_basis_kpad(n);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    make_basis_f64_kernel<<<blocks, 256, 0, st>>>(n, dense_basis_kpad(n), inverse, out);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// ---------------------------------------------------------------------------------------------
// Tile geometry
// ---------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDK = BK + 4;          // k-contiguous tiles: 36-float rows -> conflict-free ds_read_b128
constexpr int LDN = BN + 4;          // n-contiguous tile of the NN kernel
constexpr int THREADS = 256;
// f32 path: the MFMA accumulates an exact-f32 fma chain; a chain over all K (up to 7680) lets the
// rounding error grow with the running sum.  Every ACC_CHUNK k-tiles (256 k) the chain is folded
// into a second f32 accumulator and restarted, which bounds each chain by its chunk's partial sum
// (two-level summation): ~8x smaller error on the DC-dominated low frequencies at 4K for 64 extra
// VGPRs and 64 v_add_f32 per 128 MFMAs.
constexpr int ACC_CHUNK = 8;

// ---------------------------------------------------------------------------------------------
// f32 NT kernel (row pass):  OUT[m][n] = sum_k A[m][k] * B[n][k],  A: MxK, B: NxK, K-contiguous.
// ---------------------------------------------------------------------------------------------
template <bool ALIGNED>
__global__ __launch_bounds__(THREADS, 2) void dct_rows_f32_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ OUT,
    unsigned M, unsigned N, unsigned K, unsigned Kb, unsigned tiles_m, unsigned tiles_n, Epilogue ep) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][BM * LDK];   // [buf][A|B][row*LDK + k]

    unsigned tm, tn;
    tile_of_block(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
    const unsigned m0 = tm * BM, n0 = tn * BN;

    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63, wave = tid >> 6;
    const unsigned wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const unsigned lr = lane & 31, lh = lane >> 5;

    // staging: thread -> (row = tid/8 + 32p, k4 = tid%8)
    const unsigned srow = tid >> 3, sk = (tid & 7) * 4;
    const float* a_rows[4];
    const float* b_rows[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned ra = m0 + srow + 32 * p; ra = ra < M ? ra : M - 1;     // clamp: rows past the edge
        unsigned rb = n0 + srow + 32 * p; rb = rb < N ? rb : N - 1;     // feed outputs never stored
        a_rows[p] = A + (size_t)ra * K;
        b_rows[p] = B + (size_t)rb * Kb;
    }

    f32x16 acc[2][2], tot[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

    const unsigned nk = ALIGNED ? Kb / BK : (K + BK - 1) / BK;
    f32x4 ra[4], rb[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        ra[p] = load_img_k4<ALIGNED>(a_rows[p], sk, K);
        rb[p] = load_basis_k4<ALIGNED>(b_rows[p], sk, K);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        *reinterpret_cast<f32x4*>(&lds[0][0][(srow + 32 * p) * LDK + sk]) = ra[p];
        *reinterpret_cast<f32x4*>(&lds[0][1][(srow + 32 * p) * LDK + sk]) = rb[p];
    }
    __syncthreads();

    for (unsigned t = 0; t < nk; ++t) {
        const unsigned cur = t & 1;
        if (t + 1 < nk) {
            const unsigned k = (t + 1) * BK + sk;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                ra[p] = load_img_k4<ALIGNED>(a_rows[p], k, K);
                rb[p] = load_basis_k4<ALIGNED>(b_rows[p], k, K);
            }
        }
        const float* As = lds[cur][0];
        const float* Bs = lds[cur][1];
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            // lane half h holds k = 8 kg + 4 h + j at MFMA step j (same map for A and B)
            f32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const f32x4*>(&As[(wm + 32 * i + lr) * LDK + kg * 8 + lh * 4]);
                b[i] = *reinterpret_cast<const f32x4*>(&Bs[(wn + 32 * i + lr) * LDK + kg * 8 + lh * 4]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b[jn][j], acc[i][jn], 0, 0, 0);
        }
        if ((t + 1) % ACC_CHUNK == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    tot[i][jn] += acc[i][jn];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;
                }
        }
        if (t + 1 < nk) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                *reinterpret_cast<f32x4*>(&lds[cur ^ 1][0][(srow + 32 * p) * LDK + sk]) = ra[p];
                *reinterpret_cast<f32x4*>(&lds[cur ^ 1][1][(srow + 32 * p) * LDK + sk]) = rb[p];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) acc[i][jn] = tot[i][jn] + acc[i][jn];

    // C/D map of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const unsigned col = n0 + wn + 32 * jn + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < M && col < N)
                    OUT[(size_t)row * N + col] = apply_epilogue(ep, acc[i][jn][r], col);
            }
        }
}

// ---------------------------------------------------------------------------------------------
// f32 NN kernel (column pass):  OUT[z][m][n] = sum_k A[m][k] * B[z][k][n]
//   A: basis MxK (K-contiguous), B: per-frame KxN plane (N-contiguous).
// ---------------------------------------------------------------------------------------------
template <bool ALIGNED>
__global__ __launch_bounds__(THREADS, 2) void dct_cols_f32_kernel(
    const float* __restrict__ A, const float* __restrict__ Bz, float* __restrict__ OUTz,
    unsigned M, unsigned N, unsigned K, unsigned Kb, unsigned tiles_m, unsigned tiles_n,
    unsigned tiles_per_frame, Epilogue ep) {
    __shared__ __attribute__((aligned(16))) float ldsA[2][BM * LDK];
    __shared__ __attribute__((aligned(16))) float ldsB[2][BK * LDN];

    // frames are the slowest grid dimension; the XCD/L2 map is applied inside one frame
    const unsigned z = blockIdx.x / tiles_per_frame;
    unsigned tm, tn;
    tile_of_block(blockIdx.x % tiles_per_frame, tiles_per_frame, tiles_m, tiles_n, tm, tn);
    const unsigned m0 = tm * BM, n0 = tn * BN;
    const float* __restrict__ B = Bz + (size_t)z * K * N;
    float* __restrict__ OUT = OUTz + (size_t)z * M * N;

    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63, wave = tid >> 6;
    const unsigned wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const unsigned lr = lane & 31, lh = lane >> 5;

    const unsigned srow = tid >> 3, sk = (tid & 7) * 4;          // A staging (k-contiguous)
    const unsigned bk = tid >> 5, bn = (tid & 31) * 4;           // B staging: k row = bk + 8p, n4 = bn
    const float* a_rows[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned ra = m0 + srow + 32 * p; ra = ra < M ? ra : M - 1;
        a_rows[p] = A + (size_t)ra * Kb;
    }

    f32x16 acc[2][2], tot[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

    const unsigned nk = ALIGNED ? Kb / BK : (K + BK - 1) / BK;
    f32x4 ra[4], rb[4];
    auto gload = [&](unsigned t) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            ra[p] = load_basis_k4<ALIGNED>(a_rows[p], t * BK + sk, K);
            const unsigned kk = t * BK + bk + 8 * p;
            rb[p] = load_img_n4<ALIGNED>(B, kk, K, n0 + bn, N);
        }
    };
    auto lstore = [&](unsigned buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<f32x4*>(&ldsA[buf][(srow + 32 * p) * LDK + sk]) = ra[p];
            *reinterpret_cast<f32x4*>(&ldsB[buf][(bk + 8 * p) * LDN + bn]) = rb[p];
        }
    };
    gload(0);
    lstore(0);
    __syncthreads();

    for (unsigned t = 0; t < nk; ++t) {
        const unsigned cur = t & 1;
        if (t + 1 < nk) gload(t + 1);
        const float* As = ldsA[cur];
        const float* Bs = ldsB[cur];
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            f32x4 a[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[i] = *reinterpret_cast<const f32x4*>(&As[(wm + 32 * i + lr) * LDK + kg * 8 + lh * 4]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned krow = kg * 8 + lh * 4 + j;       // same k map as the A fragment
                float b[2];
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) b[jn] = Bs[krow * LDN + wn + 32 * jn + lr];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b[jn], acc[i][jn], 0, 0, 0);
            }
        }
        if ((t + 1) % ACC_CHUNK == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    tot[i][jn] += acc[i][jn];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;
                }
        }
        if (t + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) acc[i][jn] = tot[i][jn] + acc[i][jn];

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const unsigned col = n0 + wn + 32 * jn + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < M && col < N)
                    OUT[(size_t)row * N + col] = apply_epilogue(ep, acc[i][jn][r], row);
            }
        }
}

// ---------------------------------------------------------------------------------------------
// f64 kernels: v_mfma_f64_16x16x4_f64.  Lane l supplies A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15]; D holds 4 f64 per lane: col = l & 15, row = (l >> 4) + 4 reg.
// Block tile 128x128x16, 4 waves as 2x2, each wave 4x4 MFMA tiles (128 accumulator VGPRs).
// The image operand stays f32 in LDS and is widened after the read (exact); the basis is f64.
// Lane quarter q holds k = 16 kg + 4 q + j at MFMA step j (same map for both operands).
// ---------------------------------------------------------------------------------------------
constexpr int BK64 = 16;
constexpr int LDK32 = BK64 + 4;      // f32 k-contiguous tile rows (20 floats = 80 B, 16-B aligned)
constexpr int LDK64 = BK64 + 2;      // f64 k-contiguous tile rows (18 doubles = 144 B, 16-B aligned)
constexpr int LDN64 = BN + 4;

template <bool ALIGNED>
__device__ inline void load_k4_f64(const double* __restrict__ row, unsigned k, unsigned K, double v[4]) {
    v[0] = v[1] = v[2] = v[3] = 0.0;
    if (ALIGNED) {            // zero-padded rows: no predicate
        const f64x4 t = *reinterpret_cast<const f64x4*>(row + k);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k + e < K) v[e] = row[k + e];
    }
}

// Row pass, f64:  OUT[m][n] = sum_k A[m][k] (f32 image) * B[n][k] (f64 basis)
template <bool ALIGNED>
__global__ __launch_bounds__(THREADS, 2) void dct_rows_f64_kernel(
    const float* __restrict__ A, const double* __restrict__ B, float* __restrict__ OUT,
    unsigned M, unsigned N, unsigned K, unsigned Kb, unsigned tiles_m, unsigned tiles_n, Epilogue ep) {
    __shared__ __attribute__((aligned(16))) float ldsA[2][BM * LDK32];
    __shared__ __attribute__((aligned(16))) double ldsB[2][BN * LDK64];

    unsigned tm, tn;
    tile_of_block(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
    const unsigned m0 = tm * BM, n0 = tn * BN;
    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63, wave = tid >> 6;
    const unsigned wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const unsigned li = lane & 15, lq = lane >> 4;

    // staging: 128 rows x 16 k = 512 quads per operand; thread -> (row = tid/4 + 64p, k4 = tid%4)
    const unsigned srow = tid >> 2, sk = (tid & 3) * 4;
    const float* a_rows[2];
    const double* b_rows[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        unsigned ra = m0 + srow + 64 * p; ra = ra < M ? ra : M - 1;
        unsigned rb = n0 + srow + 64 * p; rb = rb < N ? rb : N - 1;
        a_rows[p] = A + (size_t)ra * K;
        b_rows[p] = B + (size_t)rb * Kb;
    }

    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};

    const unsigned nk = ALIGNED ? Kb / BK64 : (K + BK64 - 1) / BK64;
    f32x4 ra[2];
    double rb[2][4];
    auto gload = [&](unsigned t) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            ra[p] = load_img_k4<ALIGNED>(a_rows[p], t * BK64 + sk, K);
            load_k4_f64<ALIGNED>(b_rows[p], t * BK64 + sk, K, rb[p]);
        }
    };
    auto lstore = [&](unsigned buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            *reinterpret_cast<f32x4*>(&ldsA[buf][(srow + 64 * p) * LDK32 + sk]) = ra[p];
            double* d = &ldsB[buf][(srow + 64 * p) * LDK64 + sk];
            d[0] = rb[p][0]; d[1] = rb[p][1]; d[2] = rb[p][2]; d[3] = rb[p][3];
        }
    };
    gload(0);
    lstore(0);
    __syncthreads();

    for (unsigned t = 0; t < nk; ++t) {
        const unsigned cur = t & 1;
        if (t + 1 < nk) gload(t + 1);
        const float* As = ldsA[cur];
        const double* Bs = ldsB[cur];
        f32x4 a[4];
        double b[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = *reinterpret_cast<const f32x4*>(&As[(wm + 16 * i + li) * LDK32 + lq * 4]);
            const double* bp = &Bs[(wn + 16 * i + li) * LDK64 + lq * 4];
            b[i][0] = bp[0]; b[i][1] = bp[1]; b[i][2] = bp[2]; b[i][3] = bp[3];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double av = (double)a[i][j];
#pragma unroll
                for (int jn = 0; jn < 4; ++jn)
                    acc[i][jn] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b[jn][j], acc[i][jn], 0, 0, 0);
            }
        if (t + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) {
            const unsigned col = n0 + wn + 16 * jn + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned row = m0 + wm + 16 * i + lq + 4 * r;
                if (row < M && col < N)
                    OUT[(size_t)row * N + col] = apply_epilogue(ep, (float)acc[i][jn][r], col);
            }
        }
}

// Column pass, f64:  OUT[z][m][n] = sum_k A[m][k] (f64 basis) * B[z][k][n] (f32 image)
template <bool ALIGNED>
__global__ __launch_bounds__(THREADS, 2) void dct_cols_f64_kernel(
    const double* __restrict__ A, const float* __restrict__ Bz, float* __restrict__ OUTz,
    unsigned M, unsigned N, unsigned K, unsigned Kb, unsigned tiles_m, unsigned tiles_n,
    unsigned tiles_per_frame, Epilogue ep) {
    __shared__ __attribute__((aligned(16))) double ldsA[2][BM * LDK64];
    __shared__ __attribute__((aligned(16))) float ldsB[2][BK64 * LDN64];

    const unsigned z = blockIdx.x / tiles_per_frame;
    unsigned tm, tn;
    tile_of_block(blockIdx.x % tiles_per_frame, tiles_per_frame, tiles_m, tiles_n, tm, tn);
    const unsigned m0 = tm * BM, n0 = tn * BN;
    const float* __restrict__ B = Bz + (size_t)z * K * N;
    float* __restrict__ OUT = OUTz + (size_t)z * M * N;

    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63, wave = tid >> 6;
    const unsigned wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const unsigned li = lane & 15, lq = lane >> 4;

    const unsigned srow = tid >> 2, sk = (tid & 3) * 4;          // A: row = srow + 64p, k4 = sk
    const unsigned bk = tid >> 5, bn = (tid & 31) * 4;           // B: k row = bk + 8p, n4 = bn
    const double* a_rows[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        unsigned ra = m0 + srow + 64 * p; ra = ra < M ? ra : M - 1;
        a_rows[p] = A + (size_t)ra * Kb;
    }

    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};

    const unsigned nk = ALIGNED ? Kb / BK64 : (K + BK64 - 1) / BK64;
    double ra[2][4];
    f32x4 rb[2];
    auto gload = [&](unsigned t) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            load_k4_f64<ALIGNED>(a_rows[p], t * BK64 + sk, K, ra[p]);
            const unsigned kk = t * BK64 + bk + 8 * p;
            rb[p] = load_img_n4<ALIGNED>(B, kk, K, n0 + bn, N);
        }
    };
    auto lstore = [&](unsigned buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            double* d = &ldsA[buf][(srow + 64 * p) * LDK64 + sk];
            d[0] = ra[p][0]; d[1] = ra[p][1]; d[2] = ra[p][2]; d[3] = ra[p][3];
            *reinterpret_cast<f32x4*>(&ldsB[buf][(bk + 8 * p) * LDN64 + bn]) = rb[p];
        }
    };
    gload(0);
    lstore(0);
    __syncthreads();

    for (unsigned t = 0; t < nk; ++t) {
        const unsigned cur = t & 1;
        if (t + 1 < nk) gload(t + 1);
        const double* As = ldsA[cur];
        const float* Bs = ldsB[cur];
        double a[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double* ap = &As[(wm + 16 * i + li) * LDK64 + lq * 4];
            a[i][0] = ap[0]; a[i][1] = ap[1]; a[i][2] = ap[2]; a[i][3] = ap[3];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned krow = lq * 4 + j;
            double b[4];
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) b[jn] = (double)Bs[krow * LDN64 + wn + 16 * jn + li];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jn = 0; jn < 4; ++jn)
                    acc[i][jn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][j], b[jn], acc[i][jn], 0, 0, 0);
        }
        if (t + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) {
            const unsigned col = n0 + wn + 16 * jn + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned row = m0 + wm + 16 * i + lq + 4 * r;
                if (row < M && col < N)
                    OUT[(size_t)row * N + col] = apply_epilogue(ep, (float)acc[i][jn][r], row);
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------------------------
int launch_dct_rows(hipStream_t st, int precision, const float* in, float* out, size_t rows, size_t w,
                    const void* basis, Epilogue ep) {
    if (rows == 0 || w == 0) return SSW_OK;
    if (rows > 0xFFFFFFFFull || w > 0xFFFFFFull) return SSW_ERR_BAD_DIMS;
    const unsigned M = (unsigned)rows, N = (unsigned)w, K = (unsigned)w, Kb = (unsigned)dense_basis_kpad(w);
    const unsigned tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const unsigned long long nblk = (unsigned long long)tiles_m * tiles_n;
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    const bool al = (K % 4 == 0) && aligned16(in) && aligned16(basis);
    if (precision == SSW_PRECISION_F64) {
        const double* b = static_cast<const double*>(basis);
        if (al) dct_rows_f64_kernel<true><<<(unsigned)nblk, THREADS, 0, st>>>(in, b, out, M, N, K, Kb, tiles_m, tiles_n, ep);
        else    dct_rows_f64_kernel<false><<<(unsigned)nblk, THREADS, 0, st>>>(in, b, out, M, N, K, Kb, tiles_m, tiles_n, ep);
    } else {
        const float* b = static_cast<const float*>(basis);
        if (al) dct_rows_f32_kernel<true><<<(unsigned)nblk, THREADS, 0, st>>>(in, b, out, M, N, K, Kb, tiles_m, tiles_n, ep);
        else    dct_rows_f32_kernel<false><<<(unsigned)nblk, THREADS, 0, st>>>(in, b, out, M, N, K, Kb, tiles_m, tiles_n, ep);
    }
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// Plain NT GEMM on the f32 MFMA kernel of the row pass: out[m][n] = sum_k A[m][k] * B[n][k], both
// operands f32 and k-contiguous with row stride K.  Used by the one-extraction-vs-many-marks
// similarity (README.md:62 of the reference; examples/main.rs:369-415 loops Tester::similarity).
int launch_gemm_nt_f32(hipStream_t st, const float* A, size_t M, const float* B, size_t N, size_t K, float* out) {
    if (M == 0 || N == 0 || K == 0) return SSW_OK;
    if (M > 0xFFFFFFFFull || N > 0xFFFFFFull || K > 0xFFFFFFull) return SSW_ERR_BAD_DIMS;
    const unsigned m = (unsigned)M, n = (unsigned)N, k = (unsigned)K;
    const unsigned tiles_m = (m + BM - 1) / BM, tiles_n = (n + BN - 1) / BN;
    const unsigned long long nblk = (unsigned long long)tiles_m * tiles_n;
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    const Epilogue ep{1.f, 1.f};
    // aligned instance: unpredicated 16-B loads, needs K to be a whole number of k-tiles (row stride = K)
    const bool al = (k % BK == 0) && aligned16(A) && aligned16(B);
    if (al) dct_rows_f32_kernel<true><<<(unsigned)nblk, THREADS, 0, st>>>(A, B, out, m, n, k, k, tiles_m, tiles_n, ep);
    else    dct_rows_f32_kernel<false><<<(unsigned)nblk, THREADS, 0, st>>>(A, B, out, m, n, k, k, tiles_m, tiles_n, ep);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

int launch_dct_cols(hipStream_t st, int precision, const float* in, float* out, size_t n_frames,
                    size_t w, size_t h, const void* basis, Epilogue ep) {
    if (n_frames == 0 || w == 0 || h == 0) return SSW_OK;
    if (w > 0xFFFFFFull || h > 0xFFFFFFull) return SSW_ERR_BAD_DIMS;
    const unsigned M = (unsigned)h, N = (unsigned)w, K = (unsigned)h, Kb = (unsigned)dense_basis_kpad(h);
    const unsigned tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const unsigned tiles_per_frame = tiles_m * tiles_n;
    const unsigned long long nblk = (unsigned long long)tiles_per_frame * n_frames;
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    const bool al = (K % 4 == 0) && (N % 4 == 0) && aligned16(in) && aligned16(basis);
    if (precision == SSW_PRECISION_F64) {
        const double* b = static_cast<const double*>(basis);
        if (al) dct_cols_f64_kernel<true><<<(unsigned)nblk, THREADS, 0, st>>>(b, in, out, M, N, K, Kb, tiles_m, tiles_n, tiles_per_frame, ep);
        else    dct_cols_f64_kernel<false><<<(unsigned)nblk, THREADS, 0, st>>>(b, in, out, M, N, K, Kb, tiles_m, tiles_n, tiles_per_frame, ep);
    } else {
        const float* b = static_cast<const float*>(basis);
        if (al) dct_cols_f32_kernel<true><<<(unsigned)nblk, THREADS, 0, st>>>(b, in, out, M, N, K, Kb, tiles_m, tiles_n, tiles_per_frame, ep);
        else    dct_cols_f32_kernel<false><<<(unsigned)nblk, THREADS, 0, st>>>(b, in, out, M, N, K, Kb, tiles_m, tiles_n, tiles_per_frame, ep);
    }
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

}  // namespace ssw
