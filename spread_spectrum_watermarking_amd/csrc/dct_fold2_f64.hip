// Two-level even/odd folding of the basis GEMMs in f64 (canonical precision): 3/8 of the dense
// multiply-adds.
//
// Level 1 (dct_folded*.hip) splits a length-N DCT-II into the even outputs, a product on
// s[n] = x[n] + x[N-1-n], and the odd outputs, a product on d[n] = x[n] - x[N-1-n] (n < N/2).
// The even half is itself a DCT-II of length N/2 on s, so it folds once more:
//     X[4i]   = sum_{n<N/4} (s[n] + s[N/2-1-n]) * Dee[i][n]      Dee = even half basis of N/2
//     X[4i+2] = sum_{n<N/4} (s[n] - s[N/2-1-n]) * Deo[i][n]      Deo = odd  half basis of N/2
//     X[2j+1] = sum_{n<N/2}  d[n]               * Do[j][n]       Do  = odd  half basis of N
// One k-step loads x[n], x[N-1-n], x[N/2-1-n], x[N/2+n] for 16 consecutive n < N/4; these four
// values give s[n], s[N/2-1-n] (hence ss, sd) and d[n], d[N/2-1-n], so the odd product consumes two
// k-blocks (n and its mirror N/2-1-n) per step and all three products share one loop of N/64 steps.
// All sums of inputs are formed in f64 (exact), every output is rounded once to f32: still the
// correctly rounded transform (same contract as dct_folded_f64.hip / dct.hip's f64 kernels, which
// replace the rustdct calls of /root/reference/src/dct2d.rs:129-206).
//
// Block: 256 threads = 4 waves (2 x 2), two blocks per CU; tile 128 rows x 128 output columns
// (64 odd + 32 + 32 even); k-step 8 of the N/4 axis (a 16-k step needs 56 prefetch registers on top
// of the 128 accumulators and spills; a 512-thread block with 16-k steps runs at one block per CU
// and was no faster); per wave 16 MFMA 16x16x4 f64 tiles (8 odd, 4 + 4 even) and 48 MFMAs per
// k-step.  LDS double-buffered (79.9 KB), register prefetch of the next k-step, one barrier per
// step.  Lane l: li = l & 15, lq = l >> 4 holds k = 2 lq + j at MFMA step j = 0, 1; mirrored
// operands are stored in natural order and read as pair 3-lq, element 1-j.
#include "dct_common.hpp"

namespace ssw {


__device__ inline void ld4(const double* __restrict__ p, double v[4]) {
    const f64x4 t = *reinterpret_cast<const f64x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}

// ---------------------------------------------------------------------------------------------
// Row pass, forward (DCT-II).  X: M x W, OUT: M x W, W % 16 == 0.
//   Bo  : odd half basis of W      [W/2][kp1]   (kp1 = half_basis_kpad(W))
//   Bee : even half basis of W/2   [W/4][kp2]   (kp2 = half_basis_kpad(W/2))
//   Beo : odd half basis of W/2    [W/4][kp2]
// ---------------------------------------------------------------------------------------------
constexpr int K8 = 8;
constexpr int LU8 = K8 + 4;             // 12 floats per image tile row
constexpr int LB8 = K8 + 2;             // 10 doubles per basis tile row

struct Fold2Smem8 {
    float t[4][128 * LU8];
    double bo[2][64 * LB8];
    double be[2][32 * LB8];
};

__global__ __launch_bounds__(256, 2) void dct_rows_fold2_fwd_f64_k8_kernel(
    const float* __restrict__ X, const double* __restrict__ Bo, const double* __restrict__ Bee,
    const double* __restrict__ Beo, float* __restrict__ OUT, unsigned M, unsigned W, unsigned kp1,
    unsigned kp2, unsigned tiles_m, unsigned tiles_n, Epilogue ep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Fold2Smem8* sm = reinterpret_cast<Fold2Smem8*>(smem_raw);

    const unsigned Nh = W / 2, Nq = W / 4;
    unsigned tm, tn;
    tile_of_block(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
    const unsigned m0 = tm * 128, i0 = tn * 32;
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned wm = (wave >> 1) * 64, wn = (wave & 1) * 16;
    const unsigned li = lane & 15, lq = lane >> 4;

    const unsigned srow = tid >> 1, sq = tid & 1;                          // image: row 0..127, quad 0..1
    unsigned r = m0 + srow; r = r < M ? r : M - 1;
    const float* a_row = X + (size_t)r * W;
    const unsigned bblk = tid >> 7, brow = (tid >> 1) & 63;                 // Do: block, row 0..63, quad = tid & 1
    unsigned jo = 2 * i0 + brow; jo = jo < Nh ? jo : Nh - 1;
    const double* bo_row = Bo + (size_t)jo * kp1;
    const bool has_be = tid < 128;                                          // Dee / Deo: 2 x 32 rows x 2 quads
    const unsigned ewhich = (tid >> 6) & 1, erow = (tid >> 1) & 31;
    unsigned ie = i0 + erow; ie = ie < Nq ? ie : Nq - 1;
    const double* be_row = (ewhich ? Beo : Bee) + (size_t)ie * kp2;

    f64x4 aco[4][2], ace[4], acd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        aco[i][0] = (f64x4){0, 0, 0, 0}; aco[i][1] = (f64x4){0, 0, 0, 0};
        ace[i] = (f64x4){0, 0, 0, 0}; acd[i] = (f64x4){0, 0, 0, 0};
    }

    f32x4 rt[4];
    double rbo[4], rbe[4];
    bool pad_next = false;
    auto gload = [&](unsigned t) {
        const unsigned k = t * K8 + 4 * sq;
        const unsigned kc = k + 4 <= Nq ? k : Nq - 4;
        rt[0] = *reinterpret_cast<const f32x4*>(a_row + kc);
        rt[1] = *reinterpret_cast<const f32x4*>(a_row + (W - 4 - kc));
        rt[2] = *reinterpret_cast<const f32x4*>(a_row + (Nh - 4 - kc));
        rt[3] = *reinterpret_cast<const f32x4*>(a_row + (Nh + kc));
        ld4(bo_row + (bblk ? (Nh - 4 - kc) : kc), rbo);
        if (has_be) ld4(be_row + k, rbe);
        pad_next = k >= Nq;
    };
    auto lstore = [&](unsigned buf) {
        Fold2Smem8& s = sm[buf];
        *reinterpret_cast<f32x4*>(&s.t[0][srow * LU8 + 4 * sq]) = rt[0];
        *reinterpret_cast<f32x4*>(&s.t[1][srow * LU8 + 4 - 4 * sq]) = rt[1];      // natural order of the mirror block
        *reinterpret_cast<f32x4*>(&s.t[2][srow * LU8 + 4 - 4 * sq]) = rt[2];
        *reinterpret_cast<f32x4*>(&s.t[3][srow * LU8 + 4 * sq]) = rt[3];
        double* d0 = &s.bo[bblk][brow * LB8 + (bblk ? 4 - 4 * sq : 4 * sq)];
#pragma unroll
        for (int e = 0; e < 4; ++e) d0[e] = pad_next ? 0.0 : rbo[e];
        if (has_be) {
            double* de = &s.be[ewhich][erow * LB8 + 4 * sq];
#pragma unroll
            for (int e = 0; e < 4; ++e) de[e] = rbe[e];
        }
    };

    const unsigned nk = (Nq + K8 - 1) / K8;                                // kp2 is a multiple of 16 >= this * 8
    gload(0);
    lstore(0);
    __syncthreads();
    for (unsigned t = 0; t < nk; ++t) {
        const Fold2Smem8& s = sm[t & 1];
        if (t + 1 < nk) gload(t + 1);
        f64x2 bo_lo[2], bo_hi[2], bee, beo;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bo_lo[u] = *reinterpret_cast<const f64x2*>(&s.bo[0][(2 * (wn + li) + u) * LB8 + 2 * lq]);
            bo_hi[u] = *reinterpret_cast<const f64x2*>(&s.bo[1][(2 * (wn + li) + u) * LB8 + 2 * (3 - lq)]);
        }
        bee = *reinterpret_cast<const f64x2*>(&s.be[0][(wn + li) * LB8 + 2 * lq]);
        beo = *reinterpret_cast<const f64x2*>(&s.be[1][(wn + li) * LB8 + 2 * lq]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned row = wm + 16 * i + li;
            const f32x2 t0 = *reinterpret_cast<const f32x2*>(&s.t[0][row * LU8 + 2 * lq]);
            const f32x2 t1 = *reinterpret_cast<const f32x2*>(&s.t[1][row * LU8 + 2 * (3 - lq)]);
            const f32x2 t2 = *reinterpret_cast<const f32x2*>(&s.t[2][row * LU8 + 2 * (3 - lq)]);
            const f32x2 t3 = *reinterpret_cast<const f32x2*>(&s.t[3][row * LU8 + 2 * lq]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double x0 = (double)t0[j], x1 = (double)t1[1 - j], x2 = (double)t2[1 - j], x3 = (double)t3[j];
                const double d_lo = x0 - x1, d_hi = x2 - x3;
                const double s_lo = x0 + x1, s_hi = x2 + x3;
                const double ss = s_lo + s_hi, sd = s_lo - s_hi;
                aco[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(d_lo, bo_lo[0][j], aco[i][0], 0, 0, 0);
                aco[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(d_lo, bo_lo[1][j], aco[i][1], 0, 0, 0);
                ace[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(ss, bee[j], ace[i], 0, 0, 0);
                aco[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(d_hi, bo_hi[0][1 - j], aco[i][0], 0, 0, 0);
                aco[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(d_hi, bo_hi[1][1 - j], aco[i][1], 0, 0, 0);
                acd[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(sd, beo[j], acd[i], 0, 0, 0);
            }
        }
        if (t + 1 < nk) lstore((t + 1) & 1);
        __syncthreads();
    }

    const unsigned ii = i0 + wn + li;
    if (ii < Nq) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const unsigned row = m0 + wm + 16 * i + lq + 4 * rr;
                if (row >= M) continue;
                const unsigned c = 4 * ii;
                const f32x4 v = {apply_epilogue(ep, (float)ace[i][rr], c), apply_epilogue(ep, (float)aco[i][0][rr], c + 1),
                                 apply_epilogue(ep, (float)acd[i][rr], c + 2), apply_epilogue(ep, (float)aco[i][1][rr], c + 3)};
                *reinterpret_cast<f32x4*>(OUT + (size_t)row * W + c) = v;
            }
    }
}

bool dct_rows_can_fold2(size_t w, const float* in, const float* out) {
    return w >= 64 && (w % 16 == 0) && aligned16(in) && aligned16(out);
}

int launch_dct_rows_fold2_fwd_f64(hipStream_t st, const float* in, float* out, size_t rows, size_t w,
                                  const double* bo, const double* bee, const double* beo, Epilogue ep) {
    if (rows == 0) return SSW_OK;
    if (rows > 0xFFFFFFFFull || w > 0xFFFFFFull) return SSW_ERR_BAD_DIMS;
    const unsigned M = (unsigned)rows, W = (unsigned)w, Nq = W / 4;
    const unsigned kp1 = (unsigned)half_basis_kpad(w), kp2 = (unsigned)half_basis_kpad(w / 2);
    const unsigned tiles_m = (M + 127) / 128, tiles_n = (Nq + 31) / 32;
    const unsigned long long nblk = (unsigned long long)tiles_m * tiles_n;
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    const size_t smem8 = 2 * sizeof(Fold2Smem8);
    {   // per-device function attribute (79.9 KB of dynamic LDS)
        static bool attr8[64] = {false};
        int dev = 0;
        SSW_HIP_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !attr8[dev]) {
            SSW_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dct_rows_fold2_fwd_f64_k8_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem8));
            if (dev >= 0 && dev < 64) attr8[dev] = true;
        }
    }
    dct_rows_fold2_fwd_f64_k8_kernel<<<(unsigned)nblk, 256, smem8, st>>>(in, bo, bee, beo, out, M, W, kp1, kp2, tiles_m, tiles_n, ep);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

}  // namespace ssw
