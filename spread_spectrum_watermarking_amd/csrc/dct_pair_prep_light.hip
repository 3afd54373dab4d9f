// The deep forward ROW pre-pass at level 2 (pair_prep16_rows_kernel's sixteen planes, dct_pair_prep.hip) as a kernel that
// fits BESIDE the basis GEMMs (r5).
//
// Why: a GEMM block holds 222 VGPRs per lane, two blocks per CU: 448 of a SIMD's 512 registers and 96 of the CU's 160 KB of
// LDS.  pair_prep16_rows_kernel keeps the sixteen quads of a thread in registers (246 VGPRs): beside GEMMs it can only start
// where a GEMM block has retired, so the two lanes of a batch call take turns instead of overlapping.  A streaming kernel
// that stays under 64 VGPRs and 64 KB of LDS is co-resident (tools/overlap_probe.py: a copy kernel beside the transforms of
// 128 4K frames moves 3.3 TB/s while the GEMM stages slow down 1.33 x -- 62 ms of work in 43).  This kernel is that shape:
//   phase 1  the block's tile -- LINES operand lines x 32 units e x the 16 pixel runs that meet in them -- is read as whole
//            32-pixel runs (8 lanes x 48 bytes of RGB each: 384-byte reads), converted to Y (I, Q stored for the writer) and
//            parked in LDS as f32 (16.6 KB for 8 lines);
//   phase 2  one lane = one (line, e): sixteen LDS reads, col_l2_unit (dct_pair_colops.hpp: the level-2 folds and rotations
//            of one unit, the operations and the order of pair_prep16_rows_kernel), sixteen 8-byte stores; a wave holds
//            8 consecutive e of 8 lines: every store instruction writes 512 contiguous bytes of a k-block.
// Values: bit-identical to pair_prep16_rows_kernel (same Y arithmetic -- load_yiq4 --, same f64 operations in the same order).
#include "dct_pair_split.hpp"
#include "dct_pair_colops.hpp"
#include "dct_pair_yiq_load.hpp"

#include <atomic>

namespace ssw {
namespace {

constexpr int LIGHT_PITCH = 16 * 32 + 8;         // floats per line in LDS: sixteen runs of 32 pixels (+ 8: lines on different banks)

template <int SRC /*0 plane, 1 rgb f32, 2 rgb u8, 3 rgb u16*/, bool WITH_IQ, int LINES>
__global__ __launch_bounds__(16 * LINES, 8) void pair_prep16_rows_light_kernel(
    const void* __restrict__ SRCP, DeepPlanes dp, const double* __restrict__ rot1, const double* __restrict__ rot2,
    const double* __restrict__ rot3, float* __restrict__ IP, float* __restrict__ QP, unsigned rows, unsigned W, unsigned K16,
    unsigned tiles_e, unsigned unit_h, unsigned unit_hup) {
    constexpr int NT = 16 * LINES;
    extern __shared__ __attribute__((aligned(16))) float ys[];          // LINES * LIGHT_PITCH floats (dynamic: a static size makes the compiler
                                                                        // lower the occupancy target, and with it the register cap, by its own LDS estimate)
    const unsigned N8 = W / 8, N16 = W / 16;
    const unsigned e0 = (blockIdx.x % tiles_e) * 32, line0 = (blockIdx.x / tiles_e) * LINES;
    const unsigned tid = threadIdx.x;
    // image row of an operand line (natural order: the same number; fused forward transform: lines ordered by unit of the
    // column fold, see pair_prep16_rows_kernel); pad lines hold zeros
    auto row_of = [&](unsigned line, bool& pad) -> unsigned {
        pad = false;
        if (!unit_h) return line;
        const unsigned lpf = 16 * unit_hup, z = line / lpf, rem = line - z * lpf;
        pad = (rem >> 4) >= unit_h / 16;
        return z * unit_h + (pad ? 0u : col_unit_row(rem >> 4, rem & 15u, unit_h));
    };
    // ---- phase 1: LINES x 16 runs x 8 quads of pixels -> Y in LDS
#pragma unroll 2
    for (int i = 0; i < 8; ++i) {
        const unsigned t = tid + NT * i;
        const unsigned q = t & 7u, u = (t >> 3) & 15u, ll = t >> 7;
        const unsigned line = line0 + ll;
        bool pad;
        const unsigned row = row_of(line, pad);
        // run u of unit e: pixel (u/2) N8 + e (u even) or (u/2 + 1) N8 - 1 - e (u odd) for u < 8, its mirror W - 1 - p for
        // 15 - u; over e0 .. e0 + 31 an ascending (u even) or descending (u odd) run of 32 pixels, quad q of it:
        const unsigned v = u < 8 ? u : 15 - u, hv = v >> 1;
        const bool asc = (u & 1u) == 0;
        unsigned px;
        if (u < 8) px = asc ? hv * N8 + e0 + 4 * q : (hv + 1) * N8 - 32 - e0 + 4 * q;
        else       px = asc ? W - (hv + 1) * N8 + e0 + 4 * q : W - 32 - hv * N8 - e0 + 4 * q;
        // the quad's units: e0 + 4q .. (ascending) or e0 + 28 - 4q .. (descending); beyond N16 there is nothing to read
        const unsigned efirst = asc ? e0 + 4 * q : e0 + 28 - 4 * q;
        if (line >= rows || pad || efirst >= N16) continue;
        f32x4 y, iv, qv;
        if (SRC == 0) {
            y = *reinterpret_cast<const f32x4*>(static_cast<const float*>(SRCP) + (size_t)row * W + px);
        } else {
            const void* base = SRC == 3 ? static_cast<const void*>(static_cast<const uint16_t*>(SRCP) + (size_t)row * W * 3)
                             : SRC == 2 ? static_cast<const void*>(static_cast<const uint8_t*>(SRCP) + (size_t)row * W * 3)
                                        : static_cast<const void*>(static_cast<const float*>(SRCP) + (size_t)row * W * 3);
            load_yiq4<SRC - 1, WITH_IQ>(base, px, y, iv, qv);
            if (WITH_IQ) {
                *reinterpret_cast<f32x4*>(IP + (size_t)row * W + px) = iv;
                *reinterpret_cast<f32x4*>(QP + (size_t)row * W + px) = qv;
            }
        }
        *reinterpret_cast<f32x4*>(ys + ll * LIGHT_PITCH + u * 32 + 4 * q) = y;
    }
    __syncthreads();
    // ---- phase 2: LINES x 32 units, two per lane; a wave = 8 consecutive e of 8 lines
    void* const planes[16] = {dp.asp, dp.asm_, dp.bdp, dp.bdm, dp.oap, dp.obp, dp.oam, dp.obm,
                              dp.r1p, dp.r1m, dp.r2a, dp.r2b, dp.as2, dp.bd2, dp.ad2, dp.bs2};
    const unsigned kk = tid & 7u, ll = (tid >> 3) & (unsigned)(LINES - 1), eb = tid / (8u * LINES);
    const unsigned line = line0 + ll;
    if (line >= rows) return;
    bool pad;
    (void)row_of(line, pad);
    const float* yl = ys + ll * LIGHT_PITCH;
#pragma unroll 1
    for (int j = 0; j < 2; ++j) {
        const unsigned el = 8 * (eb + 2 * j) + kk, e = e0 + el;
        if (e >= K16) continue;
        const unsigned at = (unsigned)blk_index<double>(line, e, rows) * 8u;      // byte offset inside a plane: < 4 GB (the launcher checks)
        if (e >= N16 || pad) {                                    // padding of the planes; lines of the padding units
#pragma unroll
            for (int a = 0; a < 16; ++a) *reinterpret_cast<double*>(static_cast<char*>(planes[a]) + at) = 0.0;
            continue;
        }
        // col_l2_unit (dct_pair_colops.hpp) in two halves -- the differences' side, then the sums' side, each reading its
        // sixteen values from LDS again -- so that the lane stays under 64 registers; same operations, same order
        auto X = [&](int u) { return (double)yl[u * 32 + ((u & 1) ? 31 - el : el)]; };
        auto put = [&](int a, double v) { *reinterpret_cast<double*>(static_cast<char*>(planes[a]) + at) = v; };
        const unsigned Nq = W / 4;
        {
            double as, bd, ad, bs, asm_, bdm, adm, bsm;
            split_one_r(X(0) - X(15), X(3) - X(12), X(4) - X(11), X(7) - X(8), rot_load(rot1, e, Nq), as, bd, ad, bs);                  // unit e
            __builtin_amdgcn_sched_barrier(0);
            split_one_r(X(1) - X(14), X(2) - X(13), X(5) - X(10), X(6) - X(9), rot_load(rot1, N8 - 1 - e, Nq), asm_, bdm, adm, bsm);    // unit W/8 - 1 - e
            put(0, as + asm_); put(1, as - asm_); put(2, bd + bdm); put(3, bd - bdm);
            __builtin_amdgcn_sched_barrier(0);
            const double c3 = rot3[e], s3 = rot3[N16 + e];
            const double au = ad * c3 + adm * s3, bu = adm * c3 - ad * s3;
            const double av = bsm * c3 + bs * s3, bv = bs * c3 - bsm * s3;
            put(4, au + av); put(5, bu + bv); put(6, au - av); put(7, bu - bv);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            double S[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) S[u] = X(u) + X(15 - u);
            const double ss0 = S[0] + S[7], ss3 = S[3] + S[4], ss1 = S[1] + S[6], ss2 = S[2] + S[5];
            const double r1 = ss0 + ss3, r2 = ss0 - ss3, r1m = ss1 + ss2, r2m = ss1 - ss2;
            put(8, r1 + r1m); put(9, r1 - r1m);
            const double c3 = rot3[e], s3 = rot3[N16 + e];
            put(10, r2 * c3 + r2m * s3); put(11, r2m * c3 - r2 * s3);
            __builtin_amdgcn_sched_barrier(0);
            double SD[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) SD[u] = S[u] - S[7 - u];
            double o12, o13, o14, o15;
            split_one_r(SD[0], SD[1], SD[2], SD[3], rot_load(rot2, e, N8), o12, o13, o14, o15);
            put(12, o12); put(13, o13); put(14, o14); put(15, o15);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The INVERSE row pre-pass at level 2 (prep16_inv_rows_l2_kernel's sixteen planes, dct_pair_prep_staged.hip) in the same style:
// a block parks LINES whole coefficient rows in LDS (contiguous 16-byte loads, all of a lane's in flight together), then one
// lane per (line, unit k) gathers the unit's sixteen coefficients -- c[2k+1], c[W/2-1-2k], ... c[16k], c[16k+8], c[8k+4] ...:
// strides of 2 .. 16 floats, spread over the banks by 4 floats of padding per 32 -- and runs inv_col_l2_unit_lo / _hi
// (dct_pair_colops.hpp: the operations and the order of the staged kernel).  A wave = 8 consecutive units of 2 lines of 4
// k-blocks: every store instruction writes four 128-byte lines whole.
// ---------------------------------------------------------------------------------------------
constexpr int ILINES = 2;
__device__ inline unsigned ipos(unsigned j) { return j + 4u * (j >> 5); }

__global__ __launch_bounds__(128) void prep16_inv_rows_light_kernel(const float* __restrict__ X, double* __restrict__ base,
                                                                    const double* __restrict__ rot1, const double* __restrict__ rot2,
                                                                    const double* __restrict__ rot3, unsigned rows, unsigned W,
                                                                    unsigned K16, unsigned unit_h, unsigned unit_hup) {
    extern __shared__ __attribute__((aligned(16))) float ls[];          // ILINES x pitch floats
    const unsigned pitch = W + W / 8;
    const unsigned tid = threadIdx.x;
    const unsigned line0 = blockIdx.x * ILINES;
    const unsigned N16 = W / 16, Wq4 = W / 4;
    // ---- phase 1: the lines' coefficient rows (unit order of the fused inverse transform: see prep16_inv_rows_l2_kernel)
    for (unsigned q0 = 0; q0 < ILINES * Wq4; q0 += 128 * 8) {
        f32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned q = q0 + tid + 128 * i;
            v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (q >= ILINES * Wq4) continue;
            const unsigned ll = q / Wq4, j4 = q - ll * Wq4, line = line0 + ll;
            if (line >= rows) continue;
            size_t src_row = line;
            bool ok = true;
            if (unit_h) {
                const unsigned lpf = 16 * unit_hup, z = line / lpf, rem = line - z * lpf;
                ok = (rem >> 4) < unit_h / 16;
                src_row = (size_t)z * unit_h + (ok ? inv_col_unit_row(rem >> 4, rem & 15u, unit_h) : 0u);
            }
            if (ok) v[i] = *reinterpret_cast<const f32x4*>(X + src_row * W + 4 * j4);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned q = q0 + tid + 128 * i;
            if (q >= ILINES * Wq4) continue;
            const unsigned ll = q / Wq4, j4 = q - ll * Wq4;
            *reinterpret_cast<f32x4*>(ls + ll * pitch + ipos(4 * j4)) = v[i];
        }
    }
    __syncthreads();
    // ---- phase 2: (line, unit) items; item = 16 kb + 8 ll + kk: unit k = 8 kb + kk of line ll
    const unsigned Wh = W / 2, N8 = W / 8, Nq = W / 4;
    for (unsigned it = tid; it < ILINES * K16; it += 128) {
        const unsigned kk = it & 7u, ll = (it >> 3) & (unsigned)(ILINES - 1), kb = it >> 4, k = 8 * kb + kk;
        const unsigned line = line0 + ll;
        if (line >= rows || k >= K16) continue;
        const unsigned at = (unsigned)(((size_t)kb * rows + line) * 8 + kk) * 8u;        // byte offset inside a plane (< 4 GB: the launcher checks)
        const size_t pstride = (size_t)rows * K16 * sizeof(double);
        auto put = [&](int a, double v) { *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + (size_t)a * pstride + at) = v; };
        if (k >= N16) {
#pragma unroll
            for (int a = 0; a < 16; ++a) put(a, 0.0);
            continue;
        }
        const float* c = ls + ll * pitch;
        const unsigned km = N8 - 1 - k;
        auto C = [&](unsigned j) { return c[ipos(j)]; };
        {
            const float x[8] = {C(2 * k + 1), C(Wh - 1 - 2 * k), C(Wh + 2 * k + 1), C(W - 1 - 2 * k),
                                C(2 * km + 1), C(Wh - 1 - 2 * km), C(Wh + 2 * km + 1), C(W - 1 - 2 * km)};
            ColL2Tab t;
            t.ra = rot_load(rot1, k, Nq); t.rb = rot_load(rot1, km, Nq);
            t.c3 = rot3[k]; t.s3 = rot3[N16 + k];
            double o[8];
            inv_col_l2_unit_lo(x, t, o);
#pragma unroll
            for (int a = 0; a < 8; ++a) put(a, o[a]);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const float x[8] = {C(16 * k), C(16 * k + 8), C(8 * k + 4), C(8 * km + 4),
                                C(4 * k + 2), C(Wh - 2 - 4 * k), C(Wh + 4 * k + 2), C(W - 2 - 4 * k)};
            ColL2Tab t;
            t.rc = rot_load(rot2, k, N8);
            t.c3 = rot3[k]; t.s3 = rot3[N16 + k];
            double o[8];
            inv_col_l2_unit_hi(x, t, o);
#pragma unroll
            for (int a = 0; a < 8; ++a) put(8 + a, o[a]);
        }
    }
}

}  // namespace

bool dct_pair_prep_light_ok(size_t w, size_t lines) {
    return tuning(TUNE_PREP_LIGHT) != 0 && w % 64 == 0 && dct_pair_efold(w) && lines * dct_pair_split_kpad(w / 2) * sizeof(double) <= 0xFFFFFFFFull;
}

int launch_dct_pair_prep16_rows_light(hipStream_t st, int src_kind, const void* src, const DeepPlanes& dp, const double* rot1,
                                      const double* rot2, const double* rot3, float* ip, float* qp, size_t rows, size_t w, unsigned K16,
                                      unsigned unit_h, unsigned unit_hup) {
    const int LINES = tuning(TUNE_PREP_LIGHT) == 2 ? 16 : 8;         // (2: tiles of 16 lines, an A/B switch)
    const unsigned tiles_e = (K16 + 31) / 32;
    const unsigned long long nblk = (unsigned long long)((rows + LINES - 1) / LINES) * tiles_e;
    if (rows > 0xFFFFFFFFull || nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    const bool iq = ip && qp;
#define SSW_LIGHT1(SRCV, IQV, LN) pair_prep16_rows_light_kernel<SRCV, IQV, LN><<<(unsigned)nblk, 16 * LN, LN * LIGHT_PITCH * sizeof(float), st>>>( \
        src, dp, rot1, rot2, rot3, ip, qp, (unsigned)rows, (unsigned)w, K16, tiles_e, unit_h, unit_hup)
#define SSW_LIGHT(SRCV, IQV) do { if (LINES == 16) SSW_LIGHT1(SRCV, IQV, 16); else SSW_LIGHT1(SRCV, IQV, 8); } while (0)
    if (src_kind == 0) SSW_LIGHT(0, false);
    else if (src_kind == 1) { if (iq) SSW_LIGHT(1, true); else SSW_LIGHT(1, false); }
    else if (src_kind == 2) { if (iq) SSW_LIGHT(2, true); else SSW_LIGHT(2, false); }
    else                    { if (iq) SSW_LIGHT(3, true); else SSW_LIGHT(3, false); }
#undef SSW_LIGHT
#undef SSW_LIGHT1
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

bool dct_pair_inv_prep_light_ok(size_t w, size_t lines) {
    return tuning(TUNE_INV_PREP_LIGHT) != 0 && w % 128 == 0 && dct_pair_efold_inv(w) &&
           lines * dct_pair_split_kpad(w / 2) * sizeof(double) <= 0xFFFFFFFFull;
}

int launch_prep16_inv_rows_light(hipStream_t st, const float* in, size_t rows, size_t w, double* base, const double* rot1,
                                 const double* rot2, const double* rot3, unsigned K16, unsigned unit_h, unsigned unit_hup) {
    const size_t smem = (size_t)ILINES * (w + w / 8) * sizeof(float);
    if (smem > 160 * 1024 / 2) return SSW_ERR_BAD_DIMS;
    {
        static std::atomic<bool> attr_set[64];
        int dev = 0;
        SSW_HIP_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
            SSW_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(prep16_inv_rows_light_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
            if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
        }
    }
    const unsigned long long nblk = (rows + ILINES - 1) / ILINES;
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    prep16_inv_rows_light_kernel<<<(unsigned)nblk, 128, smem, st>>>(in, base, rot1, rot2, rot3, (unsigned)rows, (unsigned)w, K16, unit_h, unit_hup);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

}  // namespace ssw
