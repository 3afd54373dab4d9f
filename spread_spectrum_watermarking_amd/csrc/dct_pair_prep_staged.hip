// Deep pre-passes, r4: the three HBM-bound pre-passes that r3 left at 31-43 % of HBM -- forward columns, inverse rows,
// inverse columns (dct_pair_prep.hip: pair_prep16_cols_kernel, pair_prep16_inv_rows_kernel, pair_prep16_inv_cols_kernel)
// -- with every result staged through LDS in the layout of the k-blocked operand planes, so that
//   * a global read is a run of >= 512 contiguous bytes per row (column passes: one class-major tile of 128 memory
//     columns, dct_pair_common.hpp) or >= 512 bytes per line (row pass: 8 neighbouring 64-byte regions), and
//   * a global store instruction writes 16 bytes per lane of ONE contiguous run: the 64-byte k-block pieces of 128
//     (column passes) or 32 (row pass) consecutive operand lines = 8 KB / 2 KB.
// (r3: 32-byte pieces of lines 8 or 16 apart, 8-byte stores for the mirrored halves at 4K, four 32-byte read runs per
// row: PMC traffic 1.4-2.1 x the algorithmic bytes.)  Same operations in the same order per operand element as the r3
// kernels -- the operand planes are bit-identical (tools/prep_check.py), so is every result downstream.
//
// Reference: src/dct2d.rs:172-206 is the column pass whose strided gather / scatter these kernels replace; the f32
// store between the two passes (:152-168) stays where it was (the input of the column kernels IS that f32 plane).
#include "dct_pair_split.hpp"
#include "dct_pair_colops.hpp"

#include <cstdlib>

namespace ssw {

namespace {

constexpr unsigned CT = 128;                  // operand lines of a column-pass block = the class-major tile
constexpr unsigned SLABD = CT * 8;            // doubles of one LDS slab: CT lines x one 64-byte k-block piece

// Slab addressing: double j of line l sits at l * 8 + (j ^ sw(l)).  sw() spreads the lines that the 16 lanes of a
// ds_write_b64 group touch (4 columns apart in memory = 4, 8 or 16 lines apart) over the LDS banks.
// MODE 0: natural column order, 1: forward class-major tile, 2: inverse class-major tile.
template <int MODE>
__device__ inline unsigned slab_sw(unsigned l) {
    if (MODE == 1) return ((2u * ((l >> 5) & 3u)) | ((l >> 2) & 1u)) ^ (2u * ((l >> 3) & 1u));
    if (MODE == 2) return (l >> 4) & 7u;
    return (l >> 2) & 7u;
}
// operand line (inside the tile) of memory column m of the tile
template <int MODE>
__device__ inline unsigned tile_line(unsigned m, bool level2) {
    if (MODE == 1) return ForwardClassLayout{CT, CT, level2}.natural(m);
    if (MODE == 2) return inverse_class_natural(m, CT, CT, level2);
    return m;
}
// memory quad (4 columns) of lane l32 of a half-wave.  Forward tile: the quads of the classes with odd lines (EP EM OP
// OM) are interleaved with the even ones, so that a 16-lane write group holds 8 even and 8 odd lines.
template <int MODE>
__device__ inline unsigned lane_quad(unsigned l32) {
    return MODE == 1 ? ((l32 & 7u) | ((l32 & 8u) << 1) | ((l32 & 16u) >> 1)) : l32;
}

// Entries j (bit j of `mask`, block-uniform) of all lines of a slab -> plane[k0 + j] of the operand lines
// line_base + l, l < nl.  Consecutive lanes store consecutive 16-byte chunks: 16 lines x 64 bytes per wave instruction
// when the piece is whole.  An odd k0 (semi-deep: H/8 odd) stores single doubles.
template <int MODE>
__device__ inline void slab_store(const double* __restrict__ slab, double* __restrict__ plane, size_t lines_total, size_t line_base,
                                  unsigned nl, unsigned k0, unsigned mask, unsigned tid) {
    if (mask == 0) return;
    if ((k0 & 1u) == 0) {
#pragma unroll
        for (unsigned it = 0; it < CT * 4 / 256; ++it) {
            const unsigned w = tid + 256 * it, l = w >> 2, c = w & 3u;
            const unsigned m2 = (mask >> (2 * c)) & 3u;
            if (l >= nl || m2 == 0) continue;
            const unsigned sw = slab_sw<MODE>(l);
            f64x2 v = *reinterpret_cast<const f64x2*>(slab + l * 8 + 2 * (c ^ (sw >> 1)));
            if (sw & 1u) v = (f64x2){v[1], v[0]};
            const unsigned k = k0 + 2 * c;
            double* o = plane + ((size_t)(k >> 3) * lines_total + line_base + l) * 8 + (k & 7u);
            if (m2 == 3u) *reinterpret_cast<f64x2*>(o) = v;
            else if (m2 == 1u) o[0] = v[0];
            else o[1] = v[1];
        }
    } else {
#pragma unroll
        for (unsigned it = 0; it < CT * 8 / 256; ++it) {
            const unsigned w = tid + 256 * it, l = w >> 3, j = w & 7u;
            if (l >= nl || !((mask >> j) & 1u)) continue;
            const unsigned k = k0 + j;
            plane[((size_t)(k >> 3) * lines_total + line_base + l) * 8 + (k & 7u)] = slab[l * 8 + (j ^ slab_sw<MODE>(l))];
        }
    }
}

// zeros for plane[k] of the lines line_base + l (l < nl), k in [ka, kb)
__device__ inline void zero_range(double* __restrict__ plane, size_t lines_total, size_t line_base, unsigned nl, unsigned ka, unsigned kb, unsigned tid) {
    const unsigned nk = kb > ka ? kb - ka : 0;
    for (unsigned w = tid; w < nl * nk; w += 256) {
        const unsigned l = w / nk, k = ka + w % nk;
        plane[((size_t)(k >> 3) * lines_total + line_base + l) * 8 + (k & 7u)] = 0.0;
    }
}

// (Rot4, rot_load, split_one_r: dct_pair_colops.hpp -- shared with the GEMMs' fused epilogues)

// XCD-aware block order: blocks b, b + 8, ... run on one XCD; give each XCD a contiguous run of work ids, so that the
// blocks that complete each other's partly written 128-byte lines (neighbouring unit groups) share an L2
__device__ inline unsigned xcd_contiguous_id(unsigned bid, unsigned nblk) {
    const unsigned q = nblk / 8, r = nblk % 8, xcd = bid % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
}

// ---------------------------------------------------------------------------------------------
// Forward column pass (H % 16 == 0; SPLIT_SD = false: "semi-deep", H % 8 == 0 only -- see pair_prep16_cols_kernel).
// Block = one class-major tile of CT memory columns x one group of 8 units e = 8 g + u; thread = (unit u, memory quad):
// the 16 rows that meet in e (row v mirrors row 15 - v), 4 columns each.  Three LDS rounds of up to six slabs:
//   A  AS BD AD BS R1 R2 at unit e                       -> k-block g, whole 64-byte pieces
//   B  the same planes at the mirror units H/8 - 1 - e' of e' = 8 g - s + u, s = (8 - H/8 mod 8) mod 8: the shift makes
//      the mirrored range a whole k-block too (4K: H/8 = 270, s = 2; measured with the unshifted range, whose stores
//      are 16 + 48 bytes of two pieces: 3.42 ms per 128 frames against 2.92 with whole pieces).  These outputs need
//      rows 1, 2, 5, 6 (and their mirrors) of e' only: for s != 0 the thread loads those 8 rows again -- 6 of the 8
//      units are this block's own (cache hits), the other s the previous group's (same XCD: xcd_contiguous_id)
//   C  AS2 BD2 AD2 BS2 at e (semi-deep: SD at e, H/8-1-e, H/8+e, H/4-1-e)
// ---------------------------------------------------------------------------------------------
template <int MODE, bool SPLIT_SD>
__global__ __launch_bounds__(256) void prep16_cols_staged_kernel(const float* __restrict__ IN, DeepPlanes dp,
                                                                 const double* __restrict__ rot1, const double* __restrict__ rot2,
                                                                 unsigned W, unsigned H, unsigned K8, unsigned K16,
                                                                 unsigned n_frames, unsigned groups, unsigned tiles_c, unsigned nwork, unsigned efold) {
    __shared__ __attribute__((aligned(16))) double lds[6 * SLABD];
    const unsigned id = xcd_contiguous_id(blockIdx.x, nwork);
    const unsigned g = id % groups, zt = id / groups, ct = zt % tiles_c, z = zt / tiles_c;
    const unsigned Hh = H / 2, Hq = H / 4, H8 = H / 8, HU = SPLIT_SD ? H / 16 : (H / 8 + 1) / 2;      // HU: units
    const unsigned tid = threadIdx.x, u = tid >> 5;
    const unsigned mq = lane_quad<MODE>(tid & 31u);
    const unsigned col0 = ct * CT;
    const unsigned nl = W - col0 < CT ? W - col0 : CT;              // lines of this tile (MODE 0 only: the last tile may be short)
    unsigned off[4], sws[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned l = tile_line<MODE>(4 * mq + i, efold != 0);
        sws[i] = slab_sw<MODE>(l);
        off[i] = l * 8;
    }
    const unsigned e = 8 * g + u;
    const bool unit_ok = e < HU;
    const unsigned ec = unit_ok ? e : 0;                            // table indices stay in range
    const unsigned sh = (8u - (H8 & 7u)) & 7u;                      // round B's units: e' = e - sh
    const bool unitb_ok = e >= sh && e - sh < HU;
    const unsigned eb = unitb_ok ? e - sh : 0;
    f32x4 x[16], xb[8];
    const Rot4 ra = rot_load(rot1, ec, Hq), rb = rot_load(rot1, H8 - 1 - eb, Hq);
    const Rot4 rc = SPLIT_SD ? rot_load(rot2, ec, H8) : Rot4{0, 0, 0, 0};
    {
        unsigned c = col0 + 4 * mq;
        c = c + 4 <= W ? c : W - 4;                                // W % 4 == 0; duplicates are never stored (l >= nl)
        const float* __restrict__ Pz = IN + (size_t)z * H * W + c;
#pragma unroll
        for (int v = 0; v < 16; ++v) x[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (unit_ok) {
            const unsigned rp[8] = {e, H8 - 1 - e, H8 + e, Hq - 1 - e, Hq + e, 3 * H8 - 1 - e, 3 * H8 + e, Hh - 1 - e};
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                x[v] = *reinterpret_cast<const f32x4*>(Pz + (size_t)rp[v] * W);
                x[15 - v] = *reinterpret_cast<const f32x4*>(Pz + (size_t)(H - 1 - rp[v]) * W);
            }
        }
        // rows 1, 2, 5, 6 of unit e' and their mirrors (rows 14, 13, 10, 9)
        if (sh == 0) {
            xb[0] = x[1]; xb[1] = x[2]; xb[2] = x[5]; xb[3] = x[6]; xb[4] = x[14]; xb[5] = x[13]; xb[6] = x[10]; xb[7] = x[9];
        } else {
#pragma unroll
            for (int v = 0; v < 8; ++v) xb[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (unitb_ok) {
                const unsigned rq[4] = {H8 - 1 - eb, H8 + eb, 3 * H8 - 1 - eb, 3 * H8 + eb};
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    xb[v] = *reinterpret_cast<const f32x4*>(Pz + (size_t)rq[v] * W);
                    xb[4 + v] = *reinterpret_cast<const f32x4*>(Pz + (size_t)(H - 1 - rq[v]) * W);
                }
            }
        }
    }
    double* P8[6] = {static_cast<double*>(dp.as), static_cast<double*>(dp.bd), static_cast<double*>(dp.ad), static_cast<double*>(dp.bs),
                     static_cast<double*>(dp.r1), static_cast<double*>(dp.r2)};
    double* P16[4] = {static_cast<double*>(dp.as2), static_cast<double*>(dp.bd2), static_cast<double*>(dp.ad2), static_cast<double*>(dp.bs2)};
    const size_t lines = (size_t)n_frames * W, line_base = (size_t)z * W + col0;
    const unsigned nv = 8 * g >= HU ? 0u : (HU - 8 * g < 8 ? HU - 8 * g : 8u);       // valid units of the group
    const unsigned mlo = (1u << nv) - 1u, mhi = (0xFFu << (8 - nv)) & 0xFFu;           // entries u / 7 - u of the valid units
    // round B: entries 7 - u of the units e' < HU; e' < 0 are k >= H/8: the planes' zero padding (K8 >= the whole piece)
    const unsigned nvb = 8 * g >= HU + sh ? 0u : (HU + sh - 8 * g < 8 ? HU + sh - 8 * g : 8u);
    const unsigned mhib = (0xFFu << (8 - nvb)) & 0xFFu;
    const unsigned um = 7 - u;

    // ---- round A: unit e
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double D[8], S[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            S[v] = (double)x[v][i] + (double)x[15 - v][i];
            D[v] = (double)x[v][i] - (double)x[15 - v][i];
        }
        double o[6] = {0, 0, 0, 0, 0, 0};
        split_one_r(D[0], D[3], D[4], D[7], ra, o[0], o[1], o[2], o[3]);
        const double ss0 = S[0] + S[7], ss3 = S[3] + S[4];
        o[4] = ss0 + ss3;
        o[5] = ss0 - ss3;
#pragma unroll
        for (int a = 0; a < 6; ++a) lds[a * SLABD + off[i] + (u ^ sws[i])] = unit_ok ? o[a] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 6; ++a) slab_store<MODE>(lds + a * SLABD, P8[a], lines, line_base, nl, 8 * g, mlo, tid);
    __syncthreads();
    // ---- round B: the mirror units H/8 - 1 - e' (a whole k-block: H/8 + sh is a multiple of 8)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double S1 = (double)xb[0][i] + (double)xb[4][i], D1 = (double)xb[0][i] - (double)xb[4][i];
        const double S2 = (double)xb[1][i] + (double)xb[5][i], D2 = (double)xb[1][i] - (double)xb[5][i];
        const double S5 = (double)xb[2][i] + (double)xb[6][i], D5 = (double)xb[2][i] - (double)xb[6][i];
        const double S6 = (double)xb[3][i] + (double)xb[7][i], D6 = (double)xb[3][i] - (double)xb[7][i];
        double o[6] = {0, 0, 0, 0, 0, 0};
        split_one_r(D1, D2, D5, D6, rb, o[0], o[1], o[2], o[3]);
        const double ss1 = S1 + S6, ss2 = S2 + S5;
        o[4] = ss1 + ss2;
        o[5] = ss1 - ss2;
#pragma unroll
        for (int a = 0; a < 6; ++a) lds[a * SLABD + off[i] + (um ^ sws[i])] = unitb_ok ? o[a] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 6; ++a) slab_store<MODE>(lds + a * SLABD, P8[a], lines, line_base, nl, H8 + sh - 8 * g - 8, mhib, tid);
    __syncthreads();
    // ---- round C: the level-2 difference SD
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double S[8], SD[4];
#pragma unroll
        for (int v = 0; v < 8; ++v) S[v] = (double)x[v][i] + (double)x[15 - v][i];
#pragma unroll
        for (int v = 0; v < 4; ++v) SD[v] = S[v] - S[7 - v];
        double o[4] = {0, 0, 0, 0};
        if (SPLIT_SD) split_one_r(SD[0], SD[1], SD[2], SD[3], rc, o[0], o[1], o[2], o[3]);
        else { o[0] = SD[0]; o[1] = SD[1]; o[2] = SD[2]; o[3] = SD[3]; }      // SD at e, H/8-1-e, H/8+e, H/4-1-e
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const unsigned ent = (SPLIT_SD || !(a & 1)) ? u : um;
            lds[a * SLABD + off[i] + (ent ^ sws[i])] = unit_ok ? o[a] : 0.0;
        }
    }
    __syncthreads();
    if (SPLIT_SD) {
        // zeros beyond H/16 up to the end of the group (the planes are K16 >= 8 groups wide)
        const unsigned mk = 8 * g + 8 <= K16 ? 0xFFu : (8 * g >= K16 ? 0u : (1u << (K16 - 8 * g)) - 1u);
#pragma unroll
        for (int a = 0; a < 4; ++a) slab_store<MODE>(lds + a * SLABD, P16[a], lines, line_base, nl, 8 * g, mk, tid);
    } else {
        double* M = P16[0];
        slab_store<MODE>(lds + 0 * SLABD, M, lines, line_base, nl, 8 * g, mlo, tid);
        slab_store<MODE>(lds + 1 * SLABD, M, lines, line_base, nl, H8 - 8 * g - 8, mhi, tid);
        slab_store<MODE>(lds + 2 * SLABD, M, lines, line_base, nl, H8 + 8 * g, mlo, tid);
        slab_store<MODE>(lds + 3 * SLABD, M, lines, line_base, nl, Hq - 8 * g - 8, mhi, tid);
    }
    if (g == 0) {                                                   // padding of the planes up to their k-padded widths
#pragma unroll
        for (int a = 0; a < 6; ++a) zero_range(P8[a], lines, line_base, nl, H8, K8, tid);
        if (SPLIT_SD) {
#pragma unroll
            for (int a = 0; a < 4; ++a) zero_range(P16[a], lines, line_base, nl, 8 * groups, K16, tid);
        } else {
            zero_range(P16[0], lines, line_base, nl, Hq, K16, tid);                    // K16: kpad(H/2) here
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Forward column pass at LEVEL 2 (r4c; H % 16 == 0, dct_pair_efold_cols): the thread of unit e holds the 16 rows of unit e
// AND of its mirror unit H/8 - 1 - e, so the extra fold / rotation of pair_prep16_rows_kernel's level 2 happens in
// registers and the mirrored round B disappears.  Sixteen planes K16 wide, numbered like the row pass's:
//   round A (exact)     0 .. 3 AS+ AS- BD+ BD-,  8 9 R1+ R1-
//   round B (rotated)   4 .. 7 (a, b) of AD plus / minus (a, b) of the reversed BS,  10 11 (a, b) of R2     [table of H/4]
//   round C             12 .. 15 AS2 BD2 AD2 BS2
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void prep16_cols_l2_kernel(const float* __restrict__ IN, double* __restrict__ base,
                                                             const double* __restrict__ rot1, const double* __restrict__ rot2,
                                                             const double* __restrict__ rot3,
                                                             unsigned W, unsigned H, unsigned K16,
                                                             unsigned n_frames, unsigned groups, unsigned tiles_c, unsigned nwork, unsigned in_l2) {
    __shared__ __attribute__((aligned(16))) double lds[6 * SLABD];
    const unsigned id = xcd_contiguous_id(blockIdx.x, nwork);
    const unsigned g = id % groups, zt = id / groups, ct = zt % tiles_c, z = zt / tiles_c;
    const unsigned Hh = H / 2, Hq = H / 4, H8 = H / 8, HU = H / 16;
    const unsigned tid = threadIdx.x, u = tid >> 5;
    const unsigned mq = lane_quad<MODE>(tid & 31u);
    const unsigned col0 = ct * CT;
    const unsigned nl = W - col0 < CT ? W - col0 : CT;
    unsigned off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned l = tile_line<MODE>(4 * mq + i, in_l2 != 0);
        off[i] = l * 8 + (u ^ slab_sw<MODE>(l));
    }
    const unsigned e = 8 * g + u;
    const bool unit_ok = e < HU;
    const unsigned ec = unit_ok ? e : 0;
    f32x4 x[16];
    const Rot4 ra = rot_load(rot1, ec, Hq), rb = rot_load(rot1, H8 - 1 - ec, Hq), rc = rot_load(rot2, ec, H8);
    const double c3 = rot3[ec], s3 = rot3[HU + ec];
    {
        unsigned c = col0 + 4 * mq;
        c = c + 4 <= W ? c : W - 4;
        const float* __restrict__ Pz = IN + (size_t)z * H * W + c;
#pragma unroll
        for (int v = 0; v < 16; ++v) x[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (unit_ok) {
            const unsigned rp[8] = {e, H8 - 1 - e, H8 + e, Hq - 1 - e, Hq + e, 3 * H8 - 1 - e, 3 * H8 + e, Hh - 1 - e};
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                x[v] = *reinterpret_cast<const f32x4*>(Pz + (size_t)rp[v] * W);
                x[15 - v] = *reinterpret_cast<const f32x4*>(Pz + (size_t)(H - 1 - rp[v]) * W);
            }
        }
    }
    auto plane = [&](unsigned a) { return base + (size_t)a * n_frames * W * K16; };
    const size_t lines = (size_t)n_frames * W, line_base = (size_t)z * W + col0;
    // entries of the group inside the planes (zeros between H/16 and K16 come from the units beyond the axis)
    const unsigned mk = 8 * g + 8 <= K16 ? 0xFFu : (8 * g >= K16 ? 0u : (1u << (K16 - 8 * g)) - 1u);
    // ---- rounds A, B
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double D[8], S[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                S[v] = (double)x[v][i] + (double)x[15 - v][i];
                D[v] = (double)x[v][i] - (double)x[15 - v][i];
            }
            double as, bd, ad, bs, asm_, bdm, adm, bsm;
            split_one_r(D[0], D[3], D[4], D[7], ra, as, bd, ad, bs);            // unit e
            split_one_r(D[1], D[2], D[5], D[6], rb, asm_, bdm, adm, bsm);       // unit H/8 - 1 - e
            const double ss0 = S[0] + S[7], ss3 = S[3] + S[4], ss1 = S[1] + S[6], ss2 = S[2] + S[5];
            const double r1 = ss0 + ss3, r2 = ss0 - ss3, r1m = ss1 + ss2, r2m = ss1 - ss2;
            double o[6];
            if (rnd == 0) {
                o[0] = as + asm_; o[1] = as - asm_; o[2] = bd + bdm; o[3] = bd - bdm; o[4] = r1 + r1m; o[5] = r1 - r1m;
            } else {
                const double au = ad * c3 + adm * s3, bu = adm * c3 - ad * s3;
                const double av = bsm * c3 + bs * s3, bv = bs * c3 - bsm * s3;
                o[0] = au + av; o[1] = bu + bv; o[2] = au - av; o[3] = bu - bv;
                o[4] = r2 * c3 + r2m * s3; o[5] = r2m * c3 - r2 * s3;
            }
#pragma unroll
            for (int a = 0; a < 6; ++a) lds[a * SLABD + off[i]] = unit_ok ? o[a] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 6; ++a)
            slab_store<MODE>(lds + a * SLABD, plane(a < 4 ? 4 * rnd + a : 8 + 2 * rnd + (a - 4)), lines, line_base, nl, 8 * g, mk, tid);
        __syncthreads();
    }
    // ---- round C: the level-2 difference SD
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double S[8], SD[4];
#pragma unroll
        for (int v = 0; v < 8; ++v) S[v] = (double)x[v][i] + (double)x[15 - v][i];
#pragma unroll
        for (int v = 0; v < 4; ++v) SD[v] = S[v] - S[7 - v];
        double o[4] = {0, 0, 0, 0};
        split_one_r(SD[0], SD[1], SD[2], SD[3], rc, o[0], o[1], o[2], o[3]);
#pragma unroll
        for (int a = 0; a < 4; ++a) lds[a * SLABD + off[i]] = unit_ok ? o[a] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a) slab_store<MODE>(lds + a * SLABD, plane(12 + a), lines, line_base, nl, 8 * g, mk, tid);
    if (g == 0)
#pragma unroll
        for (int a = 0; a < 16; ++a) zero_range(plane(a), lines, line_base, nl, 8 * groups, K16, tid);
}

// ---------------------------------------------------------------------------------------------
// Inverse column pass.  Every operand element of the inverse depends on its own four (or one) coefficient rows only:
//   AS BD AD BS at unit k < H/8:    rows 2k+1, H/2-1-2k, H/2+2k+1, H-1-2k        (rotation table of H, half length H/4)
//   R1[k] = row 8k, R2[k] = row 8k+4
//   AS2 .. BS2 at unit e < H/16:    rows 4e+2, H/2-2-4e, H/2+4e+2, H-2-4e        (table of H/2)
//   semi-deep (SPLIT_MID = false):  M[q] = row 4q+2, q < H/4
// so a block takes one k-block (8 units) of one plane group for one tile of CT memory columns and every store is a whole
// 64-byte piece: task kb < K8/8: AS BD AD BS R1 R2 (six slabs); then kb < K16/8: AS2 BD2 AD2 BS2 (semi-deep: three
// k-blocks of M per block).  Units beyond the axis store the planes' zero padding.
// ---------------------------------------------------------------------------------------------
template <int MODE, bool SPLIT_MID>
__global__ __launch_bounds__(256) void prep16_inv_cols_staged_kernel(const float* __restrict__ IN, DeepPlanes dp,
                                                                     const double* __restrict__ rot1, const double* __restrict__ rot2,
                                                                     unsigned W, unsigned H, unsigned K8, unsigned K16,
                                                                     unsigned n_frames, unsigned tasks0, unsigned tasks1, unsigned tiles_c, unsigned nwork,
                                                                     unsigned l2) {
    __shared__ __attribute__((aligned(16))) double lds[3 * SLABD];
    const unsigned id = xcd_contiguous_id(blockIdx.x, nwork);
    const unsigned tasks = tasks0 + tasks1;
    const unsigned task = id % tasks, zt = id / tasks, ct = zt % tiles_c, z = zt / tiles_c;
    const unsigned Hh = H / 2, Hq = H / 4, H8 = H / 8, H16 = H / 16;
    const unsigned tid = threadIdx.x, u = tid >> 5;
    const unsigned mq = lane_quad<MODE>(tid & 31u);
    const unsigned col0 = ct * CT;
    const unsigned nl = W - col0 < CT ? W - col0 : CT;
    unsigned off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned l = tile_line<MODE>(4 * mq + i, l2 != 0);
        off[i] = l * 8 + (u ^ slab_sw<MODE>(l));
    }
    unsigned c = col0 + 4 * mq;
    c = c + 4 <= W ? c : W - 4;
    const float* __restrict__ Pz = IN + (size_t)z * H * W + c;
    auto ld = [&](unsigned r) { return *reinterpret_cast<const f32x4*>(Pz + (size_t)r * W); };
    double* P8[6] = {static_cast<double*>(dp.as), static_cast<double*>(dp.bd), static_cast<double*>(dp.ad), static_cast<double*>(dp.bs),
                     static_cast<double*>(dp.r1), static_cast<double*>(dp.r2)};
    double* P16[4] = {static_cast<double*>(dp.as2), static_cast<double*>(dp.bd2), static_cast<double*>(dp.ad2), static_cast<double*>(dp.bs2)};
    const size_t lines = (size_t)n_frames * W, line_base = (size_t)z * W + col0;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    if (task < tasks0) {
        const unsigned k = 8 * task + u;
        const bool ok = k < H8;
        const unsigned kc = ok ? k : 0;
        const f32x4 d0 = ok ? ld(2 * kc + 1) : z4, d1 = ok ? ld(Hh - 1 - 2 * kc) : z4, d2 = ok ? ld(Hh + 2 * kc + 1) : z4, d3 = ok ? ld(H - 1 - 2 * kc) : z4;
        const f32x4 r1 = ok ? ld(8 * kc) : z4, r2 = ok ? ld(8 * kc + 4) : z4;
        const Rot4 rr = rot_load(rot1, kc, Hq);
        double o[4][6];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            split_one_r((double)d0[i], (double)d1[i], (double)d2[i], (double)d3[i], rr, o[i][0], o[i][1], o[i][2], o[i][3]);
            o[i][4] = (double)r1[i];
            o[i][5] = (double)r2[i];
        }
        // two rounds of three slabs (24 KB of LDS: six blocks per CU)
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
            if (rnd) __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int a = 0; a < 3; ++a) lds[a * SLABD + off[i]] = ok ? o[i][3 * rnd + a] : 0.0;
            __syncthreads();
#pragma unroll
            for (int a = 0; a < 3; ++a) slab_store<MODE>(lds + a * SLABD, P8[3 * rnd + a], lines, line_base, nl, 8 * task, 0xFFu, tid);
        }
    } else if (SPLIT_MID) {
        const unsigned kb = task - tasks0, e = 8 * kb + u;
        const bool ok = e < H16;
        const unsigned ec = ok ? e : 0;
        const f32x4 q0 = ok ? ld(4 * ec + 2) : z4, q1 = ok ? ld(Hh - 2 - 4 * ec) : z4, q2 = ok ? ld(Hh + 4 * ec + 2) : z4, q3 = ok ? ld(H - 2 - 4 * ec) : z4;
        const Rot4 rr = rot_load(rot2, ec, H8);
        double o[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) split_one_r((double)q0[i], (double)q1[i], (double)q2[i], (double)q3[i], rr, o[i][0], o[i][1], o[i][2], o[i][3]);
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
            if (rnd) __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int a = 0; a < 2; ++a) lds[a * SLABD + off[i]] = ok ? o[i][2 * rnd + a] : 0.0;
            __syncthreads();
#pragma unroll
            for (int a = 0; a < 2; ++a) slab_store<MODE>(lds + a * SLABD, P16[2 * rnd + a], lines, line_base, nl, 8 * kb, 0xFFu, tid);
        }
    } else {
        const unsigned kb0 = 3 * (task - tasks0);                      // three k-blocks of M = c[4q+2], K16 = kpad(H/2) wide
        f32x4 v[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const unsigned q = 8 * (kb0 + s) + u;
            v[s] = q < Hq ? ld(4 * q + 2) : z4;
        }
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) lds[s * SLABD + off[i]] = (double)v[s][i];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (8 * (kb0 + s) < K16) slab_store<MODE>(lds + s * SLABD, P16[0], lines, line_base, nl, 8 * (kb0 + s), 0xFFu, tid);
    }
}

// ---------------------------------------------------------------------------------------------
// Inverse column pass at LEVEL 2 (r4c; dct_pair_efold_cols): unit k < H/16 with its mirror unit H/8 - 1 - k in one thread.
//   task < tasks0 (k-blocks of H/16 units): rows 2k+1, H/2-1-2k, H/2+2k+1, H-1-2k of unit k and of unit H/8-1-k -> planes
//       0 .. 7; rows 16k, 16k+8 -> planes 8, 9; rows 8k+4, H-4-8k (R2 at k and at its mirror) -> planes 10, 11
//   then tasks1 k-blocks of AS2 BD2 AD2 BS2 (planes 12 .. 15) as at level 1.  Four rounds of three slabs.
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void prep16_inv_cols_l2_kernel(const float* __restrict__ IN, double* __restrict__ base,
                                                                 const double* __restrict__ rot1, const double* __restrict__ rot2,
                                                                 const double* __restrict__ rot3,
                                                                 unsigned W, unsigned H, unsigned K16,
                                                                 unsigned n_frames, unsigned tasks0, unsigned tasks1, unsigned tiles_c, unsigned nwork,
                                                                 unsigned in_l2) {
    __shared__ __attribute__((aligned(16))) double lds[3 * SLABD];
    const unsigned id = xcd_contiguous_id(blockIdx.x, nwork);
    const unsigned tasks = tasks0 + tasks1;
    const unsigned task = id % tasks, zt = id / tasks, ct = zt % tiles_c, z = zt / tiles_c;
    const unsigned Hh = H / 2, Hq = H / 4, H8 = H / 8, H16 = H / 16;
    const unsigned tid = threadIdx.x, u = tid >> 5;
    const unsigned mq = lane_quad<MODE>(tid & 31u);
    const unsigned col0 = ct * CT;
    const unsigned nl = W - col0 < CT ? W - col0 : CT;
    unsigned off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned l = tile_line<MODE>(4 * mq + i, in_l2 != 0);
        off[i] = l * 8 + (u ^ slab_sw<MODE>(l));
    }
    unsigned c = col0 + 4 * mq;
    c = c + 4 <= W ? c : W - 4;
    const float* __restrict__ Pz = IN + (size_t)z * H * W + c;
    auto ld = [&](unsigned r) { return *reinterpret_cast<const f32x4*>(Pz + (size_t)r * W); };
    auto plane = [&](unsigned a) { return base + (size_t)a * n_frames * W * K16; };
    const size_t lines = (size_t)n_frames * W, line_base = (size_t)z * W + col0;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    if (task < tasks0) {
        const unsigned k = 8 * task + u;
        const bool ok = k < H16;
        const unsigned kc = ok ? k : 0, km = H8 - 1 - kc;
        const f32x4 d0 = ok ? ld(2 * kc + 1) : z4, d1 = ok ? ld(Hh - 1 - 2 * kc) : z4, d2 = ok ? ld(Hh + 2 * kc + 1) : z4, d3 = ok ? ld(H - 1 - 2 * kc) : z4;
        const f32x4 m0 = ok ? ld(2 * km + 1) : z4, m1 = ok ? ld(Hh - 1 - 2 * km) : z4, m2 = ok ? ld(Hh + 2 * km + 1) : z4, m3 = ok ? ld(H - 1 - 2 * km) : z4;
        const f32x4 xa = ok ? ld(16 * kc) : z4, xb = ok ? ld(16 * kc + 8) : z4, r2 = ok ? ld(8 * kc + 4) : z4, r2m = ok ? ld(8 * km + 4) : z4;
        const Rot4 rr = rot_load(rot1, kc, Hq), rm = rot_load(rot1, km, Hq);
        const double c3 = rot3[kc], s3 = rot3[H16 + kc];
        double o[4][12];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double as, bd, ad, bs, asm_, bdm, adm, bsm;
            split_one_r((double)d0[i], (double)d1[i], (double)d2[i], (double)d3[i], rr, as, bd, ad, bs);
            split_one_r((double)m0[i], (double)m1[i], (double)m2[i], (double)m3[i], rm, asm_, bdm, adm, bsm);
            o[i][0] = as + asm_; o[i][1] = as - asm_; o[i][2] = bd + bdm; o[i][3] = bd - bdm;
            const double au = ad * c3 + adm * s3, bu = adm * c3 - ad * s3;
            const double av = bsm * c3 + bs * s3, bv = bs * c3 - bsm * s3;
            o[i][4] = au + av; o[i][5] = bu + bv; o[i][6] = au - av; o[i][7] = bu - bv;
            o[i][8] = (double)xa[i]; o[i][9] = (double)xb[i];
            const double q = (double)r2[i], qm = (double)r2m[i];
            o[i][10] = q * c3 + qm * s3; o[i][11] = qm * c3 - q * s3;
        }
#pragma unroll
        for (int rnd = 0; rnd < 4; ++rnd) {
            if (rnd) __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int a = 0; a < 3; ++a) lds[a * SLABD + off[i]] = ok ? o[i][3 * rnd + a] : 0.0;
            __syncthreads();
#pragma unroll
            for (int a = 0; a < 3; ++a) slab_store<MODE>(lds + a * SLABD, plane(3 * rnd + a), lines, line_base, nl, 8 * task, 0xFFu, tid);
        }
    } else {
        const unsigned kb = task - tasks0, e = 8 * kb + u;
        const bool ok = e < H16;
        const unsigned ec = ok ? e : 0;
        const f32x4 q0 = ok ? ld(4 * ec + 2) : z4, q1 = ok ? ld(Hh - 2 - 4 * ec) : z4, q2 = ok ? ld(Hh + 4 * ec + 2) : z4, q3 = ok ? ld(H - 2 - 4 * ec) : z4;
        const Rot4 rr = rot_load(rot2, ec, H8);
        double o[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) split_one_r((double)q0[i], (double)q1[i], (double)q2[i], (double)q3[i], rr, o[i][0], o[i][1], o[i][2], o[i][3]);
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
            if (rnd) __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int a = 0; a < 2; ++a) lds[a * SLABD + off[i]] = ok ? o[i][2 * rnd + a] : 0.0;
            __syncthreads();
#pragma unroll
            for (int a = 0; a < 2; ++a) slab_store<MODE>(lds + a * SLABD, plane(12 + 2 * rnd + a), lines, line_base, nl, 8 * kb, 0xFFu, tid);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Inverse row pass (n % 128 == 0).  The eight regions of a thread of pair_prep16_inv_rows_kernel are two independent
// sets of four; here a thread = (line, tau), tau < n/64, takes the four 16-coefficient regions
//   g = 16 tau, n/2 - 16 - 16 tau, n/2 + 16 tau, n - 16 - 16 tau
// that meet in the units k = 8 tau .. 8 tau + 7 of the odd part (c[2k+1]: AS BD AD BS, k-block tau), in the units
// 4 tau .. 4 tau + 3 of the level-2 odd part (c[4q+2]: AS2 BD2 AD2 BS2, half a k-block) and hold two doubles of R1 = c[8q]
// and R2 = c[8q+4] each: 64 coefficients in, 64 doubles out.  Block = 32 lines x 8 neighbouring tau: every region is
// read as 8 x 64 = 512 contiguous bytes per line, and the doubles leave through LDS slabs that are images of the
// stored runs -- one slab = one 64-byte k-block piece of one plane for the block's 32 lines = 2 KB contiguous in the
// operand plane.  Four rounds of up to 24 slabs.
// ---------------------------------------------------------------------------------------------
constexpr unsigned RL = 32;                   // lines per block
constexpr unsigned RSL = RL * 8 + 2;          // doubles per row-pass slab: 2 KB + 16 bytes (bank spread of the 8 tau of a line)
constexpr unsigned RNS = 24;                  // slabs

// chunk c (16 bytes) of line l of slab s -> piece `piece` of `plane`; two slabs per sweep of the block
__device__ inline void rows_store(const double* __restrict__ lds, unsigned nslabs, double* const* planes, const unsigned* pieces, const unsigned* masks,
                                  size_t rows_total, size_t row_base, unsigned nl, unsigned tid) {
    const unsigned l = (tid & 127u) >> 2, c = tid & 3u;
    for (unsigned s = tid >> 7; s < nslabs; s += 2) {
        if (l >= nl || !((masks[s] >> c) & 1u)) continue;
        const f64x2 v = *reinterpret_cast<const f64x2*>(lds + s * RSL + l * 8 + 2 * c);
        *reinterpret_cast<f64x2*>(planes[s] + ((size_t)pieces[s] * rows_total + row_base + l) * 8 + 2 * c) = v;
    }
}

__global__ __launch_bounds__(256) void prep16_inv_rows_staged_kernel(const float* __restrict__ X, DeepPlanes dp,
                                                                     const double* __restrict__ rot1, const double* __restrict__ rot2,
                                                                     unsigned rows, unsigned W, unsigned K8, unsigned K16, unsigned tblocks) {
    __shared__ __attribute__((aligned(16))) double lds[RNS * RSL];
    __shared__ double* s_plane[RNS];
    __shared__ unsigned s_piece[RNS], s_mask[RNS];
    const unsigned tid = threadIdx.x, t = tid & 7u, lr = tid >> 3;
    const unsigned tb = blockIdx.x % tblocks, lb = blockIdx.x / tblocks;
    const unsigned NT = W / 64;                                     // region sets per line
    const unsigned tg = 8 * tb + t;
    const size_t row_base = (size_t)lb * RL;
    const unsigned nl = rows - row_base < RL ? (unsigned)(rows - row_base) : RL;
    const bool tok = tg < NT, ok = tok && lr < nl;
    const unsigned nt = NT - 8 * tb < 8 ? NT - 8 * tb : 8u;         // valid tau of the block
    const unsigned R = 16 * (tok ? tg : 0);
    const unsigned Nh = W / 2, Nq = W / 4, N8 = W / 8, N16 = W / 16;
    // planes 0 .. 5: AS BD AD BS R1 R2 (rows * K8 doubles apart), 6 .. 9: AS2 BD2 AD2 BS2 (rows * K16 apart)
    auto plane = [&](unsigned a) {
        return a < 6 ? static_cast<double*>(dp.as) + (size_t)a * rows * K8 : static_cast<double*>(dp.as2) + (size_t)(a - 6) * rows * K16;
    };
    const unsigned g[4] = {R, Nh - 16 - R, Nh + R, W - 16 - R};
    f32x4 c[4][4];
    if (ok) {
        const float* xr = X + (row_base + lr) * W;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) c[j][q] = *reinterpret_cast<const f32x4*>(xr + g[j] + 4 * q);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) c[j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // odd coefficients of region j as two ascending quads of k: element 2 i + 1 -> quad i / 4
    auto odd = [&](int j, int half) { return (f64x4){(double)c[j][2 * half][1], (double)c[j][2 * half][3], (double)c[j][2 * half + 1][1], (double)c[j][2 * half + 1][3]}; };
    // c[4q+2]: element 2 of every quad of a region, q ascending
    auto mid = [&](int j) { return (f64x4){(double)c[j][0][2], (double)c[j][1][2], (double)c[j][2][2], (double)c[j][3][2]}; };
    auto put4 = [&](unsigned slab, unsigned at, const f64x4& v) {
        *reinterpret_cast<f64x2*>(lds + slab * RSL + lr * 8 + at) = (f64x2){v[0], v[1]};
        *reinterpret_cast<f64x2*>(lds + slab * RSL + lr * 8 + at + 2) = (f64x2){v[2], v[3]};
    };
    auto flush = [&](unsigned nslabs) {
        __syncthreads();
        rows_store(lds, nslabs, s_plane, s_piece, s_mask, rows, row_base, nl, tid);
        __syncthreads();
    };
    // ---- rounds 1, 2: AS BD, then AD BS, at k-block tau (the same operations as the units A / mirror of the r3 kernel)
    {
        f64x4 as[2], bd[2], ad[2], bs[2];
#pragma unroll
        for (int half = 0; half < 2; ++half)
            split_unit<double>(odd(0, half), odd(1, 1 - half), odd(2, half), odd(3, 1 - half), rot1, R / 2 + 4 * half, Nq, as[half], bd[half], ad[half], bs[half]);
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
            if (tid < 16) {
                const unsigned tt = tid & 7u, pp = tid >> 3;
                s_plane[tid] = plane(2 * rnd + pp);
                s_piece[tid] = 8 * tb + tt;
                s_mask[tid] = tt < nt ? 0xFu : 0u;
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                put4(t, 4 * half, rnd ? ad[half] : as[half]);
                put4(8 + t, 4 * half, rnd ? bs[half] : bd[half]);
            }
            flush(16);
        }
    }
    // ---- round 3: AS2 BD2 AD2 BS2 at the units 4 tau .. 4 tau + 3: half tau & 1 of piece tau / 2 (8 tb is even)
    {
        if (tid < 16) {
            const unsigned pi = tid & 3u, pp = tid >> 2;            // piece pi of the block (tau = 2 pi, 2 pi + 1), plane pp
            const unsigned v0 = 2 * pi < nt ? 1u : 0u, v1 = 2 * pi + 1 < nt ? 1u : 0u;
            s_plane[tid] = plane(6 + pp);
            s_piece[tid] = 4 * tb + pi;
            s_mask[tid] = (v0 ? 0x3u : 0u) | (v1 ? 0xCu : 0u);
        }
        f64x4 as, bd, ad, bs;
        split_unit<double>(mid(0), mid(1), mid(2), mid(3), rot2, R / 4, N8, as, bd, ad, bs);
        const unsigned at = 4 * (t & 1u), sl = t >> 1;
        put4(sl, at, as); put4(4 + sl, at, bd); put4(8 + sl, at, ad); put4(12 + sl, at, bs);
        flush(16);
    }
    // ---- round 4: R1 = c[8q] (elements 0 of quads 0, 2), R2 = c[8q+4] (quads 1, 3): region j holds the doubles
    // k = g[j] / 8, k + 1.  The 8 tau of the block cover 16 consecutive doubles per region = two or three pieces:
    // slab 12 plane + 3 j + (piece - first piece of the block for j)
    {
        // k of tau' for region j: ascending regions base + 2 tau', descending ones base - 2 tau'
        auto kbase = [&](unsigned j) { return j == 0 ? 0u : j == 1 ? Nh / 8 - 2 : j == 2 ? Nh / 8 : W / 8 - 2; };
        auto kfirst = [&](unsigned j) {                             // smallest k of the block's valid tau
            return (j & 1u) ? kbase(j) - 2 * (8 * tb + nt - 1) : kbase(j) + 2 * (8 * tb);
        };
        if (tid < 24) {
            const unsigned pl = tid / 12, j = (tid % 12) / 3, pi = tid % 3;
            const unsigned p0 = kfirst(j) >> 3;
            unsigned m = 0;
            for (unsigned tt = 0; tt < nt; ++tt) {
                const unsigned k = (j & 1u) ? kbase(j) - 2 * (8 * tb + tt) : kbase(j) + 2 * (8 * tb + tt);
                if ((k >> 3) == p0 + pi) m |= 1u << ((k & 7u) >> 1);
            }
            s_plane[tid] = plane(4 + pl);
            s_piece[tid] = p0 + pi;
            s_mask[tid] = m;
        }
        if (tok) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned k = g[j] / 8;
                const unsigned sl = 3 * j + ((k >> 3) - (kfirst(j) >> 3));
                *reinterpret_cast<f64x2*>(lds + sl * RSL + lr * 8 + (k & 7u)) = (f64x2){(double)c[j][0][0], (double)c[j][2][0]};
                *reinterpret_cast<f64x2*>(lds + (12 + sl) * RSL + lr * 8 + (k & 7u)) = (f64x2){(double)c[j][1][0], (double)c[j][3][0]};
            }
        }
        flush(24);
    }
    if (tb == 0) {                                                  // zero padding [n/8, K8) and [n/16, K16): whole pieces
        const unsigned l = tid >> 3, ch = tid & 7u;                  // thread = (line, double of a piece)
        if (l < nl) {
            for (unsigned k = N8 + ch; k < K8; k += 8)
#pragma unroll
                for (int a = 0; a < 6; ++a) plane(a)[((size_t)(k >> 3) * rows + row_base + l) * 8 + (k & 7u)] = 0.0;
            for (unsigned k = N16 + ch; k < K16; k += 8)
#pragma unroll
                for (int a = 6; a < 10; ++a) plane(a)[((size_t)(k >> 3) * rows + row_base + l) * 8 + (k & 7u)] = 0.0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Inverse row pass at LEVEL 2 (r4c; n % 128 == 0, dct_pair_efold_inv): the operands of launches that all sum n/16 terms
// (build_pass, "deep inverse").  Sixteen planes K16 wide, numbered like the forward level-2 pass:
//   0 .. 3   AS+ AS- BD+ BD-      class E of the odd part c[2k+1] folded once more (exact)
//   4 .. 7   (a, b) of AD plus / minus (a, b) of the reversed BS: class O rotated once more
//   8, 9     c[16 s], c[16 s + 8]                     the quarter-length even part, folded once more
//   10, 11   (a, b) of R2 = c[8 q + 4] rotated
//   12 .. 15 AS2 BD2 AD2 BS2                          the half-length odd part c[4 q + 2], as at level 1
// The folds and rotations pair unit e of AS / BD / AD / BS with unit n/8 - 1 - e, i.e. k-block tau with k-block
// NT - 1 - tau (NT = n/64): a block takes 32 lines x (4 neighbouring tau from the low half + their 4 partners); thread =
// (line, slot t), slots t and 7 - t are partners.  Rounds 1 / 2 stage AS BD (AD BS) of the eight tau in LDS slabs, the
// store sweep combines slab t with slab 7 - t and writes whole 64-byte pieces of four planes; round 3 as at level 1;
// round 4: R2 rotates and c[16 s], c[16 s + 8] gather inside a thread (regions 0 <-> 3 and 1 <-> 2 are mirrors).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep16_inv_rows_l2_kernel(const float* __restrict__ X, double* __restrict__ base,
                                                                 const double* __restrict__ rot1, const double* __restrict__ rot2,
                                                                 const double* __restrict__ rot3,
                                                                 unsigned rows, unsigned W, unsigned K16, unsigned tblocks,
                                                                 unsigned unit_h, unsigned unit_hup) {
    __shared__ __attribute__((aligned(16))) double lds[RNS * RSL];
    __shared__ double* s_plane[RNS];
    __shared__ unsigned s_piece[RNS], s_mask[RNS];
    const unsigned tid = threadIdx.x, t = tid & 7u, lr = tid >> 3;
    const unsigned tb = blockIdx.x % tblocks, lb = blockIdx.x / tblocks;
    const unsigned NT = W / 64, NTh = NT / 2;                       // region sets per line; NT even
    const unsigned tl = t < 4 ? t : 7u - t;                          // the pair's low slot
    const unsigned tlow = 4 * tb + tl;
    const bool grp = t >= 4;                                         // partner side
    const size_t row_base = (size_t)lb * RL;
    const unsigned nl = rows - row_base < RL ? (unsigned)(rows - row_base) : RL;
    const unsigned nt = NTh - 4 * tb < 4 ? NTh - 4 * tb : 4u;       // low slots whose tau is in the low half
    // (a slot beyond nt holds a tau of the upper half -- the partner of another block's slot: rounds 1 / 2 leave its pair
    // to that block, rounds 3 / 4 store its own outputs like any other's: the same values twice)
    const bool ok = lr < nl;
    const unsigned tg = grp ? NT - 1 - tlow : tlow;
    const unsigned R = 16 * tg;
    const unsigned Nh = W / 2, Nq = W / 4, N8 = W / 8, N16 = W / 16;
    auto plane = [&](unsigned a) { return base + (size_t)a * rows * K16; };
    const unsigned g[4] = {R, Nh - 16 - R, Nh + R, W - 16 - R};
    f32x4 c[4][4];
    // r5, fused inverse transform (unit_h = H != 0): the operand lines of a frame are ordered (unit of the column fold, line
    // of the unit; 16 * unit_hup lines per frame, dct_pair_colops.hpp inv_col_unit_row) -- a line reads the coefficient row it
    // holds (512-byte runs per region either way); the lines of the padding units hold zeros.  `rows` counts operand lines.
    size_t src_row = row_base + lr;
    bool src_ok = ok;
    if (unit_h) {
        const unsigned line = (unsigned)(row_base + lr), lpf = 16 * unit_hup, z = line / lpf, rem = line - z * lpf;
        src_ok = ok && (rem >> 4) < unit_h / 16;
        src_row = (size_t)z * unit_h + (src_ok ? inv_col_unit_row(rem >> 4, rem & 15u, unit_h) : 0u);
    }
    if (src_ok) {
        const float* xr = X + src_row * W;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) c[j][q] = *reinterpret_cast<const f32x4*>(xr + g[j] + 4 * q);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) c[j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    auto odd = [&](int j, int half) { return (f64x4){(double)c[j][2 * half][1], (double)c[j][2 * half][3], (double)c[j][2 * half + 1][1], (double)c[j][2 * half + 1][3]}; };
    auto mid = [&](int j) { return (f64x4){(double)c[j][0][2], (double)c[j][1][2], (double)c[j][2][2], (double)c[j][3][2]}; };
    auto put4 = [&](unsigned slab, unsigned at, const f64x4& v) {
        *reinterpret_cast<f64x2*>(lds + slab * RSL + lr * 8 + at) = (f64x2){v[0], v[1]};
        *reinterpret_cast<f64x2*>(lds + slab * RSL + lr * 8 + at + 2) = (f64x2){v[2], v[3]};
    };
    auto flush = [&](unsigned nslabs) {
        __syncthreads();
        rows_store(lds, nslabs, s_plane, s_piece, s_mask, rows, row_base, nl, tid);
        __syncthreads();
    };
    // 16-byte chunk ch of piece `piece` of plane a, line l of the block
    auto out2 = [&](unsigned a, unsigned piece, unsigned l, unsigned ch, f64x2 v) {
        *reinterpret_cast<f64x2*>(plane(a) + ((size_t)piece * rows + row_base + l) * 8 + 2 * ch) = v;
    };
    // ---- rounds 1, 2: unit e = 8 tau + i of slot tl with its mirror n/8 - 1 - e = unit 7 - i of the partner slot
    {
        f64x4 as[2], bd[2], ad[2], bs[2];
#pragma unroll
        for (int half = 0; half < 2; ++half)
            split_unit<double>(odd(0, half), odd(1, 1 - half), odd(2, half), odd(3, 1 - half), rot1, R / 2 + 4 * half, Nq, as[half], bd[half], ad[half], bs[half]);
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                put4(t, 4 * half, rnd ? ad[half] : as[half]);
                put4(8 + t, 4 * half, rnd ? bs[half] : bd[half]);
            }
            __syncthreads();
            // work item = (low slot, line, chunk): 4 x 32 x 4, two per thread; a chunk = units 2 ch, 2 ch + 1
            for (unsigned w = tid; w < 512; w += 256) {
                const unsigned ch = w & 3u, l = (w >> 2) & 31u, sl = w >> 7;
                if (l >= nl || sl >= nt) continue;
                const unsigned piece = 4 * tb + sl;
                const double* pa = lds + sl * RSL + l * 8, *pb = lds + (7 - sl) * RSL + l * 8;
                const f64x2 a0 = *reinterpret_cast<const f64x2*>(pa + 2 * ch), m0 = *reinterpret_cast<const f64x2*>(pb + 6 - 2 * ch);
                const f64x2 a1 = *reinterpret_cast<const f64x2*>(pa + 8 * RSL + 2 * ch), m1 = *reinterpret_cast<const f64x2*>(pb + 8 * RSL + 6 - 2 * ch);
                if (rnd == 0) {          // AS +/- its mirror, BD +/- its mirror
                    out2(0, piece, l, ch, (f64x2){a0[0] + m0[1], a0[1] + m0[0]});
                    out2(1, piece, l, ch, (f64x2){a0[0] - m0[1], a0[1] - m0[0]});
                    out2(2, piece, l, ch, (f64x2){a1[0] + m1[1], a1[1] + m1[0]});
                    out2(3, piece, l, ch, (f64x2){a1[0] - m1[1], a1[1] - m1[0]});
                } else {                 // the rotation of pair_prep16_rows_kernel's class O: a0 = AD, a1 = BS
                    const unsigned e = 8 * piece + 2 * ch;
                    const f64x2 cc = *reinterpret_cast<const f64x2*>(rot3 + e), ss = *reinterpret_cast<const f64x2*>(rot3 + N16 + e);
                    f64x2 oap, obp, oam, obm;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const double ad_ = a0[i], adm = m0[1 - i], bs_ = a1[i], bsm = m1[1 - i];
                        const double au = ad_ * cc[i] + adm * ss[i], bu = adm * cc[i] - ad_ * ss[i];
                        const double av = bsm * cc[i] + bs_ * ss[i], bv = bs_ * cc[i] - bsm * ss[i];
                        oap[i] = au + av; obp[i] = bu + bv;
                        oam[i] = au - av; obm[i] = bu - bv;
                    }
                    out2(4, piece, l, ch, oap); out2(5, piece, l, ch, obp);
                    out2(6, piece, l, ch, oam); out2(7, piece, l, ch, obm);
                }
            }
            __syncthreads();
        }
    }
    // ---- round 3: AS2 BD2 AD2 BS2 at the units 4 tau .. 4 tau + 3: half tau & 1 of piece tau / 2; slots 2 pi, 2 pi + 1 share
    // the block's piece pi: 2 tb + pi (low side), (NT - 4 - 4 tb) / 2 + pi - 2 (partner side)
    {
        if (tid < 16) {
            const unsigned pi = tid & 3u, pp = tid >> 2;
            s_plane[tid] = plane(12 + pp);
            s_piece[tid] = pi < 2 ? 2 * tb + pi : (NT - 4 - 4 * tb) / 2 + (pi - 2);
            s_mask[tid] = 0xFu;
        }
        f64x4 as, bd, ad, bs;
        split_unit<double>(mid(0), mid(1), mid(2), mid(3), rot2, R / 4, N8, as, bd, ad, bs);
        const unsigned at = 4 * (t & 1u), sl = t >> 1;
        put4(sl, at, as); put4(4 + sl, at, bd); put4(8 + sl, at, ad); put4(12 + sl, at, bs);
        flush(16);
    }
    // ---- round 4.  R2 = c[8 q + 4]: region 0 holds q = 2 tau, 2 tau + 1 and region 3 their mirrors n/8 - 1 - q; region 1
    // holds q = n/16 - 2 - 2 tau, + 1 and region 2 their mirrors: (a, b)[q] = (r c + rm s, rm c - r s), table of n/4.
    // c[16 s], c[16 s + 8]: region j holds s = g[j] / 16.  The four slots of a side cover 8 (a, b) or 4 (c[16 s]) consecutive
    // entries per group -- whole or half pieces when n % 256 == 0, else (1920 columns) straddling two pieces: two slabs per
    // group, three sub-rounds of 16 slabs: (a, b) x {pair A, B} x side; c[16 s] x region x side; c[16 s + 8] likewise.
    {
        auto n_of = [&](unsigned pr, unsigned tau) { return pr ? N16 - 2 - 2 * tau : 2 * tau; };
        auto s_of = [&](unsigned j, unsigned tau) { return j == 0 ? tau : j == 1 ? 2 * NT - 1 - tau : j == 2 ? 2 * NT + tau : 4 * NT - 1 - tau; };
        auto tau_of = [&](unsigned side, unsigned sl) { return side ? NT - 1 - (4 * tb + sl) : 4 * tb + sl; };
        const unsigned side = grp ? 1u : 0u;
#pragma unroll
        for (int sub = 0; sub < 3; ++sub) {
            // entry (index inside its plane) of group gi, slot sl; step = doubles per slot
            auto entry = [&](unsigned gi, unsigned sd, unsigned sl) {
                return sub == 0 ? n_of(gi & 1u, tau_of(sd, sl)) : s_of(gi & 3u, tau_of(sd, sl));
            };
            const unsigned ngroups = sub == 0 ? 4u : 8u;          // sub 0: (plane a | b) x 2 pairs x 2 sides = 8 groups of 2 slabs
            if (tid < 16) {
                // slab tid = group * 2 + which piece; sub 0: group = plane * 4 + pair * 2 + side; else group = region * 2 + side
                const unsigned gr = tid >> 1, which = tid & 1u;
                const unsigned pl = sub == 0 ? 10 + (gr >> 2) : 7 + sub;
                const unsigned gi = sub == 0 ? (gr >> 1) & 1u : gr >> 1, sd = gr & 1u;
                const unsigned e0 = entry(gi, sd, 0), e3 = entry(gi, sd, 3);
                const unsigned p0 = (e0 < e3 ? e0 : e3) >> 3;
                unsigned m = 0;
                for (unsigned sl = 0; sl < 4; ++sl) {
                    const unsigned en = entry(gi, sd, sl);
                    if ((en >> 3) == p0 + which) m |= 1u << ((en & 7u) >> 1);
                }
                s_plane[tid] = plane(pl);
                s_piece[tid] = p0 + which;
                s_mask[tid] = m;
            }
            (void)ngroups;
            if (sub == 0) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const unsigned n0 = n_of(pr, tg);
                    const unsigned e0 = n_of(pr, tau_of(side, 0)), e3 = n_of(pr, tau_of(side, 3));
                    const unsigned p0 = (e0 < e3 ? e0 : e3) >> 3;
                    const int jr = pr ? 1 : 0, jm = pr ? 2 : 3;
                    const double r0 = (double)c[jr][1][0], r1 = (double)c[jr][3][0];        // R2[n0], R2[n0 + 1]
                    const double m0 = (double)c[jm][3][0], m1 = (double)c[jm][1][0];        // their mirrors
                    const f64x2 cc = *reinterpret_cast<const f64x2*>(rot3 + n0), ss = *reinterpret_cast<const f64x2*>(rot3 + N16 + n0);
                    const unsigned sl = (2 * pr + side) * 2 + ((n0 >> 3) - p0);
                    *reinterpret_cast<f64x2*>(lds + sl * RSL + lr * 8 + (n0 & 7u)) = (f64x2){r0 * cc[0] + m0 * ss[0], r1 * cc[1] + m1 * ss[1]};
                    *reinterpret_cast<f64x2*>(lds + (8 + sl) * RSL + lr * 8 + (n0 & 7u)) = (f64x2){m0 * cc[0] - r0 * ss[0], m1 * cc[1] - r1 * ss[1]};
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned sv = s_of(j, tg);
                    const unsigned e0 = s_of(j, tau_of(side, 0)), e3 = s_of(j, tau_of(side, 3));
                    const unsigned p0 = (e0 < e3 ? e0 : e3) >> 3;
                    const unsigned sl = (2 * j + side) * 2 + ((sv >> 3) - p0);
                    lds[sl * RSL + lr * 8 + (sv & 7u)] = sub == 1 ? (double)c[j][0][0] : (double)c[j][2][0];
                }
            }
            flush(16);
        }
    }
    if (tb == 0) {                                                  // zero padding [n/16, K16): whole pieces
        const unsigned l = tid >> 3, ch = tid & 7u;
        if (l < nl)
            for (unsigned k = N16 + ch; k < K16; k += 8)
#pragma unroll
                for (int a = 0; a < 16; ++a) plane(a)[((size_t)(k >> 3) * rows + row_base + l) * 8 + (k & 7u)] = 0.0;
    }
}

bool staged_enabled() {
    return tuning(TUNE_PREP_STAGED) != 0;
}

DeepPlanes planes_of(double* base, size_t lines, unsigned K8, unsigned K16) {
    DeepPlanes dp;
    double* p = base;
    const size_t p8 = lines * K8, p16 = lines * K16;
    dp.as = p; dp.bd = p + p8; dp.ad = p + 2 * p8; dp.bs = p + 3 * p8; dp.r1 = p + 4 * p8; dp.r2 = p + 5 * p8;
    p += 6 * p8;
    dp.as2 = p; dp.bd2 = p + p16; dp.ad2 = p + 2 * p16; dp.bs2 = p + 3 * p16;
    return dp;
}

}  // namespace

// The staged kernels take: natural column order (any W % 4 == 0), or class-major tiles of exactly CT columns.
bool dct_pair_prep_staged_cols_ok(size_t w, bool class_major) {
    return staged_enabled() && w % 4 == 0 && w >= 4 && (!class_major || dct_pair_class_tile(w) == CT);
}
bool dct_pair_prep_staged_rows_ok() { return staged_enabled(); }

int launch_prep16_cols_staged(hipStream_t st, const float* in, size_t n_frames, size_t w, size_t h, double* base,
                              const double* rot1, const double* rot2, bool class_major, bool semi, unsigned K8, unsigned K16, bool efold) {
    const unsigned HU = semi ? (unsigned)((h / 8 + 1) / 2) : (unsigned)(h / 16);
    const unsigned sh = (8u - (unsigned)((h / 8) & 7)) & 7u;       // shift of the mirrored units (kernel comment, round B)
    const unsigned groups = (HU + sh + 7) / 8, tiles_c = (unsigned)((w + CT - 1) / CT);
    const unsigned long long nwork = (unsigned long long)groups * tiles_c * n_frames;
    if (nwork > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    const DeepPlanes dp = planes_of(base, n_frames * w, K8, K16);
#define SSW_L(MODEV, SPLITV) prep16_cols_staged_kernel<MODEV, SPLITV><<<(unsigned)nwork, 256, 0, st>>>( \
        in, dp, rot1, rot2, (unsigned)w, (unsigned)h, K8, K16, (unsigned)n_frames, groups, tiles_c, (unsigned)nwork, efold ? 1u : 0u)
    if (class_major) { if (semi) SSW_L(1, false); else SSW_L(1, true); }
    else             { if (semi) SSW_L(0, false); else SSW_L(0, true); }
#undef SSW_L
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

int launch_prep16_inv_cols_staged(hipStream_t st, const float* in, size_t n_frames, size_t w, size_t h, double* base,
                                  const double* rot1, const double* rot2, bool class_major, bool semi, unsigned K8, unsigned K16, bool l2) {
    const unsigned tasks0 = K8 / 8, tasks1 = semi ? (K16 / 8 + 2) / 3 : K16 / 8, tiles_c = (unsigned)((w + CT - 1) / CT);
    const unsigned long long nwork = (unsigned long long)(tasks0 + tasks1) * tiles_c * n_frames;
    if (nwork > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    const DeepPlanes dp = planes_of(base, n_frames * w, K8, K16);
#define SSW_L(MODEV, SPLITV) prep16_inv_cols_staged_kernel<MODEV, SPLITV><<<(unsigned)nwork, 256, 0, st>>>( \
        in, dp, rot1, rot2, (unsigned)w, (unsigned)h, K8, K16, (unsigned)n_frames, tasks0, tasks1, tiles_c, (unsigned)nwork, l2 ? 1u : 0u)
    if (class_major) { if (semi) SSW_L(2, false); else SSW_L(2, true); }
    else             { if (semi) SSW_L(0, false); else SSW_L(0, true); }
#undef SSW_L
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

int launch_prep16_cols_l2(hipStream_t st, const float* in, size_t n_frames, size_t w, size_t h, double* base,
                          const double* rot1, const double* rot2, const double* rot3, bool class_major, bool in_l2, unsigned K16) {
    if (h % 16 != 0 || !rot3) return SSW_ERR_BAD_ARG;
    const unsigned HU = (unsigned)(h / 16), groups = (HU + 7) / 8, tiles_c = (unsigned)((w + CT - 1) / CT);
    const unsigned long long nwork = (unsigned long long)groups * tiles_c * n_frames;
    if (nwork > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
#define SSW_L(MODEV) prep16_cols_l2_kernel<MODEV><<<(unsigned)nwork, 256, 0, st>>>( \
        in, base, rot1, rot2, rot3, (unsigned)w, (unsigned)h, K16, (unsigned)n_frames, groups, tiles_c, (unsigned)nwork, in_l2 ? 1u : 0u)
    if (class_major) SSW_L(1); else SSW_L(0);
#undef SSW_L
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

int launch_prep16_inv_cols_l2(hipStream_t st, const float* in, size_t n_frames, size_t w, size_t h, double* base,
                              const double* rot1, const double* rot2, const double* rot3, bool class_major, bool in_l2, unsigned K16) {
    if (h % 16 != 0 || !rot3) return SSW_ERR_BAD_ARG;
    const unsigned tasks0 = K16 / 8, tasks1 = K16 / 8, tiles_c = (unsigned)((w + CT - 1) / CT);
    const unsigned long long nwork = (unsigned long long)(tasks0 + tasks1) * tiles_c * n_frames;
    if (nwork > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
#define SSW_L(MODEV) prep16_inv_cols_l2_kernel<MODEV><<<(unsigned)nwork, 256, 0, st>>>( \
        in, base, rot1, rot2, rot3, (unsigned)w, (unsigned)h, K16, (unsigned)n_frames, tasks0, tasks1, tiles_c, (unsigned)nwork, in_l2 ? 1u : 0u)
    if (class_major) SSW_L(2); else SSW_L(0);
#undef SSW_L
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

int launch_prep16_inv_rows_l2(hipStream_t st, const float* in, size_t rows, size_t w, double* base,
                               const double* rot1, const double* rot2, const double* rot3, unsigned K16, unsigned unit_h, unsigned unit_hup) {
    if (w % 128 != 0 || !rot3) return SSW_ERR_BAD_ARG;
    const unsigned NT = (unsigned)(w / 64), tblocks = (NT / 2 + 3) / 4;
    const unsigned long long nblk = (unsigned long long)((rows + RL - 1) / RL) * tblocks;
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    prep16_inv_rows_l2_kernel<<<(unsigned)nblk, 256, 0, st>>>(in, base, rot1, rot2, rot3, (unsigned)rows, (unsigned)w, K16, tblocks, unit_h, unit_hup);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

int launch_prep16_inv_rows_staged(hipStream_t st, const float* in, size_t rows, size_t w, double* base,
                                  const double* rot1, const double* rot2, unsigned K8, unsigned K16) {
    const unsigned NT = (unsigned)(w / 64), tblocks = (NT + 7) / 8;
    const unsigned long long nblk = (unsigned long long)((rows + RL - 1) / RL) * tblocks;
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    prep16_inv_rows_staged_kernel<<<(unsigned)nblk, 256, 0, st>>>(in, planes_of(base, rows, K8, K16), rot1, rot2, (unsigned)rows, (unsigned)w, K8, K16, tblocks);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

}  // namespace ssw
