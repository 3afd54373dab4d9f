// The level-2 fold of a forward COLUMN pass on one unit -- the arithmetic of prep16_cols_l2_kernel
// (dct_pair_prep_staged.hip), shared with the row GEMM's fused epilogue (dct_pair_f64_kernel.hpp, EPI_FWD_COLOP), which
// applies it to its own accumulators instead of reading the f32 plane between the passes back from HBM.
//
// Reference: src/dct2d.rs:152-168 stores the row pass's results as f32 before the column pass (:172-206) reads them: the
// inputs x[] here ARE those f32 values, whichever kernel hands them over.  Same operations in the same order in both
// callers (the library is built with -ffp-contract=off, nothing reassociates): bit-identical operand planes.
#pragma once
#include "dct_pair_common.hpp"

namespace ssw {

// Row of the frame that line v (0 .. 15) of unit e (< H/16) holds: rows v < 8 are e, H/8-1-e, H/8+e, H/4-1-e, H/4+e,
// 3H/8-1-e, 3H/8+e, H/2-1-e; row 15 - v is the mirror H - 1 - row(v).
__host__ __device__ inline unsigned col_unit_row(unsigned e, unsigned v, unsigned H) {
    const unsigned H8 = H / 8, m = v < 8 ? v : 15 - v;
    const unsigned r = (m >> 1) * H8 + ((m & 1u) ? H8 - 1 - e : e);
    return v < 8 ? r : H - 1 - r;
}
// ... and its inverse: (unit, line) of frame row y
__host__ __device__ inline void col_unit_of_row(unsigned y, unsigned H, unsigned& e, unsigned& v) {
    const unsigned H8 = H / 8, HU = H / 16;
    const bool mir = y >= H / 2;
    const unsigned yy = mir ? H - 1 - y : y;
    const unsigned j = yy / H8, r = yy - j * H8;
    const unsigned m = r < HU ? 2 * j : 2 * j + 1;
    e = r < HU ? r : H8 - 1 - r;
    v = mir ? 15 - m : m;
}

// the rotation tables of one unit: {cos e, sin e, cos m, sin m} of the unit and its rotation partner (rot: [0, Mh) cos psi,
// [Mh, 2 Mh) sin psi of a half length Mh)
struct Rot4 { double cc, ss, ccm, ssm; };
__device__ inline Rot4 rot_load(const double* __restrict__ rot, unsigned e, unsigned Mh) {
    return Rot4{rot[e], rot[Mh + e], rot[Mh - 1 - e], rot[2 * Mh - 1 - e]};
}
__device__ inline void split_one_r(double d0, double d1, double d2, double d3, const Rot4& r, double& as, double& bd, double& ad, double& bs) {
    const double a = d0 * r.cc + d3 * r.ss, b = d3 * r.cc - d0 * r.ss;
    const double am = d1 * r.ccm + d2 * r.ssm, bm = d2 * r.ccm - d1 * r.ssm;
    as = a + am;
    ad = a - am;
    bs = b + bm;
    bd = b - bm;
}

struct ColL2Tab { Rot4 ra, rb, rc; double c3, s3; };
// rot1 / rot2 / rot3: the tables of axes of length H, H/2, H/4
__device__ inline ColL2Tab col_l2_tab(const double* __restrict__ rot1, const double* __restrict__ rot2, const double* __restrict__ rot3,
                                      unsigned e, unsigned H) {
    const unsigned Hq = H / 4, H8 = H / 8, HU = H / 16;
    return ColL2Tab{rot_load(rot1, e, Hq), rot_load(rot1, H8 - 1 - e, Hq), rot_load(rot2, e, H8), rot3[e], rot3[HU + e]};
}

// x[v]: the sixteen f32 values of a unit's rows (col_unit_row); o[a]: entry e of the sixteen operand planes, by number:
//   0 .. 3 AS+ AS- BD+ BD-   4 .. 7 (a, b) of AD plus / minus (a, b) of the reversed BS   8 9 R1+ R1-   10 11 (a, b) of R2
//   12 .. 15 AS2 BD2 AD2 BS2
__device__ inline void col_l2_unit(const float (&x)[16], const ColL2Tab& t, double (&o)[16]) {
    double D[8], S[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        S[v] = (double)x[v] + (double)x[15 - v];
        D[v] = (double)x[v] - (double)x[15 - v];
    }
    double as, bd, ad, bs, asm_, bdm, adm, bsm;
    split_one_r(D[0], D[3], D[4], D[7], t.ra, as, bd, ad, bs);            // unit e
    split_one_r(D[1], D[2], D[5], D[6], t.rb, asm_, bdm, adm, bsm);       // unit H/8 - 1 - e
    const double ss0 = S[0] + S[7], ss3 = S[3] + S[4], ss1 = S[1] + S[6], ss2 = S[2] + S[5];
    const double r1 = ss0 + ss3, r2 = ss0 - ss3, r1m = ss1 + ss2, r2m = ss1 - ss2;
    o[0] = as + asm_; o[1] = as - asm_; o[2] = bd + bdm; o[3] = bd - bdm; o[8] = r1 + r1m; o[9] = r1 - r1m;
    const double c3 = t.c3, s3 = t.s3;
    const double au = ad * c3 + adm * s3, bu = adm * c3 - ad * s3;
    const double av = bsm * c3 + bs * s3, bv = bs * c3 - bsm * s3;
    o[4] = au + av; o[5] = bu + bv; o[6] = au - av; o[7] = bu - bv;
    o[10] = r2 * c3 + r2m * s3; o[11] = r2m * c3 - r2 * s3;
    double SD[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) SD[v] = S[v] - S[7 - v];
    split_one_r(SD[0], SD[1], SD[2], SD[3], t.rc, o[12], o[13], o[14], o[15]);
}

// ---- the INVERSE column pass at level 2 (prep16_inv_cols_l2_kernel) ------------------------------------------------------
// Coefficient row that line v of unit k (< H/16) holds, km = H/8 - 1 - k the unit's mirror:
//   v = 0 .. 3   2k+1, H/2-1-2k, H/2+2k+1, H-1-2k          (the odd part's unit k)
//   v = 4 .. 7   the same rows of unit km
//   v = 8, 9     16k, 16k+8                                 (the eighth-length even part)
//   v = 10, 11   8k+4, 8km+4                                (R2 at k and at its mirror)
//   v = 12 .. 15 4k+2, H/2-2-4k, H/2+4k+2, H-2-4k           (the half-length odd part's unit k)
__host__ __device__ inline unsigned inv_col_unit_row(unsigned k, unsigned v, unsigned H) {
    const unsigned Hh = H / 2, km = H / 8 - 1 - k;
    if (v < 8) {
        const unsigned q = v < 4 ? k : km;
        switch (v & 3u) { case 0: return 2 * q + 1; case 1: return Hh - 1 - 2 * q; case 2: return Hh + 2 * q + 1; default: return H - 1 - 2 * q; }
    }
    if (v < 12) return v == 8 ? 16 * k : v == 9 ? 16 * k + 8 : v == 10 ? 8 * k + 4 : 8 * km + 4;
    switch (v & 3u) { case 0: return 4 * k + 2; case 1: return Hh - 2 - 4 * k; case 2: return Hh + 4 * k + 2; default: return H - 2 - 4 * k; }
}
// x[v]: the sixteen f32 values of the unit's rows (the inverse row pass's results, rounded like the store between the
// passes, src/dct2d.rs:152-168); o[a]: entry k of the sixteen operand planes of prep16_inv_cols_l2_kernel, by number.
// The map is block diagonal -- planes 0 .. 7 from lines 0 .. 7, planes 8 .. 15 from lines 8 .. 15 -- and the two halves are
// separate functions so that a caller can hold one half at a time.  Tables: col_l2_tab (the same five as the forward fold).
__device__ inline void inv_col_l2_unit_lo(const float (&x)[8], const ColL2Tab& t, double (&o)[8]) {
    double as, bd, ad, bs, asm_, bdm, adm, bsm;
    split_one_r((double)x[0], (double)x[1], (double)x[2], (double)x[3], t.ra, as, bd, ad, bs);
    split_one_r((double)x[4], (double)x[5], (double)x[6], (double)x[7], t.rb, asm_, bdm, adm, bsm);
    o[0] = as + asm_; o[1] = as - asm_; o[2] = bd + bdm; o[3] = bd - bdm;
    const double c3 = t.c3, s3 = t.s3;
    const double au = ad * c3 + adm * s3, bu = adm * c3 - ad * s3;
    const double av = bsm * c3 + bs * s3, bv = bs * c3 - bsm * s3;
    o[4] = au + av; o[5] = bu + bv; o[6] = au - av; o[7] = bu - bv;
}
__device__ inline void inv_col_l2_unit_hi(const float (&x)[8], const ColL2Tab& t, double (&o)[8]) {
    o[0] = (double)x[0]; o[1] = (double)x[1];
    const double q = (double)x[2], qm = (double)x[3];
    o[2] = q * t.c3 + qm * t.s3; o[3] = qm * t.c3 - q * t.s3;
    split_one_r((double)x[4], (double)x[5], (double)x[6], (double)x[7], t.rc, o[4], o[5], o[6], o[7]);
}

// memory column, inside a class-major tile of 128 frequencies at level 2 (ForwardClassLayout{n, 128, true}), of the tile's
// natural frequency j: 8 * class(j mod 16) + j / 16
__host__ __device__ inline unsigned fwd_cm128_pos(unsigned j) {
    // class of residue r (nibble r): residues 0 8 4 12 2 14 10 6 1 15 9 7 5 11 3 13 are classes 0 .. 15
    const unsigned long long cls = 0x95F3D6A1B7C2E480ull;      // nibble r = class of residue r (tests/cpp/class_layout_test.cpp checks it against the layout)
    return 8u * (unsigned)((cls >> (4 * (j & 15u))) & 15ull) + (j >> 4);
}

}  // namespace ssw
