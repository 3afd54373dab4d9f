// The rotation + fold of the split odd half (dct_pair_prep.hip, "Split odd half") and the operand planes of a deep
// pass: shared by the pre-pass kernels of dct_pair_prep.hip and dct_pair_prep_staged.hip.
#pragma once
#include "dct_pair_common.hpp"

namespace ssw {

// the same for a single e (scalar kernels)
template <typename T>
__device__ inline void split_one(T d0, T d1, T d2, T d3, const double* __restrict__ rot, unsigned e, unsigned Mh, T& as, T& bd, T& ad, T& bs) {
    const T cc = (T)rot[e], ss = (T)rot[Mh + e], ccm = (T)rot[Mh - 1 - e], ssm = (T)rot[2 * Mh - 1 - e];
    const T a = d0 * cc + d3 * ss, b = d3 * cc - d0 * ss;
    const T am = d1 * ccm + d2 * ssm, bm = d2 * ccm - d1 * ssm;
    as = a + am;
    ad = a - am;
    bs = b + bm;
    bd = b - bm;
}
// One unit of the split: four consecutive e = base .. base + 3 of a DCT-IV input d of length M (Mh = M/2) given as
// ascending quads  dA: d[base + i], dB: d[Mh-4-base + i], dC: d[Mh+base + i], dD: d[M-4-base + i];
// rot: [0, Mh) cos psi, [Mh, 2 Mh) sin psi.  Same operations in the same order as pair_rotate_kernel.
template <typename T>
__device__ inline void split_unit(const vec4_t<T>& dA, const vec4_t<T>& dB, const vec4_t<T>& dC, const vec4_t<T>& dD,
                                  const double* __restrict__ rot, unsigned base, unsigned Mh,
                                  vec4_t<T>& as, vec4_t<T>& bd, vec4_t<T>& ad, vec4_t<T>& bs) {
    const f64x4 c = *reinterpret_cast<const f64x4*>(rot + base), s = *reinterpret_cast<const f64x4*>(rot + Mh + base);
    const f64x4 cm = *reinterpret_cast<const f64x4*>(rot + Mh - 4 - base), sm = *reinterpret_cast<const f64x4*>(rot + 2 * Mh - 4 - base);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const T d0 = dA[i], d1 = dB[3 - i], d2 = dC[i], d3 = dD[3 - i];
        const T cc = (T)c[i], ss = (T)s[i], ccm = (T)cm[3 - i], ssm = (T)sm[3 - i];
        const T a = d0 * cc + d3 * ss, b = d3 * cc - d0 * ss;
        const T am = d1 * ccm + d2 * ssm, bm = d2 * ccm - d1 * ssm;
        as[i] = a + am;
        ad[i] = a - am;
        bs[i] = b + bm;
        bd[i] = b - bm;
    }
}

struct DeepPlanes {        // device pointers of one pass's operand planes
    void *as, *bd, *ad, *bs;         // K8 wide
    void *r1, *r2;                   // K8 wide
    void *as2, *bd2, *ad2, *bs2;     // K16 wide
    // forward ROW passes at level 2 (r4b, dct_pair_efold): twelve planes K16 wide replace the six K8 wide ones --
    //   asp asm_ bdp bdm   class E folded once more: AS +/- its mirror, BD +/- its mirror (exact)
    //   oap obp oam obm    class O (a DCT-IV of AD and a DST-IV of BS, length n/8) rotated once more: (a, b) of AD plus /
    //                      minus (a, b) of the reversed BS
    //   r1p r1m            R1 +/- its mirror (exact)
    //   r2a r2b            R2 (a DCT-IV input of length n/8) rotated: (a, b)
    void *asp = nullptr, *asm_ = nullptr, *bdp = nullptr, *bdm = nullptr;
    void *oap = nullptr, *obp = nullptr, *oam = nullptr, *obm = nullptr;
    void *r1p = nullptr, *r1m = nullptr, *r2a = nullptr, *r2b = nullptr;
};

// dct_pair_prep_light.hip: the level-2 row pre-pass in the form that fits beside the GEMMs
bool dct_pair_prep_light_ok(size_t w, size_t lines);
int launch_dct_pair_prep16_rows_light(hipStream_t st, int src_kind, const void* src, const DeepPlanes& dp, const double* rot1,
                                      const double* rot2, const double* rot3, float* ip, float* qp, size_t rows, size_t w, unsigned K16,
                                      unsigned unit_h, unsigned unit_hup);

bool dct_pair_inv_prep_light_ok(size_t w, size_t lines);
int launch_prep16_inv_rows_light(hipStream_t st, const float* in, size_t rows, size_t w, double* base, const double* rot1,
                                 const double* rot2, const double* rot3, unsigned K16, unsigned unit_h, unsigned unit_hup);
// dct_pair_derived.hip: the derived frame's pruned row pass in one kernel (marks of up to 1024 entries)
struct DerivedFusedClass {
    const double *y1, *y2;     // gathered bases (y2: the sine part of a split class)
    unsigned p1, p2;           // operand planes by number
    unsigned cap, off;         // gathered rows = compact columns off .. off + cap - 1
    bool split;
};
bool dct_pair_derived_fused_ok(size_t w, unsigned n_classes, const DerivedFusedClass* cls);
int launch_dct_pair_derived_fused(hipStream_t st, int src_kind, const void* rgb, size_t lines, size_t w, const double* rot1,
                                  const double* rot2, const double* rot3, unsigned n_classes, const DerivedFusedClass* cls,
                                  float* out, unsigned cap_total);

}  // namespace ssw
