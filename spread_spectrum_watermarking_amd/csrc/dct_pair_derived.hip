// The derived frame's pruned row pass in ONE kernel (r5; verdict r4 #6): RGB frame -> Y -> level-2 operands
// -> gathered-basis GEMM -> compact plane, without the sixteen operand planes (8 B/px written by the pre-pass and read back
// once by nine launches: 17 GB of a 128-frame 4K pass).
//
// Reader::extract (/root/reference/src/algorithm.rs:556-561) reads the derived plane at the k indices of the base plane's
// ordering and nowhere else: prune.hip gathers, per frequency class, the basis rows of the columns a chunk needs, and the row
// pass multiplies every line's operands with them.  Here a block owns 16 image rows (lines) for the whole sum:
//   phase 1  (dct_pair_prep_light.hip) the tile's pixel runs -> Y -> LDS, 16 units of the fold (= four k-steps) at a time;
//   phase 2  lane (line li, unit 4 w + lq) of wave w folds its unit (col_l2_unit's operations and order): its sixteen results ARE
//            the lane's elements of the sixteen planes' MFMA A-fragments for k-step w -- written to LDS for the other waves;
//   phase 3  every wave owns a few (class, 16-column tile) jobs: per k-step A-fragment(s) from LDS, B-fragment(s) of the
//            gathered basis from L2, v_mfma_f64_16x16x4 into the job's accumulators -- the same products summed in the same
//            (ascending k) order as pair_gemm_f64_kernel's subset launches, so the compact plane is bit-identical;
//   end      acc1 + acc2 (split classes: cosine part + sine part) rounded to f32 -- the row pass's rounding point,
//            /root/reference/src/dct2d.rs:152-168 -- into the compact plane [line][cap_total].
// Applies to marks of up to 1024 entries (classes of at most 32 / 16 gathered columns); longer ones take the launches.
//
// Measured (128 4K frames): r5 4.0-4.2 ms against 3.6 (pre-pass) + 1.7 (launches) + the planes' round trip; loads alone 2.0 ms, fold +
// MFMAs alone 1.9.  r6: 3.4 ms -- every vector-memory load of the tile loop unconditional and in a fixed order (buffer loads whose
// range check drops what does not exist), so that the compiler counts its waits: with the fragment loads inside wave-uniform
// branches it had waited for vmcnt(0) in front of every k-step, the next tile's pixels included.  Loads do return in order, too:
// the next tile's pixels requested in front of the fold (-DDF_EARLY_REQUEST=1: behind the fragments of k-steps 0 and 1 only)
// measure 3.7 ms, the fragments of k-steps 2 and 3 then stand behind them.  Fragments two k-steps ahead instead of one: no change.
// -DDF_NO_LOADS / -DDF_NO_FOLD / -DDF_NO_MFMA build the timing variants the r5 numbers came from (wrong results).
#include "dct_pair_split.hpp"
#include "dct_pair_colops.hpp"
#include "dct_pair_yiq_load.hpp"

#include <atomic>
#include <type_traits>

#ifndef DF_EARLY_REQUEST
#define DF_EARLY_REQUEST 0
#endif

namespace ssw {
namespace {

typedef double f64x4d __attribute__((ext_vector_type(4)));
constexpr int DF_LINES = 16, DF_PITCH = 16 * 16 + 8, DF_MAXJ = 5;

struct DfJob {
    const double* y1;          // gathered basis of the class in MFMA fragment order (prune_gather_basis_kernel, frag): per 16 rows [k / 4][k % 4][row % 16]
    const double* y2;          // the sine part's (split classes), or nullptr
    unsigned p1, p2;           // operand planes (numbering of pair_prep16_rows_kernel's level-2 planes)
    unsigned cap, row0;        // gathered rows of the class, first row of this job's tile
    unsigned col0, ncols;      // compact columns col0 .. col0 + ncols - 1
};
struct DfJobs {
    DfJob j[4][DF_MAXJ];
    unsigned n[4];
};

template <int SRC /*1 rgb f32, 2 rgb u8, 3 rgb u16*/>
__global__ __launch_bounds__(256, 2) void prep16_derived_fused_kernel(const void* __restrict__ SRCP, const double* __restrict__ rot1,
                                                              const double* __restrict__ rot2, const double* __restrict__ rot3,
                                                              DfJobs jobs, float* __restrict__ out, unsigned rows, unsigned W,
                                                              unsigned Kp, unsigned cap_total) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ys = smem;                                                            // [16 lines][DF_PITCH] Y of the tile's 16 units
    double* afrag = reinterpret_cast<double*>(smem + DF_LINES * DF_PITCH);       // [4 k-steps][16 planes][64 lanes]
    double* tabs = afrag + 4 * 16 * 64;                                          // [14][16]: the rotation table entries of the tile's units
    const unsigned N8 = W / 8, N16 = W / 16, Nq = W / 4;
    const unsigned tid = threadIdx.x, lane = tid & 63u, li = lane & 15u, lq = lane >> 4;
    const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));      // wave-uniform: the job table is read by scalar loads
    const unsigned line0 = blockIdx.x * DF_LINES;
    const unsigned nj = jobs.n[wv];
    // r6: every vector-memory load of the loop is unconditional and comes in a fixed order (the compiler's counted s_waitcnt
    // vmcnt(N) needs that: with the fragment and pixel loads inside wave-uniform branches it waited for vmcnt(0) in front of
    // every k-step, i.e. for the next tile's pixels as well -- what r5 read as "loads complete in order").  A job that does not
    // exist, a class without a sine part, k-steps behind the padded sum and rows / units outside the frame are buffer ranges
    // of zero records or offsets behind the range: such loads return 0 without a memory access.
    __amdgpu_buffer_rsrc_t jr1[DF_MAXJ], jr2[DF_MAXJ];
    unsigned jp1[DF_MAXJ], jp2[DF_MAXJ];
    bool jon[DF_MAXJ], jon2[DF_MAXJ];
#pragma unroll
    for (int j = 0; j < DF_MAXJ; ++j) {
        const DfJob& jb = jobs.j[wv][j];
        jon[j] = (unsigned)j < nj;
        jon2[j] = jon[j] && jb.y2 != nullptr;
        const size_t at0 = jon[j] ? (size_t)(jb.row0 >> 4) * (Kp / 4) * 64 : 0;      // the job's fragment stream: [k / 4][64 lanes]
        jr1[j] = __builtin_amdgcn_make_buffer_rsrc((void*)(jon[j] ? jb.y1 + at0 : rot1), 0, jon[j] ? (Kp / 4) * 512u : 0u, 0x00020000);
        jr2[j] = __builtin_amdgcn_make_buffer_rsrc((void*)(jon2[j] ? jb.y2 + at0 : rot1), 0, jon2[j] ? (Kp / 4) * 512u : 0u, 0x00020000);
        jp1[j] = jon[j] ? jb.p1 * 64u : 0u;
        jp2[j] = jon2[j] ? jb.p2 * 64u : 0u;
    }
    constexpr unsigned PXB = SRC == 1 ? 12u : SRC == 3 ? 6u : 3u;                   // bytes per pixel of the frames
    const unsigned rows_here = rows - line0 < (unsigned)DF_LINES ? rows - line0 : (unsigned)DF_LINES;
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc((void*)(static_cast<const char*>(SRCP) + (size_t)line0 * W * PXB), 0,
                                                                      rows_here * W * PXB, 0x00020000);
    f64x4d acc[DF_MAXJ][2];
#pragma unroll
    for (int j = 0; j < DF_MAXJ; ++j) { acc[j][0] = (f64x4d){0, 0, 0, 0}; acc[j][1] = (f64x4d){0, 0, 0, 0}; }
    const float* yl = ys + li * DF_PITCH;
    const unsigned el = 4 * wv + lq;                              // this lane's unit inside a tile (phase 2): k-step wv, element lq
    // entry t of unit e: 0 .. 3 rot_load(rot1, e, W/4), 4 .. 7 rot_load(rot1, W/8 - 1 - e, W/4), 8 .. 11 rot_load(rot2, e, W/8), 12 13 rot3[e], rot3[W/16 + e]
    // (read from L2 once per tile by the block instead of four times per fold by every lane: with two or three waves per SIMD
    // four dependent global latencies per fold are not hidden)
    auto tab_src = [&](unsigned idx, unsigned e0) -> const double* {
        const unsigned e = e0 + (idx & 15u), t = idx >> 4;
        if (t >= 14 || e >= N16) return nullptr;
        const unsigned g = t >> 2, f = t & 3u;
        if (g == 3) return rot3 + (f ? N16 + e : e);
        const double* rot = g == 2 ? rot2 : rot1;
        const unsigned Mh = g == 2 ? N8 : Nq, ee = g == 1 ? N8 - 1 - e : e;
        return rot + (f == 0 ? ee : f == 1 ? Mh + ee : f == 2 ? Mh - 1 - ee : 2 * Mh - 1 - ee);
    };
    // phase 1 in two steps (dct_pair_yiq_load.hpp): `request` puts a tile's four quads per lane (and its table entry) in flight,
    // `park` converts them to Y and stores them in LDS.  Vector-memory loads complete in order, so a request may only follow the
    // LAST B-fragment load of the tile being computed (a prefetch issued earlier stands in front of every B-fragment: measured,
    // load time + compute time instead of their maximum): it is issued behind the last k-step's B-fragment loads, in front of the MFMAs of k-steps 2 and 3.
    RawQuad<SRC - 1> raw[4];
    unsigned dst[4];
    double tv = 0.0;
    bool tv_ok = false;
    auto request = [&](unsigned e0) {
        const double* tsrc = tab_src(tid, e0);
        tv = *(tsrc ? tsrc : rot3);                               // (always a load; entries outside the table are dropped by park)
        tv_ok = tsrc != nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned t = tid + 256 * i;
            const unsigned q = t & 3u, u = (t >> 2) & 15u, ll = t >> 6;
            const unsigned v = u < 8 ? u : 15 - u, hv = v >> 1;
            const bool asc = (u & 1u) == 0;
            unsigned px;
            if (u < 8) px = asc ? hv * N8 + e0 + 4 * q : (hv + 1) * N8 - 16 - e0 + 4 * q;
            else       px = asc ? W - (hv + 1) * N8 + e0 + 4 * q : W - 16 - hv * N8 - e0 + 4 * q;
            const unsigned efirst = asc ? e0 + 4 * q : e0 + 12 - 4 * q;
            const bool ok = ll < rows_here && efirst < N16;
            dst[i] = ok ? ll * DF_PITCH + u * 16 + 4 * q : 0xFFFFFFFFu;
            const unsigned off = ok ? (ll * W + px) * PXB : 0x80000000u;           // behind the range: nothing is read
            if (SRC == 1) {
                auto& w = reinterpret_cast<RawQuad<SSW_PIX_F32>&>(raw[i]).w;
                w[0] = __builtin_amdgcn_raw_buffer_load_b128(fr, off, 0, 0);
                w[1] = __builtin_amdgcn_raw_buffer_load_b128(fr, off + 16u, 0, 0);
                w[2] = __builtin_amdgcn_raw_buffer_load_b128(fr, off + 32u, 0, 0);
            } else if (SRC == 3) {
                auto& w = reinterpret_cast<RawQuad<SSW_PIX_U16>&>(raw[i]).w;
                w[0] = __builtin_amdgcn_raw_buffer_load_b64(fr, off, 0, 0);
                w[1] = __builtin_amdgcn_raw_buffer_load_b64(fr, off + 8u, 0, 0);
                w[2] = __builtin_amdgcn_raw_buffer_load_b64(fr, off + 16u, 0, 0);
            } else {
                auto& w = reinterpret_cast<RawQuad<SSW_PIX_U8>&>(raw[i]).w;
                w[0] = __builtin_amdgcn_raw_buffer_load_b32(fr, off, 0, 0);
                w[1] = __builtin_amdgcn_raw_buffer_load_b32(fr, off + 4u, 0, 0);
                w[2] = __builtin_amdgcn_raw_buffer_load_b32(fr, off + 8u, 0, 0);
            }
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (dst[i] == 0xFFFFFFFFu) continue;
            f32x4 y, iv, qv;
            yiq_of_raw4<SRC - 1, false>(raw[i], y, iv, qv);
            *reinterpret_cast<f32x4*>(ys + dst[i]) = y;
        }
        if (tid < 14 * 16) tabs[tid] = tv_ok ? tv : 0.0;
    };
#ifndef DF_NO_LOADS
    request(0);
#endif
    for (unsigned e0 = 0; e0 < N16; e0 += 16) {                   // a tile = 16 units = four k-steps of four
#ifndef DF_NO_LOADS
        // ---- phase 1: 16 lines x 16 runs x 4 quads of pixels -> Y in LDS (pair_prep16_rows_light_kernel's, on runs of 16 pixels)
        park();
        __syncthreads();
#endif
        const unsigned k0 = e0;
        // B-fragments (gathered basis rows, from L2; fragment order of prune_gather_basis_kernel: per tile of 16 gathered rows
        // [k / 4][k % 4][row % 16], a fragment = 512 contiguous bytes, rows behind the class are zero rows) of all jobs for one
        // k-step; those of the first k-step are requested before the fold, those of k-step s + 1 before the MFMAs of k-step s
        // r6: the fragments of a k-step are requested TWO k-steps ahead (those of k-steps 0 and 1 before the fold): one k-step of
        // ten MFMAs (0.27 us) did not cover a trip to L2 under load, and the next tile's pixels can be requested one k-step earlier
        double bf1[4][DF_MAXJ], bf2[4][DF_MAXJ];
        auto load_b = [&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            const unsigned koff = (k0 / 4 + ks) * 512u;           // fragment k / 4 of the stream (behind Kp / 4: zero)
#pragma unroll
            for (int j = 0; j < DF_MAXJ; ++j) {
                const u32x2 w1 = __builtin_amdgcn_raw_buffer_load_b64(jr1[j], lane * 8u, koff, 0);
                const u32x2 w2 = __builtin_amdgcn_raw_buffer_load_b64(jr2[j], lane * 8u, koff, 0);
                bf1[ks][j] = __hiloint2double((int)w1[1], (int)w1[0]);
                bf2[ks][j] = __hiloint2double((int)w2[1], (int)w2[0]);
            }
        };
        using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
        using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
#ifndef DF_NO_MFMA
        load_b(K0{});
        load_b(K1{});
#endif
#if !defined(DF_NO_LOADS) && DF_EARLY_REQUEST
        if (e0 + 16 < N16) request(e0 + 16);           // A/B: the next tile's pixels behind the first two k-steps' fragments, in front of the fold
#endif
        // ---- phase 2: wave wv folds k-step wv: lane = (line li, unit e0 + 4 wv + lq); its sixteen results are the lane's elements
        // of the sixteen planes' A-fragments for that k-step (col_l2_unit's operations and order, in two halves like the light pre-pass)
#ifndef DF_NO_FOLD
        {
            const unsigned e = e0 + el;
            double* af = afrag + (size_t)wv * 16 * 64 + lane;
            auto put = [&](int a, double v) { af[a * 64] = v; };
            if (e >= N16) {
#pragma unroll
                for (int a = 0; a < 16; ++a) put(a, 0.0);
            } else {
                auto X = [&](int u) { return (double)yl[u * 16 + ((u & 1) ? 15 - el : el)]; };
                auto T4 = [&](int t) { return Rot4{tabs[t * 16 + el], tabs[(t + 1) * 16 + el], tabs[(t + 2) * 16 + el], tabs[(t + 3) * 16 + el]}; };
                {
                    double as, bd, ad, bs, asm_, bdm, adm, bsm;
                    split_one_r(X(0) - X(15), X(3) - X(12), X(4) - X(11), X(7) - X(8), T4(0), as, bd, ad, bs);
                    __builtin_amdgcn_sched_barrier(0);
                    split_one_r(X(1) - X(14), X(2) - X(13), X(5) - X(10), X(6) - X(9), T4(4), asm_, bdm, adm, bsm);
                    put(0, as + asm_); put(1, as - asm_); put(2, bd + bdm); put(3, bd - bdm);
                    __builtin_amdgcn_sched_barrier(0);
                    const double c3 = tabs[12 * 16 + el], s3 = tabs[13 * 16 + el];
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
                    const double c3 = tabs[12 * 16 + el], s3 = tabs[13 * 16 + el];
                    put(10, r2 * c3 + r2m * s3); put(11, r2m * c3 - r2 * s3);
                    __builtin_amdgcn_sched_barrier(0);
                    double o12, o13, o14, o15;
                    split_one_r(S[0] - S[7], S[1] - S[6], S[2] - S[5], S[3] - S[4], T4(8), o12, o13, o14, o15);
                    put(12, o12); put(13, o13); put(14, o14); put(15, o15);
                }
            }
        }
#endif
        __syncthreads();
#ifndef DF_NO_MFMA
        // ---- phase 3: the tile's four k-steps; per k-step one MFMA (two: split classes) into each job's accumulators -- consecutive
        // MFMAs never touch the same accumulator (a chain of dependent f64 MFMAs issues at half rate)
        auto kstep = [&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            if (ks == 0) load_b(K2{});
            if (ks == 1) load_b(K3{});
#if !defined(DF_NO_LOADS) && !DF_EARLY_REQUEST
            if (ks == 1 && e0 + 16 < N16) request(e0 + 16);       // behind the tile's last B-fragment load (k-step 3's, just issued)
#endif
            if (k0 + 4 * ks >= N16) return;                       // block-uniform
#pragma unroll
            for (int j = 0; j < DF_MAXJ; ++j) {
                if (!jon[j]) continue;                            // wave-uniform
                const double a1 = afrag[(size_t)ks * 16 * 64 + jp1[j] + lane];
                acc[j][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bf1[ks][j], acc[j][0], 0, 0, 0);
                if (jon2[j]) {
                    const double a2 = afrag[(size_t)ks * 16 * 64 + jp2[j] + lane];
                    acc[j][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, bf2[ks][j], acc[j][1], 0, 0, 0);
                }
            }
        };
        kstep(K0{}); kstep(K1{}); kstep(K2{}); kstep(K3{});
#endif
        __syncthreads();
    }
    // ---- results: D[row = line][col = gathered column]; element r of lane (li, lq) is line 4 r + lq, column li
#pragma unroll
    for (int j = 0; j < DF_MAXJ; ++j) {
        if ((unsigned)j >= nj) continue;
        const DfJob& jb = jobs.j[wv][j];
        if (li >= jb.ncols) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned line = line0 + 4 * r + lq;
            if (line >= rows) continue;
            const double v = jb.y2 ? acc[j][0][r] + acc[j][1][r] : acc[j][0][r];
            out[(size_t)line * cap_total + jb.col0 + li] = (float)v;
        }
    }
}

}  // namespace

// classes: those of build_pruned_derived's level-2 plan (x2 != nullptr: split).  false: take the launches.
bool dct_pair_derived_fused_ok(size_t w, unsigned n_classes, const DerivedFusedClass* cls) {
    if (tuning(TUNE_DERIVED_FUSED) == 0 || w % 64 != 0 || !dct_pair_efold(w) || n_classes > 9) return false;
    unsigned load[4] = {0, 0, 0, 0}, cnt[4] = {0, 0, 0, 0};
    for (unsigned c = 0; c < n_classes; ++c) {
        if (cls[c].cap == 0) continue;
        if (cls[c].split ? cls[c].cap > 32 : cls[c].cap > 16) return false;
        for (unsigned t = 0; t * 16 < cls[c].cap; ++t) {
            unsigned best = 0;
            for (unsigned w2 = 1; w2 < 4; ++w2) if (load[w2] < load[best]) best = w2;
            load[best] += cls[c].split ? 2 : 1;
            if (++cnt[best] > (unsigned)DF_MAXJ) return false;
        }
    }
    return true;
}

int launch_dct_pair_derived_fused(hipStream_t st, int src_kind, const void* rgb, size_t lines, size_t w, const double* rot1,
                                  const double* rot2, const double* rot3, unsigned n_classes, const DerivedFusedClass* cls,
                                  float* out, unsigned cap_total) {
    if (lines == 0) return SSW_OK;
    if (lines > 0xFFFFFFFFull || src_kind < 1 || src_kind > 3) return SSW_ERR_BAD_ARG;
    DfJobs jobs;
    unsigned load[4] = {0, 0, 0, 0};
    for (int w2 = 0; w2 < 4; ++w2) jobs.n[w2] = 0;
    // split classes first (two MFMAs per k-step), every tile to the least loaded wave
    for (int pass = 0; pass < 2; ++pass)
        for (unsigned c = 0; c < n_classes; ++c) {
            if (cls[c].cap == 0 || (cls[c].split ? pass != 0 : pass != 1)) continue;
            for (unsigned t = 0; t * 16 < cls[c].cap; ++t) {
                unsigned best = 0;
                for (unsigned w2 = 1; w2 < 4; ++w2) if (load[w2] < load[best]) best = w2;
                if (jobs.n[best] >= (unsigned)DF_MAXJ) return SSW_ERR_BAD_ARG;
                const unsigned nc = cls[c].cap - 16 * t < 16 ? cls[c].cap - 16 * t : 16u;
                jobs.j[best][jobs.n[best]++] = DfJob{cls[c].y1, cls[c].split ? cls[c].y2 : nullptr, cls[c].p1, cls[c].p2, cls[c].cap, 16 * t,
                                                     cls[c].off + 16 * t, nc};
                load[best] += cls[c].split ? 2 : 1;
            }
        }
    const unsigned Kp = (unsigned)dct_pair_split_kpad(w / 2);
    const unsigned nblk = (unsigned)((lines + DF_LINES - 1) / DF_LINES);
    const size_t smem = DF_LINES * DF_PITCH * sizeof(float) + (4 * 16 * 64 + 14 * 16) * sizeof(double);
    {
        static std::atomic<bool> attr_set[64];
        int dev = 0;
        SSW_HIP_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
            for (const void* f : {reinterpret_cast<const void*>(prep16_derived_fused_kernel<1>), reinterpret_cast<const void*>(prep16_derived_fused_kernel<2>),
                                  reinterpret_cast<const void*>(prep16_derived_fused_kernel<3>)})
                SSW_HIP_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
        }
    }
    if (src_kind == 1) prep16_derived_fused_kernel<1><<<nblk, 256, smem, st>>>(rgb, rot1, rot2, rot3, jobs, out, (unsigned)lines, (unsigned)w, Kp, cap_total);
    else if (src_kind == 2) prep16_derived_fused_kernel<2><<<nblk, 256, smem, st>>>(rgb, rot1, rot2, rot3, jobs, out, (unsigned)lines, (unsigned)w, Kp, cap_total);
    else prep16_derived_fused_kernel<3><<<nblk, 256, smem, st>>>(rgb, rot1, rot2, rot3, jobs, out, (unsigned)lines, (unsigned)w, Kp, cap_total);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

}  // namespace ssw
