// Internal declarations shared by the HIP translation units of libssw_hip.so.
// gfx950 (MI355X / CDNA4) only: wave64, MFMA, 160 KiB LDS per CU.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <map>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/ssw.h"

namespace ssw {

void set_last_error(const std::string& s);

#define SSW_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess) {                                                         \
            ::ssw::set_last_error(std::string(#expr) + ": " + hipGetErrorString(e__));   \
            return (e__ == hipErrorOutOfMemory) ? SSW_ERR_OUT_OF_MEMORY : SSW_ERR_HIP;   \
        }                                                                                \
    } while (0)

#define SSW_TRY(expr)                         \
    do {                                      \
        int s__ = (expr);                     \
        if (s__ != SSW_OK) return s__;        \
    } while (0)

// ---- launchers implemented in the kernel translation units -----------------
// All are asynchronous on `st` and return the status of the launch.

// color.hip
int launch_rgb_to_yiq(hipStream_t st, const float* rgb, size_t npix, float* y, float* i, float* q);
int launch_yiq_to_rgb(hipStream_t st, const float* y, const float* i, const float* q, size_t npix,
                      float* rgb);
int launch_synth(hipStream_t st, uint32_t seed, uint32_t first_frame, size_t n_frames, size_t w,
                 size_t h, float* rgb);

// dct.hip
// Basis matrices, layout [out][sum], N x N:
//   forward: D[k][n] = 2 cos(pi k (2n+1) / 2N)          (rustdct DCT-II x the reference's x2)
//   inverse: E[n][k] = k == 0 ? 1/4 : cos(pi k (2n+1) / 2N) / 2   (rustdct DCT-III x 1/2)
size_t dense_basis_kpad(size_t n);   // row stride of a dense basis: N rounded up to 32, zero padded
int launch_make_basis_f32(hipStream_t st, size_t n, bool inverse, float* out);
int launch_make_basis_f64(hipStream_t st, size_t n, bool inverse, double* out);

struct Epilogue {    // stored value = f32(acc) * (output index == 0 ? first : base); both 1 = plain store
    float first, base;
};
// Row pass: out[r][v] = sum_c in[r][c] * basis[v][c]   for rows = n_frames*h rows of width w.
int launch_dct_rows(hipStream_t st, int precision, const float* in, float* out, size_t rows, size_t w,
                    const void* basis, Epilogue ep);
// Column pass: out[f][u][c] = sum_r basis[u][r] * in[f][r][c].
int launch_dct_cols(hipStream_t st, int precision, const float* in, float* out, size_t n_frames,
                    size_t w, size_t h, const void* basis, Epilogue ep);

// dct_folded.hip: even/odd-folded f32 GEMMs (half the multiply-adds); see the file header.
// Half bases: (N/2)x(N/2), layout [out][sum], parity 0 = even frequencies, 1 = odd.
size_t half_basis_kpad(size_t n);   // row stride of a half basis: N/2 rounded up to the k-step, zero padded
int launch_make_half_basis_f32(hipStream_t st, size_t n, bool inverse, int parity, float* out);
// false in the default build: in-kernel folding (dct_folded*.hip) and the f32 operand-ready twin (dct_pair_f32.hip) are not
// compiled in (dct_strategies_off.hip); `make ALL_STRATEGIES=1` builds the diagnostic library with every strategy
bool build_all_strategies();
bool dct_rows_can_fold(size_t w, const float* in, const float* out);
bool dct_cols_can_fold(size_t w, size_t h, const float* in, const float* out);
int launch_dct_rows_folded_f32(hipStream_t st, bool inverse, const float* in, float* out, size_t rows,
                               size_t w, const float* b_even, const float* b_odd, Epilogue ep);
int launch_dct_cols_folded_f32(hipStream_t st, bool inverse, const float* in, float* out, size_t n_frames,
                               size_t w, size_t h, const float* b_even, const float* b_odd, Epilogue ep);

// dct_folded_f64.hip: the same folding in f64 (canonical precision), f64 half bases.
int launch_make_half_basis_f64(hipStream_t st, size_t n, bool inverse, int parity, double* out);
int launch_dct_rows_folded_f64(hipStream_t st, bool inverse, const float* in, float* out, size_t rows,
                               size_t w, const double* b_even, const double* b_odd, Epilogue ep);
int launch_dct_cols_folded_f64(hipStream_t st, bool inverse, const float* in, float* out, size_t n_frames,
                               size_t w, size_t h, const double* b_even, const double* b_odd, Epilogue ep);

// dct_pair_prep.hip / dct_pair_f64.hip / dct_pair_f32.hip: "operand-ready" folded GEMMs (no VALU work in
// the MFMA loop): pre-passes write the folded operands once per pass as k-blocked planes in the GEMM's
// precision (f64 flag), the half bases are cached in the same layout.
size_t dct_pair_kpad(bool f64, size_t n);                               // row stride of operands / bases of a length-n axis
size_t dct_pair_operand_elems(bool f64, size_t n_frames, size_t w, size_t h);   // elements per operand plane
bool dct_pair_can_run(bool f64, size_t n_frames, size_t w, size_t h, const float* in, const float* out);
bool dct_pair_can_fold2(size_t len);
bool dct_pair_can_fold2_cols(size_t len);   // column passes: H % 8 == 0 suffices (1080 rows)
int launch_make_half_basis_blocked(hipStream_t st, bool f64, size_t n, bool inverse, int parity, void* out);
// one level: (S, D) forward / (E, O) inverse
int launch_dct_pair_prep(hipStream_t st, bool f64, bool is_row, bool inverse, const float* in, size_t n_frames, size_t w,
                         size_t h, void* o1, void* o2);
// two levels: (SS, SD | EE, EO) [kpad(len/2) wide] and (D | O) [kpad(len) wide] in one sweep
int launch_dct_pair_prep4(hipStream_t st, bool f64, bool is_row, bool inverse, const float* in, size_t n_frames, size_t w,
                          size_t h, void* q1, void* q2, void* p);
// first pass of a rows-first forward transform straight from interleaved RGB (u8 or f32), two levels;
// ip / qp: I and Q planes out (both or neither)
// Sample format of an interleaved RGB frame at the boundary (`u8` parameters below): what `into_rgb32f()` accepts
// (src/algorithm.rs:308, :476): f32 as it is, 8-bit v / 255, 16-bit v / 65535
enum { SSW_PIX_F32 = 0, SSW_PIX_U8 = 1, SSW_PIX_U16 = 2 };
inline size_t pix_bytes(int fmt) { return fmt == SSW_PIX_U8 ? 1 : fmt == SSW_PIX_U16 ? 2 : 4; }          // per sample
inline unsigned pix_align_mask(int fmt) { return fmt == SSW_PIX_U8 ? 3u : fmt == SSW_PIX_U16 ? 7u : 15u; }   // of a 4-pixel load
inline int pix_src_kind(int fmt) { return fmt + 1; }                     // SRC of the row pre-passes: 1 f32, 2 u8, 3 u16
bool dct_pair_can_prep_from_rgb(size_t w, size_t h, const void* rgb, int u8);
int launch_dct_pair_prep4_rows_rgb(hipStream_t st, bool f64, int u8, const void* rgb, size_t n_frames, size_t w, size_t h,
                                   void* q1, void* q2, void* p, float* ip, float* qp);
// three levels on a forward row pass: (SSS, SS-) [kpad(w/4) wide], S- [kpad(w/2)], x- [kpad(w)] from an f32
// plane (src_kind 0) or interleaved RGB f32 / u8 (1 / 2; ip / qp: I, Q planes out or null)
bool dct_pair_can_fold3(size_t len);
int launch_dct_pair_prep8_rows(hipStream_t st, bool f64, int src_kind, const void* src, size_t n_frames, size_t w, size_t h,
                               void* r1, void* r2, void* m, void* p, float* ip, float* qp);
// Writer::result fused into the last inverse pass (EPI_INV_O_RGB): the frames' I / Q planes and the RGB output
struct RgbSink {
    const float* iq_i = nullptr;
    const float* iq_q = nullptr;
    void* rgb = nullptr;          // [n][h][w][3] f32, or u8 when `u8`
    bool u8 = false;
};
int launch_dct_pair_prep8_cols(hipStream_t st, bool f64, const float* in, size_t n_frames, size_t w, size_t h,
                               void* r1, void* r2, void* m, void* p);
// kind 0: one folding level; 1 / 2: the even / odd half of two levels; sub: see dct_pair_f64.hip;
// sink (inverse column pass, kind 2 only): colour conversion in the epilogue instead of storing Y
int launch_dct_pair_gemm_f64(hipStream_t st, bool is_row, bool inverse, int kind, int sub, const double* x1, const double* x2,
                             const double* y1, const double* y2, float* out, double* tmp, size_t n_frames, size_t w,
                             size_t h, Epilogue ep, const RgbSink* sink = nullptr, double* tmp_out = nullptr, bool class_major = false);
// r5, fused transforms (both directions): how a launch takes part.
//   FUSE_ROWS_COP    row launch over the unit-ordered, padded lines whose epilogue writes the sixteen column-operand planes
//                    `cop` (forward: every launch; inverse: the four launches of the odd part); needs the rotation tables of
//                    H, H/2, H/4
//   FUSE_ROWS_LINES  row launch over the same lines with its usual epilogue (inverse: the launches that exchange A1 / T2 / E)
//   FUSE_COLS        column launch behind such a row pass: its 128-line tiles are in the row launches' class-major order
enum { FUSE_ROWS_COP = 1, FUSE_ROWS_LINES = 2, FUSE_COLS = 3 };
struct FuseCols { int mode = 0; double* cop = nullptr; const double *rot1 = nullptr, *rot2 = nullptr, *rot3 = nullptr; };
// several classes (same lines, same template instance) in one launch: single frames, whose launches are too small alone
struct PairClassDesc { int kind, sub; const double *x1, *x2, *y1, *y2; };
int launch_dct_pair_gemm_multi_f64(hipStream_t st, bool is_row, bool inverse, int n_classes, const PairClassDesc* desc, float* out,
                                   double* tmp, size_t n_frames, size_t w, size_t h, Epilogue ep, const RgbSink* sink = nullptr,
                                   double* tmp_out = nullptr, bool class_major = false, const FuseCols* fuse = nullptr);
int launch_dct_pair_gemm_f32(hipStream_t st, bool is_row, bool inverse, int kind, int sub, const float* x1, const float* x2,
                             const float* y1, const float* y2, float* out, float* tmp, size_t n_frames, size_t w,
                             size_t h, Epilogue ep, const RgbSink* sink = nullptr);

int launch_dct_pair_gemm_rows_subset_f64(hipStream_t st, const double* x, const double* y, unsigned cap, unsigned Kp, float* out,
                                         unsigned out_stride, unsigned off, size_t lines);
int launch_dct_pair_gemm_rows_subset_split_f64(hipStream_t st, const double* x1, const double* x2, const double* y1, const double* y2,
                                               unsigned cap, unsigned Kp, float* out, unsigned out_stride, unsigned off, size_t lines);
// the classes of a small pruned row pass in one or two launches (x2 == nullptr: y1 holds cap gathered rows, pairs j / j + cap/2;
// else a class of the split odd half: y1 / y2 the gathered cosine / sine rows)
struct PairSubsetClass {
    const double *x1, *x2, *y1, *y2;
    unsigned cap, Kp, off;
};
int launch_dct_pair_gemm_rows_subset_merged_f64(hipStream_t st, const PairSubsetClass* classes, unsigned n_classes, float* out,
                                                unsigned out_stride, size_t lines);
// split odd half (f64 only; dct_pair_prep.hip "Split odd half"): quarter-length cosine / sine bases (which: 0 cosE, 1 sinE,
// 2 cosO, 3 sinO), the rotation table of an axis, and the pass that turns an odd operand plane into AS | BD | AD | BS
bool dct_pair_can_split(size_t len, bool is_row);
size_t dct_pair_split_kpad(size_t len);
// tuning.hip: the process-wide table of strategy thresholds / A-B switches (ssw_tuning_set, include/ssw.h)
enum { TUNE_EFOLD_MIN, TUNE_EFOLD_INV_MIN, TUNE_EFOLD_COLS_MIN, TUNE_CLASS_TILE, TUNE_DEEP_MIN_ROWS, TUNE_DEEP_MIN_COLS, TUNE_PREP_STAGED,
       TUNE_MERGE_MAX_LINES, TUNE_BN32, TUNE_BAND_SPLIT, TUNE_FUSE_COLS, TUNE_FUSE_INV_COLS, TUNE_UPLOAD_BANDS, TUNE_SPECULATE_K, TUNE_PREP_LIGHT, TUNE_LANE_STAGGER, TUNE_DERIVED_FUSED, TUNE_INV_PREP_LIGHT, TUNE_GEMM_STAGGER, TUNE_GEMM_GROUP_M, TUNE_GEMM_GROUP_M_ROWS, TUNE_MERGE_BATCH, TUNE_TILE48, TUNE_COUNT };
long long tuning(int which);
unsigned dct_pair_class_tile(size_t len);               // tile width of the class-major plane orders (dct_pair_common.hpp)
bool dct_pair_efold(size_t len);                        // forward row passes of this length run at level 2 (r4b)
bool dct_pair_efold_inv(size_t len);                    // inverse row passes of this length run at level 2 (r4c)
bool dct_pair_efold_cols(size_t h, size_t w, bool class_major);      // column passes (both directions) of h rows run at level 2 (r4c)
int launch_prep16_cols_l2(hipStream_t st, const float* in, size_t n_frames, size_t w, size_t h, double* base,
                          const double* rot1, const double* rot2, const double* rot3, bool class_major, bool in_l2, unsigned K16);
int launch_prep16_inv_cols_l2(hipStream_t st, const float* in, size_t n_frames, size_t w, size_t h, double* base,
                              const double* rot1, const double* rot2, const double* rot3, bool class_major, bool in_l2, unsigned K16);
// LDS-staged forms of the deep pre-passes (dct_pair_prep_staged.hip; SSW_PREP_STAGED=0 keeps the r3 kernels)
bool dct_pair_prep_staged_cols_ok(size_t w, bool class_major);
bool dct_pair_prep_staged_rows_ok();
int launch_prep16_cols_staged(hipStream_t st, const float* in, size_t n_frames, size_t w, size_t h, double* base,
                              const double* rot1, const double* rot2, bool class_major, bool semi, unsigned K8, unsigned K16, bool efold);
int launch_prep16_inv_cols_staged(hipStream_t st, const float* in, size_t n_frames, size_t w, size_t h, double* base,
                                  const double* rot1, const double* rot2, bool class_major, bool semi, unsigned K8, unsigned K16, bool l2);
int launch_prep16_inv_rows_l2(hipStream_t st, const float* in, size_t rows, size_t w, double* base,
                              const double* rot1, const double* rot2, const double* rot3, unsigned K16, unsigned unit_h = 0, unsigned unit_hup = 0);
int launch_prep16_inv_rows_staged(hipStream_t st, const float* in, size_t rows, size_t w, double* base,
                                  const double* rot1, const double* rot2, unsigned K8, unsigned K16);
size_t dct_pair_split_elems(size_t n_frames, size_t w, size_t h);
size_t dct_pair_split_basis_rows(size_t len, int which);
int launch_make_split_basis_blocked(hipStream_t st, size_t n, bool inverse, int which, double* out);
int launch_make_rot_table(hipStream_t st, size_t n, double* out);
int launch_dct_pair_rotate(hipStream_t st, const double* p, const double* rot, double* sp, size_t lines, size_t len);
// deep forward row pre-pass (len % 64 == 0): D and SD split, SS folded a third time, in one sweep over the source
// (f32 plane or interleaved RGB); base: AS BD AD BS R1 R2 (lines * split_kpad(len) each), AS2 BD2 AD2 BS2 (lines * split_kpad(len/2))
bool dct_pair_can_deep_rows(size_t len);
bool dct_pair_can_deep_cols(size_t len);                 // H % 16 == 0
bool dct_pair_can_deep_inv_rows(size_t len);             // W % 128 == 0
bool dct_pair_can_semi_deep_cols(size_t len);            // H % 8 == 0, not % 16: launch_dct_pair_prep16_cols leaves SD whole
size_t dct_pair_semi_deep_elems(size_t lines, size_t len);
// deep inverse pre-passes (coefficient plane -> the same ten planes; R1 = c[8q], R2 = c[8q+4])
int launch_dct_pair_prep16_inv_rows(hipStream_t st, const float* in, size_t n_frames, size_t w, size_t h, double* base,
                                    const double* rot1, const double* rot2, const double* rot3 = nullptr, bool unit_order = false);
int launch_dct_pair_prep16_inv_cols(hipStream_t st, const float* in, size_t n_frames, size_t w, size_t h, double* base,
                                    const double* rot1, const double* rot2, bool class_major = false, const double* rot3 = nullptr);
int launch_dct_pair_prep16_cols(hipStream_t st, const float* in, size_t n_frames, size_t w, size_t h, double* base,
                                const double* rot1, const double* rot2, bool class_major = false, const double* rot3 = nullptr);
size_t dct_pair_deep_elems(size_t lines, size_t len);
int launch_dct_pair_prep16_rows(hipStream_t st, int src_kind, const void* src, size_t n_frames, size_t w, size_t h, double* base,
                                const double* rot1, const double* rot2, const double* rot3, float* ip, float* qp, bool unit_order = false);
// r5, fused forward transform: units per frame of the row pass's line order (H/16 rounded up to whole k-blocks of 8) and
// whether a transform of n frames takes it (dct_pair_f64.hip)
inline size_t dct_pair_fused_units(size_t h) { return ((h / 16 + 7) / 8) * 8; }
bool dct_pair_can_fuse_cols(size_t n_frames, size_t w, size_t h);
bool dct_pair_can_fuse_inv_cols(size_t n_frames, size_t w, size_t h);

int launch_dct_pair_gemm_rows_subset_f32(hipStream_t st, const float* x, const float* y, unsigned cap, unsigned Kp, float* out,
                                         unsigned out_stride, unsigned off, size_t lines);

// prune.hip: the derived frame's transform restricted to the frequency columns a chunk's index lists use
// v is in the class when v % mod == rem, or == rem2 (classes of the split odd half: there the output is the cosine
// part MINUS the sine part and the gathered sine row is negated); basis row (v + radd) / mod
constexpr unsigned PRUNE_NO_REM = 0xFFFFFFFFu;
constexpr uint32_t PRUNE_NEG = 0x40000000u;              // rows[] flag: negate the gathered sine row
constexpr unsigned SSW_PRUNE_INFO = 16;                  // u32 words of a chunk's info block: [0] overflow flag, [1 + c] members of class c
struct PruneClass { unsigned mod, rem, cap, off, rem2 = PRUNE_NO_REM, radd = 0; };
struct PrunePlan {
    unsigned n_classes = 0;
    PruneClass c[9];        // info[] of launch_prune_build: SSW_PRUNE_INFO words, 1 + one per class
    unsigned W = 0, cap_total = 0;
};
static_assert(SSW_PRUNE_INFO >= 1 + sizeof(PrunePlan::c) / sizeof(PruneClass), "a chunk's info block holds the flag and one count per class");
int launch_prune_build(hipStream_t st, const uint32_t* idx, size_t n_frames, size_t k, const PrunePlan& plan,
                       uint32_t* flag /*[W]*/, uint32_t* rows /*[cap_total]*/, uint32_t* pos /*[W]*/, uint32_t* info /*[SSW_PRUNE_INFO]*/);
// one gathered basis: rows[] of `src` ([kblocks][src_rows] 64-byte pieces) into `dst` ([kblocks][cap]); `negate`: rows flagged
// PRUNE_NEG change sign (f64 sine rows)
struct PruneGatherJob {
    const uint32_t* rows;
    const char* src;
    char* dst;
    unsigned cap, src_rows, kblocks, first_block;
    bool negate;
    bool frag = false;         // MFMA B-fragment order for the fused derived pass (whole tiles of 16 rows)
};
struct PruneGatherJobs {
    PruneGatherJob j[18];
    unsigned n;
};
int launch_prune_gather_bases(hipStream_t st, PruneGatherJobs jobs);
int launch_extract_pruned(hipStream_t st, const float* base, const float* compact, size_t n_frames, size_t w, size_t h,
                          size_t cap, const uint32_t* pos, const uint32_t* indices, size_t k, int method, float alpha,
                          float* out);

// select.hip
struct SelectWorkspace {
    uint32_t* hist = nullptr;       // [n_frames][2048] sample histogram
    uint32_t* ctrl = nullptr;       // [n_frames][4]: threshold digit, candidate count, 2 spare
    uint64_t* cand = nullptr;       // [n_frames][cap] candidate composite keys
    size_t frames = 0, cap = 0;
    uint32_t* fallbacks = nullptr;  // (the context's) counter of frames whose finish ran the exact whole-plane select
};
size_t select_cand_capacity(size_t k);   // candidate slots per frame needed for mark length k
size_t select_max_k();
int launch_topk(hipStream_t st, const float* coef, size_t n_frames, size_t w, size_t h, int ordering,
                size_t k, const SelectWorkspace& ws, uint32_t* indices);
// attack.hip: 8-bit boundary + CatmullRom resize (third-party `image` crate semantics)
struct ResizeTaps {                 // host side
    uint32_t max_taps = 0;
    std::vector<uint32_t> left, count;
    std::vector<float> weights;     // [out_len][max_taps]
};
struct DeviceTaps {
    uint32_t* left = nullptr;
    uint32_t* count = nullptr;
    float* weights = nullptr;
    uint32_t max_taps = 0;
    uint32_t span[8] = {0};     // reach in input samples of 1, 2, 4 .. 128 consecutive outputs (host-side, for tiling)
    bool quad_uniform = false;  // every aligned group of 4 outputs shares left and count, count <= 5 (integer up-scaling)
};
void build_resize_taps(size_t in_len, size_t out_len, ResizeTaps& t);
int launch_u8_to_f32(hipStream_t st, const uint8_t* in, size_t n, float* out);
int launch_f32_to_u8(hipStream_t st, const float* in, size_t n, uint8_t* out);
int launch_rgb8_to_yiq(hipStream_t st, const uint8_t* rgb, size_t npix, float* y, float* i, float* q);
int launch_u16_to_f32(hipStream_t st, const uint16_t* in, size_t n, float* out);
int launch_f32_to_u16(hipStream_t st, const float* in, size_t n, uint16_t* out);
int launch_rgb16_to_yiq(hipStream_t st, const uint16_t* rgb, size_t npix, float* y, float* i, float* q);
int launch_yiq_to_rgb8(hipStream_t st, const float* y, const float* i, const float* q, size_t npix, uint8_t* rgb);
size_t resize_tmp_bytes(const uint8_t* in, size_t n_frames, size_t w, size_t h, size_t nw, size_t nh, const DeviceTaps& vt,
                        const DeviceTaps& ht, const uint8_t* out);
int launch_resize_rgb8(hipStream_t st, const uint8_t* in, size_t n_frames, size_t w, size_t h, size_t nw, size_t nh,
                       const DeviceTaps& vt, const DeviceTaps& ht, float* tmp, uint8_t* out);

// sort_full.hip: the first k of all W*H-1 indices of n_frames planes in the reference's order (batched LSD radix
// sort; not hot: only Reader::indices() with a large k and marks longer than the top-k limit reach it)
int full_sort_scratch_bytes(size_t plane_len, size_t n_frames, size_t* bytes);
int launch_full_sort(hipStream_t st, const float* coef, size_t n_frames, size_t w, size_t h, int ordering, void* scratch,
                     size_t scratch_bytes, uint32_t* indices_out, size_t k);
int launch_embed(hipStream_t st, float* coef, size_t n_frames, size_t plane_len,
                 const uint32_t* indices, size_t idx_stride, const float* marks,
                 const uint32_t* mark_offsets, const uint32_t* mark_lens, size_t n_marks,
                 size_t max_len, size_t mark_stride /* between marks when mark_offsets is null */, int method, float alpha);
int launch_extract(hipStream_t st, const float* base, const float* derived, size_t n_frames,
                   size_t plane_len, const uint32_t* indices, size_t k, int method, float alpha,
                   float* out);
int launch_similarity(hipStream_t st, const float* extracted, const float* marks, size_t n_pairs,
                      size_t k, float* sims);
int launch_gemm_nt_f32(hipStream_t st, const float* A, size_t M, const float* B, size_t N, size_t K, float* out);
int launch_sim_den(hipStream_t st, const float* extracted, size_t n_ext, size_t k, float* den);
int launch_sim_scale(hipStream_t st, float* sims, const float* den, size_t n_ext, size_t n_marks);
int launch_widen_indices(hipStream_t st, const uint32_t* in, size_t n, uint64_t* out);

}  // namespace ssw

namespace ssw { namespace host { struct Transfer; } }

// ---- context ----------------------------------------------------------------
struct ssw_ctx {
    int device = 0;
    hipStream_t stream = nullptr;       // where every call enqueues: own_stream, or the caller's (ssw_ctx_set_stream)
    hipStream_t own_stream = nullptr;   // private non-blocking stream created with the context
    size_t chunk_frames = 0;      // frames per internal pass; 0 = automatic (~2^28 pixels)

    // basis cache: (N, inverse, f64, kind) -> device pointer; kind 0 = dense N x N,
    // 1 / 2 = even / odd half basis (N/2 x N/2) of the folded kernels; 3 / 4 = the same, k-blocked (operand-ready GEMMs)
    std::map<std::tuple<size_t, bool, bool, int>, void*> basis;
    bool fold = true;             // use the even/odd-folded GEMMs where the shape allows
    int fold_level = SSW_DCT_FOLDING_DEFAULT;           // 1 / 2: one folding level inside the GEMM kernel (dct_folded*.hip)
                                  // 3: operand-ready GEMMs (dct_pair_*.hip); 4: two levels; 5: + a third on long forward row passes; 6: on all

    // growable scratch
    struct Buf {
        void* p = nullptr;
        size_t bytes = 0;
    };
    // A lane = the workspace of one chunk in flight.  The batch entry points keep two chunks in flight
    // (ssw_ctx_set_overlap): while one lane's basis GEMMs run on the context's stream, the other lane's
    // HBM-bound stages (operand pre-passes, selection, colour conversion) run on `aux_stream`.
    struct Lane {
        Buf plane[4];             // y / i / q / t planes of the chunk
        Buf operand[6];           // operand planes of the operand-ready GEMMs: S|E, D|O, SS|EE, SD|EO, T, split odd (AS|BD|AD|BS)
        Buf idx;                  // [chunk][k] u32
        ssw::SelectWorkspace sel;
        Buf compact[2];           // pruned derived transform: row-pass result, column-pass result [chunk][H][cap]
        Buf gathered;             // gathered half bases of the chunk's frequency classes
        Buf prune_u32;            // flag [W] | pos [W] | rows [cap] | info [8]
        hipStream_t cur = nullptr;   // stream the lane's chain currently runs on
        hipEvent_t done = nullptr;   // recorded right after the lane's latest stage (borrowed from sync_events)
    };
    static constexpr int MAX_LANES = 2;
    Lane lane[MAX_LANES];
    hipStream_t aux_stream = nullptr;     // HBM-bound stages of the batch pipelines
    bool overlap = true;                  // two lanes / two streams in the batch entry points
    bool prune = true;                    // batch extract: derived transform only where the index lists need it
    bool split = true;                    // f64 GEMMs: odd halves as rotated quarter-length cosine + sine pairs
    std::vector<hipEvent_t> sync_events;  // cross-stream dependencies (ring)
    size_t sync_next = 0;
    Buf overflow;                         // [chunks] u32 flags of the pruned path (+ class counts)
    uint64_t pruned_chunks = 0, redone_chunks = 0;
    uint64_t pruned_columns = 0;          // sum over pruned chunks of the compact plane width (cap_total)
    // single-image handles (ssw_lib.hip): frames cross PCIe on `copy_stream` into / out of two alternating
    // device staging buffers, so the upload of the next frame runs beside the kernels of the previous one
    hipStream_t copy_stream = nullptr;
    struct FrameStage {
        Buf buf;
        static constexpr int MAX_BANDS = 4;
        hipEvent_t uploaded[MAX_BANDS] = {nullptr, nullptr, nullptr, nullptr};   // copy_stream: band b of the frame (the whole frame: [0]) is in the buffer
        hipEvent_t consumed = nullptr;    // stream: the last kernel that reads (or writes) the buffer is enqueued before
        bool in_use = false;              // `consumed` has been recorded at least once
    };
    FrameStage frame_stage[2];
    unsigned frame_stage_next = 0;
    size_t expected_k = 0;                // mark length of the context's last Reader::extract / indices (speculative selection of the next base reader)
    // device planes of destroyed handles, kept for the next handle of the same size (hipMalloc + hipFree of
    // three 4K planes cost more than the transform); all reuse is ordered on the context's stream
    std::multimap<size_t, void*> plane_pool;
    size_t plane_pool_bytes = 0;
    // RGB staging buffers of derived readers (filled on `copy_stream`, read on `stream`): kept apart with the event that
    // says when their last reader was done, so that the next derived frame's upload waits for THAT and not for everything
    // the context's stream holds (a base reader's transform queued a moment ago)
    struct RgbSpare { void* p; size_t bytes; hipEvent_t released; };
    std::vector<RgbSpare> rgb_spares;
    // workspace buffers that grew while a chain of stages was being built: stages built earlier captured the old pointer, so
    // the old allocation stays valid until the next entry into the library frees it (grow(), CtxGuard; r5)
    std::vector<void*> retired;
    uint32_t* select_fallbacks = nullptr; // device counter: frames whose candidate count fell outside [k, capacity]
    uint64_t select_frames = 0;           // frames selected since the last ssw_ctx_reset_timing
    ssw::host::Transfer* xfer = nullptr;  // pinned staging ring + copy threads (transfer.hip)
    // host-image streaming entry points (ssw_stream.hip): groups of frames cross PCIe on `copy_stream` (up) and
    // `down_stream` into / out of two alternating device buffers while the group between them is being computed
    hipStream_t down_stream = nullptr;
    struct HostStream {
        static constexpr int NB = 3;      // ring depth
        Buf in[NB], in2[NB], out[NB];     // frames of a group: input (base), second input (derived), output
        Buf marks, ext, sims;             // the whole call's marks / extracted marks / similarities
        hipEvent_t up_done[NB] = {}, k_done[NB] = {}, down_done[NB] = {};
        bool active = false;              // a streaming call is running: the ring must stay (dev_malloc's give-back skips it)
    };
    HostStream hs;
    Buf small;                    // misc (mark offsets, sims, ...)
    Buf sort_scratch;             // full-order sort (lazy, Reader::indices beyond the top-k limit)
    Buf resize_tmp;               // f32 intermediate of the resize's vertical pass
    std::map<std::pair<size_t, size_t>, ssw::DeviceTaps> taps;   // (in_len, out_len) -> filter taps

    // timing
    bool timing = false;
    double stage_ms[SSW_STAGE_COUNT] = {0};
    uint64_t stage_launches[SSW_STAGE_COUNT] = {0};
    double stage_bytes[SSW_STAGE_COUNT] = {0};    // algorithmic HBM bytes of every stage (ssw_ctx_get_traffic)
    double stage_work[SSW_STAGE_COUNT] = {0};     // executed flop (GEMM stages) / algorithmic bytes (HBM-bound stages)
    struct Pending {
        int stage;
        hipEvent_t a, b;
        int alias = -1;
    };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> free_events;
    hipEvent_t tail_event = nullptr;      // last event a stage timer recorded; shared with the next timer while tail_fresh
    hipStream_t tail_stream = nullptr;
    bool tail_fresh = false;
};
