// Operand-ready even/odd-folded basis GEMMs in f64 (canonical precision).
//
// Measured on MI355X (tools/mfma_peak.hip): v_mfma_f64_16x16x4_f64 sustains 77.6 TFLOP/s when the
// wave issues nothing else, but every VALU instruction issued next to it takes MFMA pipe time --
// ~6 cycles for a 32-bit op, ~11.5 cycles for v_cvt_f64_f32 / v_add_f64 -- even from another wave
// of the same SIMD.  The in-kernel folding of dct_folded_f64.hip spends one f64 VALU op per MFMA
// (widen + add/subtract after the LDS read) and tops out at 81 % of peak for that reason.
//
// Here the GEMM main loop contains no VALU instruction at all: global_load -> ds_write ->
// ds_read -> MFMA, with scalar address arithmetic.  Its operands are produced once per pass by
// HBM-bound pre-passes (dct_pair_prep.hip) in exactly the form the MFMA consumes:
//   forward:  S[s] = (double)x[s] + (double)x[N-1-s],  D[s] = (double)x[s] - (double)x[N-1-s]
//   inverse:  E[s] = (double)c[2s],                    O[s] = (double)c[2s+1]
// stored k-blocked: [Kp / 8][lines][8] doubles (zero padded to Kp), i.e. the 64-byte piece of every
// line that one k-step needs lies next to its neighbours' -- a block tile's k-step is ONE contiguous
// 8 KB read (whole 128-byte lines, one DRAM page) instead of 128 pieces 15 KB apart.  The half bases
// are cached in the same layout.  For the column pass the pre-pass also transposes, so that one
// "NT" kernel serves all four passes:
//   acc1[x][y] = sum_k X1[x][k] Y1[y][k],   acc2[x][y] = sum_k X2[x][k] Y2[y][k]
// with X = image operand (lines), Y = half basis (pairs).  Epilogues as in dct_folded_f64.hip:
// forward interleaves (even, odd) frequencies; inverse forms acc1 +/- acc2 for the mirrored
// positions; results are rounded once to f32 (then the reference's f32 scale factor, if any).
//
// r3: the odd halves are split once more (dct_pair_prep.hip, "Split odd half"): their launches feed this kernel two
// DIFFERENT image operands (the rotated and folded AS | BD or AD | BS) against quarter-length cosine / sine bases and
// the epilogues emit acc1 +/- acc2 (po.pm); a deep inverse adds EPI_INV_OT (the half-length even half E from its own
// even half T2 and odd part).  Row passes store through buffer instructions with per-tile lane offsets (epilogue notes
// below); deep transforms keep the plane between their passes class-major (dct_pair_common.hpp).
//
// Block: 256 threads = 4 waves as 2 x 2; block tile 128 lines x 64 pairs x 2 products; k-step 8;
// per wave 16 MFMA 16x16 tiles = 128 accumulator registers; LDS 48 KB double-buffered (XOR-swizzled
// 64-byte rows, conflict-free ds_read_b64 / ds_read2_b64), one barrier per k-step, 2 blocks per CU.
// Lane l: li = l & 15 (line / pair inside a 16x16 tile), lq = l >> 4: in half-step s lane group lq
// supplies k = 4 s + lq (the same assignment on both operands).
#include "dct_pair_common.hpp"

#include <cstdlib>
#include <type_traits>

#ifdef SSW_TILE_TRACE
namespace ssw {
// diagnostic build only: per block (thread 0) the 100 MHz wall clock at entry, after the first tile is staged, after
// the main loop and after the epilogue, plus the hardware id (XCC / SE / CU) -- tools/tile_trace.py
__device__ unsigned long long* g_tile_trace = nullptr;
__device__ unsigned int g_tile_trace_cap = 0;
__device__ unsigned int g_tile_trace_n = 0;
__device__ unsigned int* g_tile_kstep = nullptr;        // [cap][32] per-k-step stamps (tools/tile_trace.py --ksteps)
extern "C" int ssw_debug_set_tile_kstep(void* dev_ptr) {
    unsigned int* p = static_cast<unsigned int*>(dev_ptr);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tile_kstep), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
extern "C" int ssw_debug_set_tile_trace(void* dev_ptr, unsigned cap) {
    unsigned long long* p = static_cast<unsigned long long*>(dev_ptr);
    unsigned zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tile_trace), &p, sizeof(p)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tile_trace_cap), &cap, sizeof(cap)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tile_trace_n), &zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
extern "C" int ssw_debug_get_tile_trace_count(unsigned* n) {
    return hipMemcpyFromSymbol(n, HIP_SYMBOL(g_tile_trace_n), sizeof(*n)) == hipSuccess ? 0 : -1;
}
}  // namespace ssw
#endif
#include "dct_pair_f64_kernel.hpp"
#if defined(SSW_TILE_TRACE) && !defined(SSW_TILE_TRACE_FWD_ONLY)
#define SSW_INV_PART -1
#include "dct_pair_f64_inv.inc"      // the diagnostic build keeps one unit (one set of trace globals)
#endif                               // (-DSSW_TILE_TRACE_FWD_ONLY: only the forward instances trace; the inverse ones come from the normal objects: 3 min instead of 9)

namespace ssw {


// One launch of the operand-ready GEMM.  `kind` selects the epilogue:
//   0  one folding level (forward: interleave even/odd; inverse: mirror)            pairs = len/2, K = len/2
//   1  level 2, even half: X = (SS, SD) | (EE, EO), Y = half bases of len/2          pairs = len/4, K = len/4
//   2  level 2, odd half:  X = D | O (shared), Y = the two halves of the odd basis   pairs = len/4, K = len/2
//   3 / 4  the odd half split once more (see "Split odd half" in dct_pair_prep.hip): X = (AS, BD) | (AD, BS), the rotated
//      and folded odd operand; Y = (cosine, sine) rows 2i | 2i+1 of the quarter-length bases; outputs acc1 +/- acc2
//      pairs = len/8 + 1 | len/8, K = len/8
// x*, y*: k-blocked planes (y2 of kind 2 = y1 + 8 * len/4: the second row block of the same plane).
// sub (forward only): the launch belongs to the transform of length len >> sub that a deeper folding level
// applies to the even part (its frequencies are multiples of 2^sub of the full transform's).
namespace {

// the per-class part of a launch: pair count, sum length, basis lines and the output map of (kind, sub)
int pair_class_args(const PairClassDesc& d, bool is_row, bool inverse, size_t len, bool class_major, bool with_sink, bool has_tmp_out,
                    PairClassArgs& ca, PairInstance& inst) {
    const int kind = d.kind, sub = d.sub;
    // inverse: sub = 1 serves the deep inverse (the half-length transform E): kind 1 -> its even half T2, kinds 3 / 4 -> E
    if (kind < 0 || kind > 9 || sub < 0 || sub > 8 || (sub > 0 && kind == 0)) return SSW_ERR_BAD_ARG;
    // inverse: sub = 2 (level 2) serves the quarter-length even part T2: kind 1 -> ITS even half, kind 9 -> its odd part + combine
    if (inverse && sub > 1 && !(sub == 2 && (kind == 1 || kind == 9))) return SSW_ERR_BAD_ARG;
    // kinds 5 / 6 (forward): class E (kind 3) folded once more -- the even / odd rows of its bases, operands AS+ | BD- and
    // AS- | BD+ of half the length: the arithmetic of kinds 3 / 4 on a transform of half the length, frequencies 16i +/- 1
    // and 16i + 8 +/- 1 (dct_pair_common.hpp, ForwardClassLayout)
    // kinds 7 / 8 / 9 (forward, level 2): launches of class E's shape on the bases of kind 5 -- class O rotated once more
    // (operands (a, b) of AD plus / minus those of the reversed BS: frequencies 16i +/- 5, 16i +/- 3) and R2 rotated (16i +/- 4)
    const bool esplit = kind >= 5;
    const bool eshape = kind == 3 || kind == 5 || kind >= 7;       // pairs n/8 + 1 in n/8 slots (fold0), sine basis = its launch variant
    if (esplit && sub != ((inverse && kind == 9) ? 2 : 0)) return SSW_ERR_BAD_ARG;
    const bool split = kind == 3 || kind == 4 || esplit;
    if (inverse && sub >= 1 && kind != 1 && !has_tmp_out) return SSW_ERR_BAD_ARG;      // the odd part of E needs somewhere to put E
    const size_t leff = esplit ? len / 2 : (len >> sub);      // length of the (sub-)transform this class serves (level 2: of its bases)
    const unsigned fs = 1u << sub;                            // its frequencies in units of the full transform's
    if (split && leff % 8 != 0) return SSW_ERR_BAD_ARG;
    ca.x1 = d.x1; ca.x2 = d.x2; ca.y1 = d.y1; ca.y2 = d.y2;
    ca.NP = (unsigned)(kind == 0 ? leff / 2 : split ? leff / 8 : leff / 4);      // class E: n/8 + 1 pairs in n/8 slots (fold0)
    ca.Kp = (unsigned)(split ? pair_kpad<double>(leff / 4) : kind == 1 ? pair_kpad<double>(leff / 2) : pair_kpad<double>(leff));
    ca.yrows = kind == 2 ? 2 * ca.NP : eshape ? ca.NP + 1 : ca.NP;      // lines of the basis plane(s): class E's keep row n/8
#ifdef SSW_ABL_NP128        // timing-only ablation: the 135-pair column classes without their 7-pair tail tile
    if (!is_row && ca.NP == 135) ca.NP = 128;
#endif
    ca.tiles_n = (ca.NP + 63) / 64;
    ca.c1 = 0; ca.c2 = 1; ca.cs = 2; ca.pm = 0; ca.np1 = 0xFFFFFFFFu; ca.p2lo = 0; ca.bn32 = 0; ca.fold0 = 0;
    // 48-pair tiles where 64-pair ones would end in a tile of at most 16 pairs and 48 need no more tiles (135 = 48 + 48 + 39
    // instead of 64 + 64 + 7: 4K and 1080p columns); `tile48` = 0: A/B switch
    if (tuning(TUNE_TILE48) != 0 && ca.NP > 64 && (ca.NP % 64) != 0 && (ca.NP % 64) <= 16 && (ca.NP + 47) / 48 == ca.tiles_n) {
        ca.bn32 = 2;
        ca.tiles_n = (ca.NP + 47) / 48;
    }
    ca.gsh = 31; ca.e2off = 0;
    if (kind == 1) { ca.c1 = 0; ca.c2 = 2 * fs; ca.cs = 4 * fs; }
    if (kind == 2) { ca.c1 = fs; ca.c2 = fs + 2 * fs * ca.NP; ca.cs = 2 * fs; }
    if (inverse && kind == 2) { ca.c1 = 0; ca.c2 = (unsigned)(leff / 4); ca.cs = 1; }      // positions pair, pair + n/4 of the odd part
    if (split) {
        // odd frequency u = 2k+1 of the (sub-)transform; class E (kind 3) pair i: k = 4i (+), 4i-1 (-); class O: k = 4i+2 (+), 4i+1 (-)
        ca.pm = 1;
        if (eshape) ca.fold0 = (unsigned)(leff / 8);       // y2 must be the launch variant of the sine basis (row 0 = row n/8)
        if (esplit && inverse) {
            // positions of the odd part: frequency u = 2 k + 1 of the forward map below -> k; kind 9: of the quarter-length
            // even part's odd half, 2 j and 2 j - 1
            const unsigned p1[5] = {0u, 4u, 2u, 1u, 0u}, p2v[5] = {0u - 1u, 3u, 0u - 3u, 0u - 2u, 0u - 1u};
            ca.c1 = p1[kind - 5]; ca.c2 = p2v[kind - 5]; ca.cs = kind == 9 ? 2 : 8;
        }
        else if (kind == 6) { ca.c1 = 9u; ca.c2 = 7u; ca.cs = 16; }
        else if (esplit) { const unsigned r = kind == 5 ? 1u : kind == 7 ? 5u : kind == 8 ? 3u : 4u; ca.c1 = r; ca.c2 = 0u - r; ca.cs = 16; }
        else if (!inverse) {
            ca.c1 = (kind == 3 ? 1u : 5u) * fs; ca.c2 = kind == 3 ? 0u - fs : 3u * fs; ca.cs = 8 * fs;
        } else {
            ca.c1 = kind == 3 ? 0u : 2u; ca.c2 = kind == 3 ? 0u - 1u : 1u; ca.cs = 4;             // positions of the odd part
        }
    }
    if (class_major) {
        // forward row pass of a deep transform: every class writes its frequencies side by side (ForwardClassLayout,
        // dct_pair_common.hpp) instead of 4-byte pieces 16 / 32 bytes apart -- the column pre-pass puts the columns back;
        // inverse: the split classes write (and read E) at one pair of residues mod 4 (po.cm, inverse_class_pos), the
        // quarter-length even half T2 (kind 1) keeps the natural order
        if (!is_row || !((kind == 1 && sub >= 1) || split) || sub > 2 || (sub == 2 && kind != 1 && !(inverse && kind == 9))) return SSW_ERR_BAD_ARG;
        if (!inverse) {
            // po.ft = the tile: entry e of a class -> column base + (e >> gsh) * ft + (e & (2^gsh - 1)); class E's second
            // output of pair p is entry p - 1 of its "-" class (frequency 8 p - 1)
            // level 2 (dct_pair_efold(len), like the pre-pass): sixteen classes, every launch a sum of len/16 terms
            const bool l2 = dct_pair_efold(len);
            const ForwardClassLayout fl{(unsigned)len, dct_pair_class_tile(len), l2};
            typedef ForwardClassLayout F;
            int k1 = -1;
            if (l2) k1 = (kind == 1 && sub == 2) ? F::R1A : (kind == 3 && sub == 1) ? F::F_E2P : (kind == 4 && sub == 1) ? F::F_O2P
                       : kind == 5 ? F::EEP : kind == 6 ? F::EOP : kind == 7 ? F::O5 : kind == 8 ? F::O3 : kind == 9 ? F::R2A : -1;
            else k1 = (kind == 1 && sub == 1) ? F::R1 : kind == 3 ? (sub ? F::E2P : F::EP) : kind == 4 ? (sub ? F::O2P : F::OP) : -1;
            if (k1 < 0) return SSW_ERR_BAD_ARG;
            ca.cs = 1;
            ca.c1 = fl.base(k1); ca.c2 = fl.base(k1 + 1);
            ca.e2off = eshape ? 1u : 0u;
            ca.gsh = 31;
            if (fl.t != fl.n) {
                const unsigned g = fl.group(k1);
                if (g == 0 || (g & (g - 1)) != 0) return SSW_ERR_BAD_ARG;
                ca.gsh = 0;
                while ((1u << ca.gsh) < g) ++ca.gsh;
            }
        }
    }
    // template instance
    if (!inverse) {
        if (kind == 0) inst = {is_row ? EPI_FWD_ADJ : EPI_FWD, false, 0};
        else if (kind == 1) inst = {EPI_FWD, false, sub ? 1 : 0};
        else if (kind == 2) inst = {EPI_FWD, true, sub ? 1 : 0};
        else inst = {EPI_FWD, false, (is_row && (kind == 7 || (kind == 4 && sub == 0))) ? 4 : 3};      // 4: class O of the full-length split (level 2: its "+" launch), the launch bench.py prices
    } else {
        if (kind == 0) inst = {EPI_INV, false, 0};
        else if (kind == 1) inst = {EPI_INV_E, false, sub ? 1 : 0};
        else if (with_sink) inst = {EPI_INV_O_RGB, !split, 0};
        else if (sub >= 1) inst = {EPI_INV_OT, !split, 1};          // kinds 2 (semi-deep: one shared operand), 3, 4; 9 (level 2)
        else inst = {EPI_INV_O, !split, 0};
    }
    return SSW_OK;
}
}  // namespace

// One launch over `n_classes` classes (see PairMulti).  A class is described like the single launches:
//   kind 0  one folding level (forward: interleave even/odd; inverse: mirror)            pairs = len/2, K = len/2
//        1  level 2, even half: X = (SS, SD) | (EE, EO), Y = half bases of len/2          pairs = len/4, K = len/4
//        2  level 2, odd half:  X = D | O (shared), Y = the two halves of the odd basis   pairs = len/4, K = len/2
//    3 / 4  the odd half split once more (dct_pair_prep.hip "Split odd half"): X = (AS, BD) | (AD, BS), Y = (cosine,
//           sine) rows 2i | 2i+1 of the quarter-length bases; outputs acc1 +/- acc2          pairs = len/8, K = len/8
//           (class E's first and last pair share slot 0: its y2 is the launch variant of the sine basis, row 0 = row len/8)
//   sub: the class belongs to the transform of length len >> sub that a deeper folding level applies to the even part.
// r5: a forward transform of n frames whose row launches write the column operands themselves (EPI_FWD_COLOP).  Rows first,
// both passes at level 2, 128-frequency class tiles, the column planes exactly as wide as the padded units, and both passes
// on 128-line tiles (a tile of the row pass IS one k-block of the column operands; small batches keep the unfused path).
// (the launcher's own test, launch_dct_pair_gemm_multi_f64: a pass of at most merge_max_lines lines runs its eight classes as
// ONE launch, whose tile columns add up)
static bool launch_is_small(size_t lines, size_t pairs) {
    const unsigned long long classes = lines <= (size_t)tuning(TUNE_MERGE_MAX_LINES) ? 8 : 1;
    return (unsigned long long)((lines + 127) / 128) * ((pairs + 63) / 64) * classes < 448;
}
bool dct_pair_can_fuse_cols(size_t n_frames, size_t w, size_t h) {
    if (tuning(TUNE_FUSE_COLS) == 0 || n_frames == 0 || w < h || w % 128 != 0 || h % 16 != 0) return false;
    if (dct_pair_class_tile(w) != 128 || !dct_pair_efold(w) || !dct_pair_efold_cols(h, w, true)) return false;
    // ... and the column pass must reach its deep branch by itself (build_pass_impl: two && split && deep): with the public
    // thresholds lowered (ssw_tuning_set: deep_min_cols, efold_cols_min) below 128 rows it would not, and would read an f32
    // plane the fused row pass never wrote
    if (!dct_pair_can_split(h, false) || !dct_pair_can_fold2_cols(h)) return false;
    const size_t hup = dct_pair_fused_units(h);
    if (pair_kpad<double>(h / 8) != hup) return false;
    if (n_frames * 16 * hup > 0xFFFFFFFFull || n_frames * w > 0xFFFFFFFFull) return false;      // 32-bit line indices in both passes
    return !launch_is_small(n_frames * 16 * hup, w / 16) && !launch_is_small(n_frames * w, h / 16);
}
// ... and the inverse transform of such frames (deep inverse rows need whole 128-column tiles anyway)
// (measured at 4K, r5: the pre-pass it removes takes 2.6 ms per 128 frames, the four launches that take over its work grow by
// 2.7 ms -- four output positions per pair instead of the forward launches' two frequencies, i.e. twice the operand bytes
// per tile in 16-byte pieces and a second read of E: bit-identical, 8 B/px less HBM traffic, no faster.  Off by default.)
bool dct_pair_can_fuse_inv_cols(size_t n_frames, size_t w, size_t h) {
    if (tuning(TUNE_FUSE_INV_COLS) == 0 || n_frames == 0 || w < h || w % 128 != 0 || h % 16 != 0) return false;
    if (dct_pair_class_tile(w) != 128 || !dct_pair_efold_inv(w) || !dct_pair_efold_cols(h, w, true)) return false;
    if (!dct_pair_can_split(h, false) || !dct_pair_can_fold2_cols(h)) return false;      // as above
    const size_t hup = dct_pair_fused_units(h);
    if (pair_kpad<double>(h / 8) != hup) return false;
    if (n_frames * 16 * hup > 0xFFFFFFFFull || n_frames * w > 0xFFFFFFFFull) return false;
    // (the inverse pass's dependent launches run one class each: every one must fill 128-line tiles by itself)
    auto small1 = [](size_t lines, size_t pairs) { return (unsigned long long)((lines + 127) / 128) * ((pairs + 63) / 64) < 448; };
    return !small1(n_frames * 16 * hup, w / 16) && !small1(n_frames * w, h / 16);
}

int launch_dct_pair_gemm_multi_f64(hipStream_t st, bool is_row, bool inverse, int n_classes, const PairClassDesc* desc, float* out,
                                   double* tmp, size_t n_frames, size_t w, size_t h, Epilogue ep, const RgbSink* sink, double* tmp_out,
                                   bool class_major, const FuseCols* fuse) {
    if (n_frames == 0 || n_classes == 0) return SSW_OK;
    if (n_classes < 0 || n_classes > 8 || !desc) return SSW_ERR_BAD_ARG;
    if (w > 0xFFFFFFull || h > 0xFFFFFFull) return SSW_ERR_BAD_DIMS;
    // fused forward transform: the row launches (fuse->cop set) run over the unit-ordered, padded lines and write the
    // column operands; the column launches (fuse set, cop null) read their tiles in the row launches' class-major order
    const bool fuse_cop = fuse && fuse->mode == FUSE_ROWS_COP, fuse_rows = fuse_cop || (fuse && fuse->mode == FUSE_ROWS_LINES);
    const bool fuse_cols = fuse && fuse->mode == FUSE_COLS;
    if (fuse && (!(fuse_rows || fuse_cols) || fuse_rows != is_row ||
                 !(inverse ? dct_pair_can_fuse_inv_cols(n_frames, w, h) : dct_pair_can_fuse_cols(n_frames, w, h)))) return SSW_ERR_BAD_ARG;
    if (fuse_cop && (!class_major || sink || !fuse->cop || !fuse->rot1 || !fuse->rot2 || !fuse->rot3)) return SSW_ERR_BAD_ARG;
    const size_t lines = fuse_rows ? n_frames * 16 * dct_pair_fused_units(h) : is_row ? n_frames * h : n_frames * w;
    const size_t len = is_row ? w : h;
    if (lines > 0xFFFFFFFFull) return SSW_ERR_BAD_DIMS;
    const unsigned L = (unsigned)lines;
    const bool with_sink = sink && sink->rgb;
    PairMulti ml;
    PairInstance inst{0, false, 0};
    unsigned tiles_n = 0, leff0 = 0;
    for (int c = 0; c < n_classes; ++c) {
        PairInstance ic{0, false, 0};
        SSW_TRY(pair_class_args(desc[c], is_row, inverse, len, class_major, with_sink, tmp_out != nullptr, ml.c[c], ic));
        if (c == 0) { inst = ic; leff0 = (unsigned)(len >> desc[c].sub); }
        else if (!(ic == inst) && !(ic.epi == inst.epi && ic.samex == inst.samex && n_classes > 1)) return SSW_ERR_BAD_ARG;
        if (inverse && (unsigned)(len >> desc[c].sub) != leff0) return SSW_ERR_BAD_ARG;      // po.n is shared
        if ((unsigned long long)ml.c[c].Kp * L * 8 > 0xFFFFFFFFull) return SSW_ERR_BAD_DIMS;   // scalar k-block offsets are 32-bit
        tiles_n += ml.c[c].tiles_n;
    }
    if (n_classes > 1) inst.subname = inverse ? inst.subname : 3;
    // 64-line tiles when 128-line ones would not fill the 512 block slots of the chip (2 per CU)
    const bool small = (unsigned long long)((L + 127) / 128) * tiles_n < 448;
    if (fuse && small) return SSW_ERR_BAD_ARG;
    const unsigned BM = small ? 64 : 128;
    if (small)                                                     // 48-pair tiles are three pair tiles of a 128-line block
        for (int c = 0; c < n_classes; ++c)
            if (ml.c[c].bn32 == 2) { tiles_n -= ml.c[c].tiles_n; ml.c[c].bn32 = 0; ml.c[c].tiles_n = (ml.c[c].NP + 63) / 64; tiles_n += ml.c[c].tiles_n; }
    const unsigned tiles_m = (L + BM - 1) / BM;
    // ... and 32-pair tiles when that spreads such a (single-class) launch more evenly over the 256 CUs (all its blocks are
    // resident at once, so a launch takes as long as the fullest CU): balance = blocks / (256 * ceil(blocks / 256)); the
    // smaller tiles reuse their basis fragments less, hence the 8 % handicap
    if (small && n_classes == 1) {
        const int force = (int)tuning(TUNE_BN32);
        auto balance = [](unsigned long long n) { return (double)n / (256.0 * (double)((n + 255) / 256)); };
        const unsigned tn32 = (ml.c[0].NP + 31) / 32;
        const bool bn32 = force >= 0 ? force != 0 : 0.92 * balance((unsigned long long)tiles_m * tn32) > balance((unsigned long long)tiles_m * tiles_n);
        if (bn32) { ml.c[0].tiles_n = tiles_n = tn32; ml.c[0].bn32 = 1; }
    }
    unsigned long long nblk = (unsigned long long)tiles_m * tiles_n;
    // batch passes (more lines than merge_max_lines) that come with several classes run them class after class in one launch
    const bool cls_major = n_classes > 1 && (size_t)L > (size_t)tuning(TUNE_MERGE_MAX_LINES);
    if (cls_major) {
        nblk = 0;
        for (int c = 0; c < n_classes; ++c) {
            ml.cbase[c] = (unsigned)nblk;
            nblk += ((unsigned long long)tiles_m * ml.c[c].tiles_n + 7ull) & ~7ull;
        }
        for (int c = n_classes; c < 9; ++c) ml.cbase[c] = (unsigned)nblk;
        ml.cls_major = 1;
    }
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    ml.n_classes = (unsigned)n_classes; ml.L = L; ml.tiles_m = tiles_m; ml.tiles_n_total = tiles_n;
    ml.stagger = nblk > 1024 ? (unsigned)tuning(TUNE_GEMM_STAGGER) : 0u;
    // tile rows per group of the block -> tile map.  r6, verdict r5 #2: with groups of ONE tile row on the row passes a line tile's
    // tile columns are consecutive blocks, start together and share the operand panel in their XCD's L2 -- PMC FETCH_SIZE (x2) of a
    // fused forward row launch 1.95 -> 1.29 GB for 1.07 GB of operands (traffic 1.44 -> 1.11 x algorithmic), inverse row launches
    // -21 % -- and the row stage of a step takes 58.4 instead of 56.8 ms (five same-box pairs, step -1.5 %): the re-read panels
    // were Infinity-Cache hits that cost less than the lockstep of four blocks on one panel.  Both passes keep groups of 4.
    { const long long g = tuning(is_row ? TUNE_GEMM_GROUP_M_ROWS : TUNE_GEMM_GROUP_M); ml.group_m = g >= 1 && g <= 64 ? (unsigned)g : 4u; }
    PairOut po{out, tmp, (unsigned)w, (unsigned)h, (unsigned)(inverse ? leff0 : len), 0, 1, 2};
    po.tmp_out = tmp_out;
    if (class_major && inverse && desc[0].kind >= 3 && desc[0].kind <= 8) { po.cm = dct_pair_efold_inv(len) ? 2u : 1u; po.cmt = dct_pair_class_tile(len); }
    // level 2: T2 (written by kind 9, read by the half-length launches, kinds 3 / 4 sub 1) keeps the mod-4 class order, so
    // that those launches read runs instead of two doubles of every four
    if (class_major && inverse && desc[0].kind == 9) po.cm = 1;
    if (class_major && inverse && is_row && dct_pair_efold_inv(len) && (desc[0].kind == 3 || desc[0].kind == 4) && desc[0].sub == 1) po.tcm = 1;
    if (class_major && !inverse) po.ft = dct_pair_class_tile(len);
    if (fuse_cop) {
        if (inst.samex || (inverse ? (inst.epi != EPI_INV_O || po.cm != 2 || po.cmt != 128) : (inst.epi != EPI_FWD || po.ft != 128))) return SSW_ERR_BAD_ARG;
        po.cop = fuse->cop; po.cop_k16 = (unsigned)pair_kpad<double>(h / 8); po.cop_lines = (unsigned)(n_frames * w);
        po.cop_hup = (unsigned)dct_pair_fused_units(h);
        po.crot1 = fuse->rot1; po.crot2 = fuse->rot2; po.crot3 = fuse->rot3;
        inst.epi = inverse ? EPI_INV_O_COLOP : EPI_FWD_COLOP;
    }
    if (fuse_cols) po.xperm = inverse ? 2u : 1u;
    auto al = [](const void* p, unsigned a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
    po.wide = (!is_row && w % 4 == 0 && al(out, 16) && (n_frames * w) % 4 == 0 && (!tmp || al(tmp, 16)) && (!tmp_out || al(tmp_out, 16))) ? 1u : 0u;
    if (with_sink && !(al(sink->iq_i, 16) && al(sink->iq_q, 16) && al(sink->rgb, sink->u8 ? 4 : 16))) po.wide = 0;
    if (with_sink) {
        if (is_row) return SSW_ERR_BAD_ARG;
        po.iq_i = sink->iq_i; po.iq_q = sink->iq_q; po.rgb = sink->rgb; po.rgb_u8 = sink->u8 ? 1u : 0u;
    }
    ml.po = po;
#define SSW_LAUNCH_PAIR_BM(COLS, EPI, SAMEX, SUBV, BMV) \
        pair_gemm_f64_kernel<COLS, EPI, SAMEX, SUBV, BMV><<<(unsigned)nblk, PT, 0, st>>>(ml, ep)
#define SSW_LAUNCH_PAIR_SUB(COLS, EPI, SAMEX, SUBV) do { if (small) SSW_LAUNCH_PAIR_BM(COLS, EPI, SAMEX, SUBV, 64); else SSW_LAUNCH_PAIR_BM(COLS, EPI, SAMEX, SUBV, 128); } while (0)
#define SSW_LAUNCH_PAIR(COLS, EPI, SAMEX) do { if (inst.subname == 0) SSW_LAUNCH_PAIR_SUB(COLS, EPI, SAMEX, 0); else SSW_LAUNCH_PAIR_SUB(COLS, EPI, SAMEX, 1); } while (0)
#define SSW_LAUNCH_ROWCOL(EPI, SAMEX) do { if (is_row) SSW_LAUNCH_PAIR(false, EPI, SAMEX); else SSW_LAUNCH_PAIR(true, EPI, SAMEX); } while (0)
    switch (inst.epi) {
    case EPI_INV_O_COLOP: SSW_LAUNCH_PAIR_BM(false, EPI_INV_O_COLOP, false, 0, 128); break;
    case EPI_FWD_COLOP:
        if (inst.subname == 4) SSW_LAUNCH_PAIR_BM(false, EPI_FWD_COLOP, false, 4, 128);
        else SSW_LAUNCH_PAIR_BM(false, EPI_FWD_COLOP, false, 3, 128);
        break;
    case EPI_FWD_ADJ: SSW_LAUNCH_PAIR(false, EPI_FWD_ADJ, false); break;
    case EPI_FWD:
        if (inst.subname == 4) SSW_LAUNCH_PAIR_SUB(false, EPI_FWD, false, 4);
        else if (inst.subname == 3) { if (is_row) SSW_LAUNCH_PAIR_SUB(false, EPI_FWD, false, 3); else SSW_LAUNCH_PAIR_SUB(true, EPI_FWD, false, 3); }
        else if (inst.samex) SSW_LAUNCH_ROWCOL(EPI_FWD, true);
        else SSW_LAUNCH_ROWCOL(EPI_FWD, false);
        break;
    default: return launch_pair_gemm_inverse_instances(st, ml, ep, inst, is_row, small, nblk);
    }
#undef SSW_LAUNCH_ROWCOL
#undef SSW_LAUNCH_PAIR
#undef SSW_LAUNCH_PAIR_SUB
#undef SSW_LAUNCH_PAIR_BM
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// One class per launch (the batch paths).
int launch_dct_pair_gemm_f64(hipStream_t st, bool is_row, bool inverse, int kind, int sub, const double* x1, const double* x2,
                             const double* y1, const double* y2, float* out, double* tmp, size_t n_frames, size_t w,
                             size_t h, Epilogue ep, const RgbSink* sink, double* tmp_out, bool class_major) {
    const PairClassDesc d{kind, sub, x1, x2, y1, y2};
    return launch_dct_pair_gemm_multi_f64(st, is_row, inverse, 1, &d, out, tmp, n_frames, w, h, ep, sink, tmp_out, class_major);
}

// Forward row pass restricted to a gathered set of frequencies (pruned derived transform, prune.hip): one
// shared image operand `x` (k-blocked, Kp wide) against a gathered half basis `y` of `cap` rows (k-blocked,
// [Kp / 8][cap][8]); row j of y produces compact output column off + j of `out` (row stride out_stride).
// Same kernel, operands and k order as the full transform's launches -> bit-identical values.
int launch_dct_pair_gemm_rows_subset_f64(hipStream_t st, const double* x, const double* y, unsigned cap, unsigned Kp, float* out,
                                         unsigned out_stride, unsigned off, size_t lines) {
    if (lines == 0 || cap == 0) return SSW_OK;
    if (lines > 0xFFFFFFFFull || (cap & 1)) return SSW_ERR_BAD_DIMS;
    const unsigned L = (unsigned)lines, NP = cap / 2;
    const unsigned tiles_m = (L + 127) / 128, tiles_n = (NP + 63) / 64;
    const unsigned long long nblk = (unsigned long long)tiles_m * tiles_n;
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    if ((unsigned long long)Kp * L * sizeof(double) > 0xFFFFFFFFull) return SSW_ERR_BAD_DIMS;
    PairMulti ml;
    ml.c[0] = PairClassArgs{x, x, y, y + (size_t)NP * 8, NP, Kp, cap, tiles_n, off, off + NP, 1, 0, 0xFFFFFFFFu, 0, 0, 0};
    ml.n_classes = 1; ml.L = L; ml.tiles_m = tiles_m; ml.tiles_n_total = tiles_n;
    ml.po = PairOut{out, nullptr, out_stride, 0, 0, off, off + NP, 1};
    const Epilogue ep{1.f, 1.f};
    pair_gemm_f64_kernel<false, EPI_FWD, true, 2><<<(unsigned)nblk, PT, 0, st>>>(ml, ep);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// The same for a class of the split odd half: gathered cosine rows `y1` and gathered sine rows `y2` (negated where the
// output is the difference) against the class's operand pair; pair j -> compact column off + j = acc1 + acc2.
int launch_dct_pair_gemm_rows_subset_split_f64(hipStream_t st, const double* x1, const double* x2, const double* y1, const double* y2,
                                               unsigned cap, unsigned Kp, float* out, unsigned out_stride, unsigned off, size_t lines) {
    if (lines == 0 || cap == 0) return SSW_OK;
    if (lines > 0xFFFFFFFFull) return SSW_ERR_BAD_DIMS;
    const unsigned L = (unsigned)lines, NP = cap;
    const unsigned tiles_m = (L + 127) / 128, tiles_n = (NP + 63) / 64;
    const unsigned long long nblk = (unsigned long long)tiles_m * tiles_n;
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    if ((unsigned long long)Kp * L * sizeof(double) > 0xFFFFFFFFull) return SSW_ERR_BAD_DIMS;
    PairMulti ml;
    ml.c[0] = PairClassArgs{x1, x2, y1, y2, NP, Kp, cap, tiles_n, off, 0, 1, 2, 0xFFFFFFFFu, 0, 0, 0};
    ml.n_classes = 1; ml.L = L; ml.tiles_m = tiles_m; ml.tiles_n_total = tiles_n;
    ml.po = PairOut{out, nullptr, out_stride, 0, 0, off, 0, 1};
    const Epilogue ep{1.f, 1.f};
    pair_gemm_f64_kernel<false, EPI_FWD, false, 3><<<(unsigned)nblk, PT, 0, st>>>(ml, ep);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// Small passes (a single frame's derived transform: 17 tile rows, one or two tile columns per class -- nine launches of
// 13-22 us each, every one a fraction of the chip's block slots): the classes of one kind side by side in one launch's tile
// grid, like the merged launches of the full passes.  classes[c].x2 != nullptr: a class of the split odd half.
int launch_dct_pair_gemm_rows_subset_merged_f64(hipStream_t st, const PairSubsetClass* classes, unsigned n_classes, float* out,
                                                unsigned out_stride, size_t lines) {
    if (lines == 0) return SSW_OK;
    if (lines > 0xFFFFFFFFull) return SSW_ERR_BAD_DIMS;
    const unsigned L = (unsigned)lines, tiles_m = (L + 127) / 128;
    for (int split = 0; split < 2; ++split) {
        PairMulti ml;
        ml.n_classes = 0; ml.L = L; ml.tiles_m = tiles_m; ml.tiles_n_total = 0;
        auto flush = [&]() -> int {
            if (ml.n_classes == 0) return SSW_OK;
            const unsigned long long nblk = (unsigned long long)tiles_m * ml.tiles_n_total;
            if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
            ml.po = PairOut{out, nullptr, out_stride, 0, 0, 0, 0, 1};
            const Epilogue ep{1.f, 1.f};
            if (split) pair_gemm_f64_kernel<false, EPI_FWD, false, 3><<<(unsigned)nblk, PT, 0, st>>>(ml, ep);
            else       pair_gemm_f64_kernel<false, EPI_FWD, true, 2><<<(unsigned)nblk, PT, 0, st>>>(ml, ep);
            SSW_HIP_CHECK(hipGetLastError());
            ml.n_classes = 0; ml.tiles_n_total = 0;
            return SSW_OK;
        };
        for (unsigned c = 0; c < n_classes; ++c) {
            const PairSubsetClass& k = classes[c];
            if ((k.x2 != nullptr) != (split != 0) || k.cap == 0) continue;
            if (!split && (k.cap & 1)) return SSW_ERR_BAD_DIMS;
            if ((unsigned long long)k.Kp * L * sizeof(double) > 0xFFFFFFFFull) return SSW_ERR_BAD_DIMS;
            const unsigned NP = split ? k.cap : k.cap / 2, tiles_n = (NP + 63) / 64;
            if (split) ml.c[ml.n_classes] = PairClassArgs{k.x1, k.x2, k.y1, k.y2, NP, k.Kp, k.cap, tiles_n, k.off, 0, 1, 2, 0xFFFFFFFFu, 0, 0, 0};
            else       ml.c[ml.n_classes] = PairClassArgs{k.x1, k.x1, k.y1, k.y1 + (size_t)NP * 8, NP, k.Kp, k.cap, tiles_n, k.off, k.off + NP, 1, 0, 0xFFFFFFFFu, 0, 0, 0};
            ml.tiles_n_total += tiles_n;
            if (++ml.n_classes == 8) SSW_TRY(flush());
        }
        SSW_TRY(flush());
    }
    return SSW_OK;
}

}  // namespace ssw
