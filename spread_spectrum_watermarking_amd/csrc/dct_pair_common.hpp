// Shared by the operand-ready GEMMs (dct_pair_f64.hip, dct_pair_f32.hip) and their pre-passes
// (dct_pair_prep.hip).
#pragma once
#include "dct_common.hpp"

namespace ssw {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
template <typename T> using vec4_t = T __attribute__((ext_vector_type(4)));
template <typename T> using vec2_t = T __attribute__((ext_vector_type(2)));

// k-blocked operand planes: [Kp / KB][lines][KB] elements with KB * sizeof(T) = 64 bytes, i.e. the
// 64-byte piece of every line that one GEMM k-step needs lies next to its neighbours'.
template <typename T> struct KBlock { static constexpr unsigned KB = 64 / sizeof(T); };
// element (line, k) of a k-blocked plane with `rows` lines
template <typename T>
__host__ __device__ inline size_t blk_index(size_t line, unsigned k, size_t rows) {
    constexpr unsigned KB = KBlock<T>::KB;
    return ((size_t)(k / KB) * rows + line) * KB + (k % KB);
}

// Epilogues.  n = transform length, idx = output index along the transformed axis:
//   EPI_FWD    out[c1 + cs pair] = acc1, out[c2 + cs pair] = acc2          (forward, any folding level)
//   EPI_FWD_ADJ  the same with c1 = 0, c2 = 1, cs = 2 on a row pass: one 8-byte store
//   EPI_INV    out[pair] = acc1 + acc2, out[n-1-pair] = acc1 - acc2        (inverse, one level)
//   EPI_INV_E  T[pair] = acc1 + acc2, T[n/2-1-pair] = acc1 - acc2, unrounded (inverse level 2: the even half E)
//   EPI_INV_O  with n1 = pair, n2 = pair + n/4:  out[n1] = T[n1] + acc1, out[n-1-n1] = T[n1] - acc1,
//              out[n2] = T[n2] + acc2, out[n-1-n2] = T[n2] - acc2          (inverse level 2: odd part + combine)
//   EPI_INV_O_RGB  EPI_INV_O on the last pass of Writer::result (a column pass): instead of storing the Y sample it
//              converts (Y, I, Q) of that pixel to RGB like From<&YIQ32FImage> for Rgb32FImage (src/yiq.rs:187-197:
//              (y + m1 i) + m2 q per channel, clamped to [0, 1]) and stores the interleaved pixel -- f32, or
//              8-bit like into_rgb8() (round(clamp * 255)).  The colour conversion's HBM traffic (I, Q in, RGB out)
//              then runs in the shadow of the other resident block's MFMAs and the Y plane is never written.
//   EPI_INV_OT EPI_INV_O one level down (deep inverse): the odd part of the half-length transform E combined with ITS even
//              half T2 (`tmp`, length n/2 per line) into E itself, unrounded: T[n1] = T2[n1] + acc, T[n-1-n1] = T2[n1] - acc
//              (`tmp_out`, length n per line; n = the half-length transform's length)
//   EPI_FWD_COLOP (r5; forward ROW pass of a rows-first transform whose two passes run at level 2, f64): EPI_FWD's values --
//              rounded to f32 like the store between the passes (src/dct2d.rs:152-168), then the f32 per-index factor --
//              are not stored: the operand lines are ordered (frame, unit of the column fold, line of the unit), a 16-line
//              MFMA tile holds the sixteen rows of one unit, and the epilogue applies the column pre-pass's arithmetic
//              (dct_pair_colops.hpp: col_l2_unit) to them and stores the sixteen k-blocked COLUMN operand planes directly.
//              No f32 plane between the passes, no column pre-pass: 16 B/px of HBM traffic less per forward transform.
//   EPI_INV_O_COLOP (r5; the last launches of an inverse ROW pass, the same conditions): EPI_INV_O's four outputs per pair
//              -- x[n1] = E[n1] + a1, x[n-1-n1] = E[n1] - a1, the same for n2; rounded to f32 like the store between the passes
//              -- feed the inverse column pre-pass's arithmetic (inv_col_l2_unit_lo / _hi) and leave as column operands.
enum { EPI_FWD = 0, EPI_FWD_ADJ = 1, EPI_INV = 2, EPI_INV_E = 3, EPI_INV_O = 4, EPI_INV_O_RGB = 5, EPI_INV_OT = 6, EPI_FWD_COLOP = 7,
       EPI_INV_O_COLOP = 8 };

template <typename T>
struct PairOutT {
    float* out;          // f32 plane(s)
    T* tmp;              // E planes in the accumulation type (EPI_INV_E / EPI_INV_O)
    unsigned W, H;       // plane dims
    unsigned n;          // transform length (W for a row pass, H for a column pass)
    unsigned c1, c2, cs; // EPI_FWD
    // EPI_INV_O_RGB: the I and Q planes of the frames (same layout as `out`) and the interleaved RGB output
    const float* iq_i = nullptr;
    const float* iq_q = nullptr;
    void* rgb = nullptr;
    unsigned rgb_u8 = 0;
    // column passes: results leave through an LDS transpose as 16-byte pieces of whole output rows (W % 4 == 0 and
    // 16-byte aligned planes; 4-byte aligned 8-bit RGB), decided by the launcher
    unsigned wide = 0;
    // block tiles of 32 instead of 64 pairs (every tile runs the 4 x 1 wave layout): small launches whose 64-pair
    // grid would load the CUs unevenly -- a single 4K frame's column pass is 60 x 9 = 540 blocks for 256 CUs
    // r6: tile width code -- 0: 64 pairs, 1: 32 (as above), 2: 48 pairs, three pair tiles on the 4 x 1 wave grid: a class of 135 pairs
    // (4K columns) as 48 + 48 + 39 instead of 64 + 64 + 7 -- the 7-pair tile ran a whole sum's worth of operand tiles for a
    // quarter of a tile's MFMAs (without it the column pass took 15 % less time for 5 % less work)
    unsigned bn32 = 0;
    // split odd half (dct_pair_f64.hip, "rotated quarter-length pair"): the two products are a cosine and a sine
    // transform of the rotated operands and the outputs are their sum and difference: 1 = (acc1 + acc2 -> first
    // output, acc1 - acc2 -> second), 2 = the sum only (gathered rows of the pruned transform).  The first output of
    // pairs >= np1 and the second output of pairs < p2lo do not exist (the last / first pair of class E).
    unsigned pm = 0;
    unsigned np1 = 0xFFFFFFFFu, p2lo = 0;
    // class E of the split odd half has n/8 + 1 pairs of which the first has no second output (its sine row is zero) and
    // the last no first one (its cosine row is zero): with fold0 = n/8 they share pair 0 -- cosine row 0 against sine row
    // n/8 -- whose outputs are acc1 (as the first output of pair 0) and -acc2 (as the second output of pair n/8)
    unsigned fold0 = 0;
    T* tmp_out = nullptr;     // EPI_INV_OT
    unsigned cm = 0;          // inverse row pass: output line (and the E plane) in class-major order (inverse_class_pos): 1 mod 4, 2 mod 8
    unsigned tcm = 0;         // EPI_INV_OT, row pass: `tmp` (T2) is read in the mod-4 class-major order (level 2: the launch that wrote it had cm = 1)
    unsigned cmt = 0;         // ... the output line inside tiles of cmt positions (the E planes: one tile)
    // forward row pass, class-major inside tiles of ft memory columns (ft != 0): entry e = pair (output 1) or
    // p2(pair) - e2off (output 2) of its class goes to column c + (e >> gsh) * ft + (e & ((1 << gsh) - 1)); gsh = log2 of the
    // class's entries per tile (ft a power of two), or 31 for one tile over the whole line
    unsigned ft = 0, gsh = 31, e2off = 0;
    // EPI_FWD_COLOP: the sixteen column-operand planes (k-blocked: [cop_k16 / 8][cop_lines][8] doubles each, plane a at
    // cop + a * cop_lines * cop_k16), the rotation tables of axes of length H, H/2, H/4, and the units per frame of the
    // row pass's line order (H/16 rounded up to whole k-blocks)
    T* cop = nullptr;
    unsigned cop_k16 = 0, cop_lines = 0, cop_hup = 0;
    const double *crot1 = nullptr, *crot2 = nullptr, *crot3 = nullptr;
    // column pass behind such a row pass: the operand lines of a 128-line tile are in the class-major order of the row
    // launches' frequencies (fwd_cm128_pos); tile row j is staged from line m0 + fwd_cm128_pos(j), i.e. the tile's columns
    // come out natural and the epilogue is unchanged
    unsigned xperm = 0;       // 1: forward order (fwd_cm128_pos), 2: inverse order (inverse_class_pos at level 2 inside 128 positions)
};

// Column order of the intermediate plane between the two passes of a deep forward transform (row pass first): the
// frequencies of one GEMM launch class side by side, so that a launch stores runs instead of one float in 8 or 16.
// (The natural order made every row launch rewrite the whole plane: 5 x 4.2 GB of HBM writes per 128 4K frames instead
// of 4.2 GB -- PMC WRITE_SIZE -- and the clock the chip holds under that load was 10-20 % lower.)
//   class      R1     R2     E2P      E2M      O2P      O2M      EP     EM     OP     OM
//   u mod      8: 0   8: 4   16: 2    16: 14   16: 10   16: 6    8: 1   8: 7   8: 5   8: 3
//   length     t/8    t/8    t/16     t/16     t/16     t/16     t/8    t/8    t/8    t/8
// r4: the order is class-major INSIDE TILES of t natural frequencies (t = 128 when 128 divides n, else t = n: one tile):
// the t frequencies [T t, T t + t) occupy the memory columns [T t, T t + t), classes side by side.  A row launch still
// stores runs (t/8 = 16 entries = the 16 pairs of one MFMA tile), and a block of the column pre-pass now reads ONE
// contiguous 512-byte run per row for 128 consecutive operand lines (the one-tile order gave it ten runs of 64 / 32
// bytes, or -- read by memory column -- stores scattered over lines 8 or 16 apart: 2.8 TB/s).
// r4b (`level2`: row passes of 1280 columns or more, dct_pair_efold): every launch of the pass works on sums of n/16
// terms and owns two residues mod 16 -- sixteen classes of t/16 entries each, in this order:
//   class      R1A  R1B  R2A  R2B   E2P  E2M  O2P  O2M   EEP  EEM  EOP  EOM   O5  O11  O3  O13
//   u mod 16   0    8    4    12    2    14   10   6     1    15   9    7     5   11   3   13
struct ForwardClassLayout {
    unsigned n, t;
    bool level2;
    enum { R1 = 0, R2, E2P, E2M, O2P, O2M, EP, EM, OP, OM, NCLASS1 };                                    // level2 == false
    enum { R1A = 0, R1B, R2A, R2B, F_E2P, F_E2M, F_O2P, F_O2M, EEP, EEM, EOP, EOM, O5, O11, O3, O13, NCLASS2 };   // level2 == true
    __host__ __device__ ForwardClassLayout(unsigned n_, unsigned t_ = 0, bool level2_ = false) : n(n_), t(t_ ? t_ : n_), level2(level2_) {}
    __host__ __device__ int classes() const { return level2 ? (int)NCLASS2 : (int)NCLASS1; }
    // frequencies of class c: mod(c) i + res(c)
    __host__ __device__ unsigned mod(int c) const { return level2 ? 16u : ((c >= E2P && c <= O2M) ? 16u : 8u); }
    __host__ __device__ unsigned res(int c) const {
        if (level2) {
            const unsigned char r[16] = {0, 8, 4, 12, 2, 14, 10, 6, 1, 15, 9, 7, 5, 11, 3, 13};
            return r[c & 15];
        }
        const unsigned char r[10] = {0, 4, 2, 14, 10, 6, 1, 7, 5, 3};
        return r[c];
    }
    // entries of class c per tile
    __host__ __device__ unsigned group(int c) const { return t / mod(c); }
    // offset of class c inside a tile
    __host__ __device__ unsigned base(int c) const {
        const unsigned e = t / 8, s = t / 16;
        if (level2) return (unsigned)c * s;
        return c < 2 ? c * e : c < 6 ? 2 * e + (c - 2) * s : t / 2 + (c - 6) * e;
    }
    // memory column of entry i of class c
    __host__ __device__ unsigned pos(int c, unsigned i) const {
        const unsigned g = group(c);
        return (i / g) * t + base(c) + i % g;
    }
    // natural frequency of memory column p
    __host__ __device__ unsigned natural(unsigned p) const {
        const unsigned e = t / 8, s = t / 16, tb = (p / t) * t;
        p -= tb;
        int c;
        unsigned i;
        if (level2) { c = (int)(p / s); i = p - c * s; }
        else if (p < 2 * e) { c = (int)(p / e); i = p - c * e; }
        else if (p < t / 2) { const unsigned q = p - 2 * e; c = 2 + (int)(q / s); i = q - (q / s) * s; }
        else { const unsigned q = p - t / 2; c = 6 + (int)(q / e); i = q - (q / e) * e; }
        return tb + mod(c) * i + res(c);
    }
};
// tile width of the class-major orders of a line of length n (both directions): 128 when that divides n
__host__ __device__ inline unsigned class_tile(unsigned n) { return n % 128 == 0 ? 128u : n; }

// The same idea for a deep INVERSE transform's row pass: a launch of the split odd part produces the positions 4i and
// 4i-1 (class E) or 4i+2 and 4i+1 (class O) and their mirrors -- residues {0, 3} or {1, 2} mod 4 -- of the output line,
// and reads / writes the even half E at the same residues.  Class-major order of a line of length len (len % 4 == 0),
// inside tiles of t positions (t % 4 == 0, t divides len; t = len: one tile -- the order of the E planes, which only
// GEMM epilogues exchange):
//   [ m = 0 mod 4 | m = 3 mod 4 | m = 1 mod 4 | m = 2 mod 4 ], each t/4 long, m / 4 ascending.
// r4c (`l2`: inverse row passes at level 2, dct_pair_efold_inv): the four launches of the odd part produce the positions
// 8i, 8i-1 | 8i+4, 8i+3 | 8i+2, 8i-3 | 8i+1, 8i-2 and their mirrors -- residues {0, 7}, {4, 3}, {2, 5}, {1, 6} mod 8 (t % 8 == 0):
//   [ 0 | 7 | 4 | 3 | 2 | 5 | 1 | 6  mod 8 ], each t/8 long, m / 8 ascending
// (the launches of the half-length odd part, residues {0, 3} and {1, 2} mod 4, write runs of two of these classes).
__host__ __device__ inline unsigned inverse_class_pos(unsigned m, unsigned len, unsigned t = 0, bool l2 = false) {
    t = t ? t : len;
    const unsigned tb = (m / t) * t, r = m - tb;
    if (l2) {
        const unsigned run = (0x17523460u >> (4 * (r & 7u))) & 7u;      // residue -> run: 0 6 4 3 2 5 7 1
        return tb + run * (t / 8) + (r >> 3);
    }
    const unsigned c = r & 3u, q = t / 4;
    return tb + (c == 0 ? 0u : c == 3 ? q : c == 1 ? 2 * q : 3 * q) + (r >> 2);
}
__host__ __device__ inline unsigned inverse_class_natural(unsigned p, unsigned len, unsigned t = 0, bool l2 = false) {
    t = t ? t : len;
    const unsigned tb = (p / t) * t, r = p - tb;
    if (l2) {
        const unsigned q = t / 8, run = r / q, i = r - run * q;
        return tb + 8 * i + ((0x61523470u >> (4 * run)) & 7u);          // run -> residue: 0 7 4 3 2 5 1 6
    }
    const unsigned q = t / 4, c = r / q, i = r - c * q;
    return tb + 4 * i + (c == 0 ? 0u : c == 1 ? 3u : c == 2 ? 1u : 2u);
}

// yiq.rs:139-147 (f32::clamp) and :163-165, :173-175: the arithmetic of color.hip / attack.hip, per pixel
__device__ inline float pair_clamp01(float x) {
    if (x < 0.0f) return 0.0f;
    if (x > 1.0f) return 1.0f;
    return x;
}
// NB samples y[t] of pixels px[t] (where ok[t]): all I / Q loads are issued before the first store -- one memory
// latency per batch instead of one per pixel (the stores to `rgb` may alias the loads as far as the compiler knows)
// (px[t]: offset from pixel `base`, below W * H < 2^32)
template <typename T, int NB>
__device__ inline void pair_store_rgb_batch(const PairOutT<T>& po, size_t base, const unsigned (&px)[NB], const float (&y)[NB], bool ok) {
    if (!ok) return;
    const float* __restrict__ ip = po.iq_i + base;
    const float* __restrict__ qp = po.iq_q + base;
    float iv[NB], qv[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        iv[t] = ip[px[t]];
        qv[t] = qp[px[t]];
    }
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        const float r = pair_clamp01(1.0f * y[t] + 0.948262f * iv[t] + 0.624013f * qv[t]);
        const float g = pair_clamp01(1.0f * y[t] + -0.276066f * iv[t] + -0.639810f * qv[t]);
        const float b = pair_clamp01(1.0f * y[t] + -1.105450f * iv[t] + 1.729860f * qv[t]);
        if (po.rgb_u8) {
            uint8_t* o = static_cast<uint8_t*>(po.rgb) + 3 * (base + px[t]);
            o[0] = (uint8_t)roundf(pair_clamp01(r) * 255.0f);
            o[1] = (uint8_t)roundf(pair_clamp01(g) * 255.0f);
            o[2] = (uint8_t)roundf(pair_clamp01(b) * 255.0f);
        } else {
            float* o = static_cast<float*>(po.rgb) + 3 * (base + px[t]);
            o[0] = r; o[1] = g; o[2] = b;
        }
    }
}

// One quad of horizontally adjacent pixels: (Y, I, Q) -> RGB like pair_store_rgb_batch, I / Q read and RGB written as
// 16-byte pieces (f32) or 12 bytes (8-bit).  `px`: pixel index of the quad's first pixel in the frame batch.
// f32::clamp(x, 0, 1) of a finite x in one instruction (v_med3_f32); a result of -0.0 where the comparison form
// gives +0.0 compares equal and quantises to the same byte
__device__ inline float pair_clamp01_med3(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }
// into_rgb8 of a clamped channel: round(c * 255), halves away from zero, as an integer-valued float.  For x >= 0:
// floor(x + 0.5) is exact except below 0.5, where x + 0.5 can round up to 1.0 (x = 0.5 - 2^-25): those are 0.
__device__ inline float pair_round255(float c) {
    const float x = c * 255.0f;
    return x < 0.5f ? 0.0f : floorf(x + 0.5f);
}
// the twelve clamped channel values of a quad (yiq.rs:187-197: (y + m1 i) + m2 q per channel, clamped to [0, 1])
__device__ inline void pair_rgb_of_quad(const float (&y)[4], const f32x4& iv, const f32x4& qv, float (&c)[12]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        c[3 * t + 0] = pair_clamp01_med3(1.0f * y[t] + 0.948262f * iv[t] + 0.624013f * qv[t]);
        c[3 * t + 1] = pair_clamp01_med3(1.0f * y[t] + -0.276066f * iv[t] + -0.639810f * qv[t]);
        c[3 * t + 2] = pair_clamp01_med3(1.0f * y[t] + -1.105450f * iv[t] + 1.729860f * qv[t]);
    }
}
// ... as the three dwords of a quad of 8-bit pixels (into_rgb8)
__device__ inline void pair_rgb8_words(const float (&c)[12], unsigned (&w)[3]) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        unsigned v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) v = __builtin_amdgcn_cvt_pk_u8_f32(pair_round255(c[4 * d + b]), b, v);   // exact: integer-valued
        w[d] = v;
    }
}
template <typename T>
__device__ inline void pair_store_rgb_quad(const PairOutT<T>& po, size_t px, const float (&y)[4]) {
    const f32x4 iv = *reinterpret_cast<const f32x4*>(po.iq_i + px);
    const f32x4 qv = *reinterpret_cast<const f32x4*>(po.iq_q + px);
    float c[12];
    pair_rgb_of_quad(y, iv, qv, c);
    if (po.rgb_u8) {
        unsigned w[3];
        pair_rgb8_words(c, w);
        unsigned* o = reinterpret_cast<unsigned*>(static_cast<uint8_t*>(po.rgb) + 3 * px);
        o[0] = w[0]; o[1] = w[1]; o[2] = w[2];
    } else {
        f32x4* o = reinterpret_cast<f32x4*>(static_cast<float*>(po.rgb) + 3 * px);
        o[0] = (f32x4){c[0], c[1], c[2], c[3]};
        o[1] = (f32x4){c[4], c[5], c[6], c[7]};
        o[2] = (f32x4){c[8], c[9], c[10], c[11]};
    }
}

// row stride (in elements) of the operand planes / half bases of a length-n axis for precision T: n/2 rounded up to whole
// k-steps (f64: 8 per step, f32: 16), at least two of them; the f32 GEMM wants an even number of steps, the f64 one has a
// compile-time variant of its tile body for odd counts (135 -> 136 instead of 144 at 4K columns)
template <typename T> inline size_t pair_kpad(size_t n) {
    const size_t m = 2 * KBlock<T>::KB;
    return ((n / 2 + m - 1) / m) * m;
}
template <> inline size_t pair_kpad<double>(size_t n) {
    const size_t m = KBlock<double>::KB, k = ((n / 2 + m - 1) / m) * m;
    return k < 2 * m ? 2 * m : k;
}

}  // namespace ssw
