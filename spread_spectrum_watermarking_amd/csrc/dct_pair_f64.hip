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
// HBM-bound pre-passes (this file) in exactly the form the MFMA consumes:
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
// Block: 256 threads = 4 waves as 2 x 2; block tile 128 lines x 64 pairs x 2 products; k-step 8;
// per wave 16 MFMA 16x16 tiles = 128 accumulator registers; LDS 48 KB double-buffered (XOR-swizzled
// 64-byte rows, conflict-free ds_read_b128), one barrier per k-step, 2 blocks per CU.
// Lane l: li = l & 15 (line / pair inside a 16x16 tile), lq = l >> 4: MFMA step s sums
// k = 2 lq + s over the 4 lane groups (any assignment of k to MFMA slots is a valid summation
// order; A and B use the same one).
#include "dct_common.hpp"

#include <type_traits>

namespace ssw {

constexpr int PT = 256;
constexpr int PBK = 8;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Epilogues.  n = transform length, idx = output index along the transformed axis:
//   EPI_FWD    out[c1 + cs pair] = acc1, out[c2 + cs pair] = acc2          (forward, any folding level)
//   EPI_FWD_ADJ  the same with c1 = 0, c2 = 1, cs = 2 on a row pass: one 8-byte store
//   EPI_INV    out[pair] = acc1 + acc2, out[n-1-pair] = acc1 - acc2        (inverse, one level)
//   EPI_INV_E  T[pair] = acc1 + acc2, T[n/2-1-pair] = acc1 - acc2  in f64  (inverse level 2: the even half E)
//   EPI_INV_O  with n1 = pair, n2 = pair + n/4:  out[n1] = T[n1] + acc1, out[n-1-n1] = T[n1] - acc1,
//              out[n2] = T[n2] + acc2, out[n-1-n2] = T[n2] - acc2          (inverse level 2: odd part + combine)
// element (line, k) of a k-blocked operand plane with `rows` lines
__host__ __device__ inline size_t blk_index(size_t line, unsigned k, size_t rows) {
    return ((size_t)(k >> 3) * rows + line) * 8 + (k & 7);
}

enum { EPI_FWD = 0, EPI_FWD_ADJ = 1, EPI_INV = 2, EPI_INV_E = 3, EPI_INV_O = 4 };

struct PairOut {
    float* out;          // f32 plane(s)
    double* tmp;         // f64 E planes (EPI_INV_E / EPI_INV_O)
    unsigned W, H;       // plane dims
    unsigned n;          // transform length (W for a row pass, H for a column pass)
    unsigned c1, c2, cs; // EPI_FWD
};

// COLS: lines are (frame, column) and the transformed axis runs down the rows.  SAMEX: X2 == X1
// (one product's image operand feeds both basis operands; it is staged and read once).
// (A 256-line x 32-pair block tile for pair counts that 64 divides badly -- 540 at 4K -- was measured:
// equal on the shared-X launches, 10 % slower on the two-operand ones; not kept.)
template <bool COLS, int EPI, bool SAMEX>
__global__ __launch_bounds__(PT, 2) void pair_gemm_f64_kernel(
    const double* __restrict__ X1g, const double* __restrict__ X2g, const double* __restrict__ Y1g,
    const double* __restrict__ Y2g, PairOut po, unsigned L /*lines*/, unsigned NP /*pairs*/,
    unsigned Kp, unsigned yrows /*lines of the basis planes*/, unsigned tiles_m, unsigned tiles_n, Epilogue ep) {
    constexpr int NX = SAMEX ? 1 : 2;
    constexpr int BM = 128, BN = 64, XQ = 2;                               // XQ: X lines per staging thread
    __shared__ __attribute__((aligned(16))) double sX[2][NX][BM * PBK];    // [buffer][product]
    __shared__ __attribute__((aligned(16))) double sY[2][2][BN * PBK];

    unsigned tm, tn;
    tile_of_block(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
    const unsigned m0 = tm * BM, p0 = tn * BN;
    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63, wave = tid >> 6;
    const unsigned wm = (wave >> 1) * 64, wn = (wave & 1) * 32;
    const unsigned li = lane & 15, lq = lane >> 4;

    // LDS tile rows hold 8 consecutive k (64 bytes); double k of row r sits at position
    // k ^ ((r >> 1) & 7): conflict-free for the ds_read_b64 / ds_read2_b64 fragment reads (16 lanes
    // cover a 128-byte bank window exactly) and for the staging ds_write_b64.
    // staging: line = tid / 4 (+ 64 q), k-pair = tid % 4
    const unsigned srow = tid >> 2, sc = tid & 3;
    const unsigned ssw = (srow >> 1) & 7;
    unsigned xoff[XQ];
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
        unsigned r = m0 + srow + 64 * q;
        r = r < L ? r : L - 1;
        xoff[q] = ((r - m0) * 8 + 2 * sc) * 8u;
    }
    unsigned yr = p0 + srow;
    yr = yr < NP ? yr : NP - 1;
    const unsigned yoff = ((yr - p0) * 8 + 2 * sc) * 8u;
    // block-uniform buffer resources (scalar registers) at the tile's first line of k-block 0; a k-step
    // advances a scalar byte offset by one k-block (< 4 GB: checked by the launcher)
    const __amdgpu_buffer_rsrc_t x1r = __builtin_amdgcn_make_buffer_rsrc((void*)(X1g + (size_t)m0 * 8), 0, 0xFFFFFFFFu, 0x00020000);
    const __amdgpu_buffer_rsrc_t x2r = __builtin_amdgcn_make_buffer_rsrc((void*)(X2g + (size_t)m0 * 8), 0, 0xFFFFFFFFu, 0x00020000);
    const __amdgpu_buffer_rsrc_t y1r = __builtin_amdgcn_make_buffer_rsrc((void*)(Y1g + (size_t)p0 * 8), 0, 0xFFFFFFFFu, 0x00020000);
    const __amdgpu_buffer_rsrc_t y2r = __builtin_amdgcn_make_buffer_rsrc((void*)(Y2g + (size_t)p0 * 8), 0, 0xFFFFFFFFu, 0x00020000);
    const unsigned xstep = L * 64u, ystep = yrows * 64u;

    f64x4 acc1[4][2], acc2[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { acc1[i][j] = (f64x4){0, 0, 0, 0}; acc2[i][j] = (f64x4){0, 0, 0, 0}; }

    u32x4 rx1[XQ], rx2[XQ], ry1, ry2;
    auto gload = [&](unsigned t) {
        const unsigned xadv = t * xstep, yadv = t * ystep;
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            rx1[q] = __builtin_amdgcn_raw_buffer_load_b128(x1r, xoff[q], xadv, 0);
            if (!SAMEX) rx2[q] = __builtin_amdgcn_raw_buffer_load_b128(x2r, xoff[q], xadv, 0);
        }
        ry1 = __builtin_amdgcn_raw_buffer_load_b128(y1r, yoff, yadv, 0);
        ry2 = __builtin_amdgcn_raw_buffer_load_b128(y2r, yoff, yadv, 0);
    };
    const unsigned st0 = srow * PBK + ((2 * sc) ^ ssw), st1 = srow * PBK + ((2 * sc + 1) ^ ssw);
    auto put = [&](double* tile, const u32x4& v) {
        *reinterpret_cast<u32x2*>(tile + st0) = (u32x2){v[0], v[1]};
        *reinterpret_cast<u32x2*>(tile + st1) = (u32x2){v[2], v[3]};
    };
    auto lstore = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            put(&sX[buf][0][64 * q * PBK], rx1[q]);
            if (!SAMEX) put(&sX[buf][NX - 1][64 * q * PBK], rx2[q]);
        }
        put(&sY[buf][0][0], ry1);
        put(&sY[buf][1][0], ry2);
    };
    // fragment of half-step s: lane group lq supplies k = 4 s + lq
    const unsigned fsw = (li >> 1) & 7;
    unsigned rdx[2], rdy[2];
#pragma unroll
    for (int sh = 0; sh < 2; ++sh) {
        rdx[sh] = (wm + li) * PBK + ((4 * sh + lq) ^ fsw);
        rdy[sh] = (wn + li) * PBK + ((4 * sh + lq) ^ fsw);
    }
    struct Frag { double x1[4], x2[4], y1[2], y2[2]; };
    auto fread = [&](auto bufc, auto shc, Frag& f) {
        constexpr int cur = decltype(bufc)::value;
        constexpr int sh = decltype(shc)::value;
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            f.y1[jn] = sY[cur][0][rdy[sh] + 16 * jn * PBK];
            f.y2[jn] = sY[cur][1][rdy[sh] + 16 * jn * PBK];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f.x1[i] = sX[cur][0][rdx[sh] + 16 * i * PBK];
            if (!SAMEX) f.x2[i] = sX[cur][NX - 1][rdx[sh] + 16 * i * PBK];
        }
    };
    auto fmma = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const double xb = SAMEX ? f.x1[i] : f.x2[i];
                if (!COLS) {      // D[row = line][col = pair]
                    acc1[i][jn] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.x1[i], f.y1[jn], acc1[i][jn], 0, 0, 0);
                    acc2[i][jn] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb, f.y2[jn], acc2[i][jn], 0, 0, 0);
                } else {          // D[row = pair][col = line]: image columns along the lanes
                    acc1[i][jn] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.y1[jn], f.x1[i], acc1[i][jn], 0, 0, 0);
                    acc2[i][jn] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.y2[jn], xb, acc2[i][jn], 0, 0, 0);
                }
            }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;

    // Software pipeline, shifted by half a k-step: the fragments of a half-step are read from LDS
    // while the MFMAs of the previous half-step run; the tile of step t+1 is written (and the
    // barrier taken) in the middle of step t, its global loads having been issued a step earlier.
    const unsigned nk = Kp / PBK;          // even and >= 2: Kp is a multiple of 16
    Frag fa, fb;
    gload(0);
    lstore(B0{});
    __syncthreads();
    gload(1);
    fread(B0{}, B0{}, fa);
    // full step t on buffer CUR: needs t + 2 < nk
    // one LDS read behind each of the first MFMAs of a half-step (a burst of reads would stall the
    // wave at the LDS queue with its MFMAs behind it), then the staging writes two per MFMA
    auto interleave = [&](auto storec) {
        constexpr bool STORE = decltype(storec)::value != 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (STORE) {
            constexpr int NLD = XQ * NX + 2;                    // staged 16-byte loads per thread (2 LDS writes each)
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 10 - NLD, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, NLD, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 10, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto step = [&](auto curc, auto nxtc, unsigned t) {
        fread(curc, B1{}, fb);
        fmma(fa);
        lstore(nxtc);                          // loaded a whole step ago
        gload(t + 2);
        interleave(B1{});
        __syncthreads();
        fread(nxtc, B0{}, fa);
        fmma(fb);
        interleave(B0{});
    };
    unsigned t = 0;
    for (; t + 2 < nk; t += 2) {
        step(B0{}, B1{}, t);
        step(B1{}, B0{}, t + 1);
    }
    // steps nk - 2 (buffer 0) and nk - 1 (buffer 1)
    fread(B0{}, B1{}, fb);
    fmma(fa);
    interleave(B0{});
    lstore(B1{});
    __syncthreads();
    fread(B1{}, B0{}, fa);
    fmma(fb);
    interleave(B0{});
    fread(B1{}, B1{}, fb);
    fmma(fa);
    interleave(B0{});
    fmma(fb);

    // D map of 16x16x4 f64: col = lane & 15, row = (lane >> 4) + 4 reg
    const unsigned n = po.n, W = po.W, H = po.H;
    // one output line: element idx of the transformed axis lives at lp[idx * es] (and tp[idx * es] in T)
    auto emit = [&](float* lp, double* tp, size_t es, unsigned pair, double a1, double a2) {
        if (EPI == EPI_FWD || EPI == EPI_FWD_ADJ) {
            const unsigned i1 = po.c1 + po.cs * pair, i2 = po.c2 + po.cs * pair;
            if (EPI == EPI_FWD_ADJ) {
                const f32x2 v = {apply_epilogue(ep, (float)a1, i1), apply_epilogue(ep, (float)a2, i2)};
                *reinterpret_cast<f32x2*>(lp + i1) = v;
            } else {
                lp[i1 * es] = apply_epilogue(ep, (float)a1, i1);
                lp[i2 * es] = apply_epilogue(ep, (float)a2, i2);
            }
        } else if (EPI == EPI_INV) {
            lp[pair * es] = apply_epilogue(ep, (float)(a1 + a2), pair);
            lp[(n - 1 - pair) * es] = apply_epilogue(ep, (float)(a1 - a2), n - 1 - pair);
        } else if (EPI == EPI_INV_E) {
            tp[pair * es] = a1 + a2;
            tp[(n / 2 - 1 - pair) * es] = a1 - a2;
        } else {
            const unsigned n1 = pair, n2 = pair + n / 4;
            const double e1 = tp[n1 * es], e2 = tp[n2 * es];
            lp[n1 * es] = apply_epilogue(ep, (float)(e1 + a1), n1);
            lp[(n - 1 - n1) * es] = apply_epilogue(ep, (float)(e1 - a1), n - 1 - n1);
            lp[n2 * es] = apply_epilogue(ep, (float)(e2 + a2), n2);
            lp[(n - 1 - n2) * es] = apply_epilogue(ep, (float)(e2 - a2), n - 1 - n2);
        }
    };
    if (!COLS) {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const unsigned pair = p0 + wn + 16 * jn + li;
            if (pair >= NP) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned row = m0 + wm + 16 * i + lq + 4 * r;
                    if (row >= L) continue;
                    emit(po.out + (size_t)row * W, po.tmp + (size_t)row * (n / 2), 1, pair, acc1[i][jn][r], acc2[i][jn][r]);
                }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned line = m0 + wm + 16 * i + li;       // = frame * W + column
            if (line >= L) continue;
            const unsigned z = line / W, col = line - z * W;
            float* lp = po.out + (size_t)z * H * W + col;
            double* tp = po.tmp + (size_t)z * (n / 2) * W + col;
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned pair = p0 + wn + 16 * jn + lq + 4 * r;
                    if (pair >= NP) continue;
                    emit(lp, tp, W, pair, acc1[i][jn][r], acc2[i][jn][r]);
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Half bases in the k-blocked layout: [Kp / 8][n / 2][8], same values as make_half_basis_f64_kernel.
// ---------------------------------------------------------------------------------------------
__global__ void make_half_basis_blocked_f64_kernel(size_t n, bool inverse, int parity, size_t kpad, double* out) {
    const size_t nh = n / 2, total = nh * kpad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t o = i / kpad, s = i % kpad;
        double v = 0.0;
        if (s < nh) {
            const size_t freq = inverse ? 2 * s + parity : 2 * o + parity;
            const size_t pos = inverse ? o : s;
            unsigned long long a = (unsigned long long)freq * (2ull * pos + 1ull);
            a %= 4ull * n;
            const double c = cospi((double)a / (double)(2ull * n));
            v = !inverse ? 2.0 * c : (freq == 0 ? 0.25 : 0.5 * c);
        }
        out[blk_index(o, (unsigned)s, nh)] = v;
    }
}

int launch_make_half_basis_blocked_f64(hipStream_t st, size_t n, bool inverse, int parity, double* out) {
    const size_t total = (n / 2) * half_basis_kpad(n);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    make_half_basis_blocked_f64_kernel<<<blocks ? blocks : 1, 256, 0, st>>>(n, inverse, parity, half_basis_kpad(n), out);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// ---------------------------------------------------------------------------------------------
// Pre-passes (HBM-bound): f32 plane -> the f64 operand planes of the pass, k-blocked.
// One folding level:   forward  O1 = S, O2 = D;   inverse  O1 = E (even coefficients), O2 = O (odd)
// ---------------------------------------------------------------------------------------------
// Row pass: line = image row, k along the row.  Block = 32 lines x 32 k; thread = 4 consecutive k of
// one line: 128-byte read runs per line, 512-byte write runs per k-block.
template <bool INVERSE>
__global__ __launch_bounds__(256) void pair_prep_rows_kernel(const float* __restrict__ X, double* __restrict__ O1,
                                                            double* __restrict__ O2, unsigned rows, unsigned W, unsigned Kp,
                                                            unsigned tiles_k) {
    const unsigned Nh = W / 2;
    const unsigned s = (blockIdx.x % tiles_k) * 32 + (threadIdx.x & 7) * 4;
    const unsigned row = (blockIdx.x / tiles_k) * 32 + (threadIdx.x >> 3);
    if (row >= rows || s >= Kp) return;
    f64x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    if (s < Nh) {                                                 // Nh % 4 == 0
        const float* x = X + (size_t)row * W;
        if (!INVERSE) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(x + s);
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (W - 4 - s));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = (double)u[e] + (double)v[3 - e];
                b[e] = (double)u[e] - (double)v[3 - e];
            }
        } else {
            const f32x4 u = *reinterpret_cast<const f32x4*>(x + 2 * s);
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + 2 * s + 4);
            a = (f64x4){(double)u[0], (double)u[2], (double)v[0], (double)v[2]};
            b = (f64x4){(double)u[1], (double)u[3], (double)v[1], (double)v[3]};
        }
    }
    *reinterpret_cast<f64x4*>(O1 + blk_index(row, s, rows)) = a;
    *reinterpret_cast<f64x4*>(O2 + blk_index(row, s, rows)) = b;
}

// Column pass: line = (frame, column), k along the image rows: fold / split + transpose through LDS.
// Block tile: 32 k x 64 columns; written as 4 KB runs (64 lines x one 64-byte k-block piece).
template <bool INVERSE>
__global__ __launch_bounds__(256) void pair_prep_cols_kernel(const float* __restrict__ IN, double* __restrict__ O1,
                                                            double* __restrict__ O2, unsigned W, unsigned H, unsigned Kp,
                                                            unsigned n_frames, unsigned tiles_k, unsigned tiles_c) {
    __shared__ double s1[64][33];
    __shared__ double s2[64][33];
    const unsigned Hh = H / 2;
    const unsigned z = blockIdx.x / (tiles_k * tiles_c);
    const unsigned tt = blockIdx.x % (tiles_k * tiles_c);
    const unsigned k0 = (tt % tiles_k) * 32, c0 = (tt / tiles_k) * 64;
    const float* __restrict__ P = IN + (size_t)z * H * W;
    const unsigned tid = threadIdx.x;
    {
        const unsigned kr = tid >> 4, cq = (tid & 15) * 4;       // 16 k-rows per sweep, 4 columns per thread
        unsigned c = c0 + cq;
        c = c + 4 <= W ? c : W - 4;                               // W % 4 == 0; duplicates are never written out
#pragma unroll
        for (int sw = 0; sw < 2; ++sw) {
            const unsigned kl = kr + 16 * sw, k = k0 + kl;
            f32x4 u = {0.f, 0.f, 0.f, 0.f}, v = {0.f, 0.f, 0.f, 0.f};
            if (k < Hh) {
                const unsigned ra = INVERSE ? 2 * k : k, rb = INVERSE ? 2 * k + 1 : H - 1 - k;
                u = *reinterpret_cast<const f32x4*>(P + (size_t)ra * W + c);
                v = *reinterpret_cast<const f32x4*>(P + (size_t)rb * W + c);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1[cq + e][kl] = INVERSE ? (double)u[e] : (double)u[e] + (double)v[e];
                s2[cq + e][kl] = INVERSE ? (double)v[e] : (double)u[e] - (double)v[e];
            }
        }
    }
    __syncthreads();
    {
        const unsigned cl = tid & 63, kq = (tid >> 6) * 8;        // one 64-byte k-block piece of one column per thread
        const unsigned c = c0 + cl;
        if (c < W && k0 + kq < Kp) {                              // Kp % 8 == 0
            const size_t at = blk_index((size_t)z * W + c, k0 + kq, (size_t)n_frames * W);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                *reinterpret_cast<f64x2*>(O1 + at + e) = (f64x2){s1[cl][kq + e], s1[cl][kq + e + 1]};
                *reinterpret_cast<f64x2*>(O2 + at + e) = (f64x2){s2[cl][kq + e], s2[cl][kq + e + 1]};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Two-level pre-passes: f32 plane -> (SS, SD, D) forward / (EE, EO, O) inverse in one sweep
// (12 B/px of HBM traffic; a separate second-level pass over S would make it 20).
//   forward, q < n/4:  S[q] = x[q] + x[n-1-q],  S' = x[n/2-1-q] + x[n/2+q];  SS = S + S',  SD = S - S'
//                      D[q] = x[q] - x[n-1-q],  D[n/2-1-q] = x[n/2-1-q] - x[n/2+q]
//   inverse, q < n/4:  EE[q] = c[4q],  EO[q] = c[4q+2],  O[2q] = c[4q+1],  O[2q+1] = c[4q+3]
// Q1, Q2: kq = half_basis_kpad(n/2) wide; P: kp = half_basis_kpad(n) wide; k-blocked, zero padded.
// ---------------------------------------------------------------------------------------------
template <bool INVERSE>
__global__ __launch_bounds__(256) void pair_prep4_rows_kernel(const float* __restrict__ X, double* __restrict__ Q1,
                                                             double* __restrict__ Q2, double* __restrict__ P,
                                                             unsigned rows, unsigned W, unsigned Kq, unsigned Kp,
                                                             unsigned tiles_q) {
    const unsigned Nh = W / 2, Nq = W / 4;
    const unsigned q = (blockIdx.x % tiles_q) * 32 + (threadIdx.x & 7) * 4;
    const unsigned row = (blockIdx.x / tiles_q) * 32 + (threadIdx.x >> 3);
    if (row >= rows || q >= Kq) return;
    f64x4 a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
    if (q < Nq) {                                                 // Nq % 4 == 0
        const float* x = X + (size_t)row * W;
        if (!INVERSE) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(x + q);
            const f32x4 b = *reinterpret_cast<const f32x4*>(x + (Nh - 4 - q));
            const f32x4 c = *reinterpret_cast<const f32x4*>(x + (Nh + q));
            const f32x4 d = *reinterpret_cast<const f32x4*>(x + (W - 4 - q));
            f64x4 dn, dm;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double s1 = (double)a[e] + (double)d[3 - e], s2 = (double)b[3 - e] + (double)c[e];
                a1[e] = s1 + s2;
                a2[e] = s1 - s2;
                dn[e] = (double)a[e] - (double)d[3 - e];
                dm[3 - e] = (double)b[3 - e] - (double)c[e];
            }
            *reinterpret_cast<f64x4*>(P + blk_index(row, q, rows)) = dn;
            *reinterpret_cast<f64x4*>(P + blk_index(row, Nh - 4 - q, rows)) = dm;
        } else {
            f32x4 c[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) c[e] = *reinterpret_cast<const f32x4*>(x + 4 * (q + e));
#pragma unroll
            for (int e = 0; e < 4; ++e) { a1[e] = (double)c[e][0]; a2[e] = (double)c[e][2]; }
            double* o = P + blk_index(row, 2 * q, rows);           // 2 q is a multiple of 8: one whole k-block piece
            *reinterpret_cast<f64x4*>(o) = (f64x4){(double)c[0][1], (double)c[0][3], (double)c[1][1], (double)c[1][3]};
            *reinterpret_cast<f64x4*>(o + 4) = (f64x4){(double)c[2][1], (double)c[2][3], (double)c[3][1], (double)c[3][3]};
        }
    }
    *reinterpret_cast<f64x4*>(Q1 + blk_index(row, q, rows)) = a1;
    *reinterpret_cast<f64x4*>(Q2 + blk_index(row, q, rows)) = a2;
    if (q == 0)
        for (unsigned z = Nh; z < Kp; z += 4) *reinterpret_cast<f64x4*>(P + blk_index(row, z, rows)) = (f64x4){0, 0, 0, 0};
}

// Column pass: lines = (frame, column); block tile 32 q x 32 columns, transposed through LDS and
// written as 2 KB runs (32 lines x one 64-byte k-block piece).
template <bool INVERSE>
__global__ __launch_bounds__(256) void pair_prep4_cols_kernel(const float* __restrict__ IN, double* __restrict__ Q1,
                                                             double* __restrict__ Q2, double* __restrict__ P,
                                                             unsigned W, unsigned H, unsigned Kq, unsigned Kp,
                                                             unsigned n_frames, unsigned tiles_q, unsigned tiles_c) {
    __shared__ double sA[32][33], sB[32][33], sC[32][33], sD[32][33];
    const unsigned Hh = H / 2, Hq = H / 4;
    const unsigned z = blockIdx.x / (tiles_q * tiles_c);
    const unsigned tt = blockIdx.x % (tiles_q * tiles_c);
    const unsigned q0 = (tt % tiles_q) * 32, c0 = (tt / tiles_q) * 32;
    const float* __restrict__ Pz = IN + (size_t)z * H * W;
    const unsigned tid = threadIdx.x;
    {
        const unsigned qr = tid >> 3, cq = (tid & 7) * 4, q = q0 + qr;
        unsigned c = c0 + cq;
        c = c + 4 <= W ? c : W - 4;                               // W % 4 == 0; duplicates are never written out
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a, cc = a, d = a;
        if (q < Hq) {
            const unsigned ra = INVERSE ? 4 * q : q, rb = INVERSE ? 4 * q + 2 : Hh - 1 - q;
            const unsigned rc = INVERSE ? 4 * q + 1 : Hh + q, rd = INVERSE ? 4 * q + 3 : H - 1 - q;
            a = *reinterpret_cast<const f32x4*>(Pz + (size_t)ra * W + c);
            b = *reinterpret_cast<const f32x4*>(Pz + (size_t)rb * W + c);
            cc = *reinterpret_cast<const f32x4*>(Pz + (size_t)rc * W + c);
            d = *reinterpret_cast<const f32x4*>(Pz + (size_t)rd * W + c);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (!INVERSE) {
                const double s1 = (double)a[e] + (double)d[e], s2 = (double)b[e] + (double)cc[e];
                sA[cq + e][qr] = s1 + s2;
                sB[cq + e][qr] = s1 - s2;
                sC[cq + e][qr] = (double)a[e] - (double)d[e];        // D[q]
                sD[cq + e][qr] = (double)b[e] - (double)cc[e];       // D[H/2-1-q]
            } else {
                sA[cq + e][qr] = (double)a[e];                        // EE[q]
                sB[cq + e][qr] = (double)b[e];                        // EO[q]
                sC[cq + e][qr] = (double)cc[e];                       // O[2q]
                sD[cq + e][qr] = (double)d[e];                        // O[2q+1]
            }
        }
    }
    __syncthreads();
    {
        const unsigned cl = tid & 31, kq = (tid >> 5) * 4;         // 4 consecutive q of one column per thread
        const unsigned c = c0 + cl, q = q0 + kq;
        if (c < W && q < Kq) {
            const size_t line = (size_t)z * W + c, lines = (size_t)n_frames * W;
            *reinterpret_cast<f64x4*>(Q1 + blk_index(line, q, lines)) = (f64x4){sA[cl][kq], sA[cl][kq + 1], sA[cl][kq + 2], sA[cl][kq + 3]};
            *reinterpret_cast<f64x4*>(Q2 + blk_index(line, q, lines)) = (f64x4){sB[cl][kq], sB[cl][kq + 1], sB[cl][kq + 2], sB[cl][kq + 3]};
            if (q < Hq) {
                if (!INVERSE) {
                    *reinterpret_cast<f64x4*>(P + blk_index(line, q, lines)) = (f64x4){sC[cl][kq], sC[cl][kq + 1], sC[cl][kq + 2], sC[cl][kq + 3]};
                    *reinterpret_cast<f64x4*>(P + blk_index(line, Hh - 4 - q, lines)) = (f64x4){sD[cl][kq + 3], sD[cl][kq + 2], sD[cl][kq + 1], sD[cl][kq]};
                } else {
                    double* o = P + blk_index(line, 2 * q, lines);
                    *reinterpret_cast<f64x4*>(o) = (f64x4){sC[cl][kq], sD[cl][kq], sC[cl][kq + 1], sD[cl][kq + 1]};
                    *reinterpret_cast<f64x4*>(o + 4) = (f64x4){sC[cl][kq + 2], sD[cl][kq + 2], sC[cl][kq + 3], sD[cl][kq + 3]};
                }
            }
            if (q == 0)
                for (unsigned zz = Hh; zz < Kp; zz += 4) *reinterpret_cast<f64x4*>(P + blk_index(line, zz, lines)) = (f64x4){0, 0, 0, 0};
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------------------------
size_t dct_pair_operand_elems(size_t n_frames, size_t w, size_t h) {
    const size_t a = n_frames * h * half_basis_kpad(w), b = n_frames * w * half_basis_kpad(h);
    return a > b ? a : b;
}

bool dct_pair_can_run(size_t n_frames, size_t w, size_t h, const float* in, const float* out) {
    // an operand plane must stay below 4 GB (32-bit scalar offsets walk its k-blocks)
    return dct_rows_can_fold(w, in, out) && dct_cols_can_fold(w, h, in, out) && w % 8 == 0 && h % 8 == 0 &&
           dct_pair_operand_elems(n_frames, w, h) * sizeof(double) <= 0xFFFFFFFFull;
}
// second level along an axis of length len: quarter length a multiple of 4, at least one k-step pair
bool dct_pair_can_fold2(size_t len) { return len % 16 == 0 && len >= 64; }


int launch_dct_pair_prep_f64(hipStream_t st, bool is_row, bool inverse, const float* in, size_t n_frames, size_t w,
                             size_t h, double* o1, double* o2) {
    if (n_frames == 0) return SSW_OK;
    if (w > 0xFFFFFFull || h > 0xFFFFFFull || n_frames > 0xFFFFFFull) return SSW_ERR_BAD_DIMS;
    if (is_row) {
        const unsigned Kp = (unsigned)half_basis_kpad(w), tiles_k = (Kp + 31) / 32;
        const size_t rows = n_frames * h;
        const unsigned long long nblk = (unsigned long long)((rows + 31) / 32) * tiles_k;
        if (rows > 0xFFFFFFFFull || nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
        if (inverse) pair_prep_rows_kernel<true><<<(unsigned)nblk, 256, 0, st>>>(in, o1, o2, (unsigned)rows, (unsigned)w, Kp, tiles_k);
        else         pair_prep_rows_kernel<false><<<(unsigned)nblk, 256, 0, st>>>(in, o1, o2, (unsigned)rows, (unsigned)w, Kp, tiles_k);
    } else {
        const unsigned Kp = (unsigned)half_basis_kpad(h);
        const unsigned tiles_k = Kp / 32 + (Kp % 32 ? 1 : 0), tiles_c = (unsigned)((w + 63) / 64);
        const unsigned long long nblk = (unsigned long long)tiles_k * tiles_c * n_frames;
        if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
        if (inverse) pair_prep_cols_kernel<true><<<(unsigned)nblk, 256, 0, st>>>(in, o1, o2, (unsigned)w, (unsigned)h, Kp, (unsigned)n_frames, tiles_k, tiles_c);
        else         pair_prep_cols_kernel<false><<<(unsigned)nblk, 256, 0, st>>>(in, o1, o2, (unsigned)w, (unsigned)h, Kp, (unsigned)n_frames, tiles_k, tiles_c);
    }
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

int launch_dct_pair_prep4_f64(hipStream_t st, bool is_row, bool inverse, const float* in, size_t n_frames, size_t w,
                              size_t h, double* q1, double* q2, double* p) {
    if (n_frames == 0) return SSW_OK;
    if (w > 0xFFFFFFull || h > 0xFFFFFFull || n_frames > 0xFFFFFFull) return SSW_ERR_BAD_DIMS;
    const size_t len = is_row ? w : h;
    const unsigned Kp = (unsigned)half_basis_kpad(len), Kq = (unsigned)half_basis_kpad(len / 2);
    const unsigned tiles_q = (Kq + 31) / 32;
    if (is_row) {
        const size_t rows = n_frames * h;
        const unsigned long long nblk = (unsigned long long)((rows + 31) / 32) * tiles_q;
        if (rows > 0xFFFFFFFFull || nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
        if (inverse) pair_prep4_rows_kernel<true><<<(unsigned)nblk, 256, 0, st>>>(in, q1, q2, p, (unsigned)rows, (unsigned)w, Kq, Kp, tiles_q);
        else         pair_prep4_rows_kernel<false><<<(unsigned)nblk, 256, 0, st>>>(in, q1, q2, p, (unsigned)rows, (unsigned)w, Kq, Kp, tiles_q);
    } else {
        const unsigned tiles_c = (unsigned)((w + 31) / 32);
        const unsigned long long nblk = (unsigned long long)tiles_q * tiles_c * n_frames;
        if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
        if (inverse) pair_prep4_cols_kernel<true><<<(unsigned)nblk, 256, 0, st>>>(in, q1, q2, p, (unsigned)w, (unsigned)h, Kq, Kp, (unsigned)n_frames, tiles_q, tiles_c);
        else         pair_prep4_cols_kernel<false><<<(unsigned)nblk, 256, 0, st>>>(in, q1, q2, p, (unsigned)w, (unsigned)h, Kq, Kp, (unsigned)n_frames, tiles_q, tiles_c);
    }
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// One launch of the operand-ready GEMM.  `kind` selects the epilogue:
//   0  one folding level (forward: interleave even/odd; inverse: mirror)            pairs = len/2, K = len/2
//   1  level 2, even half: X = (SS, SD) | (EE, EO), Y = half bases of len/2          pairs = len/4, K = len/4
//   2  level 2, odd half:  X = D | O (shared), Y = the two halves of the odd basis   pairs = len/4, K = len/2
// x*, y*: k-blocked planes (y2 of kind 2 = y1 + 8 * len/4: the second row block of the same plane).
int launch_dct_pair_gemm_f64(hipStream_t st, bool is_row, bool inverse, int kind, const double* x1, const double* x2,
                             const double* y1, const double* y2, float* out, double* tmp, size_t n_frames, size_t w,
                             size_t h, Epilogue ep) {
    if (n_frames == 0) return SSW_OK;
    if (w > 0xFFFFFFull || h > 0xFFFFFFull) return SSW_ERR_BAD_DIMS;
    const size_t lines = is_row ? n_frames * h : n_frames * w;
    const size_t len = is_row ? w : h;
    if (lines > 0xFFFFFFFFull) return SSW_ERR_BAD_DIMS;
    const unsigned L = (unsigned)lines;
    const unsigned NP = (unsigned)(kind == 0 ? len / 2 : len / 4);
    const unsigned Kp = (unsigned)(kind == 1 ? half_basis_kpad(len / 2) : half_basis_kpad(len));
    const unsigned BM = 128, BN = 64;
    const unsigned tiles_m = (L + BM - 1) / BM, tiles_n = (NP + BN - 1) / BN;
    const unsigned long long nblk = (unsigned long long)tiles_m * tiles_n;
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    PairOut po{out, tmp, (unsigned)w, (unsigned)h, (unsigned)len, 0, 1, 2};
    if (kind == 1) { po.c1 = 0; po.c2 = 2; po.cs = 4; }
    if (kind == 2) { po.c1 = 1; po.c2 = 1 + 2 * NP; po.cs = 2; }
    const unsigned yrows = kind == 2 ? 2 * NP : NP;          // lines of the basis plane(s)
    if ((unsigned long long)Kp * L * 8 > 0xFFFFFFFFull) return SSW_ERR_BAD_DIMS;   // scalar k-block offsets are 32-bit
#define SSW_LAUNCH_PAIR(COLS, EPI, SAMEX) pair_gemm_f64_kernel<COLS, EPI, SAMEX><<<(unsigned)nblk, PT, 0, st>>>( \
        x1, x2, y1, y2, po, L, NP, Kp, yrows, tiles_m, tiles_n, ep)
    if (!inverse) {
        if (kind == 0) { if (is_row) SSW_LAUNCH_PAIR(false, EPI_FWD_ADJ, false); else SSW_LAUNCH_PAIR(true, EPI_FWD, false); }
        else if (kind == 1) { if (is_row) SSW_LAUNCH_PAIR(false, EPI_FWD, false); else SSW_LAUNCH_PAIR(true, EPI_FWD, false); }
        else { if (is_row) SSW_LAUNCH_PAIR(false, EPI_FWD, true); else SSW_LAUNCH_PAIR(true, EPI_FWD, true); }
    } else {
        if (kind == 0) { if (is_row) SSW_LAUNCH_PAIR(false, EPI_INV, false); else SSW_LAUNCH_PAIR(true, EPI_INV, false); }
        else if (kind == 1) { if (is_row) SSW_LAUNCH_PAIR(false, EPI_INV_E, false); else SSW_LAUNCH_PAIR(true, EPI_INV_E, false); }
        else { if (is_row) SSW_LAUNCH_PAIR(false, EPI_INV_O, true); else SSW_LAUNCH_PAIR(true, EPI_INV_O, true); }
    }
#undef SSW_LAUNCH_PAIR
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

}  // namespace ssw
