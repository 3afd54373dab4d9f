// Operand-ready even/odd-folded basis GEMMs in f32 (SSW_PRECISION_F32): the f32 twin of
// dct_pair_f64.hip.  v_mfma_f32_32x32x2_f32 shows the same issue behaviour as the f64 MFMA
// (tools/mfma_peak.hip: 155 TFLOP/s alone, 145 with one VALU instruction per MFMA, 135 with two), so the
// structure is the same: f32 operand planes written once per pass by the pre-passes of
// dct_pair_prep.hip (folded / split, k-blocked in 64-byte pieces = 16 floats), half bases in the same
// layout, and a GEMM loop of buffer_load -> ds_write -> ds_read -> MFMA with scalar address arithmetic.
// One VALU burst remains: every 256 k the fma chains are folded into a second accumulator set and
// restarted (the two-level accumulation of dct_folded.hip: it keeps the error on the DC scale at
// ~3e-7 instead of growing with the length of the sum) -- 64 adds per 512 MFMAs.
//
//   acc1[x][y] = sum_k X1[x][k] Y1[y][k],   acc2[x][y] = sum_k X2[x][k] Y2[y][k]
//
// Block: 256 threads = 4 waves as 2 x 2; block tile 128 lines x 64 pairs x 2 products; k-step 16;
// per wave 2 x 1 MFMA 32x32 tiles per product = 64 accumulator + 64 total registers; LDS 48 KB
// double-buffered, one barrier per k-step, 2 blocks per CU.  LDS rows are 64 bytes; the 16-byte chunk
// c of row r sits at slot c ^ ((r >> 2) & 3): conflict-free ds_read_b128 for the 32x32 lane map and
// for the staging writes.  Lane l: lr = l & 31 (line / pair inside a tile), lh = l >> 5; in half-step
// h lane group lh reads chunk 2 h + lh and supplies k = 4 (2 h + lh) + j at MFMA j (same assignment on
// both operands: a valid summation order).
#include "dct_pair_common.hpp"

#include <type_traits>

namespace ssw {

constexpr int QT = 256;
constexpr int QBK = 16;
constexpr int QCHUNK = 16;              // k-steps per accumulation chunk (256 k)
typedef PairOutT<float> PairOutF;

// SUB only names the instance (launches that serve a deeper folding level show up separately in profiles)
template <bool COLS, int EPI, bool SAMEX, int SUB = 0>
__global__ __launch_bounds__(QT, 2) void pair_gemm_f32_kernel(
    const float* __restrict__ X1g, const float* __restrict__ X2g, const float* __restrict__ Y1g,
    const float* __restrict__ Y2g, PairOutF po, unsigned L /*lines*/, unsigned NP /*pairs*/,
    unsigned Kp, unsigned yrows /*lines of the basis planes*/, unsigned tiles_m, unsigned tiles_n, Epilogue ep) {
    constexpr int NX = SAMEX ? 1 : 2;
    constexpr int BM = 128, BN = 64, XQ = 2;
    __shared__ __attribute__((aligned(16))) float sX[2][NX][BM * QBK];    // [buffer][product]
    __shared__ __attribute__((aligned(16))) float sY[2][2][BN * QBK];

    unsigned tm, tn;
    tile_of_block(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
    const unsigned m0 = tm * BM, p0 = tn * BN;
    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63, wave = tid >> 6;
    const unsigned lr = lane & 31, lh = lane >> 5;

    // staging: line = tid / 4 (+ 64 q), 16-byte chunk = tid % 4
    const unsigned srow = tid >> 2, sc = tid & 3;
    unsigned xoff[XQ];
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
        unsigned r = m0 + srow + 64 * q;
        r = r < L ? r : L - 1;
        xoff[q] = (r - m0) * 64u + sc * 16u;
    }
    unsigned yr = p0 + srow;
    yr = yr < NP ? yr : NP - 1;
    const unsigned yoff = (yr - p0) * 64u + sc * 16u;
    const __amdgpu_buffer_rsrc_t x1r = __builtin_amdgcn_make_buffer_rsrc((void*)(X1g + (size_t)m0 * 16), 0, 0xFFFFFFFFu, 0x00020000);
    const __amdgpu_buffer_rsrc_t x2r = __builtin_amdgcn_make_buffer_rsrc((void*)(X2g + (size_t)m0 * 16), 0, 0xFFFFFFFFu, 0x00020000);
    const __amdgpu_buffer_rsrc_t y1r = __builtin_amdgcn_make_buffer_rsrc((void*)(Y1g + (size_t)p0 * 16), 0, 0xFFFFFFFFu, 0x00020000);
    const __amdgpu_buffer_rsrc_t y2r = __builtin_amdgcn_make_buffer_rsrc((void*)(Y2g + (size_t)p0 * 16), 0, 0xFFFFFFFFu, 0x00020000);
    const unsigned xstep = L * 64u, ystep = yrows * 64u;

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
    const unsigned st = srow * QBK + 4 * (sc ^ ((srow >> 2) & 3));
    auto lstore = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            *reinterpret_cast<u32x4*>(&sX[buf][0][st + 64 * q * QBK]) = rx1[q];
            if (!SAMEX) *reinterpret_cast<u32x4*>(&sX[buf][NX - 1][st + 64 * q * QBK]) = rx2[q];
        }
        *reinterpret_cast<u32x4*>(&sY[buf][0][st]) = ry1;
        *reinterpret_cast<u32x4*>(&sY[buf][1][st]) = ry2;
    };
    // The wave grid is 2 x 2 (each wave 64 lines x 32 pairs: NI = 2 line tiles) unless the tile holds at
    // most 32 valid pairs, where it is 4 x 1 (each wave 32 lines x 32 pairs: NI = 1) and the tile takes
    // half the MFMAs instead of computing padding.
    auto run = [&](auto nic) {
    constexpr int NI = decltype(nic)::value;
    const unsigned wm = NI == 2 ? (wave >> 1) * 64 : wave * 32, wn = NI == 2 ? (wave & 1) * 32 : 0;
    f32x16 acc1[NI], acc2[NI], tot1[NI], tot2[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc1[i][r] = 0.f; acc2[i][r] = 0.f; tot1[i][r] = 0.f; tot2[i][r] = 0.f; }

    const unsigned fsw = (lr >> 2) & 3;
    unsigned rdx[2], rdy[2];
#pragma unroll
    for (int sh = 0; sh < 2; ++sh) {
        rdx[sh] = (wm + lr) * QBK + 4 * ((2 * sh + lh) ^ fsw);
        rdy[sh] = (wn + lr) * QBK + 4 * ((2 * sh + lh) ^ fsw);
    }
    struct Frag { f32x4 x1[NI], x2[NI], y1, y2; };
    auto fread = [&](auto bufc, auto shc, Frag& f) {
        constexpr int cur = decltype(bufc)::value;
        constexpr int sh = decltype(shc)::value;
        f.y1 = *reinterpret_cast<const f32x4*>(&sY[cur][0][rdy[sh]]);
        f.y2 = *reinterpret_cast<const f32x4*>(&sY[cur][1][rdy[sh]]);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            f.x1[i] = *reinterpret_cast<const f32x4*>(&sX[cur][0][rdx[sh] + 32 * i * QBK]);
            if (!SAMEX) f.x2[i] = *reinterpret_cast<const f32x4*>(&sX[cur][NX - 1][rdx[sh] + 32 * i * QBK]);
        }
    };
    auto fmma = [&](const Frag& f) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float xa = f.x1[i][j], xb = SAMEX ? f.x1[i][j] : f.x2[i][j];
                if (!COLS) {      // D[row = line][col = pair]
                    acc1[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, f.y1[j], acc1[i], 0, 0, 0);
                    acc2[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(xb, f.y2[j], acc2[i], 0, 0, 0);
                } else {          // D[row = pair][col = line]: image columns along the lanes
                    acc1[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.y1[j], xa, acc1[i], 0, 0, 0);
                    acc2[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.y2[j], xb, acc2[i], 0, 0, 0);
                }
            }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;

    // Same half-step-shifted software pipeline as the f64 kernel: 16 MFMAs per half-step; the next
    // half-step's fragments are read one ds_read_b128 per MFMA behind the first MFMAs, the staged tile is
    // written one ds_write_b128 per MFMA in the second half of a step, its loads issued a step earlier.
    constexpr int NMF = 8 * NI;                                 // MFMAs per half-step
    constexpr int NRD = 2 + NI * NX;                            // LDS reads per half-step
    constexpr int NLD = XQ * NX + 2;                            // staged 16-byte loads (= LDS writes) per thread
    auto interleave = [&](auto storec) {
        constexpr bool STORE = decltype(storec)::value != 0;
#pragma unroll
        for (int i = 0; i < NRD; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (STORE) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD - NLD > 0 ? NMF - NRD - NLD : 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, NLD, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    const unsigned nk = Kp / QBK;          // even and >= 2: Kp is a multiple of 32
    Frag fa, fb;
    // k-steps 0 and 1 are requested together (one memory latency at the start of a tile, not two): step 1
    // waits in registers that the fragments will use later
    gload(1);
    u32x4 nx1[XQ], nx2[XQ];
#pragma unroll
    for (int q = 0; q < XQ; ++q) { nx1[q] = rx1[q]; if (!SAMEX) nx2[q] = rx2[q]; }
    const u32x4 ny1 = ry1, ny2 = ry2;
    gload(0);
    lstore(B0{});
    __syncthreads();
#pragma unroll
    for (int q = 0; q < XQ; ++q) { rx1[q] = nx1[q]; if (!SAMEX) rx2[q] = nx2[q]; }
    ry1 = ny1; ry2 = ny2;
    fread(B0{}, B0{}, fa);
    auto step = [&](auto curc, auto nxtc, unsigned t) {      // full step t on buffer CUR: needs t + 2 < nk
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
        if (((t + 2) & (QCHUNK - 1)) == 0) {   // every MFMA of steps <= t + 1 has been issued
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                tot1[i] += acc1[i]; tot2[i] += acc2[i];
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc1[i][r] = 0.f; acc2[i][r] = 0.f; }
            }
        }
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
#pragma unroll
    for (int i = 0; i < NI; ++i) { acc1[i] = tot1[i] + acc1[i]; acc2[i] = tot2[i] + acc2[i]; }

    // C/D map of 32x32x2 f32: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const unsigned n = po.n, W = po.W, H = po.H;
    // element offsets idx * es stay below 2^32 (W * H < 2^32: indices are u32 throughout)
    auto emit = [&](float* lp, float* tp, unsigned es, unsigned pair, float a1, float a2) {
        if (EPI == EPI_FWD || EPI == EPI_FWD_ADJ) {
            const unsigned i1 = po.c1 + po.cs * pair, i2 = po.c2 + po.cs * pair;
            if (EPI == EPI_FWD_ADJ) {
                const f32x2 v = {apply_epilogue(ep, a1, i1), apply_epilogue(ep, a2, i2)};
                *reinterpret_cast<f32x2*>(lp + i1) = v;
            } else {
                lp[i1 * es] = apply_epilogue(ep, a1, i1);
                lp[i2 * es] = apply_epilogue(ep, a2, i2);
            }
        } else if (EPI == EPI_INV) {
            lp[pair * es] = apply_epilogue(ep, a1 + a2, pair);
            lp[(n - 1 - pair) * es] = apply_epilogue(ep, a1 - a2, n - 1 - pair);
        } else if (EPI == EPI_INV_E) {
            tp[pair * es] = a1 + a2;
            tp[(n / 2 - 1 - pair) * es] = a1 - a2;
        } else {
            const unsigned n1 = pair, n2 = pair + n / 4;
            const float e1 = tp[n1 * es], e2 = tp[n2 * es];
            lp[n1 * es] = apply_epilogue(ep, e1 + a1, n1);
            lp[(n - 1 - n1) * es] = apply_epilogue(ep, e1 - a1, n - 1 - n1);
            lp[n2 * es] = apply_epilogue(ep, e2 + a2, n2);
            lp[(n - 1 - n2) * es] = apply_epilogue(ep, e2 - a2, n - 1 - n2);
        }
    };
    if (!COLS) {
        const unsigned pair = p0 + wn + lr;
        if (pair < NP) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row >= L) continue;
                    emit(po.out + (size_t)row * W, po.tmp + (size_t)row * (n / 2), 1, pair, acc1[i][r], acc2[i][r]);
                }
        }
    } else {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const unsigned line = m0 + wm + 32 * i + lr;       // = frame * W + column
            if (line >= L) continue;
            const unsigned z = line / W, col = line - z * W;
            float* lp = po.out + (size_t)z * H * W + col;
            float* tp = po.tmp + (size_t)z * (n / 2) * W + col;
            if (EPI == EPI_INV_O_RGB) {
                // Writer::result in the last pass: per pair, its 4 output rows -- the even-half values, then I and Q,
                // are loaded together; offsets from the frame's pixel (row 0, col) address I, Q and RGB alike
                const size_t base = (size_t)z * H * W + col;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    float e1[2], e2[2];
                    bool okp[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int rr = r + q;
                        const unsigned pair = p0 + wn + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                        okp[q] = pair < NP;
                        const unsigned pc = okp[q] ? pair : 0;
                        e1[q] = tp[(size_t)pc * W];
                        e2[q] = tp[(size_t)(pc + n / 4) * W];
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int rr = r + q;
                        const unsigned pair = p0 + wn + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                        const unsigned n1 = okp[q] ? pair : 0, n2 = n1 + n / 4;
                        const float a1 = acc1[i][rr], a2 = acc2[i][rr];
                        const unsigned idx[4] = {n1, n - 1 - n1, n2, n - 1 - n2};
                        const float v[4] = {e1[q] + a1, e1[q] - a1, e2[q] + a2, e2[q] - a2};
                        unsigned px[4];
                        float yv[4];
#pragma unroll
                        for (int o = 0; o < 4; ++o) { px[o] = idx[o] * W; yv[o] = apply_epilogue(ep, v[o], idx[o]); }
                        pair_store_rgb_batch<float, 4>(po, base, px, yv, okp[q]);
                    }
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned pair = p0 + wn + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (pair >= NP) continue;
                emit(lp, tp, W, pair, acc1[i][r], acc2[i][r]);
            }
        }
    }
    };
    if (NP - p0 <= 32) run(std::integral_constant<int, 1>{});
    else               run(std::integral_constant<int, 2>{});
}

// kind: 0 one folding level; 1 / 2 the even / odd half of two levels (see launch_dct_pair_gemm_f64)
int launch_dct_pair_gemm_f32(hipStream_t st, bool is_row, bool inverse, int kind, int sub, const float* x1, const float* x2,
                             const float* y1, const float* y2, float* out, float* tmp, size_t n_frames, size_t w,
                             size_t h, Epilogue ep, const RgbSink* sink) {
    if (n_frames == 0) return SSW_OK;
    if (w > 0xFFFFFFull || h > 0xFFFFFFull) return SSW_ERR_BAD_DIMS;
    const size_t lines = is_row ? n_frames * h : n_frames * w;
    const size_t len = is_row ? w : h;
    if (lines > 0xFFFFFFFFull) return SSW_ERR_BAD_DIMS;
    const unsigned L = (unsigned)lines;
    if (sub < 0 || sub > 8 || (sub > 0 && (inverse || kind == 0))) return SSW_ERR_BAD_ARG;
    const size_t leff = len >> sub;                           // length of the (sub-)transform this launch serves
    const unsigned fs = 1u << sub;                            // its frequencies in units of the full transform's
    const unsigned NP = (unsigned)(kind == 0 ? leff / 2 : leff / 4);
    const unsigned Kp = (unsigned)(kind == 1 ? pair_kpad<float>(leff / 2) : pair_kpad<float>(leff));
    const unsigned tiles_m = (L + 127) / 128, tiles_n = (NP + 63) / 64;
    const unsigned long long nblk = (unsigned long long)tiles_m * tiles_n;
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    PairOutF po{out, tmp, (unsigned)w, (unsigned)h, (unsigned)len, 0, 1, 2};
    if (kind == 1) { po.c1 = 0; po.c2 = 2 * fs; po.cs = 4 * fs; }
    if (kind == 2) { po.c1 = fs; po.c2 = fs + 2 * fs * NP; po.cs = 2 * fs; }
    const unsigned yrows = kind == 2 ? 2 * NP : NP;          // lines of the basis plane(s)
    if ((unsigned long long)Kp * L * 4 > 0xFFFFFFFFull) return SSW_ERR_BAD_DIMS;   // scalar k-block offsets are 32-bit
#define SSW_LAUNCH_PAIRF(COLS, EPI, SAMEX) do { \
        if (sub == 0) pair_gemm_f32_kernel<COLS, EPI, SAMEX, 0><<<(unsigned)nblk, QT, 0, st>>>(x1, x2, y1, y2, po, L, NP, Kp, yrows, tiles_m, tiles_n, ep); \
        else          pair_gemm_f32_kernel<COLS, EPI, SAMEX, 1><<<(unsigned)nblk, QT, 0, st>>>(x1, x2, y1, y2, po, L, NP, Kp, yrows, tiles_m, tiles_n, ep); \
    } while (0)
    if (!inverse) {
        if (kind == 0) { if (is_row) SSW_LAUNCH_PAIRF(false, EPI_FWD_ADJ, false); else SSW_LAUNCH_PAIRF(true, EPI_FWD, false); }
        else if (kind == 1) { if (is_row) SSW_LAUNCH_PAIRF(false, EPI_FWD, false); else SSW_LAUNCH_PAIRF(true, EPI_FWD, false); }
        else { if (is_row) SSW_LAUNCH_PAIRF(false, EPI_FWD, true); else SSW_LAUNCH_PAIRF(true, EPI_FWD, true); }
    } else {
        if (kind == 0) { if (is_row) SSW_LAUNCH_PAIRF(false, EPI_INV, false); else SSW_LAUNCH_PAIRF(true, EPI_INV, false); }
        else if (kind == 1) { if (is_row) SSW_LAUNCH_PAIRF(false, EPI_INV_E, false); else SSW_LAUNCH_PAIRF(true, EPI_INV_E, false); }
        else if (sink && sink->rgb) {      // last pass of Writer::result: colour conversion in the epilogue
            if (is_row || sub != 0) return SSW_ERR_BAD_ARG;
            po.iq_i = sink->iq_i; po.iq_q = sink->iq_q; po.rgb = sink->rgb; po.rgb_u8 = sink->u8 ? 1u : 0u;
            SSW_LAUNCH_PAIRF(true, EPI_INV_O_RGB, true);
        }
        else { if (is_row) SSW_LAUNCH_PAIRF(false, EPI_INV_O, true); else SSW_LAUNCH_PAIRF(true, EPI_INV_O, true); }
    }
#undef SSW_LAUNCH_PAIRF
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// Forward row pass restricted to a gathered set of frequencies (pruned derived transform, prune.hip): one
// shared image operand `x` (k-blocked, Kp wide) against a gathered half basis `y` of `cap` rows (k-blocked,
// [Kp / 16][cap][16]); row j of y produces compact output column off + j of `out` (row stride out_stride).
// Same kernel, operands and k order as the full transform's launches -> bit-identical values.
int launch_dct_pair_gemm_rows_subset_f32(hipStream_t st, const float* x, const float* y, unsigned cap, unsigned Kp, float* out,
                                         unsigned out_stride, unsigned off, size_t lines) {
    if (lines == 0 || cap == 0) return SSW_OK;
    if (lines > 0xFFFFFFFFull || (cap & 1)) return SSW_ERR_BAD_DIMS;
    const unsigned L = (unsigned)lines, NP = cap / 2;
    const unsigned tiles_m = (L + 127) / 128, tiles_n = (NP + 63) / 64;
    const unsigned long long nblk = (unsigned long long)tiles_m * tiles_n;
    if (nblk > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    if ((unsigned long long)Kp * L * sizeof(float) > 0xFFFFFFFFull) return SSW_ERR_BAD_DIMS;
    PairOutF po{out, nullptr, out_stride, 0, 0, off, off + NP, 1};
    const Epilogue ep{1.f, 1.f};
    pair_gemm_f32_kernel<false, EPI_FWD, true, 2><<<(unsigned)nblk, QT, 0, st>>>(x, x, y, y + (size_t)NP * 16, po, L, NP, Kp, cap, tiles_m, tiles_n, ep);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

}  // namespace ssw
