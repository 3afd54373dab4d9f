// Coefficient ordering (top-k), embed, extract and similarity kernels.
//
// Ordering: obtain_indices_by_function (/root/reference/src/algorithm.rs:200-210) does a full
// stable descending sort of all W*H-1 non-DC coefficients with a boxed comparator, but only
// the first k (= mark length) entries are ever consumed (:396, :556-557).  Here the first k
// entries are produced directly.  Every coefficient gets a 64-bit composite key
//     (sortable(key_f32) << 32) | ~index
// -- key_f32 exactly as the comparators compute it (:214-280), sortable() = f32::total_cmp order as
// an unsigned integer, ~index so that equal keys rank lower index first (stable sort of an
// index-ascending list).  Composite keys are unique: no ties are left, and "the first k of the
// reference's list" = "the k largest composite keys, descending".
//
// Per frame (HBM-bound, ~1.02 reads of the plane, decided entirely on the device):
//   1. sample pass   -- one pseudo-randomly placed element out of every 64 goes into a 2048-bin
//                       histogram of the top 11 key bits
//   2. sample find   -- the digit d whose upper tail holds ~3k/64 samples: a conservative threshold
//                       (expected 3k survivors; fewer than k = 1000 is a -4.6 sigma event, and then step 4
//                       falls back to the exact whole-plane select)
//   3. compaction    -- the one full pass: every coefficient whose top digit >= d is appended to a
//                       candidate list (two 16-B loads in flight per thread, rare atomics; for the
//                       default energy ordering the test is one multiply and one float compare)
//   4. finish        -- one block per frame: MSD radix select (11-bit digits, LDS histogram, wave-shuffle
//                       scan) on the candidates until everything from the chosen bin upwards fits the
//                       1024-entry sort buffer (one or two rounds), gather, bitonic sort with one key per
//                       thread (shuffles inside a wave, LDS only for strides >= 64), emit ~low32.
//   Degenerate data (fewer than k candidates, or more than the candidate buffer holds: massive
//   ties, constant planes) is handled in the same finish kernel by running the exact select over
//   the whole plane instead of the candidate list -- slow but exact, and never taken by images.
#include "ssw_internal.hpp"

#include <atomic>

#include <type_traits>

namespace ssw {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DIGIT_BITS = 11;
constexpr int NBINS = 1 << DIGIT_BITS;          // 2048
constexpr size_t MAX_K = 16384;                 // 128 KiB of LDS for the in-block sort
// sampling rate of the threshold estimate: one element in 64; for long marks one in 128 (k >= 4096) or 256
// (k >= 8192) -- the sample pass pulls a cache line per 16-byte quad it reads, i.e. 1/8 of the plane at rate 1/64
// (98 us of a 1.07 ms selection on 32 8K frames with k = 10000).  The four members of a quad are neighbouring
// coefficients of a smooth spectrum, i.e. strongly correlated: count QUADS when judging the noise of the estimate
// (ADVICE r3).  The tail of the sample holds 3k / stride >= 96 elements = 24 quads at both thresholds; the exact
// whole-plane fallback needs that tail to overestimate the true density threefold: below 1e-5 per frame.
constexpr unsigned SAMPLE_STRIDE = 64, SAMPLE_STRIDE_MID = 128, SAMPLE_STRIDE_LONG = 256;
static inline unsigned sample_stride_for(size_t k) { return k >= 8192 ? SAMPLE_STRIDE_LONG : k >= 4096 ? SAMPLE_STRIDE_MID : SAMPLE_STRIDE; }
constexpr unsigned FINISH_THREADS = 1024;

size_t select_max_k() { return MAX_K; }
size_t select_cand_capacity(size_t k) { size_t c = 16 * k; return c < 65536 ? 65536 : c; }

struct KeyParams {
    int ordering;
    unsigned w;
    float s[2][2];      // ortho scaling [first_row][first_column], src/algorithm.rs:240-266
};

// f32::total_cmp order as an unsigned integer (larger == Greater)
__device__ inline uint32_t sortable(float v) {
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ inline uint32_t key_bits(const KeyParams& kp, uint32_t index, float value) {
    float key;
    if (kp.ordering == SSW_ORDER_ENERGY) {
        key = value * value;                                         // :214-221
    } else {
        const float scaled = kp.s[index < kp.w][(index % kp.w) == 0] * value;   // :252-266
        key = (kp.ordering == SSW_ORDER_ENERGY_ORTHOGONAL) ? scaled * scaled : scaled;   // :178-187
    }
    return sortable(key);
}
__device__ inline uint64_t composite_key(const KeyParams& kp, uint32_t index, float value) {
    return ((uint64_t)key_bits(kp, index, value) << 32) | (uint32_t)(~index);
}

__device__ inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// per-frame control block: [0] threshold digit, [1] candidate count ([2], [3] spare).  ctrl and the sample
// histogram are zero when a selection starts: the workspace is zero-filled at allocation and the
// finish kernel re-zeroes what it consumed.

// 1. sample pass: one pseudo-randomly placed 16-B quad out of every 256 elements (rate 1/64)
// First of 256 partial sums (LDS) at which the running total reaches `need`, found by one wave without
// barriers: lane l owns partials 4l..4l+3, a shuffle scan gives its exclusive prefix.  Returns the owner
// in [0, 256) and its exclusive prefix through `excl_out`; 256 if the total stays below `need`.
__device__ inline unsigned wave_find_owner(const uint32_t* part, uint64_t need, uint64_t* excl_out) {
    const unsigned lane = threadIdx.x & 63;
    const uint32_t a0 = part[4 * lane], a1 = part[4 * lane + 1], a2 = part[4 * lane + 2], a3 = part[4 * lane + 3];
    const uint32_t s = a0 + a1 + a2 + a3;
    uint32_t incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= (unsigned)d) incl += up;
    }
    uint64_t run = incl - s;
    unsigned owner = 256;
    uint64_t ex = 0;
    const uint32_t a[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (owner == 256 && run < need && run + a[j] >= need) { owner = 4 * lane + j; ex = run; }
        run += a[j];
    }
    const unsigned long long found = __ballot(owner != 256);      // at most one lane
    if (!found) return 256;
    const int src = __ffsll((long long)found) - 1;
    owner = __shfl(owner, src, 64);
    const uint32_t ex_lo = __shfl((uint32_t)ex, src, 64), ex_hi = __shfl((uint32_t)(ex >> 32), src, 64);
    *excl_out = ((uint64_t)ex_hi << 32) | ex_lo;
    return owner;
}

// threshold digit: smallest upper tail of the sample histogram holding >= m samples (256 threads)
__device__ inline void find_threshold_digit(const uint32_t* h, uint32_t m, uint32_t* part /*LDS, 256*/, uint32_t* thr_out) {
    const int t = threadIdx.x;
    uint32_t local[8], sum = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { local[e] = h[NBINS - 1 - (8 * t + e)]; sum += local[e]; }
    part[t] = sum;
    __shared__ unsigned owner_s;
    __shared__ uint32_t excl_s;
    __syncthreads();
    if (t < 64) {
        uint64_t ex = 0;
        const unsigned o = wave_find_owner(part, m, &ex);
        if (t == 0) { owner_s = o; excl_s = (uint32_t)ex; }
    }
    __syncthreads();
    if (t == 0 && owner_s == 256) *thr_out = 0;                   // fewer than m samples: keep everything
    if ((unsigned)t == owner_s) {
        uint32_t run = excl_s;
        int e = 0;
        for (; e < 7; ++e) {
            if (run + local[e] >= m) break;
            run += local[e];
        }
        *thr_out = (uint32_t)(NBINS - 1 - (8 * t + e));
    }
}

__global__ __launch_bounds__(256) void select_sample_kernel(const float* __restrict__ coef, size_t plane_len,
                                                            KeyParams kp, uint32_t* __restrict__ hist, unsigned sample_stride) {
    __shared__ uint32_t lh[NBINS];
    const size_t f = blockIdx.y;
    for (int i = threadIdx.x; i < NBINS; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const float* __restrict__ c = coef + f * plane_len;
    const unsigned GROUP = sample_stride * 4;                            // one quad out of GROUP / 4 (a power of two)
    const size_t groups = (plane_len + GROUP - 1) / GROUP;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const bool vec = (plane_len % 4 == 0) && ((reinterpret_cast<uintptr_t>(c) & 15) == 0);
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < groups; g += stride) {
        const size_t j0 = g * GROUP + 4 * (mix32((uint32_t)g * 0x9E3779B1u + (uint32_t)f) & (GROUP / 4 - 1));
        if (j0 >= plane_len) continue;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        unsigned cnt = 0;
        if (vec) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(c + j0);
            v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3]; cnt = 4;
        } else {
            for (; cnt < 4 && j0 + cnt < plane_len; ++cnt) v[cnt] = c[j0 + cnt];
        }
        for (unsigned e = 0; e < cnt; ++e)
            if (j0 + e != 0) atomicAdd(&lh[key_bits(kp, (uint32_t)(j0 + e), v[e]) >> (32 - DIGIT_BITS)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NBINS; i += blockDim.x) {
        const uint32_t v = lh[i];
        if (v) atomicAdd(&hist[f * NBINS + i], v);
    }
}

// 3. the one full pass over the plane: append every coefficient whose top digit >= threshold.
// ENERGY ordering (the default; keys c*c >= 0, so digit order = float order): the test is one multiply and
// one float compare against the threshold digit's lower edge, two 16-byte loads in flight per thread.
template <bool ENERGY>
__global__ __launch_bounds__(256) void select_compact_kernel(const float* __restrict__ coef, size_t plane_len,
                                                             KeyParams kp, uint32_t* __restrict__ ctrl,
                                                             const uint32_t* __restrict__ hist, uint32_t m,
                                                             uint64_t* __restrict__ cand, size_t cap) {
    const size_t f = blockIdx.y;
    // 2. the threshold digit, from the frame's sample histogram: every block works it out for itself (8 KB of
    // L2-resident counts, one wave scan) instead of waiting for a separate one-block-per-frame launch
    __shared__ uint32_t part[256];
    __shared__ uint32_t thr_s;
    find_threshold_digit(hist + f * NBINS, m, part, &thr_s);
    __syncthreads();
    const uint32_t thr = thr_s;
    uint32_t* count = &ctrl[f * 4 + 1];
    const float* __restrict__ c = coef + f * plane_len;
    uint64_t* __restrict__ out = cand + f * cap;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // lower edge of digit `thr` as a float (digits below 2^(DIGIT_BITS-1) hold negative keys: keep everything)
    const float edge = thr > (1u << (DIGIT_BITS - 1)) ? __uint_as_float((thr << (32 - DIGIT_BITS)) & 0x7FFFFFFFu) : 0.0f;
    auto append = [&](size_t j, uint32_t kb) {
        const uint32_t pos = atomicAdd(count, 1u);
        if (pos < cap) out[pos] = ((uint64_t)kb << 32) | (uint32_t)(~(uint32_t)j);
    };
    auto consider = [&](size_t j, float v) {
        if (ENERGY) {
            const float key = v * v;                                     // :214-221
            if (!(key < edge) && j != 0) append(j, sortable(key));      // NaN keys pass, like in the digit test
        } else {
            const uint32_t kb = key_bits(kp, (uint32_t)j, v);
            if ((kb >> (32 - DIGIT_BITS)) >= thr && j != 0) append(j, kb);
        }
    };
    if ((plane_len % 4 == 0) && ((reinterpret_cast<uintptr_t>(c) & 15) == 0)) {
        const size_t nquad = plane_len / 4;
        size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
        for (; q + stride < nquad; q += 2 * stride) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(c + 4 * q);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(c + 4 * (q + stride));
#pragma unroll
            for (int e = 0; e < 4; ++e) consider(4 * q + e, v0[e]);
#pragma unroll
            for (int e = 0; e < 4; ++e) consider(4 * (q + stride) + e, v1[e]);
        }
        if (q < nquad) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(c + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) consider(4 * q + e, v[e]);
        }
    } else {
        for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < plane_len; j += stride) consider(j, c[j]);
    }
}

// 4. finish: exact select + sort inside one block.  `Src` yields the composite key of item i.
struct CandSource {
    const uint64_t* cand;
    __device__ uint64_t operator()(size_t i) const { return cand[i]; }
};
struct PlaneSource {                    // item i = coefficient i + 1 (DC skipped, :204)
    const float* c;
    KeyParams kp;
    __device__ uint64_t operator()(size_t i) const { return composite_key(kp, (uint32_t)(i + 1), c[i + 1]); }
};

// Stable LSD radix sort, descending, of lbuf[0 .. rows * blockDim.x) in place: 8 passes of 8 bits.  (r4: the finish of
// marks longer than 1024 entries.  The bitonic network it replaces moves every key through LDS ~105 times -- 16384
// 64-bit keys: 55 MB of LDS traffic, 176 us on one CU, LDS-bandwidth bound; a counting pass reads and writes each key
// once: 8 x 0.26 MB.)  Wave w owns the keys [w rows 64, (w + 1) rows 64), lane l of row r the key (w rows + r) 64 + l, all
// of a thread's keys in registers between the load and the scatter of a pass, so one buffer suffices.  Per row the lanes
// of equal digit find each other with eight ballots; the lowest of them adds their number to the wave's counter of that
// digit (lcount[wave][digit], 16-bit: <= 1024 keys per wave), the others take their place behind the running count.
// blockDim.x == 1024 (16 waves), rows <= 16.
__device__ void block_radix_sort_desc(uint64_t* lbuf, unsigned rows, uint16_t* lcount, uint32_t* lscan, int first_pass) {
    constexpr int MAXR = 16, NW = FINISH_THREADS / 64;
    const unsigned tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint64_t* mine = lbuf + (size_t)w * rows * 64 + lane;         // + 64 r
    uint16_t* mycount = lcount + w * 256;
#pragma unroll 1
    for (int pass = first_pass; pass < 8; ++pass) {
        const int shift = 8 * pass;
        unsigned loc[MAXR];                                             // digit << 16 | place behind the wave's running count
        for (unsigned i = tid; i < NW * 256; i += FINISH_THREADS) lcount[i] = 0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            if ((unsigned)r >= rows) continue;                          // block-uniform
            const unsigned d = 255u - (unsigned)((mine[64 * r] >> shift) & 255ull);
            unsigned long long mask = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const unsigned long long bal = __ballot((d >> b) & 1u);
                mask &= ((d >> b) & 1u) ? bal : ~bal;
            }
            const unsigned rank = (unsigned)__popcll(mask & below);
            const unsigned base = mycount[d];                           // read by every lane of the group, then the leader adds
            if (rank == 0) mycount[d] = (uint16_t)(base + (unsigned)__popcll(mask));
            loc[r] = (d << 16) | (base + rank);
        }
        __syncthreads();
        // per digit: exclusive offsets of the waves, the digit's total
        if (tid < 256) {
            unsigned sum = 0;
            for (int w2 = 0; w2 < NW; ++w2) { const unsigned c = lcount[w2 * 256 + tid]; lcount[w2 * 256 + tid] = (uint16_t)sum; sum += c; }
            lscan[tid] = sum;
        }
        __syncthreads();
        // exclusive scan of the 256 totals: one wave, four digits per lane
        if (tid < 64) {
            unsigned s4[4], run = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) { s4[e] = lscan[4 * tid + e]; run += s4[e]; }
            unsigned incl = run;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned up = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += up; }
            unsigned ex = incl - run;
#pragma unroll
            for (int e = 0; e < 4; ++e) { lscan[4 * tid + e] = ex; ex += s4[e]; }
        }
        __syncthreads();
        uint64_t key[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            if ((unsigned)r >= rows) continue;
            const unsigned d = loc[r] >> 16;
            loc[r] = (loc[r] & 0xFFFFu) + lscan[d] + mycount[d];
            key[r] = mine[64 * r];
        }
        __syncthreads();                                               // every key of the pass is in a register
#pragma unroll
        for (int r = 0; r < MAXR; ++r)
            if ((unsigned)r < rows) lbuf[loc[r]] = key[r];
        __syncthreads();
    }
}

template <class Src>
__device__ void block_select_sort(const Src& src, size_t n, size_t k, unsigned n_pow2, uint32_t* lhist,
                                  uint32_t* lpart, uint64_t* lbuf, uint64_t* lstate,
                                  uint32_t* __restrict__ indices) {
    // lstate: [0] prefix, [1] bits decided, [2] need, [3] resolved, [4] gather counter, [5..6] scan scratch
    const unsigned tid = threadIdx.x;
    if (tid == 0) { lstate[0] = 0; lstate[1] = 0; lstate[2] = k; lstate[3] = (n == k) ? 1 : 0; lstate[4] = 0; }
    // r4: the walks over the items (select rounds, gather) keep UN loads per thread in flight -- a loop of one dependent
    // load per iteration took ~30 us per walk over the 30000 candidates of a k = 10000 mark (1 us per trip to L2 / HBM).
    // f(valid, key) is called for whole waves (the gather ballots).
    constexpr int UN = 8;
    auto walk = [&](auto&& f) {
        const size_t step = (size_t)UN * blockDim.x;
        for (size_t i0 = 0; i0 < n; i0 += step) {
            uint64_t v[UN];
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                const size_t i = i0 + tid + (size_t)j * blockDim.x;
                v[j] = i < n ? src(i) : 0ull;
            }
#pragma unroll
            for (int j = 0; j < UN; ++j) f(i0 + tid + (size_t)j * blockDim.x < n, v[j]);
        }
    };
    __syncthreads();
    while (!lstate[3]) {
        const uint64_t prefix = lstate[0];
        const int bits_done = (int)lstate[1];
        const uint64_t need = lstate[2];
        const int width = (64 - bits_done) < DIGIT_BITS ? (64 - bits_done) : DIGIT_BITS;
        const int shift = 64 - bits_done - width;
        const uint32_t mask = (1u << width) - 1u;
        for (unsigned i = tid; i < NBINS; i += blockDim.x) lhist[i] = 0;
        __syncthreads();
        // The first round sees every candidate in two or three bins (the top bits of keys above one threshold): lanes of
        // equal bin are counted with one atomic -- up to three distinct bins per wave step, the rest one by one -- instead of
        // 64 updates of one LDS word in a row
        walk([&](bool valid, uint64_t comp) {
            bool mine = valid && (bits_done == 0 || (comp >> (64 - bits_done)) == prefix);
            const uint32_t bin = (uint32_t)(comp >> shift) & mask;
            const unsigned lane = tid & 63u;
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                if (bits_done != 0) break;                              // later rounds spread over the bins: plain updates
                const unsigned long long todo = __ballot(mine);
                if (todo == 0) break;
                const int leader = __builtin_ctzll(todo);
                const uint32_t b = (uint32_t)__shfl((int)bin, leader, 64);
                const unsigned long long same = __ballot(mine && bin == b);
                if ((int)lane == leader) atomicAdd(&lhist[b], (uint32_t)__popcll(same));
                if (bin == b) mine = false;
            }
            if (mine) atomicAdd(&lhist[bin], 1u);
        });
        __syncthreads();
        // digit holding the need-th largest key: 256 threads own 8 digits each (counted from the
        // top), thread 0 scans the 256 partial sums, the owner resolves inside its 8 digits
        uint32_t local[8];
        if (tid < 256) {
            uint32_t sum = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) { local[e] = lhist[NBINS - 1 - (8 * tid + e)]; sum += local[e]; }
            lpart[tid] = sum;
        }
        __syncthreads();
        // owner of the need-th key among the 256 partials: one wave, shuffle scan, no barriers (a serial
        // scan by one thread cost ~7 us per round)
        if (tid < 64) {
            uint64_t ex = 0;
            unsigned o = wave_find_owner(lpart, need, &ex);
            if (o == 256) { o = 255; ex = 0; for (int i = 0; i < 255; ++i) ex += lpart[i]; }   // cannot happen: n >= need
            if (tid == 0) { lstate[5] = o; lstate[6] = ex; }
        }
        __syncthreads();
        if (tid == (unsigned)lstate[5]) {
            uint64_t run = lstate[6];
            int e = 0;
            for (; e < 7; ++e) {
                if (run + local[e] >= need) break;
                run += local[e];
            }
            const uint32_t d = (uint32_t)(NBINS - 1 - (8 * tid + e));
            const uint64_t new_need = need - run;
            lstate[0] = (prefix << width) | (uint64_t)(d & mask);
            lstate[1] = (uint64_t)(bits_done + width);
            lstate[2] = new_need;
            // done when the chosen bin is needed whole, when the key is exhausted, or -- the usual exit, after
            // one or two rounds instead of six -- when everything from this bin upwards fits the sort buffer
            // (the sort then orders the bin's surplus behind position k)
            lstate[3] = (local[e] == new_need || bits_done + width >= 64 || (k - new_need) + local[e] <= n_pow2) ? 1 : 0;
        }
        __syncthreads();
    }
    // gather the survivors: everything from the chosen bin upwards (>= k of them, <= n_pow2)
    {
        const uint64_t prefix = lstate[0];
        const int bits_done = (int)lstate[1];
        // one counter update per wave and step (the survivors of a wave take consecutive places), not one per survivor
        auto take = [&](bool in_range, uint64_t comp) {
            const uint64_t top = (bits_done == 0) ? 1 : (bits_done >= 64 ? comp : (comp >> (64 - bits_done)));
            const bool keep = in_range && (bits_done == 0 || top >= prefix);
            const unsigned long long m = __ballot(keep);
            if (m == 0) return;
            const unsigned lane = tid & 63u;
            unsigned base = 0;
            if (lane == (unsigned)__builtin_ctzll(m)) base = (unsigned)atomicAdd((unsigned long long*)&lstate[4], (unsigned long long)__popcll(m));
            base = __shfl(base, __builtin_ctzll(m), 64);
            const unsigned pos = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            if (keep && pos < n_pow2) lbuf[pos] = comp;
        };
        walk(take);
        __syncthreads();
        const unsigned gathered = (unsigned)lstate[4];            // k <= gathered <= n_pow2
        for (unsigned i = tid; i < n_pow2; i += blockDim.x)
            if (i >= gathered) lbuf[i] = 0ull;                    // 0 sorts behind every real key
        __syncthreads();
    }
    // bitonic sort, descending
    if (n_pow2 <= blockDim.x) {
        // one key per thread in a register: the 45 of 55 sweeps whose partner is within the wave use
        // shuffles (no LDS, no barrier); the ten with stride >= 64 exchange through LDS
        const unsigned N = blockDim.x;
        uint64_t v = tid < n_pow2 ? lbuf[tid] : 0ull;
        __syncthreads();
        for (unsigned size = 2; size <= N; size <<= 1) {
            const bool up = (tid & size) == 0;
            for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
                uint64_t other;
                if (stride >= 64) {
                    lbuf[tid] = v;
                    __syncthreads();
                    other = lbuf[tid ^ stride];
                    __syncthreads();
                } else {
                    const uint32_t lo = __shfl_xor((uint32_t)v, (int)stride, 64);
                    const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), (int)stride, 64);
                    other = ((uint64_t)hi << 32) | lo;
                }
                const bool lower = (tid & stride) == 0;
                const uint64_t mx = v > other ? v : other, mn = v > other ? other : v;
                v = (lower == up) ? mx : mn;
            }
        }
        if (tid < k) indices[tid] = ~(uint32_t)(v & 0xFFFFFFFFull);
        return;
    }
    // more keys than threads: a counting sort of the gathered keys (zero padded to whole rows of the block), the
    // histogram / partial-sum areas of the select rounds serving as its counters.  (r3: compare-exchanges through LDS,
    // 176 us for the k = 10000 finish of an 8K frame, kept below for blocks of another size.)
    if (blockDim.x == FINISH_THREADS && n_pow2 <= 16 * FINISH_THREADS) {
        const unsigned gathered = (unsigned)lstate[4] < n_pow2 ? (unsigned)lstate[4] : n_pow2;
        const unsigned rows = (gathered + FINISH_THREADS - 1) / FINISH_THREADS;
        block_radix_sort_desc(lbuf, rows, reinterpret_cast<uint16_t*>(lhist), lpart, 0);
        for (unsigned i = tid; i < k; i += blockDim.x) indices[i] = ~(uint32_t)(lbuf[i] & 0xFFFFFFFFull);
        return;
    }
    for (unsigned size = 2; size <= n_pow2; size <<= 1) {
        for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
            for (unsigned i = tid; i < n_pow2 / 2; i += blockDim.x) {
                const unsigned lo = 2 * i - (i & (stride - 1));
                const unsigned hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const uint64_t a = lbuf[lo], b = lbuf[hi];
                if ((a < b) == desc) { lbuf[lo] = b; lbuf[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (unsigned i = tid; i < k; i += blockDim.x) indices[i] = ~(uint32_t)(lbuf[i] & 0xFFFFFFFFull);
}

__global__ __launch_bounds__(FINISH_THREADS) void select_finish_kernel(
    const float* __restrict__ coef, size_t plane_len, KeyParams kp, uint32_t* __restrict__ ctrl,
    uint32_t* __restrict__ hist, const uint64_t* __restrict__ cand, size_t cap, size_t k, unsigned n_pow2,
    uint32_t* __restrict__ indices, uint32_t* __restrict__ fallbacks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t* lbuf = reinterpret_cast<uint64_t*>(smem_raw);                      // max(n_pow2, blockDim.x) entries
    uint64_t* lstate = lbuf + (n_pow2 > blockDim.x ? n_pow2 : blockDim.x);       // 8 entries
    uint32_t* lhist = reinterpret_cast<uint32_t*>(lstate + 8);                   // NBINS entries
    uint32_t* lpart = lhist + NBINS;                                             // 256 entries
    const size_t f = blockIdx.x;
    const uint32_t n = ctrl[f * 4 + 1];
    __syncthreads();
    for (unsigned i = threadIdx.x; i < NBINS; i += blockDim.x) hist[f * NBINS + i] = 0;   // ready for the next
    if (threadIdx.x == 0) { ctrl[f * 4 + 0] = 0; ctrl[f * 4 + 1] = 0; }                      // selection
    if (n >= k && n <= cap) {
        CandSource src{cand + f * cap};
        block_select_sort(src, n, k, n_pow2, lhist, lpart, lbuf, lstate, indices + f * k);
    } else {                                     // degenerate data: exact select over the whole plane
        if (threadIdx.x == 0 && fallbacks) atomicAdd(fallbacks, 1u);      // a latency cliff: visible in ssw_ctx_get_select_stats
        PlaneSource src{coef + f * plane_len, kp};
        block_select_sort(src, plane_len - 1, k, n_pow2, lhist, lpart, lbuf, lstate, indices + f * k);
    }
}

int launch_topk(hipStream_t st, const float* coef, size_t n_frames, size_t w, size_t h, int ordering,
                size_t k, const SelectWorkspace& ws, uint32_t* indices) {
    const size_t plane_len = w * h;
    if (n_frames == 0 || k == 0) return SSW_OK;
    if (plane_len > 0xFFFFFFFFull) return SSW_ERR_BAD_DIMS;
    if (k > plane_len - 1) return SSW_ERR_K_TOO_LARGE;
    if (k > MAX_K) return SSW_ERR_UNSUPPORTED;
    if (ws.frames < n_frames || ws.cap < select_cand_capacity(k)) return SSW_ERR_BAD_ARG;
    if (ordering < SSW_ORDER_ENERGY || ordering > SSW_ORDER_LEGACY) return SSW_ERR_UNSUPPORTED;

    KeyParams kp;
    kp.ordering = ordering;
    kp.w = (unsigned)w;
    {   // src/algorithm.rs:245-250, :255-265 evaluated in f32 on the host (IEEE, same bits)
        const float s_k0_w = sqrtf(1.0f / (4.0f * (float)w));
        const float s_k0_h = sqrtf(1.0f / (4.0f * (float)h));
        const float s_w = sqrtf(1.0f / (2.0f * (float)w));
        const float s_h = sqrtf(1.0f / (2.0f * (float)h));
        for (int fr = 0; fr < 2; ++fr)
            for (int fc = 0; fc < 2; ++fc) {
                volatile float sc = 1.0f;
                sc = sc * (fr ? s_k0_w : s_w);
                sc = sc * (fc ? s_k0_h : s_h);
                kp.s[fr][fc] = sc;
            }
    }
    const unsigned stride_s = sample_stride_for(k);
    const size_t groups = (plane_len + stride_s * 4 - 1) / (stride_s * 4);
    size_t sb = (groups + 256 * 2 - 1) / (256 * 2);
    if (sb < 1) sb = 1;
    if (sb > 256) sb = 256;

    // expected survivors ~ 3k (sampling noise at this depth is ~15 %: falling short of k = 1000 is a
    // -4.6 sigma event, and then the exact whole-plane select still answers); at least 32 samples deep
    uint32_t m = (uint32_t)((3 * k + stride_s - 1) / stride_s);
    if (m < 32) m = 32;
    select_sample_kernel<<<dim3((unsigned)sb, (unsigned)n_frames), 256, 0, st>>>(coef, plane_len, kp, ws.hist, stride_s);
    size_t bpf = (plane_len + 256 * 32 - 1) / (256 * 32);             // >= 32 elements per thread
    if (bpf < 1) bpf = 1;
    if (bpf > 2048) bpf = 2048;
    if (ordering == SSW_ORDER_ENERGY)
        select_compact_kernel<true><<<dim3((unsigned)bpf, (unsigned)n_frames), 256, 0, st>>>(coef, plane_len, kp, ws.ctrl, ws.hist, m, ws.cand, ws.cap);
    else
        select_compact_kernel<false><<<dim3((unsigned)bpf, (unsigned)n_frames), 256, 0, st>>>(coef, plane_len, kp, ws.ctrl, ws.hist, m, ws.cand, ws.cap);
    unsigned n_pow2 = 2;
    while (n_pow2 < k) n_pow2 <<= 1;
    const size_t smem = (size_t)(n_pow2 > FINISH_THREADS ? n_pow2 : FINISH_THREADS) * sizeof(uint64_t) + 8 * sizeof(uint64_t) +
                        (NBINS + 256) * sizeof(uint32_t);
    constexpr size_t LDS_LIMIT = 160 * 1024;
    {   // the dynamic-LDS ceiling is a per-device function attribute: set it once on every device used
        static std::atomic<bool> attr_set[64];           // contexts on several host threads: no plain bools
        int dev = 0;
        SSW_HIP_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
            SSW_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(select_finish_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT));
            if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
        }
    }
    select_finish_kernel<<<(unsigned)n_frames, FINISH_THREADS, smem, st>>>(coef, plane_len, kp, ws.ctrl, ws.hist,
                                                                            ws.cand, ws.cap, k, n_pow2, indices, ws.fallbacks);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// ---------------------------------------------------------------------------------------------
// Embed: Writer::embed_watermark (src/algorithm.rs:382-410) with insert functions :414-432.
// One thread per rank i; indices are unique so there are no write conflicts.
// ---------------------------------------------------------------------------------------------
__device__ inline float insert_fn(int method, float alpha, float original, float mark) {
    if (method == SSW_OPTION1) return original + alpha * mark;              // :414-416
    if (method == SSW_OPTION2) return original * (1.0f + alpha * mark);     // :420-424
    return original * expf(alpha * mark);                                   // :428-432
}
__device__ inline float extract_fn(int method, float alpha, float base, float derived) {
    if (method == SSW_OPTION1) return (derived - base) / alpha;             // :566-572
    if (method == SSW_OPTION2) return (derived - base) / (base * alpha);    // :576-583
    return logf(derived / base) / alpha;                                    // :587-593
}

__global__ __launch_bounds__(256) void embed_kernel(float* __restrict__ coef, size_t plane_len,
                                                    const uint32_t* __restrict__ indices, size_t idx_stride,
                                                    const float* __restrict__ marks,
                                                    const uint32_t* __restrict__ mark_offsets,
                                                    const uint32_t* __restrict__ mark_lens, size_t n_marks,
                                                    size_t max_len, size_t mark_stride, int method, float alpha) {
    const size_t f = blockIdx.y;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= max_len) return;
    const uint32_t j = indices[f * idx_stride + i];
    float* c = coef + f * plane_len + j;
    const float original = *c;
    if (n_marks == 1) {                                                     // :394-398
        const size_t off = mark_offsets ? mark_offsets[f] : f * mark_stride;
        const size_t len = mark_lens ? mark_lens[f] : max_len;
        if (i < len) *c = insert_fn(method, alpha, original, marks[off + i]);
    } else {                                                                // :399-408
        float cur = original;
        for (size_t m = 0; m < n_marks; ++m) {
            const size_t off = mark_offsets ? mark_offsets[f * n_marks + m] : (f * n_marks + m) * mark_stride;
            const size_t len = mark_lens ? mark_lens[f * n_marks + m] : max_len;
            if (i < len) {
                const float updated = insert_fn(method, alpha, original, marks[off + i]);
                const float change = updated - original;
                cur += change;
            }
        }
        *c = cur;
    }
}

int launch_embed(hipStream_t st, float* coef, size_t n_frames, size_t plane_len,
                 const uint32_t* indices, size_t idx_stride, const float* marks,
                 const uint32_t* mark_offsets, const uint32_t* mark_lens, size_t n_marks,
                 size_t max_len, size_t mark_stride, int method, float alpha) {
    if (n_frames == 0 || n_marks == 0 || max_len == 0) return SSW_OK;
    const dim3 grid((unsigned)((max_len + 255) / 256), (unsigned)n_frames);
    embed_kernel<<<grid, 256, 0, st>>>(coef, plane_len, indices, idx_stride, marks, mark_offsets,
                                       mark_lens, n_marks, max_len, mark_stride, method, alpha);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// Extract: Reader::extract_watermark (src/algorithm.rs:556-561)
__global__ __launch_bounds__(256) void extract_kernel(const float* __restrict__ base,
                                                      const float* __restrict__ derived, size_t plane_len,
                                                      const uint32_t* __restrict__ indices, size_t k,
                                                      int method, float alpha, float* __restrict__ out) {
    const size_t f = blockIdx.y;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= k) return;
    const uint32_t j = indices[f * k + i];
    out[f * k + i] = extract_fn(method, alpha, base[f * plane_len + j], derived[f * plane_len + j]);
}

int launch_extract(hipStream_t st, const float* base, const float* derived, size_t n_frames,
                   size_t plane_len, const uint32_t* indices, size_t k, int method, float alpha,
                   float* out) {
    if (n_frames == 0 || k == 0) return SSW_OK;
    const dim3 grid((unsigned)((k + 255) / 256), (unsigned)n_frames);
    extract_kernel<<<grid, 256, 0, st>>>(base, derived, plane_len, indices, k, method, alpha, out);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// The same with the derived coefficients taken from the compact plane of the pruned transform (prune.hip):
// derived value of index j = (u, v) sits at compact[f][u][pos[v]].
__global__ __launch_bounds__(256) void extract_pruned_kernel(const float* __restrict__ base,
                                                             const float* __restrict__ compact, size_t plane_len,
                                                             unsigned W, unsigned H, unsigned cap,
                                                             const uint32_t* __restrict__ pos,
                                                             const uint32_t* __restrict__ indices, size_t k,
                                                             int method, float alpha, float* __restrict__ out) {
    const size_t f = blockIdx.y;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= k) return;
    const uint32_t j = indices[f * k + i];
    const uint32_t u = j / W, v = j - u * W;
    const uint32_t p = pos[v];
    // p == none only in a chunk whose column set overflowed; the caller redoes that chunk
    const float d = p != 0xFFFFFFFFu ? compact[(f * H + u) * (size_t)cap + p] : 0.0f;
    out[f * k + i] = extract_fn(method, alpha, base[f * plane_len + j], d);
}

int launch_extract_pruned(hipStream_t st, const float* base, const float* compact, size_t n_frames, size_t w, size_t h,
                          size_t cap, const uint32_t* pos, const uint32_t* indices, size_t k, int method, float alpha,
                          float* out) {
    if (n_frames == 0 || k == 0) return SSW_OK;
    const dim3 grid((unsigned)((k + 255) / 256), (unsigned)n_frames);
    extract_pruned_kernel<<<grid, 256, 0, st>>>(base, compact, w * h, (unsigned)w, (unsigned)h, (unsigned)cap, pos, indices, k,
                                                method, alpha, out);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// Similarity: Tester::similarity (src/algorithm.rs:702-713).  The reference accumulates both
// sums sequentially in f32; rounding depends on that order, so it is kept: the products are
// formed in parallel (each is a single rounding either way), one lane does the two running sums.
__global__ __launch_bounds__(256) void similarity_kernel(const float* __restrict__ extracted,
                                                         const float* __restrict__ marks, size_t k,
                                                         float* __restrict__ sims) {
    __shared__ float pn[1024], pd[1024];
    const size_t f = blockIdx.x;
    const float* e = extracted + f * k;
    const float* m = marks + f * k;
    float nominator = 0.0f, denominator = 0.0f;
    for (size_t base = 0; base < k; base += 1024) {
        const size_t n = (k - base) < 1024 ? (k - base) : 1024;
        for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
            const float ev = e[base + i];
            pn[i] = ev * m[base + i];
            pd[i] = ev * ev;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (size_t i = 0; i < n; ++i) {
                nominator += pn[i];
                denominator += pd[i];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) sims[f] = nominator / sqrtf(denominator);
}

int launch_similarity(hipStream_t st, const float* extracted, const float* marks, size_t n_pairs,
                      size_t k, float* sims) {
    if (n_pairs == 0) return SSW_OK;
    similarity_kernel<<<(unsigned)n_pairs, 256, 0, st>>>(extracted, marks, k, sims);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

// Many-marks similarity: denominators sum(e*e) in the reference's sequential f32 order (one lane per
// extracted mark), then sims[b][j] = nom[b][j] / sqrt(den[b]) on the GEMM result.
__global__ __launch_bounds__(256) void sim_den_kernel(const float* __restrict__ extracted, size_t k,
                                                      float* __restrict__ den) {
    __shared__ float pd[1024];
    const size_t f = blockIdx.x;
    const float* e = extracted + f * k;
    float denominator = 0.0f;
    for (size_t base = 0; base < k; base += 1024) {
        const size_t n = (k - base) < 1024 ? (k - base) : 1024;
        for (size_t i = threadIdx.x; i < n; i += blockDim.x) { const float ev = e[base + i]; pd[i] = ev * ev; }
        __syncthreads();
        if (threadIdx.x == 0)
            for (size_t i = 0; i < n; ++i) denominator += pd[i];
        __syncthreads();
    }
    if (threadIdx.x == 0) den[f] = denominator;
}
__global__ void sim_scale_kernel(float* __restrict__ sims, const float* __restrict__ den, size_t n_ext, size_t n_marks) {
    const size_t total = n_ext * n_marks;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        sims[i] = sims[i] / sqrtf(den[i / n_marks]);
}
int launch_sim_den(hipStream_t st, const float* extracted, size_t n_ext, size_t k, float* den) {
    if (!n_ext) return SSW_OK;
    sim_den_kernel<<<(unsigned)n_ext, 256, 0, st>>>(extracted, k, den);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}
int launch_sim_scale(hipStream_t st, float* sims, const float* den, size_t n_ext, size_t n_marks) {
    const size_t total = n_ext * n_marks;
    if (!total) return SSW_OK;
    sim_scale_kernel<<<(unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096), 256, 0, st>>>(sims, den, n_ext, n_marks);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

__global__ void widen_kernel(const uint32_t* in, size_t n, uint64_t* out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = in[i];
}
int launch_widen_indices(hipStream_t st, const uint32_t* in, size_t n, uint64_t* out) {
    if (n == 0) return SSW_OK;
    widen_kernel<<<(unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), 256, 0, st>>>(in, n, out);
    SSW_HIP_CHECK(hipGetLastError());
    return SSW_OK;
}

}  // namespace ssw
