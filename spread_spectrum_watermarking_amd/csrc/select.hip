// Coefficient ordering (top-k), embed, extract and similarity kernels.
//
// Ordering: obtain_indices_by_function (/root/reference/src/algorithm.rs:200-210) does a full
// stable descending sort of all W*H-1 non-DC coefficients with a boxed comparator, but only
// the first k (= mark length) entries are ever consumed (:396, :556-557).  Here the first k
// entries are produced directly:
//   1. every coefficient gets a 64-bit composite key  (sortable(key_f32) << 32) | ~index
//      -- key_f32 as in the comparators (:214-280), sortable() = f32::total_cmp order as an
//      unsigned integer, ~index so that equal keys rank lower index first (stable sort of an
//      index-ascending list).  Composite keys are unique, so there are no ties left.
//   2. MSD radix select (11-bit digits) of the k-th largest composite key per frame: one
//      histogram pass over the plane per digit, decided on the device (no host round trip);
//      passes after the one that isolates the k-th key exit immediately.
//   3. compaction of the exactly-k survivors, bitonic sort of k composites in LDS, emit ~low32.
// HBM-bound: 4 B/px per pass over the coefficient plane.
#include "ssw_internal.hpp"

namespace ssw {

constexpr int DIGIT_BITS = 11;
constexpr int NBINS = 1 << DIGIT_BITS;          // 2048
constexpr int N_PASSES = 6;                     // 11*5 + 9 = 64 bits
constexpr size_t MAX_K = 16384;                 // 128 KiB of LDS for the in-block sort

size_t select_max_k() { return MAX_K; }

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

__device__ inline uint64_t composite_key(const KeyParams& kp, uint32_t index, float value) {
    float key;
    if (kp.ordering == SSW_ORDER_ENERGY) {
        key = value * value;                                         // :214-221
    } else {
        const float scaled = kp.s[index < kp.w][(index % kp.w) == 0] * value;   // :252-266
        key = (kp.ordering == SSW_ORDER_ENERGY_ORTHOGONAL) ? scaled * scaled : scaled;   // :178-187
    }
    return ((uint64_t)sortable(key) << 32) | (uint32_t)(~index);
}

// state layout per frame: [0] prefix, [1] bits decided, [2] need, [3] resolved
__device__ inline int pass_width(int bits_done) { return (64 - bits_done) < DIGIT_BITS ? (64 - bits_done) : DIGIT_BITS; }

__global__ void select_init_kernel(uint64_t* state, uint32_t* hist, uint32_t* cand_count, size_t k) {
    const size_t f = blockIdx.x;
    for (int i = threadIdx.x; i < NBINS; i += blockDim.x) hist[f * NBINS + i] = 0;
    if (threadIdx.x == 0) {
        state[f * 4 + 0] = 0;
        state[f * 4 + 1] = 0;
        state[f * 4 + 2] = k;
        state[f * 4 + 3] = 0;
        cand_count[f] = 0;
    }
}

__global__ __launch_bounds__(256) void select_hist_kernel(const float* __restrict__ coef, size_t plane_len,
                                                          KeyParams kp, const uint64_t* __restrict__ state,
                                                          uint32_t* __restrict__ hist) {
    const size_t f = blockIdx.y;
    if (state[f * 4 + 3]) return;                                    // already isolated
    __shared__ uint32_t lh[NBINS];
    for (int i = threadIdx.x; i < NBINS; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const uint64_t prefix = state[f * 4 + 0];
    const int bits_done = (int)state[f * 4 + 1];
    const int width = pass_width(bits_done);
    const int shift = 64 - bits_done - width;
    const uint32_t mask = (1u << width) - 1u;
    const float* __restrict__ c = coef + f * plane_len;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < plane_len; j += stride) {
        if (j == 0) continue;                                        // DC is skipped (:204)
        const uint64_t comp = composite_key(kp, (uint32_t)j, c[j]);
        if (bits_done == 0 || (comp >> (64 - bits_done)) == prefix)
            atomicAdd(&lh[(uint32_t)(comp >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NBINS; i += blockDim.x) {
        const uint32_t v = lh[i];
        if (v) atomicAdd(&hist[f * NBINS + i], v);
    }
}

// one block per frame: locate the digit holding the `need`-th largest key, update the state
__global__ __launch_bounds__(256) void select_find_kernel(uint64_t* __restrict__ state, uint32_t* __restrict__ hist) {
    const size_t f = blockIdx.x;
    if (state[f * 4 + 3]) return;
    __shared__ uint32_t part[256];
    __shared__ uint32_t chosen[2];
    uint32_t* h = hist + f * NBINS;
    const int bits_done = (int)state[f * 4 + 1];
    const int width = pass_width(bits_done);
    const uint64_t need = state[f * 4 + 2];
    // thread t owns the 8 digits [8t, 8t+8) counted from the TOP: digit = NBINS-1 - (8t+e)
    const int t = threadIdx.x;
    uint32_t local[8];
    uint32_t sum = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { local[e] = h[NBINS - 1 - (8 * t + e)]; sum += local[e]; }
    part[t] = sum;
    __syncthreads();
    if (t == 0) {                                                    // serial scan of 256 partials
        uint64_t run = 0;
        int owner = 255;
        for (int i = 0; i < 256; ++i) {
            if (run + part[i] >= need) { owner = i; break; }
            run += part[i];
        }
        chosen[0] = (uint32_t)owner;
        chosen[1] = (uint32_t)run;                                   // keys above the owner's digits
    }
    __syncthreads();
    if (t == (int)chosen[0]) {
        uint64_t run = chosen[1];
        int e = 0;
        for (; e < 7; ++e) {
            if (run + local[e] >= need) break;
            run += local[e];
        }
        const uint32_t digit = (uint32_t)(NBINS - 1 - (8 * t + e));
        const uint64_t new_need = need - run;
        const int new_bits = bits_done + width;
        state[f * 4 + 0] = (state[f * 4 + 0] << width) | (uint64_t)(digit & ((1u << width) - 1u));
        state[f * 4 + 1] = (uint64_t)new_bits;
        state[f * 4 + 2] = new_need;
        state[f * 4 + 3] = (local[e] == new_need || new_bits >= 64) ? 1 : 0;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) h[8 * t + e] = 0;                    // ready for the next pass
}

__global__ __launch_bounds__(256) void select_compact_kernel(const float* __restrict__ coef, size_t plane_len,
                                                             KeyParams kp, const uint64_t* __restrict__ state,
                                                             uint64_t* __restrict__ cand, size_t cap,
                                                             uint32_t* __restrict__ cand_count) {
    const size_t f = blockIdx.y;
    const uint64_t prefix = state[f * 4 + 0];
    const int bits_done = (int)state[f * 4 + 1];
    const float* __restrict__ c = coef + f * plane_len;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < plane_len; j += stride) {
        if (j == 0) continue;
        const uint64_t comp = composite_key(kp, (uint32_t)j, c[j]);
        const uint64_t top = bits_done >= 64 ? comp : (comp >> (64 - bits_done));
        if (top >= prefix) {
            const uint32_t pos = atomicAdd(&cand_count[f], 1u);
            if (pos < cap) cand[f * cap + pos] = comp;
        }
    }
}

// bitonic sort (descending) of the k survivors of one frame in LDS, then emit the indices
__global__ __launch_bounds__(1024) void select_sort_kernel(const uint64_t* __restrict__ cand, size_t cap,
                                                           size_t k, unsigned n_pow2,
                                                           uint32_t* __restrict__ indices) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t* s = reinterpret_cast<uint64_t*>(smem_raw);
    const size_t f = blockIdx.x;
    for (unsigned i = threadIdx.x; i < n_pow2; i += blockDim.x) s[i] = (i < k) ? cand[f * cap + i] : 0ull;
    __syncthreads();
    for (unsigned size = 2; size <= n_pow2; size <<= 1) {
        for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
            for (unsigned i = threadIdx.x; i < n_pow2 / 2; i += blockDim.x) {
                const unsigned lo = 2 * i - (i & (stride - 1));
                const unsigned hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const uint64_t a = s[lo], b = s[hi];
                if ((a < b) == desc) { s[lo] = b; s[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (unsigned i = threadIdx.x; i < k; i += blockDim.x) indices[f * k + i] = ~(uint32_t)(s[i] & 0xFFFFFFFFull);
}

int launch_topk(hipStream_t st, const float* coef, size_t n_frames, size_t w, size_t h, int ordering,
                size_t k, const SelectWorkspace& ws, uint32_t* indices) {
    const size_t plane_len = w * h;
    if (n_frames == 0 || k == 0) return SSW_OK;
    if (plane_len > 0xFFFFFFFFull) return SSW_ERR_BAD_DIMS;
    if (k > plane_len - 1) return SSW_ERR_K_TOO_LARGE;
    if (k > MAX_K) return SSW_ERR_UNSUPPORTED;
    if (ws.frames < n_frames || ws.cap < k) return SSW_ERR_BAD_ARG;
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
    select_init_kernel<<<(unsigned)n_frames, 256, 0, st>>>(ws.state, ws.hist, ws.cand_count, k);
    size_t bpf = (plane_len + 256 * 16 - 1) / (256 * 16);            // >= 16 elements per thread
    if (bpf < 1) bpf = 1;
    if (bpf > 1024) bpf = 1024;
    const dim3 grid((unsigned)bpf, (unsigned)n_frames);
    for (int p = 0; p < N_PASSES; ++p) {
        select_hist_kernel<<<grid, 256, 0, st>>>(coef, plane_len, kp, ws.state, ws.hist);
        select_find_kernel<<<(unsigned)n_frames, 256, 0, st>>>(ws.state, ws.hist);
    }
    select_compact_kernel<<<grid, 256, 0, st>>>(coef, plane_len, kp, ws.state, ws.cand, ws.cap, ws.cand_count);
    unsigned n_pow2 = 2;
    while (n_pow2 < k) n_pow2 <<= 1;
    const size_t smem = (size_t)n_pow2 * sizeof(uint64_t);
    static bool attr_set = false;
    if (!attr_set) {
        SSW_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(select_sort_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(MAX_K * sizeof(uint64_t))));
        attr_set = true;
    }
    select_sort_kernel<<<(unsigned)n_frames, 1024, smem, st>>>(ws.cand, ws.cap, k, n_pow2, indices);
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
                                                    size_t max_len, int method, float alpha) {
    const size_t f = blockIdx.y;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= max_len) return;
    const uint32_t j = indices[f * idx_stride + i];
    float* c = coef + f * plane_len + j;
    const float original = *c;
    if (n_marks == 1) {                                                     // :394-398
        const size_t off = mark_offsets ? mark_offsets[f] : f * max_len;
        const size_t len = mark_lens ? mark_lens[f] : max_len;
        if (i < len) *c = insert_fn(method, alpha, original, marks[off + i]);
    } else {                                                                // :399-408
        float cur = original;
        for (size_t m = 0; m < n_marks; ++m) {
            const size_t off = mark_offsets ? mark_offsets[f * n_marks + m] : (f * n_marks + m) * max_len;
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
                 size_t max_len, int method, float alpha) {
    if (n_frames == 0 || n_marks == 0 || max_len == 0) return SSW_OK;
    const dim3 grid((unsigned)((max_len + 255) / 256), (unsigned)n_frames);
    embed_kernel<<<grid, 256, 0, st>>>(coef, plane_len, indices, idx_stride, marks, mark_offsets,
                                       mark_lens, n_marks, max_len, method, alpha);
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
