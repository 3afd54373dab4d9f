// Full ordering of coefficient planes: all W*H-1 indices in the reference's order
// (obtain_indices_by_function, /root/reference/src/algorithm.rs:200-210) for callers that ask
// Reader::indices() (:506-508) or embed marks longer than the in-LDS top-k limit.
//
// Not on the hot path: the reference itself never consumes more than the mark length (:396,
// :556-557), and marks are 1 000 - 10 000 long, which select.hip serves directly.
//
// A batched least-significant-digit radix sort, written here (r2 called rocPRIM once per frame): the keys are
// built exactly like the top-k path (same comparators, f32::total_cmp order as u32), the values are the indices
// in ascending order, and four stable 8-bit passes sort by descending key -- equal keys keep ascending index
// order, which is the reference's stable sort of an index-ascending list (:205).  All frames of a group go
// through every kernel together (blockIdx.y = frame).  Per pass: digit histogram per 4096-item tile, one
// exclusive scan over (digit, tile) per frame, then a stable scatter in which a tile's items are ranked row by
// row (256 items): lanes with the same digit find each other with eight ballots, waves are ordered through LDS.
#include "ssw_internal.hpp"

namespace ssw {

struct FullKeyParams {
    int ordering;
    unsigned w;
    float s[2][2];
};

namespace {
constexpr unsigned RADIX = 256, TILE_ROWS = 16, TILE = 256 * TILE_ROWS;      // items per block

__device__ inline uint32_t sortable_u32(float v) {
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
// descending order through an ascending scatter: digit 255 first
__device__ inline unsigned digit_of(uint32_t key, unsigned shift) { return 255u - ((key >> shift) & 255u); }

__global__ void full_keys_kernel(const float* __restrict__ coef, size_t plane_len, FullKeyParams kp,
                                 uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, size_t frame_stride) {
    const float* c = coef + blockIdx.y * plane_len;
    uint32_t* ko = keys + blockIdx.y * frame_stride;
    uint32_t* vo = vals + blockIdx.y * frame_stride;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i + 1 < plane_len; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t index = (uint32_t)(i + 1);                      // DC skipped (:204)
        const float value = c[index];
        float key;
        if (kp.ordering == SSW_ORDER_ENERGY) {
            key = value * value;                                         // :214-221
        } else {
            const float scaled = kp.s[index < kp.w][(index % kp.w) == 0] * value;   // :252-266
            key = (kp.ordering == SSW_ORDER_ENERGY_ORTHOGONAL) ? scaled * scaled : scaled;
        }
        ko[i] = sortable_u32(key);
        vo[i] = index;
    }
}

// counts[frame][digit][tile]
__global__ __launch_bounds__(256) void sort_hist_kernel(const uint32_t* __restrict__ keys, size_t n, size_t frame_stride,
                                                        unsigned shift, unsigned tiles, uint32_t* __restrict__ counts) {
    __shared__ uint32_t h[RADIX];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t* k = keys + blockIdx.y * frame_stride;
    const size_t base = (size_t)blockIdx.x * TILE;
#pragma unroll 4
    for (unsigned r = 0; r < TILE_ROWS; ++r) {
        const size_t i = base + r * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[digit_of(k[i], shift)], 1u);
    }
    __syncthreads();
    counts[((size_t)blockIdx.y * RADIX + threadIdx.x) * tiles + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of one frame's counts in (digit, tile) order; one block per frame
__global__ __launch_bounds__(1024) void sort_scan_kernel(uint32_t* __restrict__ counts, unsigned tiles) {
    __shared__ uint32_t part[1024];
    uint32_t* c = counts + (size_t)blockIdx.x * RADIX * tiles;
    const size_t total = (size_t)RADIX * tiles;
    const size_t per = (total + 1023) / 1024;
    const size_t lo = threadIdx.x * per, hi = lo + per < total ? lo + per : total;
    uint32_t sum = 0;
    for (size_t i = lo; i < hi; ++i) sum += c[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x < 64) {                                   // one wave scans the 1024 partial sums, 16 per lane
        uint32_t loc[16], s = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) { loc[j] = part[16 * threadIdx.x + j]; s += loc[j]; }
        uint32_t incl = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (threadIdx.x >= (unsigned)d) incl += up;
        }
        uint32_t run = incl - s;
#pragma unroll
        for (int j = 0; j < 16; ++j) { part[16 * threadIdx.x + j] = run; run += loc[j]; }
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (size_t i = lo; i < hi; ++i) { const uint32_t v = c[i]; c[i] = run; run += v; }
}

__global__ __launch_bounds__(256) void sort_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                           uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                           size_t n, size_t frame_stride, unsigned shift, unsigned tiles,
                                                           const uint32_t* __restrict__ offsets) {
    __shared__ uint32_t run[RADIX];            // where the tile's next item of each digit goes
    __shared__ uint32_t wcount[4][RADIX];      // this row: items per (wave, digit)
    const size_t fo = blockIdx.y * frame_stride;
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    run[tid] = offsets[((size_t)blockIdx.y * RADIX + tid) * tiles + blockIdx.x];
    const size_t base = (size_t)blockIdx.x * TILE;
    for (unsigned r = 0; r < TILE_ROWS; ++r) {
        if (base + (size_t)r * 256 >= n) break;               // block-uniform
#pragma unroll
        for (int w = 0; w < 4; ++w) wcount[w][tid] = 0;
        __syncthreads();
        const size_t i = base + r * 256 + tid;
        const bool ok = i < n;
        const uint32_t key = ok ? keys_in[fo + i] : 0u, val = ok ? vals_in[fo + i] : 0u;
        const unsigned d = ok ? digit_of(key, shift) : 0u;
        // lanes of this wave that hold the same digit
        unsigned long long same = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        const unsigned before = __popcll(same & ((1ull << lane) - 1ull));
        if (ok && before == 0) wcount[wave][d] = (uint32_t)__popcll(same);
        __syncthreads();
        if (ok) {
            uint32_t pos = run[d] + before;
            for (unsigned w = 0; w < wave; ++w) pos += wcount[w][d];
            keys_out[fo + pos] = key;
            vals_out[fo + pos] = val;
        }
        __syncthreads();
        run[tid] += wcount[0][tid] + wcount[1][tid] + wcount[2][tid] + wcount[3][tid];
        __syncthreads();
    }
}
}  // namespace

// frames sorted together: keys and values in and out (16 bytes per coefficient) plus the tile counts, kept under 2 GB
static size_t full_sort_group(size_t plane_len, size_t n_frames) {
    const size_t per_frame = 16 * (plane_len - 1) + 4 * RADIX * ((plane_len - 1 + TILE - 1) / TILE) + 1024;
    size_t g = ((size_t)2 << 30) / per_frame;
    if (g < 1) g = 1;
    return g < n_frames ? g : n_frames;
}

int full_sort_scratch_bytes(size_t plane_len, size_t n_frames, size_t* bytes) {
    if (plane_len < 2) { *bytes = 16; return SSW_OK; }
    const size_t n = plane_len - 1, tiles = (n + TILE - 1) / TILE, g = full_sort_group(plane_len, n_frames);
    *bytes = g * (4 * n * sizeof(uint32_t) + RADIX * tiles * sizeof(uint32_t)) + 1024;
    return SSW_OK;
}

// first k entries of the full order of n_frames planes -> indices_out[frame][k]
int launch_full_sort(hipStream_t st, const float* coef, size_t n_frames, size_t w, size_t h, int ordering, void* scratch,
                     size_t scratch_bytes, uint32_t* indices_out, size_t k) {
    const size_t plane_len = w * h;
    if (plane_len < 2 || k == 0 || n_frames == 0) return SSW_OK;
    const size_t n = plane_len - 1;
    if (k > n) return SSW_ERR_K_TOO_LARGE;
    if (n > 0xFFFFFFFFull) return SSW_ERR_BAD_DIMS;
    FullKeyParams kp;
    kp.ordering = ordering;
    kp.w = (unsigned)w;
    {   // same f32 evaluation as select.hip (src/algorithm.rs:245-265)
        const float s_k0_w = sqrtf(1.0f / (4.0f * (float)w)), s_k0_h = sqrtf(1.0f / (4.0f * (float)h));
        const float s_w = sqrtf(1.0f / (2.0f * (float)w)), s_h = sqrtf(1.0f / (2.0f * (float)h));
        for (int fr = 0; fr < 2; ++fr)
            for (int fc = 0; fc < 2; ++fc) {
                volatile float sc = 1.0f;
                sc = sc * (fr ? s_k0_w : s_w);
                sc = sc * (fc ? s_k0_h : s_h);
                kp.s[fr][fc] = sc;
            }
    }
    const size_t tiles = (n + TILE - 1) / TILE, group = full_sort_group(plane_len, n_frames);
    if (tiles > 0x7FFFFFFFull) return SSW_ERR_BAD_DIMS;
    if (scratch_bytes < group * (4 * n + RADIX * tiles) * sizeof(uint32_t)) return SSW_ERR_BAD_ARG;
    uint32_t* keys_a = static_cast<uint32_t*>(scratch);
    uint32_t* vals_a = keys_a + group * n;
    uint32_t* keys_b = vals_a + group * n;
    uint32_t* vals_b = keys_b + group * n;
    uint32_t* counts = vals_b + group * n;
    for (size_t f0 = 0; f0 < n_frames; f0 += group) {
        const unsigned g = (unsigned)(group < n_frames - f0 ? group : n_frames - f0);
        const unsigned kb = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        full_keys_kernel<<<dim3(kb, g), 256, 0, st>>>(coef + f0 * plane_len, plane_len, kp, keys_a, vals_a, n);
        uint32_t *ki = keys_a, *vi = vals_a, *ko = keys_b, *vo = vals_b;
        for (unsigned shift = 0; shift < 32; shift += 8) {
            sort_hist_kernel<<<dim3((unsigned)tiles, g), 256, 0, st>>>(ki, n, n, shift, (unsigned)tiles, counts);
            sort_scan_kernel<<<g, 1024, 0, st>>>(counts, (unsigned)tiles);
            sort_scatter_kernel<<<dim3((unsigned)tiles, g), 256, 0, st>>>(ki, vi, ko, vo, n, n, shift, (unsigned)tiles, counts);
            uint32_t* t = ki; ki = ko; ko = t;
            t = vi; vi = vo; vo = t;
        }
        SSW_HIP_CHECK(hipGetLastError());
        // four passes: the sorted values are back in the first buffer pair
        SSW_HIP_CHECK(hipMemcpy2DAsync(indices_out + f0 * k, k * sizeof(uint32_t), vi, n * sizeof(uint32_t), k * sizeof(uint32_t), g,
                                       hipMemcpyDeviceToDevice, st));
    }
    return SSW_OK;
}

}  // namespace ssw
