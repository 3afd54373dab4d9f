// Full ordering of one coefficient plane: all W*H-1 indices in the reference's order
// (obtain_indices_by_function, /root/reference/src/algorithm.rs:200-210) for callers that ask
// Reader::indices() (:506-508) or embed marks longer than the in-LDS top-k limit.
//
// Not on the hot path: the reference itself never consumes more than the mark length (:396,
// :556-557), and marks are 1 000 - 10 000 long, which select.hip serves directly.  The keys are
// built here exactly like the top-k path (same comparators, f32::total_cmp order as u32) and handed
// to rocPRIM's device radix sort (descending, stable: equal keys keep ascending index order, which is
// the reference's stable sort of an index-ascending list).
#include <rocprim/device/device_radix_sort.hpp>

#include "ssw_internal.hpp"

namespace ssw {

struct FullKeyParams {
    int ordering;
    unsigned w;
    float s[2][2];
};

__device__ inline uint32_t sortable_u32(float v) {
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ void full_keys_kernel(const float* __restrict__ c, size_t plane_len, FullKeyParams kp,
                                 uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i + 1 < plane_len;
         i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t index = (uint32_t)(i + 1);                      // DC skipped (:204)
        const float value = c[index];
        float key;
        if (kp.ordering == SSW_ORDER_ENERGY) {
            key = value * value;
        } else {
            const float scaled = kp.s[index < kp.w][(index % kp.w) == 0] * value;
            key = (kp.ordering == SSW_ORDER_ENERGY_ORTHOGONAL) ? scaled * scaled : scaled;
        }
        keys[i] = sortable_u32(key);
        vals[i] = index;
    }
}

// scratch: 4 * (plane_len - 1) u32 (keys in/out, values in/out) + rocPRIM temporary storage.
int full_sort_scratch_bytes(size_t plane_len, size_t* bytes) {
    const size_t n = plane_len - 1;
    size_t temp = 0;
    hipError_t e = rocprim::radix_sort_pairs_desc(nullptr, temp, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                                  (uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, 32, nullptr);
    if (e != hipSuccess) { set_last_error(std::string("rocprim size query: ") + hipGetErrorString(e)); return SSW_ERR_HIP; }
    *bytes = 4 * n * sizeof(uint32_t) + ((temp + 255) / 256) * 256 + 256;
    return SSW_OK;
}

int launch_full_sort(hipStream_t st, const float* coef, size_t w, size_t h, int ordering, void* scratch,
                     size_t scratch_bytes, uint32_t* indices_out, size_t k) {
    const size_t plane_len = w * h;
    if (plane_len < 2 || k == 0) return SSW_OK;
    const size_t n = plane_len - 1;
    if (k > n) return SSW_ERR_K_TOO_LARGE;
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
    uint32_t* keys_in = static_cast<uint32_t*>(scratch);
    uint32_t* keys_out = keys_in + n;
    uint32_t* vals_in = keys_out + n;
    uint32_t* vals_out = vals_in + n;
    char* temp = reinterpret_cast<char*>(vals_out + n);
    temp = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(temp) + 255) / 256 * 256);
    size_t temp_bytes = scratch_bytes - (size_t)(temp - static_cast<char*>(scratch));
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    full_keys_kernel<<<blocks, 256, 0, st>>>(coef, plane_len, kp, keys_in, vals_in);
    SSW_HIP_CHECK(hipGetLastError());
    SSW_HIP_CHECK(rocprim::radix_sort_pairs_desc(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 32, st));
    SSW_HIP_CHECK(hipMemcpyAsync(indices_out, vals_out, k * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    return SSW_OK;
}

}  // namespace ssw
