// Calibrates what the f64 / f32 MFMA pipe sustains on this chip with no memory traffic, so that
// roofline.frac of the DCT GEMMs can be read against a measured ceiling as well as the spec peak.
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_peak tools/mfma_peak.hip ; run: /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0: MFMA only.  MODE 1: + one cvt and one f64 add per two MFMAs (fold arithmetic density).
// MODE 2: MODE 1 + one ds_read_b128 per four MFMAs.
template <int MODE>
__global__ __launch_bounds__(256, 2) void f64_loop(double* out, const float* in, int iters) {
    __shared__ float lds[4096];
    f64x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f64x4{0, 0, 0, 0};
    const int lane = threadIdx.x;
    lds[lane] = in[lane]; lds[lane + 256] = in[lane + 256];
    __syncthreads();
    double a = in[lane & 63], b = in[(lane + 7) & 63];
    float fa = in[lane & 31], fb = in[(lane + 3) & 31];
    const float4* l4 = reinterpret_cast<const float4*>(lds);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE >= 2 && (j & 3) == 0) {
                float4 v = l4[(lane + j + it) & 127];
                fa = v.x; fb = v.w;
            }
            if (MODE >= 1 && (j & 1) == 0) {
                double da = (double)fa, db = (double)fb;
                a = da + db; b = da - db;
                if (MODE == 1) { fa += 1.0f; }
            }
            acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
        }
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + lane] = s;
}

// One extra VALU instruction of a given kind per MFMA (asm volatile keeps the count exact):
// which instruction classes contend with the f64 MFMA pipe?
// KIND 0 v_add_f64, 1 v_cvt_f64_f32, 2 v_add_f32, 3 v_fma_f64, 4 v_and_b32, 5 two v_add_f32, 6 v_mov_b32
template <int KIND>
__global__ __launch_bounds__(256, 2) void f64_mix(double* out, const float* in, int iters) {
    f64x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f64x4{0, 0, 0, 0};
    const int lane = threadIdx.x;
    double a = in[lane & 63], b = in[(lane + 7) & 63];
    double d0 = a, d1 = b; float f0 = in[lane & 31], f1 = in[(lane + 5) & 31]; int i0 = lane, i1 = lane * 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (KIND == 0) asm volatile("v_add_f64 %0, %1, %2" : "=v"(d0) : "v"(d1), "v"(b));
            if (KIND == 1) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d0) : "v"(f1));
            if (KIND == 2) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f0) : "v"(f1), "v"(f1));
            if (KIND == 3) asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(d0) : "v"(d1), "v"(b));
            if (KIND == 4) asm volatile("v_and_b32 %0, %1, %2" : "=v"(i0) : "v"(i1), "v"(i1));
            if (KIND == 5) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(f0) : "v"(f1), "v"(f1));
                             asm volatile("v_add_f32 %0, %1, %2" : "=v"(f0) : "v"(f1), "v"(f1)); }
            if (KIND == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(i0) : "v"(i1));
            acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
        }
    }
    double s = d0 + f0 + i0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + lane] = s;
}

__global__ __launch_bounds__(256, 2) void f32_loop(float* out, const float* in, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    const int lane = threadIdx.x;
    float a = in[lane & 63], b = in[(lane + 7) & 63];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j & 3], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + lane] = s;
}

// f32 MFMA with one extra VALU instruction per MFMA: 0 none, 1 v_add_f32, 2 v_mov_b32, 3 two v_add_f32
template <int KIND>
__global__ __launch_bounds__(256, 2) void f32_mix(float* out, const float* in, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    const int lane = threadIdx.x;
    float a = in[lane & 63], b = in[(lane + 7) & 63];
    float f0 = in[lane & 31], f1 = in[(lane + 5) & 31]; int i0 = lane, i1 = lane * 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (KIND == 1) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f0) : "v"(f1), "v"(f1));
            if (KIND == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(i0) : "v"(i1));
            if (KIND == 3) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(f0) : "v"(f1), "v"(f1));
                             asm volatile("v_add_f32 %0, %1, %2" : "=v"(f0) : "v"(f1), "v"(f1)); }
            acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j & 3], 0, 0, 0);
        }
    }
    float s = f0 + i0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + lane] = s;
}

int main() {
    float* in; double* out;
    CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 4096 * 256 * 8));
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) >> 8) / 16777216.0f - 0.5f;
    CK(hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
        const int grid = 256 * blocks_per_cu;
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) f64_loop<0><<<grid, 256>>>(out, in, iters);
                if (mode == 1) f64_loop<1><<<grid, 256>>>(out, in, iters);
                if (mode == 2) f64_loop<2><<<grid, 256>>>(out, in, iters);
                if (mode == 3) f32_loop<<<grid, 256>>>((float*)out, in, iters);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            const double flop_per_mfma = (mode == 3) ? 2.0 * 32 * 32 * 2 : 2.0 * 16 * 16 * 4;
            const double flop = (double)grid * 4 * iters * 16 * flop_per_mfma;
            printf("%s blocks/CU=%d  %.3f ms  %.1f TFLOP/s\n",
                   mode == 0 ? "f64 mfma only      " : mode == 1 ? "f64 mfma+cvt/add   " : mode == 2 ? "f64 mfma+cvt/add+ds" : "f32 32x32x2 only   ",
                   blocks_per_cu, best, flop / best / 1e9);
        }
    }
    const char* kinds[7] = {"v_add_f64", "v_cvt_f64_f32", "v_add_f32", "v_fma_f64", "v_and_b32", "2x v_add_f32", "v_mov_b32"};
    for (int kind = 0; kind < 7; ++kind) {
        const int grid = 512; float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            switch (kind) {
                case 0: f64_mix<0><<<grid, 256>>>(out, in, iters); break;
                case 1: f64_mix<1><<<grid, 256>>>(out, in, iters); break;
                case 2: f64_mix<2><<<grid, 256>>>(out, in, iters); break;
                case 3: f64_mix<3><<<grid, 256>>>(out, in, iters); break;
                case 4: f64_mix<4><<<grid, 256>>>(out, in, iters); break;
                case 5: f64_mix<5><<<grid, 256>>>(out, in, iters); break;
                default: f64_mix<6><<<grid, 256>>>(out, in, iters); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double flop = (double)grid * 4 * iters * 16 * 2048.0;
        printf("f64 mfma + 1x %-14s per mfma, 2 blocks/CU: %.3f ms  %.1f TFLOP/s\n", kinds[kind], best, flop / best / 1e9);
    }
    const char* fk[4] = {"nothing", "v_add_f32", "v_mov_b32", "2x v_add_f32"};
    for (int kind = 0; kind < 4; ++kind) {
        const int grid = 512; float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            switch (kind) {
                case 0: f32_mix<0><<<grid, 256>>>((float*)out, in, iters); break;
                case 1: f32_mix<1><<<grid, 256>>>((float*)out, in, iters); break;
                case 2: f32_mix<2><<<grid, 256>>>((float*)out, in, iters); break;
                default: f32_mix<3><<<grid, 256>>>((float*)out, in, iters); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double flop = (double)grid * 4 * iters * 16 * 4096.0;
        printf("f32 32x32x2 mfma + 1x %-12s per mfma, 2 blocks/CU: %.3f ms  %.1f TFLOP/s\n", fk[kind], best, flop / best / 1e9);
    }
    return 0;
}
