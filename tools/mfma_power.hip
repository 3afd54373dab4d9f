// What the f64 MFMA pipe SUSTAINS on this chip (seconds of back-to-back launches, so that the power management has settled),
// by what the operands are: the r1 calibration (tools/mfma_peak.hip) multiplies two constant, 24-bit-mantissa fragments and
// runs at the full clock; the DCT GEMMs multiply folded pixel sums (25..28-bit mantissas) by cosines (53 bits), new fragments
// from LDS for every MFMA.  Reports TFLOP/s and the shader clock (s_memtime / s_memrealtime) per variant.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/mfma_power tools/mfma_power.hip ; run: tools/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0: constant fragments in registers.  1: a pool of 4 x 2 + 2 x 2 register fragments of the given data, every MFMA of a
// half-step a different pair (the GEMM's 16 MFMAs per half-step).  2: the fragments re-read from LDS every half-step (12 doubles
// per lane and half-step, like pair_gemm_f64_kernel: ds_read_b64, the tile's data in LDS).
template <int MODE>
__global__ __launch_bounds__(256, 2) void loop(double* out, const double* xa, const double* yb, int iters, unsigned long long* clk) {
    __shared__ double lds[2 * 3072];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 3072; i += 256) lds[i] = i < 2 * 2048 ? xa[(i * 7 + blockIdx.x) & 8191] : yb[(i * 5 + blockIdx.x) & 8191];
    __syncthreads();
    f64x4 acc1[4][2], acc2[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) { acc1[i][j] = f64x4{0, 0, 0, 0}; acc2[i][j] = f64x4{0, 0, 0, 0}; }
    double x1[4], x2[4], y1[2], y2[2];
    for (int i = 0; i < 4; ++i) { x1[i] = xa[(lane + 64 * i) & 8191]; x2[i] = xa[(lane + 64 * i + 256) & 8191]; }
    for (int j = 0; j < 2; ++j) { y1[j] = yb[(lane + 64 * j) & 8191]; y2[j] = yb[(lane + 64 * j + 128) & 8191]; }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    unsigned base = lane * 8;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {
            const double* b = lds + ((it & 1) ? 3072 : 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) { x1[i] = b[(base + 128 * i) & 2047]; x2[i] = b[(base + 128 * i + 64) & 2047]; }
#pragma unroll
            for (int j = 0; j < 2; ++j) { y1[j] = b[2048 + ((base + 128 * j) & 1023)]; y2[j] = b[2048 + ((base + 128 * j + 64) & 1023)]; }
            base += 8;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ii = MODE == 0 ? 0 : i, jj = MODE == 0 ? 0 : j;
                acc1[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[ii], y1[jj], acc1[i][j], 0, 0, 0);
                acc2[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x2[ii], y2[jj], acc2[i][j], 0, 0, 0);
            }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    double s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 4; ++r) s += acc1[i][j][r] + acc2[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
    double *xa, *yb, *out; unsigned long long* clk;
    CK(hipMalloc(&xa, 8192 * 8)); CK(hipMalloc(&yb, 8192 * 8)); CK(hipMalloc(&out, 512 * 256 * 8)); CK(hipMalloc(&clk, 1024 * 8));
    std::vector<double> hx(8192), hy(8192);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000, grid = 512;
    for (int data = 0; data < 3; ++data) {
        // 0: small integers (few mantissa bits), 1: sums of pixels (f32 sums: <= 28 bits) x cosines (53 bits): the DCT's operands, 2: full random mantissas
        srand(1);
        for (int i = 0; i < 8192; ++i) {
            const double u = rand() / (double)RAND_MAX, v = rand() / (double)RAND_MAX;
            hx[i] = data == 0 ? (double)(i % 7 - 3) : data == 1 ? (double)((float)u + (float)v) - 1.0 : (u - 0.5) * 3.1415926535897931;
            hy[i] = data == 0 ? (double)(i % 5 - 2) : cos(3.14159265358979 * (2 * (i % 977) + 1) * (i % 131) / 1954.0);
        }
        CK(hipMemcpy(xa, hx.data(), 8192 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(yb, hy.data(), 8192 * 8, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 3; ++mode) {
            auto launch = [&]() {
                if (mode == 0) loop<0><<<grid, 256>>>(out, xa, yb, iters, clk);
                if (mode == 1) loop<1><<<grid, 256>>>(out, xa, yb, iters, clk);
                if (mode == 2) loop<2><<<grid, 256>>>(out, xa, yb, iters, clk);
            };
            for (int w = 0; w < 200; ++w) launch();              // settle (~2 s)
            CK(hipDeviceSynchronize());
            const int reps = 100;
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> hc(1024);
            CK(hipMemcpy(hc.data(), clk, 1024 * 8, hipMemcpyDeviceToHost));
            double ghz = 0; for (int b = 0; b < 512; ++b) ghz += (double)hc[2 * b] / ((double)hc[2 * b + 1] * 10.0); ghz /= 512;
            const double flop = (double)reps * grid * 4.0 * iters * 16 * 2048.0;
            printf("data %d (%s) mode %d (%s): %.1f TFLOP/s, shader clock %.2f GHz, %.2f ms per launch\n", data,
                   data == 0 ? "small integers" : data == 1 ? "pixel sums x cosines" : "random mantissas", mode,
                   mode == 0 ? "constant fragments" : mode == 1 ? "register fragments" : "fragments from LDS", flop / (ms * 1e-3) / 1e12, ghz, ms / reps);
        }
    }
    return 0;
}
