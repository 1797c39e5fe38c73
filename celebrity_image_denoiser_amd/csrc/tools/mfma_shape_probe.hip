// mfma_shape_probe.hip — does the fp16 MFMA shape change the sustained rate under the power cap?  (not part of the product)
// Two loops on random data, operands re-read from LDS every step (ds_read_b128), same FLOPs per wave and step:
//   A: 4 x v_mfma_f32_32x32x16_f16 (2x2 tiles of a 64x64 wave tile, K = 16)      B: 16 x v_mfma_f32_16x16x32_f16 (4x4 tiles, K = 32) per TWO A-steps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ void __launch_bounds__(256, 2) k_probe(const f16x8* src, float* out, int iters) {
    __shared__ f16x8 lds[2048];                              // 32 KiB of operands
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = src[(blockIdx.x * 2048 + i) & 0xffff];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f16x8* p = lds + wave * 64 + lane;
    if (SHAPE == 0) {
        f32x16 acc[2][2] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {                    // 4 K-steps of 16: 16 MFMAs, 16 LDS reads
                f16x8 a0 = p[(s * 4 + 0) * 256 & 1792], a1 = p[((s * 4 + 1) * 256) & 1792], b0 = p[((s * 4 + 2) * 256) & 1792], b1 = p[((s * 4 + 3) * 256) & 1792];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        float s = 0;
        for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f32x4 acc[4][4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {                    // 2 K-steps of 32: 32 MFMAs, 16 LDS reads — same FLOPs as above
                f16x8 a[4], b[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { a[q] = p[((s * 8 + q) * 256) & 1792]; b[q] = p[((s * 8 + 4 + q) * 256) & 1792]; }
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[n], acc[m][n], 0, 0, 0);
            }
        }
        float s = 0;
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

int main() {
    std::vector<_Float16> h(65536 * 8);
    unsigned s = 1;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((int)(s >> 9) % 2001 - 1000) / 1000.0f); }
    f16x8* src; float* out;
    CK(hipMalloc(&src, h.size() * 2)); CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 2048 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000, grid = 2048;
    const double flops = (double)grid * 4 * iters * 16 * 32768.0;   // per wave and iteration: 16 x 32x32x16 MFMAs' worth
    for (int rep = 0; rep < 3; ++rep)
        for (int shape = 0; shape < 2; ++shape) {
            CK(hipEventRecord(e0));
            if (shape == 0) hipLaunchKernelGGL(k_probe<0>, dim3(grid), dim3(256), 0, 0, src, out, iters);
            else hipLaunchKernelGGL(k_probe<1>, dim3(grid), dim3(256), 0, 0, src, out, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::printf("%s  %.2f ms  %.1f TFLOP/s\n", shape == 0 ? "32x32x16" : "16x16x32", ms, flops / (ms * 1e-3) / 1e12);
        }
    return 0;
}
