// mfma_peak.hip — what v_mfma_f32_32x32x2_f32 sustains on this chip under our accumulator pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(const char* name, int blocks_per_cu, int iters, float* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 0.5f, 0.25f);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 0.5f, 0.25f);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); if (t < best) best = t;
    }
    const double flops = (double)grid * 4 * iters * 16 * NACC * 4096.0;
    std::printf("%-34s blocks/CU %d  %8.4f ms  %7.2f TFLOP/s (%.1f%%)\n", name, blocks_per_cu, best, flops / (best * 1e-3) / 1e12, 100 * flops / (best * 1e-3) / 1e12 / 157.3);
}

int main() {
    float* out; CK(hipMalloc(&out, (size_t)256 * 64 * 256 * sizeof(float)));  // largest grid below: 256*64 blocks x 256 threads
    for (int bpc : {1, 2, 3, 4}) {
        run<1>("1 acc (dependent chain)", bpc, 2000 / bpc, out);
        run<2>("2 acc", bpc, 1000 / bpc, out);
        run<4>("4 acc", bpc, 500 / bpc, out);
    }
    // short blocks, many of them: per-block launch overhead (like our 64 blocks/CU x 2304 MFMA/wave)
    run<4>("4 acc, 64 blocks/CU, 2304 MFMA/wave", 64, 36, out);
    run<4>("4 acc, 21 blocks/CU, 6912 MFMA/wave", 21, 108, out);
    return 0;
}
