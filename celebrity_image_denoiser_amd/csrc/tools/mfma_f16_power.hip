// mfma_f16_power.hip — what the fp16 MFMA shapes SUSTAIN under the 1,400 W cap, with and without the LDS reads of k_conv3x3_h16 beside them.
// Each variant runs for ~2 s (launches back to back) so that the power controller settles; operands are pseudo-random halfs (toggle rate matters).
//   hipcc -O3 --offload-arch=gfx950 -o mfma_f16_power mfma_f16_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

// SHAPE 0: v_mfma_f32_16x16x32_f16, 16 accumulators (64 registers), 48 MFMAs per step from 8 "A" + 12 "B" operand quads
// SHAPE 1: v_mfma_f32_32x32x16_f16, 4 accumulators (64 registers), 24 MFMAs per step from the same 20 quads
// LDSQ: operand quads re-read from LDS every step (20 = the product's ratio; 0 = registers only)
template <int SHAPE, int LDSQ>
__global__ void __launch_bounds__(256, 3) k_f16(const f32x4* src, float* out, int steps) {
    __shared__ f32x4 lds[2048];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2048; i += 256) lds[i] = src[i];
    __syncthreads();
    f32x4 q[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) q[i] = lds[(i * 64 + lane * 5) & 2047];
    f32x4 acc16[16]; f32x16 acc32[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc16[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
    for (int s = 0; s < steps; ++s) {
        if (LDSQ) {
#pragma unroll
            for (int i = 0; i < LDSQ; ++i) q[i] = lds[((s * 7 + i) * 64 + lane) & 2047];
        }
        if (SHAPE == 0) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    acc16[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, q[(t & 3) * 2 + (dy & 1)]), __builtin_bit_cast(f16x8, q[8 + dy * 4 + (t >> 2)]), acc16[t], 0, 0, 0);
        } else {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc32[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, q[(t & 1) * 4 + ks * 2 + (dy & 1)]), __builtin_bit_cast(f16x8, q[8 + dy * 4 + (t >> 1) * 2 + ks]), acc32[t], 0, 0, 0);
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) r += acc32[i][k];
    out[blockIdx.x * 256 + tid] = r;
}

// fp32: v_mfma_f32_16x16x4_f32 (the Winograd kernels' instruction), 24 accumulator tiles (96 registers) as in k_wino42_conv; per 48 MFMAs NV
// plain v_fma_f32 (the V build: 1.26 per MFMA measured = 60 per 48) and NL ds_read_b64 (the raw-tile reads)
template <int NV, int NL>
__global__ void __launch_bounds__(256, 2) k_f32(const f32x4* src, float* out, int steps) {
    __shared__ f32x4 lds[2048];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2048; i += 256) lds[i] = src[i];
    __syncthreads();
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2* l2 = reinterpret_cast<const f32x2*>(lds);
    f32x4 acc[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[12], u[8];
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = lds[(i * 64 + lane) & 2047][i & 3];
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = lds[(i * 64 + lane + 777) & 2047][i & 3];
    f32x2 r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = f32x2{0.f, 0.f};
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int g = 0; g < 12; ++g) {        // 12 groups of 4 MFMAs, side work spread evenly
#pragma unroll
            for (int j = 0; j < (NL + 11 - g) / 12; ++j) r[(g + j) & 7] = l2[(((s * 5 + g) * 64 + lane) * 2 + j) & 4095];
#pragma unroll
            for (int j = 0; j < (NV + 11 - g) / 12; ++j) v[(g + j * 5) % 12] = __builtin_fmaf(v[(g + j * 5) % 12], 0.999f, r[(g + j) & 7][0]);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[(g * 4 + i) % 24] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[(g + i) % 12], u[(g * 4 + i) & 7], acc[(g * 4 + i) % 24], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + tid] = t;
}

template <int NV, int NL>
void run32(const char* name, const f32x4* src, float* out, double seconds) {
    const int grid = 256 * 2 * 8, steps = 300;
    hipLaunchKernelGGL((k_f32<NV, NL>), dim3(grid), dim3(256), 0, 0, src, out, steps);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double first = 0, last = 0; int n = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k_f32<NV, NL>), dim3(grid), dim3(256), 0, 0, src, out, steps);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        const double tf = (double)grid * 4 * steps * 48 * 2048.0 * 50 / (t * 1e-3) / 1e12;   // 2*16*16*4 flops per MFMA
        if (n == 0) first = tf;
        last = tf; ++n;
    }
    std::printf("%-44s first 50 launches %7.1f TFLOP/s   settled (after %.1f s) %7.1f TFLOP/s  = %.3f of 157.3\n", name, first, seconds, last, last / 157.3);
    std::fflush(stdout);
}

// The wave tile priced in DESIGN.md section 5 (128 pixels x 64 channels): 32 accumulator tiles (128 registers), two waves per SIMD, per step
// 12 "A" + 12 "B" operand quads for 96 MFMAs of 16x16x32 — 0.25 ds_read_b128 per MFMA instead of 0.42
template <int LDSQ>
__global__ void __launch_bounds__(256, 2) k_f16_wide(const f32x4* src, float* out, int steps) {
    __shared__ f32x4 lds[2048];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2048; i += 256) lds[i] = src[i];
    __syncthreads();
    f32x4 q[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) q[i] = lds[(i * 64 + lane * 5) & 2047];
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < steps; ++s) {
        if (LDSQ) {
#pragma unroll
            for (int i = 0; i < LDSQ; ++i) q[i] = lds[((s * 7 + i) * 64 + lane) & 2047];
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int t = 0; t < 32; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, q[(t & 7) + (dy & 1) * 4]), __builtin_bit_cast(f16x8, q[12 + dy * 4 + (t >> 3)]), acc[t], 0, 0, 0);
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + tid] = r;
}

template <int LDSQ>
void run_wide(const char* name, const f32x4* src, float* out, double seconds) {
    const int grid = 256 * 2 * 8, steps = 450;
    hipLaunchKernelGGL((k_f16_wide<LDSQ>), dim3(grid), dim3(256), 0, 0, src, out, steps);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double first = 0, last = 0; int n = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k_f16_wide<LDSQ>), dim3(grid), dim3(256), 0, 0, src, out, steps);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        const double tf = (double)grid * 4 * steps * 96 * 16384.0 * 50 / (t * 1e-3) / 1e12;
        if (n == 0) first = tf;
        last = tf; ++n;
    }
    std::printf("%-44s first 50 launches %7.1f TFLOP/s   settled (after %.1f s) %7.1f TFLOP/s  = %.3f of 2,500\n", name, first, seconds, last, last / 2500.0);
    std::fflush(stdout);
}

template <int SHAPE, int LDSQ>
void run(const char* name, const f32x4* src, float* out, double seconds) {
    const int grid = 256 * 3 * 8, steps = 600;             // 8 rounds of three workgroups per CU, ~0.3 ms per launch
    hipLaunchKernelGGL((k_f16<SHAPE, LDSQ>), dim3(grid), dim3(256), 0, 0, src, out, steps);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double first = 0, last = 0; int n = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k_f16<SHAPE, LDSQ>), dim3(grid), dim3(256), 0, 0, src, out, steps);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        const double tf = (double)grid * 4 * steps * 48 * 16384.0 * 50 / (t * 1e-3) / 1e12;   // 48 16x16x32 (= 24 32x32x16) MFMAs per step, 2*16*16*32 flops each
        if (n == 0) first = tf;
        last = tf; ++n;
    }
    std::printf("%-44s first 50 launches %7.1f TFLOP/s   settled (after %.1f s) %7.1f TFLOP/s  = %.3f of 2,500\n", name, first, seconds, last, last / 2500.0);
    std::fflush(stdout);
}

int main(int argc, char** argv) {
    const double sec = argc > 1 ? std::atof(argv[1]) : 2.0;
    f32x4* src; float* out;
    CK(hipMalloc(&src, 2048 * sizeof(f32x4))); CK(hipMalloc(&out, (size_t)256 * 3 * 8 * 256 * sizeof(float)));
    {   // pseudo-random halfs in (-2, 2): sign/exponent/mantissa bits all toggle
        _Float16* h = (_Float16*)std::malloc(2048 * 16);
        unsigned x = 12345u;
        for (int i = 0; i < 2048 * 8; ++i) { x = x * 1664525u + 1013904223u; h[i] = (_Float16)(((int)(x >> 8) % 4001 - 2000) * 1e-3f); }
        CK(hipMemcpy(src, h, 2048 * 16, hipMemcpyHostToDevice)); std::free(h);
    }
    run<0, 0>("16x16x32 f16, registers only", src, out, sec);
    run<1, 0>("32x32x16 f16, registers only", src, out, sec);
    run<0, 20>("16x16x32 f16 + 20 ds_read_b128 per 48", src, out, sec);
    run<1, 20>("32x32x16 f16 + 20 ds_read_b128 per 24", src, out, sec);
    run<0, 10>("16x16x32 f16 + 10 ds_read_b128 per 48", src, out, sec);
    run<0, 0>("16x16x32 f16, registers only (again)", src, out, sec);
    run_wide<24>("16x16x32 f16, 128 acc, 2 waves/SIMD, 24 reads/96", src, out, sec);
    run_wide<0>("16x16x32 f16, 128 acc, 2 waves/SIMD, registers", src, out, sec);
    run32<0, 0>("16x16x4 f32, registers only", src, out, sec);
    run32<60, 0>("16x16x4 f32 + 60 v_fma per 48 (1.26/MFMA)", src, out, sec);
    run32<60, 16>("16x16x4 f32 + 60 v_fma + 16 ds_read_b64", src, out, sec);
    run32<30, 0>("16x16x4 f32 + 30 v_fma per 48", src, out, sec);
    return 0;
}
