// mix_bench.hip — how much do VALU / LDS / VMEM instructions beside the MFMA stream cost, at 1 and 2 waves per SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

// per iteration: 16 MFMAs (4 accumulators in rotation), NV v_fma, NL ds_read_b128, NG global_load_dwordx4 (1 KiB per wave)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NV, int NL, int NG, int LDSKB, bool PK = false>
__global__ void __launch_bounds__(256, 2) k_mix(const f32x4* gsrc, float* out, int iters) {
    __shared__ f32x4 lds[LDSKB * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < LDSKB * 64; i += 256) lds[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float va[16];
    f32x2 vp[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) va[i] = 0.001f * (tid + i);
#pragma unroll
    for (int i = 0; i < 8; ++i) vp[i] = f32x2{0.001f * tid, 0.002f * (tid + i)};
    f32x4 la[8], ga[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) la[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) ga[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 0.5f + lane * 1e-3f, b = 0.25f;
    const f32x4* gp = gsrc + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // a quarter of the side work per group of 4 MFMAs
#pragma unroll
            for (int j = 0; j < NL / 4; ++j) la[(g * (NL / 4) + j) & 7] = lds[((it * 13 + g * 5 + j * 7) & (LDSKB - 1)) * 64 + lane];
#pragma unroll
            for (int j = 0; j < NG / 4; ++j) ga[(g * (NG / 4) + j) & 3] = gp[((it * 4 + g + j) & 63) * 64];
#pragma unroll
            for (int j = 0; j < NV / 4; ++j) {
                if (PK) { const int q = (g * (NV / 4) + j) & 7; vp[q] = vp[q] * f32x2{1.0001f, 0.9999f} + f32x2{0.5f, 0.25f}; }   // v_pk_fma_f32 / v_pk_mul+add
                else { const int q = (g * (NV / 4) + j) & 15; va[q] = __builtin_fmaf(va[q], 1.0001f, 0.5f); }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + la[i][0] + ga[i][0], b, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += va[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += vp[i][0] + vp[i][1];
    out[blockIdx.x * 256 + tid] = s;
}

template <int NV, int NL, int NG, int LDSKB, bool PK = false>
void run(const char* name, const f32x4* src, float* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 400;
    const int grid = 256 * 8;
    hipLaunchKernelGGL((k_mix<NV, NL, NG, LDSKB, PK>), dim3(grid), dim3(256), 0, 0, src, out, iters);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_mix<NV, NL, NG, LDSKB, PK>), dim3(grid), dim3(256), 0, 0, src, out, iters);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); if (t < best) best = t;
    }
    const double flops = (double)grid * 4 * iters * 16 * 4096.0;
    std::printf("%-44s LDS %3d KiB/WG  %8.4f ms  %6.1f%% of 157.3 TF\n", name, LDSKB, best, 100 * flops / (best * 1e-3) / 1e12 / 157.3);
}

int main() {
    f32x4* src; float* out;
    CK(hipMalloc(&src, 64 * 64 * 16 + 4096)); CK(hipMemset(src, 0, 64 * 64 * 16 + 4096));
    CK(hipMalloc(&out, (size_t)256 * 8 * 256 * 4));
    // LDSKB = 64 -> 2 WG/CU (2 waves/SIMD); LDSKB = 128 -> 1 WG/CU (1 wave/SIMD)
    run<0, 0, 0, 64>("2 waves/SIMD: MFMA only", src, out);
    run<48, 0, 0, 64>("2 waves/SIMD: + 48 VALU per 16 MFMA", src, out);
    run<96, 0, 0, 64>("2 waves/SIMD: + 96 VALU", src, out);
    run<24, 0, 0, 64, true>("2 waves/SIMD: + 24 packed VALU (=48 flop-ops)", src, out);
    run<48, 0, 0, 64, true>("2 waves/SIMD: + 48 packed VALU", src, out);
    run<16, 0, 0, 64>("2 waves/SIMD: + 16 VALU", src, out);
    run<32, 0, 0, 64>("2 waves/SIMD: + 32 VALU", src, out);
    run<0, 8, 0, 64>("2 waves/SIMD: + 8 ds_read_b128", src, out);
    run<0, 0, 4, 64>("2 waves/SIMD: + 4 global_load_dwordx4", src, out);
    run<48, 8, 4, 64>("2 waves/SIMD: + 48 VALU + 8 LDS + 4 VMEM", src, out);
    run<0, 0, 0, 128>("1 wave/SIMD: MFMA only", src, out);
    run<48, 0, 0, 128>("1 wave/SIMD: + 48 VALU", src, out);
    run<96, 0, 0, 128>("1 wave/SIMD: + 96 VALU", src, out);
    run<0, 8, 0, 128>("1 wave/SIMD: + 8 ds_read_b128", src, out);
    run<0, 0, 4, 128>("1 wave/SIMD: + 4 global_load_dwordx4", src, out);
    run<48, 8, 4, 128>("1 wave/SIMD: + 48 VALU + 8 LDS + 4 VMEM", src, out);
    return 0;
}
