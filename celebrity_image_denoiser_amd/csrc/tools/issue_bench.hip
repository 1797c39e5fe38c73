// issue_bench.hip — how long does ONE wave spend ISSUING vector-memory instructions (not waiting for them)?
// s_memtime around N back-to-back instructions, no waitcnt inside; one workgroup of 4 waves per CU, L2-warm data.
// Variants: LDS-DMA dwordx4 with an M0 write per instruction (what the Winograd kernels do), LDS-DMA with one M0 value and
// immediate offsets, plain buffer loads to registers, global stores; lane addresses 16 B apart (dense) or 512 B apart
// (one pixel per 4 lanes, the kernels' pattern).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE, int N>
__global__ void __launch_bounds__(256) k(const float* src, float* dst, int bytes, int stride_bytes, unsigned long long* out) {
    __shared__ f32x4 lds[4096];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, bytes, 0x00020000);
    // lane -> (pixel = lane/4, 16-byte group = lane%4) with `stride_bytes` between pixels; a block of N*16 pixels per wave
    const unsigned base = (unsigned)(((blockIdx.x * 4 + wave) * N * 16) * stride_bytes);
    unsigned voff[N];
#pragma unroll
    for (int m = 0; m < N; ++m) voff[m] = base + (unsigned)((m * 16 + (lane >> 2)) * stride_bytes + (lane & 3) * 16);
    const unsigned lds_base = (unsigned)(uintptr_t)(&lds[0]) + wave * N * 1024;
    f32x4 r[N];
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int m = 0; m < N; ++m) {
        if (MODE == 0) {
            const unsigned d = lds_base + m * 1024;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff[m]), "s"(rsrc), "s"(d) : "memory");
        } else if (MODE == 1) {
            r[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[m], 0, 0));
        } else if (MODE == 2) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(dst) + voff[m]) = f32x4{1.f, 2.f, 3.f, (float)m};
        } else if (MODE == 3) {   // LDS-DMA one dword per lane (the pre-gfx950 form), 4 instructions per 16 bytes
            const unsigned d = lds_base + m * 1024;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, 0 offen lds" ::"v"(voff[m]), "s"(rsrc), "s"(d) : "memory");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (MODE == 1) {
        f32x4 acc = r[0];
#pragma unroll
        for (int m = 1; m < N; ++m) acc += r[m];
        if (acc[0] == 123.456f) dst[threadIdx.x] = acc[1];
    }
    if (MODE == 0 || MODE == 3) { __syncthreads(); if (lds[threadIdx.x][0] == 123.456f) dst[threadIdx.x] = 1.f; }
    if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = t1 - t0; out[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t1; }
}

template <int MODE, int N>
static int run(const char* name, const float* src, float* dst, int bytes, int stride, int blocks, unsigned long long* dout) {
    std::vector<unsigned long long> h((size_t)blocks * 8);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k<MODE, N>), dim3(blocks), dim3(256), 0, 0, src, dst, bytes, stride, dout);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> iss, wait;
    for (int i = 0; i < blocks * 4; ++i) { iss.push_back(h[(size_t)i * 2]); wait.push_back(h[(size_t)i * 2 + 1]); }
    std::sort(iss.begin(), iss.end()); std::sort(wait.begin(), wait.end());
    std::printf("%-44s stride %4d B, %3d blocks: issue %6.0f cycles per instruction (median wave; p90 %6.0f), then wait %6llu\n", name, stride, blocks,
                (double)iss[iss.size() / 2] / N, (double)iss[iss.size() * 9 / 10] / N, wait[wait.size() / 2]);
    return 0;
}

int main() {
    const int bytes = 512 << 20;
    float *src, *dst; unsigned long long* dout;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&dout, 4096 * 8 * 8));
    CK(hipMemset(src, 0, bytes)); CK(hipMemset(dst, 0, bytes));
    for (int blocks : {1, 256, 1024}) {
        for (int stride : {64, 512}) {
            if (run<0, 8>("LDS-DMA dwordx4, M0 write each", src, dst, bytes, stride, blocks, dout)) return 1;
            if (run<3, 8>("LDS-DMA dword,   M0 write each", src, dst, bytes, stride, blocks, dout)) return 1;
            if (run<1, 8>("buffer_load_dwordx4 to registers", src, dst, bytes, stride, blocks, dout)) return 1;
            if (run<2, 8>("global_store_dwordx4", src, dst, bytes, stride, blocks, dout)) return 1;
        }
    }
    return 0;
}
