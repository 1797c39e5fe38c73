#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// each lane copies 16 B from src[perm(lane)] into LDS slot lane via LDS-DMA, then reads LDS back
__global__ void k(const f32x4* src, const f32x4* zero, f32x4* out) {
    __shared__ f32x4 lds[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4* g = (lane % 5 == 4) ? zero : src + (wave * 64 + (lane ^ 1));
    const unsigned ldsbase = (unsigned)(uintptr_t)(&lds[__builtin_amdgcn_readfirstlane(wave) * 64]);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(ldsbase) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x];
}
int main() {
    f32x4 h[256], z = {0,0,0,0}, o[256];
    for (int i = 0; i < 256; ++i) h[i] = f32x4{(float)i, i + 0.25f, i + 0.5f, i + 0.75f};
    f32x4 *d, *dz, *dout;
    hipMalloc(&d, sizeof h); hipMalloc(&dz, 64); hipMalloc(&dout, sizeof o);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice); hipMemset(dz, 0, 64);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, dz, dout);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) {
        int lane = i & 63, w = i >> 6;
        float exp0 = (lane % 5 == 4) ? 0.f : (float)(w * 64 + (lane ^ 1));
        if (o[i][0] != exp0 || (lane % 5 != 4 && o[i][3] != exp0 + 0.75f)) { if (bad < 5) printf("mismatch at %d: got %f exp %f\n", i, o[i][0], exp0); ++bad; }
    }
    printf("LDS-DMA test: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad);
    return bad != 0;
}
