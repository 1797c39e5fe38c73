#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const f32x4* src, int nbytes, f32x4* out, int soff) {
    __shared__ f32x4 lds[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lds[threadIdx.x] = f32x4{-7.f, -7.f, -7.f, -7.f};       // poison: an OOB lane must overwrite it with zeros
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, nbytes, 0x00020000);
    // lane%5==4 -> out of range offset (must deliver zeros); others -> element (wave*64 + lane^1) minus soff
    const unsigned voff = (lane % 5 == 4) ? 0x7ffffff0u : (unsigned)((wave * 64 + (lane ^ 1)) * 16 - soff);
    const unsigned ldsbase = (unsigned)(uintptr_t)(&lds[__builtin_amdgcn_readfirstlane(wave) * 64]);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(voff), "s"(rsrc), "s"(ldsbase), "s"(soff) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x];
}
int main() {
    f32x4 h[256], o[256];
    for (int i = 0; i < 256; ++i) h[i] = f32x4{(float)i, i + 0.25f, i + 0.5f, i + 0.75f};
    f32x4 *d, *dout;
    if (hipMalloc(&d, sizeof h) != hipSuccess || hipMalloc(&dout, sizeof o) != hipSuccess) return 2;
    (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int soff : {0, 64}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, (int)sizeof h, dout, soff);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        (void)hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) {
            int lane = i & 63, w = i >> 6;
            float exp0 = (lane % 5 == 4) ? 0.f : (float)(w * 64 + (lane ^ 1));
            if (o[i][0] != exp0 || o[i][3] != (lane % 5 == 4 ? 0.f : exp0 + 0.75f)) { if (bad < 4) printf("soff %d mismatch at %d: got %f exp %f\n", soff, i, o[i][0], exp0); ++bad; }
        }
        printf("buffer LDS-DMA with OOB zero fill, soffset %d: %s (%d bad)\n", soff, bad ? "FAIL" : "PASS", bad);
        bad_total += bad;
    }
    return bad_total != 0;
}
