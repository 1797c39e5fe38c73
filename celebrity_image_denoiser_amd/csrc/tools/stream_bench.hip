// stream_bench.hip — how fast can 1.07 GB be READ on MI355X under the access patterns the tail/head kernels could use?
// (not part of the product).  Every variant reads the same [N][128][128][64] fp32 tensor once with 16-byte loads and
// keeps a checksum alive; what differs is who reads what, when:
//   linear      : workgroup g reads a contiguous 64 KiB block, blocks in launch order
//   rows        : workgroup = (image, band of R rows), walks its rows top to bottom, one 32 KiB row per step
//   rows-half   : same, but each row in two passes: the low 128 B of every pixel, then the high 128 B (32-channel chunks)
//   tiles-half  : workgroup = 10x34-pixel halo tile (8x32 outputs), low halves then high halves (round 1's tail)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void sink(f32x4 v, float* out) { if (v[0] + v[1] + v[2] + v[3] == 123.456f) out[0] = v[0]; }

__global__ void __launch_bounds__(256) k_linear(const f32x4* in, float* out, int blocks_per_wg) {
    const size_t base = (size_t)blockIdx.x * blocks_per_wg * 1024;     // 1024 quads = 16 KiB per block
    f32x4 acc = {0, 0, 0, 0};
    for (int b = 0; b < blocks_per_wg; ++b) {
        f32x4 v[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = in[base + (size_t)b * 1024 + m * 256 + threadIdx.x];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc += v[m];
    }
    sink(acc, out);
}
// band of R rows of image n; per row 128 px x 16 quads = 2048 quads; HALF: two passes of 8 quads per pixel
template <bool HALF, int DEPTH>
__global__ void __launch_bounds__(256) k_rows(const f32x4* in, float* out, int R, int bands) {
    const int n = blockIdx.x / bands, band = blockIdx.x % bands;
    const f32x4* img = in + (size_t)n * 128 * 2048 + (size_t)band * R * 2048;
    f32x4 acc = {0, 0, 0, 0};
    const int t = threadIdx.x;
    for (int r = 0; r < R; r += DEPTH) {
        f32x4 v[DEPTH][8];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const f32x4* row = img + (size_t)(r + d) * 2048;
            if (HALF) {
#pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf)
#pragma unroll
                    for (int m = 0; m < 4; ++m) { const int s = m * 256 + t; v[d][hlf * 4 + m] = row[(s >> 3) * 16 + hlf * 8 + (s & 7)]; }
            } else {
#pragma unroll
                for (int m = 0; m < 8; ++m) v[d][m] = row[m * 256 + t];
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int m = 0; m < 8; ++m) acc += v[d][m];
    }
    sink(acc, out);
}
// round-1 tail pattern: 10x34 halo tile, 32-channel chunks: 340 px x 8 quads = 2720 quads per chunk, 11 per thread
__global__ void __launch_bounds__(256) k_tiles(const f32x4* in, float* out, int tiles_per_xcd, int total) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = xcd * tiles_per_xcd + slot;
    if (tile >= total || slot >= tiles_per_xcd) return;
    const int n = tile >> 6, ty = (tile >> 2) & 15, tx = tile & 3;
    f32x4 acc = {0, 0, 0, 0};
    for (int hlf = 0; hlf < 2; ++hlf) {
        f32x4 v[11];
#pragma unroll
        for (int it = 0; it < 11; ++it) {
            const int s = it * 256 + threadIdx.x, p = s >> 3, c = s & 7;
            const int hy = p / 34, hx = p - hy * 34, gy = ty * 8 - 1 + hy, gx = tx * 32 - 1 + hx;
            const bool ok = s < 2720 && (unsigned)gy < 128u && (unsigned)gx < 128u;
            v[it] = ok ? in[((size_t)n * 16384 + gy * 128 + gx) * 16 + hlf * 8 + c] : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int it = 0; it < 11; ++it) acc += v[it];
    }
    sink(acc, out);
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 256;
    const size_t quads = (size_t)N * 128 * 128 * 16;
    f32x4* in; float* out;
    CK(hipMalloc(&in, quads * 16)); CK(hipMemset(in, 0, quads * 16)); CK(hipMalloc(&out, 64));
    hipStream_t s; CK(hipStreamCreate(&s));
    struct V { std::string name; std::function<void()> run; };
    std::vector<V> v;
    v.push_back({"linear 64 KiB/WG", [&] { hipLaunchKernelGGL(k_linear, dim3(quads / 4096), dim3(256), 0, s, in, out, 4); }});
    v.push_back({"linear 1 MiB/WG", [&] { hipLaunchKernelGGL(k_linear, dim3(quads / 65536), dim3(256), 0, s, in, out, 64); }});
    for (int R : {128, 64, 32, 16, 8}) {
        const int bands = 128 / R;
        v.push_back({"rows R=" + std::to_string(R) + " depth 2", [=] { hipLaunchKernelGGL((k_rows<false, 2>), dim3(N * bands), dim3(256), 0, s, in, out, R, bands); }});
        v.push_back({"rows-half R=" + std::to_string(R) + " depth 2", [=] { hipLaunchKernelGGL((k_rows<true, 2>), dim3(N * bands), dim3(256), 0, s, in, out, R, bands); }});
    }
    v.push_back({"rows R=64 depth 1", [=] { hipLaunchKernelGGL((k_rows<false, 1>), dim3(N * 2), dim3(256), 0, s, in, out, 64, 2); }});
    v.push_back({"rows R=64 depth 4", [=] { hipLaunchKernelGGL((k_rows<false, 4>), dim3(N * 2), dim3(256), 0, s, in, out, 64, 2); }});
    v.push_back({"tiles-half (round-1 tail loads)", [=] { const int total = N * 64, per = (total + 7) / 8; hipLaunchKernelGGL(k_tiles, dim3(8 * per), dim3(256), 0, s, in, out, per, total); }});
    std::vector<std::vector<float>> ms(v.size());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& x : v) x.run();
    CK(hipStreamSynchronize(s));
    for (int r = 0; r < 9; ++r)
        for (size_t i = 0; i < v.size(); ++i) {
            CK(hipEventRecord(e0, s)); v[i].run(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); ms[i].push_back(t);
        }
    CK(hipGetLastError());
    for (size_t i = 0; i < v.size(); ++i) {
        std::sort(ms[i].begin(), ms[i].end());
        std::printf("%-34s median %7.4f ms  min %7.4f ms  %6.0f GB/s\n", v[i].name.c_str(), ms[i][4], ms[i][0], quads * 16 / 1e9 / (ms[i][4] * 1e-3));
    }
    return 0;
}
