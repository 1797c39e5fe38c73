// headtail_bench.hip — timing experiments on the two HBM-bound kernels (not part of the product).
// Variants: one phase removed at a time (ABLATE), and fewer workgroups per CU (extra dynamic LDS).
#include "../conv_kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
using namespace cid;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)
struct Variant { std::string name; std::function<void(hipStream_t)> run; };

template <int ABLATE>
static Variant head(const char* name, int N, int H, int W, float* in, float* w, float* b, float* out, int extra_lds, int tpw = 0) {
    HeadArgs a{};
    a.in = in; a.w = w; a.bias = b; a.out = out; a.N = N; a.H = H; a.W = W; a.src = Window{0, 0, H, W};
    a.tiles_x = (W + TILE_W - 1) / TILE_W; a.tiles_y = (H + TILE_H - 1) / TILE_H;
    a.tiles_total = N * a.tiles_x * a.tiles_y; tile_groups(a);
    if (tpw > 0) { a.tiles_per_wg = tpw; a.groups_total = (a.tiles_total + tpw - 1) / tpw; a.groups_per_xcd = (a.groups_total + 7) / 8; }
    a.rcp_x = tile_rcp(a.tiles_x); a.rcp_xy = tile_rcp(a.tiles_x * a.tiles_y);
    const int grid = 8 * a.groups_per_xcd;
    return {name, [=](hipStream_t s) { hipLaunchKernelGGL((k_conv_head<false, ABLATE>), dim3(grid), dim3(THREADS), extra_lds, s, a); }};
}
template <int ABLATE>
static Variant tail(const char* name, int N, int H, int W, float* in, float* w, float* b, float* out, int extra_lds, int tpw = 0) {
    TailArgs a{};
    a.in = in; a.w = w; a.bias = b; a.out = out; a.N = N; a.H = H; a.W = W; a.crop = Window{0, 0, H, W};
    a.tiles_x = (W + TILE_W - 1) / TILE_W; a.tiles_y = (H + TILE_H - 1) / TILE_H;
    a.tiles_total = N * a.tiles_x * a.tiles_y; tile_groups(a);
    if (tpw > 0) { a.tiles_per_wg = tpw; a.groups_total = (a.tiles_total + tpw - 1) / tpw; a.groups_per_xcd = (a.groups_total + 7) / 8; }
    a.rcp_x = tile_rcp(a.tiles_x); a.rcp_xy = tile_rcp(a.tiles_x * a.tiles_y);
    const int grid = 8 * a.groups_per_xcd;
    return {name, [=](hipStream_t s) { hipLaunchKernelGGL((k_conv_tail<false, false, ABLATE>), dim3(grid), dim3(THREADS), extra_lds, s, a); }};
}

template <int ABLATE>
static Variant tail2(const char* name, int N, int H, int W, float* in, float* w, float* b, float* out, int rows = 0) {
    Tail2Args a{};
    a.in = in; a.w = w; a.bias = b; a.out = out; a.N = N; a.H = H; a.W = W; a.crop = Window{0, 0, H, W};
    tail2_plan(a, rows);
    const int grid = a.groups_total;
    return {name, [=](hipStream_t s) { hipLaunchKernelGGL((k_conv_tail2<false, ABLATE>), dim3(grid), dim3(THREADS), 0, s, a); }};
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 256, H = 128, W = 128;
    hipStream_t s; CK(hipStreamCreate(&s));
    float *img, *act, *w, *b;
    CK(hipMalloc(&img, (size_t)N * 3 * H * W * 4)); CK(hipMemset(img, 0, (size_t)N * 3 * H * W * 4));
    CK(hipMalloc(&act, (size_t)N * 64 * H * W * 4)); CK(hipMemset(act, 0, (size_t)N * 64 * H * W * 4));
    CK(hipMalloc(&w, 1 << 16)); CK(hipMemset(w, 0, 1 << 16));
    CK(hipMalloc(&b, 1 << 10)); CK(hipMemset(b, 0, 1 << 10));
    std::vector<Variant> v;
    v.push_back(head<0>("head base (4 WG/CU, 4 tiles/WG)", N, H, W, img, w, b, act, 0));
    v.push_back(head<0>("head 1 tile/WG", N, H, W, img, w, b, act, 0, 1));
    v.push_back(head<0>("head 2 tiles/WG", N, H, W, img, w, b, act, 0, 2));
    v.push_back(head<0>("head 8 tiles/WG", N, H, W, img, w, b, act, 0, 8));
    v.push_back(head<0>("head 16 tiles/WG", N, H, W, img, w, b, act, 0, 16));
    v.push_back(head<0>("head 3 WG/CU", N, H, W, img, w, b, act, 12 * 1024));
    v.push_back(head<0>("head 2 WG/CU", N, H, W, img, w, b, act, 36 * 1024));
    v.push_back(head<1>("head no-input-loads", N, H, W, img, w, b, act, 0));
    v.push_back(head<2>("head no-mfma", N, H, W, img, w, b, act, 0));
    v.push_back(head<4>("head no-stores", N, H, W, img, w, b, act, 0));
    v.push_back(head<3>("head stores only", N, H, W, img, w, b, act, 0));
    v.push_back(tail<0>("tail base (3 WG/CU)", N, H, W, act, w, b, img, 0));
    v.push_back(tail<0>("tail 2 WG/CU", N, H, W, act, w, b, img, 16 * 1024));
    v.push_back(tail<0>("tail 1 tile/WG", N, H, W, act, w, b, img, 0, 1));
    v.push_back(tail<0>("tail 2 tiles/WG", N, H, W, act, w, b, img, 0, 2));
    v.push_back(tail<0>("tail 8 tiles/WG", N, H, W, act, w, b, img, 0, 8));
    v.push_back(tail<0>("tail 16 tiles/WG", N, H, W, act, w, b, img, 0, 16));
    v.push_back(tail<1>("tail no-input-loads", N, H, W, act, w, b, img, 0));
    v.push_back(tail<2>("tail no-mfma", N, H, W, act, w, b, img, 0));
    v.push_back(tail<8>("tail no-epilogue", N, H, W, act, w, b, img, 0));
    v.push_back(tail<10>("tail loads only", N, H, W, act, w, b, img, 0));
    v.push_back(tail2<0>("tail2 base (default bands)", N, H, W, act, w, b, img));
    v.push_back(tail2<0>("tail2 64-row bands", N, H, W, act, w, b, img, 64));
    v.push_back(tail2<0>("tail2 32-row bands", N, H, W, act, w, b, img, 32));
    v.push_back(tail2<0>("tail2 16-row bands", N, H, W, act, w, b, img, 16));
    v.push_back(tail2<0>("tail2 128-row bands", N, H, W, act, w, b, img, 128));
    v.push_back(tail2<1>("tail2 no-dma", N, H, W, act, w, b, img));
    v.push_back(tail2<2>("tail2 no-mfma", N, H, W, act, w, b, img));
    v.push_back(tail2<4>("tail2 no-gather", N, H, W, act, w, b, img));
    v.push_back(tail2<6>("tail2 dma only", N, H, W, act, w, b, img));
    std::vector<std::vector<float>> ms(v.size());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& x : v) x.run(s);
    CK(hipStreamSynchronize(s));
    for (int r = 0; r < 9; ++r)
        for (size_t i = 0; i < v.size(); ++i) {
            CK(hipEventRecord(e0, s)); v[i].run(s); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); ms[i].push_back(t);
        }
    CK(hipGetLastError());
    const double gb = (double)N * H * W * 67 * 4 / 1e9;
    for (size_t i = 0; i < v.size(); ++i) {
        std::sort(ms[i].begin(), ms[i].end());
        const float med = ms[i][ms[i].size() / 2];
        std::printf("%-26s median %7.4f ms  min %7.4f ms  %6.0f GB/s\n", v[i].name.c_str(), med, ms[i][0], gb / (med * 1e-3));
    }
    return 0;
}
