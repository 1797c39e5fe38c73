// Experiment (round 4), third form of the split-operand prototype (see split_proto.hip): k_conv3x3_h16<..., F32IO = true> reads and writes the FP32
// tensors of the fp32 path — a drop-in shape for the Winograd launches.  While staging a 32-channel chunk of the halo tile it splits every fp32 value into
// hi = half(x) and lo = half(x - hi) (two sets of LDS planes), runs nine sub-steps per chunk (hi_x.hi_w, hi_x.lo_w, lo_x.hi_w per tap column; the host packs the
// weight sub-chunks in that order) on v_mfma_f32_16x16x32_f16 with fp32 accumulators, and stores fp32 (16 bytes per lane and pixel) — 66.6 KiB of LDS, two workgroups per CU.
//   1. numerics on hardware against a float64 convolution: plain (64 -> 64) and with the fused 2x2 max-pool (MODE 1);
//   2. time: the eight 3x3 layer shapes at B = 256 (the two pooling layers as MODE 1), singly and in forward order, beside the fp32 Winograd launches.
//   hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 -o tools/split_proto_f32 tools/split_proto_f32.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../conv_kernels.h"
#include "../conv_kernels_f16.h"
using namespace cid;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// packed weights in consumption order: [column block nb][chunk ck of 32 channels][j = 0..8][kh][channel group cg][lane = 16 kg + col][8]
//   j = 0..2: tap column kw = j of hi_w; j = 3..5: kw = j - 3 of lo_w; j = 6..8: kw = j - 6 of hi_w (met by lo_x)
static size_t packed_index(int cin, int co, int ci, int kh, int j) {
    const int nb = co >> 6, cg = co & 3, c = (co >> 2) & 15;
    const int ck = ci >> 5, kg = (ci >> 3) & 3, e = ci & 7;
    const int nchunk = cin / 32;
    return ((((((size_t)(nb * nchunk + ck) * 9 + j) * 3 + kh) * 4 + cg) * 64) + kg * 16 + c) * 8 + e;
}
static void pack_weights(const std::vector<float>& w, int C, int K, std::vector<_Float16>& out) {   // w[co][ci][kh][kw]
    out.assign((size_t)K * C * 27, (_Float16)0.f);
    for (int co = 0; co < K; ++co)
        for (int ci = 0; ci < C; ++ci)
            for (int kh = 0; kh < 3; ++kh)
                for (int kw = 0; kw < 3; ++kw) {
                    const float v = w[(((size_t)co * C + ci) * 3 + kh) * 3 + kw];
                    const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                    out[packed_index(C, co, ci, kh, kw)] = hi; out[packed_index(C, co, ci, kh, 3 + kw)] = lo; out[packed_index(C, co, ci, kh, 6 + kw)] = hi;
                }
}

template <int CIN, int COUT, int MODE>
static void launch(const float* in, const _Float16* w, const float* bias, float* out, float* pool, int N, int H, int W) {
    GemmConvArgsH a{};
    a.in = reinterpret_cast<const _Float16*>(in); a.w = w; a.bias = bias; a.out = reinterpret_cast<_Float16*>(out); a.pool = reinterpret_cast<_Float16*>(pool);
    a.N = N; a.Hin = H; a.Win = W; a.in_ps = CIN; a.Hc = H; a.Wc = W; a.Hs = H; a.Ws = W; a.out_ps = COUT; a.out_coff = 0;
    a.tiles_x = (W + TILE_W - 1) / TILE_W; a.tiles_y = (H + TILE_H - 1) / TILE_H; a.tiles_total = N * a.tiles_x * a.tiles_y;
    a.tiles_per_xcd = (a.tiles_total + 7) / 8;
    a.rcp_x = tile_rcp(a.tiles_x); a.rcp_xy = tile_rcp(a.tiles_x * a.tiles_y);
    a.walk = 0;
    constexpr int NB = COUT / NTILE;
    hipLaunchKernelGGL((k_conv3x3_h16<CIN, COUT, MODE, false, false, true>), dim3(8 * a.tiles_per_xcd * NB), dim3(THREADS), 0, 0, a);
}

__global__ void k_fill_f32(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        const float v = ((int)(x & 0xffffff) - 8388608) * (scale / 8388608.f);
        p[i] = v > 0.f ? v : 0.f;                      // post-ReLU-like
    }
}
__global__ void k_fill_h(_Float16* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (_Float16)(((int)(x & 0xffff) - 32768) * (scale / 32768.f));
    }
}

template <int MODE>
static void numerics(int H, int W) {
    constexpr int N = 2, C = 64, K = 64;
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<float> x((size_t)N * H * W * C), w((size_t)K * C * 9), b(K);
    for (auto& v : x) { const float r = u(rng); v = r < -0.3f ? 0.f : (r < 0.f ? (r + 0.3f) * -3e-3f : 2.f * r); }   // zeros, tiny values, values up to 2
    for (auto& v : w) v = 0.06f * u(rng);
    for (auto& v : b) v = 0.1f * u(rng);
    std::vector<_Float16> hw;
    pack_weights(w, C, K, hw);
    const int Hp = H / 2, Wp = W / 2;
    float *din, *dout, *dpool, *db; _Float16* dw;
    CK(hipMalloc(&din, x.size() * 4)); CK(hipMalloc(&dw, hw.size() * 2)); CK(hipMalloc(&dout, (size_t)N * H * W * K * 4)); CK(hipMalloc(&dpool, (size_t)N * Hp * Wp * K * 4 + 16)); CK(hipMalloc(&db, K * 4));
    CK(hipMemcpy(din, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), K * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dout, 0xff, (size_t)N * H * W * K * 4)); CK(hipMemset(dpool, 0xff, (size_t)N * Hp * Wp * K * 4));
    launch<C, K, MODE>(din, dw, db, dout, dpool, N, H, W);
    CK(hipDeviceSynchronize());
    std::vector<float> hout((size_t)N * H * W * K), hpool((size_t)N * Hp * Wp * K);
    CK(hipMemcpy(hout.data(), dout, hout.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hpool.data(), dpool, hpool.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> ref((size_t)N * H * W * K);
    double err = 0, err_f32 = 0, ymax = 0;
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < H; ++y)
            for (int xx = 0; xx < W; ++xx)
                for (int co = 0; co < K; ++co) {
                    double acc = b[co]; float accf = 0.f;
                    for (int kh = 0; kh < 3; ++kh)
                        for (int kw = 0; kw < 3; ++kw) {
                            const int yy = y + kh - 1, xq = xx + kw - 1;
                            if (yy < 0 || yy >= H || xq < 0 || xq >= W) continue;
                            const float* px = &x[(((size_t)n * H + yy) * W + xq) * C];
                            const float* pw = &w[(((size_t)co * C) * 3 + kh) * 3 + kw];
                            for (int ci = 0; ci < C; ++ci) { acc += (double)px[ci] * (double)pw[(size_t)ci * 9]; accf = std::fmaf(px[ci], pw[(size_t)ci * 9], accf); }
                        }
                    const size_t o = (((size_t)n * H + y) * W + xx) * K + co;
                    ref[o] = std::max(acc, 0.0);
                    err = std::max(err, std::fabs((double)hout[o] - ref[o])); err_f32 = std::max(err_f32, std::fabs((double)std::max(accf + b[co], 0.f) - ref[o])); ymax = std::max(ymax, ref[o]);
                }
    double perr = 0;
    if (MODE == 1)
        for (int n = 0; n < N; ++n)
            for (int y = 0; y < Hp; ++y)
                for (int xx = 0; xx < Wp; ++xx)
                    for (int co = 0; co < K; ++co) {
                        double m = 0;
                        for (int dy = 0; dy < 2; ++dy)
                            for (int dxx = 0; dxx < 2; ++dxx) m = std::max(m, ref[(((size_t)n * H + 2 * y + dy) * W + 2 * xx + dxx) * K + co]);
                        perr = std::max(perr, std::fabs((double)hpool[(((size_t)n * Hp + y) * Wp + xx) * K + co] - m));
                    }
    std::printf("numerics, MODE %d, 64 -> 64, 2 x %d x %d, fp32 in / fp32 out: max|y| %.2f   max|split - float64| %.3e   (one fp32 fma chain on the host: %.3e)%s", MODE, H, W, ymax, err, err_f32, MODE == 1 ? "" : "\n");
    if (MODE == 1) std::printf("   pooled tensor: %.3e\n", perr);
    CK(hipFree(din)); CK(hipFree(dw)); CK(hipFree(dout)); CK(hipFree(dpool)); CK(hipFree(db));
}

struct Layer { float *in, *out, *pool, *b; _Float16* w; };
template <int CIN, int COUT>
static Layer alloc_layer(int N, int H, int W) {
    const size_t in_n = (size_t)N * H * W * CIN, out_n = (size_t)N * H * W * COUT, w_n = (size_t)CIN * COUT * 27;
    Layer q;
    CK(hipMalloc(&q.in, in_n * 4)); CK(hipMalloc(&q.w, w_n * 2)); CK(hipMalloc(&q.out, out_n * 4)); CK(hipMalloc(&q.pool, out_n)); CK(hipMalloc(&q.b, COUT * 4));
    hipLaunchKernelGGL(k_fill_f32, dim3(4096), dim3(256), 0, 0, q.in, in_n, 1u, 2.0f);
    hipLaunchKernelGGL(k_fill_h, dim3(1024), dim3(256), 0, 0, q.w, w_n, 2u, 0.05f);
    CK(hipMemset(q.b, 0, COUT * 4));
    return q;
}
static void free_layer(Layer& q) { CK(hipFree(q.in)); CK(hipFree(q.w)); CK(hipFree(q.out)); CK(hipFree(q.pool)); CK(hipFree(q.b)); }

template <int CIN, int COUT, int MODE>
static void timing(const char* layer, int N, int H, int W, double wino_ms) {
    Layer q = alloc_layer<CIN, COUT>(N, H, W);
    for (int i = 0; i < 5; ++i) launch<CIN, COUT, MODE>(q.in, q.w, q.b, q.out, q.pool, N, H, W);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) launch<CIN, COUT, MODE>(q.in, q.w, q.b, q.out, q.pool, N, H, W);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    const double flops = 2.0 * CIN * COUT * 9 * (double)N * H * W;
    std::printf("%-13s %3d -> %3d  %3dx%-3d%s  split, fp32 in/out: %.4f ms = %6.1f TFLOP/s of fp32 work (%5.0f executed, %.2f of 2,500)   fp32 Winograd F(4x2) launch: %.3f ms   ratio %.2f\n",
                layer, CIN, COUT, H, W, MODE == 1 ? " +pool" : "      ", ms, flops / ms / 1e9, 3 * flops / ms / 1e9, 3 * flops / ms / 1e9 / 2500.0, wino_ms, ms / wino_ms);
    std::fflush(stdout);
    free_layer(q);
}

static void sequence(int N) {
    Layer a = alloc_layer<64, 64>(N, 128, 128), b = alloc_layer<64, 128>(N, 64, 64), c = alloc_layer<128, 128>(N, 64, 64), d = alloc_layer<128, 256>(N, 32, 32),
          e = alloc_layer<256, 256>(N, 32, 32), f = alloc_layer<256, 128>(N, 64, 64), g = alloc_layer<128, 128>(N, 64, 64), h = alloc_layer<128, 64>(N, 128, 128);
    auto round = [&]() {
        launch<64, 64, 1>(a.in, a.w, a.b, a.out, a.pool, N, 128, 128); launch<64, 128, 0>(b.in, b.w, b.b, b.out, b.pool, N, 64, 64);
        launch<128, 128, 1>(c.in, c.w, c.b, c.out, c.pool, N, 64, 64); launch<128, 256, 0>(d.in, d.w, d.b, d.out, d.pool, N, 32, 32);
        launch<256, 256, 0>(e.in, e.w, e.b, e.out, e.pool, N, 32, 32); launch<256, 128, 0>(f.in, f.w, f.b, f.out, f.pool, N, 64, 64);
        launch<128, 128, 0>(g.in, g.w, g.b, g.out, g.pool, N, 64, 64); launch<128, 64, 0>(h.in, h.w, h.b, h.out, h.pool, N, 128, 128);
    };
    for (int i = 0; i < 5; ++i) round();
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 30; ++i) round();
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 30;
    std::printf("the eight launches in forward order, 30 rounds back to back: %.3f ms per round (fp32 Winograd launches of the same layers in the forward: 8.347 ms)\n", ms);
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 256;
    numerics<0>(32, 32);
    numerics<0>(24, 40);          // ragged tiles: 24 = 3 x 8 rows, 40 = 32 + 8 columns
    numerics<1>(32, 64);
    if (N <= 0) return 0;
    timing<64, 64, 1>("down1.2", N, 128, 128, 1.0869);
    timing<64, 128, 0>("down2.0", N, 64, 64, 0.5069);
    timing<128, 128, 1>("down2.2", N, 64, 64, 0.9160);
    timing<128, 256, 0>("bottleneck.0", N, 32, 32, 0.4570);
    timing<256, 256, 0>("bottleneck.2", N, 32, 32, 0.8457);
    timing<256, 128, 0>("upconv2.0", N, 64, 64, 1.6963);
    timing<128, 128, 0>("upconv2.2", N, 64, 64, 0.9020);
    timing<128, 64, 0>("upconv1.0", N, 128, 128, 1.9353);
    sequence(N);
    return 0;
}
