// Experiment (round 4): the fp32 3x3 layers as SPLIT-OPERAND convolutions on the fp16 MFMA — prototype of an idea, not a product path.
//   every fp32 operand x = hi + lo, hi = half(x), lo = half(x - hi); x*w ~ hi_x*hi_w + hi_x*lo_w + lo_x*hi_w (lo*lo = 2^-22 relative, dropped);
//   that IS an fp16 convolution over 3 CIN input channels [hi_x | hi_x | lo_x] against weights [hi_w | lo_w | hi_w], fp32 accumulation.
// The product's k_conv3x3_h16 runs it unchanged with CIN' = 3 CIN (a real kernel would read hi_x once: 2 CIN channels of input instead of 3,
// the same bytes as fp32 activations); built with -DH16_SPLIT_OUT its epilogue stores the fp32 result as two halfs (hi | lo).
//   1. numerics on hardware: one layer, N = 2, 32 x 32, 64 -> 64, against a float64 convolution on the host, with and without the
//      power-of-two weight scale that keeps lo_w a normal half (does the MFMA honour fp16 denormals?);
//   2. time: the eight 3x3 layer shapes of the network at B = 256, beside the fp32 Winograd launches of profiles/r04_final_bench.json.
//   hipcc -O3 -std=c++17 -fno-slp-vectorize -DCID_EXPERIMENTS -DH16_SPLIT_OUT --offload-arch=gfx950 -o tools/split_proto tools/split_proto.hip
// Second form, -DH16_SPLIT_IN as well (tools/split_proto2): the kernel's own split sequencing (conv_kernels_f16.h) — a pixel holds [hi_x | lo_x] (2 CIN halfs = the
// bytes of the fp32 tensor), the weights [hi_w | lo_w]; a hi chunk's fragments serve six sub-steps (hi_w, then lo_w), a lo chunk's three (hi_w).
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
#ifdef H16_SPLIT_IN
constexpr int CMUL = 2;      // half channels per fp32 channel in the input tensor and in the packed weights
#else
constexpr int CMUL = 3;
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// the product's weight layout for k_conv3x3_h16 (cid_api.hip packed_index_h16), restated for a layer of `cin` input channels
static size_t packed_index(int cin, int co, int ci, int kh, int kw) {
    const int nb = co >> 6, cg = co & 3, c = (co >> 2) & 15;
    const int ck = ci >> 5, kg = (ci >> 3) & 3, e = ci & 7;
    const int nchunk = cin / 32;
    return ((((((size_t)(nb * nchunk + ck) * 3 + kw) * 3 + kh) * 4 + cg) * 64) + kg * 16 + c) * 8 + e;
}

template <int CIN3, int COUT>
static void launch(const _Float16* in, const _Float16* w, const float* bias, _Float16* out, int N, int H, int W) {
    GemmConvArgsH a{};
    a.in = in; a.w = w; a.bias = bias; a.out = out; a.pool = nullptr;
    a.N = N; a.Hin = H; a.Win = W; a.in_ps = CIN3; a.Hc = H; a.Wc = W; a.Hs = H; a.Ws = W; a.out_ps = 2 * COUT; a.out_coff = 0;
    a.tiles_x = (W + TILE_W - 1) / TILE_W; a.tiles_y = (H + TILE_H - 1) / TILE_H; a.tiles_total = N * a.tiles_x * a.tiles_y;
    a.tiles_per_xcd = (a.tiles_total + 7) / 8;
    a.rcp_x = tile_rcp(a.tiles_x); a.rcp_xy = tile_rcp(a.tiles_x * a.tiles_y);
    a.walk = 0;
    constexpr int NB = COUT / NTILE;
    hipLaunchKernelGGL((k_conv3x3_h16<CIN3, COUT, 0, false, false>), dim3(8 * a.tiles_per_xcd * NB), dim3(THREADS), 0, 0, a);
}

__global__ void k_fill(_Float16* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (_Float16)(((int)(x & 0xffff) - 32768) * (scale / 32768.f));
    }
}

static void numerics(float wscale_log2) {
    constexpr int N = 2, H = 32, W = 32, C = 64, K = 64;
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<float> x((size_t)N * H * W * C), w((size_t)K * C * 9), b(K);
    for (auto& v : x) { const float r = u(rng); v = r < -0.3f ? 0.f : (r < 0.f ? (r + 0.3f) * -3e-3f : 2.f * r); }   // post-ReLU-like: zeros, tiny values, values up to 2
    for (auto& v : w) v = 0.06f * u(rng);
    for (auto& v : b) v = 0.1f * u(rng);
    const float s = std::ldexp(1.f, (int)wscale_log2);
    std::vector<_Float16> hin((size_t)N * H * W * CMUL * C), hw((size_t)K * CMUL * C * 9);
    for (size_t p = 0; p < (size_t)N * H * W; ++p)
        for (int c = 0; c < C; ++c) {
            const float v = x[p * C + c];
            const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
            if (CMUL == 3) { hin[p * 3 * C + c] = hi; hin[p * 3 * C + C + c] = hi; hin[p * 3 * C + 2 * C + c] = lo; }
            else { hin[p * 2 * C + c] = hi; hin[p * 2 * C + C + c] = lo; }
        }
    int lo_subnormal = 0;
    for (int co = 0; co < K; ++co)
        for (int ci = 0; ci < C; ++ci)
            for (int kh = 0; kh < 3; ++kh)
                for (int kw = 0; kw < 3; ++kw) {
                    const float v = w[(((size_t)co * C + ci) * 3 + kh) * 3 + kw] * s;
                    const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                    if (lo != (_Float16)0.f && std::fabs((float)lo) < 6.104e-5f) ++lo_subnormal;
                    hw[packed_index(CMUL * C, co, ci, kh, kw)] = hi; hw[packed_index(CMUL * C, co, C + ci, kh, kw)] = lo;
                    if (CMUL == 3) hw[packed_index(3 * C, co, 2 * C + ci, kh, kw)] = hi;
                }
    std::vector<float> bs(K);
    for (int k = 0; k < K; ++k) bs[k] = b[k] * s;
    _Float16 *din, *dw, *dout; float* db;
    CK(hipMalloc(&din, hin.size() * 2)); CK(hipMalloc(&dw, hw.size() * 2)); CK(hipMalloc(&dout, (size_t)N * H * W * 2 * K * 2)); CK(hipMalloc(&db, K * 4));
    CK(hipMemcpy(din, hin.data(), hin.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, bs.data(), K * 4, hipMemcpyHostToDevice));
    launch<CMUL * C, K>(din, dw, db, dout, N, H, W);
    CK(hipDeviceSynchronize());
    std::vector<_Float16> hout((size_t)N * H * W * 2 * K);
    CK(hipMemcpy(hout.data(), dout, hout.size() * 2, hipMemcpyDeviceToHost));
    double err_split = 0, err_f32 = 0, err_hi = 0, ymax = 0;
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
                    const double ref = std::max(acc, 0.0);
                    const float f32 = std::max(accf + b[co], 0.f);
                    const size_t o = ((((size_t)n * H + y) * W + xx) * 2 * K);
                    const double got = ((double)(float)hout[o + co] + (double)(float)hout[o + K + co]) / s;
                    err_split = std::max(err_split, std::fabs(got - ref)); err_f32 = std::max(err_f32, std::fabs((double)f32 - ref));
                    err_hi = std::max(err_hi, std::fabs((double)(float)hout[o + co] / s - ref)); ymax = std::max(ymax, ref);
                }
    std::printf("numerics, 64 -> 64, 2 x 32 x 32, weights x 2^%d (%d of %d low weight pieces are subnormal halfs): max|y| %.2f   max|split - float64| %.3e   (one fp32 fma chain on the host: %.3e; the hi half alone: %.3e)\n",
                (int)wscale_log2, lo_subnormal, K * C * 9, ymax, err_split, err_f32, err_hi);
    CK(hipFree(din)); CK(hipFree(dw)); CK(hipFree(dout)); CK(hipFree(db));
}

template <int CIN, int COUT>
static void timing(const char* layer, int N, int H, int W, double wino_ms) {
    constexpr int C3 = CMUL * CIN;
    const size_t in_n = (size_t)N * H * W * C3, out_n = (size_t)N * H * W * 2 * COUT, w_n = (size_t)C3 * COUT * 9;
    _Float16 *din, *dw, *dout; float* db;
    CK(hipMalloc(&din, in_n * 2)); CK(hipMalloc(&dw, w_n * 2)); CK(hipMalloc(&dout, out_n * 2)); CK(hipMalloc(&db, COUT * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, din, in_n, 1u, 1.0f);
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, dw, w_n, 2u, 0.05f);
    CK(hipMemset(db, 0, COUT * 4));
    for (int i = 0; i < 5; ++i) launch<C3, COUT>(din, dw, db, dout, N, H, W);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) launch<C3, COUT>(din, dw, db, dout, N, H, W);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    const double flops = 2.0 * CIN * COUT * 9 * (double)N * H * W;
    std::printf("%-13s %3d -> %3d  %3dx%-3d  split (%d x CIN half channels): %.4f ms = %6.1f TFLOP/s of fp32 work (%5.0f executed, %.2f of 2,500)   fp32 Winograd F(4x2) launch: %.3f ms   ratio %.2f\n",
                layer, CIN, COUT, H, W, CMUL, ms, flops / ms / 1e9, 3 * flops / ms / 1e9, 3 * flops / ms / 1e9 / 2500.0, wino_ms, ms / wino_ms);
    std::fflush(stdout);
    CK(hipFree(din)); CK(hipFree(dw)); CK(hipFree(dout)); CK(hipFree(db));
}

// the eight launches back to back, as the forward issues them (the other four launches of the forward are not part of this tool): does the sum of the
// single-shape loops above hold when the shapes alternate under one power budget?
struct Seq { _Float16 *in, *w, *out; float* b; };
template <int CIN, int COUT>
static Seq seq_alloc(int N, int H, int W) {
    constexpr int C3 = CMUL * CIN;
    const size_t in_n = (size_t)N * H * W * C3, out_n = (size_t)N * H * W * 2 * COUT, w_n = (size_t)C3 * COUT * 9;
    Seq q;
    CK(hipMalloc(&q.in, in_n * 2)); CK(hipMalloc(&q.w, w_n * 2)); CK(hipMalloc(&q.out, out_n * 2)); CK(hipMalloc(&q.b, COUT * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, q.in, in_n, 1u, 1.0f);
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, q.w, w_n, 2u, 0.05f);
    CK(hipMemset(q.b, 0, COUT * 4));
    return q;
}
static void sequence(int N) {
    Seq a = seq_alloc<64, 64>(N, 128, 128), b = seq_alloc<64, 128>(N, 64, 64), c = seq_alloc<128, 128>(N, 64, 64), d = seq_alloc<128, 256>(N, 32, 32),
        e = seq_alloc<256, 256>(N, 32, 32), f = seq_alloc<256, 128>(N, 64, 64), g = seq_alloc<128, 128>(N, 64, 64), h = seq_alloc<128, 64>(N, 128, 128);
    auto round = [&]() {
        launch<CMUL * 64, 64>(a.in, a.w, a.b, a.out, N, 128, 128); launch<CMUL * 64, 128>(b.in, b.w, b.b, b.out, N, 64, 64);
        launch<CMUL * 128, 128>(c.in, c.w, c.b, c.out, N, 64, 64); launch<CMUL * 128, 256>(d.in, d.w, d.b, d.out, N, 32, 32);
        launch<CMUL * 256, 256>(e.in, e.w, e.b, e.out, N, 32, 32); launch<CMUL * 256, 128>(f.in, f.w, f.b, f.out, N, 64, 64);
        launch<CMUL * 128, 128>(g.in, g.w, g.b, g.out, N, 64, 64); launch<CMUL * 128, 64>(h.in, h.w, h.b, h.out, N, 128, 128);
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
    numerics(0);
    numerics(4);
    numerics(8);
    // fp32 launch times: profiles/r04_final_bench.json (B = 256, the box that measured 26,239 images/s)
    timing<64, 64>("down1.2", N, 128, 128, 1.0869);
    timing<64, 128>("down2.0", N, 64, 64, 0.5069);
    timing<128, 128>("down2.2", N, 64, 64, 0.9160);
    timing<128, 256>("bottleneck.0", N, 32, 32, 0.4570);
    timing<256, 256>("bottleneck.2", N, 32, 32, 0.8457);
    timing<256, 128>("upconv2.0", N, 64, 64, 1.6963);
    timing<128, 128>("upconv2.2", N, 64, 64, 0.9020);
    timing<128, 64>("upconv1.0", N, 128, 128, 1.9353);
    sequence(N);
    return 0;
}
