// w42_bench.hip — timing experiments on k_wino42_conv as the product launches it (walking workgroups), one layer shape at a
// time (not part of the product).  Build: make -C celebrity_image_denoiser_amd/csrc tools     Run on the GPU box: tools/w42_bench [N]
//   * every shape of the forward at B = N: median ms, executed-MFMA fraction of the 157.3 TFLOP/s fp32 matrix peak
//   * with one pipeline component removed at a time (ABLATE: wrong results by design, see wino42_kernels.h)
//   * phase sums per workgroup from s_memtime stamps (ABLATE 256): main loop / epilogue / tile boundary, in shader cycles
// Weights and inputs are random: the kernels' timing does not depend on the values (the clock the chip holds does: random data).
#include "../wino42_kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

using namespace cid;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

struct Variant { std::string name; std::function<void(hipStream_t)> run; double exec_flops; int grid; };

static float* dalloc(size_t n, float scale) {
    std::vector<float> h(n);
    uint32_t s = 12345u + (uint32_t)n;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((int)(s >> 8) % 2001 - 1000) / 1000.0f; }
    float* d; CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return d;
}

static int g_wg_per_cu = 2;   // argv[2]: 0 = one item per workgroup

struct Bufs { float *in, *u, *bias, *out, *pool; unsigned* tab; };
template <int CIN, int COUT, int H>
static Bufs& bufs(int N, int W) {   // one set per layer shape, shared by its ablation variants
    static Bufs b{};
    if (!b.in) {
        b.in = dalloc((size_t)N * H * W * CIN, 1.f);
        b.u = dalloc((size_t)CIN * COUT * 24, 0.05f);
        b.bias = dalloc(COUT, 0.1f);
        b.out = dalloc((size_t)N * H * W * COUT, 0.f);
        b.pool = dalloc((size_t)N * (H / 2) * (W / 2) * COUT, 0.f);
        std::vector<unsigned> h(wino42_slot_table<8>(nullptr));
        wino42_slot_table<8>(h.data());
        CK(hipMalloc(&b.tab, h.size() * 4));
        CK(hipMemcpy(b.tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    return b;
}

template <int CIN, int COUT, bool POOL, int ABLATE, int H>
static Variant mk(const char* name, int N, int W, unsigned long long* trace = nullptr) {
    constexpr int TC = 8, TRW = 16 / TC, NB = COUT / WN2;
    const Bufs& b = bufs<CIN, COUT, H>(N, W);
    float *in = b.in, *u = b.u, *bias = b.bias, *out = b.out, *pool = b.pool;
    unsigned* tab = b.tab;
    WinoArgs a{};
    a.in = in; a.u = u; a.bias = bias; a.out = out; a.pool = pool; a.slot_tab = tab;
    a.zout = reinterpret_cast<float*>(trace);
    a.N = N; a.Hin = H; a.Win = W; a.in_ps = CIN; a.Hc = H; a.Wc = W; a.Hs = H; a.Ws = W; a.out_ps = COUT; a.out_coff = 0;
    a.tiles_x = (W + 4 * TC - 1) / (4 * TC); a.tiles_y = (H + 2 * TRW - 1) / (2 * TRW);
    a.tiles_total = N * a.tiles_x * a.tiles_y; a.tiles_per_xcd = (a.tiles_total + 7) / 8;
    a.rcp_x = tile_rcp(a.tiles_x); a.rcp_xy = tile_rcp(a.tiles_x * a.tiles_y);
    const int items = 8 * a.tiles_per_xcd * NB, walkers = g_wg_per_cu * 256 / 8;
    int grid = items;
    a.walk = 0;
    if (g_wg_per_cu > 0 && items > 8 * walkers) { a.walk = walkers; grid = 8 * walkers; }
    const double exec = 2.0 * CIN * COUT * 3 * (double)N * H * W;   // F(4x2,3x3): 3 multiplies per output pixel and (ci, co)
    return {name, [=](hipStream_t s) { hipLaunchKernelGGL((k_wino42_conv<CIN, COUT, POOL, TC, ABLATE>), dim3(grid), dim3(THREADS), 0, s, a); }, exec, grid};
}

template <int CIN, int COUT, bool POOL, int H>
static void shape(std::vector<Variant>& v, const char* tag, int N, bool ablations) {
    const int W = H;
    auto nm = [&](const char* x) { return std::string(tag) + " " + x; };
    v.push_back(mk<CIN, COUT, POOL, 0, H>(nm("base").c_str(), N, W));
    if (!ablations) return;
    v.push_back(mk<CIN, COUT, POOL, 1, H>(nm("no-dma").c_str(), N, W));
    v.push_back(mk<CIN, COUT, POOL, 2, H>(nm("no-B-loads").c_str(), N, W));
    v.push_back(mk<CIN, COUT, POOL, 4, H>(nm("no-V-build").c_str(), N, W));
    v.push_back(mk<CIN, COUT, POOL, 8, H>(nm("no-epilogue").c_str(), N, W));
    v.push_back(mk<CIN, COUT, POOL, 15, H>(nm("mfma-only").c_str(), N, W));
}

template <int CIN, int COUT, bool POOL, int H>
static void phases(const char* tag, int N, hipStream_t s) {
    const int W = H;
    const int maxwg = 8 * 4096 * 8;
    static unsigned long long* tr = nullptr;
    if (!tr) CK(hipMalloc(&tr, (size_t)maxwg * 8 * 8));
    Variant t = mk<CIN, COUT, POOL, 256, H>("trace", N, W, tr);
    if (t.grid > maxwg) { std::printf("%s: grid too large for the trace buffer\n", tag); return; }
    t.run(s); CK(hipStreamSynchronize(s));
    CK(hipMemset(tr, 0, (size_t)maxwg * 8 * 8));
    t.run(s); CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h((size_t)t.grid * 8);
    CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
    double mainl = 0, epi = 0, bnd = 0, life = 0, tiles = 0; int cnt = 0;
    std::vector<double> lives;
    for (int i = 0; i < t.grid; ++i) {
        const unsigned long long* r = &h[(size_t)i * 8];
        if (!r[0]) continue;
        mainl += r[1]; epi += r[2]; bnd += r[3]; life += r[4] - r[0]; tiles += r[5]; ++cnt;
        lives.push_back((double)(r[4] - r[0]));
    }
    std::sort(lives.begin(), lives.end());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s)); t.run(s); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double lsum = 0; for (double v : lives) lsum += v;
    std::printf("%-22s life deciles (k cycles):", tag);
    for (int d = 0; d <= 10; ++d) std::printf(" %.0f", lives[std::min(lives.size() - 1, lives.size() * d / 10)] / 1e3);
    std::printf("  mean %.0f\n", lsum / lives.size() / 1e3);
    constexpr int NU = CIN / 8;
    // (s_memtime counters of different XCDs are not aligned: only durations inside one workgroup are compared)
    std::printf("%-22s %5d workgroups x %.1f items: per item (cycles) main loop %.0f (MFMA issue %d)  epilogue %.0f  boundary %.0f | workgroup life min %.0f median %.0f "
                "p95 %.0f max %.0f cycles; launch %.4f ms\n",
                tag, cnt, tiles / cnt, mainl / tiles, NU * 48 * 32, epi / tiles, bnd / tiles, lives.front(), lives[lives.size() / 2], lives[lives.size() * 95 / 100],
                lives.back(), ms);
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 256;
    if (argc > 2) g_wg_per_cu = std::atoi(argv[2]);
    const bool abl = argc > 3 && std::string(argv[3]) == "ablate";
    const int ROUNDS = 7;
    hipStream_t s; CK(hipStreamCreate(&s));
    std::vector<Variant> v;
    shape<64, 64, true, 128>(v, "down1.2  64->64@128 pool", N, abl);
    shape<64, 128, false, 64>(v, "down2.0  64->128@64", N, abl);
    shape<128, 128, true, 64>(v, "down2.2  128->128@64 pool", N, false);
    shape<128, 256, false, 32>(v, "bott.0   128->256@32", N, false);
    shape<256, 256, false, 32>(v, "bott.2   256->256@32", N, abl);
    shape<256, 128, false, 64>(v, "upconv2.0 256->128@64", N, false);
    shape<128, 128, false, 64>(v, "upconv2.2 128->128@64", N, false);
    shape<128, 64, false, 128>(v, "upconv1.0 128->64@128", N, abl);
    std::vector<std::vector<float>> ms(v.size());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& x : v) x.run(s);   // warm-up
    CK(hipStreamSynchronize(s));
    for (int r = 0; r < ROUNDS; ++r)
        for (size_t i = 0; i < v.size(); ++i) {
            CK(hipEventRecord(e0, s)); v[i].run(s); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); ms[i].push_back(t);
        }
    CK(hipGetLastError());
    std::printf("B = %d, %d workgroup(s) per CU (0 = one item per workgroup)\n", N, g_wg_per_cu);
    for (size_t i = 0; i < v.size(); ++i) {
        std::sort(ms[i].begin(), ms[i].end());
        const float med = ms[i][ms[i].size() / 2];
        std::printf("%-40s grid %6d  median %8.4f ms  min %8.4f ms  executed %7.2f TFLOP/s = %.3f of 157.3\n", v[i].name.c_str(), v[i].grid, med, ms[i][0],
                    v[i].exec_flops / (med * 1e-3) / 1e12, v[i].exec_flops / (med * 1e-3) / 1e12 / 157.3);
    }
    phases<64, 64, false, 128>("64->64@128 (no pool)", N, s);
    phases<64, 128, false, 64>("64->128@64", N, s);
    phases<256, 256, false, 32>("256->256@32", N, s);
    phases<128, 64, false, 128>("128->64@128", N, s);
    return 0;
}
