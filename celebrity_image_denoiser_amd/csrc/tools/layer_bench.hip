// layer_bench.hip — timing experiments on single layers of the forward (not part of the product).
// Build: make -C celebrity_image_denoiser_amd/csrc tools     Run on the GPU box: ./layer_bench
// Each variant is run ROUNDS times, interleaved with the others in one process; prints median ms
// and algorithmic TFLOP/s.  ABLATE variants compute wrong results by design (see conv_kernels.h).
#include "../conv_kernels.h"
#include "../wino64_kernels.h"
#include "../wino42_kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

using namespace cid;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

struct Variant { std::string name; std::function<void(hipStream_t)> run; double flops; };

static float* dalloc(size_t n, float scale) {
    std::vector<float> h(n);
    uint32_t s = 12345u + (uint32_t)n;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((int)(s >> 8) % 2001 - 1000) / 1000.0f; }
    float* d; CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return d;
}

template <int CIN, int COUT, int MODE, int ABLATE, int WPS>
static Variant make(const char* name, int N, int H, int W, float* in, float* w, float* bias, float* out, float* pool) {
    GemmConvArgs a{};
    a.in = in; a.w = w; a.bias = bias; a.out = out; a.pool = pool;
    a.N = N; a.Hin = H; a.Win = W; a.in_ps = CIN; a.Hc = H; a.Wc = W; a.Hs = H; a.Ws = W; a.out_ps = COUT; a.out_coff = 0;
    a.tiles_x = (W + TILE_W - 1) / TILE_W; a.tiles_y = (H + TILE_H - 1) / TILE_H;
    a.tiles_total = N * a.tiles_x * a.tiles_y; a.tiles_per_xcd = (a.tiles_total + 7) / 8;
    a.rcp_x = tile_rcp(a.tiles_x); a.rcp_xy = tile_rcp(a.tiles_x * a.tiles_y);
    constexpr int NB = (MODE == 2 ? 4 * COUT : COUT) / NTILE;
    const int grid = 8 * a.tiles_per_xcd * NB;
    const double flops = 2.0 * CIN * COUT * (MODE == 2 ? 4 : 9) * (double)N * H * W;
    return {name, [=](hipStream_t s) { hipLaunchKernelGGL((k_gemm_conv<CIN, COUT, MODE, ABLATE, WPS>), dim3(grid), dim3(THREADS), 0, s, a); }, flops};
}

template <int CIN, int COUT, bool POOL, int TC, int ABLATE>
static Variant makew64(const char* name, int N, int H, int W, float* in, float* u, float* bias, float* out, float* pool) {
    WinoArgs a{};
    a.in = in; a.u = u; a.bias = bias; a.out = out; a.pool = pool;
    static unsigned* tabs[2] = {nullptr, nullptr};
    const int ti = TC == 32 ? 0 : 1;
    if (!tabs[ti]) {
        std::vector<unsigned> h(wino_slot_table(TC, 32 / TC, nullptr));
        wino_slot_table(TC, 32 / TC, h.data());
        CK(hipMalloc(&tabs[ti], h.size() * 4));
        CK(hipMemcpy(tabs[ti], h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    a.slot_tab = tabs[ti];
    a.N = N; a.Hin = H; a.Win = W; a.in_ps = CIN; a.Hc = H; a.Wc = W; a.Hs = H; a.Ws = W; a.out_ps = COUT; a.out_coff = 0;
    constexpr int TRW = 32 / TC;
    a.tiles_x = (W + 2 * TC - 1) / (2 * TC); a.tiles_y = (H + 2 * TRW - 1) / (2 * TRW);
    a.tiles_total = N * a.tiles_x * a.tiles_y; a.tiles_per_xcd = (a.tiles_total + 7) / 8;
    a.rcp_x = tile_rcp(a.tiles_x); a.rcp_xy = tile_rcp(a.tiles_x * a.tiles_y);
    const int grid = 8 * a.tiles_per_xcd * (COUT / WN2);
    const double flops = 2.0 * CIN * COUT * 9 * (double)N * H * W;   // algorithmic (direct) FLOPs
    return {name, [=](hipStream_t s) { hipLaunchKernelGGL((k_wino64_conv<CIN, COUT, POOL, TC, ABLATE>), dim3(grid), dim3(THREADS), 0, s, a); }, flops};
}

template <int CIN, int COUT, bool POOL, int TC, int ABLATE>
static Variant makew42(const char* name, int N, int H, int W, float* in, float* u, float* bias, float* out, float* pool) {
    WinoArgs a{};
    a.in = in; a.u = u; a.bias = bias; a.out = out; a.pool = pool;
    std::vector<unsigned> h(wino42_slot_table<TC>(nullptr));
    wino42_slot_table<TC>(h.data());
    unsigned* tab; CK(hipMalloc(&tab, h.size() * 4));
    CK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    a.slot_tab = tab;
    a.N = N; a.Hin = H; a.Win = W; a.in_ps = CIN; a.Hc = H; a.Wc = W; a.Hs = H; a.Ws = W; a.out_ps = COUT; a.out_coff = 0;
    constexpr int TRW = 16 / TC;
    a.tiles_x = (W + 4 * TC - 1) / (4 * TC); a.tiles_y = (H + 2 * TRW - 1) / (2 * TRW);
    a.tiles_total = N * a.tiles_x * a.tiles_y; a.tiles_per_xcd = (a.tiles_total + 7) / 8;
    a.rcp_x = tile_rcp(a.tiles_x); a.rcp_xy = tile_rcp(a.tiles_x * a.tiles_y);
    const int grid = 8 * a.tiles_per_xcd * (COUT / WN2);
    const double flops = 2.0 * CIN * COUT * 9 * (double)N * H * W;   // algorithmic (direct) FLOPs
    return {name, [=](hipStream_t s) { hipLaunchKernelGGL((k_wino42_conv<CIN, COUT, POOL, TC, ABLATE>), dim3(grid), dim3(THREADS), 0, s, a); }, flops};
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 256;
    const int ROUNDS = 7;
    hipStream_t s; CK(hipStreamCreate(&s));
    // upconv1.0 shape: 128 -> 64 @ 128x128 ; bottleneck.2 shape: 256 -> 256 @ 32x32
    float* inA = dalloc((size_t)N * 128 * 128 * 128, 1.f);
    float* wA = dalloc((size_t)128 * 64 * 9, 0.05f);
    float* bA = dalloc(64, 0.1f);
    float* outA = dalloc((size_t)N * 128 * 128 * 64, 0.f);
    float* poolA = dalloc((size_t)N * 64 * 64 * 64, 0.f);
    float* inB = dalloc((size_t)N * 32 * 32 * 256, 1.f);
    float* wB = dalloc((size_t)256 * 256 * 9, 0.05f);
    float* bB = dalloc(256, 0.1f);
    float* outB = dalloc((size_t)N * 32 * 32 * 256, 0.f);
    std::vector<Variant> v;
    float* uA = dalloc((size_t)128 * 64 * 16, 0.05f);
    float* uB = dalloc((size_t)256 * 256 * 16, 0.05f);
    float* u42A = dalloc((size_t)128 * 64 * 24, 0.05f);
    float* u42B = dalloc((size_t)256 * 256 * 24, 0.05f);
    v.push_back(make<128, 64, 0, 0, 2>("A direct 128->64@128", N, 128, 128, inA, wA, bA, outA, poolA));
    v.push_back(makew64<128, 64, false, 32, 0>("A wino64 base", N, 128, 128, inA, uA, bA, outA, poolA));
    v.push_back(makew64<128, 64, false, 32, 1>("A wino64 no-dma", N, 128, 128, inA, uA, bA, outA, poolA));
    v.push_back(makew64<128, 64, false, 32, 2>("A wino64 no-B-loads", N, 128, 128, inA, uA, bA, outA, poolA));
    v.push_back(makew64<128, 64, false, 32, 4>("A wino64 no-A-build", N, 128, 128, inA, uA, bA, outA, poolA));
    v.push_back(makew64<128, 64, false, 32, 16>("A wino64 no-global-stores", N, 128, 128, inA, uA, bA, outA, poolA));
    v.push_back(makew64<128, 64, false, 32, 8>("A wino64 no-epilogue", N, 128, 128, inA, uA, bA, outA, poolA));
    v.push_back(makew64<128, 64, false, 32, 15>("A wino64 mfma-only", N, 128, 128, inA, uA, bA, outA, poolA));
    v.push_back(makew64<128, 64, false, 32, 47>("A wino64 mfma-only no-barrier", N, 128, 128, inA, uA, bA, outA, poolA));
    v.push_back(makew64<128, 64, false, 32, 33>("A wino64 no-dma no-barrier", N, 128, 128, inA, uA, bA, outA, poolA));
    v.push_back(makew64<128, 64, false, 32, 47 + 64>("A wino64 mfma-only no-prologue-dma", N, 128, 128, inA, uA, bA, outA, poolA));
    v.push_back(makew64<128, 64, false, 32, 128>("A wino64 no-dephase", N, 128, 128, inA, uA, bA, outA, poolA));
    v.push_back(makew42<128, 64, false, 8, 0>("A wino42 base", N, 128, 128, inA, u42A, bA, outA, poolA));
    v.push_back(makew42<128, 64, false, 8, 1>("A wino42 no-dma", N, 128, 128, inA, u42A, bA, outA, poolA));
    v.push_back(makew42<128, 64, false, 8, 2>("A wino42 no-B-loads", N, 128, 128, inA, u42A, bA, outA, poolA));
    v.push_back(makew42<128, 64, false, 8, 4>("A wino42 no-V-build", N, 128, 128, inA, u42A, bA, outA, poolA));
    v.push_back(makew42<128, 64, false, 8, 8>("A wino42 no-epilogue", N, 128, 128, inA, u42A, bA, outA, poolA));
    v.push_back(makew42<128, 64, false, 8, 15>("A wino42 mfma-only", N, 128, 128, inA, u42A, bA, outA, poolA));
    v.push_back(makew42<256, 256, false, 8, 0>("B wino42 base", N, 32, 32, inB, u42B, bB, outB, nullptr));
    v.push_back(makew42<256, 256, false, 8, 4>("B wino42 no-V-build", N, 32, 32, inB, u42B, bB, outB, nullptr));
    v.push_back(makew42<256, 256, false, 8, 15>("B wino42 mfma-only", N, 32, 32, inB, u42B, bB, outB, nullptr));
    v.push_back(make<256, 256, 0, 0, 2>("B direct 256->256@32", N, 32, 32, inB, wB, bB, outB, nullptr));
    v.push_back(makew64<256, 256, false, 16, 0>("B wino64 base", N, 32, 32, inB, uB, bB, outB, nullptr));
    v.push_back(makew64<256, 256, false, 16, 15>("B wino64 mfma-only", N, 32, 32, inB, uB, bB, outB, nullptr));
    {   // up1 shape: ConvT 128 -> 64 on 64x64 inputs (output 128x128x64); reuse inA (>= N*64*64*128) and outA
        v.push_back(make<128, 64, 2, 0, 2>("T convT 128->64@64 base", N, 64, 64, inA, wA, bA, outA, nullptr));
        v.push_back(make<128, 64, 2, 1, 2>("T no-halo-prefetch", N, 64, 64, inA, wA, bA, outA, nullptr));
        v.push_back(make<128, 64, 2, 2, 2>("T no-B-loads", N, 64, 64, inA, wA, bA, outA, nullptr));
        v.push_back(make<128, 64, 2, 8, 2>("T no-stores", N, 64, 64, inA, wA, bA, outA, nullptr));
        v.push_back(make<128, 64, 2, 15, 2>("T mfma-only", N, 64, 64, inA, wA, bA, outA, nullptr));
    }
    if (argc > 2 && std::string(argv[2]) == "trace") {
        // One launch of the transposed conv (up1 shape) with s_memtime stamps per workgroup: where does a workgroup's time go,
        // and what happens on a CU between one workgroup's stores and the next one's first MFMA?
        const int H = 64, W = 64;
        const int tiles = N * ((W + TILE_W - 1) / TILE_W) * ((H + TILE_H - 1) / TILE_H);
        const int nwg = 8 * ((tiles + 7) / 8) * 4;
        unsigned long long* tr; CK(hipMalloc(&tr, (size_t)nwg * 8 * 8)); CK(hipMemset(tr, 0, (size_t)nwg * 8 * 8));
        Variant t = make<128, 64, 2, 16, 2>("trace", N, H, W, inA, wA, bA, outA, reinterpret_cast<float*>(tr));
        t.run(s); CK(hipStreamSynchronize(s));    // warm
        CK(hipMemset(tr, 0, (size_t)nwg * 8 * 8));
        t.run(s); CK(hipStreamSynchronize(s));
        std::vector<unsigned long long> h((size_t)nwg * 8);
        CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
        std::FILE* f = std::fopen("gpurun_out/convt_trace.csv", "w");
        std::fprintf(f, "wg,t_start,t_main,t_main_end,t_stores_issued,t_stores_done,hw_id,xcc_id\n");
        for (int i = 0; i < nwg; ++i)
            if (h[(size_t)i * 8])
                std::fprintf(f, "%d,%llu,%llu,%llu,%llu,%llu,%llu,%llu\n", i, h[(size_t)i * 8], h[(size_t)i * 8 + 1], h[(size_t)i * 8 + 2],
                             h[(size_t)i * 8 + 3], h[(size_t)i * 8 + 6], h[(size_t)i * 8 + 4], h[(size_t)i * 8 + 5]);
        std::fclose(f);
        std::printf("trace written: %d workgroups\n", nwg);
        // (the F(4x2) kernel's phase trace lives in tools/w42_bench)
        {   // the same for the dominant Winograd launch (upconv1.0 shape)
            const int nwg2 = 8 * ((N * 2 * 64 + 7) / 8);
            unsigned long long* tr2; CK(hipMalloc(&tr2, (size_t)nwg2 * 16 * 8)); CK(hipMemset(tr2, 0, (size_t)nwg2 * 16 * 8));
            Variant w = makew64<128, 64, false, 32, 256>("trace64", N, 128, 128, inA, uA, bA, outA, reinterpret_cast<float*>(tr2));
            w.run(s); CK(hipStreamSynchronize(s));
            CK(hipMemset(tr2, 0, (size_t)nwg2 * 16 * 8));
            w.run(s); CK(hipStreamSynchronize(s));
            std::vector<unsigned long long> h2((size_t)nwg2 * 16);
            CK(hipMemcpy(h2.data(), tr2, h2.size() * 8, hipMemcpyDeviceToHost));
            std::FILE* f2 = std::fopen("gpurun_out/wino64_trace.csv", "w");
            std::fprintf(f2, "wg,t_start,t_main,t_main_end,t_stores_issued,t_stores_done,hw_id,xcc_id,t_table,t_dma_issued,t_dma_landed,t_dma_all,t_ep_b1,t_ep_b2,t_ep_b3,t_dma0,t_dma1\n");
            for (int i = 0; i < nwg2; ++i) {
                const unsigned long long* r = h2.data() + (size_t)i * 16;
                if (r[0])
                    std::fprintf(f2, "%d,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu\n", i, r[0], r[1], r[2], r[3], r[6], r[4], r[5],
                                 r[8], r[9], r[10], r[11], r[12], r[13], r[14], r[15], r[7]);
            }
            std::fclose(f2);
            std::printf("wino64 trace written: %d workgroups\n", nwg2);
            // main-loop duration (stamps 1 -> 2) with one component removed at a time
            auto main_median = [&](Variant v2, const char* nm) {
                (void)hipMemset(tr2, 0, (size_t)nwg2 * 16 * 8);
                v2.run(s); (void)hipStreamSynchronize(s);
                (void)hipMemset(tr2, 0, (size_t)nwg2 * 16 * 8);
                v2.run(s); (void)hipStreamSynchronize(s);
                (void)hipMemcpy(h2.data(), tr2, h2.size() * 8, hipMemcpyDeviceToHost);
                std::vector<unsigned long long> d, tot;
                for (int i = 0; i < nwg2; ++i) { const unsigned long long* r = h2.data() + (size_t)i * 16; if (r[0] && r[2] > r[1]) { d.push_back(r[2] - r[1]); tot.push_back((r[3] ? r[3] : r[2]) - r[0]); } }
                std::sort(d.begin(), d.end()); std::sort(tot.begin(), tot.end());
                std::printf("%-34s main loop median %6llu cycles, workgroup (start -> stores issued) %6llu\n", nm, d[d.size() / 2], tot[tot.size() / 2]);
            };
            float* trf = reinterpret_cast<float*>(tr2);
            main_median(makew64<128, 64, false, 32, 256>("", N, 128, 128, inA, uA, bA, outA, trf), "wino64 base");
            main_median(makew64<128, 64, false, 32, 256 + 1>("", N, 128, 128, inA, uA, bA, outA, trf), "wino64 no DMA after prologue");
            main_median(makew64<128, 64, false, 32, 256 + 2>("", N, 128, 128, inA, uA, bA, outA, trf), "wino64 B loaded once");
            main_median(makew64<128, 64, false, 32, 256 + 4>("", N, 128, 128, inA, uA, bA, outA, trf), "wino64 A built once");
            main_median(makew64<128, 64, false, 32, 256 + 32>("", N, 128, 128, inA, uA, bA, outA, trf), "wino64 no per-chunk barrier");
            main_median(makew64<128, 64, false, 32, 256 + 7>("", N, 128, 128, inA, uA, bA, outA, trf), "wino64 MFMA + barriers only");
            main_median(makew64<128, 64, false, 32, 256 + 39>("", N, 128, 128, inA, uA, bA, outA, trf), "wino64 MFMA only");
        }
        return 0;
    }
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
    for (size_t i = 0; i < v.size(); ++i) {
        std::sort(ms[i].begin(), ms[i].end());
        const float med = ms[i][ms[i].size() / 2];
        std::printf("%-28s median %8.4f ms  min %8.4f ms  %7.2f TFLOP/s  (%.1f%% of 157.3 algorithmic)\n", v[i].name.c_str(), med, ms[i][0],
                    v[i].flops / (med * 1e-3) / 1e12, 100.0 * v[i].flops / (med * 1e-3) / 1e12 / 157.3);
    }
    return 0;
}
