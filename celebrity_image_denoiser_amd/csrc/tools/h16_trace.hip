// Experiment: per-workgroup timeline of k_conv3x3_h16 (s_memtime stamps at the phase boundaries).
//   hipcc -O3 -std=c++17 -fno-slp-vectorize -DCID_EXPERIMENTS -DH16_TRACE --offload-arch=gfx950 -o tools/h16_trace tools/h16_trace.hip
//   tools/h16_trace [N=512]   -> gpurun_out/h16_trace_<cin>_<cout>.csv + summary on stdout
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include "../conv_kernels.h"
#include "../conv_kernels_f16.h"
using namespace cid;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int CIN, int COUT>
void run(int N, int H, int W, bool zero, bool walk_ok) {
    const size_t in_n = (size_t)N * H * W * CIN, out_n = (size_t)N * H * W * COUT, w_n = (size_t)CIN * COUT * 9;
    std::vector<_Float16> hin(in_n), hw(w_n);
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    for (auto& v : hin) v = zero ? (_Float16)0.f : (_Float16)u(rng);
    for (auto& v : hw) v = zero ? (_Float16)0.f : (_Float16)(u(rng) * 0.05f);
    _Float16 *din, *dw, *dout; float* dbias; unsigned long long* dtr;
    CK(hipMalloc(&din, in_n * 2)); CK(hipMalloc(&dw, w_n * 2)); CK(hipMalloc(&dout, out_n * 2)); CK(hipMalloc(&dbias, COUT * 4));
    CK(hipMemcpy(din, hin.data(), in_n * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), w_n * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dbias, 0, COUT * 4));
    GemmConvArgsH a{};
    a.in = din; a.w = dw; a.bias = dbias; a.out = dout;
    a.N = N; a.Hin = H; a.Win = W; a.in_ps = CIN; a.Hc = H; a.Wc = W; a.Hs = H; a.Ws = W; a.out_ps = COUT; a.out_coff = 0;
    a.tiles_x = (W + TILE_W - 1) / TILE_W; a.tiles_y = (H + TILE_H - 1) / TILE_H; a.tiles_total = N * a.tiles_x * a.tiles_y;
    a.tiles_per_xcd = (a.tiles_total + 7) / 8;
    a.rcp_x = tile_rcp(a.tiles_x); a.rcp_xy = tile_rcp(a.tiles_x * a.tiles_y);
    constexpr int NB = COUT / NTILE;
    int grid = 8 * a.tiles_per_xcd * NB;
    {   // the product's launch shape (cid_api.hip launch_gemm_h): three walking workgroups per CU on the layers with CIN <= 128
        int dev = 0, cus = 256;
        CK(hipGetDevice(&dev)); CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const int walkers = 3 * cus / 8;
        a.walk = 0;
        if (CIN <= 128 && walk_ok && grid > 8 * walkers && a.tiles_per_xcd >= walkers) { a.walk = walkers; grid = 8 * walkers; }
    }
    CK(hipMalloc(&dtr, (size_t)grid * 64)); CK(hipMemset(dtr, 0, (size_t)grid * 64));
    a.pool = reinterpret_cast<_Float16*>(dtr);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_conv3x3_h16<CIN, COUT, 0, false, true>), dim3(grid), dim3(THREADS), 0, 0, a);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_conv3x3_h16<CIN, COUT, 0, false, true>), dim3(grid), dim3(THREADS), 0, 0, a);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    std::vector<unsigned long long> tr((size_t)grid * 8);
    CK(hipMemcpy(tr.data(), dtr, (size_t)grid * 64, hipMemcpyDeviceToHost));
    // per ITEM (a workgroup's sums over its items / its item count): boundary (= prologue for a workgroup's first item), main loop, epilogue
    std::vector<double> pro, mainl, epi, drain, life;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < grid; ++b) {
        const unsigned long long* t = &tr[(size_t)b * 8];
        if (!t[0] || !t[5]) continue;
        const double it = (double)t[5];
        pro.push_back((double)t[3] / it); mainl.push_back((double)t[1] / it); epi.push_back((double)t[2] / it);
        drain.push_back(it); life.push_back((double)(t[4] - t[0]) / it);
        tmin = std::min(tmin, t[0]); tmax = std::max(tmax, t[4]);
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    auto mean = [](const std::vector<double>& v) { double s = 0; for (double x : v) s += x; return v.empty() ? 0.0 : s / v.size(); };
    const double flops = 2.0 * CIN * COUT * 9 * (double)N * H * W;
    std::printf("%sk_conv3x3_h16<%d,%d,0> N=%d %dx%d: %.4f ms, %.0f TFLOP/s, %zu workgroups traced\n", zero ? "[zeros] " : "", CIN, COUT, N, H, W, ms, flops / ms / 1e9, life.size());
    std::printf("  s_memtime ticks (100 MHz reference? see below): kernel span %llu\n", tmax - tmin);
    std::printf("  phase            mean      median\n");
    std::printf("  boundary      %8.0f   %8.0f   (walk = %d walkers per XCD group)\n", mean(pro), med(pro), a.walk);
    std::printf("  main loop     %8.0f   %8.0f   (MFMA issue per wave: %d cycles)\n", mean(mainl), med(mainl), CIN / 32 * 144 * 16);
    std::printf("  epilogue      %8.0f   %8.0f\n", mean(epi), med(epi));
    std::printf("  items per wg  %8.1f   %8.0f\n", mean(drain), med(drain));
    std::printf("  per item      %8.0f   %8.0f\n", mean(life), med(life));
    std::printf("  ticks per ms of the last launch: %.0f\n", (double)(tmax - tmin) / ms);
    char name[128]; std::snprintf(name, sizeof name, "gpurun_out/h16_trace_%d_%d.csv", CIN, COUT);
    if (std::FILE* f = std::fopen(name, "w")) {
        std::fprintf(f, "block,t0,t1,t2,t3,t4,hwid,xcc\n");
        for (int b = 0; b < grid; b += 7) { const unsigned long long* t = &tr[(size_t)b * 8]; std::fprintf(f, "%d,%llu,%llu,%llu,%llu,%llu,%llu,%llu\n", b, t[0], t[1], t[2], t[3], t[4], t[5], t[6]); }
        std::fclose(f);
    }
    CK(hipFree(din)); CK(hipFree(dw)); CK(hipFree(dout)); CK(hipFree(dbias)); CK(hipFree(dtr));
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 512;
    const bool zero = argc > 2 && std::string(argv[2]) == "zero";   // all-zero operands: the clock stays up, what remains is structure
    const bool walk_ok = !(argc > 3 && std::string(argv[3]) == "nowalk");
    run<128, 64>(N, 128, 128, zero, walk_ok);
    run<64, 64>(N, 128, 128, zero, walk_ok);
    run<128, 128>(N, 64, 64, zero, walk_ok);
    run<256, 128>(N / 4 * 4, 64, 64, zero, walk_ok);
    run<256, 256>(N, 32, 32, zero, walk_ok);
    return 0;
}
