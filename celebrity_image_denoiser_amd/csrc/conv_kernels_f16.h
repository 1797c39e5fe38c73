// conv_kernels_f16.h — fp16-storage / fp32-accumulate variant of the forward (BASELINE configs[4]).
//
// Same function and fused epilogues as the fp32 kernels (conv_kernels.h; reference backend/app.py:43-103), with activations and
// weights held as IEEE half in HBM/LDS and the contractions on the fp16 MFMAs with fp32 accumulators.  Bias, ReLU, max-pool and the
// transposed-convolution scatter are applied to the fp32 accumulators; results are rounded to half once, at the store.  This is a
// separate numerical contract from the fp32 path (stated tolerance in tests/test_gpu_parity.py), selected with cid_set_compute_dtype.
//
//   k_conv_head_h16   down1[0]                       v_mfma_f32_16x16x32_f16, K = 27 -> one step
//   k_conv3x3_h16     the eight 3x3 layers           v_mfma_f32_16x16x32_f16, 32-channel chunks, B by LDS-DMA
//   k_convt_t16       up2, up1                       v_mfma_f32_16x16x32_f16 (streaming: weights in registers, pixels by LDS-DMA)
//   k_conv_tail_h     upconv1[2] + tanh              v_mfma_f32_32x32x16_f16 (z = x . W per halo pixel, then nine shifted sums)
//
// The forward of this path runs at the board's power limit: the 16x16x32 shape moves half the accumulator
// registers per FLOP of 32x32x16 and sustains 1.22x its rate there (tools/mfma_shape_probe), which is why the heavy layers use it.
#pragma once
#include "conv_kernels.h"

#if (defined(H16_SPLIT_IN) || defined(H16_SPLIT_OUT)) && !defined(CID_EXPERIMENTS)
#error "the split-operand prototype is built only by csrc/tools (-DCID_EXPERIMENTS)"
#endif
#ifndef H16_ABLATE   // timing / energy experiments of csrc/tools/h16_trace only (wrong results when non-zero): 1 no B DMA after a workgroup's first item,
#define H16_ABLATE 0  // 2 no halo loads after the first item, 4 no stores, 8 fragments read from LDS once per item (MFMAs on stale registers)
#endif

namespace cid {

constexpr int HPS = 5;   // LDS slots (16 B) per pixel: 4 data (32 halfs) + 1 pad

struct GemmConvArgsH {
    const _Float16* in;   // NHWC half [N, Hin, Win, in_ps]
    const _Float16* w;    // packed: [nb][chunk][tap][kstep][ns][lane][8]
    const float* bias;    // [COUT] fp32
    _Float16* out;
    _Float16* pool;
    int N, Hin, Win, in_ps;
    int Hc, Wc, Hs, Ws;
    int out_ps, out_coff;
    int tiles_x, tiles_y, tiles_total, tiles_per_xcd;
    unsigned rcp_x, rcp_xy;   // ceil(2^32 / tiles_x), ceil(2^32 / (tiles_x*tiles_y)): division by multiply-high (host: tile_rcp)
    int walk;                 // k_conv3x3_h16: 0 = one (tile, column block) per workgroup; > 0 = tile walkers per XCD group (gridDim.x / 8)
    // k_conv3x3_h16<.., ZOUT> re-uses two fields instead of growing the struct (every kernel argument costs SGPRs, and these kernels sit at
    // the SGPR limit: two more pointers spilled 4 VGPRs in every variant): `out` = z[N][9 taps][Hs][Ws][4] halfs (co 0..2, one pad),
    // `pool` = upconv1[2]'s weights as A fragments, [3 row tiles][2 k-steps][64 lanes][8] halfs (cid_api.hip, hz_off)
};

// ---------------------------------------------------------------------------------------------
// k_convt_t16 — the transposed convolutions of the fp16-storage path: up2 / up1 = ConvTranspose2d(C, C/2, 2, stride=2) (app.py:65,73)
// as a STREAMING kernel (round 3).  At fp16 a 2x2 stride-2 ConvT is bound by HBM (171 FLOP/B at CIN = 256, 85 at CIN = 128, against a
// ridge of 312): every input pixel is read once and four output pixels are written, with no halo.  Rounds 1-2's k_convt_h (git history)
// reached 0.35 / 0.55 of that roofline: one workgroup per (tile, tap, 64 channels) re-read the input tile per tap, streamed its weights
// from L2 per tile and went through two barriers per 32-channel chunk.  Here (0.69 / 0.68, profiles/r03_ab_f16_convt.txt):
//   * wave = (tap, 64-channel block): its CIN x 64 weights stay in REGISTERS for the life of the workgroup (64 VGPRs at CIN = 128,
//     128 at CIN = 256) as the A operands of v_mfma_f32_16x16x32_f16 — channels are the MFMA ROWS, pixels the columns;
//   * workgroups are persistent and walk tiles of 32 KiB of input pixels (TP consecutive pixels of one image, all CIN channels:
//     one contiguous run of memory), brought in by LDS-DMA into one of two buffers while the other is computed on: ONE barrier per tile;
//   * LDS holds a pixel's 16-byte slots XOR-swizzled with the pixel index (phys = s ^ (p & 15)), applied on the SOURCE side of the DMA
//     (lane L always lands at L * 16): the pixel operand is read with conflict-free ds_read_b128 although pixels are 256 / 512 B apart;
//   * row m = 4 kg + r of M tile mt is channel 32 (mt >> 1) + 8 kg + 4 (mt & 1) + r (packed_index_ht): a lane's two M tiles of a pair
//     are 8 CONSECUTIVE channels of one pixel, so results leave as 16-byte stores straight from the accumulators (four lanes = 64
//     contiguous bytes of a pixel), no staging.
template <int CIN, int COUT>
struct ConvTGeom {
    static constexpr int CB = COUT / 64;            // 64-channel blocks
    static constexpr int NW = 4 * CB;               // waves per workgroup: (tap, block)
    static constexpr int S = CIN / 8;               // 16-byte slots per pixel
    static constexpr int TP = 2048 / S;             // pixels per tile (32 KiB of input)
    static constexpr int TN = TP / 16;              // 16-pixel MFMA column tiles
    static constexpr int KS = CIN / 32;             // k-steps
    static constexpr int ROUNDS = TP * S / (NW * 64);   // DMA instructions per wave and tile; 16 pixels per round
    static constexpr int BUFQ = TP * S;             // quads per buffer
    static_assert(COUT % 64 == 0 && CIN % 32 == 0 && (S == 16 || S == 32), "layer dims");
    static_assert(NW * 64 / S == 16, "a DMA round covers 16 pixels: the swizzle term (p & 15) is the same in every round");
};

// GemmConvArgsH as k_convt_t16 reads it: tiles_x = tiles per image, tiles_total = N * tiles_x, rcp_x = tile_rcp(tiles_x),
// rcp_xy = tile_rcp(Win); Hin x Win = the input image (pixel stride in_ps halfs), Hc x Wc = the part of it that is computed.
template <int CIN, int COUT>
__global__ void __launch_bounds__(64 * 4 * (COUT / 64), 2) k_convt_t16(const GemmConvArgsH a) {
    using G = ConvTGeom<CIN, COUT>;
    constexpr int S = G::S, TP = G::TP, TN = G::TN, KS = G::KS, NW = G::NW, ROUNDS = G::ROUNDS, BUFQ = G::BUFQ, CB = G::CB, TG = 4;
    static_assert(TN % TG == 0, "column tiles are processed TG at a time");
    __shared__ f32x4 lds[2 * BUFQ];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tap = wave & 3, cb = wave >> 2, kh = tap >> 1, kw = tap & 1;
    const int c16 = lane & 15, kg = lane >> 4;
    int tile = blockIdx.x;
    if (tile >= a.tiles_total) return;
    const int HW = a.Hin * a.Win;

    // ---- this wave's weights and bias, once ----
    f16x8 wf[KS][4];
    {
        const f16x8* wp = reinterpret_cast<const f16x8*>(a.w) + (size_t)((tap * CB + cb) * KS) * 4 * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) wf[ks][mt] = wp[(ks * 4 + mt) * 64];
    }
    f32x4 bias4[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) bias4[mt] = *reinterpret_cast<const f32x4*>(a.bias + cb * 64 + 32 * (mt >> 1) + 8 * kg + 4 * (mt & 1));

    // ---- input DMA: lane L of (wave, round j) fills physical quad ((j * NW + wave) * 64 + L) of the buffer ----
    const int pl0 = (wave * 64 + lane) / S, sp = lane & (S - 1);          // round 0: tile-local pixel, physical slot
    const unsigned dma_lane = (unsigned)(pl0 * a.in_ps * 2 + ((sp ^ (pl0 & 15)) * 16));
    auto image_rsrc = [&](int n) {
        const unsigned long long p = (unsigned long long)(a.in + (size_t)n * HW * a.in_ps);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), (short)0, HW * a.in_ps * 2, 0x00020000);
    };
    auto split = [&](int t, int& n, int& chunk) {                          // tile -> image, run of TP pixels inside it
        n = a.tiles_x > 1 ? (int)__umulhi((unsigned)t, a.rcp_x) : t;   // at most one too large (see the pixel decode below)
        chunk = t - n * a.tiles_x;
        if (chunk < 0) { --n; chunk += a.tiles_x; }
    };
    auto dma_tile = [&](int t, int buf) {
        int n, chunk;
        split(t, n, chunk);
        const __amdgpu_buffer_rsrc_t rsrc = image_rsrc(n);
#pragma unroll
        for (int j = 0; j < ROUNDS; ++j) {
            const int q = chunk * TP + j * 16 + pl0;                       // pixel of the image; beyond it: zeros.  The mask is carried by the
            const unsigned vo = q < HW ? dma_lane : 0x7ffffff0u;           // PER-LANE offset: nothing relies on the scalar offset being range-checked
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)&lds[buf * BUFQ + (j * NW + wave) * 64], 16, vo,
                                                     (chunk * TP + j * 16) * a.in_ps * 2, 0, 0);
        }
    };

    // ---- pixel operand: lane (c16, kg) reads slot 4 ks + kg of pixel 16 t + c16: physical quad (16 t + c16) * S + ((4 ks + kg) ^ c16) ----
    const f16x8* ldsh = reinterpret_cast<const f16x8*>(lds);
    int rd[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) rd[ks] = c16 * S + ((4 * ks + kg) ^ c16);

    // ---- output: per image a buffer over [2 Hc][2 Wc] pixels of out_ps halfs ----
    const int Ho = 2 * a.Hc, Wo = 2 * a.Wc;
    auto out_rsrc = [&](int n) {
        const unsigned long long p = (unsigned long long)(a.out + (size_t)n * Ho * Wo * a.out_ps);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), (short)0, Ho * Wo * a.out_ps * 2, 0x00020000);
    };
    const unsigned lane_chan = (unsigned)((a.out_coff + cb * 64 + 8 * kg) * 2);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    dma_tile(tile, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (;;) {
        const int next = tile + gridDim.x;
        const bool has_next = next < a.tiles_total;                         // workgroup-uniform
        if (has_next) dma_tile(next, buf ^ 1);                              // lands under this tile's MFMAs and stores

        int n, chunk;
        split(tile, n, chunk);
        const __amdgpu_buffer_rsrc_t ro = out_rsrc(n);
#pragma unroll
        for (int t0 = 0; t0 < TN; t0 += TG) {   // TG column tiles at a time: 16 TG accumulator registers beside the resident weights
            f32x4 acc[TG][4];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int t = 0; t < TG; ++t) {
                    const f16x8 px = ldsh[buf * BUFQ + (t0 + t) * 16 * S + rd[ks]];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        if (ks == 0) {
                            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                            acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][mt], px, zero, 0, 0, 0);
                        } else {
                            acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][mt], px, acc[t][mt], 0, 0, 0);
                        }
                    }
                }
            // The next tile's DMA (issued a whole tile of MFMAs ago) has landed for this wave before the LAST group's stores go out: waited
            // for here, with nothing but long-finished requests outstanding, because vector loads and stores may complete out of order
            // with respect to each other on gfx9 — a count taken after the stores would not single the loads out.
            if (has_next && t0 + TG == TN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // ---- bias (no activation: app.py:89,96), one rounding to half, tap (kh, kw) of pixel (y, x) -> output pixel (2y + kh, 2x + kw) ----
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const unsigned q = (unsigned)(chunk * TP + (t0 + t) * 16 + c16);
                unsigned y = __umulhi(q, a.rcp_xy);                         // q / Win, at most one too large (q * (rcp * Win - 2^32) < 2^32 * Win)
                int x = (int)(q - y * (unsigned)a.Win);
                if (a.Win == 1) { y = q; x = 0; }
                if (x < 0) { --y; x += a.Win; }
                const bool ok = (int)q < HW && (int)y < a.Hc && x < a.Wc;
                const unsigned po = ((2 * y + kh) * (unsigned)Wo + 2 * x + kw) * (unsigned)a.out_ps * 2 + lane_chan;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    f16x8 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = (_Float16)(acc[t][2 * hf][r] + bias4[2 * hf][r]);
                        v[4 + r] = (_Float16)(acc[t][2 * hf + 1][r] + bias4[2 * hf + 1][r]);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, ok ? po + hf * 64 : 0x7ffffff0u, 0, 0);
                }
            }
        }
        if (!has_next) break;
        __syncthreads();   // every wave's part of the next tile has landed, and every wave has left this tile's buffer
        tile = next;
        buf ^= 1;
    }
}

// Row permutation of the 16x16x32 kernels: MFMA row i of a 16-pixel group is pixel h16_prow(i) = {2,0,8,10}[i/4] + (i&1) + 4*((i>>1)&1).
__device__ __forceinline__ int h16_prow(int i) { return ((0xa802 >> (4 * (i >> 2))) & 15) + (i & 1) + 2 * (i & 2); }

// Stores of the 16x16x32 kernels (round 4: straight from the accumulators).  Column j of channel group cg is output channel 4 j + cg of
// the workgroup's 64 (packed_index_h16 / the head's packing put the weights there): in an accumulator tile, lane (column c16 = lane & 15,
// row group kg = lane >> 4) holds pixels 16 pg + {2,0,8,10}[kg] + (r&1) + 4(r>>1) (r = 0..3) of one output row, and its four channel
// groups are four CONSECUTIVE channels 4 c16 .. 4 c16 + 3 — 8 bytes as halfs.  A store instruction therefore writes four pixels'
// whole 128-byte lines (16 lanes each) with no pass through LDS: rounds 2-3 staged every row as fp32 in a wave-private LDS area
// (32 ds_write_b32 + 8 ds_read_b128 per lane and row, two wave fences, a workgroup barrier in front) to form 16-byte stores.
// Same box: the 3x3 launches with 128 or 256 output channels per pixel -2...-3 %, upconv1.0 +-0, forward +0.9...+1.4 %
// (profiles/r04_ab_f16_epilogue_layouts.txt).  NPIX = 32: a full row; NPIX = 16: the pooled row (pixels 8 pg + {1,0,4,5}[kg] + 2r,
// r = 0..1).  value(pg, cg, r) yields the finished fp32 element, rounded to half once, here.
//   non-temporal: the activations a launch writes (1 GB) are not read again before they have left the L2 anyway, but as ordinary
//   stores they displace the input tiles that neighbouring workgroups are about to share (96 walkers x 44 KB of halo tile per XCD
//   against 4 MB of L2): same box, every 3x3 launch -2...-6 %, head -5 % (profiles/r03_nt_stores.txt).  NOT for stores that rely on
//   the L2 to merge partial lines (k_convt_t16: +50 %); here every line is written whole by one instruction.
//   Addressing: raw buffer stores over ONE image (`rsrc`), a per-lane byte offset that holds for the whole row (the lane's first pixel
//   and channel quad) plus a scalar offset per (pixel group, register) — one VGPR of address instead of a 64-bit pointer per store
//   (these kernels sit at the 168-register step).  `row_off` = byte offset of the row's pixel 0, channel 0 of the column block (scalar).
template <int NPIX, typename V>
__device__ __forceinline__ void h16_store_row(int lane, V value, const __amdgpu_buffer_rsrc_t& rsrc, unsigned row_off, int stride, int xlim, bool rowok, bool full) {
    constexpr int per = NPIX == 32 ? 4 : 2;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const int c16 = lane & 15, kg = lane >> 4;
    const int pb = (0xa802 >> (4 * kg)) & 15, p0 = NPIX == 32 ? pb : pb >> 1;            // the lane's first pixel of a 16 (8)-pixel group
    const unsigned lane_off = (unsigned)((p0 * stride + 4 * c16) * 2);
#pragma unroll
    for (int pg = 0; pg < 2; ++pg)
#pragma unroll
        for (int r = 0; r < per; ++r) {
            const int dp = (NPIX / 2) * pg + (NPIX == 32 ? (r & 1) + 4 * (r >> 1) : 2 * r);   // compile-time
            const f16x4 v = {(_Float16)value(pg, 0, r), (_Float16)value(pg, 1, r), (_Float16)value(pg, 2, r), (_Float16)value(pg, 3, r)};
            const unsigned vo = (!(H16_ABLATE & 4) && (full || (rowok && p0 + dp < xlim))) ? lane_off : 0x7ffffff0u;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rsrc, vo, row_off + (unsigned)(dp * stride * 2), /*nt*/ 2);   // re-measured r4: plain stores -1.8 %
        }
}

// Epilogue of k_conv3x3_h16: bias + ReLU on the wave's two rows (+ the 2x2 max-pooled copy).
template <int COUT, int MODE, typename Args>
__device__ __forceinline__ void h16_epilogue(const Args& a, f32x4 (&acc)[2][2][4], const f32x4& bias_v, int n, int y0, int x0,
                                             int wave, int lane, int cobase) {
    const bool full = y0 + TILE_H <= a.Hs && x0 + TILE_W <= a.Ws;
    auto image_out = [&](_Float16* base, size_t elems) {   // descriptor over image n of a tensor with `elems` halfs per image (< 2^30: cid_api.hip shape_error)
        const unsigned long long p = (unsigned long long)(base + (size_t)n * elems);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), (short)0, (int)(elems * 2), 0x00020000);
    };
    {
        const __amdgpu_buffer_rsrc_t ro = image_out(a.out, (size_t)a.Hs * a.Ws * a.out_ps);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int y = __builtin_amdgcn_readfirstlane(y0 + 2 * wave + m);
            const unsigned row_off = (unsigned)(((y * a.Ws + x0) * a.out_ps + a.out_coff + cobase) * 2);
            h16_store_row<32>(lane, [&](int pg, int cg, int r) { return fmaxf(acc[m][pg][cg][r] + bias_v[cg], 0.f); }, ro, row_off, a.out_ps, a.Ws - x0, y < a.Hs, full);
#ifdef H16_SPLIT_OUT   // experiment (csrc/tools/split_proto, -DCID_EXPERIMENTS): the fp32 result kept as TWO halfs — hi = half(v) at channel c (above), lo = half(v - hi) at channel out_ps / 2 + c
            h16_store_row<32>(lane, [&](int pg, int cg, int r) { const float v = fmaxf(acc[m][pg][cg][r] + bias_v[cg], 0.f); return v - (float)(_Float16)v; }, ro,
                              row_off + (unsigned)a.out_ps, a.out_ps, a.Ws - x0, y < a.Hs, full);
#endif
        }
    }
    if (MODE == 1) {   // 2x2 max-pool, floor mode: registers (r, r+1), r even, of the wave's two rows are one window
        const int Hp = a.Hc >> 1, Wp = a.Wc >> 1;
        const int py = __builtin_amdgcn_readfirstlane((y0 >> 1) + wave);
        const __amdgpu_buffer_rsrc_t rp = image_out(a.pool, (size_t)Hp * Wp * COUT);
        const unsigned prow_off = (unsigned)(((py * Wp + (x0 >> 1)) * COUT + cobase) * 2);
        h16_store_row<16>(lane,
                          [&](int pg, int cg, int r) {
                              const float v = fmaxf(fmaxf(acc[0][pg][cg][2 * r], acc[0][pg][cg][2 * r + 1]), fmaxf(acc[1][pg][cg][2 * r], acc[1][pg][cg][2 * r + 1]));
                              return fmaxf(v + bias_v[cg], 0.f);
                          },
                          rp, prow_off, COUT, Wp - (x0 >> 1), py < Hp, false);
    }
}

// F32IO (conv_algo = "split16"): the same epilogue writing FP32 tensors — a lane's four channel groups are four consecutive channels = 16 bytes; a.out / a.pool point to float data,
// out_ps / out_coff count floats.
template <int COUT, int MODE, typename Args>
__device__ __forceinline__ void h16_epilogue_f32(const Args& a, f32x4 (&acc)[2][2][4], const f32x4& bias_v, int n, int y0, int x0, int wave, int lane, int cobase) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    const bool full = y0 + TILE_H <= a.Hs && x0 + TILE_W <= a.Ws;
    const int c16 = lane & 15, kg = lane >> 4, pb = (0xa802 >> (4 * kg)) & 15;
    // 16-byte buffer store with a scalar offset, its data registers held untouched for four more wait states: the store-data hazard of the register-soffset form
    // (wino42_kernels.h store16, profiles/r03_store_hazard.txt; tools/store_hazard_check.py found the unguarded form of this epilogue at once)
    auto store16 = [](f32x4 v, const __amdgpu_buffer_rsrc_t& rsrc, unsigned vo, unsigned soff) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), rsrc, vo, soff, /*nt*/ 2);
        asm volatile("s_nop 3" ::"v"(v) : "memory");
    };
    auto image_out = [&](float* base, size_t elems) {
        const unsigned long long p = (unsigned long long)(base + (size_t)n * elems);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), (short)0, (int)(elems * 4), 0x00020000);
    };
    {
        const __amdgpu_buffer_rsrc_t ro = image_out(reinterpret_cast<float*>(a.out), (size_t)a.Hs * a.Ws * a.out_ps);
        const unsigned lane_off = (unsigned)((pb * a.out_ps + 4 * c16) * 4);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int y = __builtin_amdgcn_readfirstlane(y0 + 2 * wave + m);
            const unsigned row_off = (unsigned)(((y * a.Ws + x0) * a.out_ps + a.out_coff + cobase) * 4);
#pragma unroll
            for (int pg = 0; pg < 2; ++pg)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int dp = 16 * pg + (r & 1) + 4 * (r >> 1);
                    const f32x4 v = {fmaxf(acc[m][pg][0][r] + bias_v[0], 0.f), fmaxf(acc[m][pg][1][r] + bias_v[1], 0.f), fmaxf(acc[m][pg][2][r] + bias_v[2], 0.f), fmaxf(acc[m][pg][3][r] + bias_v[3], 0.f)};
                    const unsigned vo = (full || (y < a.Hs && pb + dp < a.Ws - x0)) ? lane_off : 0x7ffffff0u;
                    store16(v, ro, vo, row_off + (unsigned)(dp * a.out_ps * 4));
                }
        }
    }
    if (MODE == 1) {   // 2x2 max-pool, floor mode (as h16_epilogue)
        const int Hp = a.Hc >> 1, Wp = a.Wc >> 1;
        const int py = __builtin_amdgcn_readfirstlane((y0 >> 1) + wave);
        const __amdgpu_buffer_rsrc_t rp = image_out(reinterpret_cast<float*>(a.pool), (size_t)Hp * Wp * COUT);
        const unsigned prow_off = (unsigned)(((py * Wp + (x0 >> 1)) * COUT + cobase) * 4);
        const int p0 = pb >> 1;
        const unsigned lane_off = (unsigned)((p0 * COUT + 4 * c16) * 4);
#pragma unroll
        for (int pg = 0; pg < 2; ++pg)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int dp = 8 * pg + 2 * r;
                f32x4 v;
#pragma unroll
                for (int cg = 0; cg < 4; ++cg)
                    v[cg] = fmaxf(fmaxf(fmaxf(acc[0][pg][cg][2 * r], acc[0][pg][cg][2 * r + 1]), fmaxf(acc[1][pg][cg][2 * r], acc[1][pg][cg][2 * r + 1])) + bias_v[cg], 0.f);
                const unsigned vo = (py < Hp && p0 + dp < Wp - (x0 >> 1)) ? lane_off : 0x7ffffff0u;
                store16(v, rp, vo, prow_off + (unsigned)(dp * COUT * 4));
            }
    }
}

// ZOUT epilogue of k_conv3x3_h16<128, 64> (round 4): upconv1[0] as the producer of the last layer's input.  What the fp32 path does since
// round 2 (wino42_kernels.h, ZOUT): upconv1[2] = Conv2d(64, 3, 3, padding=1) (app.py:77) is a 1x1 contraction per tap followed by a
// nine-tap shifted sum, and the contraction has no halo — so it runs HERE, on the wave's finished 2 rows x 32 pixels x 64 channels, and
// the 64-channel tensor (128 B per pixel written, then re-read with a 10x34 / 8x32 halo by k_conv_tail_h) never exists:
//     z[n][row = 3 tap + co][y][x] = sum_ci half(relu(upconv1.0))[n][y][x][ci] * W2[co][ci][tap]   56 B per pixel: 27 rows + 1 pad = 7 groups of 4 halfs
//     (first form of round 4: row 4 tap + co, 9 groups = 72 B per pixel and three row tiles of MFMAs)
// The activation is rounded to half exactly as the stored tensor was, so only z's own rounding to half is new (CPU emulation of the
// whole path, He-gain weights, 16 images: 3.2e-3 against 3.4e-3 unfused, z half vs unfused 8.9e-4; tools/emulate_f16_zout.py).
//   * the wave's pixels go to a wave-private LDS area [row][pixel][8 slots of 8 channels], slot XOR (pixel & 7): a lane's four
//     consecutive channels are one ds_write_b64 (16 lanes = one pixel's 128 B: conflict-free), and the B operand of the z product —
//     lane (column = pixel, k-group) = 8 consecutive channels — one ds_read_b128 (conflict-free by the XOR, checked exhaustively);
//   * z^T = W2' . X^T on v_mfma_f32_16x16x32_f16 with the WEIGHTS as rows: row 3 tap + co (27 of 32 rows used: two row tiles),
//     K = 64 channels = two steps -> 16 MFMAs per wave beside the 576 of the main loop; a lane of the result holds four consecutive
//     rows of ONE pixel = one group: 8 bytes as halfs, stored straight from registers (16 lanes = 16 consecutive pixels of a group plane).
// the fused last layer's partial sums: 27 rows (3 tap + co) + 1 pad per pixel as 7 groups of four halfs, z[n][group][y][x][4]
constexpr int Z_GROUPS = 7;
template <typename Args>
__device__ __forceinline__ void h16_zout_epilogue(const Args& a, f32x4* stage, f32x4 (&acc)[2][2][4], const f32x4& bias_v, int n, int y0, int x0,
                                                  int wave, int lane) {
    const int c16 = lane & 15, kg = lane >> 4;
    unsigned char* const stg = reinterpret_cast<unsigned char*>(stage + wave * 512);     // 2 rows x 32 pixels x 128 B
    const int pb = (0xa802 >> (4 * kg)) & 15;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int pg = 0; pg < 2; ++pg)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int P = 16 * pg + pb + (r & 1) + 4 * (r >> 1);
                const f16x4 v = {(_Float16)fmaxf(acc[m][pg][0][r] + bias_v[0], 0.f), (_Float16)fmaxf(acc[m][pg][1][r] + bias_v[1], 0.f),
                                 (_Float16)fmaxf(acc[m][pg][2][r] + bias_v[2], 0.f), (_Float16)fmaxf(acc[m][pg][3][r] + bias_v[3], 0.f)};
                *reinterpret_cast<f16x4*>(stg + ((m * 32 + P) * 8 + ((c16 >> 1) ^ (P & 7))) * 16 + (c16 & 1) * 8) = v;
            }
    // the last layer's weights are requested only now, with the accumulators dead (24 registers; an L2 hit per item)
    f16x8 wz[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wz[t][ks] = reinterpret_cast<const f16x8*>(a.pool)[(t * 2 + ks) * 64 + lane];
    wave_lds_fence();
    const size_t plane = (size_t)a.Hs * a.Ws;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) {
            const int P = 16 * pg + c16;                                                   // column of the z tile = pixel
            f16x8 xb[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xb[ks] = *reinterpret_cast<const f16x8*>(stg + ((m * 32 + P) * 8 + ((4 * ks + kg) ^ (P & 7))) * 16);
            const int y = y0 + 2 * wave + m, x = x0 + P;
            const bool inside = y < a.Hs && x < a.Ws;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                f32x4 z = __builtin_amdgcn_mfma_f32_16x16x32_f16(wz[t][0], xb[0], zero, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(wz[t][1], xb[1], z, 0, 0, 0);
                const int grp = 4 * t + kg;                                               // rows 16 t + 4 kg + r = 3 tap + co: row group 4 t + kg of the 7 (Z_GROUPS) a pixel has
                const f16x4 hz = {(_Float16)z[0], (_Float16)z[1], (_Float16)z[2], (_Float16)z[3]};
                if (inside && grp < Z_GROUPS) *reinterpret_cast<f16x4*>(a.out + ((((size_t)n * Z_GROUPS + grp) * plane + (size_t)y * a.Ws + x) << 2)) = hz;
            }
        }
}

// F32IO, MODE 2 (the transposed convolutions under conv_algo = "split16"): column block `blk` = (tap = blk / (COUT / 64), 64-channel block blk % (COUT / 64)); the wave's input pixel
// (y, x) becomes output pixel (2y + kh, 2x + kw) of the [2 Hc][2 Wc] tensor (out_ps floats per pixel, out_coff: the cat slice).  Bias, no activation (app.py:65,73,89,96).
template <int COUT, typename Args>
__device__ __forceinline__ void h16_epilogue_t32(const Args& a, f32x4 (&acc)[2][2][4], int n, int y0, int x0, int wave, int lane, int blk) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    constexpr int CB = COUT / 64;
    const int tap = blk / CB, cb = blk - tap * CB, kh = tap >> 1, kw = tap & 1;
    const int c16 = lane & 15, kg = lane >> 4, pb = (0xa802 >> (4 * kg)) & 15;
    const int Ho = 2 * a.Hc, Wo = 2 * a.Wc;
    const f32x4 bias_v = *reinterpret_cast<const f32x4*>(a.bias + cb * 64 + 4 * c16);
    auto store16 = [](f32x4 v, const __amdgpu_buffer_rsrc_t& rsrc, unsigned vo, unsigned soff) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), rsrc, vo, soff, 0);     // partial lines: the L2 merges the four taps' stores (not non-temporal)
        asm volatile("s_nop 3" ::"v"(v) : "memory");
    };
    const unsigned long long p = (unsigned long long)(reinterpret_cast<float*>(a.out) + (size_t)n * Ho * Wo * a.out_ps);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), (short)0, Ho * Wo * a.out_ps * 4, 0x00020000);
    const unsigned lane_off = (unsigned)(((2 * pb + kw) * a.out_ps + 4 * c16) * 4);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int y = __builtin_amdgcn_readfirstlane(y0 + 2 * wave + m);
        const unsigned row_off = (unsigned)((((2 * y + kh) * Wo + 2 * x0) * a.out_ps + a.out_coff + cb * 64) * 4);
#pragma unroll
        for (int pg = 0; pg < 2; ++pg)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int dp = 16 * pg + (r & 1) + 4 * (r >> 1);
                const f32x4 v = {acc[m][pg][0][r] + bias_v[0], acc[m][pg][1][r] + bias_v[1], acc[m][pg][2][r] + bias_v[2], acc[m][pg][3][r] + bias_v[3]};
                const unsigned vo = (y < a.Hc && x0 + pb + dp < a.Wc) ? lane_off : 0x7ffffff0u;
                store16(v, ro, vo, row_off + (unsigned)(2 * dp * a.out_ps * 4));
            }
    }
}

// ZOUT epilogue of k_conv3x3_h16<128, 64, 0, ZOUT, ., F32IO> (conv_algo = "split16"): upconv1[2]'s 64 -> 27 contraction per tap (as h16_zout_epilogue above) in the
// split-operand arithmetic of the layer itself.  The wave's finished 2 rows x 32 pixels x 64 channels (bias + ReLU, fp32) are split into hi / lo halfs and staged in a
// wave-private LDS area (hi: [row][pixel][8 slots of 8 channels], slot XOR (pixel & 7); lo: the same 8 KiB further), the weights' hi / lo pieces come as A fragments
// (row 3 tap + co, two row tiles, two k-steps; cid_api.hip hzs_off), and z^T = hi_w.hi_x + lo_w.hi_x + hi_w.lo_x accumulates in fp32: 48 MFMAs per wave.
// z leaves as FP32 into the 27 planes k_conv_tail_z reads, z[n][3 tap + co][y][x] (a lane holds four consecutive planes of one pixel: four 4-byte stores, 16 lanes = 64 bytes of a plane).
template <typename Args>
__device__ __forceinline__ void h16_zout_epilogue_f32(const Args& a, f32x4* stage, f32x4 (&acc)[2][2][4], const f32x4& bias_v, int n, int y0, int x0, int wave, int lane) {
    const int c16 = lane & 15, kg = lane >> 4;
    unsigned char* const stg = reinterpret_cast<unsigned char*>(stage + wave * 1024);     // hi: 2 rows x 32 pixels x 128 B, then lo
    const int pb = (0xa802 >> (4 * kg)) & 15;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int pg = 0; pg < 2; ++pg)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int P = 16 * pg + pb + (r & 1) + 4 * (r >> 1);
                const float v0 = fmaxf(acc[m][pg][0][r] + bias_v[0], 0.f), v1 = fmaxf(acc[m][pg][1][r] + bias_v[1], 0.f);
                const float v2 = fmaxf(acc[m][pg][2][r] + bias_v[2], 0.f), v3 = fmaxf(acc[m][pg][3][r] + bias_v[3], 0.f);
                const f16x4 hi = {(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
                const f16x4 lo = {(_Float16)(v0 - (float)hi[0]), (_Float16)(v1 - (float)hi[1]), (_Float16)(v2 - (float)hi[2]), (_Float16)(v3 - (float)hi[3])};
                unsigned char* const d = stg + ((m * 32 + P) * 8 + ((c16 >> 1) ^ (P & 7))) * 16 + (c16 & 1) * 8;
                *reinterpret_cast<f16x4*>(d) = hi;
                *reinterpret_cast<f16x4*>(d + 8192) = lo;
            }
    f16x8 wzh[2][2], wzl[2][2];                       // requested only now, with the accumulators dead
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            wzh[t][ks] = reinterpret_cast<const f16x8*>(a.pool)[(t * 2 + ks) * 64 + lane];
            wzl[t][ks] = reinterpret_cast<const f16x8*>(a.pool)[(4 + t * 2 + ks) * 64 + lane];
        }
    wave_lds_fence();
    const size_t plane = (size_t)a.Hs * a.Ws;
    float* const zout = reinterpret_cast<float*>(a.out) + (size_t)n * 27 * plane;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) {
            const int P = 16 * pg + c16;
            f16x8 xh[2], xl[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned char* s_ = stg + ((m * 32 + P) * 8 + ((4 * ks + kg) ^ (P & 7))) * 16;
                xh[ks] = *reinterpret_cast<const f16x8*>(s_);
                xl[ks] = *reinterpret_cast<const f16x8*>(s_ + 8192);
            }
            const int y = y0 + 2 * wave + m, x = x0 + P;
            const bool inside = y < a.Hs && x < a.Ws;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                f32x4 z = __builtin_amdgcn_mfma_f32_16x16x32_f16(wzh[t][0], xh[0], zero, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(wzh[t][1], xh[1], z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(wzl[t][0], xh[0], z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(wzl[t][1], xh[1], z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(wzh[t][0], xl[0], z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(wzh[t][1], xl[1], z, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * t + 4 * kg + r;                                   // plane 3 tap + co
                    if (inside && row < 27) zout[(size_t)row * plane + (size_t)y * a.Ws + x] = z[r];
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------
// 3x3 layers of the fp16-storage path, third form: v_mfma_f32_16x16x32_f16.
//
// The fp16 forward runs at the board's power cap, so what counts is energy per FLOP.  On random,
// LDS-fed operands the 16x16x32 shape sustains 1.93 PFLOP/s where 32x32x16 sustains 1.58 (tools/mfma_shape_probe: same
// FLOPs and LDS bytes per step; a 32x32 tile rewrites 16 accumulator registers per instruction, a 16x16 tile 4 for half the
// FLOPs).  Same workgroup tile as before (8x32 pixels x 64 channels, wave = 2 rows), 32-channel chunks:
//   * wave tile = 4 pixel groups (2 rows x 2 halves of 16) x 4 channel groups of 16 -> 16 accumulator tiles of 4 registers;
//     per tap: 4 A + 4 B fragments (ds_read_b128 each), 16 MFMAs;
//   * A: lane (pixel = l & 15, k-group = l >> 4) takes channels 8kg..8kg+7 -> the halo tile lies in LDS as four planes
//     [k-group][340 pixel slots] of 16-byte quads.  A ds_read_b128 is served in groups of 16 lanes — {0-3,12-15} of one k-group with
//     {4-11} of the next (MI355X_MICROARCH.md, LDS): with planes 340 = 4 (mod 16) slots apart and the MFMA rows permuted (h16_prow)
//     those cover 16 different slots (mod 16): every fragment READ is conflict-free, every (tap, row, half) address an immediate.
//     The halo WRITES (ds_write_b128, served 8 lanes at a time: two neighbouring pixels x four k-groups = two coalesced 64-byte load quads)
//     were 2-way conflicted by construction in rounds 2-3 (plane origins 0, 4, 0, 4 mod 8): most of the 18-29 % SQ_LDS_BANK_CONFLICT /
//     SQ_LDS_IDX_ACTIVE these launches showed.  Round 4 first measured the alternatives that re-order the PIECES (8 consecutive pixels of a
//     k-group per 8 lanes: conflict-free but the four lanes of a load quad then fetch from different pixels, 3x3 launches +3...4 %; lane
//     pairs of 32 bytes: +1...3 %; planes of 352 slots: writes 4-way, +1.5 %; profiles/r04_ab_f16_epilogue_layouts.txt) — coalesced loads
//     win — and then found the layout that keeps them: two pad slots between planes 1 and 2 (PLANE_GAP below) put the plane origins at
//     0, 4, 2, 6 (mod 8) while the read pairs (0,1) and (2,3) stay 4 (mod 16) apart: reads and writes conflict-free, 0.0 % measured on every
//     launch of the path, 3x3 launches -0.3...-0.8 % (profiles/r04_ab_f16_plane_gap.txt);
//   * B: packed on the host per lane, [chunk][dx][dy][channel group][lane = 16 kg + col][8] (cid_api.hip packed_index_h16);
//     one tap column (12 KiB) at a time by LDS-DMA into one of two buffers — no staging registers, and the LDS stays under a
//     third of the CU's; the halo tile of the next chunk waits in 24 registers over the chunk's last sub-step;
//   * per sub-step the wave reads its four input rows once (8 A quads) and 12 B quads for 48 MFMAs.
//   * (r3; since r4 an option, not the default: see WALK below) walking workgroups, as k_wino42_conv: with more (tile, column block) items than three workgroups per CU the grid is what is
//     resident and a workgroup walks tiles local, local + walk, ... of its XCD group, all NB column blocks of a tile back to back.
//     Under the LAST sub-step of an item the next item's first B sub-chunk is fetched into B buffer 0 (free by then: the last
//     sub-step reads buffer 1; the epilogue's staging starts behind buffer 0) and its first halo chunk into the 24 staging registers,
//     so an item's prologue (B DMA + halo request + their HBM latency, ~5-7k cycles beside 20-40k of work) is paid once per
//     workgroup.  Everything renewed per item (B offset, halo offsets, image descriptor) is derived from per-item opaque values, or
//     hipcc hoists it out of the item loop into registers this 168-register kernel does not have.
//   * (r4) WALK = false: the product's default form, one (tile, column block) item per workgroup — with the epilogue storing straight from the
//     accumulators it measures faster than walking on every layer (profiles/r04_ab_f16_walk_vs_not.txt), and as a compile-time fact it removes the
//     next-item decode, the prefetch under the last sub-step and the item boundary from the code.  WALK = true (cid_debug_half_workgroups_per_cu > 0)
//     stays a tested option with the same bits.
template <int CIN, int COUT, int MODE, bool ZOUT = false, bool WALK = false, bool F32IO = false, bool PAIR = false>
__global__ void __launch_bounds__(THREADS, F32IO ? 2 : 3) k_conv3x3_h16(const GemmConvArgsH a) {
    // PAIR (F32IO only): one workgroup computes TWO 64-channel column blocks of its tile from one staging of the input — the hi / lo planes of a chunk serve nine sub-steps into
    // `acc` (block 2 nbp) and nine into `acc2` (block 2 nbp + 1): half the loads, conversions, LDS writes and chunk seams per MFMA, half the workgroups.
    static_assert(!PAIR || (F32IO && !ZOUT && !WALK && (MODE == 2 || COUT % 128 == 0)), "PAIR: split16 layers with at least two column blocks");
#ifndef CID_EXPERIMENTS
    static_assert(H16_ABLATE == 0, "ablation variants are built only by csrc/tools (-DCID_EXPERIMENTS)");
#endif
    static_assert(MODE == 0 || MODE == 1 || (MODE == 2 && F32IO && PAIR), "3x3 layers; MODE 2 (2x2 stride-2 transposed convolution) only in the split-operand form");
    static_assert(!ZOUT || (COUT == 64 && MODE == 0), "the fused last layer contracts the 64 channels of ONE column block");
    constexpr int LW = TILE_W + 2, LH = TILE_H + 2, LPIX = LW * LH;       // 340
    constexpr int PLANE = LPIX;                                           // slots per k-group plane; 340 = 4 mod 16, see above
    static_assert(PLANE % 16 == 4, "the fragment row permutation below assumes plane stride = 4 mod 16");
    // F32IO (conv_algo = "split16" of the fp32 path): CIN counts FP32 channels; a chunk = 32 of them = 128 bytes per pixel in 8 pieces of 16 bytes, split into a hi and a lo set of planes
    // while staging; nine sub-steps per chunk (hi_x . hi_w, hi_x . lo_w, lo_x . hi_w for each tap column), the host packs the weight sub-chunks in that order
    constexpr int NSLOT = LPIX * (F32IO ? 8 : 4), NLOAD = (NSLOT + THREADS - 1) / THREADS;   // 6 (11)
    constexpr int NCHUNK = CIN / 32, NSUB = NCHUNK * (MODE == 2 ? 1 : F32IO ? 9 : 3);   // packed 12-KiB weight sub-chunks per column block
    constexpr int NB = (MODE == 2 ? 4 * COUT : COUT) / NTILE;            // column blocks; MODE 2: (tap, 64 output channels)
    constexpr int BSUB = 3 * 4 * 64;                                      // quads of one B sub-chunk (column dx: 3 dy x 4 cg), 12 KiB
    // (r4) ... and the k-group planes lie at 0, PLANE, 2 PLANE + 2, 3 PLANE + 2: the pairs (0,1) and (2,3) that share a ds_read_b128 service group stay
    // 4 (mod 16) slots apart (reads conflict-free as before), while the four plane origins are now 0, 4, 2, 6 (mod 8) — the eight lanes of a
    // ds_write_b128 service group (two neighbouring pixels x four k-groups, i.e. two coalesced 64-byte load quads) hit eight different slots:
    // the halo writes, 2-way conflicted by construction in rounds 2-3, are conflict-free too (checked exhaustively with the reads)
    constexpr int PLANE_GAP = 2;
    constexpr int HALO_SLOTS = 4 * PLANE + PLANE_GAP;
    // LDS: [B buffer 0][B buffer 1][halo planes]: 46.0 KiB -> three workgroups per CU (the epilogue stores from registers, round 4)
    constexpr int LDS_SLOTS = 2 * BSUB + (F32IO ? 2 : 1) * HALO_SLOTS;   // F32IO: hi planes, then lo planes (66.6 KiB: two workgroups per CU)
    constexpr int HB = 2 * BSUB;                                          // first halo slot
    static_assert(NSUB % 2 == 0, "the last sub-step must read B buffer 1");
    __shared__ f32x4 lds[LDS_SLOTS];

    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
    constexpr int NBW = PAIR ? NB / 2 : NB;                               // workgroups per tile
    int local = (WALK && a.walk) ? slot0 : slot0 / NBW;                   // tile index inside the XCD group (k_wino42_conv)
    int nb = (WALK && a.walk) ? 0 : (slot0 - local * NBW) * (PAIR ? 2 : 1);   // PAIR: the first of the workgroup's two column blocks
    {
        const int mt = xcd * a.tiles_per_xcd + local;
        if (!(mt < a.tiles_total && local < a.tiles_per_xcd)) return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c16 = lane & 15, kg = lane >> 4;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    auto uni = [](int v) { return (int)__builtin_amdgcn_readfirstlane(v); };

    // ---- per-item state ----
    int n, y0, x0;
    {
        int ty, tx;
        decode_tile(xcd * a.tiles_per_xcd + local, a.tiles_x, a.tiles_y, a.rcp_x, a.rcp_xy, n, ty, tx);
        y0 = ty * TILE_H; x0 = tx * TILE_W;
    }
    // the input tensor's pixel stride is the layer's CIN on every layer of this network (cat1 / cat2 hold exactly the channels their
    // consumer reads): a compile-time constant instead of a kernel argument held in an SGPR (the host asserts it)
    const size_t img_elems = (size_t)a.Hin * a.Win * CIN * (F32IO ? 2 : 1);   // in halfs (a.in is a half pointer)
    auto image_rsrc = [&](int img) {   // base through readfirstlane: the descriptor must live in SGPRs
        const unsigned long long p = (unsigned long long)(a.in + (size_t)img * img_elems);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), (short)0, a.Hin * a.Win * CIN * (F32IO ? 4 : 2), 0x00020000);
    };
    // halo pieces: piece s = it*256 + tid = (pixel s >> 2, k-group s & 3) -> LDS slot HB + (tid & 3) * PLANE + (tid >> 2) + 64 it
    bool abl_on = false;   // H16_ABLATE: a workgroup's first item runs in full (so LDS holds real data), the ablation applies from its second
    unsigned goff[NLOAD];
    auto halo_offsets = [&](int ty0, int tx0, int lane_id) {
#pragma unroll
        for (int it = 0; it < NLOAD; ++it) {
            const int sidx = it * THREADS + lane_id;
            const int p = F32IO ? sidx >> 3 : sidx >> 2, q = F32IO ? sidx & 7 : sidx & 3;
            const int hy = p / LW, hx = p - hy * LW;
            const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
            const bool ok = p < LPIX && (unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win;
            goff[it] = ok ? (F32IO ? (unsigned)(((gy * a.Win + gx) * CIN + q * 4) * 4) : (unsigned)(((gy * a.Win + gx) * CIN + q * 8) * 2)) : 0x7ffffff0u;
        }
    };
    // F32IO: piece (pixel tid >> 3 + 32 it, q = tid & 7) = fp32 channels 4q..4q+3 of the chunk = half h = q & 1 of the quad of k-group q >> 1
    const int hbase = F32IO ? HB + ((tid & 7) >> 1) * PLANE + ((tid & 4) ? PLANE_GAP : 0) + (tid >> 3)
                            : HB + (tid & 3) * PLANE + ((tid & 2) ? PLANE_GAP : 0) + (tid >> 2);
    const bool halo_last = F32IO ? (NLOAD - 1) * 32 + (tid >> 3) < LPIX : (NLOAD - 1) * 64 + (tid >> 2) < LPIX;          // does this thread's last piece exist (pixels 320..339 of 340)
    f32x4 pre[NLOAD];
    auto request_halo = [&](const __amdgpu_buffer_rsrc_t& rsrc, int ck, int zs) {
        if ((H16_ABLATE & 2) && abl_on) return;
#pragma unroll
        for (int it = 0; it < NLOAD; ++it) pre[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[it], zs + ck * (F32IO ? 128 : 64), 0));
    };
    auto halo_to_lds = [&]() {
        if ((H16_ABLATE & 2) && abl_on) return;
#pragma unroll
        for (int it = 0; it < NLOAD; ++it) {
            if (F32IO) {   // four fp32 channels -> their hi halfs (8 bytes into the hi planes) and the halfs of what hi leaves (the lo planes)
                if (it + 1 < NLOAD || halo_last) {
                    const f32x4 v = pre[it];
                    const f16x4 hi = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                    const f16x4 lo = {(_Float16)(v[0] - (float)hi[0]), (_Float16)(v[1] - (float)hi[1]), (_Float16)(v[2] - (float)hi[2]), (_Float16)(v[3] - (float)hi[3])};
                    f16x4* dst = reinterpret_cast<f16x4*>(lds + hbase + it * 32) + (tid & 1);
                    dst[0] = hi;
                    dst[2 * HALO_SLOTS] = lo;
                }
            } else if (it + 1 < NLOAD || halo_last) lds[hbase + it * 64] = pre[it];
        }
    };
    // B sub-chunk g = 3 ck + dx: 12 quads of 1 KiB, lane-contiguous in global memory -> LDS-DMA, three per wave, no registers
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, CIN * COUT * (MODE == 2 ? 4 : 9) * 2 * (F32IO ? 3 : 1), 0x00020000);
    const unsigned vlane = lane * 16;
    auto dma_b = [&](int wb, int g) {   // wb = byte offset of the item's column block in the packed weights
        if ((H16_ABLATE & 1) && abl_on) return;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int soff = wb + g * (BSUB * 16) + (wave_s + 4 * j) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)&lds[((g & 1) * BSUB) + (wave_s + 4 * j) * 64], 16, vlane, soff, 0, 0);
        }
    };
    // the item after (local, nb): the next column block of this tile, or column block 0 of the walker's next tile
    bool has_next;
    int n2, y02, x02, nb2, local2;
    auto decode_next = [&]() {
        n2 = n; y02 = y0; x02 = x0; local2 = local; nb2 = nb + 1;
        has_next = WALK && a.walk != 0;
        if (has_next && nb2 == NB) {
            nb2 = 0; local2 = local + a.walk;
            const int mt2 = xcd * a.tiles_per_xcd + local2;
            has_next = mt2 < a.tiles_total && local2 < a.tiles_per_xcd;
            if (has_next) {
                int ty2, tx2;
                decode_tile(mt2, a.tiles_x, a.tiles_y, a.rcp_x, a.rcp_xy, n2, ty2, tx2);
                y02 = ty2 * TILE_H; x02 = tx2 * TILE_W;
            }
        }
        if (!has_next) nb2 = nb;
        n2 = uni(n2); y02 = uni(y02); x02 = uni(x02); local2 = uni(local2); nb2 = uni(nb2); has_next = uni(has_next) != 0;
    };

#ifdef H16_TRACE   // experiment (csrc/tools/h16_trace): thread 0 sums s_memtime over the phases of ALL its items into a.pool (MODE 0 only), results unchanged
    unsigned long long* trace = reinterpret_cast<unsigned long long*>(a.pool) + (size_t)blockIdx.x * 8;
    unsigned long long tr_t0 = __builtin_readcyclecounter(), tr_last = tr_t0, tr_main = 0, tr_epi = 0, tr_bnd = 0, tr_items = 0;
    auto tr_lap = [&](unsigned long long& acc_) { const unsigned long long t = __builtin_readcyclecounter(); acc_ += t - tr_last; tr_last = t; };
#endif
    // ---- prologue of the workgroup's first item ----
    __amdgpu_buffer_rsrc_t rsrc_in = image_rsrc(n);
    halo_offsets(y0, x0, tid);
    dma_b(nb * NSUB * (BSUB * 16), 0);
    request_halo(rsrc_in, 0, 0);
    halo_to_lds();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // B sub-chunk 0 landed (this wave); past the barrier: every wave's part
    __syncthreads();
    decode_next();

    const f16x8* ldsh = reinterpret_cast<const f16x8*>(lds);
    // MFMA row i of a 16-pixel group is pixel prow(i) = {2,0,8,10}[i/4] + (i&1) + 4*((i>>1)&1): with the planes 4 (mod 16) slots
    // apart, the two k-groups that share a ds_read_b128 service group ({0-3,12-15} of one, {4-11} of the next) then cover 16
    // different slots (mod 16), and the halo stores (4 k-groups x 2 pixels per 8-lane group) are 2-way instead of 4-way.
    // Horizontal neighbours (2j, 2j+1) stay in one lane's four rows, which the pooled epilogue needs.
    const int prow = h16_prow(c16);
    const int abase = HB + kg * PLANE + ((kg & 2) ? PLANE_GAP : 0) + (2 * wave) * LW + prow;   // pixel (row 2*wave, column prow) of plane kg
    f32x4 acc[2][2][4];                                                   // [row m][pixel half pg][channel group cg]
    int wbase = 0, zs = 0;                                                // this item's B offset (bytes) and an opaque zero, renewed per item
    // one sub-step = one tap column dx of one chunk: A rows 0..3 of the wave (row r feeds output row m at dy = r - m), 12 B quads
    auto substep_on = [&](auto& A, auto first_tag, auto last_tag, int g, int dx, int req_ck) {   // A: the accumulator set the sub-step adds to
        constexpr bool FIRST = decltype(first_tag)::value;    // first sub-step of an item: the accumulators start from zero
        constexpr bool LAST = decltype(last_tag)::value;      // last one: fetch the NEXT item's first B sub-chunk and halo chunk instead
        const f16x8* bq = ldsh + (g & 1) * BSUB + lane;
        const f16x8* aq = ldsh + abase + dx;
        if (!LAST) {
            dma_b(wbase, g + 1);
            // the NEXT chunk's halo is requested in a chunk's MIDDLE sub-step, BEHIND that sub-step's B DMA: the seam after it then
            // waits for all but the six youngest operations (= the B DMA only), and the halo loads have two sub-steps to land instead
            // of one (round 4; rounds 2-3 requested them in front of the last sub-step)
            if (req_ck >= 0) request_halo(rsrc_in, req_ck, zs);
        } else if (has_next) {                                  // B first: the halo loads behind it in the queue then vouch for it
            // (workgroup-uniform, a scalar branch: one item per workgroup — the CIN = 256 layers — and a walker's last item have no next item;
            // rounds 2-3 fetched 12 KiB of weights and a 21 KiB halo chunk for nobody there: ADVICE r3)
            dma_b(nb2 * NSUB * (BSUB * 16) + zs, 0);
            halo_offsets(y02, x02, tid + zs);
            request_halo(image_rsrc(n2), 0, zs);
        }
        f16x8 ar[4][2], bf[3][4];
        if ((H16_ABLATE & 8) && !FIRST && abl_on) {   // no LDS reads: the MFMAs run on whatever the (opaque) registers hold
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int pg = 0; pg < 2; ++pg) asm volatile("" : "=v"(ar[r][pg]));
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) asm volatile("" : "=v"(bf[dy][cg]));
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int pg = 0; pg < 2; ++pg)
#pragma unroll
                        for (int cg = 0; cg < 4; ++cg) A[m][pg][cg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ar[m + dy][pg], bf[dy][cg], A[m][pg][cg], 0, 0, 0);
            return;
        }
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) bf[0][cg] = bq[cg * 64];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int pg = 0; pg < 2; ++pg) ar[r][pg] = aq[r * LW + pg * 16];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            if (dy < 2) {
#pragma unroll
                for (int pg = 0; pg < 2; ++pg) ar[dy + 2][pg] = aq[(dy + 2) * LW + pg * 16];
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) bf[dy + 1][cg] = bq[((dy + 1) * 4 + cg) * 64];
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int pg = 0; pg < 2; ++pg)
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg) {
                        if (FIRST && dy == 0) {
                            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                            A[m][pg][cg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ar[m + dy][pg], bf[dy][cg], zero, 0, 0, 0);
                        } else {
                            A[m][pg][cg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ar[m + dy][pg], bf[dy][cg], A[m][pg][cg], 0, 0, 0);
                        }
                    }
        }
    };
    auto substep = [&](auto first_tag, auto last_tag, int g, int dx, int req_ck = -1) { substep_on(acc, first_tag, last_tag, g, dx, req_ck); };
    // MODE 2 (ConvTranspose2d(k=2, s=2) in the split-operand form): a column block is (tap, 64 output channels), an output pixel (2y + kh, 2x + kw) a plain 1x1 product of
    // input pixel (y, x): per 32-channel chunk and block ONE sub-step of 48 MFMAs — the wave's own two rows (the halo tile's centre) from the hi planes against hi_w and
    // lo_w, from the lo planes against hi_w; the 12-KiB weight sub-chunk holds those three pieces where a 3x3 sub-chunk holds three tap rows.
    auto substep_t = [&](auto& A, auto last_tag, int g, int req_ck) {
        constexpr bool LAST = decltype(last_tag)::value;
        const f16x8* bq = ldsh + (g & 1) * BSUB + lane;
        const f16x8* aq = ldsh + abase + LW + 1;                    // centre of the halo tile: row + 1, column + 1
        if (!LAST) {
            dma_b(wbase, g + 1);
            if (req_ck >= 0) request_halo(rsrc_in, req_ck, zs);
        }
        f16x8 ax[2][2], bf[2][4];                                   // weights one piece ahead; the pixel fragments of the plane set in use (hi for pieces 0-1, lo for piece 2)
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) bf[0][cg] = bq[cg * 64];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int pg = 0; pg < 2; ++pg) ax[m][pg] = aq[m * LW + pg * 16];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            if (p < 2) {
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) bf[(p + 1) & 1][cg] = bq[((p + 1) * 4 + cg) * 64];
            }
            if (p == 2) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int pg = 0; pg < 2; ++pg) ax[m][pg] = aq[m * LW + pg * 16 + HALO_SLOTS];
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int pg = 0; pg < 2; ++pg)
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg) A[m][pg][cg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ax[m][pg], bf[p & 1][cg], A[m][pg][cg], 0, 0, 0);
        }
    };
    auto seam = [&]() {   // between sub-steps of one chunk: the next B sub-chunk has landed in every wave
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    auto seam_keep_halo = [&]() {   // the same with the NLOAD halo loads issued behind the B DMA still in flight
        static_assert(NLOAD == (F32IO ? 11 : 6), "vmcnt immediate below");
        if (H16_ABLATE & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (F32IO) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __syncthreads();
    };
    auto chunk_seam = [&]() {   // between chunks: the halo tile is replaced as well
        __syncthreads();
        halo_to_lds();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    static_assert(NCHUNK >= 2, "first and last chunk are separate instantiations");
    using T = std::true_type;
    using F = std::false_type;
    for (;;) {   // ---- one item per iteration ----
        asm volatile("s_mov_b32 %0, 0" : "=s"(zs));
        wbase = nb * NSUB * (BSUB * 16) + zs;
        const int cobase = nb * NTILE;
        const f32x4 bias_v = *reinterpret_cast<const f32x4*>(a.bias + (MODE == 2 ? 0 : cobase) + 4 * c16 + zs);   // column c16 of group cg = channel 4 c16 + cg (MODE 2: its epilogue loads its own)
#ifdef H16_TRACE
        tr_lap(tr_bnd);     // prologue of the first item / boundary of the later ones
#endif
        if constexpr (MODE == 2) {
            // two column blocks per workgroup (PAIR): per chunk one sub-step into `acc` (block nb) and one into `acc2` (block nb + 1); sub-step s reads B buffer s & 1 and
            // the DMA's base offset picks the block whose sub-chunk comes next
            const int wbA = wbase, wbB = wbase + NSUB * (BSUB * 16);
            auto aim = [&](int wb, int packed_next, int s_next) { wbase = wb + (packed_next - s_next) * (BSUB * 16); };
            f32x4 acc2[2][2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int pg = 0; pg < 2; ++pg)
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg) { acc[m][pg][cg] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[m][pg][cg] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            for (int ck = 0; ck + 1 < NCHUNK; ++ck) {
                aim(wbB, ck, 2 * ck + 1);
                substep_t(acc, F{}, 2 * ck, ck + 1); seam_keep_halo();
                aim(wbA, ck + 1, 2 * ck + 2);
                substep_t(acc2, F{}, 2 * ck + 1, -1);
                chunk_seam();
            }
            aim(wbB, NCHUNK - 1, 2 * NCHUNK - 1);
            substep_t(acc, F{}, 2 * NCHUNK - 2, -1); seam();
            wbase = wbA;
            substep_t(acc2, T{}, 2 * NCHUNK - 1, -1);
            int lane_p;
            asm volatile("v_mov_b32 %0, %1" : "=v"(lane_p) : "v"(tid & 63));
            h16_epilogue_t32<COUT>(a, acc2, n, y0, x0, wave, lane_p, nb + 1);
        } else if constexpr (PAIR) {
            // sub-step s of the sequence reads B buffer s & 1; which packed sub-chunk of which column block the NEXT one needs goes through `wbase` (dma_b fetches wbase + (s + 1) sub-chunks)
            const int wbA = wbase, wbB = wbase + NSUB * (BSUB * 16);
            auto aim = [&](int wb, int packed_next, int s_next) { wbase = wb + (packed_next - s_next) * (BSUB * 16); };
            auto dxof = [&](int j) { return j >= 6 ? j - 6 + HALO_SLOTS : j >= 3 ? j - 3 : j; };
            f32x4 acc2[2][2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int pg = 0; pg < 2; ++pg)
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg) { acc[m][pg][cg] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[m][pg][cg] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            for (int ck = 0; ck < NCHUNK; ++ck) {
                for (int j = 0; j < 9; ++j) {            // column block A on this chunk's planes
                    const int q = 18 * ck + j;
                    if (j < 8) aim(wbA, 9 * ck + j + 1, q + 1); else aim(wbB, 9 * ck, q + 1);
                    substep_on(acc, F{}, F{}, q, dxof(j), -1);
                    seam();
                }
                const int jend = ck + 1 < NCHUNK ? 9 : 8;
                for (int j = 0; j < jend; ++j) {         // column block B; the next chunk is requested in its sub-step 7
                    const int q = 18 * ck + 9 + j;
                    if (j < 8) aim(wbB, 9 * ck + j + 1, q + 1); else aim(wbA, 9 * (ck + 1), q + 1);
                    const int req = (j == 7 && ck + 1 < NCHUNK) ? ck + 1 : -1;
                    substep_on(acc2, F{}, F{}, q, dxof(j), req);
                    if (j == 8) chunk_seam(); else if (req >= 0) seam_keep_halo(); else seam();
                }
            }
            wbase = wbA;
            substep_on(acc2, F{}, T{}, 18 * NCHUNK - 1, 2 + HALO_SLOTS, -1);
            int lane_p;
            asm volatile("v_mov_b32 %0, %1" : "=v"(lane_p) : "v"(tid & 63));
            h16_epilogue_f32<COUT, MODE>(a, acc2, *reinterpret_cast<const f32x4*>(a.bias + cobase + NTILE + 4 * c16 + zs), n, y0, x0, wave, lane_p, cobase + NTILE);
        } else if constexpr (F32IO) {   // nine sub-steps per 32-channel chunk — tap columns 0..2 on the hi planes against hi_w, again against lo_w, then on the
            // lo planes against hi_w; the packed weights hold the sub-chunks in this order, so sub-step q reads sub-chunk q.  The next chunk is requested in sub-step 7.
            static_assert(!WALK, "one item per workgroup");
            substep(T{}, F{}, 0, 0); seam();
            for (int q = 1; q < NSUB - 1; ++q) {
                const int ck = q / 9, j = q - 9 * ck;
                const int dx = (j >= 6 ? j - 6 + HALO_SLOTS : j >= 3 ? j - 3 : j);          // + HALO_SLOTS: the lo planes
                const int req = (j == 7 && ck + 1 < NCHUNK) ? ck + 1 : -1;
                substep(F{}, F{}, q, dx, req);
                if (j == 8) chunk_seam(); else if (req >= 0) seam_keep_halo(); else seam();
            }
            substep(F{}, T{}, NSUB - 1, 2 + HALO_SLOTS);
        } else {
#ifdef H16_SPLIT_IN
        {   // experiment (csrc/tools/split_proto, -DCID_EXPERIMENTS): split-operand fp32 convolution.  A pixel holds [hi | lo] halfs of C = CIN / 2 fp32 channels, the packed
            // weights [hi_w | lo_w] as 2 C input channels.  Halo chunk ck < NCH (hi_x) runs SIX sub-steps — its three tap columns against hi_w, then against lo_w, from the same
            // fragments — and chunk NCH + c (lo_x) three against hi_w: 9 NCH sub-steps.  Sub-step s reads B buffer s & 1 and requests the sub-chunk of step s + 1 into the other one;
            // which packed sub-chunk that is goes through `wbase` (dma_b fetches wbase + (s + 1) sub-chunks).
            static_assert(!WALK && !ZOUT && NCHUNK % 2 == 0, "prototype: one item per workgroup, plain epilogue");
            constexpr int NCH = NCHUNK / 2, NSEQ = 9 * NCH;
            const int wbase0 = wbase;
            auto packed_of = [&](int q) { if (q < 6 * NCH) { const int c = q / 6, j = q - 6 * c; return j < 3 ? 3 * c + j : 3 * (NCH + c) + j - 3; } return q - 6 * NCH; };
            auto aim = [&](int q) { wbase = wbase0 + (packed_of(q + 1) - (q + 1)) * (BSUB * 16); };
            aim(0); substep(T{}, F{}, 0, 0); seam();
            for (int q = 1; q < NSEQ - 1; ++q) {
                int ck, j, nsub;
                if (q < 6 * NCH) { ck = q / 6; j = q - 6 * ck; nsub = 6; } else { const int t = q - 6 * NCH; ck = NCH + t / 3; j = t - 3 * (t / 3); nsub = 3; }
                const int dx = j >= 3 ? j - 3 : j;
                const int req = (j == nsub - 2 && ck + 1 < NCHUNK) ? ck + 1 : -1;
                aim(q);
                substep(F{}, F{}, q, dx, req);
                if (j == nsub - 1) chunk_seam(); else if (req >= 0) seam_keep_halo(); else seam();
            }
            wbase = wbase0;
            substep(F{}, T{}, NSEQ - 1, 2);
        }
#else
        substep(T{}, F{}, 0, 0); seam();
        substep(F{}, F{}, 1, 1, 1); seam_keep_halo();
        substep(F{}, F{}, 2, 2); chunk_seam();
        for (int ck = 1; ck + 1 < NCHUNK; ++ck) {
            substep(F{}, F{}, 3 * ck, 0); seam();
            substep(F{}, F{}, 3 * ck + 1, 1, ck + 1); seam_keep_halo();
            substep(F{}, F{}, 3 * ck + 2, 2); chunk_seam();
        }
        substep(F{}, F{}, NSUB - 3, 0); seam();
        substep(F{}, F{}, NSUB - 2, 1); seam();
        substep(F{}, T{}, NSUB - 1, 2);
#endif
        }

#ifdef H16_TRACE
        asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[1][1][3]));
        tr_lap(tr_main);
#endif
        {   // no barrier: the epilogue touches no LDS (round 4), so a wave stores while its siblings finish their MFMAs
            int lane_e;    // opaque copy of the lane id: keeps the epilogue's address arithmetic out of the item loop's registers
            asm volatile("v_mov_b32 %0, %1" : "=v"(lane_e) : "v"(tid & 63));
            if constexpr (ZOUT && F32IO) {
                __syncthreads();   // every wave has read its last fragments: the whole LDS becomes the staging area (4 x 16 KiB)
                static_assert(4 * 1024 <= LDS_SLOTS, "hi and lo staging of four waves");
                h16_zout_epilogue_f32(a, lds, acc, bias_v, n, y0, x0, wave, lane_e);
            } else if constexpr (ZOUT) {
                __syncthreads();   // every wave has read its last fragments: B buffer 1 and the halo planes become the staging area (32 KiB)
                static_assert(4 * 512 <= LDS_SLOTS - BSUB, "z staging must fit behind B buffer 0 (the next item's first B sub-chunk lands there)");
                h16_zout_epilogue(a, lds + BSUB, acc, bias_v, n, y0, x0, wave, lane_e);
            } else if constexpr (F32IO && MODE == 2) {
                h16_epilogue_t32<COUT>(a, acc, n, y0, x0, wave, lane_e, nb);
            } else if constexpr (F32IO) {
                h16_epilogue_f32<COUT, MODE>(a, acc, bias_v, n, y0, x0, wave, lane_e, cobase);
            } else {
                h16_epilogue<COUT, MODE>(a, acc, bias_v, n, y0, x0, wave, lane_e, cobase);
            }
        }
#ifdef H16_TRACE
        tr_lap(tr_epi);
        ++tr_items;
#endif
        if (!has_next) break;                                  // workgroup-uniform
        // ---- item boundary: the prefetched item becomes the current one ----
        n = n2; y0 = y02; x0 = x02; local = local2; nb = nb2;
        if (H16_ABLATE) abl_on = true;
        rsrc_in = image_rsrc(n);
        __syncthreads();   // every wave has read its last fragments of this item: the halo planes may be written again
        halo_to_lds();     // its loads are younger than the B DMA issued with them: their arrival vouches for B buffer 0 as well
        decode_next();
        __syncthreads();
    }
#ifdef H16_TRACE
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        trace[0] = tr_t0; trace[1] = tr_main; trace[2] = tr_epi; trace[3] = tr_bnd; trace[5] = tr_items;
        trace[4] = __builtin_readcyclecounter();
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Head of the fp16-storage path: down1[0] = Conv2d(3, 64, 3, padding=1) + ReLU (app.py:46-47) on v_mfma_f32_16x16x32_f16.
// K = 3 channels x 9 taps = 27 fits ONE K = 32 step (k = 3 tap + c, rows 27..31 of the packed weights are zero), so a 16-pixel x
// 16-channel tile is a single MFMA; the fp32-MFMA head needs 14 steps of 32x32x2 per 32 x 32 and was the one launch of the fp16 forward
// still priced in fp32 MFMA time (0.365 ms for 512 images against 0.21 for its 1.07 GB of stores).  The image is rounded to half
// here (as every later activation of this path is); accumulation stays fp32.
//   * input: planar [3][10][36] halfs in LDS, filled like k_conv_head's (all loads of the NEXT tile in flight under this tile's work);
//   * A: lane (pixel h16_prow(lane & 15), k-group kg) gathers its eight k = 8kg..8kg+7 with eight 2-byte LDS reads at lane-constant offsets;
//   * B: [4 channel groups][lane][8] halfs, 16 registers for the whole kernel (host: pack_head_h16);
//   * stores: h16_store_row (bias, ReLU, 8-byte stores straight from the accumulators: whole 128-byte lines), one output row at a time.
template <bool IN_U8>
__global__ void __launch_bounds__(THREADS, 4) k_conv_head_h16(const HeadArgs a) {
    // rows of 40 halfs, planes of 406: with these strides the eight 2-byte gathers of an A fragment (lane = (pixel, k-group), k = 3 tap + c) put the
    // two k-groups of each 32-lane half on disjoint banks for every tap, row and pixel group (exhaustive search over row / plane paddings, round 4;
    // rows of 36 and planes of 360 had 0.56 extra LDS cycles per gather: 46 % SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE in r3, 33 % after the staging went)
    constexpr int LW = 40, LH = TILE_H + 2, PLANE = 406;
    static_assert(PLANE >= LW * LH, "plane holds the 10-row patch");
    constexpr int IMG_BYTES = (3 * PLANE * 2 + 15) / 16 * 16;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[IMG_BYTES];   // the planar half image (2.1 KB)
    _Float16* const img_h = reinterpret_cast<_Float16*>(lds_raw);
    int grp, nb;
    if (!decode_block(a.groups_total, a.groups_per_xcd, 1, grp, nb)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c16 = lane & 15, kg = lane >> 4;

    constexpr int NS = 3 * LH * 34, NIT = (NS + THREADS - 1) / THREADS;
    const size_t img = (size_t)a.src.H * a.src.W * 3;   // elements per image in either input format
    // element s = it*256 + tid of the [3][LH][34] halo patch: plane, row, column are recomputed per tile (a few integer operations
    // against 12 registers held across the MFMAs: this kernel sits at the 128-register step)
    auto patch = [&](int it, int& c, int& hy, int& hx) {
        const int s = it * THREADS + tid;
        c = s / (LH * 34);
        const int rem = s - c * (LH * 34);
        hy = rem / 34; hx = rem - hy * 34;
        return s < NS;
    };
    float staged[NIT];
    auto request_tile = [&](int tile, int& tn, int& ty0, int& tx0) {
        int ty, tx;
        decode_tile(tile, a.tiles_x, a.tiles_y, a.rcp_x, a.rcp_xy, tn, ty, tx);
        ty0 = ty * TILE_H; tx0 = tx * TILE_W;
        const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
            IN_U8 ? (void*)(static_cast<const unsigned char*>(a.in) + (size_t)tn * img) : (void*)(static_cast<const float*>(a.in) + (size_t)tn * img),
            (short)0, (int)(IN_U8 ? img : img * 4), 0x00020000);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int c, hy, hx;
            const bool in_patch = patch(it, c, hy, hx);
            const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;                      // network-input coordinates
            const bool net = in_patch && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            const int sy = gy - a.src.top, sx = gx - a.src.left;                 // the caller's image (conv_kernels.h, Window)
            const bool ok = net && (unsigned)sy < (unsigned)a.src.H && (unsigned)sx < (unsigned)a.src.W;
            const unsigned goff = !ok ? 0x7ffffff0u : IN_U8 ? (unsigned)((sy * a.src.W + sx) * 3 + c) : (unsigned)(((c * a.src.H + sy) * a.src.W + sx) * 4);
            // the image; the black band the server pads with (-1.0 once normalised); the convolution's zero padding (0 in the
            // NORMALISED tensor, not (0/255 - 0.5)/0.5)
            const float fill = net ? -1.f : 0.f;
            if (IN_U8) {
                const float t = (float)__builtin_amdgcn_raw_buffer_load_b8(rsrc_in, goff, 0, 0);
                staged[it] = ok ? (t / 255.0f - 0.5f) / 0.5f : fill;
            } else {
                const float t = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_in, goff, 0, 0));
                staged[it] = ok ? t : fill;
            }
        }
    };
    const int tile0 = grp * a.tiles_per_wg;
    const int ntile = min(a.tiles_per_wg, a.tiles_total - tile0);   // >= 1 (decode_block), workgroup-uniform
    int n, y0, x0;
    request_tile(tile0, n, y0, x0);
    f16x8 bfr[4];
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) bfr[cg] = reinterpret_cast<const f16x8*>(a.w)[cg * 64 + lane];
    const f32x4 bias_v = *reinterpret_cast<const f32x4*>(a.bias + 4 * c16);   // column c16 of group cg = channel 4 c16 + cg
    int koff[8];   // k = 8 kg + e = 3 tap + c -> offset of (plane c, tap row, tap column); k >= 27 meets zero weights, any finite element will do
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = min(8 * kg + e, 26), tap = k / 3, c = k - 3 * tap;
        koff[e] = c * PLANE + (tap / 3) * LW + (tap % 3);
    }
    const int pbase = (2 * wave) * LW + h16_prow(c16);
#pragma unroll 1
    for (int t = 0; t < ntile; ++t) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int c, hy, hx;
            if (patch(it, c, hy, hx)) img_h[c * PLANE + hy * LW + hx] = (_Float16)staged[it];
        }
        __syncthreads();
        int nn = n, ny0 = y0, nx0 = x0;
        if (t + 1 < ntile) request_tile(tile0 + t + 1, nn, ny0, nx0);   // in flight under this tile's MFMAs and stores
        const bool full = y0 + TILE_H <= a.H && x0 + TILE_W <= a.W;
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(static_cast<_Float16*>(a.out) + (size_t)n * a.H * a.W * 64), (short)0,
                                                                            a.H * a.W * 128, 0x00020000);   // image n of t0: 128 B per pixel
#pragma unroll 1
        for (int m = 0; m < 2; ++m) {   // a row at a time: 32 accumulator registers, this kernel sits at the 128-register step
            f32x4 acc[2][4];
#pragma unroll
            for (int pg = 0; pg < 2; ++pg) {
                f16x8 af;
#pragma unroll
                for (int e = 0; e < 8; ++e) af[e] = img_h[pbase + m * LW + pg * 16 + koff[e]];
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) {
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    acc[pg][cg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bfr[cg], zero, 0, 0, 0);
                }
            }
            const int y = __builtin_amdgcn_readfirstlane(y0 + 2 * wave + m);
            h16_store_row<32>(lane, [&](int pg, int cg, int r) { return fmaxf(acc[pg][cg][r] + bias_v[cg], 0.f); }, ro, (unsigned)(((y * a.W + x0) * 64) * 2), 64, a.W - x0, y < a.H, full);
        }
        n = nn; y0 = ny0; x0 = nx0;
        if (t + 1 < ntile) __syncthreads();   // every wave is done reading the planes before the next tile overwrites them
    }
}

// ---------------------------------------------------------------------------------------------
// Tail of the fp16-storage path: upconv1[2] + tanh (app.py:77,101,103) on a half NHWC input.  Same algorithm as
// k_conv_tail (z = x . W for every halo pixel as a [352 x 64] x [64 x 32] product, then the nine shifted sums), but the
// product runs on v_mfma_f32_32x32x16_f16 straight from the half tile: 4 MFMAs per 32 pixels instead of 32, no
// half -> float conversion on the way into LDS, and all 64 channels of the tile fit the LDS that held one 32-channel fp32
// chunk, so there is one load phase per tile.  Weights: half, [4 k-steps][lane = 32*h + col][8]  (cid_api.hip, TAIL).
// The fp32 tail fed from half spent 0.63 ms at B=512 (twice the images per byte: its fp32 product alone is 0.3 ms).
template <bool OUT_U8>
__global__ void __launch_bounds__(THREADS, 2) k_conv_tail_h(const TailArgs a) {
    constexpr int LW = TILE_W + 2, LH = TILE_H + 2, LPIX = LW * LH;   // 340 halo pixels
    constexpr int MT = (LPIX + 31) / 32, LP = MT * 32;                // 11 M tiles, 352 rows
    constexpr int NSLOT = LPIX * 8, NLOAD = (NSLOT + THREADS - 1) / THREADS;   // 16-byte pieces: 8 per pixel (64 halfs)
    constexpr int ZS = 33;
    __shared__ f32x4 lds[LP * PSLOTS];
    static_assert(LP * ZS * sizeof(float) <= sizeof(lds), "z must fit where x was");
    int grp, nb;
    if (!decode_block(a.groups_total, a.groups_per_xcd, 1, grp, nb)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, h = lane >> 5;

    f16x8 wb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) wb[s] = reinterpret_cast<const f16x8*>(a.w)[s * 64 + lane];
    float bias_v[3];
#pragma unroll
    for (int co = 0; co < 3; ++co) bias_v[co] = a.bias[co];

    const size_t img_elems = (size_t)a.H * a.W * 64;
    int n, y0, x0;
    __amdgpu_buffer_rsrc_t rsrc_in;
    unsigned goff[NLOAD];
    f32x4 stage[NLOAD];
    auto request_tile = [&](int tile, int& tn, int& ty0, int& tx0) {
        int ty, tx;
        decode_tile(tile, a.tiles_x, a.tiles_y, a.rcp_x, a.rcp_xy, tn, ty, tx);
        ty0 = ty * TILE_H; tx0 = tx * TILE_W;
        rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)(static_cast<const _Float16*>(a.in) + (size_t)tn * img_elems), (short)0,
                                                    (int)(img_elems * 2), 0x00020000);
#pragma unroll
        for (int it = 0; it < NLOAD; ++it) {
            const int s = it * THREADS + tid;
            const int p = s >> 3, c = s & 7;
            const int hy = p / LW, hx = p - hy * LW;
            const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
            const bool ok = (s < NSLOT) && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            goff[it] = ok ? (unsigned)(((gy * a.W + gx) * 64 + c * 8) * 2) : 0x7ffffff0u;
        }
#pragma unroll
        for (int it = 0; it < NLOAD; ++it) stage[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, goff[it], 0, 0));
    };

    const int tile0 = grp * a.tiles_per_wg;
    const int ntile = min(a.tiles_per_wg, a.tiles_total - tile0);   // >= 1, workgroup-uniform
    request_tile(tile0, n, y0, x0);
    for (int t = 0; t < ntile; ++t) {
#pragma unroll
        for (int it = 0; it < NLOAD; ++it) {
            const int s = it * THREADS + tid;
            if (s < NSLOT) lds[lds_slot(s >> 3, s & 7)] = stage[it];
        }
        int nn = n, ny0 = y0, nx0 = x0;
        if (t + 1 < ntile) request_tile(tile0 + t + 1, nn, ny0, nx0);   // in flight under the product and the epilogue
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        f32x16 acc[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int mtile = wave + 4 * q;          // wave-uniform
            if (mtile < MT) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    // rows 340..351 of the last tile read never-written LDS: they only reach z rows nobody gathers
                    const f16x8 av = __builtin_bit_cast(f16x8, lds[lds_slot(mtile * 32 + i, 2 * s + h)]);
                    if (s == 0) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, wb[s], zero, 0, 0, 0);
                    } else {
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, wb[s], acc[q], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();   // every wave is done reading x: the LDS becomes z[352][33]
        float* zl = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int mtile = wave + 4 * q;
            if (mtile < MT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) zl[(mtile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * ZS + i] = acc[q][r];
            }
        }
        __syncthreads();
        const int row = tid >> 5, col = tid & 31;
        const int pb = row * LW + col;
        float o[3] = {bias_v[0], bias_v[1], bias_v[2]};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float* zp = zl + (pb + (tap / 3) * LW + (tap % 3)) * ZS + tap * 3;
#pragma unroll
            for (int co = 0; co < 3; ++co) o[co] += zp[co];
        }
        const int y = y0 + row, x = x0 + col;
        const int cy = y - a.crop.top, cx = x - a.crop.left;       // the caller's tensor (conv_kernels.h, Window)
        if (y < a.H && x < a.W && (unsigned)cy < (unsigned)a.crop.H && (unsigned)cx < (unsigned)a.crop.W) {
            if (OUT_U8) {
                unsigned char* op = static_cast<unsigned char*>(a.out) + ((size_t)(n * a.crop.H + cy) * a.crop.W + cx) * 3;
#pragma unroll
                for (int co = 0; co < 3; ++co) {
                    const float v = fminf(fmaxf(tanhf(o[co]) * 0.5f + 0.5f, 0.f), 1.f);
                    op[co] = (unsigned char)(v * 255.0f);
                }
            } else {
                const size_t plane = (size_t)a.crop.H * a.crop.W;
                float* op = static_cast<float*>(a.out) + (size_t)n * 3 * plane + (size_t)cy * a.crop.W + cx;
                op[0] = tanhf(o[0]);
                op[plane] = tanhf(o[1]);
                op[2 * plane] = tanhf(o[2]);
            }
        }
        __syncthreads();   // everyone is done gathering z before the next tile's pieces overwrite it
        n = nn; y0 = ny0; x0 = nx0;
    }
}

// ---------------------------------------------------------------------------------------------
// Tail of the fp16-storage path, fused form (round 4; default): k_conv3x3_h16<128, 64, 0, ZOUT> has contracted the 64 channels,
// z[n][group][y][x][4 halfs] with row 3 tap + co = 4 group + slot.  What is left of upconv1[2] + tanh (app.py:77,103) is the nine-tap shifted sum, as in k_conv_tail_z:
//     out[n][co][y][x] = tanh(bias[co] + sum_{ty,tx} z[n][3 (3 ty + tx) + co][y+ty-1][x+tx-1])        (zero outside the image)
// One thread per pixel, THIRTEEN 8-byte loads (a tap's three rows lie in one group or straddle two: taps 0, 3, 4, 7, 8 one load, the others two;
// consecutive lanes = consecutive pixels = 512 contiguous bytes per wave instruction; out-of-image taps carry an out-of-range per-lane offset),
// fp32 sums in tap order, three tanhf.  HBM-bound on 68 B per pixel (56 z + 12 out) where the 9-group form moved 84 and k_conv_tail_h 140 with a halo on top.
template <bool OUT_U8>
__global__ void __launch_bounds__(THREADS) k_conv_tail_zh(const TailZArgs a) {
    const unsigned b = blockIdx.x;
    const unsigned n = a.rcp_blocks ? __umulhi(b, a.rcp_blocks) : b;
    const unsigned p = (b - n * a.blocks_per_image) * THREADS + threadIdx.x;   // pixel index inside the image
    const size_t plane = (size_t)a.H * a.W;
    const unsigned y = a.rcp_w ? __umulhi(p, a.rcp_w) : p, x = p - y * a.W;
    const bool inside = p < plane;
    // descriptor over the image's 7 group planes of 8 bytes per pixel (<= 235 MB: H*W < 4,194,303, cid_api.hip shape_error)
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const _Float16*>(a.z) + (size_t)n * (4 * Z_GROUPS) * plane), (short)0,
                                                                        (int)(plane * (8 * Z_GROUPS)), 0x00020000);
    float o[3] = {a.bias[0], a.bias[1], a.bias[2]};
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
            const int yy = (int)y + ty - 1, xx = (int)x + tx - 1;
            const bool ok = inside && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
            const unsigned off = ok ? (unsigned)((yy * a.W + xx) * 8) : 0x7ffffff0u;   // the mask lives in the per-lane offset; the group plane is the scalar one
            const int r0 = 3 * (3 * ty + tx), g0 = r0 >> 2, g1 = (r0 + 2) >> 2;       // compile-time after unrolling
            const f16x4 v0 = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rz, off, (int)(g0 * plane * 8), 0));
            f16x4 v1 = v0;
            if (g1 != g0) v1 = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rz, off, (int)(g1 * plane * 8), 0));
#pragma unroll
            for (int co = 0; co < 3; ++co) o[co] += (float)(((r0 + co) >> 2) == g0 ? v0 : v1)[(r0 + co) & 3];
        }
    const int cy = (int)y - a.crop.top, cx = (int)x - a.crop.left;   // the caller's tensor
    if (!inside || (unsigned)cy >= (unsigned)a.crop.H || (unsigned)cx >= (unsigned)a.crop.W) return;
    const size_t oplane = (size_t)a.crop.H * a.crop.W, op_idx = (size_t)cy * a.crop.W + cx;
    if (OUT_U8) {
        unsigned char* op = static_cast<unsigned char*>(a.out) + ((size_t)n * oplane + op_idx) * 3;
#pragma unroll
        for (int co = 0; co < 3; ++co) {
            const float v = fminf(fmaxf(tanhf(o[co]) * 0.5f + 0.5f, 0.f), 1.f);
            op[co] = (unsigned char)(v * 255.0f);
        }
    } else {
        float* op = static_cast<float*>(a.out) + (size_t)n * 3 * oplane + op_idx;
        op[0] = tanhf(o[0]);
        op[oplane] = tanhf(o[1]);
        op[2 * oplane] = tanhf(o[2]);
    }
}

}  // namespace cid
