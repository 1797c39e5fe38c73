// wino64_kernels.h — 3x3 convolution (+bias, +ReLU, +optional 2x2 max-pool) as Winograd F(2x2,3x3) on the exact-f32
// matrix instruction v_mfma_f32_32x32x2_f32 (gfx950): 32 tiles x 64 output channels per workgroup, one row of the
// transformed tile per wave.
//
// Computes the same function as the reference's nn.Conv2d(k=3, p=1) + nn.ReLU (+ nn.MaxPool2d(2,2)) stages
// (backend/app.py:43-77) with 16 multiplies per 2x2 output tile and (ci, co) pair instead of 36: for each of the 16
// positions xi = (a, b) of the transformed 4x4 tile,
//       M_xi[tile][co] = sum_ci V_xi[tile][ci] * U_xi[ci][co],     V = B^T d B,  U = G g G^T,     Y(2x2) = A^T M A.
// U is computed on the host at load time (cid_api.hip, in double, rounded once).  V is never stored: each lane rebuilds
// the four V values of a row `a` from eight 16-byte LDS reads of the raw input tile (adds only).
//
//   * wave w = row a = w of B^T d B x 4 positions b x TWO 32-channel column blocks (8 accumulator tiles = 128 VGPRs):
//     every A operand feeds two MFMAs, i.e. 1 VALU instruction and 1/4 ds_read_b128 per MFMA.  VALU instructions beside
//     the MFMA stream cost matrix-pipe time beyond ~1 per MFMA (tools/mix_bench); this decomposition sits at 1.
//   * K is walked in 16-channel chunks through a double-buffered raw halo tile in LDS (pixel = 4 data slots + 1 pad slot
//     of 16 B; inside a tile row even columns first, then odd columns: conflict-free ds_read_b128); the next chunk is
//     written by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPRs, no ds_write; out-of-image and pad slots carry an
//     out-of-range offset and the buffer range check writes zeros).
//   A unit = row a, 8 input channels, both column blocks = 32 MFMAs; a 16-channel chunk = 2 units.
//   * B: 8 quads (column block nt, k-step e) per unit, each refilled for the NEXT unit right after its four MFMAs.
//   * raw tile double-buffered; the DMA of chunk c+2 is issued at the start of unit 1 of chunk c (its buffer was last
//     read during unit 0) and must have landed by the one barrier of chunk c+1, at the end of its unit 0: two units
//     (>= 4096 cycles) of slack.
//   * epilogue: the four waves exchange their column-transformed rows m'[a] through LDS; wave w then finishes column
//     block w>>1, tiles 16*(w&1) .. +16: Y[0] = (m'0 + m'1) + m'2, Y[1] = m'1 - (m'2 + m'3), bias, ReLU, optional 2x2 max-pool, 16-byte stores via LDS staging.
#pragma once
#include "conv_kernels.h"

namespace cid {

constexpr int WK = 16;        // channels per chunk
constexpr int WPS = 5;        // LDS slots (16 B) per pixel: 4 data + 1 pad
constexpr int WS32 = 36;      // staging row stride (floats) for 32-channel slabs

// Host: the slot table of k_wino64_conv<.., TC>: LDS slot s (16 B) of the raw halo tile -> packed (row, column, group).
// Must mirror the kernel's LDS order: pixel = s/5 (4 data slots + 1 pad), rows of LWS pixels, even columns then odd.
// BTR = tile rows per workgroup = 32/TC.
inline int wino_slot_table(int TC, int BTR, unsigned* out /* may be null */) {
    const int LW = 2 * TC + 2, LH = 2 * BTR + 2, LWS = (TC == 16) ? 40 : LW, HWD = LWS / 2;
    // padded to 4 * RW rounds (RW = rounds per wave), so that every wave reads RW entries unconditionally: a guarded
    // load compiles to load -> wait -> next load, i.e. RW serialised memory latencies in every workgroup's prologue
    const int LPIX = LWS * LH, NROUND = (LPIX * WPS + 63) / 64, RW = (NROUND + 3) / 4;
    if (out)
        for (int s = 0; s < 4 * RW * 64; ++s) {
            const int p = s / WPS, c = s - p * WPS;
            const int hy = p / LWS, rem = p - hy * LWS, plane = rem / HWD, hx = 2 * (rem - plane * HWD) + plane;
            out[s] = (s < NROUND * 64 && c < 4 && p < LPIX && hx < LW) ? ((unsigned)hy << 20 | (unsigned)hx << 8 | (unsigned)c) : ~0u;
        }
    return 4 * RW * 64;
}

struct WinoArgs {
    const float* in;    // NHWC [N, Hin, Win, in_ps]
    const float* u;     // packed U: [nb][chunk][round][a][nt][e][lane][b]  (cid_api.hip pack_winograd_u)
    const float* bias;  // [COUT]
    const unsigned* slot_tab;   // per LDS slot of the raw tile: (row << 20 | column << 8 | channel group), ~0u = deliver zeros (host: wino_slot_table)
    float* out;         // [N, Hs, Ws, out_ps] (+ out_coff)
    float* pool;        // POOL: [N, Hc/2, Wc/2, COUT]
    const float* zw;    // ZOUT: the NEXT layer's 64 -> 27 (tap x channel) weights, packed as for k_conv_tail
    float* zout;        // ZOUT: planar [N, 27, Hs, Ws] instead of `out`
    int N, Hin, Win, in_ps;
    int Hc, Wc, Hs, Ws;
    int out_ps, out_coff;
    int tiles_x, tiles_y, tiles_total, tiles_per_xcd;
    unsigned rcp_x, rcp_xy;   // ceil(2^32 / tiles_x), ceil(2^32 / (tiles_x*tiles_y)): division by multiply-high (host: tile_rcp)
    int walk;                 // k_wino42_conv: 0 = one (tile, column block) per workgroup; > 0 = tile walkers per XCD group (gridDim.x / 8)
};

constexpr int WN2 = 64;       // output channels per workgroup

// ABLATE (timing experiments only, tools/layer_bench; wrong results when non-zero): 1 no DMA after the prologue,
// 2 B quads loaded once, 4 A operand built once, 8 no epilogue, 16 epilogue without the global stores, 32 no per-chunk
// barrier, 64 no prologue DMA, 128 no de-phasing of the two workgroups of a CU, 256 s_memtime trace into a.pool (results stay correct).
//
// ZOUT (upconv1[0], the producer of the last layer's input): instead of storing its 64 output channels the workgroup
// applies the channel contraction of the NEXT layer, upconv1[2] = Conv2d(64, 3, 3, padding=1) (app.py:77), to them while they
// are still in LDS:  z[p][3*tap + co] = sum_ci relu(y[p][ci]) * W2[co][ci][tap]  — a [pixels x 64] x [64 x 27] product that has no
// halo, 32 more MFMAs per wave after the 512 of the main loop — and stores z as 27 planes [N, 27, H, W].  The last layer is
// then only the nine-tap shifted sum, bias and tanh (k_conv_tail_z): the 64-channel tensor (268 B per pixel written here and
// read there) never exists, 108 B per pixel of z take its place.
template <int CIN, int COUT, bool POOL, int TC, int ABLATE = 0, bool ZOUT = false>
__global__ void __launch_bounds__(THREADS, 2) k_wino64_conv(const WinoArgs a) {
    static_assert(!ZOUT || (COUT == 64 && !POOL), "ZOUT contracts exactly the 64 channels of one column-block pair");
#ifndef CID_EXPERIMENTS
    static_assert(ABLATE == 0, "ablation/trace variants are built only by csrc/tools (-DCID_EXPERIMENTS)");
#endif
    constexpr int TRW = 32 / TC;                 // tile rows per workgroup
    constexpr int LW = 2 * TC + 2, LH = 2 * TRW + 2;
    // LDS row stride in pixels.  TC=16 puts two tile rows in one 32-lane read; their slot offset (2 rows x LWS x 5 slots) must be
    // a multiple of 16 slots or the two half-rows collide in ds_read_b128's bank columns: 34 -> 40.
    constexpr int LWS = (TC == 16) ? 40 : LW;
    constexpr int LPIX = LWS * LH;
    constexpr int NROUND = (LPIX * WPS + 63) / 64;
    constexpr int RW = (NROUND + 3) / 4;
    constexpr int BUF = NROUND * 64;
    constexpr int NCHUNK = CIN / WK;
    constexpr int NB = COUT / WN2;
    constexpr int HWD = LWS / 2;
    // Same-box A/B of the whole forward, layer by layer: one DMA round per MFMA group gains 1-2.6 % over a burst on the layers with
    // CIN <= 128 and loses ~1 % on the two with CIN = 256, where two rounds per group gain 1.2-1.5 %; chosen per layer.
    // Side work of an MFMA group (DMA rounds, transform VALU, LDS reads, B refill) either in front of the group's four MFMAs or
    // one piece after each of them.  Same-box A/B, layer by layer: the interleaved form gains 0.5-2 % on the layers with
    // CIN >= 128 and loses 1.5-4 % on the two with CIN = 64.
    constexpr bool INTERLEAVE = CIN >= 128;
    constexpr int DMA_PER_GROUP = CIN <= 128 ? 1 : 2;
    static_assert(CIN % WK == 0 && COUT % WN2 == 0 && (TC == 16 || TC == 32), "layer dims");
    static_assert(NCHUNK % 2 == 0 && NCHUNK >= 4, "chunks are walked in (even, odd) buffer pairs");
    constexpr int LDS_SLOTS_K = 4096;            // 64 KiB: 2 raw buffers, later the 4x4 exchange blocks, later store staging
    static_assert(2 * BUF <= LDS_SLOTS_K && RW <= 8, "LDS budget");
    __shared__ f32x4 lds[LDS_SLOTS_K];

    int mt, nb;
    if (!decode_block(a.tiles_total, a.tiles_per_xcd, NB, mt, nb)) return;
    int n, ty, tx;
    decode_tile(mt, a.tiles_x, a.tiles_y, a.rcp_x, a.rcp_xy, n, ty, tx);
    const int y0 = ty * (2 * TRW), x0 = tx * (2 * TC);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = row a of B^T d B this wave accumulates
    const int i = lane & 31, h = lane >> 5;
    const int tr = i / TC, tc = i - tr * TC;

    // ABLATE bit 8 (trace experiment, results stay correct, non-POOL layers): thread 0 writes s_memtime stamps + HW_ID to a.pool
    unsigned long long* trace = (ABLATE & 256) ? reinterpret_cast<unsigned long long*>(a.pool) + (size_t)blockIdx.x * 16 : nullptr;
    if ((ABLATE & 256) && tid == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        trace[0] = __builtin_readcyclecounter(); trace[4] = hwid; trace[5] = xcc;
    }
    const float bias_v = a.bias[nb * WN2 + (wave >> 1) * 32 + i];   // epilogue role of wave w: column block w>>1, tiles 16*(w&1)..+16

    if (blockIdx.x < 2 * 256) {   // de-phase the two workgroups of a CU once
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        // Measured on this kernel (tools/layer_bench): without the de-phasing 2.40 ms, with it 2.27 ms (upconv1.0 shape).
        if ((hwid & 1u) && !(ABLATE & 128)) {   // bit 7 (experiment): no de-phasing
            for (int sl = 0; sl < NCHUNK / 2; ++sl) __builtin_amdgcn_s_sleep(127);
        }
    }

    // rows of the 4x4 input patch that feed row a of B^T d:  t = x + sgn*y
    //   a=0: d0 - d2   a=1: d1 + d2   a=2: d2 - d1   a=3: d1 - d3
    const int xrow = (wave == 0) ? 0 : (wave == 2) ? 2 : 1;
    const int yrow = (wave == 0 || wave == 1) ? 2 : (wave == 2) ? 1 : 3;
    const float sgn = (wave == 1) ? 1.f : -1.f;
    const int pbase = (2 * tr) * LWS + tc;
    const int xb = (pbase + xrow * LWS) * WPS + h, yb = (pbase + yrow * LWS) * WPS + h;
    auto col_off = [](int c) { return ((c & 1) * HWD + (c >> 1)) * WPS; };

    // ---- LDS-DMA sources ----
    const float* inb = a.in + (size_t)n * a.Hin * a.Win * a.in_ps;
    const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)inb, (short)0, a.Hin * a.Win * a.in_ps * 4, 0x00020000);
    unsigned voff[RW];
    {
        // All RW table entries are requested before the first is used, and the offsets are formed without branches:
        // a guarded load or a divergent `ok ? offset : sentinel` compiles to load -> wait -> branch -> next load, i.e.
        // RW serialised memory latencies in every workgroup's prologue.  The table is padded to 4*RW rounds (host).
        unsigned ent[RW];
#pragma unroll
        for (int m = 0; m < RW; ++m) ent[m] = a.slot_tab[(wave + 4 * m) * 64 + lane];
#pragma unroll
        for (int m = 0; m < RW; ++m) {
            const unsigned e = ent[m];
            const int gy = y0 - 1 + (int)(e >> 20), gx = x0 - 1 + (int)((e >> 8) & 0xfffu);
            const bool ok = e != ~0u && (unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win;
            const unsigned off = (unsigned)(((gy * a.Win + gx) * a.in_ps + (int)(e & 0xffu) * 4) * 4);
            const unsigned keep = ok ? 0xffffffffu : 0u;
            voff[m] = (off & keep) | (0x7ffffff0u & ~keep);
        }
    }
    if ((ABLATE & 256) && tid == 0) trace[8] = __builtin_readcyclecounter() + (voff[0] & 0u);   // slot table arrived, offsets formed
    const unsigned lds_base = (unsigned)(uintptr_t)(&lds[0]);
    auto dma_round = [&](int buf, int ck, int m) {   // round m of this wave: 64 slots of chunk ck -> LDS buffer `buf`
        if (wave + 4 * m < NROUND) {                   // wave-uniform
            const int soff = ck * (WK * 4);
            const unsigned dst = lds_base + (unsigned)((buf * BUF + (wave + 4 * m) * 64) * 16);
            // M0 (the LDS-DMA's wave-uniform LDS address) is the compiler's register: the asm saves and restores it, so whatever hipcc
            // keeps there across this statement survives ("m0" cannot be named as a clobber: reserved register).
            // k_wino42_conv and k_conv3x3_h16 use __builtin_amdgcn_raw_ptr_buffer_load_lds instead; with that builtin in THIS
            // kernel hipcc (ROCm 7.2) silently drops the host-side launch stub of every instantiation, so the asm form stays.
            unsigned m0_saved;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_saved) : "v"(voff[m]), "s"(rsrc_in), "s"(dst), "s"(soff) : "memory");
        }
    };
    auto dma_chunk = [&](int buf, int ck) {
#pragma unroll
        for (int m = 0; m < RW; ++m) dma_round(buf, ck, m);
    };

    f32x16 acc[2][4];   // [column block nt][position b]; first written by the zero-C MFMAs of chunk 0

    // U stream of this wave: [nb][chunk][round g2][a][nt][e][lane][b]: unit (ck, g2) is 8 KiB, quad (nt, e) 1 KiB inside it
    const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc((void*)a.u, (short)0, CIN * COUT * 16 * 4, 0x00020000);
    const int ubase = (nb * NCHUNK * 2 * 4 + wave) * 8192;   // bytes, wave-uniform
    const int ulane = lane * 16;
    auto b_load = [&](int gunit, int q) -> f32x4 {          // gunit = ck*2 + g2, q = nt*4 + e
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, ulane, ubase + gunit * (4 * 8192) + q * 1024, 0));
    };

    // ---- prologue: chunks 0 and 1 -> both LDS buffers; B of unit 0 ----
    // Only what the first unit needs is requested before the first MFMA: the B quads of unit 0 (L2 hits) and chunk 0.
    // Under load the vector-memory instructions of a cold prologue take ~1k cycles EACH to issue (s_memtime trace,
    // tools/trace_stats.py): requesting chunk 1 here as well kept the workgroup out of its main loop 4k cycles longer;
    // it is requested at the start of chunk 0's unit 0 instead and has that unit to land.
    f32x4 bq[2][4];
#pragma unroll
    for (int q = 0; q < 8; ++q) bq[q >> 2][q & 3] = b_load(0, q);
    if ((ABLATE & 256) && tid == 0) trace[7] = __builtin_readcyclecounter();    // B requested
    if (!(ABLATE & 64)) dma_chunk(0, 0);   // bit 6 (experiment): no prologue DMA
    if ((ABLATE & 256) && tid == 0) trace[15] = __builtin_readcyclecounter();   // chunk 0 requested
    if ((ABLATE & 256) && tid == 0) trace[9] = __builtin_readcyclecounter();    // DMA and B loads issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the DMA is invisible to hipcc's own wait counting
    if ((ABLATE & 256) && tid == 0) trace[10] = __builtin_readcyclecounter();   // ... and landed (this wave)
    __syncthreads();
    if ((ABLATE & 256) && tid == 0) trace[11] = __builtin_readcyclecounter();   // ... in every wave

    auto read_cols = [&](f32x4 (&xq)[2], f32x4 (&yq)[2], int bufbase, int g2, int c0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            xq[c] = lds[bufbase + xb + 2 * g2 + col_off(c0 + c)];
            yq[c] = lds[bufbase + yb + 2 * g2 + col_off(c0 + c)];
        }
    };
    auto make_t = [&](f32x4 (&t)[4], const f32x4 (&xq)[2], const f32x4 (&yq)[2], int c0) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) t[c0 + c][e] = __builtin_fmaf(sgn, yq[c][e], xq[c][e]);
    };

    f32x4 vcur[4], vnxt[4];
    {
        f32x4 xq[2], yq[2], t[4];
        read_cols(xq, yq, 0, 0, 0);
        make_t(t, xq, yq, 0);
        read_cols(xq, yq, 0, 0, 2);
        make_t(t, xq, yq, 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) {   // element-wise on purpose (no v_pk_add_f32: tools/mix_bench)
            vcur[0][e] = t[0][e] - t[2][e];
            vcur[1][e] = t[1][e] + t[2][e];
            vcur[2][e] = t[2][e] - t[1][e];
            vcur[3][e] = t[1][e] - t[3][e];
        }
    }

    // Chunk ck in LDS buffer PAR.  Unit k = 32 MFMAs (column block nt outer, k-step e, position b inner); under them the A operand of the next unit is read and built, and each B
    // quad is refilled for the next unit as soon as its four MFMAs have issued.
    auto chunk = [&](auto first_tag, auto more_tag, auto dma_tag, auto parity_tag, int ck) {
        constexpr bool FIRST = decltype(first_tag)::value;    // chunk 0: accumulators start from a zero C operand
        constexpr bool MORE = decltype(more_tag)::value;      // a chunk ck+1 exists
        constexpr bool DMA = decltype(dma_tag)::value;        // a chunk ck+2 exists: fetch it into this chunk's buffer
        constexpr int PAR = decltype(parity_tag)::value ? 1 : 0;
        constexpr int cur = PAR * BUF, nxt = BUF - cur;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool have_next_unit = (k == 0) || MORE;
            const int nbuf = (k == 0) ? cur : nxt, ng2 = 1 - k;
            f32x4 xq[2], yq[2], t[4];
            // this buffer's last reads (building unit 1) happened during unit 0, before the barrier below
            if (FIRST && k == 0 && !(ABLATE & 65)) dma_chunk(1, 1);          // chunk 1: see the prologue
            const bool build = have_next_unit && !(ABLATE & 4);
            // the eight LDS reads of the next unit's A operand go out two per MFMA group (groups 0-3), not four at a time
            auto read_col = [&](int slot, int c) {   // x and y of patch column c into pair slot `slot`
                xq[slot] = lds[nbuf + xb + 2 * ng2 + col_off(c)];
                yq[slot] = lds[nbuf + yb + 2 * ng2 + col_off(c)];
            };
            if (build) read_col(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // 8 groups of four MFMAs, group g = (column block nt = g/4, k-step e = g%4): one column block's four k-steps
            // first, i.e. the same four accumulators in rotation for 16 MFMAs (measured 1.5 % faster than alternating
            // the column blocks; the per-accumulator summation order is the same either way).
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int nt = g >> 2, e = g & 3;
                // DMA_PER_GROUP rounds per MFMA group instead of RW in a row at the start of the unit — a vector-memory
                // instruction takes 40-600 cycles to issue (tools/issue_bench) and a lone wave issues nothing else meanwhile.
                // The counted wait at the next barrier still holds: after the last round (group <= 7) come at least the 8 B
                // refills of the next unit.
                if (INTERLEAVE) {
                auto mfma = [&](int b) {
                    if (FIRST && k == 0 && e == 0) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[nt][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(vcur[b][e], bq[nt][e][b], zero, 0, 0, 0);
                    } else {
                        acc[nt][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(vcur[b][e], bq[nt][e][b], acc[nt][b], 0, 0, 0);
                    }
                };
                mfma(0);
                if (DMA && k == 1 && !(ABLATE & 1)) {
#pragma unroll
                    for (int j = 0; j < DMA_PER_GROUP; ++j)
                        if (DMA_PER_GROUP * g + j < RW) dma_round(PAR, ck + 2, DMA_PER_GROUP * g + j);
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma(1);
                if (build) {
                    if (g == 2) make_t(t, xq, yq, 0);
                    if (g == 4) {
                        make_t(t, xq, yq, 2);
#pragma unroll
                        for (int q = 0; q < 4; ++q) { vnxt[0][q] = t[0][q] - t[2][q]; vnxt[1][q] = t[1][q] + t[2][q]; }
                    }
                    if (g == 6) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { vnxt[2][q] = t[2][q] - t[1][q]; vnxt[3][q] = t[1][q] - t[3][q]; }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma(2);
                if (build) {
                    if (g == 1) read_col(1, 1);
                    if (g == 2) read_col(0, 2);
                    if (g == 3) read_col(1, 3);
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma(3);
                if (have_next_unit && !(ABLATE & 2)) bq[nt][e] = b_load(ck * 2 + k + 1, nt * 4 + e);
                __builtin_amdgcn_sched_barrier(0);
                } else {
                if (DMA && k == 1 && !(ABLATE & 1)) {
#pragma unroll
                    for (int j = 0; j < DMA_PER_GROUP; ++j)
                        if (DMA_PER_GROUP * g + j < RW) dma_round(PAR, ck + 2, DMA_PER_GROUP * g + j);
                }
                if (build) {
                    if (g == 1) read_col(1, 1);
                    if (g == 2) { make_t(t, xq, yq, 0); read_col(0, 2); }
                    if (g == 3) read_col(1, 3);
                    if (g == 4) {
                        make_t(t, xq, yq, 2);
#pragma unroll
                        for (int q = 0; q < 4; ++q) { vnxt[0][q] = t[0][q] - t[2][q]; vnxt[1][q] = t[1][q] + t[2][q]; }
                    }
                    if (g == 6) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { vnxt[2][q] = t[2][q] - t[1][q]; vnxt[3][q] = t[1][q] - t[3][q]; }
                    }
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (FIRST && k == 0 && e == 0) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[nt][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(vcur[b][e], bq[nt][e][b], zero, 0, 0, 0);
                    } else {
                        acc[nt][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(vcur[b][e], bq[nt][e][b], acc[nt][b], 0, 0, 0);
                    }
                }
                if (have_next_unit && !(ABLATE & 2)) bq[nt][e] = b_load(ck * 2 + k + 1, nt * 4 + e);
                __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (build) {
#pragma unroll
                for (int b = 0; b < 4; ++b) vcur[b] = vnxt[b];
            }
            if (MORE && k == 0 && !(ABLATE & 32)) {   // bit 5 (experiment): no per-chunk barrier
                // The DMA of chunk ck+1 was issued one chunk ago; the only vector-memory operations younger than it
                // are the B refills of the previous unit (consumed above) and of this one: at most 8 outstanding
                // means every DMA of this wave has landed; past the barrier every wave's has.
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                __syncthreads();
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    if ((ABLATE & 256) && tid == 0) trace[1] = __builtin_readcyclecounter();
    chunk(T{}, T{}, T{}, F{}, 0);
    chunk(F{}, T{}, std::integral_constant<bool, (NCHUNK > 3)>{}, T{}, 1);
    for (int ck = 2; ck + 2 < NCHUNK; ck += 2) {
        // chunk ck fetches ck+2 (exists: ck+2 < NCHUNK); chunk ck+1 fetches ck+3 (exists iff ck+3 < NCHUNK, true: NCHUNK even)
        chunk(F{}, T{}, T{}, F{}, ck);
        chunk(F{}, T{}, T{}, T{}, ck + 1);
    }
    chunk(F{}, T{}, F{}, F{}, NCHUNK - 2);
    chunk(F{}, F{}, F{}, T{}, NCHUNK - 1);
    if ((ABLATE & 256) && tid == 0) trace[2] = __builtin_readcyclecounter();

    // ---- output transform ----
    // Four code versions selected by a wave-uniform switch, so that "is this my own row / my own block" is a
    // compile-time fact (no selects).  Column transform of this wave's row: m'[nt][b'] (b' = 0,1);
    // A^T = [[1,1,1,0],[0,1,-1,-1]].
    if (ABLATE & 8) {   // keep the accumulators alive without the epilogue
        float sum = 0.f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[nt][b][r];
        if (sum == 123.456f) a.out[tid] = sum;
        return;
    }
    auto epilogue = [&](auto wave_tag) {
        constexpr int W = decltype(wave_tag)::value;
        constexpr int NT_W = W >> 1, RH = W & 1;                // epilogue role: column block, tile half
        __syncthreads();                                        // raw tiles are dead: LDS becomes the exchange area
        if ((ABLATE & 256) && tid == 0) trace[12] = __builtin_readcyclecounter();
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        // exchange block (src row a, dst wave w): 8 registers x 64 lanes of f32x2 = 4 KiB at ((a*4 + w) * 512) f32x2
        f32x2* ex = reinterpret_cast<f32x2*>(lds);
        f32x2 own[8];                                           // this wave's own row for its own outputs
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                f32x2 m;
                m[0] = acc[nt][0][r] + acc[nt][1][r] + acc[nt][2][r];
                m[1] = acc[nt][1][r] - acc[nt][2][r] - acc[nt][3][r];
                const int w = nt * 2 + (r >> 3);                // consumer of (column block nt, register half r>>3)
                if (w == W) own[r & 7] = m;
                else ex[(W * 4 + w) * 512 + (r & 7) * 64 + lane] = m;
            }
        __syncthreads();
        if ((ABLATE & 256) && tid == 0) trace[13] = __builtin_readcyclecounter();
        float y[2][2][8];   // [output row a'][column b'][register]
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            f32x2 m[4];
#pragma unroll
            for (int arow = 0; arow < 4; ++arow) m[arow] = (arow == W) ? own[rr] : ex[(arow * 4 + W) * 512 + rr * 64 + lane];
#pragma unroll
            for (int bp = 0; bp < 2; ++bp) {   // fixed summation order (golden outputs depend on it)
                y[0][bp][rr] = (m[0][bp] + m[1][bp]) + m[2][bp];
                y[1][bp][rr] = m[1][bp] - (m[2][bp] + m[3][bp]);
            }
        }
        __syncthreads();                                        // exchange area is dead: reuse as store staging
        if ((ABLATE & 256) && tid == 0) trace[14] = __builtin_readcyclecounter();
        float* stg = reinterpret_cast<float*>(lds) + W * (64 * WS32);
        // register rr of this wave is tile  T = 16*RH + (rr&3) + 8*(rr>>2) + 4*h  of the workgroup's 32
        auto tl_of = [&](int rr) { return (rr & 3) + 8 * (rr >> 2) + 4 * h; };
        float pooled[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            if (POOL) pooled[rr] = fmaxf(fmaxf(fmaxf(y[0][0][rr], y[0][1][rr]), fmaxf(y[1][0][rr], y[1][1][rr])) + bias_v, 0.f);
#pragma unroll
            for (int ap = 0; ap < 2; ++ap)
#pragma unroll
                for (int bp = 0; bp < 2; ++bp)
                    stg[(ap * 32 + 2 * tl_of(rr) + bp) * WS32 + i] = fmaxf(y[ap][bp][rr] + bias_v, 0.f);
        }
        // staged pixel sp = a' * 32 + (2*tl + b'): the wave's 16 tiles are tile row WTR, tile columns WTC .. WTC+15
        constexpr int WTR = (16 * RH) / TC, WTC = (16 * RH) % TC;
        const int wy = y0 + 2 * WTR, wx = x0 + 2 * WTC;
        if constexpr (ZOUT) {
            // Waves (NT_W = 0, RH) and (NT_W = 1, RH) have staged the two channel halves of the SAME 64 pixels (2 rows x 32
            // columns).  Wave (NT_W, RH) contracts row a' = NT_W: A fragments from both stagings (pixel stride 36 floats =
            // 9 slots: conflict-free ds_read_b128), B = the packed 64 x 32 (27 used) weights, 32 MFMAs.
            f32x4 zb[2][4];
#pragma unroll
            for (int ck = 0; ck < 2; ++ck)
#pragma unroll
                for (int g = 0; g < 4; ++g) zb[ck][g] = reinterpret_cast<const f32x4*>(a.zw)[(ck * 4 + g) * 64 + lane];
            __syncthreads();                                    // both stagings are complete
            f32x16 zacc;
#pragma unroll
            for (int ck = 0; ck < 2; ++ck) {
                const float* src = reinterpret_cast<const float*>(lds) + (ck * 2 + RH) * (64 * WS32) + (NT_W * 32 + i) * WS32 + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(src + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (ck == 0 && g == 0 && e == 0) {
                            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            zacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], zb[ck][g][e], zero, 0, 0, 0);
                        } else {
                            zacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], zb[ck][g][e], zacc, 0, 0, 0);
                        }
                    }
                }
            }
            // lane (column j = i, half h) holds z[pixel (r&3) + 8*(r>>2) + 4*h][j]: registers 4q..4q+3 are four consecutive
            // pixels of one row of plane j — one 16-byte store each (H, W are multiples of 4: a group is inside or outside)
            const int yy = wy + NT_W;
            if (i < 27 && yy < a.Hs && !(ABLATE & 16)) {
                float* zrow = a.zout + (((size_t)n * 27 + i) * a.Hs + yy) * a.Ws;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int xx = wx + 8 * q + 4 * h;
                    if (xx < a.Ws) *reinterpret_cast<f32x4*>(zrow + xx) = f32x4{zacc[4 * q], zacc[4 * q + 1], zacc[4 * q + 2], zacc[4 * q + 3]};
                }
            }
            return;
        }
        wave_lds_fence();
        const int cbase = a.out_coff + nb * WN2 + NT_W * 32;
        const bool full = (y0 + 2 * TRW <= a.Hs) && (x0 + 2 * TC <= a.Ws);
        if (ABLATE & 16) {
            // experiment: everything but the global stores
        } else if (full) {
            const int lane_off = (lane >> 3) * a.out_ps + (lane & 7) * 4;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int ap = it >> 2, xin = (it & 3) * 8;       // compile-time
                float* rowp = a.out + ((size_t)(n * a.Hs + wy + ap) * a.Ws + wx + xin) * a.out_ps + cbase;   // uniform
                const f32x4 v = *reinterpret_cast<const f32x4*>(stg + (it * 8 + (lane >> 3)) * WS32 + (lane & 7) * 4);
                *reinterpret_cast<f32x4*>(rowp + lane_off) = v;
            }
        } else {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int yy = wy + (it >> 2), xx = wx + (it & 3) * 8 + (lane >> 3);
                const f32x4 v = *reinterpret_cast<const f32x4*>(stg + (it * 8 + (lane >> 3)) * WS32 + (lane & 7) * 4);
                if (yy < a.Hs && xx < a.Ws)
                    *reinterpret_cast<f32x4*>(a.out + ((size_t)(n * a.Hs + yy) * a.Ws + xx) * a.out_ps + cbase + (lane & 7) * 4) = v;
            }
        }
        if (POOL) {
            wave_lds_fence();
            const int Hp = a.Hc >> 1, Wp = a.Wc >> 1;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) stg[tl_of(rr) * WS32 + i] = pooled[rr];
            wave_lds_fence();
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int tl = it * 8 + (lane >> 3);
                const int py = (y0 >> 1) + WTR, px = (x0 >> 1) + WTC + tl;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stg + tl * WS32 + (lane & 7) * 4);
                if (py < Hp && px < Wp)
                    *reinterpret_cast<f32x4*>(a.pool + ((size_t)(n * Hp + py) * Wp + px) * COUT + nb * WN2 + NT_W * 32 + (lane & 7) * 4) = v;
            }
        }
    };
    switch (wave) {
        case 0: epilogue(std::integral_constant<int, 0>{}); break;
        case 1: epilogue(std::integral_constant<int, 1>{}); break;
        case 2: epilogue(std::integral_constant<int, 2>{}); break;
        default: epilogue(std::integral_constant<int, 3>{}); break;
    }
    if ((ABLATE & 256) && tid == 0) {
        trace[3] = __builtin_readcyclecounter();                 // this wave's stores are issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        trace[6] = __builtin_readcyclecounter();                 // ... and written back
    }
}

}  // namespace cid
