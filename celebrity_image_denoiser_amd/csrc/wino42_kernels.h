// wino42_kernels.h — 3x3 convolution (+bias, +ReLU, +optional 2x2 max-pool) as Winograd F(4x2,3x3) — output tiles 4 pixels wide and
// 2 high — on the exact-f32 matrix instruction v_mfma_f32_16x16x4_f32 (gfx950): 16 tiles x 64 output channels per workgroup, one
// row of the transformed 4x6 tile per wave (four waves).
//
// Same function as the reference's nn.Conv2d(k=3, p=1) + nn.ReLU (+ nn.MaxPool2d(2,2)) stages (backend/app.py:43-77) with 24
// multiplies per 8 output pixels and (ci, co) pair: 3 per pixel against 9 (direct) and 4 (F(2x2,3x3), wino64_kernels.h):
//       M_xi[tile][co] = sum_ci V_xi[tile][ci] * U_xi[ci][co]  for the 24 positions xi = (a, b),   V = B2^T d B4,  U = G2 g G4^T,  Y(2x4) = A2^T M A4.
// Vertically the F(2,3) transform of k_wino64_conv; horizontally F(4,3) at the interpolation points 0, +-3/4, +-3/2, infinity:
// every entry of B4^T and A4^T is a dyadic rational (exact in fp32), and of the point sets tried this one has the smallest error
// on this network (tools/emulate_f43_winograd.py, both directions at F(4,3): 2.8e-6 on the He-gain weights against 1.3e-6 for
// direct fp32 and 5.8e-6 for the textbook points 0, +-1, +-2).  U is computed on the host in double and rounded once
// (cid_api.hip pack_winograd42_u).
//
// Why not F(4x4,3x3) (2.25 multiplies per pixel): with a row of the 6x6 tile per wave it needs six waves of ~170 registers; hipcc
// needs 225-256 for that loop, so only one such workgroup fits a CU and two of its SIMDs carry two waves against one on the
// others — measured at the speed of F(2x2) (git history: k_wino43_conv).  Four rows fit the 2-waves-per-SIMD budget of k_wino64_conv.
//
//   * wave a = row a of B2^T d: 6 positions b x 16 tiles x 64 channels = 24 accumulator tiles of 4 registers.  MFMA row m = tile
//     m of the workgroup (TC tiles per row, 16/TC rows), k = lane >> 4 = channel 4g + s of the 16-channel chunk at k-step s;
//     one V value feeds four MFMAs (the four 16-channel groups of the column block).
//   * raw halo tile in LDS, double-buffered per 16-channel chunk, written by LDS-DMA (out-of-image and pad slots: out-of-range
//     offset, the range check writes zeros).  Layout [row][x mod 4][x div 4] of pixels, pixel = 4 channel-group quads + 1 pad
//     quad (as k_wino64_conv: four DMA lanes fetch one pixel's 64 contiguous bytes; with the channel groups in separate planes
//     every lane of a round touched another 128-byte line and the rounds cost 0.45 of 2.37 ms, tools/layer_bench).  The 16 tiles
//     of a channel group lie in 16 different quads (mod 16) for every tile shape (row stride 68 / 36 / 26 pixels), and odd channel
//     groups read the other 8-byte half of their quads, so the 32 lanes of a ds_read_b64 service group cover all 64 banks.
//   * V is never stored: per 8-channel unit a lane reads, for each of the 6 patch columns, its 2 contributing rows (ds_read_b64:
//     two channels) and forms t = x +- y; the six V[b] follow from the six t (even/odd split of the +-p columns) — 40 VALU
//     instructions under the previous unit's 48 MFMAs.
//   * B (U quads: the four channel groups of one (b, k-step)) straight from L2 into a ring of six quads, refilled right after use.
//   * epilogue: column transform in registers (6 -> 4); the row transform (4 -> 2) needs all four waves' rows: wave w takes the
//     tile quarter r = w (tiles 4g + w, all 64 channels), the other three rows come through LDS (48 KiB, one pass); bias, ReLU,
//     optional 2x2 max-pool.  (r3) MFMA column j of channel group cg is output channel 4j + cg, so a lane's four groups are four
//     consecutive channels and its results leave as 16-byte quads straight from registers (sixteen lanes = one pixel's 256 bytes):
//     the transposing pass through LDS of rounds 1-2 (32 ds_write_b32 + 8 ds_read_b128 per lane and item) and the epilogue's third
//     barrier are gone (the ZOUT variant still stages: its contraction reads the pixels as MFMA operands).
//   * (r3) a workgroup WALKS tiles: work item id = blockIdx.x, + gridDim.x, ... (cid_api.hip launches about two workgroups per CU
//     when there are more items than that).  Chunk 0 of a tile lives in a third LDS buffer X that the exchange / staging area of
//     the epilogue does not touch, so the NEXT tile's chunk 0 is fetched under the current tile's last chunk (its DMA offsets are
//     formed under the last-but-one chunk) and the B ring runs on into the next tile's first quads: of the prologue that every
//     tile used to pay (slot table -> offsets -> first DMA + B requests -> landed -> barrier: ~7k cycles next to a 25-55k main
//     loop) only the first V build is left.  gridDim.x / 8 is a multiple of NB, so a workgroup keeps its column block.
#pragma once
#include "wino64_kernels.h"

namespace cid {

template <int TC>
struct W42Geom {
    static_assert(TC == 16 || TC == 8 || TC == 4, "tiles per workgroup row");
    static constexpr int TRW = 16 / TC;                       // tile rows per workgroup
    static constexpr int LW = 4 * TC + 2, LH = 2 * TRW + 2;   // raw halo tile, pixels
    static constexpr int QS = TC + 1;                         // pixels per (row, x mod 4) run
    static constexpr int RS = TC == 16 ? 68 : TC == 8 ? 36 : 22;   // row stride in pixels: tile rows (2 raw rows) 8 / 12 pixels apart (mod 16)
    static_assert(RS >= 4 * QS, "row holds four column planes");
    static constexpr int SLOTS = WPS * LH * RS;               // quads per chunk buffer: pixel = 4 channel-group quads + 1 pad
    static constexpr int NROUND = (SLOTS + 63) / 64, RW = (NROUND + 3) / 4;
    static constexpr int BUF = NROUND * 64;                   // buffer stride: whole DMA rounds (the last round's spare lanes write zeros)
    static constexpr int TABQ = RW * 64;                      // quads of one DMA offset table: 4 waves x RW rounds x 64 lanes x 4 bytes
};

// Host: LDS quad s of a chunk buffer -> packed (row << 20 | column << 8 | channel group), ~0u = deliver zeros.  Padded to 4*RW rounds.
template <int TC>
inline int wino42_slot_table(unsigned* out /* may be null */) {
    using Gm = W42Geom<TC>;
    const int n = 4 * Gm::RW * 64;
    if (out)
        for (int s = 0; s < n; ++s) {
            unsigned e = ~0u;
            if (s < Gm::SLOTS) {
                const int p = s / WPS, g = s % WPS, y = p / Gm::RS, r2 = p % Gm::RS, xp = r2 / Gm::QS, q = r2 % Gm::QS;
                const int x = 4 * q + xp;
                if (g < 4 && xp < 4 && x < Gm::LW) e = (unsigned)y << 20 | (unsigned)x << 8 | (unsigned)g;
            }
            out[s] = e;
        }
    return n;
}

// A4^T of F(4,3) at the points 0, 3/4, -3/4, 3/2, -3/2, inf applied to six values: y[i] = sum_j p_j^i m[j] (+ m[5] for i = 3)
__device__ __forceinline__ void w42_out4(const float (&m)[6], float (&y)[4]) {
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = (m[0] + s12) + s34;
    y[1] = __builtin_fmaf(1.5f, d34, 0.75f * d12);
    y[2] = __builtin_fmaf(2.25f, s34, 0.5625f * s12);
    y[3] = __builtin_fmaf(3.375f, d34, __builtin_fmaf(0.421875f, d12, m[5]));
}

// ABLATE (timing experiments only, tools/layer_bench; wrong results when non-zero): 1 no DMA after the prologue, 2 B quads loaded once,
// 4 V built once, 8 no epilogue, 256 s_memtime phase sums of thread 0 into a.zout (results stay correct, non-ZOUT layers).
template <int CIN, int COUT, bool POOL, int TC, int ABLATE = 0, bool ZOUT = false>
__global__ void __launch_bounds__(THREADS, 2) k_wino42_conv(const WinoArgs a) {
    static_assert(!ZOUT || (COUT == 64 && !POOL), "ZOUT contracts exactly the 64 channels of the workgroup's column block");
#ifndef CID_EXPERIMENTS
    static_assert(ABLATE == 0, "ablation/trace variants are built only by csrc/tools (-DCID_EXPERIMENTS)");
#endif
    using Gm = W42Geom<TC>;
    constexpr int TRW = Gm::TRW, RS = Gm::RS, QS = Gm::QS, BUF = Gm::BUF, NROUND = Gm::NROUND, RW = Gm::RW;
    constexpr int NCHUNK = CIN / WK, NU = 2 * NCHUNK;
    constexpr int NB = COUT / WN2;
    static_assert(CIN % WK == 0 && COUT % WN2 == 0, "layer dims");
    static_assert(NCHUNK % 2 == 0 && NCHUNK >= 4, "chunks are walked in (even, odd) buffer pairs");
    // LDS (quads of 16 bytes): buffer X (chunk 0 of a tile) and the two DMA offset tables (this tile's and the next one's), which
    // the epilogue must not touch, come first; then the work area: raw buffers 0 / 1, later the exchange blocks (48 KiB), later
    // store staging.  Everything the main loop reads lies below 64 KiB, so every ds_read address is one base register + a 16-bit
    // immediate.  TC = 8: 1088 + 640 + 3072 quads = 75 KiB: two workgroups per CU.
    constexpr int WORK = 3072, TABQ = Gm::TABQ;
    static_assert(2 * BUF <= WORK, "LDS budget");
    constexpr int XB = 0, TAB0 = BUF, WB = BUF + 2 * TABQ;        // buffer X, offset tables, work area
    static_assert((WB + 2 * BUF) * 16 <= 65536, "raw buffers within reach of a 16-bit ds offset");
    __shared__ f32x4 lds[BUF + 2 * TABQ + WORK];
    f32x4* const ldsw = lds + WB;                                 // the work area
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto boff = [](int buf) { return buf == 2 ? XB : WB + buf * BUF; };   // LDS quad offset of raw buffer 0, 1 or X (= 2)

    // Work items of this workgroup.  blockIdx.x % 8 labels the XCD group, which owns tiles [xcd * tiles_per_xcd, + tiles_per_xcd).
    //   a.walk == 0: one item per workgroup, (blockIdx.x / 8) = tile * NB + column block (small launches; round-2 behaviour);
    //   a.walk  > 0: the workgroup walks tiles local, local + walk, ... of its XCD group (walk = gridDim.x / 8 walkers per group) and
    //                computes ALL NB column blocks of a tile back to back: the tile's input is fetched from HBM once and the other
    //                NB - 1 reads hit this XCD's L2 (with one column block per workgroup the NB readers of a tile drifted apart
    //                and the input was fetched 2-4 times: 2.7 GB per launch for <256,256> against 0.27 GB of input).
    //   a.walk  < 0 (NB = 2 or 4 only): -walk walkers per XCD group, and an XCD group keeps ONE column block: group x computes block x % NB of
    //                the tiles of tile range x / NB (8 / NB ranges of tiles_per_xcd tiles).  Its NB-th of U (1.6 MB of bottleneck.2's 6.3 MB)
    //                then stays in the XCD's 4 MiB L2; the price is that every tile's input is read by NB XCDs (round 4).
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
    const int walkers = a.walk < 0 ? -a.walk : a.walk;
    const bool xnb = NB > 1 && a.walk < 0;
    const int tgrp = xnb ? xcd / NB : xcd;                  // tile range of this XCD group
    int local = walkers ? slot0 : slot0 / NB;               // tile index inside the range
    int nb = walkers ? (xnb ? xcd % NB : 0) : slot0 - local * NB;
    int mt = tgrp * a.tiles_per_xcd + local;
    if (!(mt < a.tiles_total && local < a.tiles_per_xcd)) return;
    int n, ty, tx;
    decode_tile(mt, a.tiles_x, a.tiles_y, a.rcp_x, a.rcp_xy, n, ty, tx);
    int y0 = ty * (2 * TRW), x0 = tx * (4 * TC);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = row a
    const int m16 = lane & 15, g = lane >> 4;
    const int tr = m16 / TC, tc = m16 - tr * TC;
    // ABLATE & 256 (tools/w42_bench): thread 0 sums, over the workgroup's tiles, the shader cycles spent in the main loop, the
    // epilogue and the tile boundary, into a.zout[blockIdx.x * 8 ...] (non-ZOUT variants): start, sums, end, tile count
    unsigned long long* trace = (ABLATE & 256) ? reinterpret_cast<unsigned long long*>(a.zout) + (size_t)blockIdx.x * 8 : nullptr;
    unsigned long long tr_start = 0, tr_main = 0, tr_epi = 0, tr_bnd = 0, tr_t = 0, tr_tiles = 0;
    if (ABLATE & 256) tr_start = tr_t = __builtin_readcyclecounter();
    auto tr_lap = [&](unsigned long long& acc) {
        if (ABLATE & 256) { const unsigned long long now = __builtin_readcyclecounter(); acc += now - tr_t; tr_t = now; }
    };

    // ---- LDS-DMA sources (same table format as k_wino64_conv) ----
    const size_t img_elems = (size_t)a.Hin * a.Win * a.in_ps;
    auto image_rsrc = [&](int img) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)img * img_elems), (short)0, a.Hin * a.Win * a.in_ps * 4, 0x00020000);
    };
    // Per-lane byte offsets of this wave's DMA rounds, parked in the LDS (one ds_read_b32 per round instead of five registers
    // held across the main loop).  Two tables: table `tpar` is the current tile's, the other one the next tile's (its chunk 0 is
    // requested under the current tile's last chunk); they are formed at tile boundaries, where few registers are live.
    // Wave-private: no barrier between writing and reading them.
    unsigned* const voff_tab = reinterpret_cast<unsigned*>(lds + TAB0) + wave * (RW * 64) + lane;
    int tpar = 0;
    int dma_soff0 = 0;   // opaque zero, renewed per tile (tile_scalars below)
    const int in_ps_log2 = __builtin_ctz((unsigned)a.in_ps);   // in_ps is a power of two (host: 64, 128, 256)
    unsigned ent[RW];
    auto load_slot_entries = [&]() {
#pragma unroll
        for (int m = 0; m < RW; ++m) ent[m] = a.slot_tab[(wave + 4 * m) * 64 + lane];
    };
    auto write_offsets = [&](int tab, int ty0, int tx0, bool live) {   // the RW rounds of the tile at (ty0, tx0); !live: every lane delivers zeros
#pragma unroll
        for (int m = 0; m < RW; ++m) {
            const unsigned e = ent[m];
            const int gy = ty0 - 1 + (int)(e >> 20), gx = tx0 - 1 + (int)((e >> 8) & 0xfffu);
            const bool ok = live && e != ~0u && (unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win;
            // full-rate arithmetic only (this runs once per tile, on ALUs the fp32 MFMAs share): a 24-bit multiply (gy, Win < 2^22: one call's
            // image has < 2^22 pixels) and a shift by log2(in_ps) (64, 128 or 256 channels) instead of two quarter-rate 32-bit multiplies
            const unsigned off = ((unsigned)(__umul24((unsigned)gy, (unsigned)a.Win) + gx) << (in_ps_log2 + 2)) + (e & 0xffu) * 16u;
            const unsigned keep = ok ? 0xffffffffu : 0u;
            voff_tab[tab * (TABQ * 4) + m * 64] = (off & keep) | (0x7ffffff0u & ~keep);
        }
    };
    // the item after this one
    bool has_next;
    int n2, y02, x02;
    int nb2, local2;
    auto decode_next = [&]() {   // the item after (local, nb)
        n2 = n; y02 = y0; x02 = x0; local2 = local; nb2 = xnb ? NB : nb + 1;
        has_next = walkers != 0;
        if (has_next && nb2 == NB) {                          // next tile of the walk
            nb2 = xnb ? nb : 0; local2 = local + walkers;
            const int mt2 = tgrp * a.tiles_per_xcd + local2;
            has_next = mt2 < a.tiles_total && local2 < a.tiles_per_xcd;
            if (has_next) {
                int ty2, tx2;
                decode_tile(mt2, a.tiles_x, a.tiles_y, a.rcp_x, a.rcp_xy, n2, ty2, tx2);
                y02 = ty2 * (2 * TRW); x02 = tx2 * (4 * TC);
            }
        }
        if (!has_next) nb2 = nb;                              // keeps the B prefetch of the last item inside U
    };
    load_slot_entries();
    decode_next();
    write_offsets(0, y0, x0, true);
    write_offsets(1, y02, x02, has_next);
    // LDS-DMA through the compiler's builtin (buffer_load_dwordx4 ... offen lds; M0 = the wave-uniform LDS address): hipcc then counts
    // the rounds in its vmcnt bookkeeping, so its waits for the B quads are exact (with inline asm they were one load early per round in
    // flight: +0.3 % same-box)
    __amdgpu_buffer_rsrc_t rsrc_in = image_rsrc(n), rsrc_next = image_rsrc(n2);   // renewed at every tile boundary
    int wave_t = wave;   // = wave, through the per-tile opaque zero: keeps the guard below a scalar compare at its place (hoisted out
                         // of the tile loop hipcc held it as a per-lane boolean — in VGPRs, spilled)
    auto dma_round = [&](const __amdgpu_buffer_rsrc_t& rsrc, int tab, int buf, int ck, int m) {   // round m of this wave: 64 quads of chunk ck -> LDS buffer `buf`
        if (4 * m + 3 < NROUND || wave_t + 4 * m < NROUND) {   // wave-uniform; a run-time test only in the last, partial set of rounds
            const int soff = dma_soff0 + ck * (WK * 4);
            const unsigned vo = voff_tab[tab * (TABQ * 4) + m * 64];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)&lds[boff(buf) + (wave + 4 * m) * 64], 16, vo, soff, 0, 0);
        }
    };
    auto dma_chunk = [&](int buf, int ck) {
#pragma unroll
        for (int m = 0; m < RW; ++m) dma_round(rsrc_in, tpar, buf, ck, m);
    };

    // ---- U stream of this wave: [nb][unit][a][q = 6*e2 + b][lane][cg]: a unit of one wave is 12 KiB, a quad 1 KiB ----
    const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc((void*)a.u, (short)0, CIN * COUT * 24 * 4, 0x00020000);
    // `ubase` is re-derived from an opaque zero at the top of every tile iteration (tile_scalars): otherwise hipcc hoists the ~200
    // loop-invariant scalar offsets (ubase + unit * 48 KiB + quad * 1 KiB, the chunks' DMA offsets) out of the tile loop, runs out of
    // SGPRs (106) and spills them into VGPR lanes — in a kernel that has no VGPR to spare (measured: 39-54 spilled VGPRs).
    int ubase = 0, ubase_next = 0;                       // this item's column block, and the next item's (its first six quads are requested under the last chunk)
    auto tile_scalars = [&]() {
        int z;
        asm volatile("s_mov_b32 %0, 0" : "=s"(z));
        ubase = (nb * NU * 4 + wave) * 12288 + z;      // bytes, wave-uniform
        ubase_next = (nb2 * NU * 4 + wave) * 12288 + z;
        dma_soff0 = z; wave_t = wave + z;
    };
    tile_scalars();
    const int ulane = lane * 16;
    auto b_load = [&](int gu, int q, bool next_item = false) -> f32x4 {   // quad q (0..11, in the order the MFMAs use them) of unit gu
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, ulane, (next_item ? ubase_next : ubase) + gu * (4 * 12288) + q * 1024, 0));
    };

    // ---- row transform of this wave (F(2,3), as k_wino64_conv):  t = x + sgn*y over patch rows
    //   a=0: d0 - d2   a=1: d1 + d2   a=2: d2 - d1   a=3: d1 - d3
    const int xrow = (wave == 0) ? 0 : (wave == 2) ? 2 : 1;
    const int yrow = (wave == 0 || wave == 1) ? 2 : (wave == 2) ? 1 : 3;
    const float sgn = (wave == 1) ? 1.f : -1.f;
    const f32x2* lds2 = reinterpret_cast<const f32x2*>(lds);
    // f32x2 index of (channel group g, patch rows xrow / yrow of tile row tr, x = 4 tc) in buffer 0, plus the 8-byte HALF of the
    // quad this lane reads for the unit being built: half = s2 ^ (g & 1).  A ds_read_b64 is served in two groups of 32 lanes
    // over 64 banks (MI355X_MICROARCH.md, LDS): with every lane on the SAME half of its 16-byte quad the 32 lanes of a group
    // (two channel groups g x 16 tiles) touch only every other 8-byte slot — a 2-way conflict by construction, the 33-41 %
    // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of round 2.  With odd channel groups on the other half the 16 tiles of g even cover
    // the 16 even slots (10 tc mod 32, + 16 tr for TC = 8, + 24 tr for TC = 4) and those of g odd the 16 odd ones: conflict-free.
    // Unit (chunk, s2) therefore holds channels 4g + 2 (s2 ^ (g & 1)) + e2 of the chunk (pack_winograd42_u packs U to match);
    // the half flips at every unit, one v_xor per base.
    int xbase = 2 * (WPS * ((2 * tr + xrow) * RS + tc) + g) + (g & 1), ybase = 2 * (WPS * ((2 * tr + yrow) * RS + tc) + g) + (g & 1);
    auto col_off = [](int c) { return 2 * WPS * ((c & 3) * QS + (c >> 2)); };   // patch column c of the tile: plane c mod 4, pixel tc + c / 4

    // ---- prologue of the workgroup's first tile ----
    // B ring: RING quads of lead between a refill and its use (a divisor of 12, the quads of a unit)
#ifdef CID_W42_RING
    constexpr int RING = CID_W42_RING;
#else
    constexpr int RING = 6;
#endif
    static_assert(RING == 6 || RING == 12, "ring = the quads of one k-step or of one unit");
    f32x4 bq[RING];
#pragma unroll
    for (int q = 0; q < RING; ++q) bq[q] = b_load(0, q);
    dma_chunk(2, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // chunk 0 has landed (this wave); past the barrier: every wave's part
    __syncthreads();

    f32x4 acc[6][4];   // [position b][channel group cg]; first written by the zero-C MFMAs of unit 0
    float vcur[6][2];
    f32x2 rawx, rawy;
    // the next unit's V: the six column sums t[c] = x + sgn*y first, then V0 = t4 + (81/64) t0 - (45/16) t2, V1/V2 = E12 +- O12,
    // V3/V4 = E34 +- O34, V5 = t5 + (81/64) t1 - (45/16) t3 (the +-p rows of B4^T share their even and odd parts): 12 + 28 VALU per unit
    float t[6][2], vnxt[6][2];
    auto read_col = [&](int bufoff, int c) {               // bufoff = 2 * buf * BUF (f32x2 units); the half is in xbase / ybase
        rawx = lds2[bufoff + xbase + col_off(c)];
        rawy = lds2[bufoff + ybase + col_off(c)];
    };
    auto flip_half = [&]() { xbase ^= 1; ybase ^= 1; };   // the next unit to be built reads the other half of every quad
    auto fold_col = [&](int c) {
#pragma unroll
        for (int e = 0; e < 2; ++e) t[c][e] = __builtin_fmaf(sgn, rawy[e], rawx[e]);
    };
    auto make_v = [&](float (&v)[6][2], int part) {       // part 0: b = 0, 1, 2; part 1: b = 3, 4, 5
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            // the odd part of a +-p pair is p (t3 - p^2 t1)... / p^2: one fma for the bracket, and its factor rides on the two fmas that
            // form the pair (round 4: 12 instead of 14 operations per channel; every constant is still an exact dyadic rational)
            if (part == 0) {
                v[0][e] = __builtin_fmaf(1.265625f, t[0][e], __builtin_fmaf(-2.8125f, t[2][e], t[4][e]));
                const float ev = __builtin_fmaf(-2.25f, t[2][e], t[4][e]);
                const float od = __builtin_fmaf(-2.25f, t[1][e], t[3][e]);          // (t3 - 9/4 t1); x 3/4 below
                v[1][e] = __builtin_fmaf(0.75f, od, ev);
                v[2][e] = __builtin_fmaf(-0.75f, od, ev);
            } else {
                const float ev = __builtin_fmaf(-0.5625f, t[2][e], t[4][e]);
                const float od = __builtin_fmaf(-0.5625f, t[1][e], t[3][e]);         // (t3 - 9/16 t1); x 3/2 below
                v[3][e] = __builtin_fmaf(1.5f, od, ev);
                v[4][e] = __builtin_fmaf(-1.5f, od, ev);
                v[5][e] = __builtin_fmaf(1.265625f, t[1][e], __builtin_fmaf(-2.8125f, t[3][e], t[5][e]));
            }
        }
    };
    auto finish_v = [&]() {
#pragma unroll
        for (int b = 0; b < 6; ++b) { vcur[b][0] = vnxt[b][0]; vcur[b][1] = vnxt[b][1]; }
    };
    auto build_first_v = [&]() {                           // V of a tile's unit 0, from buffer X
#pragma unroll
        for (int c = 0; c < 6; ++c) { read_col(2 * XB, c); fold_col(c); }
        make_v(vnxt, 0);
        make_v(vnxt, 1);
        finish_v();
#ifdef CID_W42_ASM_MFMA
#pragma unroll
        for (int b = 0; b < 6; ++b) asm volatile("" : "+v"(vcur[b][0]), "+v"(vcur[b][1]));
        asm volatile("s_nop 3");
#endif
    };
    build_first_v();
    auto next_tile = [&]() {   // at a tile boundary: the prefetched tile becomes the current one; decode its successor and form that one's offsets
        n = n2; y0 = y02; x0 = x02; local = local2; nb = nb2;
        tpar ^= 1;
        rsrc_in = rsrc_next;
        decode_next();
        rsrc_next = image_rsrc(n2);
        write_offsets(tpar ^ 1, y02, x02, has_next);   // `ent` was requested before the epilogue
        flip_half();           // NU - 1 builds flipped the half an odd number of times: back to unit 0's
        build_first_v();
    };


    // Chunk ck in LDS buffer PAR (chunk 0: buffer X): two units (s2 = 0, 1: the two channel pairs of every quad).  Unit = 12 groups of
    // four MFMAs (k-step e2 outer, position b inner, the four channel groups innermost); under them the next unit's V is built.
    auto chunk = [&](auto first_tag, auto more_tag, auto dma_tag, auto parity_tag, int ck) {
        constexpr bool FIRST = decltype(first_tag)::value;    // chunk 0: accumulators start from a zero C operand
        constexpr bool MORE = decltype(more_tag)::value;      // a chunk ck+1 exists; else: the last chunk, under which the next tile's chunk 0 is fetched
        constexpr bool DMA = decltype(dma_tag)::value;        // a chunk ck+2 exists: fetch it into the buffer that is free by then
        constexpr int PAR = decltype(parity_tag)::value ? 1 : 0;
        constexpr int RB = FIRST ? 2 : PAR, OB = FIRST ? 1 : 1 - PAR, TB = FIRST ? 0 : PAR;   // read / other / DMA-target buffer
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool build = ((k == 0) || MORE) && !(ABLATE & 4);
            const int nhalf = (k == 0) ? 2 * boff(RB) : 2 * boff(OB);   // the next unit: this buffer's other half, or the other buffer
            const int gu = ck * 2 + k;
            if (build) flip_half();
            if (FIRST && k == 0 && !(ABLATE & 1)) dma_chunk(1, 1);   // chunk 1 lands under unit 0
#pragma unroll
            for (int grp = 0; grp < 12; ++grp) {
                const int e2 = grp / 6, b = grp - 6 * e2;
                if (build) {   // column grp is read here and folded one group later, under four MFMAs
                    if (grp >= 1 && grp <= 6) fold_col(grp - 1);
                    if (grp < 6) read_col(nhalf, grp);
                    if (grp == 8) make_v(vnxt, 0);
                    if (grp == 10) make_v(vnxt, 1);
                }
                if (DMA && k == 1 && grp >= 6 && grp - 6 < RW && !(ABLATE & 1)) dma_round(rsrc_in, tpar, TB, ck + 2, grp - 6);
                // next tile, chunk 0 -> X.  Only if there is one (workgroup-uniform: a scalar branch): one item per workgroup and a walker's
                // last item used to issue the RW rounds against an all-sentinel offset table
                if (!MORE && k == 1 && grp >= 6 && grp - 6 < RW && !(ABLATE & 1) && has_next) dma_round(rsrc_next, tpar ^ 1, 2, 0, grp - 6);
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) {
#ifdef CID_W42_ASM_MFMA
                    // experiment (profiles/r03_asm_mfma_ring.txt): the accumulator tied to the destination, which hipcc's own MFMA builtins
                    // do not do (it accumulates out of place and needs 248-256 registers for this loop; 201 with this form)
                    if (FIRST && k == 0 && e2 == 0)
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc[b][cg]) : "v"(vcur[b][e2]), "v"(bq[grp % RING][cg]));
                    else
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[b][cg]) : "v"(vcur[b][e2]), "v"(bq[grp % RING][cg]));
#else
                    if (FIRST && k == 0 && e2 == 0) {
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        acc[b][cg] = __builtin_amdgcn_mfma_f32_16x16x4f32(vcur[b][e2], bq[grp % RING][cg], zero, 0, 0, 0);
                    } else {
                        acc[b][cg] = __builtin_amdgcn_mfma_f32_16x16x4f32(vcur[b][e2], bq[grp % RING][cg], acc[b][cg], 0, 0, 0);
                    }
#endif
                }
                // ring slot b: refilled with the quad six uses ahead (the other k-step of this unit, the next unit's first, or — at
                // the end of a tile — the first quads of the next tile: same column block, so the stream simply starts over)
                if (ABLATE & 2) {}
                else if (grp + RING < 12) bq[grp % RING] = b_load(gu, grp + RING);
                else if (MORE || k == 0) bq[grp % RING] = b_load(gu + 1, grp + RING - 12);
                else bq[grp % RING] = b_load(0, grp + RING - 12, true);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (build) finish_v();
#ifdef CID_W42_ASM_MFMA
            // hipcc's hazard recognizer does not look inside inline asm: a VALU result needs two wait states before an MFMA reads it
            // as SrcA.  Pin the next unit's V values here (their producers cannot sink below this point) and pad.
#pragma unroll
            for (int b = 0; b < 6; ++b) asm volatile("" : "+v"(vcur[b][0]), "+v"(vcur[b][1]));
            asm volatile("s_nop 3");
#endif
            if (MORE && k == 0) {
                // every DMA of this wave for chunk ck+1 is older than the last RING B refills (its rounds were issued under the
                // previous unit's groups 6-10, each followed by that group's refill and the twelve of this unit)
                if (RING == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                __syncthreads();
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    for (;;) {   // ---- one tile per iteration ----
    tile_scalars();
    tr_lap(tr_bnd);
    // The main loop runs at wave priority 1, epilogue and tile boundary at 0: when the sibling workgroup's wave on this SIMD is in its
    // epilogue, this wave's MFMA stream goes first and the epilogue's vector instructions take the gaps (same-box +0.85 % on the forward,
    // -2 % on the CIN = 64 layers and upconv1.0; the opposite assignment +-0: profiles/r03_ab_wino42_walk.txt).  Round 2 had measured
    // priority 2 OUTSIDE the main loop as a loss.
    __builtin_amdgcn_s_setprio(1);
    chunk(T{}, T{}, T{}, F{}, 0);
    chunk(F{}, T{}, std::integral_constant<bool, (NCHUNK > 3)>{}, T{}, 1);
    for (int ck = 2; ck + 2 < NCHUNK; ck += 2) {
        chunk(F{}, T{}, T{}, F{}, ck);
        chunk(F{}, T{}, T{}, T{}, ck + 1);
    }
    chunk(F{}, T{}, F{}, F{}, NCHUNK - 2);
    chunk(F{}, F{}, F{}, T{}, NCHUNK - 1);

    tr_lap(tr_main);
    ++tr_tiles;
    __builtin_amdgcn_s_setprio(0);
#ifdef CID_W42_ASM_MFMA
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");   // MFMA write -> VALU read of the accumulators (up to 18 wait states for an 8-pass MFMA)
#endif
    if (ABLATE & 8) {   // keep the accumulators alive without the epilogue
        float sum = 0.f;
#pragma unroll
        for (int b = 0; b < 6; ++b)
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) sum += acc[b][cg][0] + acc[b][cg][1] + acc[b][cg][2] + acc[b][cg][3];
        if (sum == 123.456f) a.out[tid] = sum;
        if (!has_next) break;
        load_slot_entries();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        next_tile();
        continue;
    }
    // ---- output transform ----
    // Everything lane-dependent below is derived from an opaque copy of the lane id made here, per tile: otherwise hipcc hoists the
    // epilogue's address arithmetic (staging and exchange addresses, store offsets) out of the tile loop and keeps it in VGPRs
    // across the main loop, which has none to spare.
    int lane_e;
    asm volatile("v_mov_b32 %0, %1" : "=v"(lane_e) : "v"(tid & 63));
    const int lane = lane_e, m16 = lane_e & 15, g = lane_e >> 4;
    // Wave w finishes tile quarter r = w: the tiles 4g + w (g = lane >> 4) for all four channel groups.  Its bias values are
    // requested here; the column transform and the exchange cover their latency.
    // MFMA column j = m16 of channel group cg is output channel 64 nb + 4 m16 + cg (pack_winograd42_u): a lane's four groups are four
    // CONSECUTIVE channels, so its results leave as 16-byte quads straight from registers — no transposing pass through LDS
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(a.bias + nb * WN2 + 4 * m16);
    if (has_next) load_slot_entries();   // for the offsets formed at the tile boundary; older than the epilogue's stores, so its wait skips them
    // step 1, in registers: mp[cg][r] = (b' = 0..3) = sum_b A4^T[b'][b] acc[b][cg][r]
    f32x4 mp[4][4];
#pragma unroll
    for (int cg = 0; cg < 4; ++cg)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float mm[6] = {acc[0][cg][r], acc[1][cg][r], acc[2][cg][r], acc[3][cg][r], acc[4][cg][r], acc[5][cg][r]};
            float yy[4];
            w42_out4(mm, yy);
            mp[cg][r] = f32x4{yy[0], yy[1], yy[2], yy[3]};
        }
    // step 2: the rows of the three other waves through LDS — block (row a, consumer w != a, cg): 64 lanes of 16 bytes (48 KiB in all) —
    //   Y[0] = (m0 + m1) + m2,  Y[1] = m1 - (m2 + m3)   (the F(2,3) output transform of k_wino64_conv, same order)
    // Output descriptors of this item's image.  Stores go through raw buffer stores: per pass the tile, its row and its first column
    // are compile-time or scalar (they follow from the pass number and the wave), so the address is ONE per-lane offset, formed once per
    // item, plus a scalar offset per store — the round-2 form decoded tile and pixel per lane and per store (~210 integer VALU
    // instructions per item and wave, a quarter of them quarter-rate 32/64-bit multiplies, on ALUs the fp32 MFMA stream shares).
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // 16-byte buffer store with a scalar offset, and its data registers kept untouched for four more wait states.  Found this round:
    // hipcc (ROCm 7.2) lets a VALU instruction overwrite the first data register of a buffer_store_dwordx4 ... s<soffset> offen one
    // instruction after it (LLVM's hazard table exempts the register-soffset form), and gfx950 then stores the NEW value: element 0 of
    // the quad was wrong, run to run, in the pooled tensor (profiles/r03_store_hazard.txt).  The empty asm keeps `v` alive across an s_nop 3.
    auto store16 = [](f32x4 v, const __amdgpu_buffer_rsrc_t& rsrc, unsigned vo, unsigned soff) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, vo, soff, 0);
        asm volatile("s_nop 3" ::"v"(v) : "memory");
    };
    auto out_rsrc = [&](const float* base, size_t elems_before, int elems) {
        const unsigned long long p = (unsigned long long)(base + elems_before);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), (short)0, elems * 4, 0x00020000);
    };
    auto epilogue = [&](auto wave_tag) {
        constexpr int W = decltype(wave_tag)::value;
        // The next tile's chunk 0 (requested under the last chunk, into buffer X) has landed for this wave — the column transform
        // above covered its latency, and that of the last B refills and the bias values — and past the barrier for every wave.
        // (Waiting at the third barrier instead, and forming the next tile's offsets and first V before the boundary barrier: -0.3 %
        // same-box, profiles/r03_ab_wino42_walk.txt.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                     // raw tiles 0 / 1 are dead: their LDS becomes the exchange area
#pragma unroll
        for (int cg = 0; cg < 4; ++cg)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r != W) ldsw[((W * 3 + (r - (r > W ? 1 : 0))) * 4 + cg) * 64 + lane] = mp[cg][r];
        __syncthreads();
        f32x4 Y[4][2];                                       // [cg][a'] = the four b' of output row a'
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) {
            f32x4 m[4];
#pragma unroll
            for (int ar = 0; ar < 4; ++ar) m[ar] = (ar == W) ? mp[cg][W] : ldsw[((ar * 3 + (W - (W > ar ? 1 : 0))) * 4 + cg) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Y[cg][0][e] = (m[0][e] + m[1][e]) + m[2][e];
                Y[cg][1][e] = m[1][e] - (m[2][e] + m[3][e]);
            }
        }
        // step 3: lane (m16, quarter g) holds pixels (a', b') of tile 4g + W for the four consecutive channels 4 m16 + cg.
        const int cbase = nb * WN2;
        const int tile_l = 4 * g + W, ttr_l = tile_l / TC, ttc_l = tile_l - ttr_l * TC;       // this lane's tile inside the workgroup's block
        auto quad = [&](int ap, int bp) {                                                    // bias + ReLU of pixel (a', b'), channels 4 m16 .. + 3
            return f32x4{fmaxf(Y[0][ap][bp] + bias4[0], 0.f), fmaxf(Y[1][ap][bp] + bias4[1], 0.f), fmaxf(Y[2][ap][bp] + bias4[2], 0.f),
                         fmaxf(Y[3][ap][bp] + bias4[3], 0.f)};
        };
        if constexpr (ZOUT) {
            // upconv1[0] as the producer of the last layer's input (see k_wino64_conv): relu(y) of the wave's 32 pixels x 64 channels is
            // staged [quarter g][pixel a'*4 + b'][64 channels] (68-float rows, 16 floats between quarters; one 16-byte write per pixel:
            // sixteen lanes write one pixel's 256 bytes); z[p][3*tap + co] = sum_ci relu(y)[p][ci] * W2[co][ci][tap] is a [32 x 64] x
            // [64 x 32 (27 used)] product, 32 MFMAs of 32x32x2, stored as 27 planes [N, 27, H, W].  MFMA row i = pixel (quarter i >> 3, i & 7).
            __syncthreads();                                 // exchange area is dead: wave-private staging
            constexpr int STR = 68, QSTR = 8 * STR + 16;
            float* stg = reinterpret_cast<float*>(ldsw) + W * (4 * QSTR);
#pragma unroll
            for (int ap = 0; ap < 2; ++ap)
#pragma unroll
                for (int bp = 0; bp < 4; ++bp) *reinterpret_cast<f32x4*>(stg + g * QSTR + (ap * 4 + bp) * STR + 4 * m16) = quad(ap, bp);
            wave_lds_fence();
            const int i32 = lane & 31, h = lane >> 5;
            f32x4 zb[2][4];
#pragma unroll
            for (int ck = 0; ck < 2; ++ck)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) zb[ck][gq] = reinterpret_cast<const f32x4*>(a.zw)[(ck * 4 + gq) * 64 + lane];
            f32x16 zacc;
            const float* src = stg + (i32 >> 3) * QSTR + (i32 & 7) * STR + 4 * h;
#pragma unroll
            for (int ck = 0; ck < 2; ++ck)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(src + 32 * ck + 8 * gq);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (ck == 0 && gq == 0 && e == 0) {
                            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            zacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], zb[ck][gq][e], zero, 0, 0, 0);
                        } else {
                            zacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], zb[ck][gq][e], zacc, 0, 0, 0);
                        }
                    }
                }
            // lane (column j = i32, half h) holds z[pixel (r&3) + 8*(r>>2) + 4*h][j]: registers 4q..4q+3 are the four pixels b' = 0..3
            // of row a' = h of tile 4q + W — one 16-byte store each into plane j
            const __amdgpu_buffer_rsrc_t rz = out_rsrc(a.zout, (size_t)n * 27 * a.Hs * a.Ws, 27 * a.Hs * a.Ws);
            const unsigned zlane = (unsigned)(((i32 * a.Hs + h) * a.Ws) * 4);              // plane i32, row offset h
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int tile = 4 * q + W, ttr = tile / TC, ttc = tile - ttr * TC;          // compile-time
                const int yb = y0 + 2 * ttr, xx = x0 + 4 * ttc;                             // scalar
                // masking lives in the PER-LANE offset only (as in every other store here): whether the scalar offset takes part
                // in the hardware range check is not documented for stores, so nothing may depend on it
                const bool in_x = xx < a.Ws;                                                // scalar
                const unsigned soff = in_x ? (unsigned)((yb * a.Ws + xx) * 4) : 0u;
                const unsigned vo = (in_x && i32 < 27 && yb + h < a.Hs) ? zlane : 0x7ffffff0u;
                store16(f32x4{zacc[4 * q], zacc[4 * q + 1], zacc[4 * q + 2], zacc[4 * q + 3]}, rz, vo, soff);
            }
            return;
        }
        // Stores straight from registers (no third barrier, no staging): one per-lane offset per item — the lane's tile and channel
        // quad — plus a scalar offset per pixel (a', b'); a wave instruction writes four pixels' 256 bytes.
        {
            const bool full = y0 + 2 * TRW <= a.Hs && x0 + 4 * TC <= a.Ws;                    // workgroup-uniform: no ragged edge in this block
            const int yl = 2 * ttr_l, xl = 4 * ttc_l;
            const __amdgpu_buffer_rsrc_t ro = out_rsrc(a.out, (size_t)n * a.Hs * a.Ws * a.out_ps, a.Hs * a.Ws * a.out_ps);
            const unsigned lane_off = (unsigned)((__umul24((unsigned)(__umul24((unsigned)yl, (unsigned)a.Ws) + xl), (unsigned)a.out_ps) + a.out_coff + cbase + 4 * m16) * 4);
#pragma unroll
            for (int ap = 0; ap < 2; ++ap)
#pragma unroll
                for (int bp = 0; bp < 4; ++bp) {
                    const unsigned soff = (unsigned)((((y0 + ap) * a.Ws + x0 + bp) * a.out_ps) * 4);      // scalar
                    const unsigned vo = (full || (y0 + yl + ap < a.Hs && x0 + xl + bp < a.Ws)) ? lane_off : 0x7ffffff0u;
                    store16(quad(ap, bp), ro, vo, soff);
                }
        }
        if (POOL) {
            const int Hp = a.Hc >> 1, Wp = a.Wc >> 1;
            const __amdgpu_buffer_rsrc_t rp = out_rsrc(a.pool, (size_t)n * Hp * Wp * COUT, Hp * Wp * COUT);
            const unsigned plane_off = (unsigned)(((__umul24((unsigned)ttr_l, (unsigned)Wp) + 2 * ttc_l) * COUT + cbase + 4 * m16) * 4);
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                f32x4 v;
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) {
                    const float mx = fmaxf(fmaxf(Y[cg][0][2 * pb], Y[cg][0][2 * pb + 1]), fmaxf(Y[cg][1][2 * pb], Y[cg][1][2 * pb + 1]));
                    v[cg] = fmaxf(mx + bias4[cg], 0.f);
                }
                const unsigned soff = (unsigned)((((y0 >> 1) * Wp + (x0 >> 1) + pb) * COUT) * 4);        // scalar
                const unsigned vo = ((y0 >> 1) + ttr_l < Hp && (x0 >> 1) + 2 * ttc_l + pb < Wp) ? plane_off : 0x7ffffff0u;
                store16(v, rp, vo, soff);
            }
        }
    };
    switch (wave) {   // four code versions: "is this my own row" is a compile-time fact
        case 0: epilogue(std::integral_constant<int, 0>{}); break;
        case 1: epilogue(std::integral_constant<int, 1>{}); break;
        case 2: epilogue(std::integral_constant<int, 2>{}); break;
        default: epilogue(std::integral_constant<int, 3>{}); break;
    }
    // Barrier discipline of the four instantiations above: each executes the same number of __syncthreads() on every path — two
    // (three in the ZOUT variant, whose `return` leaves the lambda after the third).  s_barrier is not PC-matched, so waves meeting at
    // different program counters is what the hardware does anyway; what must hold — and does, by construction of the one lambda
    // body — is the equal COUNT.
    tr_lap(tr_epi);
    if (!has_next) break;                                     // workgroup-uniform
    __syncthreads();   // every wave has left the exchange / staging area: buffers 0 / 1 may be written again
    next_tile();
    }   // ---- next tile ----
    if ((ABLATE & 256) && tid == 0) {
        trace[0] = tr_start; trace[1] = tr_main; trace[2] = tr_epi; trace[3] = tr_bnd; trace[5] = tr_tiles;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        trace[4] = __builtin_readcyclecounter();
    }
}

}  // namespace cid
