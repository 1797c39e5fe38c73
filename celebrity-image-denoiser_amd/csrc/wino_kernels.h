// wino_kernels.h — 3x3 convolution (+bias, +ReLU, +optional 2x2 max-pool) as Winograd F(2x2,3x3)
// on the exact-f32 matrix instruction v_mfma_f32_32x32x2_f32 (gfx950).
//
// Computes the same function as the reference's nn.Conv2d(k=3, p=1) + nn.ReLU (+ nn.MaxPool2d(2,2))
// stages (backend/app.py:43-77) with 16 multiplies per 2x2 output tile and (ci, co) pair instead of
// 36: for each of the 16 positions xi = (a, b) of the transformed 4x4 tile,
//       M_xi[tile][co] = sum_ci V_xi[tile][ci] * U_xi[ci][co],     V = B^T d B,  U = G g G^T,
//       Y(2x2) = A^T M A.
// U is computed on the host at load time (cid_api.hip, in double, rounded once).  V is never stored:
// each lane rebuilds the four V values of a row `a` from eight 16-byte LDS reads of the raw input tile
// (adds only), so the MFMA A operand costs the same LDS traffic as a pre-transformed image would.
//
// Work decomposition (256 threads = 4 waves = 2 pairs; 2 workgroups per CU):
//   * a pair owns 32 tiles (TC tile-columns x 32/TC tile-rows) x 32 output channels;
//   * the two waves of a pair split the 16 positions by row a: wave `half` accumulates a in {2*half,
//     2*half+1}, i.e. 8 accumulator tiles = 128 VGPRs, and runs units of 16 MFMAs
//     (one row a, 8 input channels: 4 positions b x 4 k-steps);
//   * K is walked in 16-channel chunks through a double-buffered raw halo tile in LDS
//     (pixel = 4 data slots + 1 pad slot of 16 B); the next chunk is written by LDS-DMA
//     (buffer_load_dwordx4 ... lds: no VGPRs, no ds_write; out-of-image and pad slots carry an out-of-range offset and the
//     buffer range check writes zeros)
//     under the current chunk's MFMAs, ONE barrier per chunk;
//   * B fragments (U, pre-packed per lane) stream L2 -> registers two units ahead;
//   * epilogue: the halves exchange their partial output transforms through LDS, then half h writes
//     output row 2*tile_row + h (and half 0 the pooled row): bias, ReLU, 16-byte stores via LDS.
#pragma once
#include "conv_kernels.h"

namespace cid {

constexpr int WK = 16;        // channels per chunk
constexpr int WPS = 5;        // LDS slots (16 B) per pixel: 4 data + 1 pad
constexpr int WN = 32;        // output channels per workgroup
constexpr int WS32 = 36;      // staging row stride (floats) for 32-channel slabs

// Host: the slot table of k_wino_conv<.., TC>: LDS slot s (16 B) of the raw halo tile -> packed (row, column, group).
// Must mirror the kernel's LDS order: pixel = s/5 (4 data slots + 1 pad), rows of LWS pixels, even columns then odd.
// BTR = tile rows per workgroup: 2*(32/TC) for k_wino_conv, 32/TC for k_wino64_conv.
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
    const float* u;     // packed U: [nb][chunk][round][a][e][lane][b]  (cid_api.hip pack_winograd_u)
    const float* bias;  // [COUT]
    const unsigned* slot_tab;   // per LDS slot of the raw tile: (row << 20 | column << 8 | channel group), ~0u = deliver zeros (host: wino_slot_table)
    float* out;         // [N, Hs, Ws, out_ps] (+ out_coff)
    float* pool;        // POOL: [N, Hc/2, Wc/2, COUT]
    int N, Hin, Win, in_ps;
    int Hc, Wc, Hs, Ws;
    int out_ps, out_coff;
    int tiles_x, tiles_y, tiles_total, tiles_per_xcd;
    unsigned rcp_x, rcp_xy;   // ceil(2^32 / tiles_x), ceil(2^32 / (tiles_x*tiles_y)): division by multiply-high (host: tile_rcp)
};

// ABLATE (timing experiments only, csrc/tools/layer_bench.hip; results are wrong when non-zero):
//   bit 0: no halo prefetch after chunk 0   bit 1: B fragments loaded once   bit 2: A operand built once
//   bit 3: no epilogue                      bit 4: A operand read from LDS but not transformed
//   bit 5: no prologue DMA                  bit 6: no start-up stagger of odd wave slots
template <int CIN, int COUT, bool POOL, int TC, int ABLATE>
__device__ __forceinline__ void wino_item(const WinoArgs& a, f32x4* lds, int item);

template <int CIN, int COUT, bool POOL, int TC, int ABLATE = 0>
__global__ void __launch_bounds__(THREADS, 2) k_wino_conv(const WinoArgs a) {
    constexpr int LDS_SLOTS_K = 4096;
    __shared__ f32x4 lds[LDS_SLOTS_K];
    if (ABLATE & 128) {   // experiment: two work items per workgroup, back to back (grid halved by the launcher)
        wino_item<CIN, COUT, POOL, TC, ABLATE>(a, lds, blockIdx.x);
        __syncthreads();
        wino_item<CIN, COUT, POOL, TC, ABLATE>(a, lds, blockIdx.x + gridDim.x);
    } else {
        wino_item<CIN, COUT, POOL, TC, ABLATE>(a, lds, blockIdx.x);
    }
}

template <int CIN, int COUT, bool POOL, int TC, int ABLATE>
__device__ __forceinline__ void wino_item(const WinoArgs& a, f32x4* lds, int item) {
    constexpr int TRP = 32 / TC;                 // tile rows per pair
    constexpr int BTR = 2 * TRP;                 // tile rows per workgroup
    constexpr int LW = 2 * TC + 2, LH = 2 * BTR + 2;
    // LDS row stride in pixels.  TC=16 puts two tile rows in one 32-lane read; their slot offset (2 rows x LWS x 5
    // slots) must be a multiple of 16 slots or the two half-rows collide in ds_read_b128's bank columns: 34 -> 40.
    constexpr int LWS = (TC == 16) ? 40 : LW;
    constexpr int LPIX = LWS * LH;
    constexpr int NROUND = (LPIX * WPS + 63) / 64;   // LDS-DMA wave-instructions (64 slots of 16 B) per buffer
    constexpr int RW = (NROUND + 3) / 4;             // rounds per wave
    constexpr int BUF = NROUND * 64;                 // f32x4 slots per LDS buffer (padded to whole rounds)
    constexpr int NCHUNK = CIN / WK;
    constexpr int NB = COUT / WN;
    static_assert(CIN % WK == 0 && COUT % WN == 0 && (TC == 16 || TC == 32), "layer dims");
    static_assert(RW <= 8, "DMA rounds per wave");

    constexpr int EXCH = 4 * 16 * 64;            // epilogue exchange area: 4 waves x 16 registers x 64 lanes (f32x4)
    static_assert(2 * BUF <= 4096 && EXCH <= 4096 && 4 * 64 * WS32 * sizeof(float) <= 4096 * 16, "LDS budget (64 KiB)");

    int mt, nb;
    if (!decode_block(a.tiles_total, a.tiles_per_xcd, NB, mt, nb, item)) return;
    int n, ty, tx;
    decode_tile(mt, a.tiles_x, a.tiles_y, a.rcp_x, a.rcp_xy, n, ty, tx);
    const int y0 = ty * (2 * BTR), x0 = tx * (2 * TC);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform in an SGPR
    const int pair = wave >> 1, half = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    const int tr = i / TC, tc = i - tr * TC;

    const float bias_v = a.bias[nb * WN + i];

    // De-phase the two workgroups that share a CU.  All workgroups of this launch take the same time, so the two
    // resident on a CU would start, reach their seams and finish together for the whole launch, and every
    // prologue/epilogue (LDS-DMA latency, output transform, stores) would find the SIMD's other wave in the
    // same idle phase.  The waves that landed in an odd wave slot of their SIMD (HW_ID.WAVE_ID) wait half a
    // workgroup's duration once, in the first generation only; later generations inherit the offset.
    if (!(ABLATE & 64) && item < 2 * 256) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if (hwid & 1u) {
            for (int sl = 0; sl < NCHUNK / 2; ++sl) __builtin_amdgcn_s_sleep(127);   // ~NCHUNK*4096 cycles
        }
    }

    // rows of the 4x4 input patch that feed row a of B^T d:  t = x + sgn*y
    //   a=0: d0 - d2   a=1: d1 + d2   a=2: d2 - d1   a=3: d1 - d3
    const int xrow0 = half ? 2 : 0, yrow0 = half ? 1 : 2;
    const int xrow1 = 1, yrow1 = half ? 3 : 2;
    const float sgn0 = -1.f, sgn1 = half ? -1.f : 1.f;
    // LDS pixel order inside a tile row: even columns first, then odd columns (column hx at (hx&1)*LW/2 + hx/2).
    // The 32 lanes of a read want columns 2*tc + c: with this order they are CONSECUTIVE pixels (stride 5 slots),
    // which ds_read_b128 serves conflict-free; in natural order they are 2 pixels apart = a 2-way bank conflict.
    constexpr int HWD = LWS / 2;
    const int pbase = (2 * (pair * TRP + tr)) * LWS + tc;
    const int xb0 = (pbase + xrow0 * LWS) * WPS + h, yb0 = (pbase + yrow0 * LWS) * WPS + h;
    const int xb1 = (pbase + xrow1 * LWS) * WPS + h, yb1 = (pbase + yrow1 * LWS) * WPS + h;
    auto col_off = [](int c) { return ((c & 1) * HWD + (c >> 1)) * WPS; };   // patch column c -> slot offset

    // ---- LDS-DMA sources.  Round j of a buffer fills slots [64j, 64j+64); wave w issues rounds w, w+4, ...
    // Lane l of round j owns slot s = 64j + l = pixel s/5, 16-byte group s%5 (group 4 = pad).  The source is a
    // raw buffer over this image: a lane's 32-bit offset addresses its pixel's first 4 channels, the chunk is a
    // scalar offset, and lanes that must deliver zeros (padding of the convolution, pad slots, slots past the
    // tile) carry an out-of-range offset — the buffer range check makes the DMA write zeros for them.  No vector
    // ALU work per chunk (VALU instructions beside the MFMA stream cost matrix-pipe time: tools/mix_bench).
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
    const unsigned lds_base = (unsigned)(uintptr_t)(&lds[0]);
    auto dma_rounds = [&](int buf, int m0, int m1, int ck) {   // rounds [m0, m1) of this wave, chunk ck -> buffer `buf`
        const int soff = ck * (WK * 4);
#pragma unroll
        for (int m = m0; m < m1; ++m) {
            if (wave + 4 * m < NROUND) {               // wave-uniform
                const unsigned dst = lds_base + (unsigned)((buf * BUF + (wave + 4 * m) * 64) * 16);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff[m]), "s"(rsrc_in), "s"(dst), "s"(soff) : "memory");
            }
        }
    };

    f32x16 acc[2][4];   // first written by the zero-C MFMAs of chunk 0 (no initialisation pass)

    // U stream of this wave: unit (ck, g2, u), k-step e is the 1 KiB quad at ((ck*2+g2)*16 + (2*half+u)*4 + e)*1024 bytes
    // of this column block; [nb][chunk][round][a][e][lane][b].  Raw buffer loads: lane offset in a VGPR once, everything
    // else scalar.
    const __amdgpu_buffer_rsrc_t rsrc_u = __builtin_amdgcn_make_buffer_rsrc((void*)a.u, (short)0, CIN * COUT * 16 * 4, 0x00020000);
    const int ubase = (nb * NCHUNK * 2 * 16 + 2 * half * 4) * 1024;   // bytes, wave-uniform
    const int ulane = lane * 16;
    auto b_load = [&](int gunit, int e) -> f32x4 {          // gunit = ck*4 + g2*2 + u, this wave's unit counter
        const int soff = ubase + (((gunit >> 1) * 4 + (gunit & 1)) * 4 + e) * 1024;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_u, ulane, soff, 0));
    };

    // ---- prologue: chunk 0 -> LDS buffer 0 by DMA; B of units 0 and 1 ----
    if (!(ABLATE & 32)) dma_rounds(0, 0, RW, 0);
    f32x4 bq[2][4];                                        // ring: unit k uses bq[k&1][e]
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int e = 0; e < 4; ++e) bq[d][e] = b_load(d, e);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the DMA is invisible to hipcc's own wait counting
    __syncthreads();

    // A operand of one unit (row a = 2*half+u of B^T d B, 4 channels): eight 16-byte reads of the raw
    // patch, t = x + sgn*y per column (two columns at a time, to keep few reads in flight), then the four
    // column combinations.
    auto read_cols = [&](f32x4 (&xq)[2], f32x4 (&yq)[2], int bufbase, int k, int c0) {
        const int g2 = k >> 1, u = k & 1;
        const int xb = bufbase + (u ? xb1 : xb0) + 2 * g2, yb = bufbase + (u ? yb1 : yb0) + 2 * g2;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            xq[c] = lds[xb + col_off(c0 + c)];
            yq[c] = lds[yb + col_off(c0 + c)];
        }
    };
    auto make_t = [&](f32x4 (&t)[4], const f32x4 (&xq)[2], const f32x4 (&yq)[2], int k, int c0) {
        const float sg = (k & 1) ? sgn1 : sgn0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (ABLATE & 16) { t[c0 + c] = xq[c]; t[c0 + c][0] += yq[c][1]; continue; }
#pragma unroll
            for (int e = 0; e < 4; ++e) t[c0 + c][e] = __builtin_fmaf(sg, yq[c][e], xq[c][e]);
        }
    };
    auto make_v = [&](f32x4 (&v)[4], const f32x4 (&t)[4]) {
        if (ABLATE & 16) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = t[c];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {   // element-wise on purpose: whole-vector +/- lowers to v_pk_add_f32, which costs
                v[0][e] = t[0][e] - t[2][e];   // the matrix pipe ~2.6x a plain VALU op (tools/mix_bench)
                v[1][e] = t[1][e] + t[2][e];
                v[2][e] = t[2][e] - t[1][e];
                v[3][e] = t[1][e] - t[3][e];
            }
        }
    };

    f32x4 vcur[4], vnxt[4];
    {
        f32x4 xq[2], yq[2], t[4];
        read_cols(xq, yq, 0, 0, 0);
        make_t(t, xq, yq, 0, 0);
        read_cols(xq, yq, 0, 0, 2);
        make_t(t, xq, yq, 0, 2);
        make_v(vcur, t);
    }

    // Chunk ck (LDS buffer ck&1).  Unit k runs 16 MFMAs (k-step e outer, position b inner: four independent
    // accumulators in rotation) and, under them:
    //   * reads and builds the A operand of the NEXT unit (unit 3: unit 0 of the next chunk, other buffer);
    //   * refills B quad e, two units ahead, as soon as the four MFMAs of k-step e have issued;
    //   * units 0 and 1 start the LDS-DMA of the next chunk into the other buffer (free since the previous
    //     barrier).  vmcnt retires in order: the only loads queued behind a DMA are B refills that are not
    //     consumed until two units later.
    // One barrier, at the end of unit 2, behind `s_waitcnt vmcnt(8)`: the eight B refills of units 1 and 2 are
    // the only vector-memory operations younger than the last DMA, so at most 8 outstanding means every DMA
    // of this wave has landed; past the barrier every wave's has, and nobody reads this buffer again.
    auto chunk = [&](auto first_tag, auto has_next_tag, auto parity_tag, int ck) {
        constexpr bool FIRST = decltype(first_tag)::value;         // chunk 0: accumulators start from a zero C operand
        constexpr bool NEXT = decltype(has_next_tag)::value && !(ABLATE & 1);
        constexpr int PAR = decltype(parity_tag)::value ? 1 : 0;   // LDS buffer of this chunk, compile-time: immediates
        constexpr int cur = (ABLATE & 1) ? 0 : PAR * BUF;
        constexpr int nxt = (ABLATE & 1) ? 0 : BUF - cur;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = k & 1;
            const bool have_next_unit = ((k < 3) || decltype(has_next_tag)::value) && !(ABLATE & 4);
            const int nbuf = (k < 3) ? cur : nxt, nk = (k + 1) & 3;
            f32x4 xq[2], yq[2], t[4];
            if (NEXT && k == 0) dma_rounds(1 - PAR, 0, RW / 2, ck + 1);
            if (NEXT && k == 1) dma_rounds(1 - PAR, RW / 2, RW, ck + 1);
            if (have_next_unit) read_cols(xq, yq, nbuf, nk, 0);
            const bool refill = !(ABLATE & 2) && ((k < 2) || decltype(has_next_tag)::value);
            // hipcc otherwise sinks the LDS reads of the next unit's A operand to the END of this unit (shorter live
            // ranges) and the whole read -> transform chain lands between two units, in front of the next MFMA.
            // The scheduling fences pin: reads first, transforms spread under the four MFMA groups.
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (have_next_unit) {
                    if (e == 1) { make_t(t, xq, yq, nk, 0); read_cols(xq, yq, nbuf, nk, 2); }
                    if (e == 2) {
                        make_t(t, xq, yq, nk, 2);
#pragma unroll
                        for (int q = 0; q < 4; ++q) { vnxt[0][q] = t[0][q] - t[2][q]; vnxt[1][q] = t[1][q] + t[2][q]; }
                    }
                    if (e == 3) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { vnxt[2][q] = t[2][q] - t[1][q]; vnxt[3][q] = t[1][q] - t[3][q]; }
                    }
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (FIRST && k < 2 && e == 0) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[u][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(vcur[b][e], bq[(ABLATE & 2) ? 0 : (k & 1)][e][b], zero, 0, 0, 0);
                    } else {
                        acc[u][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(vcur[b][e], bq[(ABLATE & 2) ? 0 : (k & 1)][e][b], acc[u][b], 0, 0, 0);
                    }
                }
                if (refill) bq[k & 1][e] = b_load(ck * 4 + k + 2, e);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (have_next_unit) {
#pragma unroll
                for (int b = 0; b < 4; ++b) vcur[b] = vnxt[b];
            }
            if (NEXT && k == 2) {
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                __syncthreads();
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    static_assert(NCHUNK % 2 == 0 && NCHUNK >= 4, "chunks are walked in (even, odd) buffer pairs");
    chunk(std::true_type{}, std::true_type{}, std::false_type{}, 0);
    chunk(std::false_type{}, std::true_type{}, std::true_type{}, 1);
    for (int ck = 2; ck + 2 < NCHUNK; ck += 2) {
        chunk(std::false_type{}, std::true_type{}, std::false_type{}, ck);
        chunk(std::false_type{}, std::true_type{}, std::true_type{}, ck + 1);
    }
    chunk(std::false_type{}, std::true_type{}, std::false_type{}, NCHUNK - 2);
    chunk(std::false_type{}, std::false_type{}, std::true_type{}, NCHUNK - 1);

    // ---- output transform.  m'[u][b'] = sum_b M[a][b] A[b][b'] for the two rows of this wave ----
    // A^T = [[1,1,1,0],[0,1,-1,-1]]; partial P[a'][b'] = sum over own rows a of A^T[a'][a] m'[a][b']
    //   half 0 (a=0,1): P[0] = m'0 + m'1, P[1] = m'1        half 1 (a=2,3): P[0] = m'0, P[1] = -m'0 - m'1
    if (ABLATE & 8) {   // keep the accumulators alive without the epilogue
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[u][b][r];
        if (sum == 123.456f) a.out[tid] = sum;
        return;
    }
    // Two code versions, selected by a wave-uniform branch, so that everything that depends on which half of the
    // rows this wave holds is a compile-time constant (no selects, no multiplies by 0/+-1).
    //   half 0 holds rows a = 0,1:  P[0] = m'0 + m'1 (its own output row), P[1] = m'1        (the partner's)
    //   half 1 holds rows a = 2,3:  P[0] = m'0       (the partner's),      P[1] = -(m'0+m'1) (its own row)
    auto epilogue = [&](auto half_tag) {
        constexpr bool H1 = decltype(half_tag)::value;
        __syncthreads();                                    // raw tiles are dead: LDS becomes exchange + staging
        float yrow[2][16];   // this wave's output row a' = half: columns b' = 0,1 of each tile
        float pooled[16];
        auto mprime = [&](int u, int r, float& m0, float& m1) {
            m0 = acc[u][0][r] + acc[u][1][r] + acc[u][2][r];
            m1 = acc[u][1][r] - acc[u][2][r] - acc[u][3][r];
        };
        if (POOL) {
            // the pooled value needs all four outputs of a tile: exchange both rows of the partial transform
            f32x4* ex = lds + wave * (16 * 64);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float m00, m01, m10, m11;
                mprime(0, r, m00, m01);
                mprime(1, r, m10, m11);
                f32x4 p;
                if (!H1) { p[0] = m00 + m10; p[1] = m01 + m11; p[2] = m10; p[3] = m11; }
                else     { p[0] = m00; p[1] = m01; p[2] = -(m00 + m10); p[3] = -(m01 + m11); }
                ex[r * 64 + lane] = p;
                acc[0][0][r] = p[0]; acc[0][1][r] = p[1]; acc[0][2][r] = p[2]; acc[0][3][r] = p[3];
            }
            __syncthreads();
            const f32x4* exo = lds + (wave ^ 1) * (16 * 64);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const f32x4 o = exo[r * 64 + lane];
                // Y = (half 0's partial) + (half 1's partial), in that order on both waves: bit-identical Y
                const float y00 = H1 ? o[0] + acc[0][0][r] : acc[0][0][r] + o[0];
                const float y01 = H1 ? o[1] + acc[0][1][r] : acc[0][1][r] + o[1];
                const float y10 = H1 ? o[2] + acc[0][2][r] : acc[0][2][r] + o[2];
                const float y11 = H1 ? o[3] + acc[0][3][r] : acc[0][3][r] + o[3];
                yrow[0][r] = fmaxf((H1 ? y10 : y00) + bias_v, 0.f);
                yrow[1][r] = fmaxf((H1 ? y11 : y01) + bias_v, 0.f);
                pooled[r] = fmaxf(fmaxf(fmaxf(y00, y01), fmaxf(y10, y11)) + bias_v, 0.f);
            }
        } else {
            // each half only needs the partner's partial of ITS output row: 8 bytes per value pair
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2* ex = reinterpret_cast<f32x2*>(lds) + wave * (16 * 64);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float m00, m01, m10, m11;
                mprime(0, r, m00, m01);
                mprime(1, r, m10, m11);
                f32x2 give;
                if (!H1) { give[0] = m10; give[1] = m11; acc[0][0][r] = m00 + m10; acc[0][1][r] = m01 + m11; }   // keep P[0]
                else     { give[0] = m00; give[1] = m01; acc[0][0][r] = m00 + m10; acc[0][1][r] = m01 + m11; }   // keep -P[1]
                ex[r * 64 + lane] = give;
            }
            __syncthreads();
            const f32x2* exo = reinterpret_cast<const f32x2*>(lds) + (wave ^ 1) * (16 * 64);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const f32x2 o = exo[r * 64 + lane];
                // half 0: Y[0] = P0[0] + P1[0] = keep + o;   half 1: Y[1] = P0[1] + P1[1] = o + (-(m'0+m'1)) = o - keep
                yrow[0][r] = fmaxf((H1 ? o[0] - acc[0][0][r] : acc[0][0][r] + o[0]) + bias_v, 0.f);
                yrow[1][r] = fmaxf((H1 ? o[1] - acc[0][1][r] : acc[0][1][r] + o[1]) + bias_v, 0.f);
            }
        }
        __syncthreads();                                    // exchange area is dead: reuse as store staging
        float* stg = reinterpret_cast<float*>(lds) + wave * (64 * WS32);
        auto tile_of = [&](int r) { return (r & 3) + 8 * (r >> 2) + 4 * h; };   // D row = tile index of register r
        constexpr int hrow = H1 ? 1 : 0;
        {
            // staged pixels 0..63 are, in order, the 2*TC pixels of this wave's output row (TC=16: of its two rows)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = tile_of(r);
                stg[(2 * t) * WS32 + i] = yrow[0][r];
                stg[(2 * t + 1) * WS32 + i] = yrow[1][r];
            }
            wave_lds_fence();
            // Interior tiles (workgroup-uniform test) store with a wave-uniform row pointer plus one per-lane offset,
            // bumped by scalars: no per-pixel bounds or address arithmetic on the vector ALU.
            const bool full = (y0 + 2 * BTR <= a.Hs) && (x0 + 2 * TC <= a.Ws);
            if (full) {
                const int lane_off = (lane >> 3) * a.out_ps + (lane & 7) * 4;
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int sp0 = it * 8;
                    const int ttr = (sp0 >> 1) / TC, xin = sp0 - ttr * 2 * TC;     // compile-time after unrolling
                    const int y = y0 + 2 * (pair * TRP + ttr) + hrow;
                    float* rowp = a.out + ((size_t)(n * a.Hs + y) * a.Ws + x0 + xin) * a.out_ps + a.out_coff + nb * WN;   // uniform
                    const f32x4 v = *reinterpret_cast<const f32x4*>(stg + (sp0 + (lane >> 3)) * WS32 + (lane & 7) * 4);
                    *reinterpret_cast<f32x4*>(rowp + lane_off) = v;
                }
            } else {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int sp = it * 8 + (lane >> 3);
                    const int t = sp >> 1, ttr = t / TC, ttc = t - ttr * TC;
                    const int y = y0 + 2 * (pair * TRP + ttr) + hrow, x = x0 + 2 * ttc + (sp & 1);
                    const f32x4 v = *reinterpret_cast<const f32x4*>(stg + sp * WS32 + (lane & 7) * 4);
                    if (y < a.Hs && x < a.Ws)
                        *reinterpret_cast<f32x4*>(a.out + ((size_t)(n * a.Hs + y) * a.Ws + x) * a.out_ps + a.out_coff + nb * WN + (lane & 7) * 4) = v;
                }
            }
            wave_lds_fence();
        }
        if (POOL && !H1) {
            const int Hp = a.Hc >> 1, Wp = a.Wc >> 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) stg[tile_of(r) * WS32 + i] = pooled[r];
            wave_lds_fence();
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int t = it * 8 + (lane >> 3), ttr = t / TC, ttc = t - ttr * TC;
                const int py = (y0 >> 1) + pair * TRP + ttr, px = (x0 >> 1) + ttc;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stg + t * WS32 + (lane & 7) * 4);
                if (py < Hp && px < Wp)
                    *reinterpret_cast<f32x4*>(a.pool + ((size_t)(n * Hp + py) * Wp + px) * COUT + nb * WN + (lane & 7) * 4) = v;
            }
        }
    };
    if (half) epilogue(std::true_type{}); else epilogue(std::false_type{});
}

}  // namespace cid
