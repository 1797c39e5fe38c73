// conv_kernels.h — gfx950 (CDNA4) device kernels of the denoise forward.
//
// What the reference computes here: DenoiseGenerator.forward, backend/app.py:80-103 — ten 3x3
// convolutions (+bias, +ReLU), two 2x2 max-pools, two 2x2/stride-2 transposed convolutions,
// two channel concats and a tanh.  None of this is a translation of reference code (the
// reference only calls torch.nn modules); it is written for the MI355X execution model:
//
//   * activations live in HBM as NHWC fp32, so one pixel's channels are one contiguous line;
//   * every GEMM-shaped layer (Cin,Cout multiples of 32/64) is an implicit GEMM on the exact-f32
//     matrix instruction v_mfma_f32_32x32x2_f32: M = 8x32 output pixels per workgroup, N = 64
//     output channels, K walked as (32-channel chunk) x (3x3 tap) x (8-channel group);
//   * the input halo tile (10x34 pixels x 32 channels) is staged once per chunk in LDS in 16-byte
//     slots, pixels padded to 144 B so every ds_read_b128 of an A fragment is conflict-free;
//   * the B (weight) fragments are pre-packed on the host in exactly the per-lane order the MFMA
//     wants and streamed L2 -> registers as 1 KiB coalesced wave loads, one step ahead;
//   * bias, ReLU, the 2x2 max-pool, the concat (a channel-slice store) and the transposed
//     convolution's pixel scatter are all epilogues of the producing kernel — no pool, cat or
//     copy kernel exists;
//   * the 3-channel head (Cin=3, K=27) uses the same MFMA tile on a planar LDS image, and the
//     3-channel tail (Cout=3) runs as a 27-column (tap x cout) MFMA product plus a 9-way shifted sum.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace cid {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int TILE_H = 8;    // output rows per workgroup
constexpr int TILE_W = 32;   // output columns per workgroup (= one MFMA M dimension)
constexpr int KCHUNK = 32;   // input channels staged in LDS per pass
constexpr int NTILE = 64;    // output channels per workgroup
constexpr int THREADS = 256; // 4 waves: wave w owns output rows 2w, 2w+1 of the tile

struct GemmConvArgs {
    const float* in;    // NHWC activations [N, Hin, Win, in_ps]
    const float* w;     // packed weights (pack_gemm_weights in weights_pack.h)
    const float* bias;  // [COUT]
    float* out;         // MODE 0/1: [N, Hs, Ws, out_ps]; MODE 2: [N, 2*Hc, 2*Wc, out_ps]
    float* pool;        // MODE 1: [N, Hc/2, Wc/2, COUT]
    int N, Hin, Win, in_ps;
    int Hc, Wc;         // region of output pixels this launch computes (<= Hin, Win)
    int Hs, Ws;         // region stored to `out` (<= Hc, Wc) — the top-left crop of a skip tensor
    int out_ps, out_coff;
    int tiles_x, tiles_y, tiles_total, tiles_per_xcd;
    unsigned rcp_x, rcp_xy;   // ceil(2^32 / tiles_x), ceil(2^32 / (tiles_x*tiles_y)): division by multiply-high (host: tile_rcp)
};

// 16-byte LDS slot of (tile pixel p, 4-channel group c in 0..7): pixels are padded from 8 to 9
// slots (144 B).  A wave's ds_read_b128 of one A fragment touches 32 consecutive pixels at one c
// per half; with a 36-dword pixel stride the 16 lanes of every ds_read_b128 service group land
// on 16 different 4-bank columns (9p mod 16 is a bijection), i.e. conflict-free, and every
// (tap, channel-group) address is base + immediate — no per-step address arithmetic.
constexpr int PSLOTS = 9;
__device__ __forceinline__ int lds_slot(int p, int c) { return p * PSLOTS + c; }

// XCD-aware decode of blockIdx.x -> (M tile, N block): workgroups are dealt round-robin to the
// 8 XCDs, so id%8 labels the XCD group; each group walks a contiguous range of tiles (whole
// images, neighbouring halos) and the NB column blocks of one tile run back to back on it, so
// the halo tile and the layer's weights stay in that XCD's L2.  Placement only affects speed.
__device__ __forceinline__ bool decode_block(int tiles_total, int tiles_per_xcd, int NB, int& mt, int& nb, int id = -1) {
    if (id < 0) id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    nb = slot % NB;
    mt = xcd * tiles_per_xcd + slot / NB;
    return mt < tiles_total && slot / NB < tiles_per_xcd;
}


// M tile index -> (image, tile row, tile column).  A 32-bit integer division costs ~40 VALU instructions per operand
// pair, and VALU instructions beside the matrix pipe are not free (tools/mix_bench): the host passes
// rcp = ceil(2^32 / d) and the quotient is one multiply-high, exact while value * d < 2^32 (checked on the host).
__host__ __device__ inline unsigned tile_rcp(unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }
__device__ __forceinline__ void decode_tile(int mt, int tiles_x, int tiles_y, unsigned rcp_x, unsigned rcp_xy, int& n, int& ty, int& tx) {
    const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane(mt);
    const unsigned txy = (unsigned)(tiles_x * tiles_y);
    const unsigned nn = rcp_xy ? __umulhi(m, rcp_xy) : m;          // rcp == 0 encodes divisor 1
    const unsigned rem = m - nn * txy;
    const unsigned yy = rcp_x ? __umulhi(rem, rcp_x) : rem;
    n = (int)nn; ty = (int)yy; tx = (int)(rem - yy * (unsigned)tiles_x);
}

// ---- wide store tail -------------------------------------------------------------------------
// An MFMA accumulator tile holds, per lane, ONE output channel and 16 pixels, so storing it
// directly costs one 4-byte global store per register (two 128-byte segments per instruction).
// Instead each wave transposes a [pixels][64 channels] slab through a private LDS staging area and
// writes it as 16-byte stores: 16 consecutive lanes cover one pixel's 64 channels (256 contiguous
// bytes), 4 pixels per instruction — a quarter of the store instructions at the same bytes.
constexpr int WS_STRIDE = 68;                 // floats per staged pixel row (64 + 4: ds_write_b32 of 32 lanes conflict-free)
constexpr int WS_FLOATS = 32 * WS_STRIDE;     // staging floats per wave
// fp32 stores read the staging 16 lanes per pixel (ds_read_b128, four pixels per instruction): with an UNPADDED pixel row
// (64 floats = 16 slots) the 16 lanes of every service group cover 16 different four-bank columns; at 68 floats the second
// pixel of a group is shifted by one slot onto a column of the first (PMC: 21.7 % of the head's LDS cycles were conflicts).
// The 32-lane ds_write_b32 of the accumulators is conflict-free at either stride.  Same-box A/B: the head gains 0.9 % with 64,
// the transposed convolutions lose 0.7-1.7 % (they keep 68): the stride is a template parameter of the fp32 store helpers.
constexpr int WS_UNPADDED = 64;

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Stage NPIX (16 or 32) pixels x 64 channels and store them.  `val(ns, k)` gives the value of channel
// 32*ns + (lane&31) for the k-th staged pixel of this lane (k in [0, NPIX/2)), `pix(k)` its pixel index
// in [0, NPIX); `ptr(px)` returns the global address of channel 0 of the slab for pixel px, or nullptr
// to skip it.
template <int NPIX, int WSF = WS_STRIDE, bool NT = false, typename ValFn, typename PixFn, typename PtrFn>
__device__ __forceinline__ void wide_store(float* stg, int lane, ValFn val, PixFn pix, PtrFn ptr) {
    const int i = lane & 31;
#pragma unroll
    for (int k = 0; k < NPIX / 2; ++k) {
        const int px = pix(k);
        stg[px * WSF + i] = val(0, k);
        stg[px * WSF + 32 + i] = val(1, k);
    }
    wave_lds_fence();
#pragma unroll
    for (int it = 0; it < NPIX / 4; ++it) {
        const int px = it * 4 + (lane >> 4);
        const f32x4 v = *reinterpret_cast<const f32x4*>(stg + px * WSF + (lane & 15) * 4);
        float* g = ptr(px);
        if (g) {
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(g + (lane & 15) * 4));
            else *reinterpret_cast<f32x4*>(g + (lane & 15) * 4) = v;
        }
    }
    wave_lds_fence();
}

// Interior-tile variants: the slab's pixels are `stride` floats (halfs) apart starting at the wave-uniform `base`, all in
// range.  One per-lane offset for the whole tail, the rest is scalar: no per-pixel pointer or bounds arithmetic.
template <int NPIX, int WSF = WS_STRIDE, bool NT = false, typename ValFn, typename PixFn>
__device__ __forceinline__ void wide_store_full(float* stg, int lane, ValFn val, PixFn pix, float* base, int stride) {
    const int i = lane & 31;
#pragma unroll
    for (int k = 0; k < NPIX / 2; ++k) {
        const int px = pix(k);
        stg[px * WSF + i] = val(0, k);
        stg[px * WSF + 32 + i] = val(1, k);
    }
    wave_lds_fence();
    const int lane_off = (lane >> 4) * stride + (lane & 15) * 4;
#pragma unroll
    for (int it = 0; it < NPIX / 4; ++it) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(stg + (it * 4 + (lane >> 4)) * WSF + (lane & 15) * 4);
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(base + (size_t)(it * 4) * stride + lane_off));
        else *reinterpret_cast<f32x4*>(base + (size_t)(it * 4) * stride + lane_off) = v;
    }
    wave_lds_fence();
}

// MODE 0: 3x3 conv + bias + ReLU                      -> out (channel slice of a possibly wider buffer)
// MODE 1: same, plus fused 2x2 max-pool                -> out (cropped region) and pool
// MODE 2: 2x2 stride-2 transposed conv + bias (1 tap)  -> out, pixel-scattered; N index = tap*COUT + co
//
// Pipeline of one workgroup (4 waves, 2 workgroups per CU):
//   prologue : bias -> registers; halo tile of chunk 0 -> LDS; B fragments of steps 0,1 -> registers
//   chunk ck : 36 steps of {A(st+1) <- LDS, B(st+2) <- L2, one 16-byte piece of chunk ck+1's halo
//              tile <- HBM/L2 into a register, 16 MFMAs}.  All loads fly under the MFMAs; vmcnt is
//              in-order, so the one-per-step spread of the halo loads keeps each B wait short.
//   seam     : barrier, 11 ds_write_b128 (prefetched tile -> LDS), barrier.
//   epilogue : bias/ReLU/pool/scatter straight from the accumulators (bias was loaded in the
//              prologue: a load left pending here makes hipcc wait vmcnt(0) before every guarded
//              store, and since CDNA4's vmcnt counts stores too that serialises the whole tail).
//
// ABLATE (timing experiments only, csrc/tools/layer_bench.hip; results are wrong when non-zero):
//   bit 0: no halo prefetch/restage after chunk 0   bit 1: B fragments loaded once
//   bit 2: A fragments read once per chunk          bit 3: no epilogue stores
//   bit 4 (results stay correct, MODE 2): wave 0 writes s_memtime stamps and HW_ID to a.pool (layer_bench <N> trace, tools/trace_stats.py)
template <int CIN, int COUT, int MODE, int ABLATE = 0, int WPS = 2>
__global__ void __launch_bounds__(THREADS, WPS) k_gemm_conv(const GemmConvArgs a) {
#ifndef CID_EXPERIMENTS
    static_assert(ABLATE == 0, "ablation/trace variants are built only by csrc/tools (-DCID_EXPERIMENTS)");
#endif
    constexpr int TAPS = (MODE == 2) ? 1 : 9;
    constexpr int HALO = (MODE == 2) ? 0 : 1;
    constexpr int LW = TILE_W + 2 * HALO;            // LDS tile width in pixels
    constexpr int LH = TILE_H + 2 * HALO;
    constexpr int LPIX = LW * LH;
    constexpr int NSLOT = LPIX * 8;                  // 16-byte slots holding data
    constexpr int NLOAD = (NSLOT + THREADS - 1) / THREADS;
    constexpr int NCHUNK = CIN / KCHUNK;
    constexpr int NOUT = (MODE == 2) ? 4 * COUT : COUT;
    constexpr int NB = NOUT / NTILE;
    constexpr int SPC = TAPS * 4;                    // k-steps (8 channels each) per chunk
    constexpr int NBUF = (SPC % 3 == 0) ? 3 : 2;     // B register ring; prefetch distance NBUF-1 steps
    constexpr int DIST = NBUF - 1;
    static_assert(CIN % KCHUNK == 0 && NOUT % NTILE == 0 && SPC % NBUF == 0, "layer dims");

    __shared__ f32x4 lds[LPIX * PSLOTS];

    int mt, nb;
    if (!decode_block(a.tiles_total, a.tiles_per_xcd, NB, mt, nb)) return;
    int n, ty, tx;
    decode_tile(mt, a.tiles_x, a.tiles_y, a.rcp_x, a.rcp_xy, n, ty, tx);
    const int y0 = ty * TILE_H, x0 = tx * TILE_W;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    // ABLATE bit 4 (trace experiment, MODE 2 only): wave 0 records s_memtime at four points and its HW_ID into a.pool
    unsigned long long* trace = (ABLATE & 16) ? reinterpret_cast<unsigned long long*>(a.pool) + (size_t)blockIdx.x * 8 : nullptr;
    if ((ABLATE & 16) && tid == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        trace[0] = __builtin_readcyclecounter();
        trace[4] = hwid;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        trace[5] = xcc;
    }

    // output-channel block of this workgroup (MODE 2: nb also selects the (kh,kw) tap)
    constexpr int CB = COUT / NTILE;
    const int tap2 = (MODE == 2) ? nb / CB : 0;
    const int cobase = ((MODE == 2) ? (nb - tap2 * CB) : nb) * NTILE;
    float bias_v[2];
#pragma unroll
    for (int ns = 0; ns < 2; ++ns) bias_v[ns] = a.bias[cobase + ns * 32 + i];

    // ---- this thread's pieces of the halo tile: piece `it` is LDS slot s = it*256 + tid ----
    // Raw buffer loads over this image: the piece's byte offset sits in a VGPR for the whole kernel, the chunk is a
    // scalar offset, and pieces outside the image (zero padding, ragged tiles) carry an out-of-range offset for which
    // the buffer range check returns zeros — no address or masking work on the vector ALU per chunk.
    const float* inb = a.in + (size_t)n * a.Hin * a.Win * a.in_ps;
    const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc((void*)inb, (short)0, a.Hin * a.Win * a.in_ps * 4, 0x00020000);
    unsigned goff[NLOAD];
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
        const int s = it * THREADS + tid;
        const int p = s >> 3, c = s & 7;
        const int hy = p / LW, hx = p - hy * LW;
        const int gy = y0 - HALO + hy, gx = x0 - HALO + hx;
        const bool ok = (s < NSLOT) && (unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win;
        goff[it] = ok ? (unsigned)(((gy * a.Win + gx) * a.in_ps + c * 4) * 4) : 0x7ffffff0u;
    }
    const int wslot = lds_slot(tid >> 3, tid & 7);   // slot of piece 0; piece `it` is wslot + it*32*PSLOTS
    auto halo_load = [&](int it, int ck) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, goff[it], ck * (KCHUNK * 4), 0));
    };
    auto halo_store = [&](int it, f32x4 v) {
        if ((it + 1) * THREADS <= NSLOT || it * THREADS + tid < NSLOT) lds[wslot + it * (THREADS / 8) * PSLOTS] = v;
    };

    f32x16 acc[2][2];   // first written by the zero-C MFMAs of chunk 0

    // A fragment base pixel of (row 2*wave+m, column i) inside the LDS tile, tap (0,0)
    const int pbase0 = (2 * wave) * LW + i;
    // B fragments: step g (global over chunks) is the 2 KiB at (nb*NCHUNK*SPC + g)*2048 bytes; scalar offsets only
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, CIN * NOUT * TAPS * 4, 0x00020000);
    const int wbase = nb * NCHUNK * SPC * 2048, wlane = lane * 16;
    auto b_load = [&](int gstep, int ns) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, wlane, wbase + gstep * 2048 + ns * 1024, 0));
    };

    f32x4 pre[NLOAD];
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) pre[it] = halo_load(it, 0);
    f32x4 bq[NBUF][2];
#pragma unroll
    for (int d = 0; d < DIST; ++d) {
        bq[d][0] = b_load(d, 0);
        bq[d][1] = b_load(d, 1);
    }
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) halo_store(it, pre[it]);
    __syncthreads();

    auto chunk = [&](auto first_tag, auto pref_tag, int ck) {
        constexpr bool FIRST = decltype(first_tag)::value;                  // chunk 0: accumulate onto a zero C operand
        constexpr bool PREF = decltype(pref_tag)::value && !(ABLATE & 1);   // another chunk follows
        f32x4 acur[2], anxt[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) acur[m] = lds[lds_slot(pbase0 + m * LW, h)];
#pragma unroll
        for (int st = 0; st < SPC; ++st) {
            if (st + 1 < SPC && !(ABLATE & 4)) {
                const int t2 = (st + 1) >> 2, g2 = (st + 1) & 3;
                const int off = (TAPS == 9) ? ((t2 / 3) * LW + (t2 % 3)) : 0;
#pragma unroll
                for (int m = 0; m < 2; ++m) anxt[m] = lds[lds_slot(pbase0 + m * LW + off, 2 * g2 + h)];
            }
            if ((decltype(pref_tag)::value || st + DIST < SPC) && !(ABLATE & 2)) {
                bq[(st + DIST) % NBUF][0] = b_load(ck * SPC + st + DIST, 0);
                bq[(st + DIST) % NBUF][1] = b_load(ck * SPC + st + DIST, 1);
            }
            if (PREF && st < NLOAD) pre[st] = halo_load(st, ck + 1);
            constexpr int BSEL = (ABLATE & 2) ? 0 : -1;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int ns = 0; ns < 2; ++ns) {
                        if (FIRST && st == 0 && e == 0) {
                            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[m][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[m][e], bq[BSEL < 0 ? st % NBUF : 0][ns][e], zero, 0, 0, 0);
                        } else {
                            acc[m][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[m][e], bq[BSEL < 0 ? st % NBUF : 0][ns][e], acc[m][ns], 0, 0, 0);
                        }
                    }
            if (st + 1 < SPC && !(ABLATE & 4)) {
#pragma unroll
                for (int m = 0; m < 2; ++m) acur[m] = anxt[m];
            }
            __builtin_amdgcn_sched_barrier(0);   // loads of step st stay ahead of step st+1's MFMAs
        }
        if (PREF) {
            if (SPC < NLOAD) {
#pragma unroll
                for (int it = SPC; it < NLOAD; ++it) pre[it] = halo_load(it, ck + 1);
            }
            __syncthreads();   // every wave has finished reading this chunk's tile
#pragma unroll
            for (int it = 0; it < NLOAD; ++it) halo_store(it, pre[it]);
            __syncthreads();
        }
    };
    static_assert(NCHUNK >= 2, "first and last chunk are separate instantiations");
    if ((ABLATE & 16) && tid == 0) trace[1] = __builtin_readcyclecounter();
    chunk(std::true_type{}, std::true_type{}, 0);
    for (int ck = 1; ck + 1 < NCHUNK; ++ck) chunk(std::false_type{}, std::true_type{}, ck);
    chunk(std::false_type{}, std::false_type{}, NCHUNK - 1);
    if ((ABLATE & 16) && tid == 0) trace[2] = __builtin_readcyclecounter();

    if (ABLATE & 8) {   // keep the accumulators alive without the store tail
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ns = 0; ns < 2; ++ns)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[m][ns][r];
        if (sum == 123.456f) a.out[tid] = sum;
        return;
    }

    // ---- epilogue: D[row = pixel column, col = output channel]; lane holds channel j = lane&31 ----
    // and pixel columns xo(r) = (r&3) + 8*(r>>2) + 4*h for its 16 accumulator registers r.  Each wave
    // pushes its two tile rows (and the pooled row) through wide_store; the halo tile's LDS is free now.
    __syncthreads();
    float* stg = reinterpret_cast<float*>(lds) + wave * WS_FLOATS;
    static_assert(4 * WS_FLOATS * sizeof(float) <= sizeof(lds), "staging must fit in the halo tile's LDS");
    auto xo = [&](int r) { return (r & 3) + 8 * (r >> 2) + 4 * h; };
    if (MODE == 2) {
        const int kh = tap2 >> 1, kw = tap2 & 1;
        const int Ho = 2 * a.Hc, Wo = 2 * a.Wc;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int y = y0 + 2 * wave + m;
            float* orow = a.out + ((size_t)(n * Ho + 2 * y + kh) * Wo + kw) * a.out_ps + a.out_coff + cobase;
            const int step = 2 * a.out_ps;
            const bool rowok = y < a.Hc;
            auto val = [&](int ns, int k) { return acc[m][ns][k] + bias_v[ns]; };
            if (y0 + TILE_H <= a.Hc && x0 + TILE_W <= a.Wc)
                wide_store_full<32>(stg, lane, val, xo, orow + (size_t)x0 * step, step);
            else
                wide_store<32>(stg, lane, val, xo,
                               [&](int px) -> float* { return (rowok && x0 + px < a.Wc) ? orow + (size_t)(x0 + px) * step : nullptr; });
        }
        if ((ABLATE & 16) && tid == 0) {
            trace[3] = __builtin_readcyclecounter();                 // all stores issued
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            trace[6] = __builtin_readcyclecounter();                 // ... and written back
        }
    } else {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int y = y0 + 2 * wave + m;
            float* orow = a.out + ((size_t)(n * a.Hs + y) * a.Ws) * a.out_ps + a.out_coff + cobase;
            const bool rowok = y < a.Hs;
            auto val = [&](int ns, int k) { return fmaxf(acc[m][ns][k] + bias_v[ns], 0.f); };
            if (y0 + TILE_H <= a.Hs && x0 + TILE_W <= a.Ws)
                wide_store_full<32>(stg, lane, val, xo, orow + (size_t)x0 * a.out_ps, a.out_ps);
            else
                wide_store<32>(stg, lane, val, xo,
                               [&](int px) -> float* { return (rowok && x0 + px < a.Ws) ? orow + (size_t)(x0 + px) * a.out_ps : nullptr; });
        }
        if (MODE == 1) {
            // 2x2 max-pool, floor mode (nn.MaxPool2d(2,2), app.py:48,56): the four pixels of a
            // window are registers (r, r+1) of the wave's two row tiles — no cross-lane traffic.
            const int Hp = a.Hc >> 1, Wp = a.Wc >> 1;
            const int py = (y0 >> 1) + wave;
            float* prow = a.pool + ((size_t)(n * Hp + py) * Wp) * COUT + cobase;
            const bool rowok = py < Hp;
            wide_store<16>(stg, lane,
                           [&](int ns, int q) {
                               const int r = (q & 1) * 2 + (q >> 1) * 4;
                               const float v = fmaxf(fmaxf(acc[0][ns][r], acc[0][ns][r + 1]), fmaxf(acc[1][ns][r], acc[1][ns][r + 1]));
                               return fmaxf(v + bias_v[ns], 0.f);
                           },
                           [&](int q) { return (q & 1) + 4 * (q >> 1) + 2 * h; },
                           [&](int px) -> float* { return (rowok && (x0 >> 1) + px < Wp) ? prow + (size_t)((x0 >> 1) + px) * COUT : nullptr; });
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Head: down1[0] = Conv2d(3, 64, 3, padding=1) + ReLU  (app.py:43-44).  Reads the caller's NCHW
// input directly (the NCHW -> NHWC change of layout is folded into this kernel), K = 27 padded
// to 28 = 14 MFMA k-steps; memory-bound on its 64-channel NHWC output.
// K order of the head's 14 MFMA steps.  Step s multiplies element k0(s) on lanes h=0 and k1(s) on lanes h=1, k = ci*9 + kh*3
// + kw.  The pairs are chosen so that the two elements of a step lie a FIXED distance apart in the LDS image — one column
// (steps 0-8: kw 0|1 of every (ci,kh)), one row (9-11: kw=2 of kh 0|1), one plane (12: (ci 0|1, kh 2, kw 2)) — so a lane needs
// three base addresses (base + h*distance) and every step's offset is an immediate, instead of fourteen per-lane offset
// registers.  Step 13 holds (2,2,2) alone, on the h=1 lanes one row below the step's address (2,1,2); the h=0 lanes read
// (2,1,2) itself against a ZERO weight.  Both are elements the kernel has written: a zero weight does not make an
// unwritten LDS word harmless (0 x NaN = NaN, and ReLU then turns the NaN into a wrong 0).
struct HeadStep { int addr, k0, k1, dist; };   // addr: element whose offset is the immediate; k0/k1: weights of the h=0/1 lanes (-1 = 0)
__host__ __device__ constexpr HeadStep head_step(int s) {
    return s < 9   ? HeadStep{(s / 3) * 9 + (s % 3) * 3, (s / 3) * 9 + (s % 3) * 3, (s / 3) * 9 + (s % 3) * 3 + 1, 0}
         : s < 12  ? HeadStep{(s - 9) * 9 + 2, (s - 9) * 9 + 2, (s - 9) * 9 + 5, 1}
         : s == 12 ? HeadStep{8, 8, 17, 2}
                   : HeadStep{23, -1, 26, 1};
}
// Host: (step, lane half) that multiplies element k.
inline void head_step_of(int k, int& s, int& h) {
    for (s = 0; s < 14; ++s) {
        if (head_step(s).k0 == k) { h = 0; return; }
        if (head_step(s).k1 == k) { h = 1; return; }
    }
    s = h = -1;
}

// f4: the reference server pads an upload with black to a multiple of 4, runs the network on the padded image and
// crops the padding off the result again (app.py:276-281,384-385,474-480).  Both are index arithmetic in the first and the
// last kernel:
//   head: the network input [H, W] is the caller's image [win.H, win.W] placed at (win.top, win.left); the band around it holds
//         uint8 0, i.e. (0/255 - 0.5)/0.5 = -1.0 after ToTensor + Normalize (transforms.Pad(fill=0) comes first in the reference);
//   tail: the caller's tensor [win.H, win.W] receives the window of the network output that starts at (win.top, win.left).
// Identity: {0, 0, H, W}.
struct Window { int top, left, H, W; };

struct HeadArgs {
    const void* in;     // fp32 NCHW [N,3,src.H,src.W], or (IN_U8) uint8 NHWC [N,src.H,src.W,3]
    Window src;         // where the caller's image sits inside the network input [H, W]
    const float* w;     // packed [2 ns][14 steps][64 lanes] fp32 (k_conv_head) or [4 cg][64 lanes][8] halfs (k_conv_head_h16)
    const float* bias;  // [64]
    void* out;          // NHWC [N,H,W,64]: fp32 (k_conv_head) or half (k_conv_head_h16)
    int N, H, W;
    int tiles_x, tiles_y, tiles_total;
    int tiles_per_wg, groups_total, groups_per_xcd;   // a workgroup walks tiles_per_wg consecutive tiles (host: tile_groups)
    unsigned rcp_x, rcp_xy;   // ceil(2^32 / tiles_x), ceil(2^32 / (tiles_x*tiles_y)): division by multiply-high (host: tile_rcp)
};
// Host: tiles per workgroup of the head and tail kernels — 8 / 4 once there are enough tiles to fill the chip several times
// over, else 1 (small batches: more workgroups beat hidden latency).
template <typename Args>
inline void tile_groups(Args& a) {
    // (r4, same box: 8 tiles per workgroup at B = 256 takes the head launches 2.5 % less time than 4, 1 takes 6-14 % more: gpurun A/B, both storage types)
    a.tiles_per_wg = a.tiles_total >= 16384 ? 8 : a.tiles_total >= 8192 ? 4 : 1;
    a.groups_total = (a.tiles_total + a.tiles_per_wg - 1) / a.tiles_per_wg;
    a.groups_per_xcd = (a.groups_total + 7) / 8;
}

// IN_U8: the caller's image is uint8 HWC (what PIL hands the reference); ToTensor (/255) and Normalize(0.5,0.5)
// (app.py:401-405) are applied on the fly, in fp32, with true divisions like torchvision: (u8/255 - 0.5)/0.5.
// ABLATE (timing experiments only, tools/headtail_bench.hip; wrong results when non-zero): 1 no input loads, 2 no MFMAs, 4 no stores.
//
// A workgroup walks `a.tiles_per_wg` consecutive tiles.  The input elements of the NEXT tile are requested (into
// registers) before the current tile's MFMAs and stores, so after the first tile no input latency is exposed
// (tools/headtail_bench: removing the input loads altogether was worth 0.05 of this kernel's 0.23 ms).
template <bool IN_U8, int ABLATE = 0>
__global__ void __launch_bounds__(THREADS, 4) k_conv_head(const HeadArgs a) {
#ifndef CID_EXPERIMENTS
    static_assert(ABLATE == 0, "ablation/trace variants are built only by csrc/tools (-DCID_EXPERIMENTS)");
#endif
    constexpr int LW = 36, LH = TILE_H + 2, PLANE = LW * LH;   // 34 used columns, padded to 36
    constexpr int STG16 = 16 * WS_STRIDE;                      // store staging per wave: 16 pixels x 64 channels
    __shared__ __attribute__((aligned(16))) float lds[3 * PLANE + 4 * STG16];   // input planes | store staging (21.7 KB)
    static_assert((3 * PLANE) % 4 == 0, "staging must stay 16-byte aligned");
    int grp, nb;
    if (!decode_block(a.groups_total, a.groups_per_xcd, 1, grp, nb)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, h = lane >> 5;

    // Input planes -> LDS.  All of a thread's loads are issued before the first is used (a load -> wait -> LDS write
    // loop serialises four memory latencies per workgroup): raw buffer loads over the tile's image, out-of-image
    // elements carry an out-of-range offset and read as zero.  Element s = it*256 + tid of the [3][LH][34] halo patch:
    // its plane/row/column and its LDS index do not depend on the tile.
    constexpr int NS = 3 * LH * 34, NIT = (NS + THREADS - 1) / THREADS;
    const size_t img = (size_t)a.src.H * a.src.W * 3;   // elements per image in either input format
    int pc[NIT], phy[NIT], phx[NIT], lidx[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int s = it * THREADS + tid;
        pc[it] = s / (LH * 34);
        const int rem = s - pc[it] * (LH * 34);
        phy[it] = rem / 34; phx[it] = rem - phy[it] * 34;
        lidx[it] = s < NS ? pc[it] * PLANE + phy[it] * LW + phx[it] : -1;
    }
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
            const int gy = ty0 - 1 + phy[it], gx = tx0 - 1 + phx[it];          // network-input coordinates
            const bool net = lidx[it] >= 0 && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            const int sy = gy - a.src.top, sx = gx - a.src.left;                 // the caller's image
            const bool ok = net && (unsigned)sy < (unsigned)a.src.H && (unsigned)sx < (unsigned)a.src.W;
            const unsigned goff = !ok ? 0x7ffffff0u : IN_U8 ? (unsigned)((sy * a.src.W + sx) * 3 + pc[it]) : (unsigned)(((pc[it] * a.src.H + sy) * a.src.W + sx) * 4);
            if (ABLATE & 1) { staged[it] = (float)tid; continue; }
            // three kinds of element: the image; the black band the server pads with (-1.0 once normalised); the convolution's
            // zero padding outside the network input, which applies to the NORMALISED tensor: 0, not (0/255 - 0.5)/0.5
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
    float bw[2][14], bias_v[2];
#pragma unroll
    for (int ns = 0; ns < 2; ++ns) {
        bias_v[ns] = a.bias[ns * 32 + i];   // loaded here, not in the store tail (see k_gemm_conv)
#pragma unroll
        for (int s = 0; s < 14; ++s) bw[ns][s] = a.w[(ns * 14 + s) * 64 + lane];
    }
    // One output row (32 pixels x 64 channels) at a time: 32 accumulator registers instead of 64 and a 16-pixel store
    // slab per wave in its own LDS region, so that four workgroups fit a CU and one's stores, another's MFMAs and a
    // third's input latency overlap (tools/headtail_bench: 2 -> 3 -> 4 workgroups per CU = 0.305 -> 0.267 -> 0.235 ms).
    float* stg = lds + 3 * PLANE + wave * STG16;
    const int pbase = (2 * wave) * LW + i;
#pragma unroll 1
    for (int t = 0; t < ntile; ++t) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (lidx[it] >= 0) lds[lidx[it]] = staged[it];
        __syncthreads();
        int nn = n, ny0 = y0, nx0 = x0;
        if (t + 1 < ntile) request_tile(tile0 + t + 1, nn, ny0, nx0);   // in flight under this tile's MFMAs and stores
        const bool full = y0 + TILE_H <= a.H && x0 + TILE_W <= a.W;
#pragma unroll 1
        for (int m = 0; m < 2; ++m) {   // not unrolled: hipcc would interleave the two rows and double the live accumulators
            f32x16 acc[2];   // first written by the zero-C MFMAs of step 0
            const int base_col = pbase + m * LW + h, base_row = pbase + m * LW + h * LW, base_plane = pbase + m * LW + h * PLANE;
#pragma unroll
            for (int s = 0; s < 14; ++s) {
                const HeadStep hs = head_step(s);   // folds to constants after unrolling
                const int o0 = (hs.addr / 9) * PLANE + ((hs.addr % 9) / 3) * LW + (hs.addr % 3);   // immediate
                const float av = lds[(hs.dist == 0 ? base_col : hs.dist == 1 ? base_row : base_plane) + o0];
#pragma unroll
                for (int ns = 0; ns < 2; ++ns) {
                    if ((ABLATE & 2) && s > 0) { acc[ns][s] += av * bw[ns][s]; continue; }
                    if (s == 0) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bw[ns][s], zero, 0, 0, 0);
                    } else {
                        acc[ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bw[ns][s], acc[ns], 0, 0, 0);
                    }
                }
            }
            if (ABLATE & 4) {
                float sum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[0][r] + acc[1][r];
                if (sum == 123.456f) static_cast<float*>(a.out)[tid] = sum;
                continue;
            }
            const int y = y0 + 2 * wave + m;
            const bool rowok = y < a.H;
#pragma unroll
            for (int q = 0; q < 2; ++q) {   // accumulator registers 8q..8q+7 are pixels 16q..16q+15 of the row
                auto val = [&](int ns, int k) { return fmaxf(acc[ns][8 * q + k] + bias_v[ns], 0.f); };
                auto pix = [&](int k) { return (k & 3) + 8 * (k >> 2) + 4 * h; };
                const int xq = x0 + 16 * q;
                float* orow = static_cast<float*>(a.out) + ((size_t)(n * a.H + y) * a.W) * 64;
                if (full)
                    wide_store_full<16, WS_UNPADDED, true>(stg, lane, val, pix, orow + (size_t)xq * 64, 64);
                else
                    wide_store<16, WS_UNPADDED, true>(stg, lane, val, pix,
                                                [&](int px) -> float* { return (rowok && xq + px < a.W) ? orow + (size_t)(xq + px) * 64 : nullptr; });
            }
        }
        n = nn; y0 = ny0; x0 = nx0;
        if (t + 1 < ntile) __syncthreads();   // every wave is done reading the planes before the next tile overwrites them
    }
}

// ---------------------------------------------------------------------------------------------
// Tail: upconv1[2] = Conv2d(64, 3, 3, padding=1) followed by torch.tanh  (app.py:77,101,103).
// Cout = 3 is no GEMM, but (tap, cout) = 27 columns is: the kernel first computes, for every pixel p
// of the 10x34 halo tile, z[p][3*tap + co] = sum_ci x[p][ci] * W[co][ci][tap] as a [352 x 64] x [64 x 32]
// MFMA product (27 of 32 columns used; weights held in 32 registers), parks z in the LDS the input
// tile occupied, and then every output pixel adds its nine shifted z entries, the bias, applies
// tanh and is written straight into the caller's NCHW tensor (the NHWC -> NCHW change of layout is
// folded in).  HBM-bound on the 64-channel NHWC input it reads once.
struct TailArgs {
    const void* in;     // NHWC [N,H,W,64]: fp32, or (IN_F16) half from the fp16-storage path
    const float* w;     // packed [2 chunk][4 group][64 lanes][4]  (cid_api.hip packed_index, TAIL)
    const float* bias;  // [3]
    void* out;          // fp32 NCHW [N,3,crop.H,crop.W], or (OUT_U8) uint8 NHWC [N,crop.H,crop.W,3]
    Window crop;        // the window of the network output [H, W] the caller's tensor receives
    int N, H, W;
    int tiles_x, tiles_y, tiles_total;
    int tiles_per_wg, groups_total, groups_per_xcd;   // a workgroup walks tiles_per_wg consecutive tiles (host: tail_groups)
    unsigned rcp_x, rcp_xy;   // ceil(2^32 / tiles_x), ceil(2^32 / (tiles_x*tiles_y)): division by multiply-high (host: tile_rcp)
};
// OUT_U8: the reference's view transform and PIL conversion folded in: y*0.5+0.5, clamp to [0,1] (app.py:435),
// then ToPILImage's mul(255).byte() — truncation, not rounding (app.py:471-472; denoisegan_eval.py:97-98).
// ABLATE (timing experiments only): 1 no input loads, 2 no MFMAs, 8 no gather/tanh/store epilogue.
//
// A workgroup walks `a.tiles_per_wg` consecutive tiles.  The loads of a tile's second 32-channel chunk fly under the
// first chunk's MFMAs, and the NEXT tile's first chunk under the second chunk's MFMAs and the epilogue, in the same
// registers: after the first tile no memory latency is exposed (tools/headtail_bench: the load phase alone is
// 0.17 ms of the 0.30 ms this kernel took when it ran load -> product -> load -> product -> epilogue).
template <bool OUT_U8, bool IN_F16 = false, int ABLATE = 0>
__global__ void __launch_bounds__(THREADS, 2) k_conv_tail(const TailArgs a) {
#ifndef CID_EXPERIMENTS
    static_assert(ABLATE == 0, "ablation/trace variants are built only by csrc/tools (-DCID_EXPERIMENTS)");
#endif
    constexpr int LW = TILE_W + 2, LH = TILE_H + 2, LPIX = LW * LH;   // 340 halo pixels
    constexpr int MT = (LPIX + 31) / 32, LP = MT * 32;                // 11 M tiles, 352 rows
    constexpr int NSLOT = LPIX * 8, NLOAD = (NSLOT + THREADS - 1) / THREADS;
    constexpr int ZS = 33;                                            // z row stride in floats (conflict-free)
    __shared__ f32x4 lds[LP * PSLOTS];
    static_assert(LP * ZS * sizeof(float) <= sizeof(lds), "z must fit where x was");
    int grp, nb;
    if (!decode_block(a.groups_total, a.groups_per_xcd, 1, grp, nb)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, h = lane >> 5;

    f32x4 wb[2][4];
#pragma unroll
    for (int ck = 0; ck < 2; ++ck)
#pragma unroll
        for (int g = 0; g < 4; ++g) wb[ck][g] = reinterpret_cast<const f32x4*>(a.w)[(ck * 4 + g) * 64 + lane];
    float bias_v[3];
#pragma unroll
    for (int co = 0; co < 3; ++co) bias_v[co] = a.bias[co];

    // halo pieces by raw buffer loads over the tile's image: per-piece byte offsets computed once per tile, the 32-channel
    // chunk is a scalar offset, out-of-image pieces carry an out-of-range offset (the range check returns the zero padding)
    const size_t img_elems = (size_t)a.H * a.W * 64;
    int n, y0, x0;
    __amdgpu_buffer_rsrc_t rsrc_in;
    unsigned goff[NLOAD];
    f32x4 stage[NLOAD];
    auto setup_tile = [&](int tile, int& tn, int& ty0, int& tx0) {
        int ty, tx;
        decode_tile(tile, a.tiles_x, a.tiles_y, a.rcp_x, a.rcp_xy, tn, ty, tx);
        ty0 = ty * TILE_H; tx0 = tx * TILE_W;
        rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
            IN_F16 ? (void*)(static_cast<const _Float16*>(a.in) + (size_t)tn * img_elems) : (void*)(static_cast<const float*>(a.in) + (size_t)tn * img_elems),
            (short)0, (int)(img_elems * (IN_F16 ? 2 : 4)), 0x00020000);
#pragma unroll
        for (int it = 0; it < NLOAD; ++it) {
            const int s = it * THREADS + tid;
            const int p = s >> 3, c = s & 7;
            const int hy = p / LW, hx = p - hy * LW;
            const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
            const bool ok = (s < NSLOT) && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            goff[it] = ok ? (unsigned)(((gy * a.W + gx) * 64 + c * 4) * (IN_F16 ? 2 : 4)) : 0x7ffffff0u;
        }
    };
    auto load_chunk = [&](int ck) {
#pragma unroll
        for (int it = 0; it < NLOAD; ++it) {
            if (ABLATE & 1) { stage[it] = f32x4{(float)tid, 1.f, 2.f, 3.f}; continue; }
            if (IN_F16) {
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(rsrc_in, goff[it], ck * (KCHUNK * 2), 0);
                const f16x4 hv = __builtin_bit_cast(f16x4, raw);
#pragma unroll
                for (int e = 0; e < 4; ++e) stage[it][e] = (float)hv[e];
            } else {
                stage[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, goff[it], ck * (KCHUNK * 4), 0));
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int it = 0; it < NLOAD; ++it) {
            const int s = it * THREADS + tid;
            if (s < NSLOT) lds[lds_slot(s >> 3, s & 7)] = stage[it];
        }
    };
    f32x16 acc[3];
    auto product = [&](auto first_tag, int ck) {
        constexpr bool FIRST = decltype(first_tag)::value;   // chunk 0 starts the accumulators from a zero C operand
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int mtile = wave + 4 * t;          // wave-uniform
            if (mtile < MT) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // rows 340..351 of the last tile read never-written LDS: they only reach z rows nobody gathers
                    const f32x4 av = lds[lds_slot(mtile * 32 + i, 2 * g + h)];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (ABLATE & 2) { acc[t][4 * g + e] = av[e] * wb[ck][g][e]; continue; }
                        if (FIRST && g == 0 && e == 0) {
                            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], wb[ck][g][e], zero, 0, 0, 0);
                        } else {
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], wb[ck][g][e], acc[t], 0, 0, 0);
                        }
                    }
                }
            }
        }
    };

    const int tile0 = grp * a.tiles_per_wg;
    const int ntile = min(a.tiles_per_wg, a.tiles_total - tile0);   // >= 1 (decode_block), workgroup-uniform
    setup_tile(tile0, n, y0, x0);
    load_chunk(0);
    for (int t = 0; t < ntile; ++t) {
        store_chunk();                       // chunk 0 of this tile (requested during the previous tile)
        load_chunk(1);                       // in flight under chunk 0's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        product(std::true_type{}, 0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        store_chunk();
        int nn = n, ny0 = y0, nx0 = x0;
        if (t + 1 < ntile) {                 // next tile's chunk 0: in flight under chunk 1's MFMAs and the epilogue
            setup_tile(tile0 + t + 1, nn, ny0, nx0);
            load_chunk(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        product(std::false_type{}, 1);
        __syncthreads();   // every wave is done reading x: the LDS becomes z[352][33]
        if (!(ABLATE & 8)) {
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
            const int cy = y - a.crop.top, cx = x - a.crop.left;       // the caller's tensor
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
        } else {
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[0][r] + acc[1][r] + acc[2][r];
            if (sum == 123.456f) static_cast<float*>(a.out)[tid] = sum;
        }
        n = nn; y0 = ny0; x0 = nx0;
    }
}


// ---------------------------------------------------------------------------------------------
// Tail, second decomposition (the default for images up to 128 pixels wide): a workgroup owns a BAND of rows of one
// image and slides down it, one row per step.
//
//   z[p][3*tap + co] = sum_ci x[p][ci] * W[co][ci][tap]        once per PIXEL p  (k_conv_tail: once per pixel of every
//                                                               tile's halo, 340 per 256 outputs)
//   out[y][x][co]    = bias[co] + sum_{ty,tx} z[(y+ty-1, x+tx-1)][3*(3*ty+tx) + co],   then tanh
//
//   * a z row (<= 128 pixels) = 4 MFMA M-tiles, one per wave: [32 pixels x 64 ci] x [64 x 32 (27 used)], 32 MFMAs;
//   * every wave is self-sufficient for x: it requests, TWO ROWS ahead and straight into registers, exactly the pixels of
//     its own M-tile (half-row chunks of 32 pixels x 32 channels = 4 KiB, 16 VGPRs per chunk, four chunks in flight),
//     transposes them through a wave-private 4 KiB LDS buffer (16-byte ds_write, ds_read_b128 A fragments) and runs its
//     MFMAs — no workgroup barrier stands between a load and its use, so one wave's memory latency never stalls
//     another.  The data in flight lives in the register file (512 KiB per CU), not in LDS: three workgroups fit a CU
//     (44 KiB of LDS each) with 192 KiB of loads in flight per CU.  hipcc counts the waits itself (plain loads/stores);
//   * the 16-byte quads of a pixel are stored XOR-swizzled — physical quad = logical quad ^ ((pixel >> 1) & 7) — which
//     makes every ds_read_b128 of an A fragment conflict-free without padding: within each of the instruction's four
//     16-lane service groups the 16 pixels then cover all 16 four-bank columns;
//   * a z row goes to LDS once ([130 entries][27], entry e = column e-1; entries 0 and 129 are the zero padding; two
//     slots, alternating) and is gathered once: it contributes to output rows zr-1 (ty=2), zr (ty=1) and zr+1 (ty=0),
//     whose partial sums live in registers — lane (i, h) of wave w owns pixel 32w+i; h=0 sums the taps (ty0: tx0,1,2),
//     (ty1: tx0,1), h=1 the taps (ty1: tx2), (ty2: tx0,1,2); the halves meet through one v_permlane32_swap per channel
//     when a row completes;
//   * ONE barrier per row (between a z row's write and its gather: neighbouring pixels belong to other waves);
//   * straight-line row loop: rows outside the image are read through an empty buffer descriptor (zeros, no traffic) and
//     rows outside the band are "stored" at an out-of-range offset, so no branch splits hipcc's wait counting.
struct Tail2Args {
    const float* in;    // NHWC [N,H,W,64] fp32
    const float* w;     // packed as for k_conv_tail: [2 chunk][4 group][64 lanes][4]
    const float* bias;  // [3]
    void* out;          // fp32 NCHW [N,3,crop.H,crop.W], or (OUT_U8) uint8 NHWC [N,crop.H,crop.W,3]
    Window crop;        // the window of the network output [H, W] the caller's tensor receives
    int N, H, W;
    int band_rows, bands_per_image, groups_total;   // host: tail2_plan
    unsigned rcp_bands;                              // ceil(2^32 / bands_per_image)
};
constexpr int T2_MAXW = 128;
// Host: rows per band — as tall as possible (every band re-reads two halo rows) while the launch still has three
// workgroups for every CU.
inline void tail2_plan(Tail2Args& a, int rows = 0) {
    int r = 64;
    while (r > 8 && (long long)a.N * ((a.H + r - 1) / r) < 768) r >>= 1;
    if (rows > 0) r = rows;
    a.band_rows = r;
    a.bands_per_image = (a.H + r - 1) / r;
    a.groups_total = a.N * a.bands_per_image;
    a.rcp_bands = tile_rcp((unsigned)a.bands_per_image);
}

// ABLATE (timing experiments only): 1 no input loads, 2 no MFMAs, 4 no gather/tanh/stores.
template <bool OUT_U8, int ABLATE = 0>
__global__ void __launch_bounds__(THREADS, 3) k_conv_tail2(const Tail2Args a) {
#ifndef CID_EXPERIMENTS
    static_assert(ABLATE == 0, "ablation/trace variants are built only by csrc/tools (-DCID_EXPERIMENTS)");
#endif
    constexpr int WAVE_SLOTS = 32 * 8;                     // wave-private x buffer: 32 pixels x 8 quads of 16 B = 4 KiB
    constexpr int NPW = 4;                                 // pieces (16 B per lane, 8 pixels) per wave and chunk
    constexpr int ZE = T2_MAXW + 2, ZC = 27;               // z row: 130 entries x 27 columns
    constexpr int ZROW = ZE * ZC + 2;                      // floats per z slot (16-byte multiple)
    static_assert((ZROW % 4) == 0, "layout");
    __shared__ f32x4 lds[4 * WAVE_SLOTS + (2 * ZROW + 4) / 4];   // 16,384 + 28,096 + 16 B
    float* const zl = reinterpret_cast<float*>(lds + 4 * WAVE_SLOTS);
    constexpr int ZERO_AT = 2 * ZROW;                      // floats ZERO_AT .. +2 stay zero (the padded taps read them)

    const int grp = blockIdx.x;
    if (grp >= a.groups_total) return;
    const unsigned ug = (unsigned)__builtin_amdgcn_readfirstlane(grp);
    const int n = (int)(a.rcp_bands ? __umulhi(ug, a.rcp_bands) : ug);
    const int band = grp - n * a.bands_per_image;
    const int r0 = band * a.band_rows, r1 = min(r0 + a.band_rows, a.H);   // output rows [r0, r1)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;

    // ---- weights (B operand) and bias: once per workgroup ----
    f32x4 wb[2][4];
#pragma unroll
    for (int ck = 0; ck < 2; ++ck)
#pragma unroll
        for (int g = 0; g < 4; ++g) wb[ck][g] = reinterpret_cast<const f32x4*>(a.w)[(ck * 4 + g) * 64 + lane];
    float bias_v[3];
#pragma unroll
    for (int co = 0; co < 3; ++co) bias_v[co] = a.bias[co];

    // ---- this wave's input pieces: piece m covers pixels 32*wave + 8*m .. +8; lane -> (pixel, logical quad lane & 7) ----
    // One buffer descriptor per ROW (num_records = the row's bytes, or 0 for a row outside the image): pixels past the row
    // end and whole zero rows are out of range and read as zeros; piece m is piece 0's offset + m * 2 KiB, and its LDS slot
    // is piece 0's + 64 m (8 pixels further on).
    const float* inb = a.in + (size_t)n * a.H * a.W * 64;
    const int pl0 = lane >> 3, q0 = lane & 7;              // pixel within the wave's 32, quad
    const unsigned voff0 = (unsigned)((32 * wave + pl0) * 256 + q0 * 16);
    f32x4* const xw = lds + wave * WAVE_SLOTS;             // this wave's buffer
    int wslot[NPW];
#pragma unroll
    for (int m = 0; m < NPW; ++m) { const int pl = 8 * m + pl0; wslot[m] = pl * 8 + (q0 ^ ((pl >> 1) & 7)); }
    const int zr_first = r0 - 1, zr_last = r1;             // z rows needed: [r0-1, r1]
    f32x4 stage[4][NPW];                                   // chunk (row zr_first + k, half) lives in stage[2*(k&1) + half]
    auto request = [&](auto slot_tag, int zr, int half) {
        constexpr int SLOT = decltype(slot_tag)::value;
        if (ABLATE & 1) return;
        const bool ok = zr >= 0 && zr < a.H && zr <= zr_last;
        const int zc = ok ? zr : 0;
        const __amdgpu_buffer_rsrc_t rsrc_row = __builtin_amdgcn_make_buffer_rsrc((void*)(inb + (size_t)zc * a.W * 64), (short)0, ok ? a.W * 256 : 0, 0x00020000);
#pragma unroll
        for (int m = 0; m < NPW; ++m)
            stage[SLOT][m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_row, voff0 + m * 2048, half * 128, 0));
    };
    auto to_lds = [&](auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        if (ABLATE & 1) return;
#pragma unroll
        for (int m = 0; m < NPW; ++m) xw[wslot[m]] = stage[SLOT][m];
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;
    request(S0{}, zr_first, 0);
    request(S1{}, zr_first, 1);
    request(S2{}, zr_first + 1, 0);
    request(S3{}, zr_first + 1, 1);
    // The weights were requested BEFORE the four chunks; using them here makes hipcc wait for them now (they are the oldest
    // loads in flight: the chunks stay in flight).  Without this their wait sits in front of the loop's first MFMAs, where
    // its count (set by the first iteration) would force all but seven of the loads in flight to land in EVERY iteration.
#pragma unroll
    for (int ck = 0; ck < 2; ++ck)
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" ::"v"(wb[ck][g]));

    // ---- output addressing: two buffer stores per row; lanes (and rows) with nothing to store carry an out-of-range offset ----
    const int px = 32 * wave + i;                          // this lane's pixel (column of the network output)
    const int cx = px - a.crop.left;                       // ... and of the caller's tensor
    const bool pxok = px < a.W && (unsigned)cx < (unsigned)a.crop.W;
    const size_t plane = (size_t)a.crop.H * a.crop.W;
    const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(
        OUT_U8 ? (void*)(static_cast<unsigned char*>(a.out) + (size_t)n * plane * 3) : (void*)(static_cast<float*>(a.out) + (size_t)n * plane * 3),
        (short)0, (int)(OUT_U8 ? plane * 3 : plane * 12), 0x00020000);
    // store 1: channel h of pixel px; store 2: channel 2 (h = 0 lanes only)
    const unsigned ooff1 = pxok ? (OUT_U8 ? (unsigned)(cx * 3 + h) : (unsigned)((h * plane + cx) * 4)) : 0x7ffffff0u;
    const unsigned ooff2 = (pxok && h == 0) ? (OUT_U8 ? (unsigned)(cx * 3 + 2) : (unsigned)((2 * plane + cx) * 4)) : 0x7ffffff0u;

    // ---- A fragments: logical quad 2g+h of pixel i of the wave's buffer, swizzled: slot = 8 i + ((2g + h) ^ swz) = (8 i + (h ^ swz)) ^ 2g ----
    const int aslot0 = i * 8 + (h ^ ((i >> 1) & 7));

    // ---- gather addresses (floats into a z slot): group U completes a row, group V is carried to the next step ----
    //   h=0: U = ty1 (tx0, tx1, zero)   V = ty0 (tx0, tx1, tx2)        h=1: U = ty2 (tx0, tx1, tx2)   V = ty1 (tx2, zero, zero)
    // The padded taps point at the zero triple: their address must not move with the slot, hence two bases per group.
    auto tap_at = [&](int ty, int tx) { return (px + tx) * ZC + (3 * ty + tx) * 3; };
    const int ua0 = h ? tap_at(2, 0) : tap_at(1, 0), ua1 = h ? tap_at(2, 1) : tap_at(1, 1);
    const int va0 = h ? tap_at(1, 2) : tap_at(0, 0);
    const int ua2[2] = {h ? tap_at(2, 2) : ZERO_AT, h ? tap_at(2, 2) + ZROW : ZERO_AT};
    const int va1[2] = {h ? ZERO_AT : tap_at(0, 1), h ? ZERO_AT : tap_at(0, 1) + ZROW};
    const int va2[2] = {h ? ZERO_AT : tap_at(0, 2), h ? ZERO_AT : tap_at(0, 2) + ZROW};

    // zero padding of both z slots (entries 0 and 129) and the zero triple: written once, never overwritten
    if (tid < ZC) { zl[tid] = 0.f; zl[(ZE - 1) * ZC + tid] = 0.f; zl[ZROW + tid] = 0.f; zl[ZROW + (ZE - 1) * ZC + tid] = 0.f; }
    if (tid < 4) zl[ZERO_AT + tid] = 0.f;

    f32x16 acc;
    float carry[3] = {0.f, 0.f, 0.f}, tprev[3] = {0.f, 0.f, 0.f};
    auto mfma_chunk = [&](auto first_tag, auto half_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int HALF = decltype(half_tag)::value;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int sl = aslot0 ^ (2 * g);
            if (g) asm volatile("" : "+v"(sl));            // keep the xor at its use: hoisted out of the row loop it costs three VGPRs (and spilled)
            const f32x4 av = xw[sl];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (ABLATE & 2) { acc[4 * g + e] = av[e] * wb[HALF][g][e]; continue; }
                if (FIRST && g == 0 && e == 0) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], wb[HALF][g][e], zero, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], wb[HALF][g][e], acc, 0, 0, 0);
                }
            }
        }
    };
    // gather z row `zr` (slot Z) into the partial sums; emit output row zr-1 (dropped by the range check outside the band)
    auto gather_emit = [&](auto z_tag, int zr) {
        constexpr int Z = decltype(z_tag)::value;
        if (ABLATE & 4) return;
        const float* zs = zl + Z * ZROW;
        float tot[3];
#pragma unroll
        for (int co = 0; co < 3; ++co) {
            const float u = (zs[ua0 + co] + zs[ua1 + co]) + zl[ua2[Z] + co];
            const float v = (zs[va0 + co] + zl[va1[Z] + co]) + zl[va2[Z] + co];
            const float t = carry[co] + u;                 // h=0: a(zr) = ty0(zr-1) + ty1'(zr);  h=1: d(zr-1) = ty1''(zr-1) + ty2(zr)
            carry[co] = v;
            const float wv = h ? t : tprev[co];            // h=0: a(zr-1), h=1: d(zr-1): the two parts of output row zr-1
            tprev[co] = t;
            // v_permlane32_swap x, y exchanges x[32..63] with y[0..31]: with y a copy of x the results are {lo, lo} and
            // {hi, hi}, whose sum is a + d in both halves.  Inline asm, not __builtin_amdgcn_permlane32_swap: hipcc (ROCm 7.2)
            // turned the sum of the builtin's two results into 2 x the first one.  s_nop: VALU write -> permlane read hazard.
            float wx = wv, wy = wv;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(wx), "+v"(wy));
            tot[co] = (wx + wy) + bias_v[co];
        }
        const int y = zr - 1, cy = y - a.crop.top;         // network-output row, row of the caller's tensor
        const bool emit = y >= r0 && y < r1 && (unsigned)cy < (unsigned)a.crop.H;   // workgroup-uniform
        const float v1 = tanhf(h ? tot[1] : tot[0]), v2 = tanhf(tot[2]);
        if (OUT_U8) {
            const int so = emit ? cy * a.crop.W * 3 : 0;
            const float q1 = fminf(fmaxf(v1 * 0.5f + 0.5f, 0.f), 1.f), q2 = fminf(fmaxf(v2 * 0.5f + 0.5f, 0.f), 1.f);
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(q1 * 255.0f), rsrc_out, emit ? ooff1 : 0x7ffffff0u, so, 0);
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(q2 * 255.0f), rsrc_out, emit ? ooff2 : 0x7ffffff0u, so, 0);
        } else {
            const int so = emit ? cy * a.crop.W * 4 : 0;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), rsrc_out, emit ? ooff1 : 0x7ffffff0u, so, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v2), rsrc_out, emit ? ooff2 : 0x7ffffff0u, so, 0);
        }
    };

    // One row step.  P = parity of the step: its chunks sit in stage[2P], stage[2P+1] and its z row goes to slot P; a stage
    // slot that has gone to LDS is re-requested at once for the row two steps down.  Within a wave LDS operations execute in
    // order, so the wave-private buffer needs no barrier: write half 0, read its fragments, MFMAs, write half 1, read, MFMAs.
    auto step = [&](auto parity_tag, int zr) {
        constexpr int P = decltype(parity_tag)::value;
        using SA = std::integral_constant<int, 2 * P>; using SB = std::integral_constant<int, 2 * P + 1>;
        to_lds(SA{});
        request(SA{}, zr + 2, 0);
        mfma_chunk(std::true_type{}, std::integral_constant<int, 0>{});
        to_lds(SB{});
        request(SB{}, zr + 2, 1);
        mfma_chunk(std::false_type{}, std::integral_constant<int, 1>{});
        if (i < ZC) {                                      // lane holds column i = 3*tap + co of pixels (r&3) + 8*(r>>2) + 4*h
#pragma unroll
            for (int r = 0; r < 16; ++r) zl[P * ZROW + (32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h + 1) * ZC + i] = acc[r];
        }
        __syncthreads();                                   // z row zr is complete; every wave is past its gather of row zr-1
        gather_emit(parity_tag, zr);
    };
    for (int zr = zr_first; zr <= zr_last; zr += 2) {      // an odd count of rows runs one extra step on an empty row (nothing stored)
        step(std::integral_constant<int, 0>{}, zr);
        step(std::integral_constant<int, 1>{}, zr + 1);
    }
}

// ---------------------------------------------------------------------------------------------
// Tail, third form (the default with the Winograd kernels): the producer, k_wino64_conv<128, 64, .., ZOUT>, has already
// contracted the 64 channels, z[n][3*tap + co][y][x] (27 planes).  What is left of upconv1[2] + tanh (app.py:77,103) is
//     out[n][co][y][x] = tanh(bias[co] + sum_{ty,tx} z[n][3*(3*ty+tx) + co][y+ty-1][x+tx-1])          (zero outside the image)
// Every z element is read by exactly one output, so there is nothing to stage: one thread per pixel, 27 coalesced 4-byte
// loads (plane = scalar offset, the nine shifted pixel offsets in registers, out-of-image taps carry an out-of-range
// offset and read as zero), three tanh, three stores.  HBM-bound on 120 B per pixel (108 z + 12 out).
struct TailZArgs {
    const float* z;     // [N, 27, H, W]
    const float* bias;  // [3]
    void* out;          // fp32 NCHW [N,3,crop.H,crop.W], or (OUT_U8) uint8 NHWC [N,crop.H,crop.W,3]
    Window crop;        // the window of the network output [H, W] the caller's tensor receives
    int N, H, W;
    int blocks_per_image;   // ceil(H*W / 256)
    unsigned rcp_w, rcp_blocks;
};
template <bool OUT_U8>
__global__ void __launch_bounds__(THREADS) k_conv_tail_z(const TailZArgs a) {
    const unsigned b = blockIdx.x;
    const unsigned n = a.rcp_blocks ? __umulhi(b, a.rcp_blocks) : b;
    const unsigned p = (b - n * a.blocks_per_image) * THREADS + threadIdx.x;   // pixel index inside the image
    const size_t plane = (size_t)a.H * a.W;
    const unsigned y = a.rcp_w ? __umulhi(p, a.rcp_w) : p, x = p - y * a.W;
    const bool inside = p < plane;
    // descriptor over the image's 27 planes (<= 453 MB: H*W < 4,194,303, cid_api.hip shape_error); the scalar plane offset
    // takes part in the hardware's range check on this part, so the descriptor cannot be a single plane
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)(a.z + (size_t)n * 27 * plane), (short)0, (int)(plane * 108), 0x00020000);
    float o[3] = {a.bias[0], a.bias[1], a.bias[2]};
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
            const int yy = (int)y + ty - 1, xx = (int)x + tx - 1;
            const bool ok = inside && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
            const unsigned off = ok ? (unsigned)((yy * a.W + xx) * 4) : 0x7ffffff0u;
#pragma unroll
            for (int co = 0; co < 3; ++co)   // plane (3*tap + co) is a scalar offset; out-of-image taps: out-of-range vector offset
                o[co] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, off, (int)(((3 * ty + tx) * 3 + co) * plane * 4), 0));
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

// ---------------------------------------------------------------------------------------------
// The reference's view transform as a stand-alone pass (app.py:435 `y*0.5+0.5` clamp(0,1), :471-472 ToPILImage = mul(255).byte(),
// truncating; denoise_eavl_iter.py:97-110 saves such a view of EVERY fed-back iteration): fp32 NCHW tanh-range tensor -> uint8 NHWC
// image, the same arithmetic as k_conv_tail_z<true>'s store.  For callers that need both the fp32 tensor (fed back) and its image.
// One thread per pixel: three coalesced plane reads, three bytes out.  HBM-bound, 15 B per pixel.
__global__ void __launch_bounds__(THREADS) k_view_u8(const float* __restrict__ in, unsigned char* __restrict__ out, size_t plane, size_t pixels) {
    const size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;    // pixel index over the batch
    if (i >= pixels) return;
    const size_t n = i / plane, p = i - n * plane;
    const float* ip = in + n * 3 * plane + p;
    unsigned char* op = out + i * 3;
#pragma unroll
    for (int co = 0; co < 3; ++co) {
        const float v = fminf(fmaxf(ip[co * plane] * 0.5f + 0.5f, 0.f), 1.f);
        op[co] = (unsigned char)(v * 255.0f);
    }
}

// ---------------------------------------------------------------------------------------------
// k_convt_s32 — up1 = ConvTranspose2d(128, 64, 2, stride=2) (app.py:73,96) as a streaming kernel (round 3; the fp16 path's k_convt_t16
// is the same design, conv_kernels_f16.h).  A 2x2 stride-2 transposed convolution has no halo: every input pixel is read once and
// produces four output pixels.  k_gemm_conv MODE 2 ran it at 0.75 of the fp32 MFMA peak: one workgroup per (8x32 tile, tap), weights
// streamed from L2 per k-step, the input tile staged chunk by chunk behind barriers, a prologue and a staged epilogue per 256 MFMAs.  Here:
//   * wave = tap: its CIN x 64 weights stay in REGISTERS for the life of the workgroup (CIN VGPRs) as the A operands of
//     v_mfma_f32_16x16x4_f32 — channels are the MFMA rows, pixels the columns;
//   * persistent workgroups (two per CU) walk tiles of TP consecutive pixels of one image (32 KiB, one contiguous run of memory) brought
//     in by LDS-DMA into one of two buffers while the other is computed on: one barrier per tile;
//   * a pixel's 16-byte slots lie XOR-swizzled with the pixel index in LDS (phys = s ^ (p & 15)), applied on the source side of the DMA:
//     the pixel operand is read with conflict-free ds_read_b128 (one quad = the B values of four consecutive MFMAs) although pixels are
//     512 B apart;
//   * row 4 kg + r of M tile mt is channel 16 mt + 4 kg + r: a lane's four accumulator values are four consecutive channels and leave
//     as 16-byte stores straight from registers (four lanes = 64 contiguous bytes of a pixel), no staging.
// GemmConvArgs as this kernel reads it: tiles_x = tiles per image, tiles_total = N * tiles_x, rcp_x = tile_rcp(tiles_x),
// rcp_xy = tile_rcp(Win); Hin x Win = the input image (pixel stride in_ps floats), Hc x Wc = the part of it that is computed.
template <int CIN, int COUT>
struct ConvTGeom32 {
    static constexpr int NW = 4;                    // waves per workgroup = taps
    static constexpr int S = CIN / 4;               // 16-byte slots per pixel
    static constexpr int TP = 2048 / S;             // pixels per tile (32 KiB of input)
    static constexpr int TN = TP / 16;              // 16-pixel MFMA column tiles
    static constexpr int G = CIN / 16;              // 16-channel groups: one quad per lane = four k-steps
    static constexpr int PPR = NW * 64 / S;         // pixels per DMA round
    static constexpr int ROUNDS = TP / PPR;
    static constexpr int BUFQ = TP * S;             // quads per buffer
    static_assert(COUT == 64 && CIN == 128, "weights of one tap x 64 channels in CIN registers: up1");
    static_assert(16 % PPR == 0 && S >= 16, "the swizzle term (p & 15) repeats every 16 / PPR rounds");
};

template <int CIN, int COUT>
__global__ void __launch_bounds__(THREADS, 2) k_convt_s32(const GemmConvArgs a) {
    using Gm = ConvTGeom32<CIN, COUT>;
    constexpr int S = Gm::S, TP = Gm::TP, TN = Gm::TN, G = Gm::G, NW = Gm::NW, ROUNDS = Gm::ROUNDS, BUFQ = Gm::BUFQ, PPR = Gm::PPR;
    constexpr int NV = 16 / PPR;                    // distinct per-lane DMA offsets
    constexpr int TG = 2;
    static_assert(TN % TG == 0, "column tiles are processed TG at a time");
    __shared__ f32x4 lds[2 * BUFQ];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tap = wave, kh = tap >> 1, kw = tap & 1;
    const int c16 = lane & 15, kg = lane >> 4;
    int tile = blockIdx.x;
    if (tile >= a.tiles_total) return;
    const int HW = a.Hin * a.Win;

    // ---- this wave's weights and bias, once: wf[g][j][mt] = W[ci = 16 g + 4 (lane >> 4) + j][co = 16 mt + (lane & 15)][tap] ----
    float wf[G][4][4];
    {
        const float* wp = a.w + (size_t)tap * G * 16 * 64 + lane;
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) wf[g][j][mt] = wp[((g * 4 + j) * 4 + mt) * 64];
    }
    f32x4 bias4[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) bias4[mt] = *reinterpret_cast<const f32x4*>(a.bias + 16 * mt + 4 * kg);

    // ---- input DMA: lane L of (wave, round j) fills physical quad ((j * NW + wave) * 64 + L) of the buffer ----
    const int sp = lane & (S - 1);
    int pl[NV];
    unsigned dma_lane[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        pl[v] = ((v * NW + wave) * 64 + lane) / S;                          // tile-local pixel in round v (+ PPR * NV per NV rounds)
        dma_lane[v] = (unsigned)(pl[v] * a.in_ps * 4 + ((sp ^ (pl[v] & 15)) * 16));
    }
    auto image_rsrc = [&](int n) {
        const unsigned long long p = (unsigned long long)(a.in + (size_t)n * HW * a.in_ps);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), (short)0, HW * a.in_ps * 4, 0x00020000);
    };
    auto split = [&](int t, int& n, int& chunk) {                          // tile -> image, run of TP pixels inside it
        n = a.tiles_x > 1 ? (int)__umulhi((unsigned)t, a.rcp_x) : t;       // at most one too large (see the pixel decode below)
        chunk = t - n * a.tiles_x;
        if (chunk < 0) { --n; chunk += a.tiles_x; }
    };
    auto dma_tile = [&](int t, int buf) {
        int n, chunk;
        split(t, n, chunk);
        const __amdgpu_buffer_rsrc_t rsrc = image_rsrc(n);
#pragma unroll
        for (int j = 0; j < ROUNDS; ++j) {
            const int base = chunk * TP + (j / NV) * 16;                   // pixel of the image that round j - j % NV starts at
            const int q = base + pl[j % NV];                               // beyond the image: zeros.  The mask is carried by the PER-LANE
            const unsigned vo = q < HW ? dma_lane[j % NV] : 0x7ffffff0u;   // offset: nothing relies on the scalar offset being range-checked
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)&lds[buf * BUFQ + (j * NW + wave) * 64], 16, vo,
                                                     base * a.in_ps * 4, 0, 0);
        }
    };

    // ---- pixel operand: lane (c16, kg) reads slot 4 g + kg of pixel 16 t + c16: physical quad (16 t + c16) * S + ((4 g + kg) ^ c16) ----
    int rd[G];
#pragma unroll
    for (int g = 0; g < G; ++g) rd[g] = c16 * S + ((4 * g + kg) ^ c16);

    // ---- output: per image a buffer over [2 Hc][2 Wc] pixels of out_ps floats ----
    const int Ho = 2 * a.Hc, Wo = 2 * a.Wc;
    auto out_rsrc = [&](int n) {
        const unsigned long long p = (unsigned long long)(a.out + (size_t)n * Ho * Wo * a.out_ps);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), (short)0, Ho * Wo * a.out_ps * 4, 0x00020000);
    };
    const unsigned lane_chan = (unsigned)((a.out_coff + 4 * kg) * 4);
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
        for (int t0 = 0; t0 < TN; t0 += TG) {   // TG column tiles at a time: 16 TG accumulator registers beside the CIN of the weights
            f32x4 acc[TG][4];
            // the quads of group g + 1 are requested before the MFMAs of group g (hipcc sinks LDS reads to their first use otherwise: an
            // exposed LDS round trip per 32 MFMAs)
            f32x4 px[TG], pxn[TG];
#pragma unroll
            for (int t = 0; t < TG; ++t) px[t] = lds[buf * BUFQ + (t0 + t) * 16 * S + rd[0]];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (g + 1 < G) {
#pragma unroll
                    for (int t = 0; t < TG; ++t) pxn[t] = lds[buf * BUFQ + (t0 + t) * 16 * S + rd[g + 1]];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < TG; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) {
                            if (g == 0 && j == 0) {
                                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                                acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[g][j][mt], px[t][j], zero, 0, 0, 0);
                            } else {
                                acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[g][j][mt], px[t][j], acc[t][mt], 0, 0, 0);
                            }
                        }
                __builtin_amdgcn_sched_barrier(0);
                if (g + 1 < G) {
#pragma unroll
                    for (int t = 0; t < TG; ++t) px[t] = pxn[t];
                }
            }
            // The next tile's DMA (issued a whole tile of MFMAs ago) has landed for this wave before the LAST group's stores go out: waited
            // for here, with nothing but long-finished requests outstanding, because vector loads and stores may complete out of order
            // with respect to each other on gfx9 — a count taken after the stores would not single the loads out.
            if (has_next && t0 + TG == TN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // ---- bias (no activation: app.py:96), tap (kh, kw) of pixel (y, x) -> output pixel (2y + kh, 2x + kw) ----
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const unsigned q = (unsigned)(chunk * TP + (t0 + t) * 16 + c16);
                unsigned y = __umulhi(q, a.rcp_xy);                         // q / Win, at most one too large (q * (rcp * Win - 2^32) < 2^32 * Win)
                int x = (int)(q - y * (unsigned)a.Win);
                if (a.Win == 1) { y = q; x = 0; }
                if (x < 0) { --y; x += a.Win; }
                const bool ok = (int)q < HW && (int)y < a.Hc && x < a.Wc;
                const unsigned po = ((2 * y + kh) * (unsigned)Wo + 2 * x + kw) * (unsigned)a.out_ps * 4 + lane_chan;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const f32x4 v = acc[t][mt] + bias4[mt];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, ok ? po + mt * 64 : 0x7ffffff0u, 0, 0);
                }
            }
        }
        if (!has_next) break;
        __syncthreads();   // every wave's part of the next tile has landed, and every wave has left this tile's buffer
        tile = next;
        buf ^= 1;
    }
}

}  // namespace cid
