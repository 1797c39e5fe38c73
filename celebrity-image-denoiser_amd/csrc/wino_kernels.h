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
//     (pixel = 4 data slots + 1 pad slot of 16 B); the next chunk is fetched global -> registers ->
//     LDS under the current chunk's MFMAs, ONE barrier per chunk;
//   * B fragments (U, pre-packed per lane) stream L2 -> registers one unit ahead;
//   * epilogue: the halves exchange their partial output transforms through LDS, then half h writes
//     output row 2*tile_row + h (and half 0 the pooled row): bias, ReLU, 16-byte stores via LDS.
#pragma once
#include "conv_kernels.h"

namespace cid {

constexpr int WK = 16;        // channels per chunk
constexpr int WPS = 5;        // LDS slots (16 B) per pixel: 4 data + 1 pad
constexpr int WN = 32;        // output channels per workgroup
constexpr int WS32 = 36;      // staging row stride (floats) for 32-channel slabs

struct WinoArgs {
    const float* in;    // NHWC [N, Hin, Win, in_ps]
    const float* u;     // packed U: [nb][chunk][round][a][b][lane][4]
    const float* bias;  // [COUT]
    float* out;         // [N, Hs, Ws, out_ps] (+ out_coff)
    float* pool;        // POOL: [N, Hc/2, Wc/2, COUT]
    int N, Hin, Win, in_ps;
    int Hc, Wc, Hs, Ws;
    int out_ps, out_coff;
    int tiles_x, tiles_y, tiles_total, tiles_per_xcd;
};

// ABLATE (timing experiments only, csrc/tools/layer_bench.hip; results are wrong when non-zero):
//   bit 0: no halo prefetch after chunk 0   bit 1: B fragments loaded once   bit 2: A operand built once
//   bit 3: no epilogue                      bit 4: A operand read from LDS but not transformed
template <int CIN, int COUT, bool POOL, int TC, int ABLATE = 0>
__global__ void __launch_bounds__(THREADS, 2) k_wino_conv(const WinoArgs a) {
    constexpr int TRP = 32 / TC;                 // tile rows per pair
    constexpr int BTR = 2 * TRP;                 // tile rows per workgroup
    constexpr int LW = 2 * TC + 2, LH = 2 * BTR + 2, LPIX = LW * LH;
    constexpr int NSLOT = LPIX * 4;
    constexpr int NPIECE = (NSLOT + THREADS - 1) / THREADS;
    constexpr int BUF = LPIX * WPS;              // f32x4 slots per LDS buffer
    constexpr int NCHUNK = CIN / WK;
    constexpr int NB = COUT / WN;
    static_assert(CIN % WK == 0 && COUT % WN == 0 && (TC == 16 || TC == 32), "layer dims");
    static_assert(NPIECE <= 8, "halo pieces are spread over the first three units of a chunk");

    constexpr int EXCH = 4 * 16 * 64;            // epilogue exchange area: 4 waves x 16 registers x 64 lanes (f32x4)
    constexpr int LDS_SLOTS = (2 * BUF > EXCH) ? 2 * BUF : EXCH;
    __shared__ f32x4 lds[LDS_SLOTS];
    static_assert(4 * 64 * WS32 * sizeof(float) <= sizeof(lds), "store staging");

    int mt, nb;
    if (!decode_block(a.tiles_total, a.tiles_per_xcd, NB, mt, nb)) return;
    const int tx = mt % a.tiles_x;
    const int ty = (mt / a.tiles_x) % a.tiles_y;
    const int n = mt / (a.tiles_x * a.tiles_y);
    const int y0 = ty * (2 * BTR), x0 = tx * (2 * TC);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform in an SGPR
    const int pair = wave >> 1, half = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    const int tr = i / TC, tc = i - tr * TC;

    const float bias_v = a.bias[nb * WN + i];

    // rows of the 4x4 input patch that feed row a of B^T d:  t = x + sgn*y
    //   a=0: d0 - d2   a=1: d1 + d2   a=2: d2 - d1   a=3: d1 - d3
    const int xrow0 = half ? 2 : 0, yrow0 = half ? 1 : 2;
    const int xrow1 = 1, yrow1 = half ? 3 : 2;
    const float sgn0 = -1.f, sgn1 = half ? -1.f : 1.f;
    const int pbase = (2 * (pair * TRP + tr)) * LW + 2 * tc;
    const int xb0 = (pbase + xrow0 * LW) * WPS + h, yb0 = (pbase + yrow0 * LW) * WPS + h;
    const int xb1 = (pbase + xrow1 * LW) * WPS + h, yb1 = (pbase + yrow1 * LW) * WPS + h;

    // ---- halo pieces of this thread: piece `it` is data slot s = it*256 + tid of the raw tile ----
    const float* inb = a.in + (size_t)n * a.Hin * a.Win * a.in_ps;
    int goff[NPIECE];
    unsigned okmask = 0;
#pragma unroll
    for (int it = 0; it < NPIECE; ++it) {
        const int s = it * THREADS + tid;
        const int p = s >> 2, c = s & 3;
        const int hy = p / LW, hx = p - hy * LW;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool ok = (s < NSLOT) && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
        goff[it] = ok ? ((gy * a.Win + gx) * a.in_ps + c * 4) : 0;
        okmask |= (ok ? 1u : 0u) << it;
    }
    const int wslot = (tid >> 2) * WPS + (tid & 3);   // piece `it` lands at wslot + it*64*WPS
    auto halo_load = [&](int it, int ck) -> f32x4 {
        const float* cb = inb + ck * WK;                   // wave-uniform
        return *reinterpret_cast<const f32x4*>(cb + goff[it]);
    };
    auto halo_store = [&](int buf, int it, f32x4 v) {
        if (!((okmask >> it) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};   // zero padding of the convolution
        if ((it + 1) * THREADS <= NSLOT || it * THREADS + tid < NSLOT) lds[buf * BUF + wslot + it * 64 * WPS] = v;
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][b][r] = 0.f;

    // U stream of this wave: unit (ck, g2, u) is 4 quads of 1 KiB at ((ck*2+g2)*4 + 2*half+u)*4*64 f32x4
    // (wave-uniform base in SGPRs + the lane index as a 32-bit offset: no 64-bit VALU address arithmetic)
    const f32x4* up = reinterpret_cast<const f32x4*>(a.u) + ((size_t)nb * NCHUNK * 2 * 16 + 2 * half * 4) * 64;
    auto load_b = [&](f32x4 (&dst)[4], int unit_in_chunk, int ck) {   // unit_in_chunk = g2*2 + u
        const int g2 = unit_in_chunk >> 1, u = unit_in_chunk & 1;
        const f32x4* q = up + ((size_t)((ck * 2 + g2) * 4 + u) * 4) * 64;
#pragma unroll
        for (int b = 0; b < 4; ++b) dst[b] = q[b * 64 + lane];
    };

    // ---- prologue: chunk 0 -> LDS buffer 0, first B unit ----
    {
        f32x4 pre[NPIECE];
#pragma unroll
        for (int it = 0; it < NPIECE; ++it) pre[it] = halo_load(it, 0);
#pragma unroll
        for (int it = 0; it < NPIECE; ++it) halo_store(0, it, pre[it]);
    }
    f32x4 bq[2][4];
    load_b(bq[0], 0, 0);
    __syncthreads();

    auto chunk = [&](auto more_tag, int ck) {
        constexpr bool MORE = decltype(more_tag)::value;   // another chunk follows: prefetch it
        constexpr bool PREF = MORE && !(ABLATE & 1);
        const int cur = (ABLATE & 1) ? 0 : (ck & 1) * BUF;
        f32x4 hp[3];
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {                      // unit k = (round g2 = k>>1, row u = k&1)
            const int g2 = k >> 1, u = k & 1;
            // next unit's B fragments
            if (!(ABLATE & 2)) {
                if (k < 3) load_b(bq[(k + 1) & 1], k + 1, ck);
                else if (MORE) load_b(bq[0], 0, ck + 1);
            }
            // halo pieces of the next chunk: units 0,1,2 each issue up to 3 and write the previous unit's
            if (PREF) {
                if (k >= 1) {
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if ((k - 1) * 3 + j < NPIECE) halo_store(((ck + 1) & 1), (k - 1) * 3 + j, hp[j]);
                }
                if (k < 3) {
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if (k * 3 + j < NPIECE) hp[j] = halo_load(k * 3 + j, ck + 1);
                }
            }
            // A operand: V[a][0..3] for 4 channels, rebuilt from rows x,y of the raw patch
            const int xb = (u ? xb1 : xb0) + 2 * g2, yb = (u ? yb1 : yb0) + 2 * g2;
            const float sg = u ? sgn1 : sgn0;
            if (!(ABLATE & 4) || (k == 0 && ck == 0)) {
                f32x4 t[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 xv = lds[cur + xb + c * WPS];
                    const f32x4 yv = lds[cur + yb + c * WPS];
                    if (ABLATE & 16) { t[c] = xv; t[c][0] += yv[1]; continue; }
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[c][e] = __builtin_fmaf(sg, yv[e], xv[e]);
                }
                if (ABLATE & 16) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = t[c];
                } else {
                    v[0] = t[0] - t[2];
                    v[1] = t[1] + t[2];
                    v[2] = t[2] - t[1];
                    v[3] = t[1] - t[3];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[u][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[b][e], bq[(ABLATE & 2) ? 0 : (k & 1)][b][e], acc[u][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (PREF) __syncthreads();   // next buffer is complete, and everybody is done reading this one
    };
    for (int ck = 0; ck + 1 < NCHUNK; ++ck) chunk(std::true_type{}, ck);
    chunk(std::false_type{}, NCHUNK - 1);

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
    const float c01 = half ? 0.f : 1.f, c10 = half ? -1.f : 0.f, c11 = half ? -1.f : 1.f;
    __syncthreads();                                    // raw tiles are dead: LDS becomes exchange + staging
    f32x4* ex = lds + wave * (16 * 64);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float m00 = acc[0][0][r] + acc[0][1][r] + acc[0][2][r], m01 = acc[0][1][r] - acc[0][2][r] - acc[0][3][r];
        const float m10 = acc[1][0][r] + acc[1][1][r] + acc[1][2][r], m11 = acc[1][1][r] - acc[1][2][r] - acc[1][3][r];
        f32x4 p;
        p[0] = __builtin_fmaf(c01, m10, m00);
        p[1] = __builtin_fmaf(c01, m11, m01);
        p[2] = c10 * m00 + c11 * m10;
        p[3] = c10 * m01 + c11 * m11;
        ex[r * 64 + lane] = p;
        acc[0][0][r] = p[0]; acc[0][1][r] = p[1]; acc[0][2][r] = p[2]; acc[0][3][r] = p[3];
    }
    __syncthreads();
    const f32x4* exo = lds + (wave ^ 1) * (16 * 64);
    float yrow[2][16];   // this wave's output row a' = half: columns b' = 0,1 of each tile
    float pooled[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const f32x4 o = exo[r * 64 + lane];
        // add in a fixed order (half 0's partial first) so both halves see bit-identical Y
        const float y00 = half ? o[0] + acc[0][0][r] : acc[0][0][r] + o[0];
        const float y01 = half ? o[1] + acc[0][1][r] : acc[0][1][r] + o[1];
        const float y10 = half ? o[2] + acc[0][2][r] : acc[0][2][r] + o[2];
        const float y11 = half ? o[3] + acc[0][3][r] : acc[0][3][r] + o[3];
        yrow[0][r] = fmaxf((half ? y10 : y00) + bias_v, 0.f);
        yrow[1][r] = fmaxf((half ? y11 : y01) + bias_v, 0.f);
        if (POOL) pooled[r] = fmaxf(fmaxf(fmaxf(y00, y01), fmaxf(y10, y11)) + bias_v, 0.f);
    }
    __syncthreads();                                    // exchange area is dead: reuse as store staging
    float* stg = reinterpret_cast<float*>(lds) + wave * (64 * WS32);
    auto tile_of = [&](int r) { return (r & 3) + 8 * (r >> 2) + 4 * h; };   // D row = tile index of register r
    {
        // staged pixel sp = 2*tile + b'  ->  output (y, x)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = tile_of(r);
            stg[(2 * t) * WS32 + i] = yrow[0][r];
            stg[(2 * t + 1) * WS32 + i] = yrow[1][r];
        }
        wave_lds_fence();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int sp = it * 8 + (lane >> 3);
            const int t = sp >> 1, ttr = t / TC, ttc = t - ttr * TC;
            const int y = y0 + 2 * (pair * TRP + ttr) + half, x = x0 + 2 * ttc + (sp & 1);
            const f32x4 v = *reinterpret_cast<const f32x4*>(stg + sp * WS32 + (lane & 7) * 4);
            if (y < a.Hs && x < a.Ws)
                *reinterpret_cast<f32x4*>(a.out + ((size_t)(n * a.Hs + y) * a.Ws + x) * a.out_ps + a.out_coff + nb * WN + (lane & 7) * 4) = v;
        }
        wave_lds_fence();
    }
    if (POOL && half == 0) {
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
}

}  // namespace cid
