// cid_api.hip — the C ABI of include/cid.h: weight repacking, workspace planning and the launch
// sequence of one forward.  Host code; the device kernels are in conv_kernels.h.
//
// Layer table = the reference module's declaration order, backend/app.py:42-78.
#include "../../include/cid.h"
#include "conv_kernels.h"
#include "wino64_kernels.h"
#include "wino42_kernels.h"
#include "conv_kernels_f16.h"

#include <dlfcn.h>

#include <atomic>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

using namespace cid;

struct ncclUniqueIdBytes { char internal[128]; };   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value

enum Kind { HEAD, CONV, CONVT, TAIL };
struct LayerDef { const char* name; Kind kind; int cin, cout; };
constexpr int NL = 12;
const LayerDef kLayers[NL] = {
    {"down1.0", HEAD, 3, 64},        {"down1.2", CONV, 64, 64},        {"down2.0", CONV, 64, 128},
    {"down2.2", CONV, 128, 128},     {"bottleneck.0", CONV, 128, 256}, {"bottleneck.2", CONV, 256, 256},
    {"up2", CONVT, 256, 128},        {"upconv2.0", CONV, 256, 128},    {"upconv2.2", CONV, 128, 128},
    {"up1", CONVT, 128, 64},         {"upconv1.0", CONV, 128, 64},     {"upconv1.2", TAIL, 64, 3},
};
const char* kKernelNames[NL] = {
    "k_conv_head", "k_gemm_conv<64, 64, 1,", "k_gemm_conv<64, 128, 0,", "k_gemm_conv<128, 128, 1,",
    "k_gemm_conv<128, 256, 0,", "k_gemm_conv<256, 256, 0,", "k_gemm_conv<256, 128, 2,", "k_gemm_conv<256, 128, 0,",
    "k_gemm_conv<128, 128, 0,", "k_convt_s32<128, 64>", "k_gemm_conv<128, 64, 0,", "k_conv_tail",
};

const char* kHalfKernelNames[NL] = {
    "k_conv_head_h16", "k_conv3x3_h16<64, 64, 1,", "k_conv3x3_h16<64, 128, 0,", "k_conv3x3_h16<128, 128, 1,",
    "k_conv3x3_h16<128, 256, 0,", "k_conv3x3_h16<256, 256, 0,", "k_convt_t16<256, 128>", "k_conv3x3_h16<256, 128, 0,",
    "k_conv3x3_h16<128, 128, 0,", "k_convt_t16<128, 64>", "k_conv3x3_h16<128, 64, 0, false,", "k_conv_tail_h<",
};
const char* kWino64KernelNames[NL] = {
    nullptr, "k_wino64_conv<64, 64, true,", "k_wino64_conv<64, 128, false,", "k_wino64_conv<128, 128, true,",
    "k_wino64_conv<128, 256, false,", "k_wino64_conv<256, 256, false,", nullptr, "k_wino64_conv<256, 128, false,",
    "k_wino64_conv<128, 128, false,", nullptr, "k_wino64_conv<128, 64, false,", nullptr,
};
const char* kSplitKernelNames[NL] = {   // conv_algo = "split16": the eight 3x3 layers on k_conv3x3_h16<..., F32IO = true>; everything else as the direct configuration
    nullptr, "k_conv3x3_h16<64, 64, 1, false, false, true,", "k_conv3x3_h16<64, 128, 0, false, false, true,", "k_conv3x3_h16<128, 128, 1, false, false, true,",
    "k_conv3x3_h16<128, 256, 0, false, false, true,", "k_conv3x3_h16<256, 256, 0, false, false, true,", "k_conv3x3_h16<256, 128, 2, false, false, true,", "k_conv3x3_h16<256, 128, 0, false, false, true,",
    "k_conv3x3_h16<128, 128, 0, false, false, true,", "k_conv3x3_h16<128, 64, 2, false, false, true,", "k_conv3x3_h16<128, 64, 0, false, false, true,", nullptr,
};
const char* kWino42KernelNames[NL] = {
    nullptr, "k_wino42_conv<64, 64, true,", "k_wino42_conv<64, 128, false,", "k_wino42_conv<128, 128, true,",
    "k_wino42_conv<128, 256, false,", "k_wino42_conv<256, 256, false,", nullptr, "k_wino42_conv<256, 128, false,",
    "k_wino42_conv<128, 128, false,", nullptr, "k_wino42_conv<128, 64, false,", nullptr,
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline unsigned long long cdiv_ull(unsigned long long a, unsigned long long b) { return (a + b - 1) / b; }

size_t ref_weight_count(const LayerDef& L) { return (size_t)L.cin * L.cout * (L.kind == CONVT ? 4 : 9); }
size_t packed_weight_count(const LayerDef& L) {
    switch (L.kind) {
        case HEAD: return 2 * 14 * 64;
        case TAIL: return 2 * 4 * 64 * 4;   // [chunk][group][lane][4], columns 27..31 stay zero
        default: return ref_weight_count(L);
    }
}

// float offsets of each layer's packed weights / bias inside the blob; segments 256-byte aligned
//   [direct-kernel segments][Winograd U of the eight 3x3 GEMM layers][reference-layout copy of all 24 tensors]
// The reference-layout copy makes the blob self-describing (state_dict() after a broadcast is exact: U is
// not invertible bit-for-bit).
struct BlobLayout {
    size_t w_off[NL], b_off[NL], u_off[NL], u42_off[NL], h_off[NL], raw_w_off[NL], raw_b_off[NL], tab_off[2], tab42_off[2], hz_off, s_off[NL], hzs_off, total;
    BlobLayout() {
        size_t o = 0;
        for (int l = 0; l < NL; ++l) {
            w_off[l] = o; o = align_up(o + packed_weight_count(kLayers[l]), 64);
            b_off[l] = o; o = align_up(o + kLayers[l].cout, 64);
        }
        for (int l = 0; l < NL; ++l) {
            u_off[l] = o;
            if (kLayers[l].kind == CONV) o = align_up(o + (size_t)kLayers[l].cin * kLayers[l].cout * 16, 64);
        }
        for (int l = 0; l < NL; ++l) {   // fp16-storage path: half weights (2 per float slot), in each kernel's fragment order
            h_off[l] = o;
            if (kLayers[l].kind == CONV || kLayers[l].kind == CONVT) o = align_up(o + (ref_weight_count(kLayers[l]) + 1) / 2, 64);
            if (kLayers[l].kind == TAIL) o = align_up(o + 4 * 64 * 8 / 2, 64);   // k_conv_tail_h: [4 k-steps][64 lanes][8] halfs
            if (kLayers[l].kind == HEAD) o = align_up(o + 4 * 64 * 8 / 2, 64);   // k_conv_head_h16: [4 channel groups][64 lanes][8] halfs
        }
        for (int l = 0; l < NL; ++l) {
            raw_w_off[l] = o; o = align_up(o + ref_weight_count(kLayers[l]), 64);
            raw_b_off[l] = o; o = align_up(o + kLayers[l].cout, 64);
        }
        tab_off[0] = o; o = align_up(o + wino_slot_table(32, 1, nullptr), 64);   // Winograd LDS slot tables, TC = 32 and 16
        tab_off[1] = o; o = align_up(o + wino_slot_table(16, 2, nullptr), 64);
        for (int l = 0; l < NL; ++l) {   // Winograd F(4x2,3x3): U at 24 positions
            u42_off[l] = o;
            if (kLayers[l].kind == CONV) o = align_up(o + (size_t)kLayers[l].cin * kLayers[l].cout * 24, 64);
        }
        tab42_off[0] = o; o = align_up(o + wino42_slot_table<8>(nullptr), 64);   // its LDS slot tables, TC = 8 and 4
        tab42_off[1] = o; o = align_up(o + wino42_slot_table<4>(nullptr), 64);
        hz_off = o; o = align_up(o + 3 * 2 * 64 * 8 / 2, 64);   // fp16 path, fused last layer: upconv1[2] as A fragments [2 row tiles][2 k-steps][64 lanes][8] halfs (room for three, as round 4's first form had)
        for (int l = 0; l < NL; ++l) {   // conv_algo = "split16": hi | lo | hi half pieces of the eight 3x3 layers' weights, in the order k_conv3x3_h16<F32IO> consumes them (27 halfs per weight)
            s_off[l] = o;
            if (kLayers[l].kind == CONV) o = align_up(o + ((size_t)kLayers[l].cin * kLayers[l].cout * 27 + 1) / 2, 64);
        }
        for (int l = 0; l < NL; ++l)   // split16: the two transposed convolutions' hi | lo | hi pieces, [block = (tap, 64 channels)][chunk][piece][cg][lane][8] halfs (12 halfs per weight)
            if (kLayers[l].kind == CONVT) { s_off[l] = o; o = align_up(o + ((size_t)kLayers[l].cin * kLayers[l].cout * 12 + 1) / 2, 64); }
        hzs_off = o; o = align_up(o + 2 * 2 * 2 * 64 * 8 / 2, 64);   // split16, fused last layer: upconv1[2] as A fragments [hi | lo][2 row tiles][2 k-steps][64 lanes][8] halfs
        total = o;
    }
};
const BlobLayout kBlob;

// Index in the packed segment of reference weight element (co, ci, kh, kw) of layer L.
// GEMM layers: [nb][chunk][tap][group g][ns][lane = 32*h + j][e] with
//   ci = 32*chunk + 8*g + 4*h + e  and  n' = 64*nb + 32*ns + j  (n' = co, or tap*COUT + co for convT)
// — lane (h, j) of v_mfma_f32_32x32x2_f32 holds B[k = h][col = j]; e walks the 4 MFMAs of a group.
size_t packed_index(const LayerDef& L, int co, int ci, int kh, int kw) {
    switch (L.kind) {
        case HEAD: {
            int s, h;
            head_step_of(ci * 9 + kh * 3 + kw, s, h);   // the head's K order (conv_kernels.h head_step)
            return (size_t)((co >> 5) * 14 + s) * 64 + h * 32 + (co & 31);
        }
        case TAIL: {   // B[k = ci][col = 3*tap + co] of the tail's 64 x 32 product
            const int col = (kh * 3 + kw) * 3 + co, ck = ci >> 5, g = (ci >> 3) & 3, h = (ci >> 2) & 1, e = ci & 3;
            return (size_t)((ck * 4 + g) * 64 + h * 32 + col) * 4 + e;
        }
        case CONVT:
            if (L.cin == 128) {   // up1, k_convt_s32: [tap][g][j][mt][lane = 16*kga + row] — A[row][k = kga] of v_mfma_f32_16x16x4_f32 for k-step j
                const int g = ci >> 4, kga = (ci >> 2) & 3, j = ci & 3, mt = co >> 4, row = co & 15;   // of group g: ci = 16 g + 4 kga + j, co = 16 mt + row
                return (((((size_t)(kh * 2 + kw) * (L.cin / 16) + g) * 4 + j) * 4 + mt) * 64) + kga * 16 + row;
            }
            [[fallthrough]];
        case CONV: {
            const int taps = L.kind == CONV ? 9 : 1;
            const int tap = L.kind == CONV ? kh * 3 + kw : 0;
            const int np = L.kind == CONV ? co : (kh * 2 + kw) * L.cout + co;
            const int nb = np >> 6, ns = (np >> 5) & 1, j = np & 31;
            const int ck = ci >> 5, g = (ci >> 3) & 3, h = (ci >> 2) & 1, e = ci & 3;
            const int nchunk = L.cin / 32;
            return ((((size_t)(nb * nchunk + ck) * taps + tap) * 4 + g) * 2 + ns) * 256 + (h * 32 + j) * 4 + e;
        }
    }
    return 0;
}
// Index of (co, ci, kh, kw) in the reference tensor: Conv2d [Cout,Cin,3,3]; ConvTranspose2d [Cin,Cout,2,2].
size_t ref_index(const LayerDef& L, int co, int ci, int kh, int kw) {
    if (L.kind == CONVT) return (((size_t)ci * L.cout + co) * 2 + kh) * 2 + kw;
    return (((size_t)co * L.cin + ci) * 3 + kh) * 3 + kw;
}

// Winograd F(2x2,3x3) filter transform U = G g G^T (reference Conv2d weight [Cout,Cin,3,3] -> 16 values per
// (co, ci)), in double, rounded once to fp32, laid out for k_wino64_conv:
//   [nb = co/64][chunk = ci/16][round = (ci/8)%2][a][nt = (co/32)%2][e][lane = 32*h + j][b],  ci = 16*chunk + 8*round + 4*h + e, co = 64*nb + 32*nt + j
// (one 16-byte quad per lane = the four positions b of k-step e, so a quad's registers free up after 4 MFMAs)
void pack_winograd_u(const LayerDef& L, const float* w, float* dst) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int nchunk = L.cin / 16;
    for (int co = 0; co < L.cout; ++co)
        for (int ci = 0; ci < L.cin; ++ci) {
            const float* g = w + ((size_t)co * L.cin + ci) * 9;
            double tmp[4][3];
            for (int a = 0; a < 4; ++a)
                for (int q = 0; q < 3; ++q) tmp[a][q] = G[a][0] * g[0 * 3 + q] + G[a][1] * g[1 * 3 + q] + G[a][2] * g[2 * 3 + q];
            const int nb = co >> 5, j = co & 31, ck = ci >> 4, g2 = (ci >> 3) & 1, h = (ci >> 2) & 1, e = ci & 3;
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b) {
                    const double u = tmp[a][0] * G[b][0] + tmp[a][1] * G[b][1] + tmp[a][2] * G[b][2];
                    dst[(((((((size_t)(nb >> 1) * nchunk + ck) * 2 + g2) * 4 + a) * 2 + (nb & 1)) * 4 + e) * 64 + h * 32 + j) * 4 + b] = (float)u;
                }
        }
}

// Winograd F(4x2,3x3) filter transform U = G2 g G4^T — rows by the F(2,3) matrix of pack_winograd_u, columns by F(4,3) at the
// points 0, 3/4, -3/4, 3/2, -3/2, inf (24 values per (co, ci)) —, in double, rounded once to fp32, laid out for k_wino42_conv:
//   [nb = co/64][unit][a][q = 6*e2 + b][lane = 16*g + j][cg],   ci = 16*(unit/2) + 4*g + 2*((unit%2) ^ (g&1)) + e2,  co = 64*nb + 4*j + cg
// (one 16-byte quad per lane = the four channel groups of position (a, b) at k-step e2: one V value, four MFMAs)
void pack_winograd42_u(const LayerDef& L, const float* w, float* dst) {
    static const double G2[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    static const double G4[6][3] = {{64.0 / 81, 0, 0},
                                    {-128.0 / 243, -32.0 / 81, -8.0 / 27},
                                    {-128.0 / 243, 32.0 / 81, -8.0 / 27},
                                    {32.0 / 243, 16.0 / 81, 8.0 / 27},
                                    {32.0 / 243, -16.0 / 81, 8.0 / 27},
                                    {0, 0, 1}};
    const int nunit = L.cin / 8;
    for (int co = 0; co < L.cout; ++co)
        for (int ci = 0; ci < L.cin; ++ci) {
            const float* g = w + ((size_t)co * L.cin + ci) * 9;
            double tmp[4][3];
            for (int a = 0; a < 4; ++a)
                for (int q = 0; q < 3; ++q) tmp[a][q] = G2[a][0] * g[0 * 3 + q] + G2[a][1] * g[1 * 3 + q] + G2[a][2] * g[2 * 3 + q];
            const int nb = co >> 6, j = (co >> 2) & 15, cg = co & 3;   // column j of channel group cg = channel 4j + cg: see the kernel's epilogue
            // lane group gg reads the 8-byte half (ci >> 1) & 1 of its LDS quad for unit s2 = half ^ (gg & 1): odd channel groups take
            // the halves in the other order, which makes the kernel's ds_read_b64 conflict-free (wino42_kernels.h, xbase / ybase)
            const int ck = ci >> 4, gg = (ci >> 2) & 3, s2 = ((ci >> 1) & 1) ^ (gg & 1), e2 = ci & 1;
            const int unit = ck * 2 + s2;
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 6; ++b) {
                    const double u = tmp[a][0] * G4[b][0] + tmp[a][1] * G4[b][1] + tmp[a][2] * G4[b][2];
                    dst[((((((size_t)nb * nunit + unit) * 4 + a) * 12 + (6 * e2 + b)) * 64) + gg * 16 + j) * 4 + cg] = (float)u;
                }
        }
}

// k_convt_t16: [tap][cb][k-step][mt][lane = 16*kga + row][8] halfs — the A operand of v_mfma_f32_16x16x32_f16: A[row][k = 8 kga + e] with
// ci = 32*kstep + 8*kga + e and row `row` of M tile `mt` = channel 64 cb + 32 (mt >> 1) + 8 (row >> 2) + 4 (mt & 1) + (row & 3).
size_t packed_index_ht(const LayerDef& L, int co, int ci, int kh, int kw) {
    const int tap = kh * 2 + kw, cb = co >> 6, c = co & 63;
    const int mt = ((c >> 5) << 1) | ((c >> 2) & 1), row = (((c >> 3) & 3) << 2) | (c & 3);
    const int ks = ci >> 5, kga = (ci >> 3) & 3, e = ci & 7;
    const int CB = L.cout / 64, KS = L.cin / 32;
    return (((((size_t)(tap * CB + cb) * KS + ks) * 4 + mt) * 64) + kga * 16 + row) * 8 + e;
}

// k_conv3x3_h16: [nb][32-ch chunk][dx][dy][cg = co%4][lane = 16*kg + (co/4)%16][8] halfs with ci = 32*chunk + 8*kg + e —
// lane (col, kg) of v_mfma_f32_16x16x32_f16 holds B[k = 8kg..8kg+7][col]; column col of channel group cg is output channel
// 64 nb + 4 col + cg, so a lane's four accumulator tiles are four consecutive channels (the kernel stores them as 8 bytes);
// one (chunk, dx) is a 12 KiB LDS-DMA unit.
size_t packed_index_h16(const LayerDef& L, int co, int ci, int kh, int kw) {
    const int nb = co >> 6, cg = co & 3, c = (co >> 2) & 15;
    const int ck = ci >> 5, kg = (ci >> 3) & 3, e = ci & 7;
    const int nchunk = L.cin / 32;
    return ((((((size_t)(nb * nchunk + ck) * 3 + kw) * 3 + kh) * 4 + cg) * 64) + kg * 16 + c) * 8 + e;
}

// conv_algo = "split16" (k_conv3x3_h16<F32IO>): [column block nb][chunk ck of 32 fp32 channels][j = 0..8][kh][channel group cg][lane = 16 kg + col][8] halfs, where sub-chunk
// j = 0..2 is tap column kw = j of hi_w (met by hi_x), j = 3..5 kw = j - 3 of lo_w (met by hi_x), j = 6..8 kw = j - 6 of hi_w again (met by lo_x); hi_w = half(w), lo_w = half(w - hi_w)
size_t packed_index_s16(const LayerDef& L, int co, int ci, int kh, int j) {
    const int nb = co >> 6, cg = co & 3, c = (co >> 2) & 15;
    const int ck = ci >> 5, kg = (ci >> 3) & 3, e = ci & 7;
    const int nchunk = L.cin / 32;
    return ((((((size_t)(nb * nchunk + ck) * 9 + j) * 3 + kh) * 4 + cg) * 64) + kg * 16 + c) * 8 + e;
}

struct Dims {
    int N, H, W, H1, W1, H2, W2, Hu2, Wu2, Hu1, Wu1;
};
bool make_dims(int N, int H, int W, Dims& d) {
    if (N < 1 || H < 4 || W < 4) return false;
    d.N = N; d.H = H; d.W = W;
    d.H1 = H / 2; d.W1 = W / 2; d.H2 = d.H1 / 2; d.W2 = d.W1 / 2;       // floor-mode pools, app.py:48,56
    d.Hu2 = 2 * d.H2; d.Wu2 = 2 * d.W2; d.Hu1 = 2 * d.Hu2; d.Wu1 = 2 * d.Wu2;  // x2 transposed convs, app.py:65,73
    return true;
}

// Why an [N,3,H,W] forward is not accepted, or nullptr.  Besides the reference's own limit (H, W >= 4: ATen raises "Output
// size is too small", app.py:80-103) there are two of this implementation's:
//   * the kernels address ONE image's activations through a raw buffer descriptor with 32-bit byte offsets, and a lane
//     that must deliver zeros (convolution padding, ragged tiles) carries the offset 0x7ffffff0, which has to lie beyond
//     the descriptor's range.  The widest per-image tensor is cat1 = [4*(H/4), 4*(W/4), 128] fp32 (512 bytes per pixel);
//   * tile decode divides by multiply-high with 32-bit reciprocals: exact while (tiles of the launch) x (tiles per image) < 2^32,
//     i.e. N x t^2 < 2^32 with t = the most tiles (or blocks) per image of any launch: every tiling in use is listed below.
// Larger inputs are an error (CID_ERR_SHAPE), never wrong results: split the batch, or the image into stripes
// (api.serve_u8 does).
constexpr unsigned long long kZeroSentinel = 0x7ffffff0ull;
const char* shape_error(int N, int H, int W, Dims& d) {
    if (!make_dims(N, H, W, d)) return "N >= 1 and H, W >= 4 required (output size is too small)";
    if ((unsigned long long)H * W * 512ull >= kZeroSentinel)
        return "image too large for one call: H*W must stay below 4,194,303 pixels (32-bit per-image addressing); split it into stripes";
    unsigned long long t0 = 0;
    const unsigned long long per_image[] = {
        cdiv_ull(W, 32) * cdiv_ull(H, 4), cdiv_ull(W, 16) * cdiv_ull(H, 8),          // k_wino42_conv: 8 x 2 / 4 x 4 tiles of 4x2 pixels
        cdiv_ull(W, 64) * cdiv_ull(H, 2), cdiv_ull(W, 32) * cdiv_ull(H, 4),          // k_wino64_conv: 32 x 1 / 16 x 2 tiles of 2x2 pixels
        cdiv_ull(W, TILE_W) * cdiv_ull(H, TILE_H),                                   // head, tail, k_gemm_conv, the fp16 kernels
        cdiv_ull((unsigned long long)H * W, THREADS),                                // k_conv_tail_z: one thread per pixel
    };
    for (unsigned long long t : per_image) t0 = t > t0 ? t : t0;
    if ((unsigned long long)N * t0 * t0 >= (1ull << 32)) return "batch x image too large for one call (split the batch)";
    return nullptr;
}

// activation arena (floats), NHWC
enum Buf { T0, CAT1, P1, T1, CAT2, P2, T2, BT, T3, D2, T4, NBUF };
struct Plan { size_t off[NBUF]; size_t total_bytes; };
Plan make_plan(const Dims& d) {
    const size_t s0 = (size_t)d.N * d.H * d.W, s1 = (size_t)d.N * d.H1 * d.W1, s2 = (size_t)d.N * d.H2 * d.W2;
    const size_t su2 = (size_t)d.N * d.Hu2 * d.Wu2, su1 = (size_t)d.N * d.Hu1 * d.Wu1;
    const size_t sz[NBUF] = {s0 * 64, su1 * 128, s1 * 64, s1 * 128, su2 * 256, s2 * 128, s2 * 256, s2 * 256, su2 * 128, su2 * 128, su1 * 64};
    Plan p; size_t o = 0;
    for (int b = 0; b < NBUF; ++b) { p.off[b] = o; o = align_up(o + sz[b], 64); }
    p.total_bytes = o * sizeof(float);
    return p;
}

}  // namespace

struct cid_handle_s {
    std::vector<float> staging;        // packed host blob
    bool have[NL][2];
    const float* dev_blob = nullptr;
    std::string err;
    int dtype = CID_DTYPE_F32;         // storage type of activations/weights between the first and last kernel
    int tail_algo = CID_TAIL_FUSED;    // last layer: see cid_set_tail_algo
    int algo = CID_ALGO_WINOGRAD42;    // 3x3 GEMM layers: Winograd F(4x2,3x3) (default), Winograd F(2x2,3x3) or CID_ALGO_DIRECT (9-tap implicit GEMM)
    std::vector<hipEvent_t> tev;       // armed timing events, (NL+1) per forward
    int tev_forwards = 0, tev_used = 0;
    cid_handle_s() : staging(kBlob.total, 0.f) {
        std::memset(have, 0, sizeof(have));
        wino_slot_table(32, 1, reinterpret_cast<unsigned*>(staging.data() + kBlob.tab_off[0]));
        wino_slot_table(16, 2, reinterpret_cast<unsigned*>(staging.data() + kBlob.tab_off[1]));
        wino42_slot_table<8>(reinterpret_cast<unsigned*>(staging.data() + kBlob.tab42_off[0]));
        wino42_slot_table<4>(reinterpret_cast<unsigned*>(staging.data() + kBlob.tab42_off[1]));
    }
};

namespace {

// The fused form lives in the epilogue of upconv1[0]'s kernel: the Winograd kernels on the fp32 path (not the 9-tap direct one), k_conv3x3_h16 on the
// fp16-storage path (one 3x3 algorithm there, so always).
bool fused_tail_active(cid_handle_t h) {
    return h->tail_algo == CID_TAIL_FUSED && (h->dtype == CID_DTYPE_F16 || h->algo != CID_ALGO_DIRECT);
}

int fail(cid_handle_t h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}

bool find_key(const char* key, int& layer, int& is_bias) {
    if (!key) return false;
    const std::string k(key);
    for (int l = 0; l < NL; ++l) {
        const std::string n(kLayers[l].name);
        if (k == n + ".weight") { layer = l; is_bias = 0; return true; }
        if (k == n + ".bias") { layer = l; is_bias = 1; return true; }
    }
    return false;
}

template <typename F>
void for_each_weight(const LayerDef& L, F f) {
    const int kk = L.kind == CONVT ? 2 : 3;
    for (int co = 0; co < L.cout; ++co)
        for (int ci = 0; ci < L.cin; ++ci)
            for (int kh = 0; kh < kk; ++kh)
                for (int kw = 0; kw < kk; ++kw) f(co, ci, kh, kw);
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

struct TileGrid { int tx, ty, total, per_xcd; };
TileGrid tiles_for(int N, int Hc, int Wc) {
    TileGrid g;
    g.tx = cdiv(Wc, TILE_W); g.ty = cdiv(Hc, TILE_H);
    g.total = N * g.tx * g.ty;
    g.per_xcd = cdiv(g.total, 8);
    return g;
}

template <int CIN, int COUT, int MODE>
hipError_t launch_gemm(hipStream_t s, const float* blob, int layer, const float* in, int Hin, int Win, int in_ps,
                       float* out, int out_ps, int out_coff, int Hc, int Wc, int Hs, int Ws, float* pool, int N) {
    GemmConvArgs a;
    a.in = in; a.w = blob + kBlob.w_off[layer]; a.bias = blob + kBlob.b_off[layer];
    a.out = out; a.pool = pool;
    a.N = N; a.Hin = Hin; a.Win = Win; a.in_ps = in_ps;
    a.Hc = Hc; a.Wc = Wc; a.Hs = Hs; a.Ws = Ws; a.out_ps = out_ps; a.out_coff = out_coff;
    const TileGrid g = tiles_for(N, Hc, Wc);
    a.tiles_x = g.tx; a.tiles_y = g.ty; a.tiles_total = g.total; a.tiles_per_xcd = g.per_xcd;
        a.rcp_x = tile_rcp(g.tx); a.rcp_xy = tile_rcp(g.tx * g.ty);
    constexpr int NB = (MODE == 2 ? 4 * COUT : COUT) / NTILE;
    hipLaunchKernelGGL((k_gemm_conv<CIN, COUT, MODE>), dim3(8 * g.per_xcd * NB), dim3(THREADS), 0, s, a);
    return hipGetLastError();
}

template <int CIN, int COUT, bool POOL, int TC>
hipError_t launch_wino64_tc(hipStream_t s, const WinoArgs& base) {
    WinoArgs a = base;
    constexpr int TRW = 32 / TC;
    a.tiles_x = cdiv(a.Wc, 2 * TC); a.tiles_y = cdiv(a.Hc, 2 * TRW);
    a.tiles_total = a.N * a.tiles_x * a.tiles_y; a.tiles_per_xcd = cdiv(a.tiles_total, 8);
    a.rcp_x = tile_rcp(a.tiles_x); a.rcp_xy = tile_rcp(a.tiles_x * a.tiles_y);
    hipLaunchKernelGGL((k_wino64_conv<CIN, COUT, POOL, TC>), dim3(8 * a.tiles_per_xcd * (COUT / WN2)), dim3(THREADS), 0, s, a);
    return hipGetLastError();
}

// upconv1[0] with the channel contraction of upconv1[2] folded into its epilogue: z planes instead of the 64-channel tensor.
template <int TC>
hipError_t launch_wino64_z_tc(hipStream_t s, const WinoArgs& base) {
    WinoArgs a = base;
    constexpr int TRW = 32 / TC;
    a.tiles_x = cdiv(a.Wc, 2 * TC); a.tiles_y = cdiv(a.Hc, 2 * TRW);
    a.tiles_total = a.N * a.tiles_x * a.tiles_y; a.tiles_per_xcd = cdiv(a.tiles_total, 8);
    a.rcp_x = tile_rcp(a.tiles_x); a.rcp_xy = tile_rcp(a.tiles_x * a.tiles_y);
    hipLaunchKernelGGL((k_wino64_conv<128, 64, false, TC, 0, true>), dim3(8 * a.tiles_per_xcd), dim3(THREADS), 0, s, a);
    return hipGetLastError();
}
int wino42_grid(WinoArgs& a, int nb);
extern int g_half_wg_per_cu;
int device_cus();
template <int TC>
hipError_t launch_wino42_z_tc(hipStream_t s, const WinoArgs& base, const float* blob, int tab) {
    WinoArgs a = base;
    constexpr int TRW = 16 / TC;
    a.slot_tab = reinterpret_cast<const unsigned*>(blob + kBlob.tab42_off[tab]);
    a.tiles_x = cdiv(a.Wc, 4 * TC); a.tiles_y = cdiv(a.Hc, 2 * TRW);
    a.tiles_total = a.N * a.tiles_x * a.tiles_y; a.tiles_per_xcd = cdiv(a.tiles_total, 8);
    a.rcp_x = tile_rcp(a.tiles_x); a.rcp_xy = tile_rcp(a.tiles_x * a.tiles_y);
    const int grid = wino42_grid(a, 1);
    hipLaunchKernelGGL((k_wino42_conv<128, 64, false, TC, 0, true>), dim3(grid), dim3(THREADS), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_upconv1_0_z(int algo, hipStream_t s, const float* blob, const float* in, int Hc, int Wc, float* zout, int N) {
    WinoArgs a;
    a.in = in; a.u = blob + kBlob.u_off[10]; a.bias = blob + kBlob.b_off[10];
    a.out = nullptr; a.pool = nullptr; a.zw = blob + kBlob.w_off[11]; a.zout = zout;
    a.N = N; a.Hin = Hc; a.Win = Wc; a.in_ps = 128; a.Hc = Hc; a.Wc = Wc; a.Hs = Hc; a.Ws = Wc;
    a.out_ps = 64; a.out_coff = 0;
    a.tiles_x = a.tiles_y = a.tiles_total = a.tiles_per_xcd = 0;
    a.rcp_x = a.rcp_xy = 0; a.walk = 0;
    if (algo == CID_ALGO_SPLIT16) {   // k_conv3x3_h16<128, 64, 0, ZOUT, ., F32IO>: `out` = the 27 fp32 z planes, `pool` = upconv1[2]'s hi | lo fragments
        GemmConvArgsH g;
        g.in = reinterpret_cast<const _Float16*>(in); g.w = reinterpret_cast<const _Float16*>(blob + kBlob.s_off[10]); g.bias = blob + kBlob.b_off[10];
        g.out = reinterpret_cast<_Float16*>(zout); g.pool = const_cast<_Float16*>(reinterpret_cast<const _Float16*>(blob + kBlob.hzs_off));
        g.N = N; g.Hin = Hc; g.Win = Wc; g.in_ps = 128; g.Hc = Hc; g.Wc = Wc; g.Hs = Hc; g.Ws = Wc; g.out_ps = 64; g.out_coff = 0;
        const TileGrid tg = tiles_for(N, Hc, Wc);
        g.tiles_x = tg.tx; g.tiles_y = tg.ty; g.tiles_total = tg.total; g.tiles_per_xcd = tg.per_xcd;
        g.rcp_x = tile_rcp(tg.tx); g.rcp_xy = tile_rcp(tg.tx * tg.ty); g.walk = 0;
        hipLaunchKernelGGL((k_conv3x3_h16<128, 64, 0, true, false, true>), dim3(8 * tg.per_xcd), dim3(THREADS), 0, s, g);
        return hipGetLastError();
    }
    if (algo == CID_ALGO_WINOGRAD42) {
        a.u = blob + kBlob.u42_off[10];
        if (Wc > 16) return launch_wino42_z_tc<8>(s, a, blob, 0);
        return launch_wino42_z_tc<4>(s, a, blob, 1);
    }
    a.slot_tab = reinterpret_cast<const unsigned*>(blob + kBlob.tab_off[Wc > 32 ? 0 : 1]);
    return Wc > 32 ? launch_wino64_z_tc<32>(s, a) : launch_wino64_z_tc<16>(s, a);
}
hipError_t launch_tail_z(hipStream_t s, const float* z, const float* bias, void* out, const Window& crop, int N, int H, int W, bool u8) {
    TailZArgs a;
    a.z = z; a.bias = bias; a.out = out; a.crop = crop; a.N = N; a.H = H; a.W = W;
    a.blocks_per_image = cdiv(H * W, THREADS);
    a.rcp_w = tile_rcp((unsigned)W); a.rcp_blocks = tile_rcp((unsigned)a.blocks_per_image);
    if (u8) hipLaunchKernelGGL((k_conv_tail_z<true>), dim3(N * a.blocks_per_image), dim3(THREADS), 0, s, a);
    else hipLaunchKernelGGL((k_conv_tail_z<false>), dim3(N * a.blocks_per_image), dim3(THREADS), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_tail_zh(hipStream_t s, const void* z, const float* bias, void* out, const Window& crop, int N, int H, int W, bool u8) {
    TailZArgs a;
    a.z = static_cast<const float*>(z); a.bias = bias; a.out = out; a.crop = crop; a.N = N; a.H = H; a.W = W;
    a.blocks_per_image = cdiv(H * W, THREADS);
    a.rcp_w = tile_rcp((unsigned)W); a.rcp_blocks = tile_rcp((unsigned)a.blocks_per_image);
    if (u8) hipLaunchKernelGGL((k_conv_tail_zh<true>), dim3(N * a.blocks_per_image), dim3(THREADS), 0, s, a);
    else hipLaunchKernelGGL((k_conv_tail_zh<false>), dim3(N * a.blocks_per_image), dim3(THREADS), 0, s, a);
    return hipGetLastError();
}

// Grid of a k_wino42_conv launch over `tiles_per_xcd` tiles per XCD group and NB column blocks.  Small launches: one workgroup
// per (tile, column block), a.walk = 0.  Once there are more items than the chip holds at a time — two workgroups per CU (LDS
// 75 KiB, <= 256 VGPRs) — the grid is what is resident and the workgroups WALK: a.walk = grid / 8 walkers per XCD group, each
// taking tiles local, local + walk, ... with all NB column blocks of a tile back to back.  CID_WINO42_WG_PER_CU (environment) and
// cid_debug_winograd_workgroups_per_cu (development / testing aids): 0 = never walk (the round-2 behaviour), k = k workgroups per CU.
// The environment values are clamped like the debug setters' (0 = never walk ... the occupancy the kernel's LDS use admits).
int env_wg_per_cu(const char* name, int dflt, int hi) {
    const char* e = std::getenv(name);
    if (!e) return dflt;
    const int v = std::atoi(e);
    return v < 0 ? 0 : v > hi ? hi : v;
}
int g_wino42_wg_per_cu = env_wg_per_cu("CID_WINO42_WG_PER_CU", 2, 2);
// k_conv3x3_h16, same meaning.  Default 0 since round 4: with the epilogue storing straight from the accumulators (no LDS staging, no barrier in front
// of it) one workgroup per item is FASTER than three walkers per CU on every layer that walked — same box, three boxes: forward +2.3...+2.8 %, down1.2 -7...-8 %,
// upconv1.0 -5.5...-6.7 %, upconv2.2 -2.6...-3.2 %, down2.2 -1.5...-2.6 % (profiles/r04_ab_f16_walk_vs_not.txt) — where round 3's kernels had gained 2.4 % from walking.
int g_half_wg_per_cu = env_wg_per_cu("CID_HALF_WG_PER_CU", 0, 3);
// CID_WINO42_XNB (environment, measurement aid): bit mask over the column-block counts NB (2, 4) whose walking launches give every XCD group ONE
// column block (a.walk < 0, wino42_kernels.h) instead of walking all NB blocks of a tile back to back.  Default 0: profiles/r04_xnb_experiment.txt.
int g_wino42_xnb = env_wg_per_cu("CID_WINO42_XNB", 0, 6);
// CU count of the CURRENT device (the one the launch goes to), cached per device id: a process may drive unlike devices.
int device_cus() {
    constexpr int MAXDEV = 64;
    static std::atomic<int> cache[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return 256;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    cache[dev].store(n, std::memory_order_relaxed);
    return n;
}
int wino42_grid(WinoArgs& a, int nb) {
    const int per_cu = g_wino42_wg_per_cu;
    const int cus = device_cus();
    const int items = 8 * a.tiles_per_xcd * nb;
    const int walkers = per_cu * cus / 8;                    // per XCD group
    a.walk = 0;
    // Walkers are indexed by TILE (a walker runs all nb column blocks of its tiles back to back), so walking needs at least one
    // tile per walker: with fewer, only tiles_per_xcd of the slots would work, each nb items deep, where one item per workgroup
    // spreads the same items over every CU (mid-size batches, N = 17..48 on the bottleneck layers).
    if (per_cu <= 0 || walkers < 1 || items <= 8 * walkers || a.tiles_per_xcd < walkers) return items;   // same-box against round 3's rule: profiles/r04_ab_midbatch_walk_rule.txt
    a.walk = walkers;
    if ((nb == 2 || nb == 4) && (g_wino42_xnb & nb)) {        // one column block per XCD group: 8 / nb tile ranges
        a.tiles_per_xcd = cdiv(a.tiles_total, 8 / nb);
        a.walk = -walkers;
    }
    return 8 * walkers;
}

template <int CIN, int COUT, bool POOL, int TC>
hipError_t launch_wino42_tc(hipStream_t s, const WinoArgs& base, const float* blob, int tab) {
    WinoArgs a = base;
    constexpr int TRW = 16 / TC;
    a.slot_tab = reinterpret_cast<const unsigned*>(blob + kBlob.tab42_off[tab]);
    a.tiles_x = cdiv(a.Wc, 4 * TC); a.tiles_y = cdiv(a.Hc, 2 * TRW);
    a.tiles_total = a.N * a.tiles_x * a.tiles_y; a.tiles_per_xcd = cdiv(a.tiles_total, 8);
    a.rcp_x = tile_rcp(a.tiles_x); a.rcp_xy = tile_rcp(a.tiles_x * a.tiles_y);
    const int grid = wino42_grid(a, COUT / WN2);
    hipLaunchKernelGGL((k_wino42_conv<CIN, COUT, POOL, TC>), dim3(grid), dim3(THREADS), 0, s, a);
    return hipGetLastError();
}

// One 3x3 GEMM layer, by the handle's algorithm: MODE 0/1 of k_gemm_conv or Winograd.
template <int CIN, int COUT, int MODE>
hipError_t launch_conv3x3(int algo, hipStream_t s, const float* blob, int layer, const float* in, int Hin, int Win, int in_ps,
                          float* out, int out_ps, int out_coff, int Hc, int Wc, int Hs, int Ws, float* pool, int N) {
    if (algo == 0) return launch_gemm<CIN, COUT, MODE>(s, blob, layer, in, Hin, Win, in_ps, out, out_ps, out_coff, Hc, Wc, Hs, Ws, pool, N);
    if (algo == CID_ALGO_SPLIT16) {
        // split-operand convolution on the fp16 MFMA, fp32 tensors in and out (conv_kernels_f16.h, F32IO): one (8x32-pixel tile, 64-channel column block) per workgroup, two per CU
        if (in_ps != CIN) return hipErrorInvalidValue;   // the kernel takes the pixel stride of its input as CIN (true of every layer of this network)
        GemmConvArgsH a;
        a.in = reinterpret_cast<const _Float16*>(in); a.w = reinterpret_cast<const _Float16*>(blob + kBlob.s_off[layer]); a.bias = blob + kBlob.b_off[layer];
        a.out = reinterpret_cast<_Float16*>(out); a.pool = reinterpret_cast<_Float16*>(pool);
        a.N = N; a.Hin = Hin; a.Win = Win; a.in_ps = in_ps; a.Hc = Hc; a.Wc = Wc; a.Hs = Hs; a.Ws = Ws; a.out_ps = out_ps; a.out_coff = out_coff;
        const TileGrid g = tiles_for(N, Hc, Wc);
        a.tiles_x = g.tx; a.tiles_y = g.ty; a.tiles_total = g.total; a.tiles_per_xcd = g.per_xcd;
        a.rcp_x = tile_rcp(g.tx); a.rcp_xy = tile_rcp(g.tx * g.ty); a.walk = 0;
        if constexpr (COUT >= 128) {   // two column blocks per workgroup from one staging of the input tile (conv_kernels_f16.h, PAIR)
            static const int pair = env_wg_per_cu("CID_SPLIT_PAIR", 1, 1);   // measurement aid: 0 = one column block per workgroup (same box: the six launches +8 % slower, profiles/r04_ab_split16_pair.txt)
            if (pair) {
                hipLaunchKernelGGL((k_conv3x3_h16<CIN, COUT, MODE, false, false, true, true>), dim3(8 * g.per_xcd * (COUT / NTILE / 2)), dim3(THREADS), 0, s, a);
                return hipGetLastError();
            }
        }
        hipLaunchKernelGGL((k_conv3x3_h16<CIN, COUT, MODE, false, false, true>), dim3(8 * g.per_xcd * (COUT / NTILE)), dim3(THREADS), 0, s, a);
        return hipGetLastError();
    }
    WinoArgs a;
    a.in = in; a.u = blob + kBlob.u_off[layer]; a.bias = blob + kBlob.b_off[layer]; a.slot_tab = nullptr;
    a.out = out; a.pool = pool; a.zw = nullptr; a.zout = nullptr;
    a.N = N; a.Hin = Hin; a.Win = Win; a.in_ps = in_ps; a.Hc = Hc; a.Wc = Wc; a.Hs = Hs; a.Ws = Ws;
    a.out_ps = out_ps; a.out_coff = out_coff;
    a.tiles_x = a.tiles_y = a.tiles_total = a.tiles_per_xcd = 0;
    a.rcp_x = a.rcp_xy = 0; a.walk = 0;
    if (algo == CID_ALGO_WINOGRAD42) {   // 16 tiles of 4x2 pixels per workgroup: 8 x 2 (32x4 pixels), or 4 x 4 (16x8) for rows of 16 pixels or fewer.
        // 16 x 1 (64x2 pixels) fetches 4 input rows for 2 of output: same-box 24.4k images/s against 24.8k (8 x 2) and 24.7k (4 x 4 everywhere)
        a.u = blob + kBlob.u42_off[layer];
        if (Wc > 16) return launch_wino42_tc<CIN, COUT, MODE == 1, 8>(s, a, blob, 0);
        return launch_wino42_tc<CIN, COUT, MODE == 1, 4>(s, a, blob, 1);
    }
    // 32 tile-columns (64 pixels) per workgroup when the rows are wide enough, else 16 x 2 tile-rows
    a.slot_tab = reinterpret_cast<const unsigned*>(blob + kBlob.tab_off[Wc > 32 ? 0 : 1]);
    return Wc > 32 ? launch_wino64_tc<CIN, COUT, MODE == 1, 32>(s, a) : launch_wino64_tc<CIN, COUT, MODE == 1, 16>(s, a);
}

template <int CIN, int COUT, int MODE, bool ZOUT = false>
hipError_t launch_gemm_h(hipStream_t s, const float* blob, int layer, const void* in, int Hin, int Win, int in_ps,
                         void* out, int out_ps, int out_coff, int Hc, int Wc, int Hs, int Ws, void* pool, int N);

// One GEMM-shaped layer under the handle's storage type and algorithm.  The arena regions are sized for fp32; the
// fp16-storage path keeps its half tensors in the front half of the same regions.
template <int CIN, int COUT, int MODE>
hipError_t launch_layer(cid_handle_t h, hipStream_t s, const float* blob, int layer, float* in, int Hin, int Win, int in_ps,
                        float* out, int out_ps, int out_coff, int Hc, int Wc, int Hs, int Ws, float* pool, int N) {
    if (h->dtype == CID_DTYPE_F16)
        return launch_gemm_h<CIN, COUT, MODE>(s, blob, layer, in, Hin, Win, in_ps, out, out_ps, out_coff, Hc, Wc, Hs, Ws, pool, N);
    if constexpr (MODE == 2) {
        static const int split_t = env_wg_per_cu("CID_SPLIT_CONVT", 1, 1);   // measurement aid: 0 = the transposed convolutions stay on the fp32-MFMA kernels under conv_algo "split16"
        if (h->algo == CID_ALGO_SPLIT16 && split_t) {
            // ConvTranspose2d(k=2, s=2) in the split-operand form: (tap, 64 channels) column blocks, two per workgroup, over 8x32-pixel tiles of the INPUT
            if (in_ps != CIN) return hipErrorInvalidValue;
            GemmConvArgsH a;
            a.in = reinterpret_cast<const _Float16*>(in); a.w = reinterpret_cast<const _Float16*>(blob + kBlob.s_off[layer]); a.bias = blob + kBlob.b_off[layer];
            a.out = reinterpret_cast<_Float16*>(out); a.pool = nullptr;
            a.N = N; a.Hin = Hin; a.Win = Win; a.in_ps = in_ps; a.Hc = Hc; a.Wc = Wc; a.Hs = Hs; a.Ws = Ws; a.out_ps = out_ps; a.out_coff = out_coff;
            const TileGrid g = tiles_for(N, Hc, Wc);
            a.tiles_x = g.tx; a.tiles_y = g.ty; a.tiles_total = g.total; a.tiles_per_xcd = g.per_xcd;
            a.rcp_x = tile_rcp(g.tx); a.rcp_xy = tile_rcp(g.tx * g.ty); a.walk = 0;
            hipLaunchKernelGGL((k_conv3x3_h16<CIN, COUT, 2, false, false, true, true>), dim3(8 * g.per_xcd * (4 * COUT / NTILE / 2)), dim3(THREADS), 0, s, a);
            return hipGetLastError();
        }
    }
    if constexpr (MODE == 2 && CIN == 128) {   // up1: streaming form, persistent workgroups (two per CU) over runs of TP input pixels
        using CG = ConvTGeom32<CIN, COUT>;
        GemmConvArgs a;
        a.in = in; a.w = blob + kBlob.w_off[layer]; a.bias = blob + kBlob.b_off[layer];
        a.out = out; a.pool = nullptr;
        a.N = N; a.Hin = Hin; a.Win = Win; a.in_ps = in_ps;
        a.Hc = Hc; a.Wc = Wc; a.Hs = Hs; a.Ws = Ws; a.out_ps = out_ps; a.out_coff = out_coff;
        a.tiles_x = (Hin * Win + CG::TP - 1) / CG::TP; a.tiles_y = 1; a.tiles_total = N * a.tiles_x; a.tiles_per_xcd = 0;
        a.rcp_x = tile_rcp(a.tiles_x); a.rcp_xy = tile_rcp(Win);
        const int wgs = device_cus() * 2;
        hipLaunchKernelGGL((k_convt_s32<CIN, COUT>), dim3(a.tiles_total < wgs ? a.tiles_total : wgs), dim3(THREADS), 0, s, a);
        return hipGetLastError();
    } else if constexpr (MODE == 2)
        return launch_gemm<CIN, COUT, MODE>(s, blob, layer, in, Hin, Win, in_ps, out, out_ps, out_coff, Hc, Wc, Hs, Ws, pool, N);
    else
        return launch_conv3x3<CIN, COUT, MODE>(h->algo, s, blob, layer, in, Hin, Win, in_ps, out, out_ps, out_coff, Hc, Wc, Hs, Ws, pool, N);
}

hipError_t launch_head(hipStream_t s, const HeadArgs& a, int grid, bool u8, bool f16) {
    if (u8 && f16) hipLaunchKernelGGL((k_conv_head_h16<true>), dim3(grid), dim3(THREADS), 0, s, a);
    else if (u8) hipLaunchKernelGGL((k_conv_head<true>), dim3(grid), dim3(THREADS), 0, s, a);
    else if (f16) hipLaunchKernelGGL((k_conv_head_h16<false>), dim3(grid), dim3(THREADS), 0, s, a);
    else hipLaunchKernelGGL((k_conv_head<false>), dim3(grid), dim3(THREADS), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_tail(hipStream_t s, const TailArgs& a, int grid, bool u8, bool f16) {
    if (u8 && f16) hipLaunchKernelGGL((k_conv_tail_h<true>), dim3(grid), dim3(THREADS), 0, s, a);
    else if (u8) hipLaunchKernelGGL((k_conv_tail<true, false>), dim3(grid), dim3(THREADS), 0, s, a);
    else if (f16) hipLaunchKernelGGL((k_conv_tail_h<false>), dim3(grid), dim3(THREADS), 0, s, a);
    else hipLaunchKernelGGL((k_conv_tail<false, false>), dim3(grid), dim3(THREADS), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_tail2(hipStream_t s, const Tail2Args& a, bool u8) {
    if (u8) hipLaunchKernelGGL((k_conv_tail2<true>), dim3(a.groups_total), dim3(THREADS), 0, s, a);
    else hipLaunchKernelGGL((k_conv_tail2<false>), dim3(a.groups_total), dim3(THREADS), 0, s, a);
    return hipGetLastError();
}

template <int CIN, int COUT, int MODE, bool ZOUT>
hipError_t launch_gemm_h(hipStream_t s, const float* blob, int layer, const void* in, int Hin, int Win, int in_ps,
                         void* out, int out_ps, int out_coff, int Hc, int Wc, int Hs, int Ws, void* pool, int N) {
    GemmConvArgsH a;
    a.in = static_cast<const _Float16*>(in); a.w = reinterpret_cast<const _Float16*>(blob + kBlob.h_off[layer]);
    a.bias = blob + kBlob.b_off[layer];
    a.out = static_cast<_Float16*>(out); a.pool = static_cast<_Float16*>(pool);
    if (ZOUT) a.pool = const_cast<_Float16*>(reinterpret_cast<const _Float16*>(blob + kBlob.hz_off));   // ZOUT: `out` = the z planes, `pool` = the last layer's weights (read only)
    a.N = N; a.Hin = Hin; a.Win = Win; a.in_ps = in_ps;
    a.Hc = Hc; a.Wc = Wc; a.Hs = Hs; a.Ws = Ws; a.out_ps = out_ps; a.out_coff = out_coff;
    const TileGrid g = tiles_for(N, Hc, Wc);
    a.tiles_x = g.tx; a.tiles_y = g.ty; a.tiles_total = g.total; a.tiles_per_xcd = g.per_xcd;
        a.rcp_x = tile_rcp(g.tx); a.rcp_xy = tile_rcp(g.tx * g.ty);
    constexpr int NB = (MODE == 2 ? 4 * COUT : COUT) / NTILE;
    a.walk = 0;
    if constexpr (MODE == 2 && !ZOUT) {
        // streaming form: persistent workgroups (two waves per SIMD: 2 per CU of four waves, 1 of eight) over runs of TP input pixels
        using CG = ConvTGeom<CIN, COUT>;
        a.tiles_x = (Hin * Win + CG::TP - 1) / CG::TP; a.tiles_total = N * a.tiles_x;
        a.rcp_x = tile_rcp(a.tiles_x); a.rcp_xy = tile_rcp(Win);
        const int wgs = device_cus() * 2 / CG::CB;
        hipLaunchKernelGGL((k_convt_t16<CIN, COUT>), dim3(a.tiles_total < wgs ? a.tiles_total : wgs), dim3(64 * CG::NW), 0, s, a);
    } else if constexpr (MODE != 2) {
        if (in_ps != CIN) return hipErrorInvalidValue;   // k_conv3x3_h16 takes the pixel stride of its input as CIN (true of every layer of this network)
        // walking workgroups (conv_kernels_f16.h): three per CU (47 KiB of LDS, <= 168 VGPRs) once there are more items than that —
        // on the layers with CIN <= 128, where an item is short beside its prologue (same-box: down1.2 -15 %, down2.0 -11 %, the
        // CIN = 128 layers -0.3...-1.5 %).  With CIN = 256 walking LOSES 3-5 %: a tile's NB column blocks then run one after the
        // other in one workgroup and the 174 KB halo tile of the second pass has left the XCD's L2 (96 walkers x 174 KB), while
        // sibling workgroups dispatched back to back share one fetch (profiles/r03_ab_f16_walk.txt).
        int grid = 8 * g.per_xcd * NB;
        const int walkers = g_half_wg_per_cu * device_cus() / 8;
        if (CIN <= 128 && g_half_wg_per_cu > 0 && walkers >= 1 && grid > 8 * walkers && g.per_xcd >= walkers) { a.walk = walkers; grid = 8 * walkers; }   // >= one tile per walker, as in wino42_grid
        // the pooling launches (MODE 1) never walk: their walking variants were the last kernels of the library with spilled registers (4 VGPRs, 20 B of
        // scratch per lane) and walking is slower than one item per workgroup on them by the widest margin (-7...-10 % on down1.2)
        if constexpr (MODE == 0) {
            if (a.walk) { hipLaunchKernelGGL((k_conv3x3_h16<CIN, COUT, MODE, ZOUT, true>), dim3(grid), dim3(THREADS), 0, s, a); return hipGetLastError(); }
        }
        a.walk = 0; grid = 8 * g.per_xcd * NB;
        hipLaunchKernelGGL((k_conv3x3_h16<CIN, COUT, MODE, ZOUT, false>), dim3(grid), dim3(THREADS), 0, s, a);
    }
    return hipGetLastError();
}

// (H, W) = the network input.  `src` places the caller's image inside it (the band around it is uint8 0 = -1.0 normalised), `crop`
// is the window of the network output the caller's tensor receives; null = identity (no padding, whole output).
int run_forward(cid_handle_t h, const void* in, int in_fmt, void* out, int out_fmt, int N, int H, int W, void* ws, size_t ws_bytes,
                hipStream_t s, hipEvent_t* ev /* NL+1 events or null */, const Window* src_win = nullptr, const Window* crop_win = nullptr) {
    if (!h) return CID_ERR_INVALID;
    if (!in || !out || !ws) return fail(h, CID_ERR_INVALID, "cid_forward: null pointer");
    if (!h->dev_blob) return fail(h, CID_ERR_STATE, "cid_forward: no device weights attached (call cid_upload_weights or cid_attach_weights)");
    Dims d;
    if (const char* why = shape_error(N, H, W, d)) {
        char m[256];
        std::snprintf(m, sizeof m, "cid_forward: input [%d,3,%d,%d] not accepted: %s", N, H, W, why);
        return fail(h, CID_ERR_SHAPE, m);
    }
    const Plan p = make_plan(d);
    if (ws_bytes < p.total_bytes) return fail(h, CID_ERR_WORKSPACE, "cid_forward: workspace smaller than cid_workspace_bytes()");
    if ((in_fmt != CID_FMT_F32_NCHW && in_fmt != CID_FMT_U8_NHWC) || (out_fmt != CID_FMT_F32_NCHW && out_fmt != CID_FMT_U8_NHWC))
        return fail(h, CID_ERR_INVALID, "cid_forward: unknown tensor format");
    if (((uintptr_t)ws & 255) || ((uintptr_t)h->dev_blob & 255) || (in_fmt == CID_FMT_F32_NCHW && ((uintptr_t)in & 3)))
        return fail(h, CID_ERR_WORKSPACE, "cid_forward: workspace/weights must be 256-byte aligned, fp32 input 4-byte aligned");
    float* base = static_cast<float*>(ws);
    float* B[NBUF];
    for (int b = 0; b < NBUF; ++b) B[b] = base + p.off[b];
    const float* blob = h->dev_blob;
    hipError_t e = hipSuccess;
    int li = 0;
#define STEP(call)                                                                         \
    do {                                                                                   \
        if (ev && hipEventRecord(ev[li], s) != hipSuccess) e = hipErrorUnknown;            \
        if (e == hipSuccess) e = (call);                                                   \
        if (e != hipSuccess) {                                                             \
            return fail(h, CID_ERR_HIP, std::string("launch '") + kLayers[li].name + "' failed: " + hipGetErrorString(e)); \
        }                                                                                  \
        ++li;                                                                              \
    } while (0)

    {   // down1[0]: Conv 3->64 + ReLU, NCHW in -> NHWC t0            app.py:43-44
        HeadArgs a;
        a.in = in; a.w = blob + (h->dtype == CID_DTYPE_F16 ? kBlob.h_off[0] : kBlob.w_off[0]); a.bias = blob + kBlob.b_off[0]; a.out = B[T0];
        a.N = N; a.H = H; a.W = W;
        a.src = src_win ? *src_win : Window{0, 0, H, W};
        const TileGrid g = tiles_for(N, H, W);
        a.tiles_x = g.tx; a.tiles_y = g.ty; a.tiles_total = g.total;
        tile_groups(a);
        a.rcp_x = tile_rcp(g.tx); a.rcp_xy = tile_rcp(g.tx * g.ty);
        STEP(launch_head(s, a, 8 * a.groups_per_xcd, in_fmt == CID_FMT_U8_NHWC, h->dtype == CID_DTYPE_F16));
    }
    // down1[2] + ReLU -> e1 into cat1[:, 64:128] (cropped to Hu1 x Wu1), pool1 -> p1     app.py:45-48,97-100
    STEP((launch_layer<64, 64, 1>(h, s, blob, 1, B[T0], H, W, 64, B[CAT1], 128, 64, 2 * d.H1, 2 * d.W1, d.Hu1, d.Wu1, B[P1], N)));
    // down2[0] + ReLU                                                                    app.py:51-52
    STEP((launch_layer<64, 128, 0>(h, s, blob, 2, B[P1], d.H1, d.W1, 64, B[T1], 128, 0, d.H1, d.W1, d.H1, d.W1, nullptr, N)));
    // down2[2] + ReLU -> e2 into cat2[:, 128:256] (cropped), pool2 -> p2                 app.py:53-56,90-93
    STEP((launch_layer<128, 128, 1>(h, s, blob, 3, B[T1], d.H1, d.W1, 128, B[CAT2], 256, 128, d.Hu2, d.Wu2, d.Hu2, d.Wu2, B[P2], N)));
    // bottleneck                                                                         app.py:59-62
    STEP((launch_layer<128, 256, 0>(h, s, blob, 4, B[P2], d.H2, d.W2, 128, B[T2], 256, 0, d.H2, d.W2, d.H2, d.W2, nullptr, N)));
    STEP((launch_layer<256, 256, 0>(h, s, blob, 5, B[T2], d.H2, d.W2, 256, B[BT], 256, 0, d.H2, d.W2, d.H2, d.W2, nullptr, N)));
    // up2: ConvT 256->128 -> cat2[:, 0:128]                                              app.py:65,89
    STEP((launch_layer<256, 128, 2>(h, s, blob, 6, B[BT], d.H2, d.W2, 256, B[CAT2], 256, 0, d.H2, d.W2, d.H2, d.W2, nullptr, N)));
    // upconv2                                                                            app.py:67-70
    STEP((launch_layer<256, 128, 0>(h, s, blob, 7, B[CAT2], d.Hu2, d.Wu2, 256, B[T3], 128, 0, d.Hu2, d.Wu2, d.Hu2, d.Wu2, nullptr, N)));
    STEP((launch_layer<128, 128, 0>(h, s, blob, 8, B[T3], d.Hu2, d.Wu2, 128, B[D2], 128, 0, d.Hu2, d.Wu2, d.Hu2, d.Wu2, nullptr, N)));
    // up1: ConvT 128->64 -> cat1[:, 0:64]                                                app.py:73,96
    STEP((launch_layer<128, 64, 2>(h, s, blob, 9, B[D2], d.Hu2, d.Wu2, 128, B[CAT1], 128, 0, d.Hu2, d.Wu2, d.Hu2, d.Wu2, nullptr, N)));
    const bool fused_tail = fused_tail_active(h);
    const Window crop = crop_win ? *crop_win : Window{0, 0, d.Hu1, d.Wu1};
    if (fused_tail && h->dtype == CID_DTYPE_F16) {
        // the same on the fp16-storage path: z[N][9][Hu1][Wu1][4] halfs into the t4 region                         app.py:75-77
        STEP((launch_gemm_h<128, 64, 0, true>(s, blob, 10, B[CAT1], d.Hu1, d.Wu1, 128, B[T4], 64, 0, d.Hu1, d.Wu1, d.Hu1, d.Wu1, nullptr, N)));
        STEP(launch_tail_zh(s, B[T4], blob + kBlob.b_off[11], out, crop, N, d.Hu1, d.Wu1, out_fmt == CID_FMT_U8_NHWC));
    } else if (fused_tail) {
        // upconv1[0] + ReLU, with upconv1[2]'s channel contraction in its epilogue: z planes into the t4 region     app.py:75-77
        STEP(launch_upconv1_0_z(h->algo, s, blob, B[CAT1], d.Hu1, d.Wu1, B[T4], N));
        // the nine-tap shifted sum + bias + tanh, -> NCHW out                              app.py:77,103
        STEP(launch_tail_z(s, B[T4], blob + kBlob.b_off[11], out, crop, N, d.Hu1, d.Wu1, out_fmt == CID_FMT_U8_NHWC));
    } else {
    // upconv1[0] + ReLU                                                                  app.py:75-76
    STEP((launch_layer<128, 64, 0>(h, s, blob, 10, B[CAT1], d.Hu1, d.Wu1, 128, B[T4], 64, 0, d.Hu1, d.Wu1, d.Hu1, d.Wu1, nullptr, N)));
    // upconv1[2] + tanh, NHWC t4 -> NCHW out                                            app.py:77,103
    if (h->dtype == CID_DTYPE_F32 && d.Wu1 <= T2_MAXW && h->tail_algo == CID_TAIL_BANDS) {   // row-band kernel: z once per pixel
        Tail2Args a;
        a.in = B[T4]; a.w = blob + kBlob.w_off[11]; a.bias = blob + kBlob.b_off[11]; a.out = out; a.crop = crop;
        a.N = N; a.H = d.Hu1; a.W = d.Wu1;
        tail2_plan(a);
        STEP(launch_tail2(s, a, out_fmt == CID_FMT_U8_NHWC));
    } else {   // 8x32 tiles with halo: images wider than 128 pixels, and the fp16-storage path
        TailArgs a;
        a.in = B[T4]; a.w = blob + (h->dtype == CID_DTYPE_F16 ? kBlob.h_off[11] : kBlob.w_off[11]); a.bias = blob + kBlob.b_off[11]; a.out = out; a.crop = crop;
        a.N = N; a.H = d.Hu1; a.W = d.Wu1;
        const TileGrid g = tiles_for(N, d.Hu1, d.Wu1);
        a.tiles_x = g.tx; a.tiles_y = g.ty; a.tiles_total = g.total;
        tile_groups(a);
        a.rcp_x = tile_rcp(g.tx); a.rcp_xy = tile_rcp(g.tx * g.ty);
        STEP(launch_tail(s, a, 8 * a.groups_per_xcd, out_fmt == CID_FMT_U8_NHWC, h->dtype == CID_DTYPE_F16));
    }
    }
#undef STEP
    if (ev && hipEventRecord(ev[NL], s) != hipSuccess) return fail(h, CID_ERR_HIP, "hipEventRecord failed");
    return CID_OK;
}

}  // namespace

extern "C" {

const char* cid_version(void) { return "cid 0.4.0 (gfx950, fp32 MFMA: Winograd F(4x2,3x3) / F(2x2,3x3) + implicit GEMM; fp16-storage path: 16x16x32 MFMA)"; }

int cid_create(cid_handle_t* out) {
    if (!out) return CID_ERR_INVALID;
    *out = new (std::nothrow) cid_handle_s();
    return *out ? CID_OK : CID_ERR_INVALID;
}

void cid_destroy(cid_handle_t h) {
    if (h) for (hipEvent_t e : h->tev) (void)hipEventDestroy(e);
    delete h;
}

const char* cid_last_error(cid_handle_t h) { return h ? h->err.c_str() : "null handle"; }

const char* cid_param_key(int i) {
    static std::string keys[CID_NUM_PARAMS];
    if (i < 0 || i >= CID_NUM_PARAMS) return nullptr;
    if (keys[i].empty()) keys[i] = std::string(kLayers[i / 2].name) + ((i & 1) ? ".bias" : ".weight");
    return keys[i].c_str();
}

int cid_set_weight(cid_handle_t h, const char* key, const float* data, const int64_t* shape, int ndim) {
    if (!h) return CID_ERR_INVALID;
    if (!key || !data || !shape) return fail(h, CID_ERR_INVALID, "cid_set_weight: null argument");
    int l, is_bias;
    if (!find_key(key, l, is_bias)) return fail(h, CID_ERR_KEY, std::string("unexpected key '") + key + "' in state_dict");
    const LayerDef& L = kLayers[l];
    int64_t want[4]; int wn;
    if (is_bias) { want[0] = L.cout; wn = 1; }
    else if (L.kind == CONVT) { want[0] = L.cin; want[1] = L.cout; want[2] = 2; want[3] = 2; wn = 4; }
    else { want[0] = L.cout; want[1] = L.cin; want[2] = 3; want[3] = 3; wn = 4; }
    bool ok = ndim == wn;
    for (int i = 0; ok && i < wn; ++i) ok = shape[i] == want[i];
    if (!ok) {
        std::string m = std::string("size mismatch for ") + key + ": expected [";
        for (int i = 0; i < wn; ++i) m += (i ? "," : "") + std::to_string(want[i]);
        m += "], got [";
        for (int i = 0; i < ndim; ++i) m += (i ? "," : "") + std::to_string(shape[i]);
        return fail(h, CID_ERR_SHAPE, m + "]");
    }
    if (is_bias) {
        std::memcpy(h->staging.data() + kBlob.b_off[l], data, sizeof(float) * L.cout);
        std::memcpy(h->staging.data() + kBlob.raw_b_off[l], data, sizeof(float) * L.cout);
    } else {
        float* dst = h->staging.data() + kBlob.w_off[l];
        for_each_weight(L, [&](int co, int ci, int kh, int kw) { dst[packed_index(L, co, ci, kh, kw)] = data[ref_index(L, co, ci, kh, kw)]; });
        std::memcpy(h->staging.data() + kBlob.raw_w_off[l], data, sizeof(float) * ref_weight_count(L));
        if (L.kind == CONV) pack_winograd_u(L, data, h->staging.data() + kBlob.u_off[l]);
        if (L.kind == CONV) pack_winograd42_u(L, data, h->staging.data() + kBlob.u42_off[l]);
        if (L.kind == TAIL) {   // half copy for k_conv_tail_h: [k-step s][lane = 32*h + col][e], ci = 16*s + 8*h + e, col = 3*tap + co
            _Float16* hd = reinterpret_cast<_Float16*>(h->staging.data() + kBlob.h_off[l]);
            for_each_weight(L, [&](int co, int ci, int kh, int kw) {
                const int col = (kh * 3 + kw) * 3 + co, s = ci >> 4, hh = (ci >> 3) & 1, e = ci & 7;
                hd[((size_t)s * 64 + hh * 32 + col) * 8 + e] = (_Float16)data[ref_index(L, co, ci, kh, kw)];
            });
            // and for the fused form (h16_zout_epilogue): the A operand of z^T = W2' . X^T on v_mfma_f32_16x16x32_f16, row 3 tap + co
            // (27 rows, rows 27..31 stay zero): [row tile t][k-step ks][lane = 16 kga + row][e] with ci = 32 ks + 8 kga + e
            _Float16* hz = reinterpret_cast<_Float16*>(h->staging.data() + kBlob.hz_off);
            std::memset(hz, 0, 3 * 2 * 64 * 8 * sizeof(_Float16));
            for_each_weight(L, [&](int co, int ci, int kh, int kw) {
                const int row = 3 * (kh * 3 + kw) + co, t = row >> 4, ks = ci >> 5, kga = (ci >> 3) & 3, e = ci & 7;
                hz[(((size_t)t * 2 + ks) * 64 + kga * 16 + (row & 15)) * 8 + e] = (_Float16)data[ref_index(L, co, ci, kh, kw)];
            });
            // the same fragments as hi | lo pieces for the split-operand form (h16_zout_epilogue_f32)
            _Float16* hzs = reinterpret_cast<_Float16*>(h->staging.data() + kBlob.hzs_off);
            std::memset(hzs, 0, 2 * 2 * 2 * 64 * 8 * sizeof(_Float16));
            for_each_weight(L, [&](int co, int ci, int kh, int kw) {
                const int row = 3 * (kh * 3 + kw) + co, t = row >> 4, ks = ci >> 5, kga = (ci >> 3) & 3, e = ci & 7;
                const float v = data[ref_index(L, co, ci, kh, kw)];
                const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                const size_t at = (((size_t)t * 2 + ks) * 64 + kga * 16 + (row & 15)) * 8 + e;
                hzs[at] = hi; hzs[4 * 64 * 8 + at] = lo;
            });
        }
        if (L.kind == CONVT) {   // split-operand pieces of a transposed convolution (k_conv3x3_h16<..., 2, ., ., F32IO, PAIR>): piece p = 0 hi_w, 1 lo_w, 2 hi_w again (met by lo_x)
            _Float16* sp = reinterpret_cast<_Float16*>(h->staging.data() + kBlob.s_off[l]);
            const int cbs = L.cout / 64, nchunk = L.cin / 32;
            for_each_weight(L, [&](int co, int ci, int kh, int kw) {
                const float v = data[ref_index(L, co, ci, kh, kw)];
                const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                const int blk = (kh * 2 + kw) * cbs + (co >> 6), cg = co & 3, c = (co >> 2) & 15, ck = ci >> 5, kg = (ci >> 3) & 3, e = ci & 7;
                for (int p = 0; p < 3; ++p)
                    sp[((((((size_t)blk * nchunk + ck) * 3 + p) * 4 + cg) * 64) + kg * 16 + c) * 8 + e] = p == 1 ? lo : hi;
            });
        }
        if (L.kind == CONV) {   // split-operand pieces (conv_algo = "split16")
            _Float16* sp = reinterpret_cast<_Float16*>(h->staging.data() + kBlob.s_off[l]);
            for_each_weight(L, [&](int co, int ci, int kh, int kw) {
                const float v = data[ref_index(L, co, ci, kh, kw)];
                const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                sp[packed_index_s16(L, co, ci, kh, kw)] = hi; sp[packed_index_s16(L, co, ci, kh, 3 + kw)] = lo; sp[packed_index_s16(L, co, ci, kh, 6 + kw)] = hi;
            });
        }
        if (L.kind == CONV || L.kind == CONVT) {
            _Float16* hd = reinterpret_cast<_Float16*>(h->staging.data() + kBlob.h_off[l]);
            for_each_weight(L, [&](int co, int ci, int kh, int kw) {
                hd[L.kind == CONV ? packed_index_h16(L, co, ci, kh, kw) : packed_index_ht(L, co, ci, kh, kw)] = (_Float16)data[ref_index(L, co, ci, kh, kw)];
            });
        }
        if (L.kind == HEAD) {   // k_conv_head_h16: B[k = 3 tap + c][co], rows 27..31 zero; lane (col = (co / 4) % 16, kg) of group co % 4 holds k = 8kg..8kg+7
            _Float16* hd = reinterpret_cast<_Float16*>(h->staging.data() + kBlob.h_off[l]);
            for (int co = 0; co < 64; ++co)
                for (int k = 0; k < 32; ++k) {
                    const int tap = k / 3, c = k % 3;
                    const float v = k < 27 ? data[ref_index(L, co, c, tap / 3, tap % 3)] : 0.f;
                    hd[((co & 3) * 64 + (k >> 3) * 16 + ((co >> 2) & 15)) * 8 + (k & 7)] = (_Float16)v;
                }
        }
    }
    h->have[l][is_bias] = true;
    return CID_OK;
}

int cid_get_weight(cid_handle_t h, const char* key, float* out, size_t count) {
    if (!h) return CID_ERR_INVALID;
    if (!key || !out) return fail(h, CID_ERR_INVALID, "cid_get_weight: null argument");
    int l, is_bias;
    if (!find_key(key, l, is_bias)) return fail(h, CID_ERR_KEY, std::string("unknown key '") + key + "'");
    const LayerDef& L = kLayers[l];
    const size_t need = is_bias ? (size_t)L.cout : ref_weight_count(L);
    if (count != need) return fail(h, CID_ERR_SHAPE, std::string("cid_get_weight: wrong element count for ") + key);
    // the blob carries a reference-layout copy of every tensor (the Winograd U is not invertible bit-for-bit)
    std::memcpy(out, h->staging.data() + (is_bias ? kBlob.raw_b_off[l] : kBlob.raw_w_off[l]), sizeof(float) * need);
    return CID_OK;
}

int cid_missing_weights(cid_handle_t h, int* missing) {
    if (!h || !missing) return CID_ERR_INVALID;
    int m = 0;
    for (int l = 0; l < NL; ++l) m += !h->have[l][0] + !h->have[l][1];
    *missing = m;
    return CID_OK;
}

size_t cid_packed_weights_bytes(void) { return kBlob.total * sizeof(float); }

int cid_upload_weights(cid_handle_t h, void* device_blob, void* stream) {
    if (!h) return CID_ERR_INVALID;
    if (!device_blob) return fail(h, CID_ERR_INVALID, "cid_upload_weights: null device pointer");
    if ((uintptr_t)device_blob & 255) return fail(h, CID_ERR_WORKSPACE, "cid_upload_weights: blob must be 256-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemcpyAsync(device_blob, h->staging.data(), kBlob.total * sizeof(float), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);   // staging is pageable host memory owned by the handle
    if (e != hipSuccess) return fail(h, CID_ERR_HIP, std::string("cid_upload_weights: ") + hipGetErrorString(e));
    h->dev_blob = static_cast<const float*>(device_blob);
    return CID_OK;
}

int cid_export_packed(cid_handle_t h, void* host_out, size_t bytes) {
    if (!h) return CID_ERR_INVALID;
    if (!host_out) return fail(h, CID_ERR_INVALID, "cid_export_packed: null pointer");
    if (bytes != kBlob.total * sizeof(float)) return fail(h, CID_ERR_SHAPE, "cid_export_packed: size != cid_packed_weights_bytes()");
    std::memcpy(host_out, h->staging.data(), bytes);
    return CID_OK;
}

int cid_import_packed(cid_handle_t h, const void* host_in, size_t bytes) {
    if (!h) return CID_ERR_INVALID;
    if (!host_in) return fail(h, CID_ERR_INVALID, "cid_import_packed: null pointer");
    if (bytes != kBlob.total * sizeof(float)) return fail(h, CID_ERR_SHAPE, "cid_import_packed: size != cid_packed_weights_bytes()");
    std::memcpy(h->staging.data(), host_in, bytes);
    std::memset(h->have, 1, sizeof(h->have));
    return CID_OK;
}

int cid_attach_weights(cid_handle_t h, const void* device_blob) {
    if (!h) return CID_ERR_INVALID;
    if (!device_blob) return fail(h, CID_ERR_INVALID, "cid_attach_weights: null device pointer");
    if ((uintptr_t)device_blob & 255) return fail(h, CID_ERR_WORKSPACE, "cid_attach_weights: blob must be 256-byte aligned");
    h->dev_blob = static_cast<const float*>(device_blob);
    return CID_OK;
}

int cid_out_shape(int H, int W, int* Ho, int* Wo) {
    if (!Ho || !Wo) return CID_ERR_INVALID;
    if (H < 4 || W < 4) return CID_ERR_SHAPE;
    *Ho = 4 * (H / 4);
    *Wo = 4 * (W / 4);
    return CID_OK;
}

int cid_workspace_bytes(int N, int H, int W, size_t* bytes) {
    if (!bytes) return CID_ERR_INVALID;
    Dims d;
    if (shape_error(N, H, W, d)) return CID_ERR_SHAPE;
    *bytes = make_plan(d).total_bytes;
    return CID_OK;
}

int cid_forward(cid_handle_t h, const float* in, float* out, int N, int H, int W, void* ws, size_t ws_bytes, void* stream) {
    return cid_forward_ex(h, in, CID_FMT_F32_NCHW, out, CID_FMT_F32_NCHW, N, H, W, ws, ws_bytes, stream);
}

int cid_forward_ex(cid_handle_t h, const void* in, int in_fmt, void* out, int out_fmt, int N, int H, int W,
                   void* ws, size_t ws_bytes, void* stream) {
    hipEvent_t* ev = nullptr;
    if (h && h->tev_used < h->tev_forwards) ev = h->tev.data() + (size_t)h->tev_used * (NL + 1);
    const int rc = run_forward(h, in, in_fmt, out, out_fmt, N, H, W, ws, ws_bytes, static_cast<hipStream_t>(stream), ev);
    if (ev && rc == CID_OK) ++h->tev_used;
    return rc;
}

int cid_forward_padded(cid_handle_t h, const void* in, int in_fmt, void* out, int out_fmt, int N, int H, int W,
                       int pad_left, int pad_top, int pad_right, int pad_bottom, void* ws, size_t ws_bytes, void* stream) {
    if (!h) return CID_ERR_INVALID;
    if (pad_left < 0 || pad_top < 0 || pad_right < 0 || pad_bottom < 0 || H < 1 || W < 1 || pad_left > 4096 || pad_top > 4096 || pad_right > 4096 || pad_bottom > 4096)
        return fail(h, CID_ERR_INVALID, "cid_forward_padded: paddings must lie in [0, 4096] and the image must not be empty");
    const long long Hp = (long long)H + pad_top + pad_bottom, Wp = (long long)W + pad_left + pad_right;
    if (Hp > 0x7fffffff || Wp > 0x7fffffff) return fail(h, CID_ERR_SHAPE, "cid_forward_padded: padded size overflows");
    int Ho = 0, Wo = 0;
    if (cid_out_shape((int)Hp, (int)Wp, &Ho, &Wo) != CID_OK) return fail(h, CID_ERR_SHAPE, "cid_forward_padded: padded image smaller than 4x4 (output size is too small)");
    // the reference crops [top, top + H) x [left, left + W) out of the network's output (app.py:474-480): that window must exist
    if (pad_top + H > Ho || pad_left + W > Wo) {
        char m[256];
        std::snprintf(m, sizeof m, "cid_forward_padded: the crop window [%d+%d, %d+%d] does not fit the network output %dx%d of the padded image %lldx%lld "
                                   "(pad to a multiple of 4, app.py:276-281)", pad_top, H, pad_left, W, Ho, Wo, Hp, Wp);
        return fail(h, CID_ERR_SHAPE, m);
    }
    const Window src{pad_top, pad_left, H, W}, crop{pad_top, pad_left, H, W};
    hipEvent_t* ev = nullptr;
    if (h->tev_used < h->tev_forwards) ev = h->tev.data() + (size_t)h->tev_used * (NL + 1);
    const int rc = run_forward(h, in, in_fmt, out, out_fmt, N, (int)Hp, (int)Wp, ws, ws_bytes, static_cast<hipStream_t>(stream), ev, &src, &crop);
    if (ev && rc == CID_OK) ++h->tev_used;
    return rc;
}

int cid_view_u8(const float* in_nchw, void* out_u8_nhwc, int N, int H, int W, void* stream) {
    if (!in_nchw || !out_u8_nhwc || N < 1 || H < 1 || W < 1) return CID_ERR_INVALID;
    const size_t plane = (size_t)H * W, pixels = plane * N;
    if ((pixels + THREADS - 1) / THREADS > 0x7fffffffull) return CID_ERR_SHAPE;
    hipLaunchKernelGGL(k_view_u8, dim3((unsigned)((pixels + THREADS - 1) / THREADS)), dim3(THREADS), 0, static_cast<hipStream_t>(stream),
                       in_nchw, static_cast<unsigned char*>(out_u8_nhwc), plane, pixels);
    return hipGetLastError() == hipSuccess ? CID_OK : CID_ERR_HIP;
}

int cid_debug_winograd_workgroups_per_cu(int k) {
    const int prev = g_wino42_wg_per_cu;
    if (k >= 0) g_wino42_wg_per_cu = k > 2 ? 2 : k;   // two is what the kernel's LDS use (75 KiB) admits
    return prev;
}
int cid_debug_winograd_column_block_per_xcd(int mask) {
    const int prev = g_wino42_xnb;
    if (mask >= 0) g_wino42_xnb = mask & 6;
    return prev;
}
int cid_debug_half_workgroups_per_cu(int k) {
    const int prev = g_half_wg_per_cu;
    if (k >= 0) g_half_wg_per_cu = k > 3 ? 3 : k;     // three: 47 KiB of LDS, 168 VGPRs
    return prev;
}

int cid_timing_begin(cid_handle_t h, int max_forwards) {
    if (!h) return CID_ERR_INVALID;
    if (max_forwards < 1 || max_forwards > 4096) return fail(h, CID_ERR_INVALID, "cid_timing_begin: max_forwards out of range");
    if (!h->tev.empty()) return fail(h, CID_ERR_STATE, "cid_timing_begin: timing already armed");
    h->tev.resize((size_t)max_forwards * (NL + 1));
    for (size_t i = 0; i < h->tev.size(); ++i)
        if (hipEventCreate(&h->tev[i]) != hipSuccess) {
            for (size_t j = 0; j < i; ++j) (void)hipEventDestroy(h->tev[j]);
            h->tev.clear();
            return fail(h, CID_ERR_HIP, "cid_timing_begin: hipEventCreate failed");
        }
    h->tev_forwards = max_forwards;
    h->tev_used = 0;
    return CID_OK;
}

int cid_timing_end(cid_handle_t h, void* stream, float* launch_ms_sum, int* forwards) {
    if (!h) return CID_ERR_INVALID;
    if (!launch_ms_sum || !forwards) return fail(h, CID_ERR_INVALID, "cid_timing_end: null pointer");
    if (h->tev.empty()) return fail(h, CID_ERR_STATE, "cid_timing_end: timing not armed");
    int rc = CID_OK;
    if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) rc = fail(h, CID_ERR_HIP, "cid_timing_end: stream sync failed");
    for (int l = 0; l < NL; ++l) launch_ms_sum[l] = 0.f;
    for (int f = 0; rc == CID_OK && f < h->tev_used; ++f)
        for (int l = 0; l < NL; ++l) {
            float ms = 0.f;
            const hipEvent_t* e = h->tev.data() + (size_t)f * (NL + 1);
            if (hipEventElapsedTime(&ms, e[l], e[l + 1]) != hipSuccess) { rc = fail(h, CID_ERR_HIP, "cid_timing_end: hipEventElapsedTime failed"); break; }
            launch_ms_sum[l] += ms;
        }
    *forwards = h->tev_used;
    for (hipEvent_t e : h->tev) (void)hipEventDestroy(e);
    h->tev.clear();
    h->tev_forwards = h->tev_used = 0;
    return rc;
}

int cid_forward_timed(cid_handle_t h, const float* in, float* out, int N, int H, int W, void* ws, size_t ws_bytes,
                      void* stream, float* launch_ms) {
    if (!h) return CID_ERR_INVALID;
    if (!launch_ms) return fail(h, CID_ERR_INVALID, "cid_forward_timed: null launch_ms");
    hipEvent_t ev[NL + 1];
    int made = 0;
    for (; made <= NL; ++made)
        if (hipEventCreate(&ev[made]) != hipSuccess) break;
    int rc = made == NL + 1 ? CID_OK : fail(h, CID_ERR_HIP, "hipEventCreate failed");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (rc == CID_OK) rc = run_forward(h, in, CID_FMT_F32_NCHW, out, CID_FMT_F32_NCHW, N, H, W, ws, ws_bytes, s, ev);
    if (rc == CID_OK && hipStreamSynchronize(s) != hipSuccess) rc = fail(h, CID_ERR_HIP, "cid_forward_timed: stream sync failed");
    if (rc == CID_OK)
        for (int l = 0; l < NL; ++l)
            if (hipEventElapsedTime(&launch_ms[l], ev[l], ev[l + 1]) != hipSuccess) rc = fail(h, CID_ERR_HIP, "hipEventElapsedTime failed");
    for (int i = 0; i < made; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}

const char* cid_launch_name(int i) { return (i >= 0 && i < NL) ? kLayers[i].name : nullptr; }
const char* cid_launch_kernel(cid_handle_t h, int i) {
    if (i < 0 || i >= NL) return nullptr;
    if (h && h->dtype == CID_DTYPE_F16) {
        if (i >= 10 && fused_tail_active(h)) return i == 10 ? "k_conv3x3_h16<128, 64, 0, true," : "k_conv_tail_zh<";
        return kHalfKernelNames[i];
    }
    if (h && h->algo == CID_ALGO_SPLIT16) {
        if (i >= 10 && fused_tail_active(h)) return i == 10 ? "k_conv3x3_h16<128, 64, 0, true, false, true," : "k_conv_tail_z<";
        return kSplitKernelNames[i] ? kSplitKernelNames[i] : kKernelNames[i];
    }
    if (h && h->algo == CID_ALGO_WINOGRAD42 && kWino42KernelNames[i]) return kWino42KernelNames[i];
    if (h && h->algo != CID_ALGO_DIRECT && kWino64KernelNames[i]) return kWino64KernelNames[i];
    return kKernelNames[i];
}

int cid_set_conv_algo(cid_handle_t h, int algo) {
    if (!h) return CID_ERR_INVALID;
    if (algo != CID_ALGO_DIRECT && algo != CID_ALGO_WINOGRAD64 && algo != CID_ALGO_WINOGRAD42 && algo != CID_ALGO_SPLIT16) return fail(h, CID_ERR_INVALID, "cid_set_conv_algo: unknown algorithm");
    h->algo = algo;
    return CID_OK;
}
int cid_set_tail_algo(cid_handle_t h, int algo) {
    if (!h) return CID_ERR_INVALID;
    if (algo != CID_TAIL_FUSED && algo != CID_TAIL_BANDS && algo != CID_TAIL_TILES) return fail(h, CID_ERR_INVALID, "cid_set_tail_algo: unknown algorithm");
    h->tail_algo = algo;
    return CID_OK;
}
int cid_get_tail_algo(cid_handle_t h, int* algo) {
    if (!h || !algo) return CID_ERR_INVALID;
    *algo = h->tail_algo;
    return CID_OK;
}
int cid_set_compute_dtype(cid_handle_t h, int dtype) {
    if (!h) return CID_ERR_INVALID;
    if (dtype != CID_DTYPE_F32 && dtype != CID_DTYPE_F16) return fail(h, CID_ERR_INVALID, "cid_set_compute_dtype: unknown dtype");
    h->dtype = dtype;
    return CID_OK;
}
int cid_get_compute_dtype(cid_handle_t h, int* dtype) {
    if (!h || !dtype) return CID_ERR_INVALID;
    *dtype = h->dtype;
    return CID_OK;
}
// Testing aid: leave NaN in every byte of every CU's LDS.  LDS is not cleared between kernels, so whatever the forward's
// kernels read from LDS without having written it is NaN afterwards (a zero weight does not hide that: 0 x NaN = NaN).
__global__ void __launch_bounds__(256) k_poison_lds() {
    __shared__ float lds[8192];   // 32 KiB: five of these blocks fit a CU and together cover its 160 KiB
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = __builtin_nanf("");
    __syncthreads();
    for (int sl = 0; sl < 8; ++sl) __builtin_amdgcn_s_sleep(127);   // stay resident while the CU's other slots fill up
    const float v = lds[threadIdx.x];
    asm volatile("" ::"v"(v));
}
int cid_debug_poison_lds(void* stream) {
    hipLaunchKernelGGL(k_poison_lds, dim3(256 * 5 * 4), dim3(256), 0, static_cast<hipStream_t>(stream));
    return hipGetLastError() == hipSuccess ? CID_OK : CID_ERR_HIP;
}

int cid_get_conv_algo(cid_handle_t h, int* algo) {
    if (!h || !algo) return CID_ERR_INVALID;
    *algo = h->algo;
    return CID_OK;
}

// ---- where the reference module's stage outputs live in the arena (tests: per-stage parity) ----
int cid_stage_view(const char* stage, int N, int H, int W, size_t* offset_bytes, int* C, int* Hs, int* Ws, int* pixel_stride, int* channel_offset) {
    if (!stage || !offset_bytes || !C || !Hs || !Ws || !pixel_stride || !channel_offset) return CID_ERR_INVALID;
    Dims d;
    if (shape_error(N, H, W, d)) return CID_ERR_SHAPE;
    const Plan p = make_plan(d);
    struct Row { const char* name; Buf buf; int c, hs, ws, ps, coff; };
    const Row rows[] = {
        {"down1", CAT1, 64, d.Hu1, d.Wu1, 128, 64},      // e1, top-left crop Hu1 x Wu1 of H x W       app.py:81,97-100
        {"pool1", P1, 64, d.H1, d.W1, 64, 0},            //                                            app.py:82
        {"down2", CAT2, 128, d.Hu2, d.Wu2, 256, 128},    // e2, top-left crop Hu2 x Wu2 of H1 x W1     app.py:84,90-93
        {"pool2", P2, 128, d.H2, d.W2, 128, 0},          //                                            app.py:85
        {"bottleneck", BT, 256, d.H2, d.W2, 256, 0},     //                                            app.py:87
        {"up2", CAT2, 128, d.Hu2, d.Wu2, 256, 0},        //                                            app.py:89
        {"upconv2", D2, 128, d.Hu2, d.Wu2, 128, 0},      //                                            app.py:94
        {"up1", CAT1, 64, d.Hu1, d.Wu1, 128, 0},         //                                            app.py:96
        {"upconv1.0", T4, 64, d.Hu1, d.Wu1, 64, 0},      // after its ReLU (app.py:75-76); with CID_TAIL_FUSED the region holds
                                                         // the 27 z planes [N, 27, Hu1, Wu1] instead
    };
    for (const Row& r : rows)
        if (std::strcmp(stage, r.name) == 0) {
            *offset_bytes = p.off[r.buf] * sizeof(float);
            *C = r.c; *Hs = r.hs; *Ws = r.ws; *pixel_stride = r.ps; *channel_offset = r.coff;
            return CID_OK;
        }
    return CID_ERR_KEY;
}

// ---- multi-GPU: the one collective of the job, without PyTorch in it ----
// RCCL is bound at run time from the process (the library that created the caller's communicator), not linked: libcid.so
// loads on a box without RCCL, and a host that already has an RCCL (PyTorch ships its own librccl.so.1) keeps exactly one.
}  // extern "C"
namespace {
struct Rccl {
    typedef int (*GetUniqueId)(void*);
    typedef int (*CommInitRank)(void**, int, ncclUniqueIdBytes, int);
    typedef int (*CommDestroy)(void*);
    typedef int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t);
    typedef const char* (*GetErrorString)(int);
    typedef int (*CommCount)(void*, int*);
    GetUniqueId get_unique_id = nullptr; CommInitRank comm_init_rank = nullptr; CommDestroy comm_destroy = nullptr;
    Broadcast broadcast = nullptr; GetErrorString error_string = nullptr; CommCount comm_count = nullptr;
    std::string why;
    bool ok = false;
};
const Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);          // the copy this process already uses
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) { x.why = std::string("librccl.so.1 not found: ") + dlerror(); return x; }
        x.get_unique_id = (Rccl::GetUniqueId)dlsym(lib, "ncclGetUniqueId");
        x.comm_init_rank = (Rccl::CommInitRank)dlsym(lib, "ncclCommInitRank");
        x.comm_destroy = (Rccl::CommDestroy)dlsym(lib, "ncclCommDestroy");
        x.broadcast = (Rccl::Broadcast)dlsym(lib, "ncclBroadcast");
        x.error_string = (Rccl::GetErrorString)dlsym(lib, "ncclGetErrorString");
        x.comm_count = (Rccl::CommCount)dlsym(lib, "ncclCommCount");   // optional: only cid_comm_count needs it
        x.ok = x.get_unique_id && x.comm_init_rank && x.comm_destroy && x.broadcast && x.error_string;
        if (!x.ok) x.why = "librccl.so.1 lacks ncclGetUniqueId/ncclCommInitRank/ncclCommDestroy/ncclBroadcast";
        return x;
    }();
    return r;
}
int rccl_fail(cid_handle_t h, const char* what, int rc) {
    return fail(h, CID_ERR_HIP, std::string(what) + ": " + (rccl().error_string ? rccl().error_string(rc) : "RCCL error"));
}
}  // namespace
extern "C" {

int cid_comm_available(void) { return rccl().ok ? 1 : 0; }
int cid_comm_unique_id(void* id128) {
    if (!id128) return CID_ERR_INVALID;
    if (!rccl().ok) return CID_ERR_STATE;
    return rccl().get_unique_id(id128) == 0 ? CID_OK : CID_ERR_HIP;
}
int cid_comm_init_rank(void** comm, int nranks, const void* id128, int rank) {
    if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return CID_ERR_INVALID;
    if (!rccl().ok) return CID_ERR_STATE;
    ncclUniqueIdBytes id;
    std::memcpy(&id, id128, sizeof id);
    return rccl().comm_init_rank(comm, nranks, id, rank) == 0 ? CID_OK : CID_ERR_HIP;
}
int cid_comm_destroy(void* comm) {
    if (!comm) return CID_ERR_INVALID;
    if (!rccl().ok) return CID_ERR_STATE;
    return rccl().comm_destroy(comm) == 0 ? CID_OK : CID_ERR_HIP;
}
int cid_comm_count(void* comm, int* nranks) {
    if (!comm || !nranks) return CID_ERR_INVALID;
    if (!rccl().ok || !rccl().comm_count) return CID_ERR_STATE;
    return rccl().comm_count(comm, nranks) == 0 ? CID_OK : CID_ERR_HIP;
}

int cid_broadcast_weights(cid_handle_t h, void* comm, int root, int rank, void* stream) {
    if (!h) return CID_ERR_INVALID;
    if (!comm || root < 0 || rank < 0) return fail(h, CID_ERR_INVALID, "cid_broadcast_weights: null communicator or negative rank");
    if (!h->dev_blob) return fail(h, CID_ERR_STATE, "cid_broadcast_weights: no device blob attached (root: cid_upload_weights; receivers: cid_attach_weights on an allocated buffer)");
    if (!rccl().ok) return fail(h, CID_ERR_STATE, "cid_broadcast_weights: " + rccl().why);
    hipStream_t s = static_cast<hipStream_t>(stream);
    void* blob = const_cast<float*>(h->dev_blob);
    const size_t bytes = kBlob.total * sizeof(float);
    const int rc = rccl().broadcast(blob, blob, bytes, /*ncclUint8*/ 1, root, comm, s);   // in place: one 26 MB message, root -> everyone
    if (rc != 0) return rccl_fail(h, "ncclBroadcast", rc);
    if (rank != root) {   // receivers: refresh the host staging copy, so cid_get_weight / state_dict() return what the kernels use
        hipError_t e = hipMemcpyAsync(h->staging.data(), blob, bytes, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail(h, CID_ERR_HIP, std::string("cid_broadcast_weights: ") + hipGetErrorString(e));
        std::memset(h->have, 1, sizeof(h->have));
    }
    return CID_OK;
}

int cid_launch_work(int i, int N, int H, int W, double* flops, double* bytes) {
    if (i < 0 || i >= NL || !flops || !bytes) return CID_ERR_INVALID;
    Dims d;
    if (!make_dims(N, H, W, d)) return CID_ERR_SHAPE;
    // output pixels (conv) / input pixels (convT) per image, as the reference evaluates each layer
    const double s0 = (double)d.H * d.W, s1 = (double)d.H1 * d.W1, s2 = (double)d.H2 * d.W2;
    const double su2 = (double)d.Hu2 * d.Wu2, su1 = (double)d.Hu1 * d.Wu1;
    const double pix[NL] = {s0, s0, s1, s1, s2, s2, s2, su2, su2, su2, su1, su1};
    const LayerDef& L = kLayers[i];
    const double taps = L.kind == CONVT ? 4 : 9;
    *flops = 2.0 * L.cin * L.cout * taps * pix[i] * N;
    const double out_pix = L.kind == CONVT ? 4 * pix[i] : pix[i];
    *bytes = 4.0 * (N * (pix[i] * L.cin + out_pix * L.cout) + (double)ref_weight_count(L) + L.cout);
    return CID_OK;
}

// The same per LAUNCH under the handle's configuration.  It differs from the per-layer figures (a) on launches 1 and 3, which write the pooled
// tensor beside their own output (the reference's MaxPool2d modules are fused into them), and (b) when the last layer's
// channel contraction is fused into upconv1[0]'s launch (CID_TAIL_FUSED): launch 10 then carries the FLOPs of both layers and
// writes 27 z planes instead of 64 channels; launch 11 is the shifted sum (no multiply-adds) over those planes.
int cid_launch_work_ex(cid_handle_t h, int i, int N, int H, int W, double* flops, double* bytes) {
    const int rc = cid_launch_work(i, N, H, W, flops, bytes);
    if (rc != CID_OK) return rc;
    Dims d;
    make_dims(N, H, W, d);
    // launches 1 and 3 (down1[2], down2[2]) also write the 2x2-pooled tensor from their epilogue (pool1 / pool2, app.py:48,56: SURVEY 8a rows a3 / a6 —
    // the pools' reads are what the fusion removes, their writes remain): in every configuration of the handle
    if (i == 1) *bytes += 4.0 * N * (double)d.H1 * d.W1 * kLayers[1].cout;
    if (i == 3) *bytes += 4.0 * N * (double)d.H2 * d.W2 * kLayers[3].cout;
    if (!h || !fused_tail_active(h) || i < 10) return rc;
    const double px = (double)N * d.Hu1 * d.Wu1;
    double f11, b11;
    cid_launch_work(11, N, H, W, &f11, &b11);
    // z elements per pixel: 27 fp32 planes; on the fp16-storage path 7 groups x 4 halfs (27 rows and a pad), counted here in
    // fp32-sized elements like every other activation of that path (bench.py halves the bytes between the first and the last tensor)
    const double zc = h->dtype == CID_DTYPE_F16 ? 28 : 27;
    if (i == 10) {
        *flops += f11;
        *bytes += 4.0 * px * (zc - 64) + 4.0 * ref_weight_count(kLayers[11]);
    } else {
        *flops = 0.0;
        *bytes = 4.0 * (px * (zc + 3) + 3);
    }
    return CID_OK;
}

}  // extern "C"
