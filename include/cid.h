/*
 * cid.h — C ABI of the MI355X-native denoise forward ("cid" = celebrity image denoiser).
 *
 * The reference has no FFI: its boundary for this path is the torch.nn.Module protocol on the
 * object stored in PT_MODELS["denoise"] (reference backend/app.py:319-320).  Each entry point
 * below names the reference interface it stands in for.  All pointers are plain host or device
 * addresses; no torch types cross this boundary.  Device memory (weights blob, workspace, input,
 * output) is owned by the caller — in the Python host layer that is PyTorch-ROCm's allocator.
 *
 * Threading: a handle is not re-entrant — one forward in flight per handle, matching the
 * reference's single-threaded use (backend/app.py:358-359,433).
 * Errors: every function returns CID_OK (0) or a CID_ERR_* code; cid_last_error(h) holds the text.
 */
#ifndef CID_H_
#define CID_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cid_handle_s* cid_handle_t;

enum {
    CID_OK = 0,
    CID_ERR_INVALID = 1,   /* null pointer / bad argument                                   */
    CID_ERR_SHAPE = 2,     /* tensor shape not accepted (e.g. H or W < 4, wrong weight dims) */
    CID_ERR_KEY = 3,       /* unknown state_dict key                                        */
    CID_ERR_STATE = 4,     /* call out of order (forward before weights attached, ...)      */
    CID_ERR_WORKSPACE = 5, /* workspace too small or misaligned                             */
    CID_ERR_HIP = 6        /* a HIP runtime call or kernel launch failed                    */
};

/* Number of parameter tensors (24) and scalars (1,827,587) of the module.
 * reference: DenoiseGenerator.__init__, backend/app.py:39-78. */
#define CID_NUM_PARAMS 24
#define CID_NUM_PARAM_ELEMS 1827587

/* Kernel launches of one forward, in order (names: cid_launch_name). */
#define CID_NUM_LAUNCHES 12

const char* cid_version(void);

/* DenoiseGenerator()  — backend/app.py:320 (constructor; here: host-side state only). */
int cid_create(cid_handle_t* out);
/* garbage collection of the module. */
void cid_destroy(cid_handle_t h);
/* Python exception text — the reference raises; this ABI returns codes + this string. */
const char* cid_last_error(cid_handle_t h);

/*
 * Module.load_state_dict, one tensor at a time — backend/app.py:272 (via load_state_safely,
 * :257-274).  `key` is a reference state_dict key ("down1.0.weight", "up2.bias", ...; any
 * "module." prefix already stripped by the caller), `host_data` fp32 in the reference layout
 * (Conv2d [Cout,Cin,3,3], ConvTranspose2d [Cin,Cout,2,2], bias [Cout]), `shape`/`ndim` its
 * dims.  The tensor is repacked into the kernels' layout inside the handle's host staging blob.
 * Unknown key -> CID_ERR_KEY; wrong shape -> CID_ERR_SHAPE (the reference's size-mismatch error).
 */
int cid_set_weight(cid_handle_t h, const char* key, const float* host_data, const int64_t* shape, int ndim);

/* Module.state_dict()[key] — backend/trainingcode/denoise_gan_code/training.py:361.
 * Unpacks the staged tensor back into reference layout (`host_out` holds `count` floats). */
int cid_get_weight(cid_handle_t h, const char* key, float* host_out, size_t count);

/* How many of the 24 tensors have not been set since cid_create (strict=True check). */
int cid_missing_weights(cid_handle_t h, int* missing);

/* i-th state_dict key in reference order, NULL if i is out of range. */
const char* cid_param_key(int i);

/* Size in bytes of the packed device weights blob (the unit the multi-GPU path broadcasts). */
size_t cid_packed_weights_bytes(void);

/*
 * Module.to(device) for the parameters — backend/app.py:320.  Copies the packed staging blob
 * to `device_blob` (caller-owned, >= cid_packed_weights_bytes(), 256-byte aligned) on `stream`
 * (hipStream_t, may be NULL) and attaches it.
 */
int cid_upload_weights(cid_handle_t h, void* device_blob, void* stream);

/* Copy the packed host staging blob out (`bytes` must equal cid_packed_weights_bytes()): the
 * exact bytes cid_upload_weights sends to the device, for transports that move it themselves. */
int cid_export_packed(cid_handle_t h, void* host_out, size_t bytes);
/* Replace the host staging blob with packed bytes produced by another handle's
 * cid_export_packed / device blob; afterwards all 24 tensors count as set and cid_get_weight
 * returns them in reference layout. */
int cid_import_packed(cid_handle_t h, const void* host_in, size_t bytes);

/* Adopt a device blob that already holds packed weights (e.g. filled by an RCCL broadcast from
 * the rank that ran cid_upload_weights).  No copy. */
int cid_attach_weights(cid_handle_t h, const void* device_blob);

/* Output spatial size of the forward: Ho = 4*floor(H/4), Wo = 4*floor(W/4) (two floor-mode
 * 2x2 pools, two x2 transposed convs, skip tensors cropped top-left) — backend/app.py:80-103.
 * H or W < 4 -> CID_ERR_SHAPE (the reference raises "Output size is too small"). */
int cid_out_shape(int H, int W, int* Ho, int* Wo);

/* Bytes of scratch device memory one forward of [N,3,H,W] needs (activation arena, NHWC fp32). */
int cid_workspace_bytes(int N, int H, int W, size_t* bytes);

/*
 * Module.__call__(x) / forward — backend/app.py:433 (net(x_pt)), :80-103.
 * in_nchw : device, fp32, contiguous [N,3,H,W], values nominally in [-1,1]  (app.py:401-406)
 * out_nchw: device, fp32, contiguous [N,3,Ho,Wo], tanh range                (app.py:103)
 * workspace: device scratch, 256-byte aligned, >= cid_workspace_bytes(N,H,W)
 * stream  : hipStream_t (NULL = default stream).  Asynchronous: returns after enqueueing.
 */
int cid_forward(cid_handle_t h, const float* in_nchw, float* out_nchw, int N, int H, int W,
                void* workspace, size_t workspace_bytes, void* stream);

/*
 * The forward with the reference's pre/post-processing folded into the first and last kernel
 * (SURVEY.md 8f row f1).  Formats:
 *   CID_FMT_F32_NCHW  fp32 [N,3,H,W] — what cid_forward takes/returns
 *   CID_FMT_U8_NHWC   uint8 [N,H,W,3], PIL/numpy image layout.  As input it is normalised on the fly,
 *                     (u8/255 - 0.5)/0.5 in fp32 = ToTensor + Normalize(0.5,0.5), backend/app.py:401-405;
 *                     as output it is (uint8)(clamp(y*0.5+0.5, 0, 1)*255), truncating like
 *                     ToPILImage's mul(255).byte(), backend/app.py:435,471-472.
 * Any combination is allowed; cid_forward == cid_forward_ex(F32_NCHW, F32_NCHW).
 */
enum { CID_FMT_F32_NCHW = 0, CID_FMT_U8_NHWC = 1 };
int cid_forward_ex(cid_handle_t h, const void* in, int in_fmt, void* out, int out_fmt, int N, int H, int W,
                   void* workspace, size_t workspace_bytes, void* stream);

/*
 * The reference's view transform alone — backend/app.py:435 (y*0.5+0.5, clamp to [0,1]) and :471-472 (ToPILImage: mul(255).byte(),
 * truncating), as the iterated caller applies it to EVERY fed-back iteration (denoise_eavl_iter.py:97-110): a device fp32
 * [N,3,H,W] tensor in tanh range -> device uint8 [N,H,W,3].  Same arithmetic as cid_forward_ex's CID_FMT_U8_NHWC output, for callers that
 * keep the fp32 tensor (to feed it back) and also want its image.  No handle: it needs no weights.
 */
int cid_view_u8(const float* in_nchw, void* out_u8_nhwc, int N, int H, int W, void* stream);

/*
 * The reference server's handling of arbitrary upload sizes (backend/app.py:276-281 get_padding, :384-385
 * transforms.Pad(padding, fill=0) in front of ToTensor/Normalize, :474-480 crop of the result), as index arithmetic in the
 * first and the last kernel — no padded copy of the image and no uncropped output exist:
 *   in   : [N,3,H,W] fp32 or [N,H,W,3] uint8 — the caller's image, UNPADDED
 *   the network runs on [H + pad_top + pad_bottom, W + pad_left + pad_right]; the band around the image is uint8 0, i.e.
 *          (0/255 - 0.5)/0.5 = -1.0 (for CID_FMT_F32_NCHW input the band is -1.0 as well)
 *   out  : [N,3,H,W] fp32 or [N,H,W,3] uint8 — rows [pad_top, pad_top + H) and columns [pad_left, pad_left + W) of the
 *          network's output; that window must exist (it does when the padded size is a multiple of 4, the reference's rule),
 *          otherwise CID_ERR_SHAPE
 * workspace: cid_workspace_bytes(N, H + pad_top + pad_bottom, W + pad_left + pad_right).  With all pads zero and H, W multiples
 * of 4 this is cid_forward_ex.  Pads outside [0, 4096] -> CID_ERR_INVALID.
 */
int cid_forward_padded(cid_handle_t h, const void* in, int in_fmt, void* out, int out_fmt, int N, int H, int W,
                       int pad_left, int pad_top, int pad_right, int pad_bottom, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Same forward, with a HIP event recorded on `stream` around every kernel launch; synchronises
 * the stream and writes the CID_NUM_LAUNCHES per-launch durations in milliseconds to launch_ms.
 * Measurement aid for bench.py's roofline object; not part of the reference surface.
 */
int cid_forward_timed(cid_handle_t h, const float* in_nchw, float* out_nchw, int N, int H, int W,
                      void* workspace, size_t workspace_bytes, void* stream, float* launch_ms);

/*
 * Per-launch timing across many forwards without a host sync per forward (bench.py's timed
 * region).  cid_timing_begin arms the handle: each of the next `max_forwards` cid_forward calls
 * records CID_NUM_LAUNCHES+1 HIP events on its stream.  cid_timing_end synchronises `stream`,
 * writes the per-launch durations summed over the recorded forwards (milliseconds) to
 * launch_ms_sum[CID_NUM_LAUNCHES] and the number of forwards to *forwards, and disarms.
 */
int cid_timing_begin(cid_handle_t h, int max_forwards);
int cid_timing_end(cid_handle_t h, void* stream, float* launch_ms_sum, int* forwards);

/* Name of the i-th launch ("down1.0", ..., "upconv1.2"), the reference layer(s) it computes. */
const char* cid_launch_name(int i);
/* Device kernel symbol prefix of the i-th launch under the handle's current algorithm (to match
 * rocprofv3 kernel-trace rows). */
const char* cid_launch_kernel(cid_handle_t h, int i);

/*
 * Algorithm of the eight GEMM-shaped 3x3 convolutions (down1[2] ... upconv1[0]); head, tail and the
 * transposed convolutions are unaffected (except under CID_ALGO_SPLIT16, which runs the transposed convolutions in its own arithmetic too).  All compute the reference's nn.Conv2d(k=3,p=1) in fp32 (exact-fp32 MFMA):
 *   CID_ALGO_DIRECT     implicit GEMM, 9 taps: 9 multiplies per output pixel and (ci,co)
 *   CID_ALGO_WINOGRAD64 Winograd F(2x2,3x3): 4 multiplies per pixel (round 1's default)
 *   CID_ALGO_WINOGRAD42 Winograd F(4x2,3x3), tiles 4 wide x 2 high, interpolation points 0, +-3/4, +-3/2, inf: 3 multiplies
 *                       per pixel: the default
 *   CID_ALGO_SPLIT16    (round 4, OPT-IN, not exact-fp32 MFMA) split-operand convolution on the fp16 MFMA: fp32 tensors in and out; every fp32 operand is
 *                       taken as hi + lo with hi = half(x), lo = half(x - hi) (22 bits of mantissa), every product as hi*hi + hi*lo + lo*hi on
 *                       v_mfma_f32_16x16x32_f16 with fp32 accumulators (lo*lo, 2^-22 relative, dropped).  9 multiplies per pixel (direct form), each on three
 *                       half products.  Measured against a float64 evaluation (profiles/r04_accuracy_study.txt, He-gain weights, faces / white noise, units of
 *                       1e-6): 5.1 / 8.6 where the exact-fp32 direct kernel has 4.2 / 6.3, the Winograd default 2.2 / 4.0 and ATen fp32 2.2 / 2.5 — the least accurate
 *                       of the four algorithms, inside the 1e-5 contract with the thinnest margin.  It passes every 1e-5 parity test of the suite, but its arithmetic type is
 *                       "fp32 operands as two halfs, fp16 MFMA, fp32 accumulate": the default and the headline benchmark stay on CID_ALGO_WINOGRAD42.
 *                       CID_TAIL_FUSED works under it: upconv1[2]'s contraction runs in the same split-operand arithmetic in upconv1[0]'s epilogue and
 *                       leaves the 27 fp32 planes k_conv_tail_z sums.
 * (value 1 was round 1's first Winograd kernel, removed: same bits as WINOGRAD64, slower.)
 * No reference counterpart (the reference leaves the choice to ATen/oneDNN/cuDNN).
 */
enum { CID_ALGO_DIRECT = 0, CID_ALGO_WINOGRAD64 = 2, CID_ALGO_WINOGRAD42 = 3, CID_ALGO_SPLIT16 = 4 };

/*
 * Storage type of activations and weights between the first and the last kernel (BASELINE configs[4]):
 *   CID_DTYPE_F32  the reference's arithmetic (default): fp32 storage, exact-fp32 MFMA
 *   CID_DTYPE_F16  IEEE half storage, v_mfma_f32_32x32x16_f16 with fp32 accumulators, bias/ReLU/pool in fp32,
 *                  one rounding to half per stored element; direct implicit GEMM for all ten GEMM layers.
 * The caller-side tensors (cid_forward / cid_forward_ex) keep their formats; only the arena and the weight
 * segments read change.  A different numerical contract from the reference's fp32 (tolerances: tests/).
 */
enum { CID_DTYPE_F32 = 0, CID_DTYPE_F16 = 1 };   /* (the half path's kernels use v_mfma_f32_16x16x32_f16 since round 2) */
int cid_set_compute_dtype(cid_handle_t h, int dtype);
int cid_get_compute_dtype(cid_handle_t h, int* dtype);
int cid_set_conv_algo(cid_handle_t h, int algo);
int cid_get_conv_algo(cid_handle_t h, int* algo);
/*
 * How the last layer (upconv1[2] = Conv2d(64,3,3,p=1) + tanh, backend/app.py:77,103) runs on the fp32 path; same function:
 *   CID_TAIL_FUSED  (default) its 64 -> 27 (tap x channel) contraction runs in the epilogue of upconv1[0]'s kernel, on the
 *                   tile still in LDS; the last launch is the nine-tap shifted sum + bias + tanh over 27 fp32 planes (CID_DTYPE_F32; needs
 *                   a Winograd algorithm, with CID_ALGO_DIRECT the handle behaves as CID_TAIL_BANDS) or over 7 planes of 4 halfs (the same 27
 *                   rows 3 tap + co and a pad; CID_DTYPE_F16, round 4).
 *   CID_TAIL_BANDS  separate kernel: a workgroup slides down a band of rows, the contraction is computed once per pixel
 *                   (images up to 128 pixels wide, wider ones take CID_TAIL_TILES)
 *   CID_TAIL_TILES  separate kernel: 8x32-pixel tiles, the contraction is computed over each tile's halo (round 1's kernel)
 */
enum { CID_TAIL_FUSED = 0, CID_TAIL_BANDS = 1, CID_TAIL_TILES = 2 };
int cid_set_tail_algo(cid_handle_t h, int algo);
int cid_get_tail_algo(cid_handle_t h, int* algo);
/* Algorithmic work of the i-th launch for an [N,3,H,W] forward: conv/convT FLOPs (2*MAC) and
 * fp32 bytes (input activations + output activations + weights, each once) — SURVEY.md 8(a). */
int cid_launch_work(int i, int N, int H, int W, double* flops, double* bytes);
/* The same per LAUNCH under the handle's configuration: launches 1 and 3 also count the 2x2-pooled tensor they write (pool1 / pool2,
 * backend/app.py:48,56, run in their epilogues: SURVEY.md 8(a) rows a3 / a6 less the pools' reads); with CID_TAIL_FUSED launch 10 also carries
 * upconv1[2]'s FLOPs and writes 27 planes instead of 64 channels, and launch 11 only sums, adds the bias and applies tanh. */
int cid_launch_work_ex(cid_handle_t h, int i, int N, int H, int W, double* flops, double* bytes);

/*
 * Where the output of one of the reference module's stages lives in the workspace of an [N,3,H,W] forward (NHWC,
 * fp32 — or half elements from the same base when the compute dtype is CID_DTYPE_F16).  `stage` is the attribute name of
 * the reference module whose forward-hook output it is (backend/app.py:42-78): "down1", "pool1", "down2", "pool2",
 * "bottleneck", "up2", "upconv2", "up1", plus "upconv1.0" (upconv1[0] after its ReLU: the last layer's input; with
 * CID_TAIL_FUSED that region holds the 27 planes z[N,27,Hs,Ws] instead and the view does not apply).  Element (n, y, x, c) of the stage is at
 *     offset_bytes/elem_size + ((n*Hs + y)*Ws + x)*pixel_stride + channel_offset + c        for y < Hs, x < Ws, c < C.
 * Skip tensors ("down1", "down2") are stored only over the region the concat keeps (top-left crop, app.py:90-92,97-99).
 * "upconv1" (pre-tanh) is never stored: it is fused into the last kernel.  Testing aid for per-stage parity.
 */
int cid_stage_view(const char* stage, int N, int H, int W, size_t* offset_bytes, int* C, int* Hs, int* Ws,
                   int* pixel_stride, int* channel_offset);

/*
 * Multi-GPU (one process per GPU, batch sharded): the job's ONE collective — an RCCL broadcast of the
 * packed weights blob over xGMI from the rank that loaded the checkpoint — as a C entry point, so that a host which is
 * not PyTorch can distribute weights too.  The reference has no distributed code; nothing here replaces a reference call.
 *   comm   : an ncclComm_t (opaque pointer) spanning the ranks; create it with RCCL directly or with the three helpers
 *            below (thin wrappers over ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy; the 128-byte unique id
 *            travels from rank 0 to the others by whatever control channel the host has).
 *   before : root has run cid_upload_weights; every other rank has attached an allocated, 256-byte aligned device buffer
 *            of cid_packed_weights_bytes() with cid_attach_weights.
 *   after  : every rank's attached blob holds root's weights (in place, on `stream`); non-root handles have also
 *            refreshed their host copy (the call synchronises `stream` there), so cid_get_weight returns the new tensors.
 * RCCL is resolved at run time from the process (dlopen of librccl.so.1): CID_ERR_STATE if it is not available.
 */
int cid_comm_available(void);   /* 1 if RCCL could be resolved in this process (the three calls below can work), else 0; no side effect */
int cid_comm_unique_id(void* id128);
int cid_comm_init_rank(void** comm, int nranks, const void* id128, int rank);
int cid_comm_destroy(void* comm);
int cid_comm_count(void* comm, int* nranks);   /* ncclCommCount: how many ranks RCCL itself says the communicator spans */
int cid_broadcast_weights(cid_handle_t h, void* comm, int root, int rank, void* stream);

/*
 * Testing aid (no reference counterpart): fills the LDS of every CU with NaN on `stream`.  LDS is not cleared between
 * kernels, so a forward enqueued after it exposes any kernel that reads LDS words it has not written.
 */
int cid_debug_poison_lds(void* stream);
/*
 * Testing / measurement aid (no reference counterpart; process-wide): workgroups per CU of the Winograd F(4x2) launches.
 * k >= 1: a launch with more (tile, column block) items than k workgroups per CU is run by that many WALKING workgroups
 * (default 2; the kernel's LDS use admits no more).  0: every item gets its own workgroup.  Both must give the same bits.
 * Returns the previous value; a negative argument only queries.
 */
int cid_debug_winograd_workgroups_per_cu(int k);
/* Measurement aid (process-wide; default 0): bit mask over the column-block counts NB (2, 4) whose WALKING Winograd F(4x2) launches give every XCD
 * group ONE column block of a tile range instead of all NB blocks of its tiles back to back (profiles/r04_xnb_experiment.txt: bit-identical results,
 * -3 % fabric traffic, +-0 time).  Returns the previous mask; a negative argument only queries. */
int cid_debug_winograd_column_block_per_xcd(int mask);
/* The same for the 3x3 launches of the fp16-storage path (k_conv3x3_h16): k walking workgroups per CU (at most 3), 0 = one item per workgroup — the
 * DEFAULT since round 4 (measured faster once the epilogue lost its LDS staging: profiles/r04_ab_f16_walk_vs_not.txt); walking stays a tested option for the
 * launches without a fused pool (down1[2] and down2[2] always take one item per workgroup). */
int cid_debug_half_workgroups_per_cu(int k);

#ifdef __cplusplus
}
#endif
#endif /* CID_H_ */
