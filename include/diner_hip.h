/*
 * diner_hip.h -- C ABI of libdiner_hip.so: DINER's volumetric-rendering hot path on MI355X (gfx950).
 *
 * The reference (malteprinzler/diner) is pure Python/PyTorch and has no FFI of its own; the
 * boundary it offers is the Python module API of the src/models Python modules.  Each entry point below replaces
 * the body of one of those Python functions (cited as file:line in /root/reference) and is what a
 * maintainer would bind with ctypes from those modules -- see INTEGRATION.md.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HIP) to contiguous row-major fp32 unless stated;
 *   - every call only ENQUEUES work on the caller's stream (hipStream_t passed as void*);
 *     no hidden synchronisation, no global mutable state besides the thread-local error string, and no
 *     device allocation -- with one exception: diner_mlp_create() hipMallocs the buffers of the packed-weights
 *     handle it returns (freed by diner_mlp_destroy) and synchronises `stream` once before it returns;
 *   - return value: 0 = ok, negative = DINER_E_*; diner_last_error() gives the message;
 *   - the caller (PyTorch's caching allocator in the Python host) owns all buffers; the library
 *     owns only the packed-weights handle created by diner_mlp_create().
 */
#ifndef DINER_HIP_H
#define DINER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DINER_ABI_VERSION 6

#define DINER_E_INVALID     (-1)  /* bad argument (null pointer, size, unsupported configuration) */
#define DINER_E_UNSUPPORTED (-2)  /* configuration outside what the kernels are built for        */
#define DINER_E_HIP         (-3)  /* a HIP runtime call failed                                     */

/* Per-object scene state: what the reference keeps on PixelNeRF / SpatialEncoder after encode()
 * (pixelnerf.py:47-51; image_encoder.py:232-236, :290-291). */
typedef struct DinerScene {
  const float* latent_cl;   /* (NV, Hf, Wf, C)  feature map, CHANNELS-LAST (re-laid-out once per encode) */
  const float* latent_proj; /* (3, NV, Hf, Wf, C) lin_z[b](latent)+bias, written by diner_scene_prepare_f32; the field
                               entry points gather from these maps (resnetfc.py:153-155 hoisted out of the sample loop).
                               Maps 1 and 2 also carry the fc_1 bias of the block before them (the two constants are added to
                               the residual stream at the same point): opaque to the caller, valid for the MLP handle that
                               prepared them */
  const float* depth;       /* (NV, Hs, Ws)     source depth maps, 0 = background                          */
  const float* depth_std;   /* (NV, Hs, Ws)     depth standard deviation                                    */
  const float* normals;     /* (NV, 3, Hs, Ws)  normal maps (planar, as produced by depth2normal)          */
  /* ---- the only HOST pointers of this struct (suffix _host): three tiny camera arrays, read on the host at call time and
   *      copied by value into the kernel arguments (SGPRs); never dereferenced on the device ---- */
  const float* poses_host;  /* HOST (NV, 4, 4)  world->camera extrinsics, row-major (pixelnerf.py:47)               */
  const float* focal_host;  /* HOST (NV, 2)     fx, fy                               (pixelnerf.py:49)               */
  const float* c_host;      /* HOST (NV, 2)     cx, cy                               (pixelnerf.py:48)               */
  const float* std_pad_scale; /* (100)          multipliers exp(e/12*ln2), e = 0..99, of the exponential std padding
                                                 (torch_helpers.py:110-120 with pad_size=100, pad_double_width=12,
                                                 image_encoder.py:185-194); computed by the host exactly as the
                                                 reference does so that padded sigma values are bit-identical        */
  float img_w, img_h;       /* PixelNeRF.image_shape = [W, H] (pixelnerf.py:50-51)                         */
  float feature_padding;    /* SpatialEncoder.feature_padding (image_encoder.py:59), 32 in the shipped configs */
  int32_t nv, C, Hf, Wf, Hs, Ws;
  uint64_t proj_stamp;      /* diner_mlp_stamp() of the handle that wrote latent_proj (set by the caller after
                               diner_scene_prepare_f32).  The field entry points return DINER_E_INVALID when it is not the
                               stamp of the handle they are called with: maps prepared with another (or an older) handle carry
                               that handle's biases */
  const void* latent_proj_f16; /* ABI v4: the same three maps as fp16 (3, NV, Hf, Wf, C), channels in the order the plain-fp16 field
                               kernel consumes them, written by diner_scene_prepare_f16 from latent_proj (same proj_stamp).  Only
                               DINER_PRECISION_F16 reads it (half the gather bytes of that mode); NULL for the other modes */
  uint64_t proj_stamp_f16;  /* ABI v5: diner_mlp_stamp() of the handle whose latent_proj the fp16 copy was made from (set by the caller after
                               diner_scene_prepare_f16).  DINER_PRECISION_F16 returns DINER_E_INVALID when it is not the stamp of the handle
                               it is called with: a caller that re-ran diner_scene_prepare_f32 with new weights but not _prepare_f16 would
                               otherwise render from stale fp16 maps without an error */
} DinerScene;

/* ResnetFC parameters as the reference stores them (nn.Linear: weight (out,in), bias (out));
 * resnetfc.py:72-127.  Host or device pointers are both accepted by diner_mlp_create (flag). */
typedef struct DinerMlpParams {
  int32_t d_in, d_latent, d_hidden, d_out, n_blocks, combine_layer;
  /* positional encoding that produces the d_in inputs (pixelnerf.py:15-18, positional_encoding.py:12-31): the field
   * entry points encode x_c (3 -> 3*(2F+1)) and the depth difference (1 -> 2F+1) in registers.  num_freqs must be 6
   * and include_input non-zero (d_in = 55); freq_factor is honoured (6.28 in every shipped config). */
  int32_t num_freqs, include_input;
  float freq_factor;
  const float* lin_in_w;  const float* lin_in_b;       /* (d_hidden, d_in), (d_hidden)          */
  const float* lin_out_w; const float* lin_out_b;      /* (d_out, d_hidden), (d_out)            */
  const float* const* fc0_w; const float* const* fc0_b; /* n_blocks x (d_hidden,d_hidden),(d_hidden) */
  const float* const* fc1_w; const float* const* fc1_b;
  const float* const* lin_z_w; const float* const* lin_z_b; /* min(combine_layer,n_blocks) x (d_hidden,d_latent) */
} DinerMlpParams;

typedef struct DinerMlp DinerMlp;   /* opaque: weights re-packed into MFMA-fragment order, resident in HBM */

int         diner_abi_version(void);
const char* diner_last_error(void);

/* ---- packed weights ---------------------------------------------------------------------------
 * Packs the ResnetFC parameters (DEVICE pointers) into stage-tile order on `stream`.
 * Supported: d_in=55, d_latent=512, d_hidden=512, d_out=4, n_blocks=5, combine_layer=3, num_freqs=6, include_input=1
 * (the configuration of configs/train_dtu.yaml:39-50 and train_facescape.yaml), NV = 4; anything else returns
 * DINER_E_UNSUPPORTED before any device work.  A failed call leaves no allocation behind. */
int diner_mlp_create(const DinerMlpParams* p, void* stream, DinerMlp** out);
int diner_mlp_destroy(DinerMlp* mlp);
/* ABI v6: new parameter values into an EXISTING handle, packed on `stream` -- no allocation and no synchronisation (diner_mlp_create
 * hipMallocs ~12 buffers, waits for the stream once, and diner_mlp_destroy hipFrees them: three device-wide synchronisations per optimiser
 * step when a training loop re-creates its handle; resnetfc.py:72-127 under the optimiser's in-place updates, diner.py:217-290).  The
 * parameter tensors must stay valid until the packing has run on `stream`.  The handle gets a new diner_mlp_stamp(): maps projected with
 * the old values are refused.  The weight range is not read back: diner_mlp_weights_fit_f16x3 and the inference entry points read it
 * (one stream wait) the first time they need it after an update; the fused training forward never does -- it tests the range on the device.
 * flags: DINER_MLP_UPDATE_TRAIN_ONLY packs only what diner_field_train_forward_fused_f32 / _batch_f32 read (the four-wave f16x3 layouts,
 * biases, the constants of the projected maps): the inference entry points then return DINER_E_INVALID for this handle until an update
 * without the flag. */
#define DINER_MLP_UPDATE_TRAIN_ONLY 1
int diner_mlp_update(DinerMlp* mlp, const DinerMlpParams* p, int flags, void* stream);
/* Largest |weight| over all weight matrices (biases stay fp32 in every mode), reduced on the device at pack time (one 4-byte read back, after a
 * stream synchronise): DINER_PRECISION_F16X3 / _F16 carry the weights x16 as fp16 and need it below 1024.
 * Returns 1 when the f16 modes may be used with this handle, 0 when not, <0 on error; *max_abs (optional) receives the value. */
int diner_mlp_weights_fit_f16x3(const DinerMlp* mlp, float* max_abs);
/* Identity of a packed-weights handle: unique per diner_mlp_create call of the process, never 0.  Goes into
 * DinerScene.proj_stamp after diner_scene_prepare_f32(scene, mlp, ...). */
uint64_t diner_mlp_stamp(const DinerMlp* mlp);
/* Number of field launches with this handle whose fp16-operand pass left the fp16 range (or met a non-finite input) and were
 * recomputed by the exact-fp32 kernels on the device (see DINER_PRECISION_* below) -- a 2-3x slower launch that is otherwise
 * invisible.  Waits for `stream` (one 4-byte read back); `reset` != 0 zeroes the counter. */
int diner_mlp_fallback_count(const DinerMlp* mlp, long long* launches, int reset, void* stream);

/* ---- a1+a2+a3+a4: NeRFRendererDGS.sample_depthguided + fill_up_uniform_samples ---------------
 * (nerf_renderer.py:39-63, :65-190, :367-397; torch_helpers.py:215-223; image_encoder.py:148-223)
 *   rays         (NR, 8)  [o(3), d(3), near, far]
 *   t_base       (n_cand) the stratification offsets torch.linspace(0, 1-1/n_cand, n_cand)
 *   noise_*      explicit noise (parity mode) or NULL -> counter-based Philox4x32-10 keyed by (`seed`, ray_index0 + i) for ray i of
 *                the call: ray_index0 = index of rays[0] in the caller's ray list, so that a frame rendered with one seed does not
 *                depend on how its rays are split into batches or sharded across GPUs (the reference's noise depends on both)
 *                noise_coarse (NR,n_cand) U[0,1); noise_gauss (NR,G) N(0,1); noise_fill (NR,K) U[0,1)
 *                indexed by the SORTED column of the empty slot
 *   z_out        (NR, K)  ascending z per ray
 *   z_unfilled   optional (NR, K): the sampler output before the fill (zeros = empty), or NULL
 * Limits: n_cand <= 1024, K <= 256, 0 <= G <= K. */
int diner_sample_depthguided_f32(const DinerScene* scene, const float* rays, int NR, int n_cand, int K, int G,
                                 float depth_diff_max, const float* t_base,
                                 const float* noise_coarse, const float* noise_gauss, const float* noise_fill,
                                 uint64_t seed, long long ray_index0, float* z_out, float* z_unfilled, void* stream);

/* Stage-level entry for tests: fill_up_uniform_samples alone (nerf_renderer.py:367-397). */
int diner_fill_uniform_f32(const float* z_in, const float* rays, int NR, int K, const float* noise_fill,
                           uint64_t seed, long long ray_index0, float* z_out, void* stream);

/* ---- per-scene preparation: hoist of the three lin_z projections (resnetfc.py:153-155) ---------
 * lin_z[b] is linear and bilinear/border weights sum to 1, so lin_z[b](interp(latent)) == interp(lin_z[b](latent)):
 * the projections are applied once per feature-map pixel here instead of once per (sample, view).  Must be called
 * (and scene->latent_proj set to the output buffer of diner_scene_proj_bytes(scene) bytes) whenever the latent map or
 * the lin_z parameters change, before any diner_field_* / diner_render_f32 call on that scene. */
size_t diner_scene_proj_bytes(const DinerScene* scene);
int diner_scene_prepare_f32(const DinerScene* scene, const DinerMlp* mlp, float* latent_proj_out, void* stream);
/* ABI v4, DINER_PRECISION_F16 only ("fp16 MLP on MFMA", create_prediction_folder.py:44-47 with configs/evaluate_diner_on_facescape.yaml):
 * fp16 copy of scene->latent_proj (which must be set and current) into latent_proj_f16_out, diner_scene_proj_f16_bytes(scene) bytes;
 * the caller then stores the pointer in scene->latent_proj_f16.  Values beyond the fp16 range become inf and end up in the launch's
 * overflow flag (the gated exact-fp32 pass recomputes it from latent_proj). */
size_t diner_scene_proj_f16_bytes(const DinerScene* scene);
int diner_scene_prepare_f16(const DinerScene* scene, void* latent_proj_f16_out, void* stream);

/* ---- arithmetic of the MLP GEMMs: a PER-CALL argument of the field entry points (no process-wide state) ---------
 * DINER_PRECISION_FP32  exact fp32 MFMA (v_mfma_f32_16x16x4_f32), an fmaf chain per output (mlp.hip).
 * DINER_PRECISION_F16X3 "f16x3" split products: each fp32 product a*w is evaluated as a_hi*w_hi + a_hi*w_lo + a_lo*w_hi
 *                       on the fp16 MFMA with fp32 accumulation and power-of-two pre-scaling; ~2^-21 relative error per
 *                       product, end-to-end results within fp32 round-off class of the reference (parity tests hold it
 *                       to the same 1e-4 bar as FP32).  Feature-sliced kernels (mlp_h3n.hip).  Requires |w| < 1024 and
 *                       hidden activations < 6.5e4 (weights are checked by the host; see diner_mlp_weights_fit_f16x3);
 *                       one projected map (NV*Hf*Wf*2 KB) must stay below 4 GiB, else DINER_E_UNSUPPORTED.
 * DINER_PRECISION_F16   plain fp16 operands (hi parts only, one MFMA per product), fp32 accumulation, same kernels:
 *                       BASELINE configs[4] ("fp16 MLP on MFMA").  ~1e-3 relative on rendered colours: OUTSIDE the
 *                       1e-4 parity bar, never a default.
 * diner_mlp_forward_f32 (explicit matrices), diner_scene_prepare_f32 and the training path always use exact fp32. */
#define DINER_PRECISION_FP32  0
#define DINER_PRECISION_F16X3 1
#define DINER_PRECISION_F16   3   /* (2 is retired: it meant the parity-grade split mode in ABI v1 and is rejected, so that an old
                                     caller passing the bare integer cannot end up in the reduced-precision mode) */

/* ---- a5+a6+a7+a8: PixelNeRF.forward at ray samples -------------------------------------------
 * (pixelnerf.py:55-145; positional_encoding.py:33-53; image_encoder.py:97-170; resnetfc.py:129-159)
 * Points are o + z*d for every (ray, sample); view directions are the ray directions.
 *   precision  DINER_PRECISION_* (above)
 *   field_out  (NR*K, 4) = [sigmoid(r,g,b), relu(sigma)]
 *   workspace  diner_field_workspace_bytes(NR*K) bytes of device scratch */
size_t diner_field_workspace_bytes(long long n_points);
int diner_field_from_rays_f32(const DinerScene* scene, const DinerMlp* mlp, const float* rays, const float* z,
                              int NR, int K, int precision, float* field_out, void* workspace, void* stream);
/* Same, explicit points / view directions (P,3): PixelNeRF.forward(xyz, viewdirs) (pixelnerf.py:55). */
int diner_field_from_points_f32(const DinerScene* scene, const DinerMlp* mlp, const float* xyz, const float* viewdirs,
                                long long P, int precision, float* field_out, void* workspace, void* stream);

/* ---- a7 alone: ResnetFC.forward(zx, combine_dim) on an explicit (NV, B, d_latent+d_in) matrix --
 * workspace: diner_mlp_forward_workspace_bytes(B) bytes. */
size_t diner_mlp_forward_workspace_bytes(long long B);
int diner_mlp_forward_f32(const DinerMlp* mlp, const float* zx, long long B, float* out /* (B,4) raw */,
                          void* workspace, void* stream);

/* ---- a9: compositing (nerf_renderer.py:299-301, :341-360) ------------------------------------
 *   field (NR*K,4), z (NR,K), rays (NR,8) -> rgb (NR,3), depth (NR), weights (NR,K) or NULL */
int diner_composite_f32(const float* field, const float* z, const float* rays, int NR, int K, int white_bkgd,
                        float* rgb_out, float* depth_out, float* weights_out, void* stream);

/* ---- a10: NeRFRendererDGS.composite / forward (nerf_renderer.py:286-365, :399-424) -----------
 * field + composite in one call; `field_ws` is (NR*K,4) scratch the caller provides. */
int diner_render_f32(const DinerScene* scene, const DinerMlp* mlp, const float* rays, const float* z, int NR, int K,
                     int white_bkgd, int precision, float* rgb_out, float* depth_out, float* weights_out,
                     float* field_ws, void* workspace, void* stream);

/* ---- stage-level entries that back the reference's small public methods ---------------------- */
/* PositionalEncoding.forward (positional_encoding.py:33-53): x (N,d_in) -> (N, d_in*(2F+include_input)) */
int diner_posenc_f32(const float* x, long long N, int d_in, int num_freqs, float freq_factor, int include_input,
                     float* out, void* stream);
/* SpatialEncoder.index / index_depth / index_depth_std / index_normal (image_encoder.py:97-223):
 *   uv (NV, N, 2) in [-1,1] -> out (NV, Cout, N); mode: 0 latent (bilinear/border, Cout=C),
 *   1 depth (nearest/border), 2 depth_std (nearest on the 100px exponential padding, zeros), 3 normal. */
int diner_index_f32(const DinerScene* scene, int mode, const float* uv, long long N, float* out, void* stream);

/* ---- per-image preparation either side of the renderer ------
 * depth2normal (reference src/util/depth2normal.py:7-87): dmap (N,1,H,W) device, K (N,3,3) device ->
 *   normals (N,3,H,W) device: normalize(cross(down - up, right - left)) of the back-projected pixel centres with
 *   replicate padding; pixels with a background neighbour take the un-cleaned normal of the pixel shifted away from
 *   the hole; 0 where depth == 0 (NaN where the reference produces 0/0). */
int diner_depth2normal_f32(const float* dmap, const float* K, int N, int H, int W, float* normals, void* stream);
/* gen_rays (reference src/util/cam_geometry.py:5-48): rays [ray0, ray0 + n_rays) of the row-major (H, W) pixel-centre
 *   ray list of each of B cameras -> out (B, n_rays, 8) device = [origin(3), world direction(3), near, far].
 *   extrinsics (B,4,4) world->camera, intrinsics (B,3,3), z_near / z_far (B): HOST float32; B <= 16 per call.
 *   A sharded rank passes its own [ray0, ray0 + n_rays) (diner_amd/render.py::shard_range). */
int diner_gen_rays_f32(const float* extrinsics, const float* intrinsics, const float* z_near, const float* z_far, int B,
                       int W, int H, long long ray0, long long n_rays, float* out, void* stream);

/* Image output (row f3): what DINER.create_prediction_folder does with a rendered image before it is written
 * (diner.py:119-133): torchvision.utils.save_image's quantisation, uint8(clamp(v * 255 + 0.5, 0, 255)), of a (3,H,W)
 * image into (H,W,3) bytes; and torch_cmap (torch_helpers.py:42-75): per-image min / max (NaNs skipped; the keys come back
 * as order-preserving int32 images of the floats, see diner_amd/imageio.py), then the matplotlib lookup
 * index = int((x - vmin) / (vmax - vmin) * 256) clipped to [0, 255] in float64 through a 256 x 3 uint8 table that the
 * host quantised the same way. */
int diner_quantize_rgb_u8(const float* img, int H, int W, unsigned char* out, void* stream);
int diner_minmax_f32(const float* x, long long n, float* out2, void* stream);
int diner_colormap_u8(const float* x, long long n, const unsigned char* lut_u8, double vmin, double vmax,
                      unsigned char* out, void* stream);

/* ---- training path: building blocks of the un-fused forward that keeps activations and
 * of the backward pass that torch autograd performs in DINER.calc_losses (diner.py:217-290); driven by
 * diner_amd/train.py.  All pointers are device pointers unless noted; enqueue-only on `stream`.
 *
 * General fp32 GEMM on the matrix cores, C (M x N, row stride ldc) = op(A) (M x K) . op(B) (K x N), row-major:
 *   flags: 1 A is stored K x M, 2 B is stored N x K (torch Linear weights: y = x W^T), 4 / 8 relu applied to A / B while
 *   loading, 16 C += result, 32 atomicAdd into C (required for k_split > 1: weight gradients reduce over ~1e4 rows),
 *   64 exact fp32 MFMA products.  Without flag 64 the product runs as "bf16x6": each fp32 operand is split into three bf16
 *   terms and the six products above 2^-24 are accumulated in fp32 on the bf16 MFMA (2.6x the fp32 MFMA peak; bf16 has fp32's
 *   exponent range, so no scaling and no range limits) -- products as accurate as fp32 rounding itself (ragged and skinny
 *   shapes ride along in zero-padded 128 x 128 x 32 tiles);
 *   bias (N) added to every row, mask (M x N, stride ldc): C = 0 where mask <= 0 (relu adjoint). */
int diner_gemm_f32(const float* A, const float* B, float* C, long long M, int N, int K, int lda, int ldb, int ldc, int flags,
                   const float* bias, const float* mask, int k_split, void* stream);
/* The 512 x 512 layer products of the training path on the feature-sliced kernel (train_lin512.hip, entry point in train_512.hip; what the two calls below use for
 * the forward and data-gradient products of every fc_0 / fc_1 / lin_z layer): Y (M, ldy) [+]= act(X (M, ldx)) op(W) [+ bias] [+ resid],
 * W (512, 512) row-major fp32; transpose = 0: op(W) = W^T (y = x W^T, the nn.Linear forward), 1: op(W) = W (dx = dy W).
 *   flags: 1 relu on X while it is staged, 2 Y += result, 4 (transpose = 0 only) the f16x3 arithmetic of the inference kernels -- two
 *   fp16 planes per operand, three product terms: half the MFMAs, the same accuracy class, |X| < 65504 (the training forward runs its
 *   products this way first and repeats one in bf16x6 when an operand was out of range; this entry does not);
 *   bias (512) / resid (M, ldy) / mask (M, ldy: Y = 0 where mask <= 0) or NULL;
 *   wpack: diner_linear512_pack_bytes() bytes of device scratch (W is packed into three bf16 planes there, then multiplied).
 * Products as accurate as fp32 rounding (six-term split-bf16, no range limits).  ldx, ldy multiples of 4, pointers 16-byte aligned. */
size_t diner_linear512_pack_bytes(void);
int diner_linear512_f32(const float* X, const float* W, float* Y, long long M, int ldx, int ldy, int transpose, int flags,
                        const float* bias, const float* resid, const float* mask, void* wpack, void* stream);
/* The weight / bias gradient of a 512 x 512 layer on the persistent kernel of train_wgrad512.hip (what diner_field_train_backward_f32 uses):
 * dW (512, 512) += dY^T act(X), db (512, or NULL) += column sums of dY, over M >= 1 rows of dY (M, ldy) and X (M, ldx); relu_x != 0 applies
 * relu to X while it is staged.  Both outputs are ACCUMULATED: zero them first.  scratch: NULL (row chunks add into dW with atomics) or
 * diner_wgrad512_scratch_bytes() bytes of device memory (every chunk stores its partial dW there and one pass sums them: what the training
 * step uses; 32 atomics per element cost as much as the products of a 20480-row batch).  ldy even, ldx a multiple of 4, dY 8-byte and
 * X 16-byte aligned.  Six-term split-bf16 products (fp32-class accuracy, no range limits). */
size_t diner_wgrad512_scratch_bytes(void);
int diner_wgrad512_f32(const float* dY, const float* X, float* dW, float* db, long long M, int ldy, int ldx, int relu_x, void* scratch,
                       void* stream);
/* Per-(view, point) MLP inputs of PixelNeRF.forward (pixelnerf.py:91-128) for explicit points xyz / viewdirs (P,3):
 *   feat (NV*P, 64) the 55 encoded inputs zero-padded, tap_row (NV*P, 4) int32 texel rows of the channels-last latent,
 *   tap_w (NV*P, 4) bilinear weights, lat (NV*P, 512) the interpolated latent (SpatialEncoder.index). */
int diner_train_inputs_f32(const DinerScene* scene, const float* xyz, const float* viewdirs, long long P, float freq_factor,
                           float* feat, int* tap_row, float* tap_w, float* lat, void* stream);
/* adjoint of the latent interpolation: d_latent_cl[tap_row][c] += tap_w * d_lat[col][c] (atomic; caller zero-fills) */
int diner_scatter_latent_grad_f32(const float* d_lat, const int* tap_row, const float* tap_w, long long cols,
                                  float* d_latent_cl, void* stream);
/* y (PC) = mean over nv slabs of x (nv, PC) (resnetfc.py:150-152); adjoint != 0: y (nv, PC) = x (PC) / nv */
int diner_view_mean_f32(const float* x, int nv, long long PC, float* y, int adjoint, void* stream);
/* db (N) += column sums of dY (M x N, row stride ld): bias gradients (caller zero-fills) */
int diner_colsum_f32(const float* dY, long long M, int N, int ld, float* db, void* stream);
/* dout == NULL: out (P,4) = [sigmoid(raw[:, :3]), relu(raw[:, 3])] (pixelnerf.py:139-143), raw has row stride ld;
 * dout (P,4) given: out (P, ld) = its adjoint with respect to raw (columns >= 4 zero) */
int diner_field_act_f32(const float* raw, const float* dout, long long P, int ld, float* out, void* stream);
/* adjoint of diner_composite_f32 with respect to the field values (nerf_renderer.py:299-301, :341-360):
 *   g_rgb (NR,3), g_depth (NR) or NULL -> d_field (NR,K,4).  z and rays carry no gradient (the sampler is no_grad). */
int diner_composite_bwd_f32(const float* field, const float* z, const float* rays, int NR, int K, int white_bkgd,
                            const float* g_rgb, const float* g_depth, float* d_field, void* stream);

/* (n, HW, C) channels-last -> (n, C, HW): the latent gradient of diner_field_train_backward_f32 / diner_scatter_latent_grad_f32 in the
 * layout of the encoder's feature map (SpatialEncoder.latent, image_encoder.py:82-95), through LDS tiles. */
int diner_channels_last_to_nchw_f32(const float* src, int n, long long HW, int C, float* dst, void* stream);

/* The whole training forward / backward of the field for one object as one call each (the sequences of the building
 * blocks above that PixelNeRF.forward + ResnetFC.forward and their autograd adjoints amount to; pixelnerf.py:55-145,
 * resnetfc.py:129-159).  `params`: DEVICE parameter tensors in nn.Linear layout (d_in=55, d_latent=d_hidden=512, 5 blocks,
 * combine_layer=3); `workspace`: diner_field_train_workspace_bytes(P, nv) bytes, written by the forward (saved
 * pre-activations, taps, interpolated latent, the 13 weight matrices of the 512 x 512 layers packed in both orientations) and consumed
 * by the backward of the same call pair -- which therefore takes `params` with the VALUES the forward saw (an optimiser step belongs
 * after the backward); it also holds the backward's scratch (partial weight-gradient tiles: 13 x 33.6 MB).
 *   forward : xyz, viewdirs (P,3) -> out (P,4) = [sigmoid rgb, relu sigma]
 *   backward: d_out (P,4) -> `grads` (same structure as `params`, device buffers of the parameters' shapes, overwritten)
 *             and d_latent_cl (nv,Hf,Wf,512) or NULL (gradient of the channels-last feature map, overwritten). */
size_t diner_field_train_workspace_bytes(long long P, int nv);
int diner_field_train_forward_f32(const DinerScene* scene, const DinerMlpParams* params, const float* xyz,
                                  const float* viewdirs, long long P, float* out, void* workspace, void* stream);
int diner_field_train_backward_f32(const DinerScene* scene, const DinerMlpParams* params, const DinerMlpParams* grads,
                                   long long P, const float* d_out, void* workspace, float* d_latent_cl, void* stream);
/* Round 5: the workspace in two parts.  `saved_bytes`: what the forward leaves for the backward (per object: 10.8 GiB of the 15.4 at 4096
 * rays x 40 samples); `scratch_bytes`: the work buffers of either call (the backward's dx / dH / d_lat, partial weight-gradient tiles; the
 * forward's hand-over buffers) -- nothing in them lives from the forward to the backward, so the objects of a step, whose calls run one
 * after the other on a stream, can share ONE scratch buffer.  The _s entry points take the two parts separately (scratch NULL: the work
 * buffers follow the saved part in `workspace`, which then holds diner_field_train_workspace_bytes = saved + scratch bytes). */
int diner_field_train_workspace_split(long long P, int nv, size_t* saved_bytes, size_t* scratch_bytes);
int diner_field_train_forward_s_f32(const DinerScene* scene, const DinerMlpParams* params, const float* xyz, const float* viewdirs,
                                    long long P, float* out, void* workspace, void* scratch, void* stream);
int diner_field_train_backward_s_f32(const DinerScene* scene, const DinerMlpParams* params, const DinerMlpParams* grads, long long P,
                                     const float* d_out, void* workspace, void* scratch, float* d_latent_cl, void* stream);
/* Round 5 (the Python host uses it for objects with at least ~0.7 of a feature map of sample points; DINER_TRAIN_FUSED_FWD=1 / 0 forces it): the training forward on the INFERENCE kernels -- the f16x3
 * per-view and post kernels of diner_field_from_points_f32 in variants that store the pre-activations into the places of `workspace`
 * the layer-wise forward uses (diner_field_train_ws_layout), so that diner_field_train_backward_f32 follows unchanged.  `mlp`: the
 * packed-weights handle of THIS step's parameters.  latent_proj_out (diner_scene_proj_bytes; free again when the call's work is done):
 * the library projects scene->latent_cl through lin_z[0..2] into it on the training products' kernel (f16x3 with its bf16x6 repeat)
 * and gathers from it (round 6: only the texel rows the batch's taps name are projected -- a 64 x 64 ray patch touches 2-3 % of a 400 x 300 map;
 * the buffer must hold FINITE values before its first use (zero it once): a texel that another rounding of a tap's last bit would name is read
 * with a weight of ~1e-7; DINER_TRAIN_PROJ_TOUCHED=0: the whole map); NULL: `scene->latent_proj` prepared with `mlp` (diner_scene_prepare_f32) is used.  The exact repeat is the layer-wise
 * forward, enqueued by this call behind the fused kernels and gated on their range flag (no host synchronisation);
 * diner_field_train_fused_overflowed reads that flag back after a stream wait (a test aid).  DINER_E_UNSUPPORTED (weights outside the fp16
 * split, a projected map of 4 GiB or more): call diner_field_train_forward_f32 (pixelnerf.py:55-145, resnetfc.py:129-159). */
int diner_field_train_forward_fused_f32(const DinerScene* scene, const DinerMlp* mlp, const DinerMlpParams* p, const float* xyz,
                                        const float* viewdirs, long long P, float* out, void* workspace, void* scratch,
                                        float* latent_proj_out, void* stream);
int diner_field_train_fused_overflowed(const void* workspace, long long P, int nv, int* overflowed, void* stream);
/* ABI v6: the SB objects of a training step in ONE call pair (DINER.calc_losses renders SB = 4 objects x 4096 rays per step,
 * diner.py:217-290, configs/train_dtu.yaml:16,52-63; the reference's batched tensors (SB, B, ...) of pixelnerf.py:55-145 amount to this).
 * The ResnetFC layers are scene-independent: only the inputs / gather (forward) and the view-mean adjoint / latent-gradient scatter
 * (backward) are per object.  scenes: HOST array of n_obj scene pointers (same nv; same map sizes not required); xyz / viewdirs
 * (n_obj, P, 3), out / d_out (n_obj, P, 4), object-major.  `saved` / `scratch`: diner_field_train_batch_workspace_split bytes (the layout
 * of diner_field_train_workspace_split for n_obj * P points, rows object-major: diner_field_train_ws_layout(n_obj * P, nv) locates the saved
 * pre-activations).  forward: re-packs the PERSISTENT handle `mlp` from `p` (diner_mlp_update, training subset; no host synchronisation),
 * packs the step's 13 matrices once, then per object projects its latent map into latent_proj_scratch (the largest object's
 * diner_scene_proj_bytes; reused object after object) and runs the fused storing kernels with the gated layer-wise repeat behind them.
 * backward: the 13 x (data gradient, weight gradient) products ONCE over n_obj x P x nv rows -- n_obj times fewer launches, one partial-tile
 * sum, weight gradients of the step summed in-kernel -- then per object the scatter into d_latent_cl[o] (HOST array of n_obj device
 * pointers, entries or the array may be NULL).  map_scratch (or NULL): one map-shaped plane, diner_scene_proj_bytes / 3 bytes of the largest
 * object (the forward's latent_proj_scratch serves) -- with it the adjoint of the three lin_z terms runs in MAP space over the texel rows the
 * batch touches (dWz_b = D_b^T L, d latent = sum_b D_b Wz_b with D_b = the stream's gradient scattered through the bilinear taps) instead of
 * six products over the per-view sample rows; NULL: those products.  DINER_E_UNSUPPORTED as diner_field_train_forward_fused_f32 (before
 * anything is enqueued). */
int diner_field_train_batch_workspace_split(long long P, int nv, int n_obj, size_t* saved_bytes, size_t* scratch_bytes);
int diner_field_train_forward_batch_f32(const DinerScene* const* scenes, int n_obj, DinerMlp* mlp, const DinerMlpParams* p, const float* xyz,
                                        const float* viewdirs, long long P, float* out, void* saved, void* scratch,
                                        float* latent_proj_scratch, void* stream);
int diner_field_train_backward_batch_f32(const DinerScene* const* scenes, int n_obj, const DinerMlpParams* p, const DinerMlpParams* grads,
                                         long long P, const float* d_out, void* saved, void* scratch, float* const* d_latent_cl,
                                         float* map_scratch, void* stream);
/* Test aid: float offsets into the training workspace of the pre-activations the forward saved -- [0..4] X_b, the residual stream
 * entering block b (P*nv rows of 512 for b < 3, P rows behind the view mean), [5..9] H_b, the fc_0 outputs of block b, [10] the
 * stream entering lin_out (P x 512), [11] lin_out's raw outputs (P x 4).  The signs of these values are the relu decisions of the
 * forward (resnetfc.py:61-69): tests evaluate the reference's backward conditioned on them.  n >= 12. */
int diner_field_train_ws_layout(long long P, int nv, long long* float_offsets, int n);

/* ---- ABI v5: generic-shape slow path -------------------------------------------------------------------------------------------
 * ResnetFC.forward (resnetfc.py:129-159) for ANY configuration the reference's constructor accepts (resnetfc.py:72-127: d_hidden
 * default 128, any n_blocks / combine_layer / d_in / d_latent / d_out, Softplus for beta > 0; combine_type "average" is the only one
 * the reference implements, :9-14) and any number of views: the layers chained on the general exact-fp32 MFMA GEMM (diner_gemm_f32 with
 * its exact flag), one launch per layer, the mean over the views at block `combine_layer`.  The fused field kernels remain the path of
 * the shipped configuration; this one exists so that "same constructor kwargs" is "same behaviour" (training through it: the _train_
 * entry points below, ABI v6).  `p` holds DEVICE pointers to the parameters as nn.Linear stores them; no packing, no handle.
 *   zx   (nv, B, d_latent + d_in) row-major, the latent part first (resnetfc.py:140-142)
 *   out  (B, d_out) when 0 <= combine_layer < n_blocks (views averaged inside the network), else (nv, B, d_out)
 *   workspace: diner_mlp_generic_workspace_bytes(p, nv, B) bytes */
size_t diner_mlp_generic_workspace_bytes(const DinerMlpParams* p, int nv, long long B);
int diner_mlp_generic_forward_f32(const DinerMlpParams* p, float beta, const float* zx, int nv, long long B, float* out,
                                  void* workspace, void* stream);
/* ABI v6: training through the generic path (resnetfc.py:72-159 under torch autograd; e.g. ResnetFC's default d_hidden = 128).  The forward
 * keeps the pre-activations in `workspace` (diner_mlp_generic_train_workspace_bytes: (2 n_blocks + 1) saved tensors + three temporaries);
 * the backward chains data / weight / bias gradients, the adjoint of the view mean and -- when d_zx is not NULL -- the gradient with
 * respect to zx on the same exact-fp32 GEMM.  grads: device buffers of the parameters' shapes (overwritten).  n_blocks <= 64. */
size_t diner_mlp_generic_train_workspace_bytes(const DinerMlpParams* p, int nv, long long B);
int diner_mlp_generic_train_forward_f32(const DinerMlpParams* p, float beta, const float* zx, int nv, long long B, float* out,
                                        void* workspace, void* stream);
int diner_mlp_generic_backward_f32(const DinerMlpParams* p, const DinerMlpParams* grads, float beta, const float* zx, int nv, long long B,
                                   const float* d_out, void* workspace, float* d_zx, void* stream);
/* ... and the adjoint of the bilinear latent lookup of diner_field_inputs_generic_f32 (image_encoder.py:97-146 under autograd): d_latent_cl
 * (nv, Hf, Wf, C) channels-last, overwritten, from the first C columns of d_zx (nv, P, d_row); points given as xyz / viewdirs (P, 3). */
int diner_field_inputs_generic_bwd_f32(const DinerScene* scene, const float* xyz, const float* viewdirs, long long P, int d_row,
                                       const float* d_zx, float* d_latent_cl, void* stream);
/* The matrix PixelNeRF.forward hands to its MLP (pixelnerf.py:84-128) for any positional encoding / latent width / NV <= 4:
 * zx (nv, P, C + d_in) with d_in = 4 (2 num_freqs + include_input) + 3, rows [latent (bilinear / border) ; poscode(x_c) ; R d ;
 * poscode(depth_nearest - z_c)].  Point source: (rays (NR,8), z (NR,K), K) with P = NR K, or (xyz, viewdirs) (P,3) with rays == NULL. */
int diner_field_inputs_generic_f32(const DinerScene* scene, const float* rays, const float* z, int K, const float* xyz,
                                   const float* viewdirs, long long P, int num_freqs, int include_input, float freq_factor,
                                   float* zx, void* stream);

/* ---- measurement aid (bench.py): per-kernel durations of the two field kernels ------------------
 * With profiling enabled every field call brackets k_field_pre / k_field_post with HIP events on the
 * launch stream; diner_profile_collect waits for them, returns the summed durations (ms), the number
 * of launches of each kernel and the points they processed, and resets the log. */
int diner_profile_enable(int enable);
int diner_profile_collect(double* pre_ms, double* post_ms, long long* launches, long long* points);

#ifdef __cplusplus
}
#endif
#endif /* DINER_HIP_H */
