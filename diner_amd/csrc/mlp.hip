// Fused per-sample radiance-field evaluation: projection + positional encoding + bilinear feature gather +
// ResnetFC, with every activation resident in MFMA accumulator registers.
//
// Replaces PixelNeRF.forward (reference pixelnerf.py:55-145), PositionalEncoding.forward
// (positional_encoding.py:33-53), SpatialEncoder.index / index_depth (image_encoder.py:97-170) and
// ResnetFC.forward (resnetfc.py:61-69, :129-159).
//
// Design (gfx950):
//  * GEMMs run TRANSPOSED on v_mfma_f32_16x16x4_f32 (exact fp32, 157 TFLOP/s peak): D[feature][point] =
//    sum_k W[feature][k] * act[k][point].  A = weights, B = activations, D = activations of the next
//    layer.  The D register layout of that instruction (lane (q,pt) holds features 4q..4q+3 of a 16-row
//    tile for point pt) is exactly the B layout of the next layer's k-steps once the contraction index is
//    permuted, and the permutation is absorbed by the order in which the weights are packed.  So a
//    wave's 16 (point,view) columns x 512 features = 128 registers never leave the register file between
//    lin_in and the view-mean: no LDS round trip, no HBM round trip for activations.
//  * one wave = one source view x 16 consecutive points; a 256-thread workgroup = the 4 views of those
//    points.  Two 128-register accumulator sets (residual stream x, hidden net) + fragments fit the
//    512-entry unified VGPR/AGPR file at one wave per SIMD.
//  * weights (13.8 MB fp32, L2 / Infinity-Cache resident) are streamed by all four waves through a
//    2 x 32 KB LDS ring with global_load_lds (DMA one stage ahead, one barrier per 32 KB stage, A
//    fragments double-buffered one k-group ahead), packed on the host side of the C ABI into the exact
//    ds_read_b128 fragment order (conflict-free, lane-linear).
//  * the 512-channel latent is gathered straight into B-operand registers from the channels-last map:
//    each lane reads 16 B (4 channels) per tap, the 4 lanes of a point cover one 64 B segment.
//  * lin_z hoist: `lin_z[b]` is linear and bilinear / border interpolation weights sum to one, so
//    lin_z[b](interp(F)) == interp(lin_z[b](F)) (+bias).  The three 512->512 projections of the latent
//    (30 % of the network's FLOPs, resnetfc.py:153-155) are applied ONCE per feature-map pixel by
//    k_hoist_linz when a scene is prepared (0.36 TFLOP at 400x300 vs 326 TFLOP per rendered frame); the
//    per-sample work becomes a gather-add of the projected maps, fused into the B-operand production of
//    the following fc_0.  Results differ from the reference only by fp32 rounding (parity tests).
//  * the view-mean boundary (resnetfc.py:148-151) splits the network into two persistent kernels;
//    the hand-over is 8 KB/point of pre-mean activations stored in accumulator layout (coalesced 1 KB
//    wave stores).
#include <atomic>
#include <mutex>
#include <vector>
#include "field_common.hpp"

namespace diner {

struct DinerMlpImpl {
  float* w_hoist;  // kHoistStages x 8192 floats, stage-tile order: lin_z[0], lin_z[1], lin_z[2]
  float* w_pre;    // kPreStages   x 8192 floats: lin_in, then per block b<3: fc_0, fc_1
  float* w_post;   // kPostStages  x 8192 floats: per block b=3,4: fc_0, fc_1, then lin_out
  float* b_hoist;  // lin_z biases, 3 x 512
  float* b_pre;    // lin_in, then per block b<3: fc_0, fc_1  -> 7 x 512
  float* b_post;   // per block b=3,4: fc_0, fc_1 -> 4 x 512, then lin_out (4, padded to 16)
  // split-precision (f16x3 / f16) copies for the feature-sliced kernels of mlp_h3n.hip: fp16 hi/lo fragments and biases, x16
  float* hn_w;     // n-split packing of lin_in + the 6 per-view layers + the 4 post layers
  float* hn_w_out; // lin_out fragments
  float* hn_b_pre; // 7 x 512, x16
  float* hn_b_post;// 4 x 512 x16, then lin_out bias (scale 1, padded to 16)
  float* wmax_dev; // max |parameter| (device scalar, reduced at pack time)
  float wmax;      // ... read back at the end of diner_mlp_create; after diner_mlp_update: when first asked for (wmax_known)
  int wmax_known;  // 0 after diner_mlp_update (packing enqueued, nothing read back): mlp_wmax() reads it back on demand
  hipStream_t pack_stream;   // the stream the last packing was enqueued on (the lazy read back of wmax waits for it)
  int train_only;  // diner_mlp_update(DINER_MLP_UPDATE_TRAIN_ONLY): only what the fused training forward reads is current
  float freq_factor;   // PositionalEncoding.freq_factor of the inputs this MLP was trained on
  unsigned int* fallback_dev;   // launches recomputed by the gated exact-fp32 pass (device counter)
  uint64_t stamp;      // identity of this handle (DinerScene.proj_stamp)
};

// max |weight| of the handle on the host: known after diner_mlp_create; after diner_mlp_update it is read back (one synchronisation
// of the stream the packing was enqueued on) the first time somebody asks -- the inference entry points choose their kernels by it on the host, the fused training
// forward never asks (it gates on the device scalar)
static std::mutex g_wmax_mu;
static int mlp_wmax(const DinerMlpImpl* m, float* out) {
  std::lock_guard<std::mutex> lock(g_wmax_mu);
  DinerMlpImpl* im = const_cast<DinerMlpImpl*>(m);
  if (!im->wmax_known) {
    DINER_HIP_OK(hipMemcpyAsync(&im->wmax, im->wmax_dev, sizeof(float), hipMemcpyDeviceToHost, im->pack_stream));
    DINER_HIP_OK(hipStreamSynchronize(im->pack_stream));
    im->wmax_known = 1;
  }
  *out = im->wmax;
  return 0;
}
static int mlp_fits(const DinerMlpImpl* m, bool* fits) {
  float w = 0.0f;
  int rc = mlp_wmax(m, &w);
  if (rc) return rc;
  *fits = w == w && w < 1024.0f;
  return 0;
}

// mlp_h3n.hip
int h3n_alloc(float** w, float** w_out, float** b_pre, float** b_post);
int h3n_pack(const DinerMlpParams* p, hipStream_t stream, float* w, float* w_out, float* b_pre, float* b_post, bool train_only);
int h3n_set_attributes();
void h3n_launch_pre(const SceneDev& sc, const FieldArgs& fa, const float* w, const float* b, int grid, bool split,
                    unsigned* tile_counter, hipStream_t stream, const SaveActs* sv = nullptr);
void h3n_launch_post(const PostArgs& pa, const float* w, const float* w_out, int grid, bool split, unsigned* tile_counter,
                     hipStream_t stream, const SaveActs* sv = nullptr);

// ------------------------------------------------------------------------------------------------------
// weight packing (runs once per parameter version, on the device)
//   stage tile layout [mo 8][ml 4][lane 64][4]:  W[128 mg + 16 mo + (lane&15)][64 kc + 16 ml + 4 (lane>>4) + j]
//   stages of a layer are ordered s = 4 kc + mg
// ------------------------------------------------------------------------------------------------------
__global__ void k_pack_layer(const float* __restrict__ W, int rows, int cols, int n_kc, float* __restrict__ dst) {
  // W is (rows, cols) row-major nn.Linear weight; rows padded to 512 and cols to 64*n_kc with zeros
  const int total = n_kc * 4 * kStageFloats;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i & 3, lane = (i >> 2) & 63, ml = (i >> 8) & 3, mo = (i >> 10) & 7, s = i >> 13;
    const int kc = s >> 2, mg = s & 3;
    const int row = 128 * mg + 16 * mo + (lane & 15);
    const int col = 64 * kc + 16 * ml + 4 * (lane >> 4) + j;
    dst[i] = (row < rows && col < cols) ? W[(size_t)row * cols + col] : 0.0f;
  }
}
// lin_out: one stage, layout [m 32][lane 64][4]: Wout[lane&15][16 m + 4 (lane>>4) + j], rows >= d_out are zero
__global__ void k_pack_lin_out(const float* __restrict__ W, int rows, int cols, float* __restrict__ dst) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kStageFloats; i += gridDim.x * blockDim.x) {
    const int j = i & 3, lane = (i >> 2) & 63, m = i >> 8;
    const int row = lane & 15, col = 16 * m + 4 * (lane >> 4) + j;
    dst[i] = (row < rows && col < cols) ? W[(size_t)row * cols + col] : 0.0f;
  }
}
__global__ void k_add_vec(const float* __restrict__ src, int n, float* __restrict__ dst) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = dst[i] + src[i];
}
__global__ void k_copy_pad(const float* __restrict__ src, int n, int n_pad, float* __restrict__ dst) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += gridDim.x * blockDim.x)
    dst[i] = i < n ? src[i] : 0.0f;
}

// ------------------------------------------------------------------------------------------------------
// device building blocks
// ------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) float lds_float;

// LDS-DMA of one 32 KB stage = 32 pieces of 1 KB (one wave-wide global_load_lds_dwordx4 each); wave w moves the
// contiguous pieces 8w..8w+7.  The instruction's immediate offset applies to the global AND the LDS address, so one
// address / M0 pair covers four pieces.  Piece j of a wave is issued from step j of the stage (see stage_compute):
// the DMA is spread over the first half of the stage instead of an 8-instruction burst after the barrier.
// one 32 KB stage = 128 output features (accumulators acc[8 mg .. 8 mg+7]) x 64 k (B operands bop[0..15]).
// The stage is walked in 16 steps of 8 MFMAs: step (ml, mp) multiplies the two A fragments (mo = 2 mp, 2 mp + 1)
// of k-group ml into their two accumulators, alternating between them so that back-to-back MFMAs never hit the
// same accumulator (40-cycle dependent latency vs 32-cycle issue).  Everything that is not an MFMA is spread over
// the steps so that the in-order wave never leaves the matrix pipe idle for long:
//   * the two ds_read_b128 of step s+2 are issued in front of the MFMAs of step s (24 VGPRs of fragments);
//   * one 1 KB LDS-DMA piece of the NEXT stage goes out in each of steps 0..7;
//   * a per-step hook does 1/16 of the VALU work that prepares the B operands of the next 64-feature chunk.
// sched_barrier pins MFMA / LDS / VMEM order between steps; VALU and SALU may still float.
#ifndef DINER_PIN_SCHEDULE
#define DINER_PIN_SCHEDULE 1
#endif
#if DINER_PIN_SCHEDULE
#define DINER_STEP_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define DINER_STEP_FENCE()
#endif

struct NoHook {
  template <int STEP, int PIECE>
  __device__ __forceinline__ void run() {}
};

#define DINER_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, ACC, 0, 0, 0)

// One step = 8 MFMAs on two accumulators, with every other kind of instruction placed in the gaps BETWEEN
// individual MFMAs (an in-order wave can hide about five single-issue instructions under one 32-cycle MFMA; a clump of
// a dozen at a step boundary leaves the matrix pipe idle): two ds_read_b128 for step s+2, one LDS-DMA piece of the
// next stage, and four pieces of the B-operand preparation hook.
template <int MG, int STEP, class Hook>
__device__ __forceinline__ void stage_step(WeightStream& ws, const f32x4* __restrict__ cur, const f32x4 (&f)[2],
                                           f32x4 (&fnext)[2], const float (&bop)[16], f32x4 (&acc)[kTiles],
                                           Hook& hook) {
  constexpr int ml = STEP >> 2, mp = STEP & 3;
  constexpr int a0 = 8 * MG + 2 * mp, a1 = a0 + 1;
  constexpr int S2 = STEP + 2 < 16 ? STEP + 2 : 15;
  constexpr int ml2 = S2 >> 2, mp2 = S2 & 3;
  DINER_MFMA(acc[a0], f[0][0], bop[4 * ml + 0]);
#ifndef DINER_ABL_NO_LDS
  if constexpr (STEP + 2 < 16) fnext[0] = cur[((2 * mp2) * 4 + ml2) * 64];
#endif
  DINER_MFMA(acc[a1], f[1][0], bop[4 * ml + 0]);
  hook.template run<STEP, 0>();
  DINER_MFMA(acc[a0], f[0][1], bop[4 * ml + 1]);
#ifndef DINER_ABL_NO_LDS
  if constexpr (STEP + 2 < 16) fnext[1] = cur[((2 * mp2 + 1) * 4 + ml2) * 64];
#endif
  DINER_MFMA(acc[a1], f[1][1], bop[4 * ml + 1]);
  hook.template run<STEP, 1>();
  DINER_MFMA(acc[a0], f[0][2], bop[4 * ml + 2]);
  ws.template dma_step<STEP>();
  DINER_MFMA(acc[a1], f[1][2], bop[4 * ml + 2]);
  hook.template run<STEP, 2>();
  DINER_MFMA(acc[a0], f[0][3], bop[4 * ml + 3]);
  hook.template run<STEP, 3>();
  DINER_MFMA(acc[a1], f[1][3], bop[4 * ml + 3]);
#if DINER_PIN_SCHEDULE
  // the order the scheduler must realise inside this step: MFMAs with at most a few other instructions between them
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // VALU (hook piece 0)
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // VMEM (LDS-DMA piece / tap load)
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#endif
  DINER_STEP_FENCE();
}

template <int MG, class Hook>
__device__ __forceinline__ void stage_compute(WeightStream& ws, const f32x4* __restrict__ cur, const float (&bop)[16],
                                              f32x4 (&acc)[kTiles], Hook& hook) {
  f32x4 fa[2], fb[2], fc[2];
#ifdef DINER_ABL_NO_LDS
  asm volatile("" : "=v"(fa[0]), "=v"(fa[1]), "=v"(fb[0]), "=v"(fb[1]), "=v"(fc[0]), "=v"(fc[1]));
#else
  fa[0] = cur[(0 * 4 + 0) * 64];
  fa[1] = cur[(1 * 4 + 0) * 64];
  fb[0] = cur[(2 * 4 + 0) * 64];
  fb[1] = cur[(3 * 4 + 0) * 64];
#endif
#define DINER_STEP(S_, FUSE, FLOAD) stage_step<MG, (S_)>(ws, cur, FUSE, FLOAD, bop, acc, hook);
  DINER_STEP(0, fa, fc)  DINER_STEP(1, fb, fa)  DINER_STEP(2, fc, fb)
  DINER_STEP(3, fa, fc)  DINER_STEP(4, fb, fa)  DINER_STEP(5, fc, fb)
  DINER_STEP(6, fa, fc)  DINER_STEP(7, fb, fa)  DINER_STEP(8, fc, fb)
  DINER_STEP(9, fa, fc)  DINER_STEP(10, fb, fa) DINER_STEP(11, fc, fb)
  DINER_STEP(12, fa, fc) DINER_STEP(13, fb, fa) DINER_STEP(14, fc, fb)
  DINER_STEP(15, fa, fc)
#undef DINER_STEP
}

template <int MG>
__device__ __forceinline__ void stage_mma(WeightStream& ws, const float (&bop)[16], f32x4 (&acc)[kTiles]) {
  NoHook h;
  stage_compute<MG>(ws, ws.begin(), bop, acc, h);
}
template <int MG, class Hook>
__device__ __forceinline__ void stage_mma_hook(WeightStream& ws, const float (&bop)[16], f32x4 (&acc)[kTiles],
                                               Hook& hook) {
  stage_compute<MG>(ws, ws.begin(), bop, acc, hook);
}

template <int KC>
__device__ __forceinline__ void bops_relu(const f32x4 (&src)[kTiles], float (&bop)[16]) {
#pragma unroll
  for (int ml = 0; ml < 4; ++ml)
#pragma unroll
    for (int r = 0; r < 4; ++r) bop[4 * ml + r] = fmaxf(src[4 * KC + ml][r], 0.0f);
}

__device__ __forceinline__ void add_bias(f32x4 (&acc)[kTiles], const float* __restrict__ bias, int q) {
#pragma unroll
  for (int mo = 0; mo < kTiles; ++mo) acc[mo] += *reinterpret_cast<const f32x4*>(bias + 16 * mo + 4 * q);
}
__device__ __forceinline__ void set_bias(f32x4 (&acc)[kTiles], const float* __restrict__ bias, int q) {
#pragma unroll
  for (int mo = 0; mo < kTiles; ++mo) acc[mo] = *reinterpret_cast<const f32x4*>(bias + 16 * mo + 4 * q);
}

// hook: B operands of chunk KCN = relu(src[4 KCN ..]) -- one element per step of the preceding stage
template <int KCN>
struct ReluNext {
  const f32x4 (&src)[kTiles];
  float (&bop)[16];
  template <int STEP, int PIECE>
  __device__ __forceinline__ void run() {
    if constexpr (PIECE == 0) bop[STEP] = fmaxf(src[4 * KCN + (STEP >> 2)][STEP & 3], 0.0f);
  }
};

// dst (+)= W . relu(src)  : a full 512x512 layer, 32 stages.  The B operands of chunk kc+1 are produced inside the
// last stage of chunk kc.
__device__ __forceinline__ void layer_from_acc(WeightStream& ws, const f32x4 (&src)[kTiles], f32x4 (&dst)[kTiles]) {
  float bopA[16], bopB[16];
  bops_relu<0>(src, bopA);
#define DINER_KC(KC_, CUR, NXT)                                       \
  {                                                                   \
    stage_mma<0>(ws, CUR, dst);                                       \
    stage_mma<1>(ws, CUR, dst);                                       \
    stage_mma<2>(ws, CUR, dst);                                       \
    if constexpr ((KC_) < 7) {                                        \
      ReluNext<((KC_) < 7 ? (KC_) + 1 : 7)> hk{src, NXT};             \
      stage_mma_hook<3>(ws, CUR, dst, hk);                            \
    } else {                                                          \
      stage_mma<3>(ws, CUR, dst);                                     \
    }                                                                 \
  }
  DINER_KC(0, bopA, bopB) DINER_KC(1, bopB, bopA) DINER_KC(2, bopA, bopB) DINER_KC(3, bopB, bopA)
  DINER_KC(4, bopA, bopB) DINER_KC(5, bopB, bopA) DINER_KC(6, bopA, bopB) DINER_KC(7, bopB, bopA)
#undef DINER_KC
}

// hook: chunk KCN of the hoisted layer: blend the 4 taps of the projected map (step 4 ml), add into the residual
// stream (4 ml + 1), relu -> B operands (4 ml + 2, 4 ml + 3)
template <int KCN>
struct HoistNext {
  const f32x4 (&raw)[16];
  const Taps& t;
  f32x4 (&x)[kTiles];
  float (&bop)[16];
  f32x4 v;
  // one vector component (PIECE) per MFMA gap: blend in step 4 ml, add into x in 4 ml + 1, relu in 4 ml + 2 / + 3
  template <int STEP, int PIECE>
  __device__ __forceinline__ void run() {
    constexpr int ml = STEP >> 2, j = STEP & 3, c = PIECE;
    if constexpr (j == 0)
      v[c] = raw[0 + ml][c] * t.w[0] + raw[4 + ml][c] * t.w[1] + raw[8 + ml][c] * t.w[2] + raw[12 + ml][c] * t.w[3];
    if constexpr (j == 1) x[4 * KCN + ml][c] += v[c];
    if constexpr (j == 2 && c < 2) bop[4 * ml + c] = fmaxf(x[4 * KCN + ml][c], 0.0f);
    if constexpr (j == 3 && c < 2) bop[4 * ml + 2 + c] = fmaxf(x[4 * KCN + ml][2 + c], 0.0f);
  }
};
// hook that only issues the tap loads of the following chunk right after the barrier of a stage (step 0)
template <int KCN>
struct TapsIssue {
  const float* __restrict__ tz;
  const Taps& t;
  int q;
  f32x4 (&raw)[16];
  // 16 float4 loads: one per step of the stage, in its first gap
  template <int STEP, int PIECE>
  __device__ __forceinline__ void run() {
    if constexpr (PIECE == 0) {
      constexpr int tap = STEP >> 2, ml = STEP & 3;
      raw[tap * 4 + ml] = *reinterpret_cast<const f32x4*>(tz + t.off[tap] + 64 * KCN + 16 * ml + 4 * q);
    }
  }
};

// net = fc_0(relu(x + interp(lin_z[b](latent)))) with the gather-add of the hoisted projection fused into the
// B-operand production.  Chunk kc+1's taps are requested in stage 0 of chunk kc (right after its barrier) and are
// blended / added / relu'd step by step inside stage 3 of chunk kc.
__device__ __forceinline__ void layer_fc0_hoisted(WeightStream& ws, const float* __restrict__ tz, const Taps& t, int q,
                                                  f32x4 (&x)[kTiles], f32x4 (&net)[kTiles]) {
  f32x4 raw[16];
  float bopA[16], bopB[16];
  taps_load(tz, t, 0, q, raw);
  {
    HoistNext<0> h0{raw, t, x, bopA};
#define DINER_H0(S_) h0.template run<S_, 0>(); h0.template run<S_, 1>(); h0.template run<S_, 2>(); h0.template run<S_, 3>();
    DINER_H0(0) DINER_H0(1) DINER_H0(2) DINER_H0(3) DINER_H0(4) DINER_H0(5) DINER_H0(6) DINER_H0(7)
    DINER_H0(8) DINER_H0(9) DINER_H0(10) DINER_H0(11) DINER_H0(12) DINER_H0(13) DINER_H0(14) DINER_H0(15)
#undef DINER_H0
  }
#define DINER_KC(KC_, CUR, NXT)                                                    \
  {                                                                                \
    if constexpr ((KC_) < 7) {                                                     \
      TapsIssue<((KC_) < 7 ? (KC_) + 1 : 7)> ti{tz, t, q, raw};                    \
      stage_mma_hook<0>(ws, CUR, net, ti);                                         \
    } else {                                                                       \
      stage_mma<0>(ws, CUR, net);                                                  \
    }                                                                              \
    stage_mma<1>(ws, CUR, net);                                                    \
    stage_mma<2>(ws, CUR, net);                                                    \
    if constexpr ((KC_) < 7) {                                                     \
      HoistNext<((KC_) < 7 ? (KC_) + 1 : 7)> hk{raw, t, x, NXT};                   \
      stage_mma_hook<3>(ws, CUR, net, hk);                                         \
    } else {                                                                       \
      stage_mma<3>(ws, CUR, net);                                                  \
    }                                                                              \
  }
  DINER_KC(0, bopA, bopB) DINER_KC(1, bopB, bopA) DINER_KC(2, bopA, bopB) DINER_KC(3, bopB, bopA)
  DINER_KC(4, bopA, bopB) DINER_KC(5, bopB, bopA) DINER_KC(6, bopA, bopB) DINER_KC(7, bopB, bopA)
#undef DINER_KC
}

// hook: explicit rows (hoist kernel): load the next 64 inputs of the row in step 0, copy them to B operands in step 8+
template <int KCN>
struct RowsNext {
  const float* __restrict__ row;
  int q;
  f32x4 (&raw)[4];
  float (&bop)[16];
  template <int STEP, int PIECE>
  __device__ __forceinline__ void run() {
    if constexpr (STEP < 4 && PIECE == 0)
      raw[STEP] = *reinterpret_cast<const f32x4*>(row + 64 * KCN + 16 * STEP + 4 * q);
  }
};

// dst += W . row   for an explicit 512-float row per lane-column (the hoist kernel): B operands straight from memory
__device__ __forceinline__ void layer_from_rows(WeightStream& ws, const float* __restrict__ row, int q,
                                                f32x4 (&dst)[kTiles]) {
  f32x4 raw[4];
  float bopA[16], bopB[16];
#pragma unroll
  for (int ml = 0; ml < 4; ++ml) raw[ml] = *reinterpret_cast<const f32x4*>(row + 16 * ml + 4 * q);
#pragma unroll
  for (int i = 0; i < 16; ++i) bopA[i] = raw[i >> 2][i & 3];
#define DINER_KC(KC_, CUR, NXT)                                                    \
  {                                                                                \
    if constexpr ((KC_) < 7) {                                                     \
      RowsNext<((KC_) < 7 ? (KC_) + 1 : 7)> rn{row, q, raw, NXT};                  \
      stage_mma_hook<0>(ws, CUR, dst, rn);                                         \
    } else {                                                                       \
      stage_mma<0>(ws, CUR, dst);                                                  \
    }                                                                              \
    stage_mma<1>(ws, CUR, dst);                                                    \
    stage_mma<2>(ws, CUR, dst);                                                    \
    stage_mma<3>(ws, CUR, dst);                                                    \
    if constexpr ((KC_) < 7) {                                                     \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) NXT[i] = raw[i >> 2][i & 3];  \
    }                                                                              \
  }
  DINER_KC(0, bopA, bopB) DINER_KC(1, bopB, bopA) DINER_KC(2, bopA, bopB) DINER_KC(3, bopB, bopA)
  DINER_KC(4, bopA, bopB) DINER_KC(5, bopB, bopA) DINER_KC(6, bopA, bopB) DINER_KC(7, bopB, bopA)
#undef DINER_KC
}

__global__ __launch_bounds__(256, 1) void k_field_pre(SceneDev sc, FieldArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, pt = lane & 15;
  const int v = wave;                                   // one wave per source view
  const long long n_tiles = (a.P + kPtsPerWave - 1) / kPtsPerWave;
  if (a.gate && *a.gate == 0) return;                   // fp32 fall-back pass of an fp16-operand call: nothing overflowed

  WeightStream ws;
  ws.base = a.w_pre;
  ws.lds = smem;
  ws.n_stages = kPreStages;
  ws.wave = wave;
  ws.lane = lane;
  ws.start();

  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    long long p = tile * kPtsPerWave + pt;
    if (p >= a.P) p = a.P - 1;                          // tail lanes shadow the last point; stores are whole tiles
                                                        // into a workspace padded to a multiple of 16 points
    Taps taps;
    float feat[16];
    field_frontend(sc, a, v, q, p, taps, feat);

    f32x4 x[kTiles], net[kTiles];
    // ---- lin_in: x = W_in f + b_in                                             (resnetfc.py:141)
    set_bias(x, a.b_pre, q);
    stage_mma<0>(ws, feat, x);
    stage_mma<1>(ws, feat, x);
    stage_mma<2>(ws, feat, x);
    stage_mma<3>(ws, feat, x);
    // ---- blocks 0..2 (per view)                                                 (:145-157, :61-69)
    for (int b = 0; b < 3; ++b) {
      const float* bias = a.b_pre + kHidden * (1 + 2 * b);
      set_bias(net, bias, q);
      layer_fc0_hoisted(ws, a.tz + (size_t)b * a.tz_stride, taps, q, x, net);   // x += lin_z[b](latent); net = fc_0(relu(x))
      add_bias(x, bias + kHidden, q);
      layer_from_acc(ws, net, x);                                                // x += fc_1(relu(net))
    }
    // ---- mean over the 4 views (= the 4 waves) and hand-over to the second kernel in accumulator layout
    view_mean_store(smem + kRing * kStageFloats, x, 0.25f, reinterpret_cast<f32x4*>(a.xpre) + (size_t)tile * (kTiles * 64),
                    wave, lane);
  }
  ws.drain();   // the last (unused) stage DMAs must land before the LDS is released
}

struct HoistArgs {
  const float* src;    // (rows, 512) channels-last latent pixels (or explicit latent rows)
  float* dst;          // (3, rows, 512): lin_z[b](src) + bias
  long long rows;
  const float* w_hoist;
  const float* b_hoist;
};

// Projects every feature-map pixel through lin_z[0..2] once per scene (see the header comment).
__global__ __launch_bounds__(256, 1) void k_hoist_linz(HoistArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, pt = lane & 15;
  const long long n_tiles = (a.rows + 63) / 64;
  WeightStream ws;
  ws.base = a.w_hoist;
  ws.lds = smem;
  ws.n_stages = kHoistStages;
  ws.wave = wave;
  ws.lane = lane;
  ws.start();
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long row_raw = tile * 64 + wave * 16 + pt;
    const long long row = row_raw < a.rows ? row_raw : a.rows - 1;
    const float* src = a.src + (size_t)row * kLatent;
    for (int b = 0; b < 3; ++b) {
      f32x4 acc[kTiles];
      set_bias(acc, a.b_hoist + kHidden * b, q);
      layer_from_rows(ws, src, q, acc);
      if (row_raw < a.rows) {
        f32x4* out = reinterpret_cast<f32x4*>(a.dst + ((size_t)b * a.rows + row) * kLatent) + q;
#pragma unroll
        for (int mo = 0; mo < kTiles; ++mo) out[mo * 4] = acc[mo];
      }
    }
  }
  ws.drain();
}

__global__ __launch_bounds__(256, 1) void k_field_post(PostArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, pt = lane & 15;
  const long long n_t16 = (a.P + kPtsPerWave - 1) / kPtsPerWave;
  const long long n_tiles = (n_t16 + 3) / 4;           // 4 waves x 16 points per workgroup tile
  if (a.gate && *a.gate == 0) return;
  if (a.gate && a.fallback_count && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.fallback_count, 1u);

  WeightStream ws;
  ws.base = a.w_post;
  ws.lds = smem;
  ws.n_stages = kPostStages;
  ws.wave = wave;
  ws.lane = lane;
  ws.start();

  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    long long t16 = tile * 4 + wave;
    const bool live = t16 < n_t16;
    if (!live) t16 = n_t16 - 1;
    f32x4 x[kTiles], net[kTiles];
    // ---- view-averaged activations written by k_field_pre (resnetfc.py:148-151)
    {
      const f32x4* in = reinterpret_cast<const f32x4*>(a.xpre) + (size_t)t16 * (kTiles * 64) + lane;
#pragma unroll
      for (int mo = 0; mo < kTiles; ++mo) x[mo] = in[mo * 64];
    }
    // ---- blocks 3, 4
    for (int b = 0; b < 2; ++b) {
      const float* bias = a.b_post + 2 * kHidden * b;
      set_bias(net, bias, q);
      layer_from_acc(ws, x, net);
      add_bias(x, bias + kHidden, q);
      layer_from_acc(ws, net, x);
    }
    // ---- lin_out (one stage: 16 padded output rows x 512)                       (resnetfc.py:158)
    {
      const f32x4* st4 = ws.begin();
      stage_prefetch(ws.dma_src, ws.dma_dst, wave, lane);      // DMA of the stage after lin_out (no step loop here)
      f32x4 o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < kTiles; ++m) {
        const f32x4 a4 = st4[m * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          o[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[r], fmaxf(x[m][r], 0.0f), o[r], 0, 0, 0);
      }
      f32x4 res = (o[0] + o[1]) + (o[2] + o[3]);
      res += *reinterpret_cast<const f32x4*>(a.b_post + 4 * kHidden + 4 * q);
      const long long p = t16 * kPtsPerWave + pt;
      if (live && q == 0 && p < a.P) {
        if (!a.raw) {
          res[0] = 1.0f / (1.0f + expf(-res[0]));
          res[1] = 1.0f / (1.0f + expf(-res[1]));
          res[2] = 1.0f / (1.0f + expf(-res[2]));
          res[3] = fmaxf(res[3], 0.0f);
        }
        reinterpret_cast<f32x4*>(a.out)[p] = res;
      }
    }
  }
  ws.drain();
}

// split an explicit (NV, B, 512+55) ResnetFC input into aligned latent rows and 64-padded feature rows
__global__ void k_split_zx(const float* __restrict__ zx, long long rows, float* __restrict__ lat,
                           float* __restrict__ feat) {
  const long long total = rows * (kLatent + kDInPad);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / (kLatent + kDInPad);
    const int c = (int)(i - row * (kLatent + kDInPad));
    if (c < kLatent) lat[row * kLatent + c] = zx[row * (kLatent + kDIn) + c];
    else {
      const int f = c - kLatent;
      feat[row * kDInPad + f] = f < kDIn ? zx[row * (kLatent + kDIn) + kLatent + f] : 0.0f;
    }
  }
}

// max |x| over n floats into *dst (non-negative floats order like their bit patterns)
__global__ void k_absmax(const float* __restrict__ x, long long n, float* __restrict__ dst) {
  float m = 0.0f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = fabsf(x[i]);
    m = (v > m || v != v) ? v : m;                        // a NaN parameter wins (and fails the range check)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float t = __shfl_xor(m, o, 64);
    m = (t > m || t != t) ? t : m;
  }
  if ((threadIdx.x & 63) == 0) {
    if (m != m) atomicExch(reinterpret_cast<unsigned*>(dst), 0x7fc00000u);
    else atomicMax(reinterpret_cast<unsigned*>(dst), __float_as_uint(m));
  }
}

// fp32 projected maps -> fp16 in the plain-fp16 kernel's channel order (mlp_h3n.hip, GatherSideH): position 128 w + 32 mp + 8 q + 4 t + i
// of a texel's 512 halves holds channel 128 w + 16 (2 mp + t) + 4 q + i.  One thread per 8 output halves (two f32x4 reads 64 B apart).
__global__ __launch_bounds__(256) void k_proj_to_f16(const float* __restrict__ src, long long rows, _Float16* __restrict__ dst) {
  typedef _Float16 h8v __attribute__((ext_vector_type(8)));
  const long long n = rows * 64;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i >> 6;
    const int o = (int)(i & 63), w = o >> 4, mp = (o >> 2) & 3, q = o & 3;
    const float* s = src + row * kLatent + 128 * w + 32 * mp + 4 * q;
    const f32x4 a = *reinterpret_cast<const f32x4*>(s), b = *reinterpret_cast<const f32x4*>(s + 16);
    h8v h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h[j] = (_Float16)a[j];
      h[4 + j] = (_Float16)b[j];
    }
    reinterpret_cast<h8v*>(dst)[i] = h;
  }
}

// ---- per-device launch state: function attributes (dynamic LDS above 64 KB) are per device, and so is the CU count ----
constexpr int kMaxDevices = 64;
struct DeviceState {
  std::mutex mu;
  bool attrs[kMaxDevices] = {};
  int cus[kMaxDevices] = {};
};
static DeviceState g_dev;
static constexpr size_t kFp32LdsBytes = (kRing * kStageFloats + kExchFloats) * sizeof(float);

// Returns the CU count of the current device (>0) after making sure the kernels' attributes are set on it; <0 on error.
static int prepare_device() {
  int dev = 0;
  DINER_HIP_OK(hipGetDevice(&dev));
  DINER_CHECK_ARG(dev >= 0 && dev < kMaxDevices, "device index %d outside [0,%d)", dev, kMaxDevices);
  std::lock_guard<std::mutex> lock(g_dev.mu);
  if (!g_dev.attrs[dev]) {
    DINER_HIP_OK(hipFuncSetAttribute((const void*)k_field_pre, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFp32LdsBytes));
    DINER_HIP_OK(hipFuncSetAttribute((const void*)k_field_post, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFp32LdsBytes));
    DINER_HIP_OK(hipFuncSetAttribute((const void*)k_hoist_linz, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFp32LdsBytes));
    int rc = h3n_set_attributes();
    if (rc) return rc;
    hipDeviceProp_t prop;
    DINER_HIP_OK(hipGetDeviceProperties(&prop, dev));
    g_dev.cus[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    g_dev.attrs[dev] = true;
  }
  return g_dev.cus[dev];
}

// ---- optional per-kernel timing (HIP events on the launch stream), used by bench.py for the roofline ----
struct KernelTimer {
  std::mutex mu;
  bool enabled = false;
  std::vector<hipEvent_t> ev;      // triples: before pre, between pre and post, after post
  std::vector<long long> points;
};
static KernelTimer g_timer;

static size_t xpre_bytes(long long P, int /*nv*/) {
  const long long n_t16 = (P + kPtsPerWave - 1) / kPtsPerWave;
  return (size_t)n_t16 * kTiles * 64 * sizeof(f32x4);        // view-averaged hand-over: 2 KB per point
}
constexpr size_t kFlagBytes = 256;                           // overflow flag of the fp16-operand kernels (+ padding)

// precision: DINER_PRECISION_*.  The fp16-operand modes are followed by a GATED pass of the exact-fp32 kernels: the post
// kernel raises a device flag when a raw output is not finite (an activation left the fp16 range, or an input was not
// finite to begin with), and only then do the fp32 kernels (which return at once otherwise) recompute the launch.  No
// host synchronisation, no silent inf/NaN from a checkpoint with large activations.
static int launch_field(const SceneDev* sc, const DinerMlpImpl* m, FieldArgs fa, int nv, float* out, int raw,
                        void* workspace, int precision, hipStream_t stream) {
  DINER_CHECK_ARG(precision == DINER_PRECISION_FP32 || precision == DINER_PRECISION_F16X3 || precision == DINER_PRECISION_F16,
                  "field: precision must be DINER_PRECISION_FP32 (0), _F16X3 (1) or _F16 (3), got %d (2 is retired)", precision);
  const int cus = prepare_device();
  if (cus < 0) return cus;
  // explicit matrices (diner_mlp_forward_f32) and weights outside the fp16 range always take the exact kernels
  DINER_CHECK_ARG(!m->train_only, "field: this handle was last packed with diner_mlp_update(DINER_MLP_UPDATE_TRAIN_ONLY) -- only the "
                  "layouts of the fused training forward are current; update it without the flag (or create one) for inference");
  bool fits = false;
  {
    int rcf = mlp_fits(m, &fits);
    if (rcf) return rcf;
  }
  const bool use_hn = precision != DINER_PRECISION_FP32 && !fa.direct_feat && fits;
  if (use_hn && fa.tz_stride * sizeof(float) >= ((size_t)1 << 32)) {
    set_error("field: one projected feature map is %.1f GiB; the fp16-operand kernels address it with 32-bit offsets "
              "(< 4 GiB) -- use DINER_PRECISION_FP32 for this scene", (double)(fa.tz_stride * sizeof(float)) / (1u << 30));
    return DINER_E_UNSUPPORTED;
  }
  const bool split = precision != DINER_PRECISION_F16;
  if (use_hn && !split && !fa.tz16) {
    set_error("field: DINER_PRECISION_F16 gathers from the fp16 copy of the projected maps -- call diner_scene_prepare_f16 and set "
              "scene->latent_proj_f16 (or use DINER_PRECISION_F16X3 / _FP32)");
    return DINER_E_INVALID;
  }
  fa.w_pre = m->w_pre;
  fa.b_pre = m->b_pre;
  fa.xpre = (float*)workspace;
  fa.freq_factor = m->freq_factor;
  fa.gate = nullptr;
  int* flag = reinterpret_cast<int*>((char*)workspace + xpre_bytes(fa.P, nv));
  const long long n_t16 = (fa.P + kPtsPerWave - 1) / kPtsPerWave;
  SceneDev dummy;
  if (!sc) {
    memset(&dummy, 0, sizeof(dummy));
    dummy.nv = nv;
    sc = &dummy;
  }
  const int grid_pre = (int)(n_t16 < cus ? n_t16 : cus);
  const long long n_tiles = (n_t16 + 3) / 4;
  const int grid_post = (int)(n_tiles < cus ? n_tiles : cus);
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
  bool timed;
  {
    std::lock_guard<std::mutex> lock(g_timer.mu);
    timed = g_timer.enabled;
  }
  if (timed) {
    DINER_HIP_OK(hipEventCreate(&e0));
    DINER_HIP_OK(hipEventCreate(&e1));
    DINER_HIP_OK(hipEventCreate(&e2));
  }
  PostArgs pa{(const float*)workspace, m->w_post, m->b_post, out, fa.P, nv, raw, nullptr, nullptr, m->fallback_dev};
  if (use_hn) {
    DINER_HIP_OK(hipMemsetAsync(flag, 0, 24 * sizeof(int), stream));     // overflow flag + the tile counters (8 queues) of the two kernels
    if (timed) DINER_HIP_OK(hipEventRecord(e0, stream));
    h3n_launch_pre(*sc, fa, m->hn_w, m->hn_b_pre, grid_pre, split, reinterpret_cast<unsigned*>(flag) + 8, stream);
    DINER_LAUNCH_OK();
    if (timed) DINER_HIP_OK(hipEventRecord(e1, stream));
    PostArgs pn = pa;
    pn.b_post = m->hn_b_post;
    pn.overflow = flag;
    h3n_launch_post(pn, m->hn_w, m->hn_w_out, grid_post, split, reinterpret_cast<unsigned*>(flag) + 16, stream);
    DINER_LAUNCH_OK();
    if (timed) DINER_HIP_OK(hipEventRecord(e2, stream));
    fa.gate = flag;                 // the exact kernels below only run when the flag was raised
    pa.gate = flag;
  } else if (timed) {
    DINER_HIP_OK(hipEventRecord(e0, stream));
  }
  hipLaunchKernelGGL(k_field_pre, dim3(grid_pre), dim3(256), kFp32LdsBytes, stream, *sc, fa);
  DINER_LAUNCH_OK();
  if (timed && !use_hn) DINER_HIP_OK(hipEventRecord(e1, stream));
  hipLaunchKernelGGL(k_field_post, dim3(grid_post), dim3(256), kFp32LdsBytes, stream, pa);
  DINER_LAUNCH_OK();
  if (timed) {
    if (!use_hn) DINER_HIP_OK(hipEventRecord(e2, stream));
    std::lock_guard<std::mutex> lock(g_timer.mu);
    g_timer.ev.push_back(e0);
    g_timer.ev.push_back(e1);
    g_timer.ev.push_back(e2);
    g_timer.points.push_back(fa.P);
  }
  return 0;
}

static int launch_hoist(const DinerMlpImpl* m, const float* src, long long rows, float* dst, hipStream_t stream) {
  DINER_CHECK_ARG(!m->train_only, "scene_prepare: this handle was last packed with DINER_MLP_UPDATE_TRAIN_ONLY (its exact-fp32 lin_z pack is stale)");
  const int cus = prepare_device();
  if (cus < 0) return cus;
  HoistArgs ha{src, dst, rows, m->w_hoist, m->b_hoist};
  const long long n_tiles = (rows + 63) / 64;
  hipLaunchKernelGGL(k_hoist_linz, dim3((unsigned)(n_tiles < cus ? n_tiles : cus)), dim3(256), kFp32LdsBytes, stream, ha);
  DINER_LAUNCH_OK();
  return 0;
}

}  // namespace diner

using namespace diner;

struct DinerMlp {
  DinerMlpImpl impl;
};

static int mlp_alloc(DinerMlpImpl& im) {
  DINER_HIP_OK(hipMalloc(&im.w_hoist, (size_t)kHoistStages * kStageFloats * sizeof(float)));
  DINER_HIP_OK(hipMalloc(&im.w_pre, (size_t)kPreStages * kStageFloats * sizeof(float)));
  DINER_HIP_OK(hipMalloc(&im.w_post, (size_t)kPostStages * kStageFloats * sizeof(float)));
  DINER_HIP_OK(hipMalloc(&im.b_hoist, 3 * kHidden * sizeof(float)));
  DINER_HIP_OK(hipMalloc(&im.b_pre, 7 * kHidden * sizeof(float)));
  DINER_HIP_OK(hipMalloc(&im.b_post, (4 * kHidden + 16) * sizeof(float)));
  DINER_HIP_OK(hipMalloc(&im.wmax_dev, sizeof(float)));
  DINER_HIP_OK(hipMalloc(&im.fallback_dev, sizeof(unsigned int)));
  return h3n_alloc(&im.hn_w, &im.hn_w_out, &im.hn_b_pre, &im.hn_b_post);
}

// Packs the parameters into the handle's buffers on `stream` (no allocation, no synchronisation).  train_only: only what the fused
// training forward reads -- the four-wave f16x3 layouts, the lin_out packs, the biases of those kernels, the constants of the projected
// maps (b_hoist) and the weight range; the exact-fp32 stage tiles and the eight-wave layouts keep their old contents.
static int mlp_pack_into(const DinerMlpParams* p, hipStream_t stream, DinerMlpImpl& im, bool train_only) {
  im.pack_stream = stream;
  im.wmax_known = 0;
  DINER_HIP_OK(hipMemsetAsync(im.wmax_dev, 0, sizeof(float), stream));
  auto pack = [&](const float* W, int rows, int cols, int n_kc, float* dst) {
    if (!train_only) hipLaunchKernelGGL(k_pack_layer, dim3(256), dim3(256), 0, stream, W, rows, cols, n_kc, dst);
    hipLaunchKernelGGL(k_absmax, dim3(64), dim3(256), 0, stream, W, (long long)rows * cols, im.wmax_dev);
  };
  auto bias = [&](const float* b, int n, int n_pad, float* dst) {
    hipLaunchKernelGGL(k_copy_pad, dim3(4), dim3(256), 0, stream, b, n, n_pad, dst);   // (biases stay fp32 in every mode)
  };
  auto bias_x = [&](const float* b, int n, int n_pad, float* dst) { if (!train_only) bias(b, n, n_pad, dst); };   // (exact kernels only)
  for (int b = 0; b < 3; ++b) bias(p->lin_z_b[b], kHidden, kHidden, im.b_hoist + kHidden * b);
  float* wp = im.w_pre;
  pack(p->lin_in_w, kHidden, kDIn, 1, wp);
  wp += 4 * kStageFloats;
  bias_x(p->lin_in_b, kHidden, kHidden, im.b_pre);
  for (int b = 0; b < 3; ++b) {
    pack(p->lin_z_w[b], kHidden, kLatent, 8, im.w_hoist + (size_t)b * kStagesPerLayer * kStageFloats);
    pack(p->fc0_w[b], kHidden, kHidden, 8, wp);   wp += kStagesPerLayer * kStageFloats;
    pack(p->fc1_w[b], kHidden, kHidden, 8, wp);   wp += kStagesPerLayer * kStageFloats;
    float* bb = im.b_pre + kHidden * (1 + 2 * b);
    bias_x(p->fc0_b[b], kHidden, kHidden, bb);
    // x_(b+1) = x_b + fc_1(..) + b1[b] + interp(lin_z[b+1](latent) + bz[b+1]): the two constants of blocks 0 and 1 travel together
    // in the projected map of the NEXT block (the interpolation weights sum to one), so the per-view kernels have no separate
    // fc_1 bias pass for those blocks -- a read-modify-write of a whole accumulator block per GEMM in the feature-sliced kernel
    if (b < 2) {
      if (!train_only) DINER_HIP_OK(hipMemsetAsync(bb + kHidden, 0, kHidden * sizeof(float), stream));
      hipLaunchKernelGGL(k_add_vec, dim3(2), dim3(256), 0, stream, p->fc1_b[b], kHidden, im.b_hoist + kHidden * (b + 1));
    } else {
      bias_x(p->fc1_b[b], kHidden, kHidden, bb + kHidden);
    }
  }
  wp = im.w_post;
  for (int b = 3; b < 5; ++b) {
    pack(p->fc0_w[b], kHidden, kHidden, 8, wp); wp += kStagesPerLayer * kStageFloats;
    pack(p->fc1_w[b], kHidden, kHidden, 8, wp); wp += kStagesPerLayer * kStageFloats;
    float* bb = im.b_post + 2 * kHidden * (b - 3);
    bias_x(p->fc0_b[b], kHidden, kHidden, bb);
    bias_x(p->fc1_b[b], kHidden, kHidden, bb + kHidden);
  }
  if (!train_only) hipLaunchKernelGGL(k_pack_lin_out, dim3(32), dim3(256), 0, stream, p->lin_out_w, 4, kHidden, wp);
  hipLaunchKernelGGL(k_absmax, dim3(8), dim3(256), 0, stream, p->lin_out_w, (long long)4 * kHidden, im.wmax_dev);
  bias_x(p->lin_out_b, 4, 16, im.b_post + 4 * kHidden);
  DINER_LAUNCH_OK();
  return h3n_pack(p, stream, im.hn_w, im.hn_w_out, im.hn_b_pre, im.hn_b_post, train_only);
}

static std::atomic<uint64_t> g_next_stamp{1};
extern "C" int diner_mlp_create(const DinerMlpParams* p, void* stream_, DinerMlp** out) {
  DINER_CHECK_ARG(p && out, "mlp_create: null argument");
  int rc = check_mlp_config(p, "mlp_create", /*poscode=*/true);
  if (rc) return rc;
  DinerMlp* m = new DinerMlp();
  memset(&m->impl, 0, sizeof(m->impl));
  m->impl.freq_factor = p->freq_factor;
  m->impl.stamp = g_next_stamp.fetch_add(1);
  rc = mlp_alloc(m->impl);
  if (!rc) DINER_HIP_OK(hipMemsetAsync(m->impl.fallback_dev, 0, sizeof(unsigned int), (hipStream_t)stream_));
  if (!rc) rc = mlp_pack_into(p, (hipStream_t)stream_, m->impl, false);
  // The call returns once packing has completed (the caller may free or overwrite the source tensors) and the weight
  // range is known on the host: one 4-byte read back per diner_mlp_create.
  float w = 0.0f;
  if (!rc) rc = mlp_wmax(&m->impl, &w);
  if (rc) {                       // nothing is left behind by a failed create
    diner_mlp_destroy(m);
    return rc;
  }
  *out = m;
  return 0;
}

// ABI v6: new parameter values into an existing handle -- no allocation, no synchronisation (the training step's packed weights: one
// persistent handle per model, re-packed on the stream every step).  The handle gets a new stamp: maps projected with the old values are refused.
extern "C" int diner_mlp_update(DinerMlp* m, const DinerMlpParams* p, int flags, void* stream_) {
  DINER_CHECK_ARG(m && p, "mlp_update: null argument");
  DINER_CHECK_ARG((flags & ~DINER_MLP_UPDATE_TRAIN_ONLY) == 0, "mlp_update: unknown flags 0x%x", flags);
  int rc = check_mlp_config(p, "mlp_update", /*poscode=*/true);
  if (rc) return rc;
  m->impl.freq_factor = p->freq_factor;
  m->impl.stamp = g_next_stamp.fetch_add(1);
  m->impl.wmax_known = 0;
  m->impl.train_only = (flags & DINER_MLP_UPDATE_TRAIN_ONLY) ? 1 : 0;
  return mlp_pack_into(p, (hipStream_t)stream_, m->impl, m->impl.train_only != 0);
}

extern "C" int diner_mlp_destroy(DinerMlp* m) {
  if (!m) return 0;
  float* bufs[] = {m->impl.w_hoist, m->impl.w_pre, m->impl.w_post, m->impl.b_hoist, m->impl.b_pre, m->impl.b_post,
                   m->impl.hn_w, m->impl.hn_w_out, m->impl.hn_b_pre, m->impl.hn_b_post, m->impl.wmax_dev,
                   (float*)m->impl.fallback_dev};
  for (float* b : bufs)
    if (b) hipFree(b);
  delete m;
  return 0;
}

extern "C" int diner_mlp_weights_fit_f16x3(const DinerMlp* mlp, float* max_abs) {
  DINER_CHECK_ARG(mlp, "mlp_weights_fit_f16x3: null handle");
  float w = 0.0f;
  int rc = mlp_wmax(&mlp->impl, &w);      // (after diner_mlp_update: read back here, behind the packing's stream)
  if (rc) return rc;
  if (max_abs) *max_abs = w;
  return (w == w && w < 1024.0f) ? 1 : 0;
}

extern "C" uint64_t diner_mlp_stamp(const DinerMlp* mlp) { return mlp ? mlp->impl.stamp : 0; }

extern "C" int diner_mlp_fallback_count(const DinerMlp* mlp, long long* launches, int reset, void* stream_) {
  DINER_CHECK_ARG(mlp && launches, "mlp_fallback_count: null argument");
  hipStream_t stream = (hipStream_t)stream_;
  unsigned int n = 0;
  DINER_HIP_OK(hipMemcpyAsync(&n, mlp->impl.fallback_dev, sizeof(n), hipMemcpyDeviceToHost, stream));
  if (reset) DINER_HIP_OK(hipMemsetAsync(mlp->impl.fallback_dev, 0, sizeof(n), stream));
  DINER_HIP_OK(hipStreamSynchronize(stream));
  *launches = (long long)n;
  return 0;
}

extern "C" int diner_profile_enable(int enable) {
  std::lock_guard<std::mutex> lock(g_timer.mu);
  g_timer.enabled = enable != 0;
  return 0;
}

// Sums the recorded kernel durations since the last call (waits for the recorded events), then clears them.
extern "C" int diner_profile_collect(double* pre_ms, double* post_ms, long long* launches, long long* points) {
  std::lock_guard<std::mutex> lock(g_timer.mu);
  double a = 0.0, b = 0.0;
  long long pts = 0;
  const size_t n = g_timer.points.size();
  for (size_t i = 0; i < n; ++i) {
    float t0 = 0.f, t1 = 0.f;
    DINER_HIP_OK(hipEventSynchronize(g_timer.ev[3 * i + 2]));
    DINER_HIP_OK(hipEventElapsedTime(&t0, g_timer.ev[3 * i], g_timer.ev[3 * i + 1]));
    DINER_HIP_OK(hipEventElapsedTime(&t1, g_timer.ev[3 * i + 1], g_timer.ev[3 * i + 2]));
    a += t0;
    b += t1;
    pts += g_timer.points[i];
    for (int k = 0; k < 3; ++k) hipEventDestroy(g_timer.ev[3 * i + k]);
  }
  g_timer.ev.clear();
  g_timer.points.clear();
  if (pre_ms) *pre_ms = a;
  if (post_ms) *post_ms = b;
  if (launches) *launches = (long long)n;
  if (points) *points = pts;
  return 0;
}

extern "C" size_t diner_field_workspace_bytes(long long n_points) {
  if (n_points <= 0) return 0;
  // view-averaged pre-mean activations in accumulator layout (2 KB / point) + the overflow flag
  return xpre_bytes(n_points, kMaxViews) + kFlagBytes;
}

extern "C" size_t diner_mlp_forward_workspace_bytes(long long B) {
  if (B <= 0) return 0;
  // as above plus the aligned split of the explicit zx matrix and the three projected copies of its latent rows
  return xpre_bytes(B, kMaxViews) + kFlagBytes + (size_t)B * kMaxViews * (kLatent + kDInPad + 3 * kLatent) * sizeof(float);
}

static int check_field_scene(const DinerScene* scene, const DinerMlp* mlp, SceneDev* sd, int precision) {
  int rc = make_scene_dev(scene, sd);
  if (rc) return rc;
  DINER_CHECK_ARG(scene->proj_stamp == mlp->impl.stamp,
                  "field: scene->latent_proj was prepared with another packed-weights handle (proj_stamp %llu, this handle %llu): "
                  "the projected maps carry that handle's lin_z / fc_1 biases -- call diner_scene_prepare_f32 with this handle and "
                  "store diner_mlp_stamp() in scene->proj_stamp", (unsigned long long)scene->proj_stamp,
                  (unsigned long long)mlp->impl.stamp);
  DINER_CHECK_ARG(precision != DINER_PRECISION_F16 || !scene->latent_proj_f16 || scene->proj_stamp_f16 == mlp->impl.stamp,
                  "field: scene->latent_proj_f16 was made from maps of another packed-weights handle (proj_stamp_f16 %llu, this handle "
                  "%llu) -- call diner_scene_prepare_f16 after diner_scene_prepare_f32 and store diner_mlp_stamp() in scene->proj_stamp_f16",
                  (unsigned long long)scene->proj_stamp_f16, (unsigned long long)mlp->impl.stamp);
  DINER_CHECK_ARG(scene->nv == kMaxViews, "field: the fused kernel is built for NV=%d source views (got %d)", kMaxViews,
                  scene->nv);
  DINER_CHECK_ARG(scene->C == kLatent, "field: latent size %d != %d", scene->C, kLatent);
  DINER_CHECK_ARG(scene->depth, "field: depth map missing");
  DINER_CHECK_ARG(scene->latent_proj, "field: scene->latent_proj is null -- call diner_scene_prepare_f32 once per "
                                      "(scene, MLP weights) first");
  DINER_CHECK_ARG(scene->Hf > 0 && scene->Wf > 0 && scene->Hs > 0 && scene->Ws > 0, "field: bad map sizes");
  return 0;
}

extern "C" int diner_field_from_rays_f32(const DinerScene* scene, const DinerMlp* mlp, const float* rays,
                                         const float* z, int NR, int K, int precision, float* field_out, void* workspace,
                                         void* stream) {
  DINER_CHECK_ARG(scene && mlp && rays && z && field_out && workspace, "field_from_rays: null pointer argument");
  DINER_CHECK_ARG(NR > 0 && K > 0, "field_from_rays: bad sizes NR=%d K=%d", NR, K);
  SceneDev sd;
  int rc = check_field_scene(scene, mlp, &sd, precision);
  if (rc) return rc;
  FieldArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.rays = rays;
  fa.z = z;
  fa.K = K;
  fa.P = (long long)NR * K;
  fa.tz = scene->latent_proj;
  fa.tz16 = scene->latent_proj_f16;
  fa.tz_stride = (size_t)sd.nv * sd.Hf * sd.Wf * kLatent;
  return launch_field(&sd, &mlp->impl, fa, sd.nv, field_out, 0, workspace, precision, (hipStream_t)stream);
}

extern "C" int diner_field_from_points_f32(const DinerScene* scene, const DinerMlp* mlp, const float* xyz,
                                           const float* viewdirs, long long P, int precision, float* field_out,
                                           void* workspace, void* stream) {
  DINER_CHECK_ARG(scene && mlp && xyz && viewdirs && field_out && workspace, "field_from_points: null pointer argument");
  DINER_CHECK_ARG(P > 0, "field_from_points: P must be positive");
  SceneDev sd;
  int rc = check_field_scene(scene, mlp, &sd, precision);
  if (rc) return rc;
  FieldArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.xyz = xyz;
  fa.viewdirs = viewdirs;
  fa.K = 1;
  fa.P = P;
  fa.tz = scene->latent_proj;
  fa.tz16 = scene->latent_proj_f16;
  fa.tz_stride = (size_t)sd.nv * sd.Hf * sd.Wf * kLatent;
  return launch_field(&sd, &mlp->impl, fa, sd.nv, field_out, 0, workspace, precision, (hipStream_t)stream);
}

// Training forward on the inference kernels (round 5; train.hip, DINER_TRAIN_FUSED_FWD): the f16x3 per-view and post kernels of the
// field path, storing the pre-activations the backward needs into `sv`.  `workspace`: diner_field_workspace_bytes(P) bytes (hand-over +
// flags).  No exact-fp32 repeat behind it: *overflow_flag (device, zeroed here) stays raised when an activation left the fp16 range --
// the caller checks it (the saved activations are then not usable).  Weights outside the fp16 split: DINER_E_UNSUPPORTED (the caller
// keeps the layer-wise forward).
// (3, 512): the constants the projected latent maps carry -- plane b: lin_z[b]'s bias, planes 1 and 2 also fc_1's bias of the block before
// (diner_mlp_create folds them: the interpolation weights sum to one)
const float* mlp_hoist_bias(const DinerMlp* mlp) { return mlp->impl.b_hoist; }

// host-known reasons for which field_forward_save would return DINER_E_UNSUPPORTED, checked before the caller enqueues anything
int field_forward_save_supported(const DinerScene* scene, const DinerMlp* mlp) {
  const DinerMlpImpl* im = &mlp->impl;
  if (im->wmax_known && !(im->wmax == im->wmax && im->wmax < 1024.0f)) {
    set_error("field_forward_save: weights outside the fp16 split (max |w| %g)", (double)im->wmax);
    return DINER_E_UNSUPPORTED;
  }
  if ((size_t)scene->nv * scene->Hf * scene->Wf * kLatent * sizeof(float) >= ((size_t)1 << 32)) {
    set_error("field_forward_save: projected map beyond the 32-bit addressing of the fp16-operand kernels");
    return DINER_E_UNSUPPORTED;
  }
  return 0;
}

__global__ void k_wmax_gate(const float* __restrict__ wmax, int* __restrict__ flag) {
  const float w = *wmax;
  if (!(w == w && w < 1024.0f)) *flag = 1;
}

int field_forward_save(const DinerScene* scene, const DinerMlp* mlp, const float* xyz, const float* viewdirs, long long P, float* out,
                       void* workspace, const SaveActs& sv, int** overflow_flag, hipStream_t stream) {
  SceneDev sd;
  int rc = check_field_scene(scene, mlp, &sd, DINER_PRECISION_F16X3);
  if (rc) return rc;
  const DinerMlpImpl* im = &mlp->impl;
  const int cus = prepare_device();
  if (cus < 0) return cus;
  // weights outside the fp16 split (max |w| >= 1024): known on the host after diner_mlp_create -> DINER_E_UNSUPPORTED (the caller keeps the
  // layer-wise forward); after diner_mlp_update (nothing read back) the launch's flag is raised ON THE DEVICE in front of the kernels --
  // the caller's gated layer-wise repeat then redoes the object, no host synchronisation
  if (im->wmax_known && !(im->wmax == im->wmax && im->wmax < 1024.0f)) {
    set_error("field_forward_save: weights outside the fp16 split (max |w| %g)", (double)im->wmax);
    return DINER_E_UNSUPPORTED;
  }
  FieldArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.xyz = xyz;
  fa.viewdirs = viewdirs;
  fa.K = 1;
  fa.P = P;
  fa.tz = scene->latent_proj;
  fa.tz_stride = (size_t)sd.nv * sd.Hf * sd.Wf * kLatent;
  if (fa.tz_stride * sizeof(float) >= ((size_t)1 << 32)) {
    set_error("field_forward_save: projected map beyond the 32-bit addressing of the fp16-operand kernels");
    return DINER_E_UNSUPPORTED;
  }
  fa.xpre = (float*)workspace;
  fa.freq_factor = im->freq_factor;
  int* flag = reinterpret_cast<int*>((char*)workspace + xpre_bytes(P, sd.nv));
  DINER_HIP_OK(hipMemsetAsync(flag, 0, 24 * sizeof(int), stream));
  if (!im->wmax_known) hipLaunchKernelGGL(k_wmax_gate, dim3(1), dim3(1), 0, stream, im->wmax_dev, flag);
  const long long n_t16 = (P + kPtsPerWave - 1) / kPtsPerWave;
  const int grid_pre = (int)(n_t16 < cus ? n_t16 : cus);
  const long long n_tiles = (n_t16 + 3) / 4;
  const int grid_post = (int)(n_tiles < cus ? n_tiles : cus);
  // DINER_TRAIN_NOSAVE=1 (timing aid, WRONG gradients): the per-view kernel without its stores -- what the saved tensors cost
  static const bool nosave = [] { const char* e = getenv("DINER_TRAIN_NOSAVE"); return e && *e == '1'; }();
  h3n_launch_pre(sd, fa, im->hn_w, im->hn_b_pre, grid_pre, true, reinterpret_cast<unsigned*>(flag) + 8, stream, nosave ? nullptr : &sv);
  DINER_LAUNCH_OK();
  PostArgs pn{(const float*)workspace, im->w_post, im->hn_b_post, out, P, sd.nv, 0, nullptr, flag, im->fallback_dev};
  h3n_launch_post(pn, im->hn_w, im->hn_w_out, grid_post, true, reinterpret_cast<unsigned*>(flag) + 16, stream, &sv);
  DINER_LAUNCH_OK();
  if (overflow_flag) *overflow_flag = flag;
  return 0;
}

extern "C" int diner_mlp_forward_f32(const DinerMlp* mlp, const float* zx, long long B, float* out, void* workspace,
                                     void* stream_) {
  DINER_CHECK_ARG(mlp && zx && out && workspace, "mlp_forward: null pointer argument");
  DINER_CHECK_ARG(B > 0, "mlp_forward: B must be positive");
  hipStream_t stream = (hipStream_t)stream_;
  char* ws = (char*)workspace;
  float* lat = (float*)(ws + xpre_bytes(B, kMaxViews) + kFlagBytes);
  float* feat = lat + (size_t)kMaxViews * B * kLatent;
  float* tz = feat + (size_t)kMaxViews * B * kDInPad;
  const long long rows = (long long)kMaxViews * B;
  hipLaunchKernelGGL(k_split_zx, dim3(2048), dim3(256), 0, stream, zx, rows, lat, feat);
  DINER_LAUNCH_OK();
  int rc = launch_hoist(&mlp->impl, lat, rows, tz, stream);       // lin_z[0..2] of the explicit latent rows
  if (rc) return rc;
  FieldArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.direct_feat = feat;
  fa.tz = tz;
  fa.tz_stride = (size_t)rows * kLatent;
  fa.K = 1;
  fa.P = B;
  return launch_field(nullptr, &mlp->impl, fa, kMaxViews, out, 1, workspace, DINER_PRECISION_FP32, stream);
}

extern "C" size_t diner_scene_proj_bytes(const DinerScene* scene) {
  if (!scene || scene->nv <= 0 || scene->Hf <= 0 || scene->Wf <= 0) return 0;
  return (size_t)3 * scene->nv * scene->Hf * scene->Wf * kLatent * sizeof(float);
}

extern "C" int diner_scene_prepare_f32(const DinerScene* scene, const DinerMlp* mlp, float* latent_proj_out,
                                       void* stream) {
  DINER_CHECK_ARG(scene && mlp && latent_proj_out, "scene_prepare: null pointer argument");
  DINER_CHECK_ARG(scene->latent_cl && scene->C == kLatent && scene->Hf > 0 && scene->Wf > 0 && scene->nv > 0,
                  "scene_prepare: channels-last latent (NV,Hf,Wf,%d) missing", kLatent);
  return launch_hoist(&mlp->impl, scene->latent_cl, (long long)scene->nv * scene->Hf * scene->Wf, latent_proj_out,
                      (hipStream_t)stream);
}

extern "C" size_t diner_scene_proj_f16_bytes(const DinerScene* scene) { return diner_scene_proj_bytes(scene) / 2; }

extern "C" int diner_scene_prepare_f16(const DinerScene* scene, void* latent_proj_f16_out, void* stream) {
  DINER_CHECK_ARG(scene && latent_proj_f16_out, "scene_prepare_f16: null pointer argument");
  DINER_CHECK_ARG(scene->latent_proj && scene->C == kLatent && scene->Hf > 0 && scene->Wf > 0 && scene->nv > 0,
                  "scene_prepare_f16: scene->latent_proj (diner_scene_prepare_f32) missing");
  DINER_CHECK_ARG((reinterpret_cast<size_t>(latent_proj_f16_out) & 15) == 0, "scene_prepare_f16: output must be 16-byte aligned");
  const long long rows = 3ll * scene->nv * scene->Hf * scene->Wf;
  hipLaunchKernelGGL(k_proj_to_f16, dim3(4096), dim3(256), 0, (hipStream_t)stream, scene->latent_proj, rows, (_Float16*)latent_proj_f16_out);
  DINER_LAUNCH_OK();
  return 0;
}

extern "C" int diner_render_f32(const DinerScene* scene, const DinerMlp* mlp, const float* rays, const float* z, int NR,
                                int K, int white_bkgd, int precision, float* rgb_out, float* depth_out, float* weights_out,
                                float* field_ws, void* workspace, void* stream) {
  DINER_CHECK_ARG(field_ws, "render: field scratch missing");
  int rc = diner_field_from_rays_f32(scene, mlp, rays, z, NR, K, precision, field_ws, workspace, stream);
  if (rc) return rc;
  return diner_composite_f32(field_ws, z, rays, NR, K, white_bkgd, rgb_out, depth_out, weights_out, stream);
}
