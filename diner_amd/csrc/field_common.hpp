// Shared between the fp32 field kernels (mlp.hip) and the split-precision variant (mlp_h3.hip): constants, the LDS-DMA
// weight stream, bilinear taps, the per-sample front end and the kernel argument blocks.
#pragma once
#include "common.hpp"

namespace diner {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kHidden = 512;
constexpr int kLatent = 512;
constexpr int kDIn = 55;
constexpr int kDInPad = 64;
constexpr int kTiles = kHidden / 16;          // 32 accumulator tiles of 16 features
constexpr int kStageFloats = 8192;            // 32 KB: 128 output features x 64 k
constexpr int kStagesPerLayer = 32;           // 8 k-chunks x 4 feature groups
constexpr int kHoistStages = 3 * kStagesPerLayer;          // lin_z[0..2]                      =  96
constexpr int kPreStages = 4 + 3 * 2 * kStagesPerLayer;    // lin_in + 3 x (fc_0, fc_1)        = 196
constexpr int kPostStages = 2 * 2 * kStagesPerLayer + 1;   // 2 x (fc_0, fc_1) + lin_out      = 129
constexpr int kPtsPerWave = 16;

// Timing-experiment switches (tools/ablate.sh): DINER_ABL_NO_DMA / _NO_BARRIER / _NO_LDS remove one ingredient of the
// stage loop to price it.  Results are WRONG when any is set; the shipped library is built with none.
template <int J>
__device__ __forceinline__ void stage_dma_piece(const float* __restrict__ gsrc, float* lds_dst, int wave, int lane) {
#ifdef DINER_ABL_NO_DMA
  return;
#endif
  constexpr int h = J >> 2, o = (J & 3) * 1024;
  const __attribute__((address_space(1))) void* g =
      (const __attribute__((address_space(1))) void*)(gsrc + wave * 2048 + lane * 4 + h * 1024);
  __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)(lds_dst + wave * 2048 + h * 1024);
  __builtin_amdgcn_global_load_lds(g, l, 16, o, 0);
}
__device__ __forceinline__ void stage_prefetch(const float* __restrict__ gsrc, float* lds_dst, int wave, int lane) {
  stage_dma_piece<0>(gsrc, lds_dst, wave, lane);
  stage_dma_piece<1>(gsrc, lds_dst, wave, lane);
  stage_dma_piece<2>(gsrc, lds_dst, wave, lane);
  stage_dma_piece<3>(gsrc, lds_dst, wave, lane);
  stage_dma_piece<4>(gsrc, lds_dst, wave, lane);
  stage_dma_piece<5>(gsrc, lds_dst, wave, lane);
  stage_dma_piece<6>(gsrc, lds_dst, wave, lane);
  stage_dma_piece<7>(gsrc, lds_dst, wave, lane);
}

// The weight stream of a persistent workgroup: a fixed cyclic sequence of 32 KB stages flowing through a
// double-buffered LDS ring.  While stage s is consumed, the DMA of stage s+1 is issued (pieces spread over the
// first 8 steps of stage s) and has the rest of the stage (> 2000 matrix-pipe cycles) to land.
constexpr int kRing = 2;

struct WeightStream {
  const float* base;   // packed stages in global memory
  float* lds;          // kRing x kStageFloats
  int n_stages;
  int issue;           // index (in the cyclic sequence) of the stage whose DMA is issued during the current stage
  int slot;            // ring slot of the stage about to be consumed
  int wave, lane;
  const float* dma_src;   // set by begin(): source / destination of the DMA pieces of this stage
  float* dma_dst;

  __device__ __forceinline__ void start() {
    stage_prefetch(base, lds, wave, lane);
    issue = n_stages > 1 ? 1 : 0;
    slot = 0;
  }
  // Begin consuming the stage in `slot`: one barrier per stage publishes it (every wave has waited for its own DMA
  // pieces) and retires the previous stage, whose slot then receives the DMA of the next one.
  __device__ __forceinline__ const f32x4* begin() {
#ifndef DINER_ABL_NO_BARRIER
    __builtin_amdgcn_s_waitcnt(0x0f70);       // vmcnt(0): everything outstanding is at least half a stage old
    __syncthreads();
#endif
    dma_src = base + (size_t)issue * kStageFloats;
    dma_dst = lds + (slot ^ 1) * kStageFloats;
    issue = (issue + 1 == n_stages) ? 0 : issue + 1;
    const f32x4* cur = reinterpret_cast<const f32x4*>(lds + slot * kStageFloats) + lane;
    slot ^= 1;
    return cur;
  }
  template <int STEP>
  __device__ __forceinline__ void dma_step() {
    if constexpr (STEP < 8) stage_dma_piece<STEP>(dma_src, dma_dst, wave, lane);
  }
  __device__ __forceinline__ void drain() { __builtin_amdgcn_s_waitcnt(0x0f70); }
};

// Deeper ring for kernels whose stages are short compared with the L2 / Infinity-Cache latency (the f16x3 kernels: a
// stage lasts ~1 us): RING slots, the DMA of stage k + RING - 1 is issued during stage k.  At the start of stage k only
// the pieces of stage k must have landed; the 8 (RING - 2) younger DMA instructions of this wave (stages k+1 ..) stay
// in flight, hence a COUNTED vmcnt (any other younger VMEM instruction only makes the wait more conservative) and a RAW
// s_barrier (__syncthreads() would drain the LDS-DMA queue).
template <int RING>
struct WeightStreamDeep {
  static constexpr int D = RING - 1;
  const float* base;
  float* lds;
  int n_stages;
  int issue;
  int slot;
  int wave, lane;
  const float* dma_src;
  float* dma_dst;

  __device__ __forceinline__ void start() {
    issue = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      stage_prefetch(base + (size_t)issue * kStageFloats, lds + d * kStageFloats, wave, lane);
      issue = (issue + 1 == n_stages) ? 0 : issue + 1;
    }
    slot = 0;
  }
  __device__ __forceinline__ const f32x4* begin() {
#ifndef DINER_ABL_NO_BARRIER
    constexpr int keep = 8 * (D - 1);
    __builtin_amdgcn_s_waitcnt(0x0f70 | (keep & 15) | ((keep >> 4) << 14));
    __builtin_amdgcn_s_barrier();
#endif
    int sd = slot + D;
    if (sd >= RING) sd -= RING;
    dma_src = base + (size_t)issue * kStageFloats;
    dma_dst = lds + sd * kStageFloats;
    issue = (issue + 1 == n_stages) ? 0 : issue + 1;
    const f32x4* cur = reinterpret_cast<const f32x4*>(lds + slot * kStageFloats) + lane;
    slot = (slot + 1 == RING) ? 0 : slot + 1;
    return cur;
  }
  template <int STEP>
  __device__ __forceinline__ void dma_step() {
    if constexpr (STEP < 8) stage_dma_piece<STEP>(dma_src, dma_dst, wave, lane);
  }
  __device__ __forceinline__ void drain() { __builtin_amdgcn_s_waitcnt(0x0f70); }
};

// View mean at the hand-over between the two field kernels (resnetfc.py:148-151): the 4 waves of a workgroup hold the 4
// source views of the same 16 points, so the mean is an in-LDS exchange -- four rounds of 8 accumulator tiles through a
// 32 KB buffer behind the weight ring.  Wave w sums tiles 2w, 2w+1 of each round over the views ((v0+v1)+v2)+v3, scales
// and stores 2 KB per point in accumulator layout.  Raw barriers + lgkmcnt only: the ring's LDS-DMA stays in flight.
constexpr int kExchFloats = 4 * 8 * 64 * 4;     // 32 KB
__device__ __forceinline__ void view_mean_store(float* exch, const f32x4 (&x)[kTiles], float scale, f32x4* __restrict__ out_tile,
                                                int wave, int lane) {
  f32x4* e = reinterpret_cast<f32x4*>(exch);
#define DINER_ROUND(R_)                                                                              \
  {                                                                                                  \
    __builtin_amdgcn_s_waitcnt(0xc07f);       /* lgkmcnt(0): my reads of the previous round are done */ \
    __builtin_amdgcn_s_barrier();                                                                    \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) e[(wave * 8 + j) * 64 + lane] = x[8 * (R_) + j];   \
    __builtin_amdgcn_s_waitcnt(0xc07f);                                                              \
    __builtin_amdgcn_s_barrier();                                                                    \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                               \
      const int j = 2 * wave + jj;                                                                   \
      f32x4 s = e[(0 * 8 + j) * 64 + lane];                                                          \
      s += e[(1 * 8 + j) * 64 + lane];                                                               \
      s += e[(2 * 8 + j) * 64 + lane];                                                               \
      s += e[(3 * 8 + j) * 64 + lane];                                                               \
      out_tile[(8 * (R_) + j) * 64 + lane] = s * scale;                                              \
    }                                                                                                \
  }
  DINER_ROUND(0) DINER_ROUND(1) DINER_ROUND(2) DINER_ROUND(3)
#undef DINER_ROUND
}

// bilinear taps of one (point, view): float offsets into a channels-last (.., 512) map + blend weights
struct Taps {
  size_t off[4];
  float w[4];
};

__device__ __forceinline__ void taps_load(const float* __restrict__ map, const Taps& t, int kc, int q,
                                          f32x4 (&raw)[16]) {
#pragma unroll
  for (int tap = 0; tap < 4; ++tap)
#pragma unroll
    for (int ml = 0; ml < 4; ++ml)
      raw[tap * 4 + ml] = *reinterpret_cast<const f32x4*>(map + t.off[tap] + 64 * kc + 16 * ml + 4 * q);
}

// MLP input feature f of [x_c(3), 36 sin/cos of x_c, R d (3), dd, 12 sin/cos of dd]  (pixelnerf.py:96-128)
__device__ __forceinline__ float input_feature(int f, const float* xc, const float* vd, float dd, float freq_factor) {
  float arg;
  int j;
  if (f < 3) return f == 0 ? xc[0] : (f == 1 ? xc[1] : xc[2]);
  if (f < 39) {
    j = (f - 3) / 3;
    const int d = (f - 3) - 3 * j;
    arg = d == 0 ? xc[0] : (d == 1 ? xc[1] : xc[2]);
  } else if (f < 42) {
    return f == 39 ? vd[0] : (f == 40 ? vd[1] : vd[2]);
  } else if (f == 42) {
    return dd;
  } else if (f < kDIn) {
    j = f - 43;
    arg = dd;
  } else {
    return 0.0f;
  }
  const float freq = __fmul_rn(freq_factor, (float)(1 << (j >> 1)));            // positional_encoding.py:18
  const float phase = (j & 1) ? 1.57079637050628662109375f : 0.0f;               // fp32(pi/2), :30
  return sin_posenc(__fmaf_rn(arg, freq, phase));                                 // addcmul is fused, :46
}

struct FieldArgs {
  // point source: (rays, z) with K samples per ray, or explicit xyz / viewdirs, or a pre-split zx matrix
  const float* rays;
  const float* z;
  const float* xyz;
  const float* viewdirs;
  const float* direct_feat;     // (NV*P, 64)   explicit MLP inputs (ResnetFC.forward on a matrix); tz rows = (3, NV*P, 512)
  const float* tz;              // hoisted projections: (3, NV, Hf, Wf, 512) of the scene, or (3, NV*P, 512) rows
  size_t tz_stride;             // floats between lin_z[b] and lin_z[b+1] maps
  const void* tz16;             // the same maps in fp16, channels in the plain-fp16 kernel's order (DinerScene.latent_proj_f16), or null
  long long P;
  int K;
  float freq_factor;            // PositionalEncoding.freq_factor of the MLP handle (6.28 in every shipped config)
  const int* gate;              // optional: the kernel returns at once when *gate == 0 (device-side fp32 fall-back, mlp.hip)
  const float* w_pre;
  const float* b_pre;
  float* xpre;                  // (P/16 tiles, NV, 32, 64) f32x4
};

// sample point p of the call: position (and viewing direction) from explicit xyz / viewdirs or from (ray, z)
__device__ __forceinline__ void load_point(const FieldArgs& a, long long p, float& px, float& py, float& pz, float& dx,
                                           float& dy, float& dz) {
  if (a.xyz) {
    px = a.xyz[p * 3 + 0]; py = a.xyz[p * 3 + 1]; pz = a.xyz[p * 3 + 2];
    dx = a.viewdirs[p * 3 + 0]; dy = a.viewdirs[p * 3 + 1]; dz = a.viewdirs[p * 3 + 2];
  } else {
    const long long ray = p / a.K;
    const float* r = a.rays + ray * 8;
    const float zz = a.z[p];
    dx = r[3]; dy = r[4]; dz = r[5];
    px = __fadd_rn(r[0], __fmul_rn(zz, dx));                                 // nerf_renderer.py:304
    py = __fadd_rn(r[1], __fmul_rn(zz, dy));
    pz = __fadd_rn(r[2], __fmul_rn(zz, dz));
  }
}

// bilinear / border taps of view v on the padded feature map at normalised image position (u, w)
// (image_encoder.py:112-123)
__device__ __forceinline__ void bilinear_taps(int Wf, int Hf, float feature_padding, int v, float u, float w, Taps& taps) {
  const float su = __fmul_rn(u, __fdiv_rn(__fsub_rn((float)Wf, __fmul_rn(feature_padding, 2.0f)), (float)Wf));
  const float sv = __fmul_rn(w, __fdiv_rn(__fsub_rn((float)Hf, __fmul_rn(feature_padding, 2.0f)), (float)Hf));
  const float fx = clip_border(unnormalize(su, Wf), Wf), fy = clip_border(unnormalize(sv, Hf), Hf);
  const float x0f = floorf(fx), y0f = floorf(fy);
  const float wx = fx - x0f, wy = fy - y0f;
  const int x0 = (int)x0f, y0 = (int)y0f;
  const int x1 = min(x0 + 1, Wf - 1), y1 = min(y0 + 1, Hf - 1);
  const size_t base = (size_t)v * Hf * Wf;
  taps.off[0] = (base + (size_t)y0 * Wf + x0) * kLatent;
  taps.off[1] = (base + (size_t)y0 * Wf + x1) * kLatent;
  taps.off[2] = (base + (size_t)y1 * Wf + x0) * kLatent;
  taps.off[3] = (base + (size_t)y1 * Wf + x1) * kLatent;
  taps.w[0] = (1.0f - wy) * (1.0f - wx);
  taps.w[1] = (1.0f - wy) * wx;
  taps.w[2] = wy * (1.0f - wx);
  taps.w[3] = wy * wx;
}
__device__ __forceinline__ void bilinear_taps(const SceneDev& sc, int v, float u, float w, Taps& taps) {
  bilinear_taps(sc.Wf, sc.Hf, sc.feature_padding, v, u, w, taps);
}

// Everything per (sample point, view) that precedes the MLP: world->camera transform, projection, nearest depth tap,
// the 55 encoded inputs (as lin_in B operands: feat[4 m + r] = input 16 m + 4 q + r) and the four bilinear taps.
__device__ __forceinline__ void field_frontend(const SceneDev& sc, const FieldArgs& a, int v, int q, long long p,
                                               Taps& taps, float (&feat)[16]) {
  if (a.direct_feat) {
    taps.off[0] = taps.off[1] = taps.off[2] = taps.off[3] = ((size_t)v * a.P + p) * kLatent;
    taps.w[0] = 1.0f;
    taps.w[1] = taps.w[2] = taps.w[3] = 0.0f;
    const float* fr = a.direct_feat + ((size_t)v * a.P + p) * kDInPad;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(fr + 16 * m + 4 * q);
#pragma unroll
      for (int r = 0; r < 4; ++r) feat[4 * m + r] = t4[r];
    }
  } else {
    float px, py, pz, dx, dy, dz;
    load_point(a, p, px, py, pz, dx, dy, dz);
    float xc[3], vd[3];
    world_to_cam(sc.R[v], sc.t[v], px, py, pz, xc[0], xc[1], xc[2]);           // pixelnerf.py:91-93
    vd[0] = rot_row(sc.R[v] + 0, dx, dy, dz);                                  // :100
    vd[1] = rot_row(sc.R[v] + 3, dx, dy, dz);
    vd[2] = rot_row(sc.R[v] + 6, dx, dy, dz);
    const float u = project_axis(xc[0], xc[2], sc.focal[v][0], sc.c[v][0], sc.img_w);   // :105-108
    const float w = project_axis(xc[1], xc[2], sc.focal[v][1], sc.c[v][1], sc.img_h);
    // nearest depth tap -> distance-to-depth code (:114-116)
    const int ix = nearest_border(u, sc.Ws), iy = nearest_border(w, sc.Hs);
    const float dd = __fsub_rn(sc.depth[(size_t)v * sc.Hs * sc.Ws + (size_t)iy * sc.Ws + ix], xc[2]);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) feat[4 * m + r] = input_feature(16 * m + 4 * q + r, xc, vd, dd, a.freq_factor);
    bilinear_taps(sc, v, u, w, taps);
  }

}

// Training forward on the inference kernels (round 5, k_train_fwd_pre / k_train_fwd_post): where the pre-activations the backward needs are
// stored -- fp32, row-major (rows, 512), values at scale 1.  X[b]: the residual stream entering block b, H[b]: fc_0's output of block b;
// b < 3: rows v P + p (per view), b >= 3 and x_last (the stream entering lin_out): rows p; raw: lin_out's outputs (P, 4).
struct SaveActs {
  float* X[5];
  float* H[5];
  float* x_last;
  float* raw;
  unsigned* bX[5];     // the relu decisions of X[b] / H[b] as bits, 16 dwords per row (train_lin512.hpp, Lin512Args.maskbits)
  unsigned* bH[5];
};
struct PostArgs {
  const float* xpre;
  const float* w_post;
  const float* b_post;
  float* out;          // (P, 4)
  long long P;
  int nv;
  int raw;             // 1: ResnetFC.forward output; 0: sigmoid(rgb), relu(sigma) (pixelnerf.py:139-143)
  const int* gate;     // optional: return at once when *gate == 0
  int* overflow;       // optional (fp16-operand kernels): set to 1 when a raw lin_out value is not finite
  unsigned int* fallback_count;   // optional (gated exact pass): +1 per launch that actually recomputes
};

}  // namespace diner
