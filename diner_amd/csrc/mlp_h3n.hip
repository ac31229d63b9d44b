// f16x3 field kernel, "n-split" variant of the per-view part (experimental, diner_set_precision(2)).
//
// mlp_h3.hip keeps the fp32 design: one wave = 16 (point,view) columns x all 512 features, weights streamed through
// LDS and read by all four waves -- with fp16 MFMAs the matrix pipe is no longer the bound, the LDS fragment reads are
// (each 16x16x32 MFMA consumes a fresh 1 KB weight fragment, four waves read the same ones).  Here the work is split the
// other way: wave w owns output features [128 w, 128 w + 128) for ALL 64 columns of the workgroup (4 views x 16 points).
//   * weights are wave-private: streamed straight global -> VGPR (16 KB per k32 block per wave, software-prefetched),
//     each A fragment feeds 4 column groups x {hi,lo}: 12 MFMAs per (hi, lo) fragment pair instead of 3;
//   * activations are exchanged between layers through a 128 KB LDS buffer already in B-operand form (fp16 hi / lo,
//     1/16 scale folded in): every wave converts its 128-feature slice, two barriers per layer instead of 32;
//   * LDS traffic per MFMA drops 8x, the view mean is a register sum over the four column groups.
// Arithmetic, scaling and results are those of mlp_h3.hip (same products, same accumulation order over k).
#include <vector>
#include "field_common.hpp"

namespace diner {
namespace h3n {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr float kScale = 16.0f, kInvScale = 1.0f / 16.0f;
constexpr int kSlice = 8;                 // accumulator row tiles per wave (128 features)
constexpr int kGroups = 4;                // column groups = source views
constexpr int kBHalfs = 16 * kGroups * 2 * 64 * 8;      // B buffer: [t 16][g 4][hl 2][lane 64] h8 = 128 KB
constexpr size_t kLdsBytes = (size_t)kBHalfs * 2 + 4 * 16 * 32;   // + taps exchange (4 groups x 16 columns x 32 B)

#define DINER_HN_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, ACC, 0, 0, 0)

struct TapRec {            // per column: 4 tap offsets (float4 units into a projected map) + 4 blend weights
  unsigned off[4];
  float w[4];
};

struct Args {
  FieldArgs fa;
  const _Float16* w;       // n-split packed weights: lin_in, then per block b<3: fc_0, fc_1
  const float* b;          // biases x16: lin_in, then per block: fc_0, fc_1  (7 x 512)
};

// packed weights of one layer with KT k32 blocks: [w 4][t KT][mo 8][hl 2][lane 64][8]
__device__ __forceinline__ const h8* wfrag(const _Float16* layer, int KT, int wave, int t, int mo, int hl, int lane) {
  return reinterpret_cast<const h8*>(layer) + ((((size_t)wave * KT + t) * 8 + mo) * 2 + hl) * 64 + lane;
}

// acc[mo][g] += W[slice rows][all k] . B[k][cols g]   (B from the LDS exchange buffer, A straight from global)
template <int KT>
__device__ __forceinline__ void gemm(const _Float16* __restrict__ layer, const h8* __restrict__ B, int wave, int lane,
                                     f32x4 (&acc)[kSlice][kGroups]) {
  h8 a_cur[16], a_nxt[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a_cur[i] = *wfrag(layer, KT, wave, 0, i >> 1, i & 1, lane);
#pragma unroll 1
  for (int t = 0; t < KT; ++t) {
    const int tn = t + 1 < KT ? t + 1 : t;
#pragma unroll
    for (int i = 0; i < 16; ++i) a_nxt[i] = *wfrag(layer, KT, wave, tn, i >> 1, i & 1, lane);
    h8 bh[kGroups], bl[kGroups];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      bh[g] = B[((t * kGroups + g) * 2 + 0) * 64 + lane];
      bl[g] = B[((t * kGroups + g) * 2 + 1) * 64 + lane];
    }
    // three product kinds, accumulators revisited 32 MFMAs apart
#pragma unroll
    for (int g = 0; g < kGroups; ++g)
#pragma unroll
      for (int mo = 0; mo < kSlice; ++mo) DINER_HN_MFMA(acc[mo][g], a_cur[2 * mo], bh[g]);
#pragma unroll
    for (int g = 0; g < kGroups; ++g)
#pragma unroll
      for (int mo = 0; mo < kSlice; ++mo) DINER_HN_MFMA(acc[mo][g], a_cur[2 * mo + 1], bh[g]);
#pragma unroll
    for (int g = 0; g < kGroups; ++g)
#pragma unroll
      for (int mo = 0; mo < kSlice; ++mo) DINER_HN_MFMA(acc[mo][g], a_cur[2 * mo], bl[g]);
#pragma unroll
    for (int i = 0; i < 16; ++i) a_cur[i] = a_nxt[i];
  }
}

__device__ __forceinline__ void split8(const f32x4& lo4, const f32x4& hi4, float scale, h8& h, h8& l) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = fmaxf(j < 4 ? lo4[j] : hi4[j - 4], 0.0f) * scale;
    const _Float16 hh = (_Float16)v;
    h[j] = hh;
    l[j] = (_Float16)(v - (float)hh);
  }
}

// publish relu(acc)/16 of this wave's 128-feature slice as B operands (k32 blocks 4w .. 4w+3) for all 4 column groups
__device__ __forceinline__ void publish(h8* __restrict__ B, int wave, int lane, const f32x4 (&acc)[kSlice][kGroups]) {
#pragma unroll
  for (int tl = 0; tl < 4; ++tl)
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      h8 h, l;
      split8(acc[2 * tl][g], acc[2 * tl + 1][g], kInvScale, h, l);
      const int t = 4 * wave + tl;
      B[((t * kGroups + g) * 2 + 0) * 64 + lane] = h;
      B[((t * kGroups + g) * 2 + 1) * 64 + lane] = l;
    }
}

__device__ __forceinline__ void set_bias(f32x4 (&acc)[kSlice][kGroups], const float* __restrict__ bias, int wave, int q) {
#pragma unroll
  for (int mo = 0; mo < kSlice; ++mo) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 128 * wave + 16 * mo + 4 * q);
#pragma unroll
    for (int g = 0; g < kGroups; ++g) acc[mo][g] = bv;
  }
}
__device__ __forceinline__ void add_bias(f32x4 (&acc)[kSlice][kGroups], const float* __restrict__ bias, int wave, int q) {
#pragma unroll
  for (int mo = 0; mo < kSlice; ++mo) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 128 * wave + 16 * mo + 4 * q);
#pragma unroll
    for (int g = 0; g < kGroups; ++g) acc[mo][g] += bv;
  }
}

// xs[mo][g] += 16 * interp(lin_z[b](latent)) for this wave's feature slice and all four column groups
__device__ __forceinline__ void gather_add(const float* __restrict__ tz, const TapRec (&tp)[kGroups], int wave, int q,
                                           f32x4 (&xs)[kSlice][kGroups]) {
  const f32x4* m4 = reinterpret_cast<const f32x4*>(tz) + 32 * wave + q;       // float4 index of feature 128 w + 4 q
#pragma unroll
  for (int g = 0; g < kGroups; ++g) {
    f32x4 raw[kSlice][4];
#pragma unroll
    for (int mo = 0; mo < kSlice; ++mo)
#pragma unroll
      for (int tap = 0; tap < 4; ++tap) raw[mo][tap] = m4[(size_t)tp[g].off[tap] * 128 + 4 * mo];
#pragma unroll
    for (int mo = 0; mo < kSlice; ++mo) {
      const f32x4 v = raw[mo][0] * tp[g].w[0] + raw[mo][1] * tp[g].w[1] + raw[mo][2] * tp[g].w[2] + raw[mo][3] * tp[g].w[3];
      xs[mo][g] += v * kScale;
    }
  }
}

__global__ __launch_bounds__(256, 1) void k_field_pre_h3n(SceneDev sc, Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  h8* B = reinterpret_cast<h8*>(smem);
  TapRec* taps_lds = reinterpret_cast<TapRec*>(reinterpret_cast<char*>(smem) + (size_t)kBHalfs * 2);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, pt = lane & 15;
  const FieldArgs& fa = a.fa;
  const long long n_tiles = (fa.P + kPtsPerWave - 1) / kPtsPerWave;
  const _Float16* w_in = a.w;                                   // [4][2][8][2][64][8]  = 4 * 2 * 16 KB
  const _Float16* w_blk = a.w + (size_t)4 * 2 * 8192;           // then 6 layers of 4 * 16 * 16 KB
  constexpr size_t kLayerHalfs = (size_t)4 * 16 * 8192;

  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    long long p = tile * kPtsPerWave + pt;
    if (p >= fa.P) p = fa.P - 1;
    Taps taps;
    float feat[16];
    field_frontend(sc, fa, /*view=*/wave, q, p, taps, feat);
    __syncthreads();                              // previous tile's readers of B / taps are done
    {   // publish lin_in B operands (scale 1) for column group `wave` and this column's taps
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        h8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = feat[8 * t + j];
          const _Float16 hh = (_Float16)v;
          h[j] = hh;
          l[j] = (_Float16)(v - (float)hh);
        }
        B[((t * kGroups + wave) * 2 + 0) * 64 + lane] = h;
        B[((t * kGroups + wave) * 2 + 1) * 64 + lane] = l;
      }
      if (q == 0) {
        TapRec r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          r.off[k] = (unsigned)(taps.off[k] >> 9);          // float offset -> units of 512 floats (one texel row)
          r.w[k] = taps.w[k];
        }
        taps_lds[wave * 16 + pt] = r;
      }
    }
    __syncthreads();
    TapRec tp[kGroups];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) tp[g] = taps_lds[g * 16 + pt];

    f32x4 xs[kSlice][kGroups], ns[kSlice][kGroups];
    set_bias(xs, a.b, wave, q);
    gemm<2>(w_in, B, wave, lane, xs);
    for (int b = 0; b < 3; ++b) {
      const float* bias = a.b + kHidden * (1 + 2 * b);
      gather_add(fa.tz + (size_t)b * fa.tz_stride, tp, wave, q, xs);
      __syncthreads();                            // everybody finished reading the previous B
      publish(B, wave, lane, xs);
      __syncthreads();
      set_bias(ns, bias, wave, q);
      gemm<16>(w_blk + (size_t)(2 * b) * kLayerHalfs, B, wave, lane, ns);
      __syncthreads();
      publish(B, wave, lane, ns);
      __syncthreads();
      add_bias(xs, bias + kHidden, wave, q);
      gemm<16>(w_blk + (size_t)(2 * b + 1) * kLayerHalfs, B, wave, lane, xs);
    }
    // view mean = mean over the four column groups; hand-over at scale 1 in accumulator layout (row tile 8 w + mo)
    f32x4* out = reinterpret_cast<f32x4*>(fa.xpre) + (size_t)tile * (kTiles * 64) + lane;
#pragma unroll
    for (int mo = 0; mo < kSlice; ++mo)
      out[(8 * wave + mo) * 64] = (((xs[mo][0] + xs[mo][1]) + xs[mo][2]) + xs[mo][3]) * (0.25f * kInvScale);
  }
}

// layer packing: [w 4][t KT][mo 8][hl 2][lane 64][8]: W[128 w + 16 mo + (lane&15)][32 t + 16 (j>>2) + 4 (lane>>4) + (j&3)] * scale
__global__ void k_pack_layer_h3n(const float* __restrict__ W, int rows, int cols, int KT, float scale,
                                 _Float16* __restrict__ dst) {
  const long long total = (long long)4 * KT * 8192;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63, hl = (i >> 9) & 1, mo = (i >> 10) & 7;
    const int wt = (int)(i >> 13), t = wt % KT, w = wt / KT;
    const int row = 128 * w + 16 * mo + (lane & 15);
    const int col = 32 * t + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
    const float x = (row < rows && col < cols) ? W[(size_t)row * cols + col] * scale : 0.0f;
    const _Float16 h = (_Float16)x;
    dst[i] = hl ? (_Float16)(x - (float)h) : h;
  }
}

}  // namespace h3n

int h3n_pack(const DinerMlpParams* p, hipStream_t stream, float** w_out) {
  using namespace h3n;
  const size_t halfs = (size_t)4 * 2 * 8192 + (size_t)6 * 4 * 16 * 8192;
  DINER_HIP_OK(hipMalloc(w_out, halfs * sizeof(_Float16)));
  _Float16* wp = (_Float16*)*w_out;
  hipLaunchKernelGGL(k_pack_layer_h3n, dim3(256), dim3(256), 0, stream, p->lin_in_w, kHidden, kDIn, 2, kScale, wp);
  wp += (size_t)4 * 2 * 8192;
  for (int b = 0; b < 3; ++b) {
    hipLaunchKernelGGL(k_pack_layer_h3n, dim3(512), dim3(256), 0, stream, p->fc0_w[b], kHidden, kHidden, 16, kScale, wp);
    wp += (size_t)4 * 16 * 8192;
    hipLaunchKernelGGL(k_pack_layer_h3n, dim3(512), dim3(256), 0, stream, p->fc1_w[b], kHidden, kHidden, 16, kScale, wp);
    wp += (size_t)4 * 16 * 8192;
  }
  DINER_LAUNCH_OK();
  return 0;
}
int h3n_set_attributes() {
  DINER_HIP_OK(hipFuncSetAttribute((const void*)h3n::k_field_pre_h3n, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)h3n::kLdsBytes));
  return 0;
}
void h3n_launch_pre(const SceneDev& sc, const FieldArgs& fa, const float* w, const float* b, int grid, hipStream_t stream) {
  h3n::Args a{fa, (const _Float16*)w, b};
  hipLaunchKernelGGL(h3n::k_field_pre_h3n, dim3(grid), dim3(256), h3n::kLdsBytes, stream, sc, a);
}

}  // namespace diner
