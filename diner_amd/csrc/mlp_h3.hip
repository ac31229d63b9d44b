// Split-precision variant of the field kernels ("f16x3"): every fp32 GEMM product a*w is evaluated as
//     a_hi*w_hi + a_hi*w_lo + a_lo*w_hi        (a = a_hi + a_lo, w = w_hi + w_lo, all four parts fp16)
// on v_mfma_f32_16x16x32_f16 with fp32 accumulation: three fp16 MFMAs (each 16x faster than the fp32 MFMA per
// flop) replace one fp32 product, the dropped a_lo*w_lo term is ~2^-22 relative.  To keep the low parts of the
// (small) weights out of the fp16 subnormal range the whole network runs at a power-of-two scale: weights, biases
// and the hoisted lin_z maps are multiplied by 16 (exact), accumulators hold 16x the activations, and the
// fp32 -> (hi, lo) conversion of every B operand folds the 1/16 back in (exact).  Same register-resident design
// as mlp.hip (transposed GEMMs, activations in accumulators, weights through an LDS ring); the stage tile holds
// hi and lo fragments of a 128 x 64 weight block in v_mfma_f32_16x16x32_f16 A-operand order.
//
// Selected with diner_set_precision(1); the exact-fp32 kernels of mlp.hip remain the default.
#include <vector>
#include "field_common.hpp"

namespace diner {
namespace h3 {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#ifndef DINER_H3_RING
#define DINER_H3_RING 3
#endif
typedef WeightStreamDeep<DINER_H3_RING> WStream;     // 3 x 32 KB ring: DMA two stages (~2 us) ahead (4 and 5 slots measured no better)

constexpr float kScale = 16.0f, kInvScale = 1.0f / 16.0f;

// B operands of one 64-feature chunk: two k32 blocks, hi and lo parts (8 fp16 per lane each)
struct BOp {
  h8 hi[2], lo[2];
};

// element E (= 4 ml + r, the chunk-local index of accumulator tile ml, register r) -> (k32 block, slot)
template <int E>
__device__ __forceinline__ void split_store(BOp& b, float v) {
  constexpr int ml = E >> 2, tb = ml >> 1, j = 4 * (ml & 1) + (E & 3);
  const _Float16 h = (_Float16)v;
  const _Float16 l = (_Float16)(v - (float)h);
  b.hi[tb][j] = h;
  b.lo[tb][j] = l;
}

#define DINER_H3_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, ACC, 0, 0, 0)

struct NoHook {
  template <int STEP, int PIECE>
  __device__ __forceinline__ void run() {}
};

// stage tile (32 KB): [mo 8][tb 2][hl 2][lane 64] h8 ; fragment (mo, tb, hl) at h8 index ((mo*2+tb)*2+hl)*64 + lane
struct Frag {
  h8 v[4];   // hi(mo0), lo(mo0), hi(mo1), lo(mo1)
};
#ifndef DINER_H3_AHEAD
#define DINER_H3_AHEAD 2
#endif
constexpr int kAhead = DINER_H3_AHEAD;   // fragment reads are issued this many steps (6 MFMAs each) before their use
template <int STEP, int W>
__device__ __forceinline__ h8 frag_addr(const h8* __restrict__ cur) {
  constexpr int tb = STEP >> 2, mp = STEP & 3;
  constexpr int mo = 2 * mp + (W >> 1), hl = W & 1;
#ifdef DINER_ABL_NO_LDS
  h8 r;
  asm volatile("" : "=v"(r));
  return r;
#else
  return cur[((mo * 2 + tb) * 2 + hl) * 64];
#endif
}

// One step = 6 MFMAs on two accumulators (2 row tiles x {hi*hi, lo*hi, hi*lo}); the four fragment reads of step
// s+2, one LDS-DMA piece of the next stage and the hook's VALU go into the gaps.
template <int MG, int STEP, class Hook>
__device__ __forceinline__ void stage_step(WStream& ws, const h8* __restrict__ cur, const Frag& f, Frag& fnext,
                                           const BOp& bop, f32x4 (&acc)[kTiles], Hook& hook) {
  constexpr int tb = STEP >> 2, mp = STEP & 3;
  constexpr int a0 = 8 * MG + 2 * mp, a1 = a0 + 1;
  constexpr int S2 = STEP + kAhead < 8 ? STEP + kAhead : 7;
  DINER_H3_MFMA(acc[a0], f.v[0], bop.hi[tb]);
  if constexpr (STEP + kAhead < 8) fnext.v[0] = frag_addr<S2, 0>(cur);
  hook.template run<STEP, 0>();
  DINER_H3_MFMA(acc[a1], f.v[2], bop.hi[tb]);
  if constexpr (STEP + kAhead < 8) fnext.v[1] = frag_addr<S2, 1>(cur);
  hook.template run<STEP, 1>();
  DINER_H3_MFMA(acc[a0], f.v[1], bop.hi[tb]);
  if constexpr (STEP + kAhead < 8) fnext.v[2] = frag_addr<S2, 2>(cur);
  hook.template run<STEP, 2>();
  DINER_H3_MFMA(acc[a1], f.v[3], bop.hi[tb]);
  if constexpr (STEP + kAhead < 8) fnext.v[3] = frag_addr<S2, 3>(cur);
  hook.template run<STEP, 3>();
  DINER_H3_MFMA(acc[a0], f.v[0], bop.lo[tb]);
  ws.template dma_step<STEP>();
  hook.template run<STEP, 4>();
  DINER_H3_MFMA(acc[a1], f.v[2], bop.lo[tb]);
  hook.template run<STEP, 5>();
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
  __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // VALU
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // VMEM
  __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
  __builtin_amdgcn_sched_barrier(0);
}

template <int MG, class Hook>
__device__ __forceinline__ void stage_compute(WStream& ws, const h8* __restrict__ cur, const BOp& bop,
                                              f32x4 (&acc)[kTiles], Hook& hook) {
  // kAhead + 1 rotating fragment buffers: step s uses buffer s % (kAhead+1) and fills the buffer of step s + kAhead
  Frag fr[kAhead + 1];
#define DINER_FL(S_) fr[S_].v[0] = frag_addr<S_, 0>(cur); fr[S_].v[1] = frag_addr<S_, 1>(cur); \
                     fr[S_].v[2] = frag_addr<S_, 2>(cur); fr[S_].v[3] = frag_addr<S_, 3>(cur);
  DINER_FL(0) DINER_FL(1)
  if constexpr (kAhead > 2) { DINER_FL(2) }
#undef DINER_FL
#define DINER_STEP(S_) stage_step<MG, (S_)>(ws, cur, fr[(S_) % (kAhead + 1)], fr[((S_) + kAhead) % (kAhead + 1)], bop, acc, hook);
  DINER_STEP(0) DINER_STEP(1) DINER_STEP(2) DINER_STEP(3) DINER_STEP(4) DINER_STEP(5) DINER_STEP(6) DINER_STEP(7)
#undef DINER_STEP
}

template <int MG>
__device__ __forceinline__ void stage_mma(WStream& ws, const BOp& bop, f32x4 (&acc)[kTiles]) {
  NoHook h;
  stage_compute<MG>(ws, reinterpret_cast<const h8*>(ws.begin()), bop, acc, h);
}
template <int MG, class Hook>
__device__ __forceinline__ void stage_mma_hook(WStream& ws, const BOp& bop, f32x4 (&acc)[kTiles], Hook& hook) {
  stage_compute<MG>(ws, reinterpret_cast<const h8*>(ws.begin()), bop, acc, hook);
}

__device__ __forceinline__ void add_bias(f32x4 (&acc)[kTiles], const float* __restrict__ bias, int q) {
#pragma unroll
  for (int mo = 0; mo < kTiles; ++mo) acc[mo] += *reinterpret_cast<const f32x4*>(bias + 16 * mo + 4 * q);
}
__device__ __forceinline__ void set_bias(f32x4 (&acc)[kTiles], const float* __restrict__ bias, int q) {
#pragma unroll
  for (int mo = 0; mo < kTiles; ++mo) acc[mo] = *reinterpret_cast<const f32x4*>(bias + 16 * mo + 4 * q);
}

// B operands of chunk KC from 16x-scaled accumulators: relu, undo the scale, split
template <int KC>
__device__ __forceinline__ void bops_relu(const f32x4 (&src)[kTiles], BOp& b) {
#define DINER_E(E_) split_store<E_>(b, fmaxf(src[4 * KC + ((E_) >> 2)][(E_) & 3], 0.0f) * kInvScale);
  DINER_E(0) DINER_E(1) DINER_E(2) DINER_E(3) DINER_E(4) DINER_E(5) DINER_E(6) DINER_E(7)
  DINER_E(8) DINER_E(9) DINER_E(10) DINER_E(11) DINER_E(12) DINER_E(13) DINER_E(14) DINER_E(15)
#undef DINER_E
}

// hook (last stage of a chunk): two elements of the next chunk's B operands per step
template <int KCN>
struct ReluNext {
  const f32x4 (&src)[kTiles];
  BOp& bop;
  template <int STEP, int PIECE>
  __device__ __forceinline__ void run() {
    if constexpr (PIECE == 0 || PIECE == 3) {
      constexpr int e = 2 * STEP + (PIECE == 3 ? 1 : 0);
      split_store<e>(bop, fmaxf(src[4 * KCN + (e >> 2)][e & 3], 0.0f) * kInvScale);
    }
  }
};

__device__ __forceinline__ void layer_from_acc(WStream& ws, const f32x4 (&src)[kTiles], f32x4 (&dst)[kTiles]) {
  BOp bopA, bopB;
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

// hook (last stage of a chunk of the hoisted layer): step 2 ml blends the taps of accumulator tile ml (x16 scale),
// step 2 ml + 1 adds them into the residual stream and splits relu(x)/16 into the next chunk's B operands.
template <int KCN>
struct HoistNext {
  const f32x4 (&raw)[16];
  const Taps& t;
  f32x4 (&x)[kTiles];
  BOp& bop;
  f32x4 v;
  template <int STEP, int PIECE>
  __device__ __forceinline__ void run() {
    constexpr int ml = STEP >> 1, half = STEP & 1;
    if constexpr (PIECE < 4) {
      constexpr int c = PIECE;
      if constexpr (half == 0) {
        v[c] = (raw[0 + ml][c] * t.w[0] + raw[4 + ml][c] * t.w[1] + raw[8 + ml][c] * t.w[2] + raw[12 + ml][c] * t.w[3]) *
               kScale;
      } else {
        x[4 * KCN + ml][c] += v[c];
        split_store<4 * ml + c>(bop, fmaxf(x[4 * KCN + ml][c], 0.0f) * kInvScale);
      }
    }
  }
};
template <int KCN>
struct TapsIssue {
  const float* __restrict__ tz;
  const Taps& t;
  int q;
  f32x4 (&raw)[16];
  template <int STEP, int PIECE>
  __device__ __forceinline__ void run() {
    if constexpr (PIECE == 0 || PIECE == 3) {
      constexpr int i = 2 * STEP + (PIECE == 3 ? 1 : 0), tap = i >> 2, ml = i & 3;
      raw[tap * 4 + ml] = *reinterpret_cast<const f32x4*>(tz + t.off[tap] + 64 * KCN + 16 * ml + 4 * q);
    }
  }
};

__device__ __forceinline__ void layer_fc0_hoisted(WStream& ws, const float* __restrict__ tz, const Taps& t, int q,
                                                  f32x4 (&x)[kTiles], f32x4 (&net)[kTiles]) {
  f32x4 raw[16];
  BOp bopA, bopB;
  taps_load(tz, t, 0, q, raw);
  {
    HoistNext<0> h0{raw, t, x, bopA};
#define DINER_H0(S_) h0.template run<S_, 0>(); h0.template run<S_, 1>(); h0.template run<S_, 2>(); h0.template run<S_, 3>();
    DINER_H0(0) DINER_H0(1) DINER_H0(2) DINER_H0(3) DINER_H0(4) DINER_H0(5) DINER_H0(6) DINER_H0(7)
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

// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void k_field_pre_h3(SceneDev sc, FieldArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, pt = lane & 15;
  const int v = wave;
  const long long n_tiles = (a.P + kPtsPerWave - 1) / kPtsPerWave;
  WStream ws;
  ws.base = a.w_pre;           // fp16 hi/lo stage tiles (same 32 KB stage size)
  ws.lds = smem;
  ws.n_stages = kPreStages;
  ws.wave = wave;
  ws.lane = lane;
  ws.start();
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    long long p = tile * kPtsPerWave + pt;
    if (p >= a.P) p = a.P - 1;
    Taps taps;
    float feat[16];
    field_frontend(sc, a, v, q, p, taps, feat);
    f32x4 x[kTiles], net[kTiles];
    // ---- lin_in (inputs are at scale 1: split without the 1/16)
    {
      BOp bf;
#define DINER_E(E_) split_store<E_>(bf, feat[E_]);
      DINER_E(0) DINER_E(1) DINER_E(2) DINER_E(3) DINER_E(4) DINER_E(5) DINER_E(6) DINER_E(7)
      DINER_E(8) DINER_E(9) DINER_E(10) DINER_E(11) DINER_E(12) DINER_E(13) DINER_E(14) DINER_E(15)
#undef DINER_E
      set_bias(x, a.b_pre, q);
      stage_mma<0>(ws, bf, x);
      stage_mma<1>(ws, bf, x);
      stage_mma<2>(ws, bf, x);
      stage_mma<3>(ws, bf, x);
    }
    for (int b = 0; b < 3; ++b) {
      const float* bias = a.b_pre + kHidden * (1 + 2 * b);
      set_bias(net, bias, q);
      layer_fc0_hoisted(ws, a.tz + (size_t)b * a.tz_stride, taps, q, x, net);
      add_bias(x, bias + kHidden, q);
      layer_from_acc(ws, net, x);
    }
    // view mean + hand-over at scale 1 (0.25 / 16 folded into one exact power-of-two factor)
    view_mean_store(smem + DINER_H3_RING * kStageFloats, x, 0.25f * kInvScale,
                    reinterpret_cast<f32x4*>(a.xpre) + (size_t)tile * (kTiles * 64), wave, lane);
  }
  ws.drain();
}

__global__ __launch_bounds__(256, 1) void k_field_post_h3(PostArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, pt = lane & 15;
  const long long n_t16 = (a.P + kPtsPerWave - 1) / kPtsPerWave;
  const long long n_tiles = (n_t16 + 3) / 4;
  WStream ws;
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
    {
      const f32x4* in = reinterpret_cast<const f32x4*>(a.xpre) + (size_t)t16 * (kTiles * 64) + lane;
#pragma unroll
      for (int mo = 0; mo < kTiles; ++mo) x[mo] = in[mo * 64] * kScale;
    }
    for (int b = 0; b < 2; ++b) {
      const float* bias = a.b_post + 2 * kHidden * b;
      set_bias(net, bias, q);
      layer_from_acc(ws, x, net);
      add_bias(x, bias + kHidden, q);
      layer_from_acc(ws, net, x);
    }
    // ---- lin_out: stage layout [t 16][hl 2][lane 64] h8 (rows >= 4 are zero)
    {
      const h8* st = reinterpret_cast<const h8*>(ws.begin());
      stage_prefetch(ws.dma_src, ws.dma_dst, wave, lane);
      f32x4 o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        h8 bh, bl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float val = fmaxf(x[2 * t + (j >> 2)][j & 3], 0.0f) * kInvScale;
          const _Float16 h = (_Float16)val;
          bh[j] = h;
          bl[j] = (_Float16)(val - (float)h);
        }
        const h8 ah = st[(t * 2 + 0) * 64], al = st[(t * 2 + 1) * 64];
        DINER_H3_MFMA(o[t & 3], ah, bh);
        DINER_H3_MFMA(o[(t + 1) & 3], al, bh);
        DINER_H3_MFMA(o[(t + 2) & 3], ah, bl);
      }
      f32x4 res = ((o[0] + o[1]) + (o[2] + o[3])) * kInvScale;
      res += *reinterpret_cast<const f32x4*>(a.b_post + 4 * kHidden + 4 * q);       // lin_out bias kept at scale 1
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

// ---- packing ------------------------------------------------------------------------------------------------
// stage (32 KB = 16384 halfs): [mo 8][tb 2][hl 2][lane 64][8]; element j of lane (q = lane>>4, i = lane&15):
//   W[128 mg + 16 mo + i][32 (2 kc + tb) + 16 (j>>2) + 4 q + (j&3)] * scale, split into hi / lo
__global__ void k_pack_layer_h3(const float* __restrict__ W, int rows, int cols, int n_kc, float scale,
                                _Float16* __restrict__ dst) {
  const int total = n_kc * 4 * 16384;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63, hl = (i >> 9) & 1, tb = (i >> 10) & 1, mo = (i >> 11) & 7, s = i >> 14;
    const int kc = s >> 2, mg = s & 3;
    const int row = 128 * mg + 16 * mo + (lane & 15);
    const int col = 32 * (2 * kc + tb) + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
    const float w = (row < rows && col < cols) ? W[(size_t)row * cols + col] * scale : 0.0f;
    const _Float16 h = (_Float16)w;
    dst[i] = hl ? (_Float16)(w - (float)h) : h;
  }
}
// lin_out stage: [t 16][hl 2][lane 64][8]: Wout[lane&15][32 t + 16 (j>>2) + 4 (lane>>4) + (j&3)] * scale
__global__ void k_pack_lin_out_h3(const float* __restrict__ W, int rows, int cols, float scale, _Float16* __restrict__ dst) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 16384; i += gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63, hl = (i >> 9) & 1, t = i >> 10;
    const int row = lane & 15, col = 32 * t + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
    const float w = (row < rows && col < cols) ? W[(size_t)row * cols + col] * scale : 0.0f;
    const _Float16 h = (_Float16)w;
    dst[i] = hl ? (_Float16)(w - (float)h) : h;
  }
}
__global__ void k_scale_pad(const float* __restrict__ src, int n, int n_pad, float scale, float* __restrict__ dst) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += gridDim.x * blockDim.x)
    dst[i] = i < n ? src[i] * scale : 0.0f;
}

}  // namespace h3

// ---- host side ------------------------------------------------------------------------------------------------
int h3_pack(const DinerMlpParams* p, hipStream_t stream, float** w_pre, float** w_post, float** b_pre, float** b_post) {
  using namespace h3;
  DINER_HIP_OK(hipMalloc(w_pre, (size_t)kPreStages * kStageFloats * sizeof(float)));
  DINER_HIP_OK(hipMalloc(w_post, (size_t)kPostStages * kStageFloats * sizeof(float)));
  DINER_HIP_OK(hipMalloc(b_pre, 7 * kHidden * sizeof(float)));
  DINER_HIP_OK(hipMalloc(b_post, (4 * kHidden + 16) * sizeof(float)));
  auto pack = [&](const float* W, int rows, int cols, int n_kc, float* dst) {
    hipLaunchKernelGGL(k_pack_layer_h3, dim3(256), dim3(256), 0, stream, W, rows, cols, n_kc, kScale, (_Float16*)dst);
  };
  auto bias = [&](const float* b, int n, int n_pad, float scale, float* dst) {
    hipLaunchKernelGGL(k_scale_pad, dim3(4), dim3(256), 0, stream, b, n, n_pad, scale, dst);
  };
  float* wp = *w_pre;
  pack(p->lin_in_w, kHidden, kDIn, 1, wp);
  wp += 4 * kStageFloats;
  bias(p->lin_in_b, kHidden, kHidden, kScale, *b_pre);
  for (int b = 0; b < 3; ++b) {
    pack(p->fc0_w[b], kHidden, kHidden, 8, wp); wp += kStagesPerLayer * kStageFloats;
    pack(p->fc1_w[b], kHidden, kHidden, 8, wp); wp += kStagesPerLayer * kStageFloats;
    float* bb = *b_pre + kHidden * (1 + 2 * b);
    bias(p->fc0_b[b], kHidden, kHidden, kScale, bb);
    bias(p->fc1_b[b], kHidden, kHidden, kScale, bb + kHidden);
  }
  wp = *w_post;
  for (int b = 3; b < 5; ++b) {
    pack(p->fc0_w[b], kHidden, kHidden, 8, wp); wp += kStagesPerLayer * kStageFloats;
    pack(p->fc1_w[b], kHidden, kHidden, 8, wp); wp += kStagesPerLayer * kStageFloats;
    float* bb = *b_post + 2 * kHidden * (b - 3);
    bias(p->fc0_b[b], kHidden, kHidden, kScale, bb);
    bias(p->fc1_b[b], kHidden, kHidden, kScale, bb + kHidden);
  }
  hipLaunchKernelGGL(k_pack_lin_out_h3, dim3(64), dim3(256), 0, stream, p->lin_out_w, 4, kHidden, kScale, (_Float16*)wp);
  bias(p->lin_out_b, 4, 16, 1.0f, *b_post + 4 * kHidden);
  DINER_LAUNCH_OK();
  return 0;
}

static constexpr size_t kH3LdsBytes = ((size_t)DINER_H3_RING * kStageFloats + kExchFloats) * sizeof(float);
int h3_set_attributes(size_t) {
  const size_t lds_bytes = kH3LdsBytes;
  DINER_HIP_OK(hipFuncSetAttribute((const void*)h3::k_field_pre_h3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  DINER_HIP_OK(hipFuncSetAttribute((const void*)h3::k_field_post_h3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  return 0;
}
void h3_launch_pre(const SceneDev& sc, const FieldArgs& fa, int grid, size_t, hipStream_t stream) {
  hipLaunchKernelGGL(h3::k_field_pre_h3, dim3(grid), dim3(256), kH3LdsBytes, stream, sc, fa);
}
void h3_launch_post(const PostArgs& pa, int grid, size_t, hipStream_t stream) {
  hipLaunchKernelGGL(h3::k_field_post_h3, dim3(grid), dim3(256), kH3LdsBytes, stream, pa);
}

}  // namespace diner
