// Training path, the layer-sized products with a 512 x 512 weight matrix (forward y = act(x) W^T + b and the data gradient
// dx = dy W of every ResnetFC layer, resnetfc.py:61-69 / :129-159 under torch autograd in DINER.calc_losses, diner.py:217-290):
// the inference kernels' decomposition with the training path's arithmetic.
//
//   * arithmetic: "bf16x6" as in train.hip -- every fp32 operand is split into three bf16 terms (8 + 8 + 8 mantissa bits, fp32's
//     exponent range: no scaling, no range limit) and the six products above 2^-24 are accumulated in fp32 on
//     v_mfma_f32_32x32x16_bf16;
//   * decomposition: D[feature][row] = sum_k W[feature][k] x[row][k].  A persistent workgroup (one per CU, one wave per SIMD) takes
//     tiles of 64 rows; wave w owns output features [128 w, 128 w + 128) of all 64 rows = 4 x 2 MFMA tiles = 128 accumulator
//     registers.  The weights (A operand) are wave-private: packed once per parameter version into three bf16 planes in the wave's
//     consumption order and streamed global -> VGPR through a register ring, 12 KB per k16 step (every fragment feeds 2 x {1..3}
//     MFMAs of 32 clocks: the 64 B/clk vector-memory path is ~half busy);
//   * the activations (B operand) are staged by the four waves together: each converts its share of the next 64 x 128 slab
//     (fp32 -> three bf16 planes, optional relu) as a side task of the current slab's 384 MFMAs and writes it to LDS in B-fragment
//     order; two 48 KB slab buffers, one barrier per slab;
//   * epilogue straight from the accumulators: bias, residual, relu mask of the saved pre-activation, "+=".
// The general kernel of train.hip (128 x 128 x 32 tiles, operands split while staged, two barriers per k-tile) stays for the
// weight gradients and the ragged / skinny products; on a 327680 x 512 x 512 product it reaches 112-137 TFLOP/s fp32-equivalent.
#include <atomic>
#include <utility>
#include "field_common.hpp"
#include "train_lin512.hpp"
// (compiled as part of train_512.hip, which holds the kernel entry points and the launchers)

namespace diner {
namespace train {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 hf8 __attribute__((ext_vector_type(8)));
constexpr float kF16Scale = 16.0f, kF16InvScale = 1.0f / 16.0f;      // f16x3 arithmetic: the packed weights carry this factor

constexpr int kStepsPerSlab = 8;                // k16 steps per staged slab (128 contraction indices)
constexpr int kSlabs = 512 / (16 * kStepsPerSlab);
#ifndef DINER_L512_STEP_PAD       // bytes between the k16 steps of a slab buffer beyond their fragments, see lin512_body (staging writes)
#define DINER_L512_STEP_PAD 32
#endif
constexpr int kStepPad = DINER_L512_STEP_PAD;
constexpr size_t kLdsBytes512 = (size_t)2 * kStepsPerSlab * (2 * 3 * 1024 + kStepPad);  // two slab buffers of [step 8][row half CT][plane 3] 1 KB fragments: 96 KB at CT = 2
constexpr size_t kLdsBytes512F16W2 = (size_t)2 * kStepsPerSlab * (2 * 2 * 1024 + kStepPad);  // f16x3 at CT = 2: 66 KB (two workgroups per CU, experiment)
constexpr size_t kLdsBytes512F16 = (size_t)2 * kStepsPerSlab * (4 * 2 * 1024 + kStepPad);  // f16x3 (two planes) at CT = 4 (128-row tiles): 128 KB

// fp32 -> three bf16 planes (round to nearest each time; the residuals are exact in fp32), 8 values at once
__device__ __forceinline__ void split3x8(const float (&v)[8], bf8& p0, bf8& p1, bf8& p2) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 a0 = (__bf16)v[j];
    const float r1 = v[j] - (float)a0;
    const __bf16 a1 = (__bf16)r1;
    const float r2 = r1 - (float)a1;
    p0[j] = a0;
    p1[j] = a1;
    p2[j] = (__bf16)r2;
  }
}

// f16x3 staging in single full-rate instructions (the idiom of the inference kernels' operand conversion, mlp_h3n.hip): two values ->
// one packed fp16 word, the residuals x - float(half) by v_fma_mix_f32 reading the half in place
__device__ __forceinline__ unsigned cvt_pk_f16_w(float a, float b) {
  unsigned d;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float resid_lo_w(unsigned h, float v) {       // v - float(low half of h)
  float d;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(v));
  return d;
}
__device__ __forceinline__ float resid_hi_w(unsigned h, float v) {       // v - float(high half of h)
  float d;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(v));
  return d;
}

template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl(f, std::make_integer_sequence<int, N>{});
}

// Weight packing: dst[(((w * 32 + s) * 4 + rt) * 3 + pl) * 64 + lane][j] = plane pl of
//   W[128 w + 32 rt + (lane & 31)][16 s + 8 (lane >> 5) + j]      (transpose = 0: forward, W is (out, in) as nn.Linear stores it)
//   W[16 s + 8 (lane >> 5) + j][128 w + 32 rt + (lane & 31)]      (transpose = 1: data gradient, the roles of out / in swap)
// mode 0 / 1: three bf16 planes, forward / transposed; mode 2 / 3 (forward / transposed, the f16x3 arithmetic of lin512_body<.., AR = 1>):
// two fp16 planes hi / lo of 16 W -- dst[(((w * 32 + s) * 4 + rt) * 2 + pl) * 64 + lane][j] (the factor keeps the lo parts of small weights
// normal; the body's epilogue takes it out again).  wbad: raised when a hi part is not finite (16 |w| >= 65520 or NaN)
__device__ __forceinline__ void pack_w512(const float* __restrict__ W, int mode, __bf16* __restrict__ dst, int* __restrict__ wbad = nullptr) {
  const int total = 4 * 32 * 4 * 64;              // (w, s, rt, lane) slots of 8 values x 3 (2) planes
  const bool transpose = mode == 1 || mode == 3;
  bool bad = false;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int lane = i & 63, rt = (i >> 6) & 3, s = (i >> 8) & 31, w = i >> 13;
    const int f = 128 * w + 32 * rt + (lane & 31), k0 = 16 * s + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = transpose ? W[(size_t)(k0 + j) * 512 + f] : W[(size_t)f * 512 + k0 + j];
    if (mode >= 2) {
      hf8 hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = v[j] * kF16Scale;
        hi[j] = (_Float16)x;
        lo[j] = (_Float16)(x - (float)hi[j]);
        bad |= !(fabsf((float)hi[j]) <= 65504.0f);
      }
      hf8* d = reinterpret_cast<hf8*>(dst) + ((((size_t)w * 32 + s) * 4 + rt) * 2) * 64 + lane;
      d[0] = hi;
      d[64] = lo;
      continue;
    }
    bf8 p0, p1, p2;
    split3x8(v, p0, p1, p2);
    bf8* d = reinterpret_cast<bf8*>(dst) + ((((size_t)w * 32 + s) * 4 + rt) * 3) * 64 + lane;
    d[0] = p0;
    d[64] = p1;
    d[128] = p2;
  }
  if (bad && wbad) *wbad = 1;
}
__global__ void k_pack_w512(const float* __restrict__ W, int transpose, __bf16* __restrict__ dst) { pack_w512(W, transpose, dst); }
// blockIdx.y = matrix, blockIdx.z = pack mode (0 forward, 1 transposed, 2 / 3 the same in fp16 hi / lo): all 512 x 512 weights of a training
// step in one launch
__global__ void k_pack_w512_many(PackMany w, char* __restrict__ base, int* __restrict__ wbad) {
  pack_w512(w.W[blockIdx.y], blockIdx.z, reinterpret_cast<__bf16*>(base + (size_t)(13 * blockIdx.z + blockIdx.y) * kL512PackBytes), wbad);
}

#ifndef DINER_L512_RING
#define DINER_L512_RING 2
#endif
#ifndef DINER_L512_EPI_PD      // groups of tensor-term requests in flight in the epilogue of the 128-row shape (1 = round 6's first version)
#define DINER_L512_EPI_PD 1
#endif
// -DDINER_L512_PROF (measurement build, tools/prof_l512.sh): shader clocks per wave summed over the launches since the last read --
// [0] whole tile loop, [1] MFMA slab loops without their barriers, [2] slab barriers, [3] epilogues, [4] tiles, [5] waves, [6] prologue
#ifdef DINER_L512_PROF
__device__ unsigned long long g_l512_prof[8];
#define L512_CLK() __builtin_readcyclecounter()
#else
#define L512_CLK() 0ull
#endif
#define DINER_BF16_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, ACC, 0, 0, 0)
#define DINER_F16_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf8, A), __builtin_bit_cast(hf8, B), ACC, 0, 0, 0)

// R: weight ring depth in k16 steps (kStepsPerSlab must be a multiple of it); CT: 32-row halves per tile -- 2 = 64-row tiles (the
// throughput shape), 1 = 32-row tiles for small M (the reference training batch has 20480 rows: 320 tiles of 64 on 256 CUs is two rounds
// for 1.25 rounds of work; 640 tiles of 32 balance better although a tile then feeds each weight fragment half as many MFMAs)
// FH: feature halves -- 1: a workgroup computes all 512 features of its rows; 2: workgroup pairs share a tile, each computes 256 features
// (wave w: 64 features = 2 MFMA row tiles) -- half-cost units for the ragged last round of a small M (lin512_launch)
// (bid, nblk: the workgroup's index and count within its launch part -- k_lin512_plan runs two shapes in one launch)
// AR: arithmetic -- 0 = bf16x6 (three bf16 planes per operand, six product terms: no range limit), 1 = f16x3 (two fp16 planes, three
// terms, weights x16: the inference kernels' arithmetic; half the MFMAs; an operand beyond the fp16 range raises a.ovf and the caller
// recomputes the product with AR = 0 -- the training FORWARD only, whose operands are activations)
// NW (round 5): waves per workgroup.  8 = two waves per SIMD (512 threads, 256 registers per wave): wave w owns 64 features (the second half
// of the wave slice w / 2 for odd w: the addressing of the shared 32-row shape, FH = 2, inside ONE workgroup) and stages half as many rows.
template <int R, int CT, int FH, int AR = 0, int NW = 4, bool MDEV = false>
__device__ __forceinline__ void lin512_body(const Lin512Args& a, const int bid, const int nblk) {
  static_assert(NW == 4 || (NW == 8 && FH == 1), "eight waves: whole tiles only");
  constexpr int NP = AR == 1 ? 2 : 3, NT = AR == 1 ? 3 : 6;  // planes per operand, product terms
  constexpr int kRows = 32 * CT, kSlabFrags = kStepsPerSlab * CT * NP, NQ = 16 * CT / NW;      // NQ: staging requests per wave and slab
  constexpr int kStageRows = 32 * CT / NW;                   // rows of the tile this wave stages
  // A staging write puts 8 bytes per lane into the fragments of EIGHT k16 steps at once (a lane holds 4 consecutive k of one row); with the
  // steps a multiple of 1 KB apart those are 8 lanes per LDS bank (PMC, round 4: 70 % of the kernel's LDS-active cycles were bank-conflict
  // cycles).  kStepPad = 32 bytes between the steps spreads them over the banks (2 lanes per bank are left: the two 8-row halves of a
  // fragment are 512 bytes apart by the MFMA layout); the fragment reads stay 1 KB contiguous per instruction.
  constexpr int kStepBytes = CT * NP * 1024 + kStepPad, kSlabBytes = kStepsPerSlab * kStepBytes;
  constexpr int NRT = 16 / (FH * NW), NF = NP * NRT;         // MFMA row (= feature) tiles per wave, weight fragments per k16 step
#ifdef DINER_L512_OLD_EPI      // A/B build: the direct epilogue for every shape
  constexpr bool kLdsEpi = false;
#else
  constexpr bool kLdsEpi = CT == 4 && NW == 4 && FH == 1 && AR == 1;      // the epilogue leaves through LDS (below)
#endif
  if (a.gate && *a.gate == 0) return;                        // fall-back launch of an f16x3 product that stayed in range: nothing to do
  if (a.gate2 && *a.gate2 == 0) return;
  long long Mrows = a.M;
  if constexpr (MDEV) {                                      // a row list whose length only the device knows (k_lin512_rows; a separate
    if (a.skip_silent && *a.skip_silent != 0) return;        // instantiation: one more live value costs the 128-row shapes spills)
    if (a.m_dev) {
      const long long md = *a.m_dev;
      Mrows = md < Mrows ? md : Mrows;
    }
  }
  if (a.skip && *a.skip != 0) {                              // f16x3 launch of a step whose weights do not fit: the bf16x6 twin works
    if (a.ovf && threadIdx.x == 0) {                         // (the forward's twin is gated on this product's flag)
      *a.ovf = 1;
      if (a.ovf2) *a.ovf2 = 1;
    }
    return;
  }
  // AR = 1 with amax_in: the staged operand times sx = 2^(14 - E), E the exponent of the operand's maximum; the epilogue takes it out
  float sx = 1.0f, inv_scale = kF16InvScale;
  if constexpr (AR == 1) {
    if (a.amax_in) {
      const unsigned e = (*a.amax_in >> 23) & 0xffu;          // (0 / tiny / inf / NaN maxima: no scaling)
      if (e >= 32u && e < 255u) {
        sx = __uint_as_float((268u - e) << 23);
        inv_scale = kF16InvScale * __uint_as_float((e - 14u) << 23);
      }
    }
  }
  unsigned y_max = 0;                                        // amax_out: running maximum of the |Y| bit patterns this lane has stored
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char* lds_ptr;
  typedef __attribute__((address_space(3))) bf8* lds_bf8;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long long n_tiles = (Mrows + kRows - 1) / kRows;
  const int relu_floor1 = (a.flags & kL512ReluIn) ? 0 : (int)0x80000000, relu_floor2 = a.relu2 ? 0 : (int)0x80000000;
  int relu_floor = relu_floor1;                              // of the slab being converted (two contraction segments: set per slab)
  const int n_slabs = a.X2 ? 2 * kSlabs : kSlabs;            // slabs kSlabs .. 2 kSlabs - 1: the second segment (X2, Wp2)
  lds_ptr lbase = (lds_ptr)smem + lane * 16;                 // lane's 16 B slot in fragment 0 of slab buffer 0
  // ---- staging share of this wave: rows [8 CT w, 8 CT (w + 1)) of the tile, all 128 contraction indices of the slab.  Request i (0..NQ-1)
  // reads rows 8 CT w + 2 i and + 1 whole: lanes 0..31 one row (512 contiguous bytes), lanes 32..63 the next -- 8 cache lines per
  // instruction (a first version read B-fragment-shaped pieces, 32 rows x 16 B per instruction: 32+ lines each, and the slab
  // staging cost 16 % of the kernel).  A lane then holds 4 consecutive k of one row: three 8-byte pieces of B fragments.
  // Buffer loads (as in wgrad512_body): a descriptor over the tile's rows made from scalars, the lane's part of the offset in one register
  // for the whole kernel, request and slab as the scalar offset; rows past M read as zeros (their results are never stored).
  f32x4 xst[NQ];
  const unsigned xvoff = ((unsigned)(kStageRows * wave + (lane >> 5)) * (unsigned)a.ldx + 4u * (lane & 31)) * 4u;
  auto request_one = [&](int i, long long tile, int slab) {
    const long long row0 = tile * kRows;
    long long left = Mrows - row0;
    left = left < 0 ? 0 : (left > kRows ? kRows : left);
    const float* src = slab >= kSlabs ? a.X2 : a.X;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src + (size_t)row0 * a.ldx), 0,
                                                                        (int)(left * a.ldx * 4), 0x00020000);
    xst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, xvoff, ((unsigned)(2 * i) * (unsigned)a.ldx + 128u * (slab & (kSlabs - 1))) * 4u, 0));
  };
  auto request_slab = [&](long long tile, int slab) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) request_one(i, tile, slab);
  };
  typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) bf4* lds_bf4;
  // where this lane's 4 values go inside a slab buffer: step s = k / 16, lane' = (row & 31) + 32 ((k / 8) & 1), element k & 7
  const int k4 = 4 * (lane & 31);
  const int st_off = (k4 >> 4) * kStepBytes + (32 * ((k4 >> 3) & 1)) * 16 + (k4 & 7) * 2;
  bf4 sp0, sp1, sp2;                                         // the request being converted (two halves, see the slab loop)
  unsigned x_max = 0;                                        // AR = 1: packed running maximum of the |hi| halves this lane has produced
  auto stash_half = [&](int i, int half) {
    if constexpr (AR == 1) {
      // two values -> one hi word, one lo word; the halves are watched instead of the fp32 values: a half that is inf (0x7c00) or NaN
      // is an operand that left the fp16 range
      const float x0 = __int_as_float(max(__float_as_int(xst[i][2 * half]), relu_floor)) * sx;
      const float x1 = __int_as_float(max(__float_as_int(xst[i][2 * half + 1]), relu_floor)) * sx;
      const unsigned h = cvt_pk_f16_w(x0, x1);
      const unsigned l = cvt_pk_f16_w(resid_lo_w(h, x0), resid_hi_w(h, x1));
      typedef unsigned short us2 __attribute__((ext_vector_type(2)));
      x_max = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(us2, x_max), __builtin_bit_cast(us2, h & 0x7fff7fffu)));
      typedef unsigned u2 __attribute__((ext_vector_type(2)));
      u2 w0 = __builtin_bit_cast(u2, sp0), w1 = __builtin_bit_cast(u2, sp1);
      w0[half] = h;
      w1[half] = l;
      sp0 = __builtin_bit_cast(bf4, w0);
      sp1 = __builtin_bit_cast(bf4, w1);
      return;
    }
#pragma unroll
    for (int j = 2 * half; j < 2 * half + 2; ++j) {
      const float x = __int_as_float(max(__float_as_int(xst[i][j]), relu_floor));      // relu, or the identity (floor INT_MIN): one instruction
      const __bf16 a0 = (__bf16)x;
      const float r1 = x - (float)a0;
      const __bf16 a1 = (__bf16)r1;
      sp0[j] = a0;
      sp1[j] = a1;
      sp2[j] = (__bf16)(r1 - (float)a1);
    }
  };
  auto stash_write = [&](int buf, int i) {
    const int r = kStageRows * wave + 2 * i + (lane >> 5);   // row within the tile
    lds_ptr d = (lds_ptr)smem + buf * kSlabBytes + st_off + ((r >> 5) * NP) * 1024 + (r & 31) * 16;
    *(lds_bf4)(d) = sp0;
    *(lds_bf4)(d + 1024) = sp1;
    if constexpr (NP == 3) *(lds_bf4)(d + 2048) = sp2;
  };
  auto stash_req = [&](int buf, int i) {                     // convert + write request i of the slab in flight
    stash_half(i, 0);
    stash_half(i, 1);
    stash_write(buf, i);
  };
  // ---- weights: wave-private stream, NF fragments (NRT row tiles x 3 planes) per k16 step out of the 12 of the packed wave slice
  typedef const __attribute__((address_space(1))) char* gptr;
  const int half = FH == 2 ? (bid & 1) : 0;
  const int wslice = NW == 8 ? (wave >> 1) : FH == 2 ? 2 * half + (wave >> 1) : wave;      // 128-feature slice of the packed weights
  const int rt0 = (FH == 2 || NW == 8) ? 2 * (wave & 1) : 0;                 // first of this wave's row tiles inside the slice
  const gptr wbase = (gptr)(reinterpret_cast<const char*>(a.Wp)) + ((size_t)wslice * 32 * (4 * NP) + NP * rt0) * 1024;
  const gptr wbase2 = (gptr)(reinterpret_cast<const char*>(a.Wp2 ? a.Wp2 : a.Wp)) + ((size_t)wslice * 32 * (4 * NP) + NP * rt0) * 1024;
  const int step_mask = 8 * n_slabs - 1;                     // k16 steps of a tile - 1 (the weight stream repeats per tile)
  const unsigned woff = lane * 16;
  bf8 wr[R][NF];
  auto load_w = [&](bf8 (&dst)[NF], int step, int first, int count) {        // fragments [first, first + count) of step (0..31)
#ifdef DINER_L512_ABL_W       // ablation (wrong results): a 24 KB weight working set per wave, i.e. no L2 latency on the weight stream
    step &= 1;
#endif
    gptr p = (step >= 32 ? wbase2 : wbase) + (size_t)(step & 31) * (4 * NP) * 1024;
    asm volatile("" : "+s"(p));                  // scalar base + per-lane 32-bit offset + immediate: no address registers per load
#pragma unroll
    for (int i = first; i < first + count; ++i) dst[i] = *(const __attribute__((address_space(1))) bf8*)(p + woff + i * 1024);
  };

  long long tile = bid / FH;
  const int tile_stride = nblk / FH;
  if (tile >= n_tiles) return;
  // prologue: slab 0 of the first tile
  request_slab(tile, 0);
#pragma unroll
  for (int i = 0; i < NQ; ++i) stash_req(0, i);
  request_slab(tile, 1);                                     // rolling: the slab after the next one is always in flight
  __syncthreads();

  f32x16 acc[NRT][CT];
  int unit = 0;                                              // slabs processed by this workgroup so far (buffer = unit & 1)
  [[maybe_unused]] unsigned long long pf_slab = 0, pf_bar = 0, pf_epi = 0, pf_tiles = 0;
  [[maybe_unused]] const unsigned long long pf_t0 = L512_CLK();
  // weight ring: the first R - 1 steps; from then on every step requests the step R - 1 ahead of it (the stream repeats per tile)
  sfor<R - 1>([&](auto S) { load_w(wr[decltype(S)::value], decltype(S)::value, 0, NF); });
  for (; tile < n_tiles; tile += tile_stride) {
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[rt][ct][e] = 0.0f;
#pragma nounroll
    for (int slab = 0; slab < n_slabs; ++slab, ++unit) {
      [[maybe_unused]] const unsigned long long pf_a = L512_CLK();
      relu_floor = (slab + 1 >= kSlabs && slab + 1 < n_slabs) ? relu_floor2 : relu_floor1;      // the slab converted during this one
      const int buf = unit & 1;
      // Staging, branch-free so that the conversion can be scheduled between the MFMAs: during this slab the NEXT slab (requested one
      // slab ago, in xst) is converted and written to the other buffer, one request per k16 step, and each request register is
      // re-armed at once with the slab after that.  Past the workgroup's last slab the requests repeat valid addresses (harmless).
      const bool last = slab == n_slabs - 1;
      long long t2 = slab >= n_slabs - 2 ? tile + tile_stride : tile;          // tile / slab two slabs ahead
      const int s2 = (slab + 2) & (n_slabs - 1);
      if (t2 >= n_tiles) t2 = tile;
      (void)last;
      lds_ptr rb = lbase + buf * kSlabBytes;
      asm volatile("" : "+v"(rb));
      // B fragments [parity][row part][plane]; CT = 4: ONE buffer, each row part re-read for the next step right after its last use in this
      // one (groups run row part by row part: part c is free from group 3 (c + 1) on; the last part at the top of the next step) -- 32
      // registers less, which the 128-row shape needs
      constexpr bool kRollB = CT == 4;
      bf8 bb[kRollB ? 1 : 2][CT][NP];
      auto load_b1 = [&](int par, int s, int h) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) bb[par][h][pl] = *(lds_bf8)(rb + s * kStepBytes + (h * NP + pl) * 1024);
      };
      auto load_b = [&](int par, int s) {
#pragma unroll
        for (int h = 0; h < CT; ++h) load_b1(par, s, h);
      };
      load_b(0, 0);
      sfor<kStepsPerSlab>([&](auto S) {
        constexpr int s = decltype(S)::value;
        const int gstep = slab * kStepsPerSlab + s;
        bf8 (&wc)[NF] = wr[s % R];
        // NG = NT x CT groups (row half ct, product term t): one MFMA on each of the wave's row tiles -- consecutive MFMAs never share an
        // accumulator; the next step's weights are requested in the first groups (NF fragments, LW per group)
        constexpr int NG = NT * CT, LW = (NF + NG - 1) / NG < 2 ? 2 : (NF + NG - 1) / NG;
        sfor<NG>([&](auto G) {
          constexpr int g = decltype(G)::value, ct = g / NT, t = g % NT;
          // smallest terms first.  bf16x6: (a2 b0) (a0 b2) (a1 b1) (a1 b0) (a0 b1) (a0 b0);  f16x3: (lo hi) (hi lo) (hi hi)
          constexpr int ia = AR == 1 ? (t == 0 ? 1 : 0) : (t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0);
          constexpr int ib = AR == 1 ? (t == 1 ? 1 : 0) : (t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (LW * g < NF) load_w(wr[(s + R - 1) % R], (gstep + R - 1) & step_mask, LW * g, (LW * g + LW <= NF ? LW : NF - LW * g));
          if constexpr (kRollB) {
            if constexpr (t == 0 && ct > 0 && s + 1 < kStepsPerSlab) load_b1(0, s + 1, ct - 1);
            if constexpr (g == 0 && s > 0) load_b1(0, s, CT - 1);
          } else {
            if constexpr (g == (NG >= 12 ? 7 : NG >= 6 ? 3 : 1) && s + 1 < kStepsPerSlab) load_b((s + 1) & 1, s + 1);
          }
          // staging side task: one of the slab's NQ requests per step (steps 0 .. NQ-1), in three groups
          // // (which steps makes no measurable difference)
          constexpr int g0 = AR == 1 ? NG - 3 : (CT == 2 ? 8 : 2);
          constexpr int g1 = 3;                                // CT = 4 (16 requests per slab): a second request per step, s + 8, in groups 3..5
#ifndef DINER_L512_ABL_X
          if constexpr (s < NQ && g == g0) stash_half(s, 0);
          if constexpr (s < NQ && g == g0 + 1) stash_half(s, 1);
          if constexpr (s < NQ && g == g0 + 2) {
            stash_write(buf ^ 1, s);
            request_one(s, t2, s2);
          }
          if constexpr (s + 8 < NQ && g == g1) stash_half(s + 8, 0);
          if constexpr (s + 8 < NQ && g == g1 + 1) stash_half(s + 8, 1);
          if constexpr (s + 8 < NQ && g == g1 + 2) {
            stash_write(buf ^ 1, s + 8);
            request_one(s + 8, t2, s2);
          }
#endif
          const bf8 b = bb[kRollB ? 0 : (s & 1)][ct][ib];
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) {
            if constexpr (AR == 1) DINER_F16_MFMA(acc[rt][ct], wc[NP * rt + ia], b);
            else DINER_BF16_MFMA(acc[rt][ct], wc[NP * rt + ia], b);
          }
          if constexpr ((s < NQ && (g == g0 || g == g0 + 1)) || (s + 8 < NQ && (g == g1 || g == g1 + 1))) {      // the conversion between the MFMAs, not in front of them: <= 6 vector-ALU slots per MFMA
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
              __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
          }
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) asm volatile("" : "+a"(acc[rt][ct]));
        });
      });
      [[maybe_unused]] const unsigned long long pf_b = L512_CLK();
      __syncthreads();                                       // slab buffer `buf` is free, the next one is complete
#ifdef DINER_L512_PROF
      pf_slab += pf_b - pf_a;
      pf_bar += L512_CLK() - pf_b;
#endif
    }
    [[maybe_unused]] const unsigned long long pf_e = L512_CLK();
    if constexpr (kLdsEpi) {
      // ---- epilogue through LDS (round 6; the 128-row f16x3 shape).  In the D layout a lane holds ONE row and 4 consecutive features per
      // register quad: a 16-byte store instruction touches 32 rows x 32 bytes -- 32 cache lines for 1 KB -- and the phase timer
      // (tools/prof_l512.py, profiles/r06_l512_phase_timer.txt) booked 31 k of a 91 k-clock tile on the epilogue (69 k of 129 k with one
      // tensor term read the same way).  Here every 32-row part goes through the wave's share of the slab buffer that has just been
      // consumed and comes back ROW-CONTIGUOUS: lanes 0..31 hold the 512 bytes of the wave's feature slice of one row, lanes 32..63 the
      // next row's -- every store / residual / accumulate access is 2 x 512 contiguous bytes.  The region is exactly the bytes THIS wave
      // writes when it stages (row part `wave` of each of the 8 steps: NP KB per step), nobody reads them before the next slab barrier:
      // no barrier around the epilogue.  16-byte slot s of row r sits at slot s ^ (r & 15): conflict-free writes (16 rows per pass) and reads.
      // Buffer descriptors over the tile's valid rows: rows past M read as zeros and their stores are dropped.
      typedef __attribute__((address_space(3))) f32x4* lds_f4;
      const int fb = (unit - 1) & 1;
      lds_ptr eb = (lds_ptr)smem + fb * kSlabBytes + wave * (NP * 1024);
      // every per-lane value of the epilogue is derived HERE from an opaque copy of the lane id: derived from `lane` itself they are
      // loop-invariant, the compiler hoists ~40 of them (16 read and 16 write addresses among them) out of the tile loop, keeps them across
      // the slab loop and spills -- and a spill reload between the epilogue's stores is a load that waits for those stores' acknowledgement
      int el = lane;
      asm volatile("" : "+v"(el));
      const int er = el & 31, eh = el >> 5;
      const long long row0 = tile * kRows;
      const long long left_ll = Mrows - row0;
      const int left = left_ll > kRows ? kRows : (int)left_ll;
      const int left_lane = left - eh;                                          // row 2 i + eh of a part is valid iff 32 ct + 2 i < left_lane
      const unsigned rowbytes = (unsigned)a.ldy * 4u;
      auto rsrc = [&](const float* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p + (size_t)row0 * a.ldy), 0, left * a.ldy * 4, 0x00020000); };
      const __amdgpu_buffer_rsrc_t rs_y = rsrc(a.Y);
      const unsigned gvoff = ((unsigned)eh * (unsigned)a.ldy + 128u * wslice + 4u * er) * 4u;
      lds_ptr wbase_e = eb + (er >> 2) * kStepBytes + (er & 3) * 512;          // write side: this lane's row er
      const int wx = ((er & 15) ^ eh) * 16;                                     // (8 rt + 2 q4 + eh) ^ (er & 15) = (8 rt + 2 q4) ^ wx / 16
      lds_ptr rbase_e = eb + eh * 512;                                          // read side: row 2 i + eh, slot er ^ (row & 15) = (er ^ eh) ^ 2 (i & 7)
      const int rx = (er ^ eh) * 16;
      f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + 128 * wslice + 4 * er);
      if (a.bias2) bv += *reinterpret_cast<const f32x4*>(a.bias2 + 128 * wslice + 4 * er);
      const bool has_bias = a.bias || a.bias2;
      // The tensor terms are requested ONE GROUP (4 instructions = 8 rows) AHEAD: the requests of group g + 1 are issued in front of the
      // stores of group g.  Memory operations of a wave retire in order, so a load issued behind a store waits for that store's
      // acknowledgement -- with loads, wait, stores per group every group paid a store round trip plus a load round trip (78 k clocks per tile
      // with one residual; the direct epilogue: 69 k).  Registers decide the rest (the staging requests of the next tile, the first step
      // of the weight ring and the other parts' accumulators stay live here; a spill reload is a load behind stores as well), and so does
      // the control flow: with one uniform branch per optional term inside the unrolled groups the compiler waited for every mask-bits
      // dword and spilled it.  So the term set is a COMPILE-TIME parameter of the epilogue (PK: the pipelined 16-byte term -- 0 none, 1 the
      // residual, 2 the old output of an accumulating product; BITS: mask bits) and the five sets the training step has are dispatched
      // once per tile; anything else (a second residual, fp32 masks: measurement switches) takes the GEN variant, which reads its
      // terms in place.
      constexpr int GN = 4, GPC = 16 / GN, NGRP = GPC * CT;                     // instructions per group, groups per 32-row part / per tile
      const bool accum = (a.flags & kL512Accum) != 0;
      const unsigned bvoff = ((unsigned)eh * 16u + 4u * wslice + (er & 3)) * 4u;      // mask bits: the lane's features 4 er .. + 3 of the wave's 128-feature
      const int sh = 4 * (er >> 2);                                                   // slice: dword 4 wslice + (er & 3) of the row's 16, bits 4 (er >> 2) + c
      auto rsrc_bits = [&]() { return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(a.maskbits + (size_t)row0 * 16), 0, left * 64, 0x00020000); };
      auto epilogue = [&](auto PKc, auto BITSc, auto GENc) {
        constexpr int PK = decltype(PKc)::value;
        constexpr bool BITS = decltype(BITSc)::value, GEN = decltype(GENc)::value;
        const __amdgpu_buffer_rsrc_t rs_p = rsrc(PK == 1 ? a.resid : a.Y);
        const __amdgpu_buffer_rsrc_t rs_mb = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>((BITS ? a.maskbits : (const unsigned*)a.Y) + (size_t)row0 * 16), 0, left * 64, 0x00020000);
        // a ring of PD groups of term requests in flight (DINER_L512_EPI_PD).  Measured (profiles/r06_train_wgrad_phase_timer.txt): 2 groups ahead
        // change nothing (accumulating epilogue 22.3 k clocks per tile, step 120.9 ms against 120.9), 3 cost 70 spill instructions and 1 ms:
        // the epilogue is not waiting for latency any more.  Default 1.
        constexpr int PD = (PK != 0 || BITS) ? DINER_L512_EPI_PD : 1;
        f32x4 pre[PD][GN];
        unsigned mb[PD][GN];
        auto issue = [&](f32x4 (&tp)[GN], unsigned (&tm)[GN], int g) {
#pragma unroll
          for (int j = 0; j < GN; ++j) {
            const int rw = 32 * (g / GPC) + 2 * (GN * (g % GPC) + j);
            if constexpr (PK != 0) tp[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, gvoff, (unsigned)rw * rowbytes, 0));
            if constexpr (BITS) tm[j] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_mb, bvoff, (unsigned)rw * 64u, 0);
          }
        };
        sfor<PD>([&](auto Pi) {
          constexpr int pg = decltype(Pi)::value;
          if constexpr (pg < NGRP) issue(pre[pg], mb[pg], pg);
        });
        sfor<NGRP>([&](auto Gi) {
          constexpr int g = decltype(Gi)::value, ct = g / GPC, gi = g % GPC, slot = g % PD;
          if constexpr (gi == 0) {                                              // this part's accumulators -> the wave's LDS region
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) {
                f32x4 t;
#pragma unroll
                for (int c = 0; c < 4; ++c) t[c] = acc[rt][ct][4 * q4 + c] * inv_scale;
                *(lds_f4)(wbase_e + (wx ^ ((8 * rt + 2 * q4) * 16))) = t;
              }
          }
          f32x4 v[GN];
#pragma unroll
          for (int j = 0; j < GN; ++j) {
            const int i = GN * gi + j;
            v[j] = *(lds_f4)(rbase_e + (i >> 1) * kStepBytes + (i & 1) * 1024 + (rx ^ ((i & 7) * 32)));
          }
          f32x4 cur[GN];
          unsigned curb[GN];
#pragma unroll
          for (int j = 0; j < GN; ++j) {
            cur[j] = pre[slot][j];
            curb[j] = mb[slot][j];
          }
          if constexpr (g + PD < NGRP) issue(pre[slot], mb[slot], g + PD);      // (in front of this group's stores; rows past M: the descriptors return zeros)
#pragma unroll
          for (int j = 0; j < GN; ++j) {
            const int rw = 32 * ct + 2 * (GN * gi + j);
            const unsigned so = (unsigned)rw * rowbytes;
            if (has_bias) v[j] += bv;
            if constexpr (PK == 1) v[j] += cur[j];
            if constexpr (GEN) {
              if (a.resid) v[j] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc(a.resid), gvoff, so, 0));
              if (a.resid2) v[j] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc(a.resid2), gvoff, so, 0));
              if (a.mask) {
                const f32x4 m = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc(a.mask), gvoff, so, 0));
#pragma unroll
                for (int c = 0; c < 4; ++c) v[j][c] = m[c] > 0.0f ? v[j][c] : 0.0f;
              }
              if (a.maskbits) {
                const unsigned m = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc_bits(), bvoff, (unsigned)rw * 64u, 0) >> sh;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[j][c] = ((m >> c) & 1u) ? v[j][c] : 0.0f;
              }
              if (accum) v[j] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_y, gvoff, so, 0));
            }
            if constexpr (BITS) {
              const unsigned m = curb[j] >> sh;
#pragma unroll
              for (int c = 0; c < 4; ++c) v[j][c] = ((m >> c) & 1u) ? v[j][c] : 0.0f;
            }
            if constexpr (PK == 2) v[j] += cur[j];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[j]), rs_y, gvoff, so, 0);
            if (a.amax_out && rw < left_lane) {
#pragma unroll
              for (int c = 0; c < 4; ++c) y_max = max(y_max, __float_as_uint(v[j][c]) & 0x7fffffffu);
            }
          }
        });
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      using BT = std::integral_constant<bool, true>;
      using BF = std::integral_constant<bool, false>;
      const bool generic = a.resid2 || a.mask || (a.resid && (accum || a.maskbits));
      if (generic) epilogue(I0{}, BF{}, BT{});
      else if (a.maskbits) {
        if (accum) epilogue(I2{}, BT{}, BF{});
        else epilogue(I0{}, BT{}, BF{});
      } else if (accum) epilogue(I2{}, BF{}, BF{});
      else if (a.resid) epilogue(I1{}, BF{}, BF{});
      else epilogue(I0{}, BF{}, BF{});
    } else {
    // ---- epilogue: D layout of a 32 x 32 tile: lane holds row (of x) = lane & 31, features 8 (e >> 2) + 4 (lane >> 5) + (e & 3).
      // One 32-row half at a time into registers (the weight ring's are free here), then one pass per optional term -- a branch per term and
      // half, not one per term and four values (130 uniform branches per tile as first written).
  #pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const long long row = tile * kRows + 32 * ct + (lane & 31);
        if (row >= Mrows) continue;
        constexpr int NV = 4 * NRT;
        f32x4 v[NV];
        const size_t at0 = (size_t)row * a.ldy + 128 * wslice + 32 * rt0 + 4 * (lane >> 5);        // + 32 rt + 8 q4
  #pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
  #pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
  #pragma unroll
            for (int c = 0; c < 4; ++c) v[4 * rt + q4][c] = acc[rt][ct][4 * q4 + c];
            if constexpr (AR == 1) v[4 * rt + q4] *= inv_scale;
          }
        if (a.bias) {
          const float* bp = a.bias + 128 * wslice + 32 * rt0 + 4 * (lane >> 5);
  #pragma unroll
          for (int n = 0; n < NV; ++n) v[n] += *reinterpret_cast<const f32x4*>(bp + 32 * (n >> 2) + 8 * (n & 3));
        }
        if (a.bias2) {
          const float* bp = a.bias2 + 128 * wslice + 32 * rt0 + 4 * (lane >> 5);
  #pragma unroll
          for (int n = 0; n < NV; ++n) v[n] += *reinterpret_cast<const f32x4*>(bp + 32 * (n >> 2) + 8 * (n & 3));
        }
        if (a.resid) {
  #pragma unroll
          for (int n = 0; n < NV; ++n) v[n] += *reinterpret_cast<const f32x4*>(a.resid + at0 + 32 * (n >> 2) + 8 * (n & 3));
        }
        if (a.resid2) {
  #pragma unroll
          for (int n = 0; n < NV; ++n) v[n] += *reinterpret_cast<const f32x4*>(a.resid2 + at0 + 32 * (n >> 2) + 8 * (n & 3));
        }
        if (a.mask) {
  #pragma unroll
          for (int n = 0; n < NV; ++n) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(a.mask + at0 + 32 * (n >> 2) + 8 * (n & 3));
  #pragma unroll
            for (int c = 0; c < 4; ++c) v[n][c] = m[c] > 0.0f ? v[n][c] : 0.0f;
          }
        }
        if (a.maskbits) {
          // one 16-byte load per row: the 128 decisions of the wave's feature slice; v[n][c] is feature 32 (rt0 + rt) + 8 q4 + 4 h + c of it
          // (n = 4 rt + q4, h = lane / 32): dword 2 (q4 % 2) + h, bit 8 (rt0 + rt) + 4 (q4 / 2) + c
          const u32x4 mb = *reinterpret_cast<const u32x4*>(a.maskbits + (size_t)row * 16 + 4 * wslice);
          const bool hi = lane >= 32;
          const unsigned m0 = (hi ? mb[1] : mb[0]) >> (8 * rt0), m1 = (hi ? mb[3] : mb[2]) >> (8 * rt0);
  #pragma unroll
          for (int n = 0; n < NV; ++n) {
            const unsigned m = (n & 1) ? m1 : m0;
  #pragma unroll
            for (int c = 0; c < 4; ++c) v[n][c] = ((m >> (8 * (n >> 2) + 4 * ((n & 3) >> 1) + c)) & 1u) ? v[n][c] : 0.0f;
          }
        }
        if (a.flags & kL512Accum) {
  #pragma unroll
          for (int n = 0; n < NV; ++n) v[n] += *reinterpret_cast<const f32x4*>(a.Y + at0 + 32 * (n >> 2) + 8 * (n & 3));
        }
  #pragma unroll
        for (int n = 0; n < NV; ++n) *reinterpret_cast<f32x4*>(a.Y + at0 + 32 * (n >> 2) + 8 * (n & 3)) = v[n];
        if (a.amax_out) {
  #pragma unroll
          for (int n = 0; n < NV; ++n)
  #pragma unroll
            for (int c = 0; c < 4; ++c) y_max = max(y_max, __float_as_uint(v[n][c]) & 0x7fffffffu);
        }
      }
    }
#ifdef DINER_L512_PROF
    pf_epi += L512_CLK() - pf_e;
    ++pf_tiles;
#endif
  }
#ifdef DINER_L512_PROF
  if (lane == 0) {
    atomicAdd(&g_l512_prof[0], L512_CLK() - pf_t0);
    atomicAdd(&g_l512_prof[1], pf_slab);
    atomicAdd(&g_l512_prof[2], pf_bar);
    atomicAdd(&g_l512_prof[3], pf_epi);
    atomicAdd(&g_l512_prof[4], pf_tiles);
    atomicAdd(&g_l512_prof[5], 1ull);
  }
#endif
  if (a.amax_out) {                                          // (a NaN's pattern is the largest: it reaches the consumer, which then does not scale)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) y_max = max(y_max, (unsigned)__shfl_xor((int)y_max, o, 64));
    if (lane == 0 && y_max) atomicMax(a.amax_out, y_max);
  }
  if constexpr (AR == 1) {                                   // an operand beyond the fp16 range (or not finite): the caller's bf16x6 launch recomputes
    if (a.ovf && ((x_max & 0xffffu) >= 0x7c00u || (x_max >> 16) >= 0x7c00u)) {      // a hi half was inf or NaN
      *a.ovf = 1;
      if (a.ovf2) *a.ovf2 = 1;
    }
  }
}

#ifdef DINER_L512_PROF
}  // namespace train
}  // namespace diner
extern "C" int diner_debug_l512_prof(unsigned long long* out8, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out8, HIP_SYMBOL(diner::train::g_l512_prof), 8 * sizeof(unsigned long long));
  if (reset) {
    unsigned long long z[8] = {};
    hipMemcpyToSymbol(HIP_SYMBOL(diner::train::g_l512_prof), z, sizeof(z));
  }
  return 0;
}
namespace diner {
namespace train {
#endif
// ---- host side ---------------------------------------------------------------------------------------------------------------
int lin512_pack(const float* W, int transpose, void* dst, hipStream_t stream) {
  hipLaunchKernelGGL(k_pack_w512, dim3(128), dim3(256), 0, stream, W, transpose, (__bf16*)dst);
  DINER_LAUNCH_OK();
  return 0;
}

int lin512_pack_many(const PackMany& w, int n, void* base, hipStream_t stream, int modes, int* wbad) {
  hipLaunchKernelGGL(k_pack_w512_many, dim3(128, n, modes), dim3(256), 0, stream, w, (char*)base, wbad);
  DINER_LAUNCH_OK();
  return 0;
}

}  // namespace train
}  // namespace diner
