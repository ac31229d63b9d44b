// Training path, the weight gradients of the 512 x 512 layers: dW[f][k] += sum_m dy[m][f] act(x)[m][k], db[f] += sum_m dy[m][f]
// (the adjoint of y = act(x) W^T + b with respect to W and b; resnetfc.py:61-69 / :129-159 under torch autograd in DINER.calc_losses,
// diner.py:217-290) -- the third product of every layer, on the same persistent one-wave-per-SIMD scheme as train_lin512.hip.
//
//   * arithmetic: "bf16x6" (three bf16 planes per fp32 operand, six v_mfma_f32_32x32x16_bf16 products, fp32 accumulation);
//   * the contraction runs over the ROWS m.  A workgroup owns one 128 (f) x 256 (k) tile of dW -- 8 tiles cover the matrix -- and one
//     chunk of the rows (8 tiles x up to 32 chunks = one workgroup per CU); the four waves split the tile 2 x 2, 64 f x 128 k each:
//     2 x 4 MFMA tiles = 128 accumulator registers (18 operand fragments from LDS per k16 step for 48 MFMAs), added to dW with atomics at the end (dW is zeroed by the caller);
//   * both operands are stored with the contraction index OUTERMOST (row-major (M, 512) matrices), but an MFMA lane wants 8 consecutive
//     contraction indices of ONE column.  No transpose is needed: a lane that loads the same 4 (x) or 2 (dy) columns of 8 consecutive
//     rows holds exactly four / two such lane-fragments.  Per 32-row slab a wave requests 8 x 16 B + 8 x 8 B per lane (its own 64
//     columns of x and 32 columns of dy), converts them to bf16 planes as a side task of the previous slab's MFMAs and writes them to
//     LDS in fragment order; two 72 KB slab buffers, one barrier per slab (96 MFMAs per wave);
//   * the bias gradient is the row sum of the dy operand: the staging lanes of the workgroups with k-tile 0 keep it in two registers.
// Replaces k_gemm_bf16x6<2> (train.hip) for these shapes: 128 x 128 x 32 tiles with two barriers per k-tile and 64-way split-K
// reached 138-142 TFLOP/s on 327680 rows and 93 us per product on the reference batch's 20480 rows.
#include <atomic>
#include <utility>
#include "field_common.hpp"
#include "train_lin512.hpp"
// (compiled as part of train_512.hip, which holds the kernel entry points and the launchers)

namespace diner {
namespace train {

typedef __bf16 bf8w __attribute__((ext_vector_type(8)));
typedef float f32x16w __attribute__((ext_vector_type(16)));
typedef float f32x2w __attribute__((ext_vector_type(2)));

constexpr int kWgFragsPerStep = (4 + 8) * 3;                     // [A: 4 f tiles | B: 8 k tiles][plane 3] fragments of 1 KB per k16 step
constexpr int kWgSlabBytes = 2 * kWgFragsPerStep * 1024;         // two steps (32 rows) per slab: 72 KB
constexpr size_t kLdsBytesWgrad = (size_t)2 * kWgSlabBytes;      // two slab buffers (f16x3: two planes, 96 KB; its 256 x 256-tile body: 128 KB)
constexpr int kWgMaxChunks = 64;                                 // row chunks the scratch holds (one partial dW + db per chunk)
constexpr int kWgPlanChunks = 32;                                // chunks of the 128 x 256-tile plan (8 tiles x 32 chunks = 256 workgroups)
constexpr int kWgPlanChunksWide = 64;                            // chunks of the 256 x 256-tile plan (4 tiles x 64 chunks = 256 workgroups)

struct Wgrad512Args {
  const float* dY;       // (M, ldy)
  const float* X;        // (M, ldx)
  float* dW;             // (512, 512), atomically accumulated (zeroed by the caller)
  float* db;             // 512 or null, atomically accumulated
  float* part;           // null, or (n_chunks, 512, 512) scratch: every chunk STORES its partial dW there and k_wgrad512_reduce sums them
                         // (32 atomics per element of dW cost as much as the products of the reference batch's 20480 rows)
  long long M;
  int ldy, ldx, relu_x;
  int n_chunks;          // row chunks (grid = 8 * n_chunks)
  long long rows_per_chunk;      // multiple of 32
  // round 4, the f16x3 instance (AR = 1): dY staged times the power of two that brings *amax_dy into [2^14, 2^15) (Lin512Args.amax_in), X as
  // it is; skip / gate: the part does nothing when *skip != 0 / unless *gate != 0 (the f16x3 launch and its bf16x6 twin)
  const unsigned* amax_dy;
  const int* skip;
  const int* gate;
  int wide;              // f16x3 launch: 256 x 256 tiles (wgrad512_body_wide: grid = 4 * n_chunks)
};

template <class F, int... I>
__device__ __forceinline__ void wfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void wfor(F&& f) {
  wfor_impl(f, std::make_integer_sequence<int, N>{});
}

#define DINER_WG_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, ACC, 0, 0, 0)
#define DINER_WG_MFMA_F16(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf8, A), __builtin_bit_cast(hf8, B), ACC, 0, 0, 0)

// (bid: the workgroup's index within the weight-gradient part of its launch, train_512.hip)
// AR: 0 = bf16x6 (three bf16 planes per operand, six product terms); 1 = f16x3 (two fp16 planes, three terms: half the MFMAs), see Wgrad512Args
template <int AR = 0>
__device__ __forceinline__ void wgrad512_body(const Wgrad512Args& a, const int bid) {
  constexpr int NP = AR == 1 ? 2 : 3;                        // planes per operand
  constexpr int kFrags = (4 + 8) * NP, kSlab = 2 * kFrags * 1024;      // fragments per k16 step, bytes per slab buffer
  if (a.gate && *a.gate == 0) return;
  if (a.skip && *a.skip != 0) return;
  float sy = 1.0f, inv_sy = 1.0f;
  if constexpr (AR == 1) {
    if (a.amax_dy) {
      const unsigned e = (*a.amax_dy >> 23) & 0xffu;
      if (e >= 32u && e < 255u) {
        sy = __uint_as_float((268u - e) << 23);
        inv_sy = __uint_as_float((e - 14u) << 23);
      }
    }
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char* lds_ptr;
  typedef __attribute__((address_space(3))) bf8w* lds_bf8;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  // workgroup b: tile (b / 8) % 8, chunk b % 8 + 8 (b / 64) -- the 8 tiles of one chunk read the same rows and run on one XCD
  const int tile = (bid >> 3) & 7, chunk = (bid & 7) + 8 * (bid >> 6);
  const int ft = tile >> 1, kt2 = tile & 1;                  // f range [128 ft, +128), k range [256 kt2, +256)
  const long long m_begin = (long long)chunk * a.rows_per_chunk;
  long long m_end = m_begin + a.rows_per_chunk;
  if (m_end > a.M) m_end = a.M;
  if (m_begin >= a.M) return;
  const int n_slabs = (int)((m_end - m_begin + 31) / 32);

  // ---- staging: wave w owns columns [64 w, +64) of the tile's x range and [32 w, +32) of its dy range, all 32 rows of a slab.
  // Request j (0..7) of x: lanes 16 g .. 16 g + 15 read row 8 g + j (g = row group = (step, half)), 4 columns each;
  // of dy: the same rows, 2 columns each.  After the 8 requests a lane holds rows 8 g .. 8 g + 7 of its columns.
  const int g = lane >> 4, li = lane & 15;
  f32x4 xr[8];
  f32x2w yr[8];
  // Buffer loads: a descriptor per operand over the chunk's rows (base and size in scalar registers), the lane's part of the offset in one
  // register for the whole kernel, the (slab, request) part as the scalar offset -- no vector-ALU address arithmetic per load, and rows
  // past the chunk read as zeros through the descriptor's range check instead of a compare and six selects per row (the staging's other
  // vector-ALU work -- addresses, selects -- was as much as the conversion it exists for).
  const unsigned x_bytes = (unsigned)((m_end - m_begin) * (long long)a.ldx * 4), y_bytes = (unsigned)((m_end - m_begin) * (long long)a.ldy * 4);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X + (size_t)m_begin * a.ldx), 0, (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dY + (size_t)m_begin * a.ldy), 0, (int)y_bytes, 0x00020000);
  const unsigned xvoff = (unsigned)(8 * g) * (unsigned)a.ldx * 4u + (unsigned)(256 * kt2 + 64 * wave + 4 * li) * 4u;
  const unsigned yvoff = (unsigned)(8 * g) * (unsigned)a.ldy * 4u + (unsigned)(128 * ft + 32 * wave + 2 * li) * 4u;
  typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
  auto request = [&](int j, long long m0) {                  // rows of the slab that starts at m0
    const unsigned row = (unsigned)(m0 - m_begin) + (unsigned)j;               // (+ 8 g in the lane's offset)
    xr[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xvoff, row * (unsigned)a.ldx * 4u, 0));
    yr[j] = __builtin_bit_cast(f32x2w, __builtin_amdgcn_raw_buffer_load_b64(yrs, yvoff, row * (unsigned)a.ldy * 4u, 0));
  };
  // LDS slots of this lane's fragments inside a slab buffer: step s = g >> 1, lane' = column & 31 + 32 (g & 1)
  lds_ptr sbase = (lds_ptr)smem + (g >> 1) * (kFrags * 1024) + (32 * (g & 1)) * 16;
  // x columns 64 w + 4 li + c -> B tile 2 w + (li >> 3), lane' += 4 (li & 7) + c;   dy columns 32 w + 2 li + c -> A tile w, lane' += 2 li + c
  const int xslot = ((4 + 2 * wave + (li >> 3)) * NP) * 1024 + (4 * (li & 7)) * 16;
  const int yslot = (wave * NP) * 1024 + (2 * li) * 16;
  const int relu_floor = a.relu_x ? 0 : (int)0x80000000;
  float rs0 = 0.0f, rs1 = 0.0f;                              // row sums of this lane's two dy columns (bias gradient)
  bf8w sp0, sp1, sp2;                                        // the lane-fragment being converted (two halves of 4 rows)
  auto split_half = [&](const float (&v)[4], int half) {
    if constexpr (AR == 1) {                                 // fp16 hi / lo in single instructions (train_lin512.hip, the inference kernels' idiom)
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      u4 w0 = __builtin_bit_cast(u4, sp0), w1 = __builtin_bit_cast(u4, sp1);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned h = cvt_pk_f16_w(v[2 * j], v[2 * j + 1]);
        w0[2 * half + j] = h;
        w1[2 * half + j] = cvt_pk_f16_w(resid_lo_w(h, v[2 * j]), resid_hi_w(h, v[2 * j + 1]));
      }
      sp0 = __builtin_bit_cast(bf8w, w0);
      sp1 = __builtin_bit_cast(bf8w, w1);
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const __bf16 a0 = (__bf16)v[j];
      const float r1 = v[j] - (float)a0;
      const __bf16 a1 = (__bf16)r1;
      sp0[4 * half + j] = a0;
      sp1[4 * half + j] = a1;
      sp2[4 * half + j] = (__bf16)(r1 - (float)a1);
    }
  };
  auto store_frag = [&](lds_ptr d) {
    *(lds_bf8)(d) = sp0;
    *(lds_bf8)(d + 1024) = sp1;
    if constexpr (NP == 3) *(lds_bf8)(d + 2048) = sp2;
  };
  // unit i (0..5): columns 0..3 of the lane's x block, then 0..1 of its dy block; half 0 / 1 = rows 0..3 / 4..7 (+ the store)
  auto stash_half = [&](int buf, int i, int half) {
    float v[4];
    if (i < 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j)             // relu as an integer maximum with a uniform floor (0, or INT_MIN = the identity): one instruction
        v[j] = __int_as_float(max(__float_as_int(xr[4 * half + j][i]), relu_floor));
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = yr[4 * half + j][i - 4];
      const float sum = (v[0] + v[1]) + (v[2] + v[3]);
      if (i == 4) rs0 += sum; else rs1 += sum;
      if constexpr (AR == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= sy;
      }
    }
    split_half(v, half);
    if (half == 1) store_frag(sbase + buf * kSlab + (i < 4 ? xslot + i * 16 : yslot + (i - 4) * 16));
  };
  // ---- prologue: slab 0 staged, slab 1 requested
#pragma unroll
  for (int j = 0; j < 8; ++j) request(j, m_begin);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    stash_half(0, i, 0);
    stash_half(0, i, 1);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) request(j, m_begin + 32);      // (past the chunk: zeros)
  __syncthreads();

  const int wf = wave >> 1, wk = wave & 1;                    // this wave's 64 f x 128 k quarter of the tile
  f32x16w acc[2][4];
#pragma unroll
  for (int fi = 0; fi < 2; ++fi)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[fi][kt][e] = 0.0f;
  lds_ptr lbase = (lds_ptr)smem + lane * 16;
#pragma nounroll
  for (int slab = 0; slab < n_slabs; ++slab) {
    const int buf = slab & 1;
    const long long m_next2 = m_begin + 32ll * (slab + 2);    // the slab requested while this one multiplies
    lds_ptr rb = lbase + buf * kSlab;
    asm volatile("" : "+v"(rb));
    wfor<2>([&](auto S) {
      constexpr int s = decltype(S)::value;
      bf8w af[2][NP], bfr[2][NP];
#pragma unroll
      for (int fi = 0; fi < 2; ++fi)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) af[fi][pl] = *(lds_bf8)(rb + (s * kFrags + (2 * wf + fi) * NP + pl) * 1024);
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) bfr[0][pl] = *(lds_bf8)(rb + (s * kFrags + (4 + 4 * wk) * NP + pl) * 1024);
      wfor<8>([&](auto G) {
        constexpr int gi = decltype(G)::value, kt = gi >> 1, fi = gi & 1;
        constexpr int u = s * 8 + gi;                        // 16 groups of 6 MFMAs per slab
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (fi == 0 && kt + 1 < 4) {
#pragma unroll
          for (int pl = 0; pl < NP; ++pl)
            bfr[(kt + 1) & 1][pl] = *(lds_bf8)(rb + (s * kFrags + (4 + 4 * wk + kt + 1) * NP + pl) * 1024);
        }
        // staging side task: the next slab's six lane-fragment columns, each in two halves, groups 1..12; requests re-armed in group 13
        if constexpr (u >= 1 && u <= 12) stash_half(buf ^ 1, (u - 1) >> 1, (u - 1) & 1);
        if constexpr (u == 13) {
#pragma unroll
          for (int j = 0; j < 8; ++j) request(j, m_next2);
        }
        // smallest terms first
        if constexpr (AR == 1) {
          const bf8w b0 = bfr[kt & 1][0], b1 = bfr[kt & 1][1];
          DINER_WG_MFMA_F16(acc[fi][kt], af[fi][1], b0);
          DINER_WG_MFMA_F16(acc[fi][kt], af[fi][0], b1);
          DINER_WG_MFMA_F16(acc[fi][kt], af[fi][0], b0);
        } else {
          const bf8w b0 = bfr[kt & 1][0], b1 = bfr[kt & 1][1], b2 = bfr[kt & 1][NP - 1];
          DINER_WG_MFMA(acc[fi][kt], af[fi][NP - 1], b0);
          DINER_WG_MFMA(acc[fi][kt], af[fi][0], b2);
          DINER_WG_MFMA(acc[fi][kt], af[fi][1], b1);
          DINER_WG_MFMA(acc[fi][kt], af[fi][1], b0);
          DINER_WG_MFMA(acc[fi][kt], af[fi][0], b1);
          DINER_WG_MFMA(acc[fi][kt], af[fi][0], b0);
        }
        if constexpr (u >= 1 && u <= 12) {                   // the conversion between the MFMAs: <= 7 vector-ALU slots per 32-clock MFMA
#pragma unroll
          for (int i = 0; i < (AR == 1 ? 3 : 6); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x002, AR == 1 ? 8 : 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
        }
        asm volatile("" : "+a"(acc[fi][kt]));
      });
    });
    __syncthreads();                                         // slab buffer `buf` is free, the next one is complete
  }

  // ---- epilogue: D layout of a 32 x 32 tile: lane holds column (k) = lane & 31, rows (f) = 8 (e >> 2) + 4 (lane >> 5) + (e & 3)
#pragma unroll
  for (int fi = 0; fi < 2; ++fi)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int k = 256 * kt2 + 32 * (4 * wk + kt) + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int f = 128 * ft + 32 * (2 * wf + fi) + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
        const float val = AR == 1 ? acc[fi][kt][e] * inv_sy : acc[fi][kt][e];
        if (a.part) a.part[((size_t)chunk * 512 + f) * 512 + k] = val;
        else atomicAdd(a.dW + (size_t)f * 512 + k, val);
      }
    }
  if (a.db && kt2 == 0) {                                    // one k tile column of workgroups adds the row sums of dy
    const int f = 128 * ft + 32 * wave + 2 * li;
    if (a.part) {                                            // the four row groups of a column summed across lanes, one plain store
      rs0 += __shfl_xor(rs0, 16); rs0 += __shfl_xor(rs0, 32);
      rs1 += __shfl_xor(rs1, 16); rs1 += __shfl_xor(rs1, 32);
      if (lane < 16) {
        float* pdb = a.part + (size_t)kWgMaxChunks * 512 * 512 + (size_t)chunk * 512;
        pdb[f] = rs0;
        pdb[f + 1] = rs1;
      }
    } else {
      atomicAdd(a.db + f, rs0);
      atomicAdd(a.db + f + 1, rs1);
    }
  }
}

// ---- round 4: the f16x3 weight gradient on 256 (f) x 256 (k) tiles.  At three MFMAs per product the 128 x 256-tile body above is bound by its
// STAGING (48 values per thread and 32-row slab to convert, for 96 MFMAs per wave: the conversion no longer fits in the MFMAs' shadow); a
// 256 x 256 tile converts 64 values for 192 MFMAs -- a third less conversion per MFMA.  4 tiles x up to 64 row chunks = 256 workgroups (the 4
// tiles of a chunk read the same rows and sit on one XCD); waves 2 x 2 over the tile: 128 f x 128 k each = 4 x 4 accumulator tiles = 256
// registers; dy is staged like x (4 columns per lane); slabs of 64 KB ([A: 8 f tiles | B: 8 k tiles][hi, lo] per k16 step), two buffers.
__device__ __forceinline__ void wgrad512_body_wide(const Wgrad512Args& a, const int bid) {
  constexpr int NP = 2, kFrags = 16 * NP, kSlab = 2 * kFrags * 1024;
  if (a.gate && *a.gate == 0) return;
  if (a.skip && *a.skip != 0) return;
  float sy = 1.0f, inv_sy = 1.0f;
  if (a.amax_dy) {
    const unsigned e = (*a.amax_dy >> 23) & 0xffu;
    if (e >= 32u && e < 255u) {
      sy = __uint_as_float((268u - e) << 23);
      inv_sy = __uint_as_float((e - 14u) << 23);
    }
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char* lds_ptr;
  typedef __attribute__((address_space(3))) bf8w* lds_bf8;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  // workgroup b: tile (b / 8) % 4, chunk b % 8 + 8 (b / 32)
  const int tile = (bid >> 3) & 3, chunk = (bid & 7) + 8 * (bid >> 5);
  const int ft = tile >> 1, kt2 = tile & 1;                  // f range [256 ft, +256), k range [256 kt2, +256)
  const long long m_begin = (long long)chunk * a.rows_per_chunk;
  long long m_end = m_begin + a.rows_per_chunk;
  if (m_end > a.M) m_end = a.M;
  if (m_begin >= a.M) return;
  const int n_slabs = (int)((m_end - m_begin + 31) / 32);
  // staging: wave w owns columns [64 w, +64) of the tile's x range AND of its dy range, all 32 rows of a slab; request j (0..7): lanes
  // 16 g .. 16 g + 15 read row 8 g + j, 4 columns each -- after the 8 requests a lane holds rows 8 g .. 8 g + 7 of its 4 + 4 columns
  const int g = lane >> 4, li = lane & 15;
  f32x4 xr[8], yr[8];
  const unsigned x_bytes = (unsigned)((m_end - m_begin) * (long long)a.ldx * 4), y_bytes = (unsigned)((m_end - m_begin) * (long long)a.ldy * 4);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X + (size_t)m_begin * a.ldx), 0, (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dY + (size_t)m_begin * a.ldy), 0, (int)y_bytes, 0x00020000);
  const unsigned xvoff = (unsigned)(8 * g) * (unsigned)a.ldx * 4u + (unsigned)(256 * kt2 + 64 * wave + 4 * li) * 4u;
  const unsigned yvoff = (unsigned)(8 * g) * (unsigned)a.ldy * 4u + (unsigned)(256 * ft + 64 * wave + 4 * li) * 4u;
  typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
  auto request = [&](int j, long long m0) {
    const unsigned row = (unsigned)(m0 - m_begin) + (unsigned)j;
    xr[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xvoff, row * (unsigned)a.ldx * 4u, 0));
    yr[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yrs, yvoff, row * (unsigned)a.ldy * 4u, 0));
  };
  // LDS slots: step s = g >> 1, lane' = column & 31 + 32 (g & 1); columns 64 w + 4 li + c -> tile 2 w + (li >> 3), lane' += 4 (li & 7) + c
  lds_ptr sbase = (lds_ptr)smem + (g >> 1) * (kFrags * 1024) + (32 * (g & 1)) * 16;
  const int yslot = ((2 * wave + (li >> 3)) * NP) * 1024 + (4 * (li & 7)) * 16;
  const int xslot = ((8 + 2 * wave + (li >> 3)) * NP) * 1024 + (4 * (li & 7)) * 16;
  const int relu_floor = a.relu_x ? 0 : (int)0x80000000;
  float rs[4] = {0.0f, 0.0f, 0.0f, 0.0f};                   // row sums of this lane's four dy columns (bias gradient)
  u32x4w sp0, sp1;                                           // the lane-fragment being converted (two halves of 4 rows): hi / lo words
  // unit i (0..7): columns 0..3 of the lane's x block, then 0..3 of its dy block; half 0 / 1 = rows 0..3 / 4..7 (+ the store)
  auto stash_half = [&](int buf, int i, int half) {
    float v[4];
    if (i < 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = __int_as_float(max(__float_as_int(xr[4 * half + j][i]), relu_floor));
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = yr[4 * half + j][i - 4];
      rs[i - 4] += (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] *= sy;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned h = cvt_pk_f16_w(v[2 * j], v[2 * j + 1]);
      sp0[2 * half + j] = h;
      sp1[2 * half + j] = cvt_pk_f16_w(resid_lo_w(h, v[2 * j]), resid_hi_w(h, v[2 * j + 1]));
    }
    if (half == 1) {
      lds_ptr d = sbase + buf * kSlab + (i < 4 ? xslot + i * 16 : yslot + (i - 4) * 16);
      *(lds_bf8)(d) = __builtin_bit_cast(bf8w, sp0);
      *(lds_bf8)(d + 1024) = __builtin_bit_cast(bf8w, sp1);
    }
  };
#pragma unroll
  for (int j = 0; j < 8; ++j) request(j, m_begin);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    stash_half(0, i, 0);
    stash_half(0, i, 1);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) request(j, m_begin + 32);      // (past the chunk: zeros)
  __syncthreads();

  const int wf = wave >> 1, wk = wave & 1;                    // this wave's 128 f x 128 k quarter of the tile
  f32x16w acc[4][4];
#pragma unroll
  for (int fi = 0; fi < 4; ++fi)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[fi][kt][e] = 0.0f;
  lds_ptr lbase = (lds_ptr)smem + lane * 16;
#pragma nounroll
  for (int slab = 0; slab < n_slabs; ++slab) {
    const int buf = slab & 1;
    const long long m_next2 = m_begin + 32ll * (slab + 2);
    lds_ptr rb = lbase + buf * kSlab;
    asm volatile("" : "+v"(rb));
    wfor<2>([&](auto S) {
      constexpr int s = decltype(S)::value;
      bf8w af[4][NP], bfr[2][NP];
#pragma unroll
      for (int fi = 0; fi < 4; ++fi)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) af[fi][pl] = *(lds_bf8)(rb + (s * kFrags + (4 * wf + fi) * NP + pl) * 1024);
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) bfr[0][pl] = *(lds_bf8)(rb + (s * kFrags + (8 + 4 * wk) * NP + pl) * 1024);
      wfor<16>([&](auto G) {
        constexpr int gi = decltype(G)::value, kt = gi >> 2, fi = gi & 3;
        constexpr int u = s * 16 + gi;                       // 32 groups of 3 MFMAs per slab
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (fi == 0 && kt + 1 < 4) {
#pragma unroll
          for (int pl = 0; pl < NP; ++pl) bfr[(kt + 1) & 1][pl] = *(lds_bf8)(rb + (s * kFrags + (8 + 4 * wk + kt + 1) * NP + pl) * 1024);
        }
        // staging side task: the next slab's eight lane-fragment columns, each in two halves, groups 1..16; requests re-armed in group 17
        if constexpr (u >= 1 && u <= 16) stash_half(buf ^ 1, (u - 1) >> 1, (u - 1) & 1);
        if constexpr (u == 17) {
#pragma unroll
          for (int j = 0; j < 8; ++j) request(j, m_next2);
        }
        const bf8w b0 = bfr[kt & 1][0], b1 = bfr[kt & 1][1];
        DINER_WG_MFMA_F16(acc[fi][kt], af[fi][1], b0);       // smallest terms first
        DINER_WG_MFMA_F16(acc[fi][kt], af[fi][0], b1);
        DINER_WG_MFMA_F16(acc[fi][kt], af[fi][0], b0);
        if constexpr (u >= 1 && u <= 16) {
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
        }
        asm volatile("" : "+a"(acc[fi][kt]));
      });
    });
    __syncthreads();
  }
  // epilogue: D layout of a 32 x 32 tile: lane holds column (k) = lane & 31, rows (f) = 8 (e >> 2) + 4 (lane >> 5) + (e & 3)
#pragma unroll
  for (int fi = 0; fi < 4; ++fi)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int k = 256 * kt2 + 32 * (4 * wk + kt) + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int f = 256 * ft + 32 * (4 * wf + fi) + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
        const float val = acc[fi][kt][e] * inv_sy;
        if (a.part) a.part[((size_t)chunk * 512 + f) * 512 + k] = val;
        else atomicAdd(a.dW + (size_t)f * 512 + k, val);
      }
    }
  if (a.db && kt2 == 0) {
    const int f = 256 * ft + 64 * wave + 4 * li;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      rs[c] += __shfl_xor(rs[c], 16);
      rs[c] += __shfl_xor(rs[c], 32);
    }
    if (lane < 16) {
      if (a.part) {
        float* pdb = a.part + (size_t)kWgMaxChunks * 512 * 512 + (size_t)chunk * 512;
#pragma unroll
        for (int c = 0; c < 4; ++c) pdb[f + c] = rs[c];
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) atomicAdd(a.db + f + c, rs[c]);
      }
    }
  }
}

// ---- round 5: the same 256 x 256 tiles on EIGHT waves (512 threads, two waves per SIMD, 256 registers each).  One wave per SIMD leaves the
// matrix pipe idle whenever that wave converts, waits at the slab barrier or for a fragment (both 512-layer bodies run at ~0.47 of the
// pipe, profiles/r05_train_bwd_split.txt); two waves fill each other's gaps (the micro-benchmark of the inference kernels: 0.98 of the pipe
// against 0.8).  Waves 2 (f) x 4 (k): 128 f x 64 k each = 4 x 2 accumulator tiles = 128 registers.  Staging: waves 0..3 convert the x
// operand (columns [64 w, +64) of the tile's k range), waves 4..7 the dy operand (columns [64 (w - 4), +64) of its f range) -- the lane
// mapping, the LDS layout (64 KB slabs, two buffers) and the partial-tile layout of wgrad512_body_wide; 32 values per thread and slab for
// 48 MFMAs.  Its own kernel (k_wgrad512_w8: 512 threads); the data gradient of the layer is the launch in front of it.
#ifdef DINER_L512_PROF
__device__ unsigned long long g_wg_prof[8];      // wgrad512_body_w8: [0] clocks per wave, [1] slab sections, [2] slab barriers, [3] slabs, [4] waves
#endif
__device__ __forceinline__ void wgrad512_body_w8(const Wgrad512Args& a, const int bid) {
  constexpr int NP = 2, kFrags = 16 * NP, kSlab = 2 * kFrags * 1024;
  if (a.gate && *a.gate == 0) return;
  if (a.skip && *a.skip != 0) return;
  float sy = 1.0f, inv_sy = 1.0f;
  if (a.amax_dy) {
    const unsigned e = (*a.amax_dy >> 23) & 0xffu;
    if (e >= 32u && e < 255u) {
      sy = __uint_as_float((268u - e) << 23);
      inv_sy = __uint_as_float((e - 14u) << 23);
    }
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char* lds_ptr;
  typedef __attribute__((address_space(3))) bf8w* lds_bf8;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int tile = (bid >> 3) & 3, chunk = (bid & 7) + 8 * (bid >> 5);      // as wgrad512_body_wide
  const int ft = tile >> 1, kt2 = tile & 1;
  const long long m_begin = (long long)chunk * a.rows_per_chunk;
  long long m_end = m_begin + a.rows_per_chunk;
  if (m_end > a.M) m_end = a.M;
  if (m_begin >= a.M) return;
  const int n_slabs = (int)((m_end - m_begin + 31) / 32);
  const bool is_y = wave >= 4;                               // this wave stages dy (else x)
  const int ws = wave & 3;                                   // its 64-column block of that operand
  const int g = lane >> 4, li = lane & 15;
  f32x4 vr[8];                                               // rows 8 g .. 8 g + 7 of the lane's 4 columns
  const int ld = is_y ? a.ldy : a.ldx;
  const float* src = is_y ? a.dY : a.X;
  const unsigned s_bytes = (unsigned)((m_end - m_begin) * (long long)ld * 4);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src + (size_t)m_begin * ld), 0, (int)s_bytes, 0x00020000);
  const unsigned voff = (unsigned)(8 * g) * (unsigned)ld * 4u + (unsigned)(256 * (is_y ? ft : kt2) + 64 * ws + 4 * li) * 4u;
  auto request = [&](int j, long long m0) {
    const unsigned row = (unsigned)(m0 - m_begin) + (unsigned)j;
    vr[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, row * (unsigned)ld * 4u, 0));
  };
  lds_ptr sbase = (lds_ptr)smem + (g >> 1) * (kFrags * 1024) + (32 * (g & 1)) * 16;
  const int slot = (((is_y ? 0 : 8) + 2 * ws + (li >> 3)) * NP) * 1024 + (4 * (li & 7)) * 16;
  // Bank conflicts of the staging writes: one store instruction puts column i of every lane's 4-column block into its fragment -- 16-byte
  // slots 4 (li & 7) + i, a stride of 64 bytes: the 64 lanes hit a quarter of the banks (the round-4 counters: 64-70 % of the LDS-active
  // cycles were conflict cycles), and with eight waves reading 96 KB of fragments per k16 step on top of 4 x 32 KB-equivalents of
  // conflicted writes the LDS, not the matrix pipe, sets the pace.  Slots are swizzled inside every 1 KB fragment: slot L lives at
  // L ^ ((L >> 3) & 3) -- the eight lanes of a row group then cover all eight 16-byte bank groups, and a reader's 64 lanes still read 64
  // distinct slots.  For the writer (L >> 3) & 3 = (li >> 1) & 3 whatever i: the slot offset of column i is (16 i) ^ sw16.
  const int sw16 = ((li >> 1) & 3) * 16;
  const int relu_floor = (!is_y && a.relu_x) ? 0 : (int)0x80000000;
  const float scale = is_y ? sy : 1.0f;
  float rs[4] = {0.0f, 0.0f, 0.0f, 0.0f};                   // dy waves: row sums of the lane's four columns (bias gradient, unscaled)
  typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
  u32x4w sp0, sp1;
  auto stash_half = [&](int buf, int i, int half) {          // column i (0..3) of the lane's block, rows 0..3 / 4..7 (+ the store)
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __int_as_float(max(__float_as_int(vr[4 * half + j][i]), relu_floor));
    rs[i] += (v[0] + v[1]) + (v[2] + v[3]);                  // (x waves: never read)
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] *= scale;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned h = cvt_pk_f16_w(v[2 * j], v[2 * j + 1]);
      sp0[2 * half + j] = h;
      sp1[2 * half + j] = cvt_pk_f16_w(resid_lo_w(h, v[2 * j]), resid_hi_w(h, v[2 * j + 1]));
    }
    if (half == 1) {
      lds_ptr d = sbase + buf * kSlab + slot + ((i * 16) ^ sw16);
      *(lds_bf8)(d) = __builtin_bit_cast(bf8w, sp0);
      *(lds_bf8)(d + 1024) = __builtin_bit_cast(bf8w, sp1);
    }
  };
#pragma unroll
  for (int j = 0; j < 8; ++j) request(j, m_begin);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    stash_half(0, i, 0);
    stash_half(0, i, 1);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) request(j, m_begin + 32);      // (past the chunk: zeros)
  __syncthreads();

#ifdef DINER_L512_PROF
  unsigned long long pw_mf = 0, pw_bar = 0;
  const unsigned long long pw_t0 = __builtin_readcyclecounter();
#endif
  const int wf = wave >> 2, wk = wave & 3;                    // this wave's 128 f x 64 k part of the tile
  f32x16w acc[4][2];
#pragma unroll
  for (int fi = 0; fi < 4; ++fi)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[fi][kt][e] = 0.0f;
  lds_ptr lbase = (lds_ptr)smem + (lane ^ ((lane >> 3) & 3)) * 16;      // (the swizzled slot of fragment lane `lane`)
#pragma nounroll
  for (int slab = 0; slab < n_slabs; ++slab) {
#ifdef DINER_L512_PROF
    const unsigned long long pw_a = __builtin_readcyclecounter();
#endif
    const int buf = slab & 1;
    const long long m_next2 = m_begin + 32ll * (slab + 2);
    lds_ptr rb = lbase + buf * kSlab;
    asm volatile("" : "+v"(rb));
    wfor<2>([&](auto S) {
      constexpr int s = decltype(S)::value;
      bf8w af[4][NP], bfr[2][NP];
#pragma unroll
      for (int fi = 0; fi < 4; ++fi)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) af[fi][pl] = *(lds_bf8)(rb + (s * kFrags + (4 * wf + fi) * NP + pl) * 1024);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) bfr[kt][pl] = *(lds_bf8)(rb + (s * kFrags + (8 + 2 * wk + kt) * NP + pl) * 1024);
      wfor<8>([&](auto G) {
        constexpr int gi = decltype(G)::value, kt = gi >> 2, fi = gi & 3;
        constexpr int u = s * 8 + gi;                        // 16 groups of 3 MFMAs per slab
        __builtin_amdgcn_sched_barrier(0);
        // staging side task: the next slab's four lane-fragment columns, each in two halves, groups 1..8; requests re-armed in group 9
        // (round 6, measured and not kept: both halves of a column per group in groups 1..4 and the requests in group 5 -- 800 clocks more lead
        // time for the loads -- is SLOWER: 4.78 k against 4.63 k clocks per slab, 125.4 against 124.1 ms per step, profiles/r06_train_wgrad_phase_timer.txt)
        if constexpr (u >= 1 && u <= 8) stash_half(buf ^ 1, (u - 1) >> 1, (u - 1) & 1);
        if constexpr (u == 9) {
#pragma unroll
          for (int j = 0; j < 8; ++j) request(j, m_next2);
        }
        const bf8w b0 = bfr[kt][0], b1 = bfr[kt][1];
        DINER_WG_MFMA_F16(acc[fi][kt], af[fi][1], b0);       // smallest terms first
        DINER_WG_MFMA_F16(acc[fi][kt], af[fi][0], b1);
        DINER_WG_MFMA_F16(acc[fi][kt], af[fi][0], b0);
        if constexpr (u >= 1 && u <= 8) {
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
        }
        asm volatile("" : "+a"(acc[fi][kt]));
      });
    });
#ifdef DINER_L512_PROF
    const unsigned long long pw_b = __builtin_readcyclecounter();
    __syncthreads();
    pw_mf += pw_b - pw_a;
    pw_bar += __builtin_readcyclecounter() - pw_b;
#else
    __syncthreads();
#endif
  }
#ifdef DINER_L512_PROF
  if (lane == 0) {
    atomicAdd(&g_wg_prof[0], __builtin_readcyclecounter() - pw_t0);
    atomicAdd(&g_wg_prof[1], pw_mf);
    atomicAdd(&g_wg_prof[2], pw_bar);
    atomicAdd(&g_wg_prof[3], (unsigned long long)n_slabs);
    atomicAdd(&g_wg_prof[4], 1ull);
  }
#endif
#pragma unroll
  for (int fi = 0; fi < 4; ++fi)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int k = 256 * kt2 + 32 * (2 * wk + kt) + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int f = 256 * ft + 32 * (4 * wf + fi) + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
        const float val = acc[fi][kt][e] * inv_sy;
        if (a.part) a.part[((size_t)chunk * 512 + f) * 512 + k] = val;
        else atomicAdd(a.dW + (size_t)f * 512 + k, val);
      }
    }
  if (a.db && kt2 == 0 && is_y) {
    const int f = 256 * ft + 64 * ws + 4 * li;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      rs[c] += __shfl_xor(rs[c], 16);
      rs[c] += __shfl_xor(rs[c], 32);
    }
    if (lane < 16) {
      if (a.part) {
        float* pdb = a.part + (size_t)kWgMaxChunks * 512 * 512 + (size_t)chunk * 512;
#pragma unroll
        for (int c = 0; c < 4; ++c) pdb[f + c] = rs[c];
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) atomicAdd(a.db + f + c, rs[c]);
      }
    }
  }
}

// dW[i] (+)= sum over the chunks of part[c][i]; db likewise from the partial row sums behind the tiles.  Four independent running sums:
// the 32 loads of a thread must not wait for each other (the pass read its 32 MB at 2 TB/s with one)
__global__ __launch_bounds__(256) void k_wgrad512_reduce(const float* __restrict__ part, int n_chunks, int overwrite, float* __restrict__ dW,
                                                         float* __restrict__ db) {
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= 512 * 512) return;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 s[4] = {overwrite ? zero : *reinterpret_cast<const f32x4*>(dW + i), zero, zero, zero};
  int c = 0;
  for (; c + 4 <= n_chunks; c += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(part + (size_t)(c + u) * 512 * 512 + i));
  }
  for (; c < n_chunks; ++c) s[0] += *reinterpret_cast<const f32x4*>(part + (size_t)c * 512 * 512 + i);
  *reinterpret_cast<f32x4*>(dW + i) = (s[0] + s[1]) + (s[2] + s[3]);
  if (db && i < 512) {
    const float* pdb = part + (size_t)kWgMaxChunks * 512 * 512;
    f32x4 t = overwrite ? zero : *reinterpret_cast<const f32x4*>(db + i);
    for (int c2 = 0; c2 < n_chunks; ++c2) t += *reinterpret_cast<const f32x4*>(pdb + (size_t)c2 * 512 + i);
    *reinterpret_cast<f32x4*>(db + i) = t;
  }
}

// the same for up to 13 layers in one launch (blockIdx.y = layer): the training step sums the partial tiles of all its 512 x 512 layers at
// the end of the backward instead of behind every product (13 launches of 13 us and their serialisation points -> one)
__global__ __launch_bounds__(256) void k_wgrad512_reduce_many(WgReduceJobs jobs, int overwrite) {
  const WgReduceJob j = jobs.job[blockIdx.y];
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= 512 * 512) return;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 s[4] = {overwrite ? zero : *reinterpret_cast<const f32x4*>(j.dW + i), zero, zero, zero};
  int c = 0;
  for (; c + 4 <= j.n_chunks; c += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(j.part + (size_t)(c + u) * 512 * 512 + i));
  }
  for (; c < j.n_chunks; ++c) s[0] += *reinterpret_cast<const f32x4*>(j.part + (size_t)c * 512 * 512 + i);
  *reinterpret_cast<f32x4*>(j.dW + i) = (s[0] + s[1]) + (s[2] + s[3]);
  if (j.db && i < 512) {
    const float* pdb = j.part + (size_t)kWgMaxChunks * 512 * 512;
    f32x4 t = overwrite ? zero : *reinterpret_cast<const f32x4*>(j.db + i);
    for (int c2 = 0; c2 < j.n_chunks; ++c2) t += *reinterpret_cast<const f32x4*>(pdb + (size_t)c2 * 512 + i);
    *reinterpret_cast<f32x4*>(j.db + i) = t;
  }
}

int wgrad512_reduce_many(const WgReduceJobs& jobs, int n, bool overwrite, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_wgrad512_reduce_many, dim3(256, n), dim3(256), 0, stream, jobs, overwrite ? 1 : 0);
  DINER_LAUNCH_OK();
  return 0;
}

size_t wgrad512_part_bytes() { return (size_t)kWgMaxChunks * (512 * 512 + 512) * sizeof(float); }


// ---- round 6: lin_in's weight / bias gradient, dW (512, 55) = dy^T feat, db = column sums of dy, over the step's 2.6 M per-view rows -----------
// The general bf16x6 GEMM served this product with 128 x 128 tiles of which half the columns are padding and four N-tiles that each re-read
// the encoded inputs: 3.1 ms per SB 4 x 4096 x 40 step at 2 TB/s for 6 GB of operands (profiles/r06_train_batched_kernel_stats.md).  Here: one
// persistent workgroup per row chunk, eight waves (two per SIMD); wave w owns features [64 w, 64 w + 64) x all 64 (padded) inputs = 4
// accumulator tiles; per 32-row slab every wave stages its 64 columns of dy (fp32 -> fp16 hi / lo of dy * 2^(14 - E), as wgrad512_body_w8)
// and 8 columns of feat (encoded inputs: |x| of a few units, no scaling); 24 MFMAs per wave and slab -- HBM-bound by a wide margin (72 KB per
// slab and CU for 1.5 k MFMA clocks).  Partial products go to dW / db with float atomics (both zeroed by the caller).
struct WgradInArgs {
  const float* dY;       // (M, ldy) gradient of lin_in's output
  const float* F;        // (M, ldf >= 64) encoded inputs, columns >= n_in ignored
  float* dW;             // (512, n_in) row-major, += (zeroed by the caller)
  float* db;             // (512), +=
  long long M, rows_per_chunk;
  int ldy, ldf, n_in;
  const unsigned* amax_dy;
};
constexpr int kWgInFrags = (16 + 2) * 2;                        // 1 KB fragments per k16 step: 16 feature tiles of dy + 2 input tiles, two planes each
constexpr int kWgInSlab = 2 * kWgInFrags * 1024;                // 72 KB per slab buffer
constexpr size_t kLdsBytesWgradIn = (size_t)2 * kWgInSlab;      // 144 KB

__global__ __launch_bounds__(512, 1) void k_wgrad_in_f16x3(WgradInArgs a) {
  float sy = 1.0f, inv_sy = 1.0f;
  if (a.amax_dy) {
    const unsigned e = (*a.amax_dy >> 23) & 0xffu;
    if (e >= 32u && e < 255u) {
      sy = __uint_as_float((268u - e) << 23);
      inv_sy = __uint_as_float((e - 14u) << 23);
    }
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char* lds_ptr;
  typedef __attribute__((address_space(3))) bf8w* lds_bf8;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long long m_begin = (long long)blockIdx.x * a.rows_per_chunk;
  long long m_end = m_begin + a.rows_per_chunk;
  if (m_end > a.M) m_end = a.M;
  if (m_begin >= a.M) return;
  const int n_slabs = (int)((m_end - m_begin + 31) / 32);
  const int g = lane >> 4, li = lane & 15;
  f32x4 vr[8];                                               // dy: rows 8 g .. 8 g + 7 of the lane's 4 columns (64 w + 4 li ..)
  float vf[8];                                               // feat: the same rows of column 8 w + li (lanes li < 8)
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dY + (size_t)m_begin * a.ldy), 0,
                                                                        (int)((m_end - m_begin) * (long long)a.ldy * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_f = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.F + (size_t)m_begin * a.ldf), 0,
                                                                        (int)((m_end - m_begin) * (long long)a.ldf * 4), 0x00020000);
  const unsigned voff_y = (unsigned)(8 * g) * (unsigned)a.ldy * 4u + (unsigned)(64 * wave + 4 * li) * 4u;
  const unsigned voff_f = (unsigned)(8 * g) * (unsigned)a.ldf * 4u + (unsigned)(8 * wave + (li & 7)) * 4u;
  auto request = [&](long long m0) {
    const unsigned row0 = (unsigned)(m0 - m_begin);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      vr[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_y, voff_y, (row0 + j) * (unsigned)a.ldy * 4u, 0));
      vf[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_f, voff_f, (row0 + j) * (unsigned)a.ldf * 4u, 0));
    }
  };
  // where the lane's values go: step s = g / 2, row half g % 2; dy column i of the lane: feature 64 w + 4 li + i -> tile 2 w + li / 8, fragment
  // lane 4 (li % 8) + i + 32 half; 16-byte slots swizzled inside every fragment as in wgrad512_body_w8 (slot L at L ^ ((L >> 3) & 3))
  lds_ptr sbase = (lds_ptr)smem + (g >> 1) * (kWgInFrags * 1024) + (32 * (g & 1)) * 16;
  const int slot_y = ((2 * wave + (li >> 3)) * 2) * 1024 + (4 * (li & 7)) * 16;
  const int sw16 = ((li >> 1) & 3) * 16;
  // feat column 8 w + li (li < 8): tile (8 w + li) / 32 = w / 4, fragment lane 8 (w % 4) + li + 32 half, swizzle ((L >> 3) & 3) = w % 4
  const int slot_f = ((16 + (wave >> 2)) * 2) * 1024 + (((8 * (wave & 3) + (li & 7)) ^ (wave & 3)) * 16);
  float rs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4w sp0, sp1;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = vr[4 * half + j][i];
        rs[i] += (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= sy;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const unsigned h = cvt_pk_f16_w(v[2 * j], v[2 * j + 1]);
          sp0[2 * half + j] = h;
          sp1[2 * half + j] = cvt_pk_f16_w(resid_lo_w(h, v[2 * j]), resid_hi_w(h, v[2 * j + 1]));
        }
      }
      lds_ptr d = sbase + buf * kWgInSlab + slot_y + ((i * 16) ^ sw16);
      *(lds_bf8)(d) = __builtin_bit_cast(bf8w, sp0);
      *(lds_bf8)(d + 1024) = __builtin_bit_cast(bf8w, sp1);
    }
    if (li < 8) {
      u32x4w sp0, sp1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned h = cvt_pk_f16_w(vf[2 * j], vf[2 * j + 1]);
        sp0[j] = h;
        sp1[j] = cvt_pk_f16_w(resid_lo_w(h, vf[2 * j]), resid_hi_w(h, vf[2 * j + 1]));
      }
      lds_ptr d = sbase + buf * kWgInSlab + slot_f;
      *(lds_bf8)(d) = __builtin_bit_cast(bf8w, sp0);
      *(lds_bf8)(d + 1024) = __builtin_bit_cast(bf8w, sp1);
    }
  };
  request(m_begin);
  stash(0);
  request(m_begin + 32);                                     // (past the chunk: zeros)
  __syncthreads();
  f32x16w acc[2][2];
#pragma unroll
  for (int fi = 0; fi < 2; ++fi)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[fi][kt][e] = 0.0f;
  lds_ptr lbase = (lds_ptr)smem + (lane ^ ((lane >> 3) & 3)) * 16;
#pragma nounroll
  for (int slab = 0; slab < n_slabs; ++slab) {
    const int buf = slab & 1;
    lds_ptr rb = lbase + buf * kWgInSlab;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf8w af[2][2], bfr[2][2];
#pragma unroll
      for (int fi = 0; fi < 2; ++fi)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) af[fi][pl] = *(lds_bf8)(rb + (s * kWgInFrags + (2 * wave + fi) * 2 + pl) * 1024);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) bfr[kt][pl] = *(lds_bf8)(rb + (s * kWgInFrags + (16 + kt) * 2 + pl) * 1024);
#pragma unroll
      for (int fi = 0; fi < 2; ++fi)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          DINER_WG_MFMA_F16(acc[fi][kt], af[fi][1], bfr[kt][0]);       // smallest terms first
          DINER_WG_MFMA_F16(acc[fi][kt], af[fi][0], bfr[kt][1]);
          DINER_WG_MFMA_F16(acc[fi][kt], af[fi][0], bfr[kt][0]);
        }
    }
    stash(buf ^ 1);                                          // the next slab (requested one slab ago) into the other buffer
    request(m_begin + 32ll * (slab + 2));
    __syncthreads();
  }
#pragma unroll
  for (int fi = 0; fi < 2; ++fi)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int k = 32 * kt + (lane & 31);
      if (k >= a.n_in) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int f = 64 * wave + 32 * fi + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
        atomicAdd(a.dW + (size_t)f * a.n_in + k, acc[fi][kt][e] * inv_sy);
      }
    }
  if (a.db) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      rs[c] += __shfl_xor(rs[c], 16);
      rs[c] += __shfl_xor(rs[c], 32);
    }
    if (lane < 16) {
#pragma unroll
      for (int c = 0; c < 4; ++c) atomicAdd(a.db + 64 * wave + 4 * li + c, rs[c]);
    }
  }
}

#ifdef DINER_L512_PROF
}  // namespace train
}  // namespace diner
extern "C" int diner_debug_wgrad_prof(unsigned long long* out8, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out8, HIP_SYMBOL(diner::train::g_wg_prof), 8 * sizeof(unsigned long long));
  if (reset) {
    unsigned long long z[8] = {};
    hipMemcpyToSymbol(HIP_SYMBOL(diner::train::g_wg_prof), z, sizeof(z));
  }
  return 0;
}
namespace diner {
namespace train {
#endif
}  // namespace train
}  // namespace diner
