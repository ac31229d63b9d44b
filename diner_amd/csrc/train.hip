// Training path of the field MLP and the compositor: un-fused forward that keeps the
// activations, and the backward pass, as plain building blocks driven by diner_amd/train.py:
//   * one general fp32 GEMM on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 products) with the epilogues the
//     ResnetFC backward needs: relu on either operand, bias, "+=" output, relu mask from a saved pre-activation,
//     split-K with atomics for the weight gradients;
//   * the per-(view, point) inputs (55 encoded features, 4 bilinear taps, interpolated latent) and the scatter-add of
//     the latent gradient through the same taps;
//   * view mean / its adjoint, the output activations and their adjoint, bias-gradient column sums;
//   * the adjoint of the compositor.
// Training batches are small (reference: 128 rays x 40 samples x 4 views = 20 k columns per object and step,
// configs/train_dtu.yaml:55-65), so these kernels are written for clarity: the fused inference kernels stay the fast path.
// Reference: ResnetFC.forward resnetfc.py:129-159, PixelNeRF.forward pixelnerf.py:55-145, NeRFRendererDGS.composite
// nerf_renderer.py:286-365, differentiated by torch autograd in DINER.calc_losses (diner.py:217-290).
#include <atomic>
#include "field_common.hpp"
#include "train_lin512.hpp"

namespace diner {
namespace train {

constexpr int BM = 64, BN = 64, BK = 16, PAD = 4;
enum : int { kTA = 1, kTB = 2, kReluA = 4, kReluB = 8, kAccum = 16, kAtomic = 32, kExact = 64,
             kNoXcdOrder = 128 /* internal: plain blockIdx tile order (A/B measurement, DINER_TRAIN_NO_XCD=1) */ };

struct GemmArgs {
  const float* A;       // op(A) is M x K: stored [M][lda] (or [K][lda] with kTA)
  const float* B;       // op(B) is K x N: stored [K][ldb] (or [N][ldb] with kTB)
  float* C;             // M x N, [M][ldc]
  const float* bias;    // N or null: added to every row
  const float* mask;    // M x N (ldc) or null: C *= (mask > 0)   (relu adjoint with the saved pre-activation)
  const float* resid;   // M x N (ldc) or null: added to the product (bf16x6 kernel only): C = resid + A B (+ bias) without a copy of resid
  float* rowsum;        // M or null (bf16x6 kernel, op(A) stored with kTA only): rowsum[m] += sum_k op(A)[m][k] (atomic) -- the bias gradient
                        // of a layer is the row sum of the dy^T operand of its weight-gradient product, read by that product anyway
  long long M;
  int N, K, lda, ldb, ldc, flags, k_chunk;      // k_chunk: K range per blockIdx.z (split-K, needs kAtomic)
  const int* gate;      // null, or: the launch does nothing unless *gate != 0 (bf16x6 kernel: the layer-wise repeat behind the fused training forward)
  const int* k_dev;     // null, or (bf16x6 kernel): the contraction runs over min(K, *k_dev) indices -- a row list whose length only the device knows
};

// C tile 64 x 64 per workgroup, four waves 2 x 2, each 32 x 32 = 2 x 2 MFMA tiles; operands staged k-major in LDS.
__global__ __launch_bounds__(256) void k_gemm(GemmArgs g) {
  __shared__ float As[BK][BM + PAD], Bs[BK][BN + PAD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long long m0 = (long long)blockIdx.y * BM;
  const int n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * g.k_chunk, kend = min(g.K, kbeg + g.k_chunk);
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool ta = g.flags & kTA, tb = g.flags & kTB, ra = g.flags & kReluA, rb = g.flags & kReluB;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    // 64 x 16 elements of each operand, four per thread; the fast index follows the storage order
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + 256 * e;
      int am, ak;
      if (ta) { am = idx & 63; ak = idx >> 6; } else { ak = idx & 15; am = idx >> 4; }
      float av = 0.0f;
      if (m0 + am < g.M && k0 + ak < kend)
        av = ta ? g.A[(size_t)(k0 + ak) * g.lda + (m0 + am)] : g.A[(size_t)(m0 + am) * g.lda + (k0 + ak)];
      As[ak][am] = ra ? fmaxf(av, 0.0f) : av;
      int bn, bk;
      if (tb) { bk = idx & 15; bn = idx >> 4; } else { bn = idx & 63; bk = idx >> 6; }
      float bv = 0.0f;
      if (n0 + bn < g.N && k0 + bk < kend)
        bv = tb ? g.B[(size_t)(n0 + bn) * g.ldb + (k0 + bk)] : g.B[(size_t)(k0 + bk) * g.ldb + (n0 + bn)];
      Bs[bk][bn] = rb ? fmaxf(bv, 0.0f) : bv;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[kk + (lane >> 4)][32 * wm + 16 * i + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[kk + (lane >> 4)][32 * wn + 16 * j + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // D layout: lane holds rows 4 (lane >> 4) .. + 3 of column lane & 15
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + 32 * wn + 16 * j + (lane & 15);
      if (n >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long m = m0 + 32 * wm + 16 * i + 4 * (lane >> 4) + r;
        if (m >= g.M) continue;
        float v = acc[i][j][r];
        float* c = g.C + (size_t)m * g.ldc + n;
        if (g.bias && blockIdx.z == 0) v += g.bias[n];
        if (g.mask && !(g.mask[(size_t)m * g.ldc + n] > 0.0f)) v = 0.0f;
        if (g.flags & kAtomic) atomicAdd(c, v);
        else if (g.flags & kAccum) *c += v;
        else *c = v;
      }
    }
}

// The same GEMM with 128 x 128 x 16 tiles for the layer-sized products (N, M >= 128): each wave owns 64 x 64 = 4 x 4 MFMA
// tiles, so one k-step of 4 costs 8 LDS operand reads for 16 MFMAs (the 64 x 64 kernel: 4 for 4), and the next k-tile's
// global loads are issued before the current tile's MFMAs.
constexpr int BM2 = 128, BN2 = 128;
__global__ __launch_bounds__(256) void k_gemm128(GemmArgs g) {
  __shared__ float As[2][BK][BM2 + PAD], Bs[2][BK][BN2 + PAD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long long m0 = (long long)blockIdx.y * BM2;
  const int n0 = blockIdx.x * BN2;
  const int kbeg = blockIdx.z * g.k_chunk, kend = min(g.K, kbeg + g.k_chunk);
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool ta = g.flags & kTA, tb = g.flags & kTB, ra = g.flags & kReluA, rb = g.flags & kReluB;
  // Operand staging: two float4 per thread and operand along the storage's contiguous dimension when the whole tile is
  // in range and 16 B aligned (every layer-sized product of the MLP), else the element-wise path below.
  const bool vec_a = ((g.lda & 3) == 0) && ((reinterpret_cast<size_t>(g.A) & 15) == 0) && m0 + BM2 <= g.M;
  const bool vec_b = ((g.ldb & 3) == 0) && ((reinterpret_cast<size_t>(g.B) & 15) == 0) && n0 + BN2 <= g.N;
  float ra_v[8], rb_v[8];
  auto fetch = [&](int k0) {                       // 128 x 16 elements of each operand, eight per thread
    const bool kfull = k0 + BK <= kend;
    if (vec_a && kfull) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int idx = tid + 256 * e;             // float4 index
        const float* src = ta ? g.A + (size_t)(k0 + (idx >> 5)) * g.lda + (m0 + 4 * (idx & 31))
                              : g.A + (size_t)(m0 + (idx >> 2)) * g.lda + (k0 + 4 * (idx & 3));
        const f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
        for (int c = 0; c < 4; ++c) ra_v[4 * e + c] = ra ? fmaxf(v[c], 0.0f) : v[c];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = tid + 256 * e;
        int am, ak;
        if (ta) { am = idx & 127; ak = idx >> 7; } else { ak = idx & 15; am = idx >> 4; }
        float av = 0.0f;
        if (m0 + am < g.M && k0 + ak < kend)
          av = ta ? g.A[(size_t)(k0 + ak) * g.lda + (m0 + am)] : g.A[(size_t)(m0 + am) * g.lda + (k0 + ak)];
        ra_v[e] = ra ? fmaxf(av, 0.0f) : av;
      }
    }
    if (vec_b && kfull) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int idx = tid + 256 * e;
        const float* src = tb ? g.B + (size_t)(n0 + (idx >> 2)) * g.ldb + (k0 + 4 * (idx & 3))
                              : g.B + (size_t)(k0 + (idx >> 5)) * g.ldb + (n0 + 4 * (idx & 31));
        const f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
        for (int c = 0; c < 4; ++c) rb_v[4 * e + c] = rb ? fmaxf(v[c], 0.0f) : v[c];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = tid + 256 * e;
        int bn, bk;
        if (tb) { bk = idx & 15; bn = idx >> 4; } else { bn = idx & 127; bk = idx >> 7; }
        float bv = 0.0f;
        if (n0 + bn < g.N && k0 + bk < kend)
          bv = tb ? g.B[(size_t)(n0 + bn) * g.ldb + (k0 + bk)] : g.B[(size_t)(k0 + bk) * g.ldb + (n0 + bn)];
        rb_v[e] = rb ? fmaxf(bv, 0.0f) : bv;
      }
    }
  };
  auto stash = [&](int buf, int k0) {
    const bool kfull = k0 + BK <= kend;
    if (vec_a && kfull) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int idx = tid + 256 * e;
        if (ta) {
          *reinterpret_cast<f32x4*>(&As[buf][idx >> 5][4 * (idx & 31)]) = (f32x4){ra_v[4 * e], ra_v[4 * e + 1], ra_v[4 * e + 2], ra_v[4 * e + 3]};
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) As[buf][4 * (idx & 3) + c][idx >> 2] = ra_v[4 * e + c];
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = tid + 256 * e;
        int am, ak;
        if (ta) { am = idx & 127; ak = idx >> 7; } else { ak = idx & 15; am = idx >> 4; }
        As[buf][ak][am] = ra_v[e];
      }
    }
    if (vec_b && kfull) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int idx = tid + 256 * e;
        if (tb) {
#pragma unroll
          for (int c = 0; c < 4; ++c) Bs[buf][4 * (idx & 3) + c][idx >> 2] = rb_v[4 * e + c];
        } else {
          *reinterpret_cast<f32x4*>(&Bs[buf][idx >> 5][4 * (idx & 31)]) = (f32x4){rb_v[4 * e], rb_v[4 * e + 1], rb_v[4 * e + 2], rb_v[4 * e + 3]};
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = tid + 256 * e;
        int bn, bk;
        if (tb) { bk = idx & 15; bn = idx >> 4; } else { bn = idx & 127; bk = idx >> 7; }
        Bs[buf][bk][bn] = rb_v[e];
      }
    }
  };
  fetch(kbeg);
  stash(0, kbeg);
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) fetch(k0 + BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[buf][kk + (lane >> 4)][64 * wm + 16 * i + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[buf][kk + (lane >> 4)][64 * wn + 16 * j + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) stash(buf ^ 1, k0 + BK);
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + 64 * wn + 16 * j + (lane & 15);
      if (n >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long m = m0 + 64 * wm + 16 * i + 4 * (lane >> 4) + r;
        if (m >= g.M) continue;
        float v = acc[i][j][r];
        float* c = g.C + (size_t)m * g.ldc + n;
        if (g.bias && blockIdx.z == 0) v += g.bias[n];
        if (g.mask && !(g.mask[(size_t)m * g.ldc + n] > 0.0f)) v = 0.0f;
        if (g.flags & kAtomic) atomicAdd(c, v);
        else if (g.flags & kAccum) *c += v;
        else *c = v;
      }
    }
}

// ---- the layer-sized products on the bf16 matrix pipe: "bf16x6" --------------------------------------------------------------
// fp32 MFMA peaks at 157 TFLOP/s; the bf16 MFMA at 2.4 PFLOP/s.  Each fp32 operand is split into three bf16 terms
// (a = a0 + a1 + a2: 8 + 8 + 8 mantissa bits, bf16 has fp32's exponent range, so -- unlike an fp16 split -- no scaling and no
// range restriction: loss gradients of 1e-8 are as safe as activations of 1e4) and the six products with at least 2^-16 weight
//   a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)
// are accumulated in fp32 on v_mfma_f32_32x32x16_bf16; the dropped terms are below 2^-24 of |a||b|.  Six MFMAs at 16x the fp32 MFMA
// rate = 2.6x its peak, with products as accurate as fp32's own rounding.
// Workgroup tile 128 x 128 x 32, four waves 2 x 2 of 64 x 64 (2 x 2 MFMA tiles, 64 accumulator registers); the operands are split
// ONCE per tile while they are staged into LDS (fragment order [plane 3][k16 block 2][lane half 2][row 128][8 bf16]: one conflict-
// free ds_read_b128 per fragment); the next k-tile's global loads are issued before the current tile's MFMAs.  Two workgroups per
// CU (48 KB LDS each) let one convert while the other multiplies.
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int XM = 128, XN = 128, XK = 32;
constexpr int kPlaneElems = 2 * 2 * 128 * 8;          // bf16 elements of one plane of one operand tile (8 KB)

__device__ __forceinline__ void split3(const float (&v)[8], bf8& p0, bf8& p1, bf8& p2) {
#ifdef DINER_BF16X6_NOSPLIT      // ablation (wrong results): what the on-the-fly split costs
#pragma unroll
  for (int j = 0; j < 8; ++j) p0[j] = p1[j] = p2[j] = (__bf16)v[j];
  return;
#endif
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

// One operand tile (128 rows of the output dimension x 32 of the contraction) from global memory into registers.
//   kc: the operand is stored with the CONTRACTION index contiguous (row stride ld): thread t takes row t/2, 16 consecutive k
//   mc: stored with the OUTPUT index contiguous: thread t takes rows 2 (t % 64), +1 and the 8 contraction indices of chunk t/64
struct TileRegs {
  float v[2][8];        // two chunks of 8 consecutive contraction indices (kc: chunks 2 (t&1), +1 of row t/2; mc: chunk t/64 of two rows)
};
__device__ __forceinline__ void tile_fetch(TileRegs& r, const float* __restrict__ P, int ld, bool kc, long long mn0, long long MN,
                                           int k0, int kend, bool relu, int tid) {
  if (kc) {
    const long long row = mn0 + (tid >> 1);
    const int kb = k0 + 16 * (tid & 1);
    const bool fast = row < MN && kb + 16 <= kend && (ld & 3) == 0 && ((reinterpret_cast<size_t>(P) & 15) == 0);
    if (fast) {
      const f32x4* src = reinterpret_cast<const f32x4*>(P + (size_t)row * ld + kb);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x4 x = src[e];
#pragma unroll
        for (int c = 0; c < 4; ++c) r.v[e >> 1][4 * (e & 1) + c] = x[c];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e)
        r.v[e >> 3][e & 7] = (row < MN && kb + e < kend) ? P[(size_t)row * ld + kb + e] : 0.0f;
    }
  } else {
    const long long row = mn0 + 2 * (tid & 63);
    const int kb = k0 + 8 * (tid >> 6);
    const bool fast = row + 1 < MN && kb + 8 <= kend && (ld & 1) == 0 && ((reinterpret_cast<size_t>(P) & 7) == 0);
    if (fast) {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const f32x2 x = *reinterpret_cast<const f32x2*>(P + (size_t)(kb + e) * ld + row);
        r.v[0][e] = x[0];
        r.v[1][e] = x[1];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          r.v[q][e] = (row + q < MN && kb + e < kend) ? P[(size_t)(kb + e) * ld + row + q] : 0.0f;
    }
  }
  if (relu) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) r.v[q][e] = fmaxf(r.v[q][e], 0.0f);
  }
}
// ... split and written to LDS: slot (k16 block, lane half h, row) holds contraction indices 16 blk + 8 h + 0..7 of that row
__device__ __forceinline__ void tile_stash(const TileRegs& r, __bf16* lds, bool kc, int tid) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int row, chunk;                   // chunk = 2 blk + h
    if (kc) { row = tid >> 1; chunk = 2 * (tid & 1) + q; }
    else { row = 2 * (tid & 63) + q; chunk = tid >> 6; }
    bf8 p0, p1, p2;
    split3(r.v[q], p0, p1, p2);
    bf8* dst = reinterpret_cast<bf8*>(lds) + chunk * 128 + row;
    dst[0] = p0;
    dst[kPlaneElems / 8] = p1;
    dst[2 * (kPlaneElems / 8)] = p2;
  }
}

// EPI: 0 plain, 1 + resid in the epilogue, 2 + row sums of op(A) (separate instantiations: the plain kernel must stay at 3 waves per SIMD)
template <int EPI>
__global__ __launch_bounds__(256, 2) void k_gemm_bf16x6(GemmArgs g) {
  if (g.gate && *g.gate == 0) return;
  __shared__ __attribute__((aligned(16))) __bf16 As[3 * kPlaneElems], Bs[3 * kPlaneElems];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order.  Workgroups go to the 8 XCDs round-robin by their linear id, so the (up to four) N-tiles that share
  // a 128-row block of A -- consecutive ids -- would land on different XCDs and each pull that block through its own L2.  Remap:
  // the workgroups of XCD x take the x-th eighth of the tile list, in order, so tiles that share operands run on one XCD at
  // about the same time (split-K: the 16 tiles of a K chunk share both operand slices).
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    if (!(g.flags & kNoXcdOrder) && (total & 7) == 0) {
      const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      const unsigned logical = (lin & 7) * (total >> 3) + (lin >> 3);
      bx = logical % gridDim.x;
      by = (logical / gridDim.x) % gridDim.y;
      bz = logical / (gridDim.x * gridDim.y);
    }
  }
  const long long m0 = (long long)by * XM;
  const int n0 = bx * XN;
  const int Keff = g.k_dev ? min(g.K, *g.k_dev) : g.K;
  if (bz * g.k_chunk >= Keff) return;                         // (k_dev: the chunks past the list's end)
  const bool ta = g.flags & kTA, tb = g.flags & kTB, ra = g.flags & kReluA, rb = g.flags & kReluB;
  // op(A) is M x K: stored [M][lda] (contraction contiguous) unless kTA; op(B) is K x N: stored [K][ldb] (output contiguous) unless kTB
  const bool a_kc = !ta, b_kc = tb;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
  TileRegs ra_t, rb_t;
  const bool do_rowsum = EPI == 2 && g.rowsum && bx == 0;          // (one N-tile column of workgroups sums the rows of op(A))
  float rs0 = 0.0f, rs1 = 0.0f;
  // round 6: the workgroup takes chunks bz, bz + gridDim.z, ... into ONE accumulator (a launch covers K in gridDim.z chunks and this loop runs
  // once, except a device-side contraction length: there the host sizes the grid for a few dozen chunk walkers instead of one workgroup
  // per chunk of the list's capacity -- 6816 workgroups of which 368 had work, at ~10 ns of dispatch each)
  for (int kbeg = bz * g.k_chunk; kbeg < Keff; kbeg += (int)gridDim.z * g.k_chunk) {
  const int kend = min(Keff, kbeg + g.k_chunk);
  tile_fetch(ra_t, g.A, g.lda, a_kc, m0, g.M, kbeg, kend, ra, tid);
  tile_fetch(rb_t, g.B, g.ldb, b_kc, n0, g.N, kbeg, kend, rb, tid);
  for (int k0 = kbeg; k0 < kend; k0 += XK) {
    __syncthreads();                                   // the previous tile's fragment reads are done
    if (EPI == 2 && do_rowsum) {                       // op(A) stored [K][lda]: this thread holds rows 2 (tid % 64), +1, 8 contraction indices
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        rs0 += ra_t.v[0][e];
        rs1 += ra_t.v[1][e];
      }
    }
    tile_stash(ra_t, As, a_kc, tid);
    tile_stash(rb_t, Bs, b_kc, tid);
    __syncthreads();
    if (k0 + XK < kend) {                              // next tile's loads fly under this tile's MFMAs
      tile_fetch(ra_t, g.A, g.lda, a_kc, m0, g.M, k0 + XK, kend, ra, tid);
      tile_fetch(rb_t, g.B, g.ldb, b_kc, n0, g.N, k0 + XK, kend, rb, tid);
    }
    const bf8* Af = reinterpret_cast<const bf8*>(As);
    const bf8* Bf = reinterpret_cast<const bf8*>(Bs);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      bf8 a[2][3], b[2][3];
      const int slot = (2 * blk + (lane >> 5)) * 128 + (lane & 31);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          a[i][pl] = Af[pl * (kPlaneElems / 8) + slot + 64 * wm + 32 * i];
          b[i][pl] = Bf[pl * (kPlaneElems / 8) + slot + 64 * wn + 32 * i];
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // smallest terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
        }
    }
  }
  }
  if (EPI == 2 && do_rowsum) {
    const long long row = m0 + 2 * (tid & 63);
    if (row < g.M) atomicAdd(g.rowsum + row, rs0);
    if (row + 1 < g.M) atomicAdd(g.rowsum + row + 1, rs1);
  }
  // D layout of the 32x32 tile: lane holds column lane & 31, rows 8 (e >> 2) + 4 (lane >> 5) + (e & 3)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + 64 * wn + 32 * j + (lane & 31);
      if (n >= g.N) continue;
      const float bias = (g.bias && bz == 0) ? g.bias[n] : 0.0f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const long long m = m0 + 64 * wm + 32 * i + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
        if (m >= g.M) continue;
        float v = acc[i][j][e] + bias;
        if (EPI == 1) v += g.resid[(size_t)m * g.ldc + n];
        float* c = g.C + (size_t)m * g.ldc + n;
        if (g.mask && !(g.mask[(size_t)m * g.ldc + n] > 0.0f)) v = 0.0f;
        if (g.flags & kAtomic) atomicAdd(c, v);
        else if (g.flags & kAccum) *c += v;
        else *c = v;
      }
    }
}

// ---- per-(view, point) inputs ---------------------------------------------------------------------------------
// one 64-lane wave per (view, 16 points): the front end of the inference kernels, written out instead of consumed
__global__ __launch_bounds__(256) void k_train_inputs(SceneDev sc, FieldArgs fa, float* __restrict__ feat,
                                                      int* __restrict__ tap_row, float* __restrict__ tap_w) {
  const int lane = threadIdx.x & 63, q = lane >> 4, pt = lane & 15;
  const long long n_t16 = (fa.P + 15) / 16;
  const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= n_t16 * sc.nv) return;
  const int v = (int)(wid / n_t16);
  const long long p = (wid - (long long)v * n_t16) * 16 + pt;
  if (p >= fa.P) return;
  Taps taps;
  float f[16];
  field_frontend(sc, fa, v, q, p, taps, f);
  const size_t col = (size_t)v * fa.P + p;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) feat[col * kDInPad + 16 * m + 4 * q + r] = f[4 * m + r];
  if (q == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      tap_row[col * 4 + k] = (int)(taps.off[k] / kLatent);
      tap_w[col * 4 + k] = taps.w[k];
    }
  }
}

// lat[col][c] = sum_k w_k latent_cl[row_k][c]      (SpatialEncoder.index, bilinear / border, image_encoder.py:97-146)
__global__ __launch_bounds__(256) void k_gather_latent(const float* __restrict__ latent_cl, const int* __restrict__ tap_row,
                                                       const float* __restrict__ tap_w, long long cols,
                                                       float* __restrict__ lat, const int* __restrict__ gate = nullptr) {
  if (gate && *gate == 0) return;
  const long long col = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col >= cols) return;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(latent_cl + (size_t)tap_row[col * 4 + k] * kLatent + 256 * h + 4 * lane);
      s += t * tap_w[col * 4 + k];
    }
    *reinterpret_cast<f32x4*>(lat + (size_t)col * kLatent + 256 * h + 4 * lane) = s;
  }
}

// adjoint: d_latent_cl[row_k][c] += w_k d_lat[col][c]
__global__ __launch_bounds__(256) void k_scatter_latent(const float* __restrict__ d_lat, const int* __restrict__ tap_row,
                                                        const float* __restrict__ tap_w, long long cols,
                                                        float* __restrict__ d_latent_cl) {
  const long long col = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col >= cols) return;
  const int lane = threadIdx.x & 63;
  for (int c = lane; c < kLatent; c += 64) {
    const float gv = d_lat[(size_t)col * kLatent + c];
#pragma unroll
    for (int k = 0; k < 4; ++k) atomicAdd(d_latent_cl + (size_t)tap_row[col * 4 + k] * kLatent + c, gv * tap_w[col * 4 + k]);
  }
}

// The same adjoint with the taps of 64 consecutive columns merged first: consecutive columns are samples along a ray in one view, their
// bilinear taps fall on a few dozen texels of the epipolar segment, and the 42 M atomics of the reference batch (200 us, 5 % of the step;
// 3.2 ms of the 2048-ray step) become one atomic per distinct texel and channel.  A workgroup (1) finds the distinct texels of its 256 taps
// (first occurrence = leader, compacted by a block prefix sum) and sorts the taps by texel (counting sort in LDS), (2) brings its 64 x 512
// block of d_lat into LDS, (3) per distinct texel sums w * d_lat over that texel's taps -- independent LDS reads, no read-modify-write
// chain (a first version accumulated into a (slot, channel) table in LDS: one dependent LDS round trip per tap) -- and adds the sum to the
// texel with one atomic per channel.
// round 6: COLS columns per workgroup as a template parameter (DINER_TRAIN_SCATTER_COLS: 64 columns stage 128 KB of d_lat in LDS = one
// workgroup per CU, 32 / 16 columns let 2 / 4 share a CU); the map-space lin_z adjoint hands the columns over SORTED by texel (perm), where
// 64 neighbours share a handful of texels.
#ifdef DINER_L512_PROF
__device__ unsigned long long g_scat_prof[8];      // clocks of thread 0 per phase: [0] load issue + taps, [1] wait for the block, [2] leaders .. sort, [3] sums + atomics, [4] workgroups
#define SCAT_T(i) do { if (t == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); atomicAdd(&g_scat_prof[i], now_ - pt_); pt_ = now_; } } while (0)
#else
#define SCAT_T(i) do { } while (0)
#endif
template <int COLS>
__global__ __launch_bounds__(256) void k_scatter_latent_merged(const float* __restrict__ d_lat, const int* __restrict__ tap_row,
                                                               const float* __restrict__ tap_w, long long cols,
                                                               float* __restrict__ d_latent_cl, const int* __restrict__ perm) {
  // round 6, from the kernel's phase timer (55 k clocks per workgroup: 36 k in the load loop -- a dynamic trip count, one 8-byte load per
  // thread and trip, each waited for -- 5 k in a serial leader search + a one-thread prefix sum, 13 k in the sums, one texel after the
  // other with 256 threads x 2 channels): (2) all of the block's rows requested at once, 16 bytes per lane; (1) the leader = the smallest
  // matching index from a vectorised compare against all taps, the prefix sum by one wave; (3) a WAVE per texel (64 lanes x 8 channels,
  // four taps per trip): four texels in flight per workgroup.
  constexpr int NT = 4 * COLS;                                                    // taps of the workgroup (threads 0 .. NT - 1 own one each)
  constexpr int NL = COLS / 2;                                                    // row requests per thread: a trip brings two rows (128 lanes x 16 B each)
  extern __shared__ __attribute__((aligned(16))) char smem_scat[];
  float* dl = reinterpret_cast<float*>(smem_scat);                               // [COLS][512]
  __shared__ __attribute__((aligned(16))) int s_id[NT];
  __shared__ int s_uniq[NT], s_start[NT + 1], s_fill[NT], s_wave_n[4];
  __shared__ float s_tw[NT];
  __shared__ short s_lead_slot[NT];
  __shared__ unsigned char s_tcol[NT];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#ifdef DINER_L512_PROF
  const bool no_atomics = g_scat_prof[7] != 0;               // measurement switch, read once in front of everything
  unsigned long long pt_ = __builtin_readcyclecounter();
#endif
  // (2) the block of d_lat, COLS rows of 2 KB (rows past the end: the last row again, never referenced).  round 6: the workgroup walks
  // blocks blockIdx.x, + gridDim.x, ... and requests block i + 1 (rows into registers, its own tap) in front of the phases of block i
  // (a workgroup per block spends 31 k of its 46 k clocks waiting for its rows; measured: the walk is worth 0.5 % of the step at 64 columns
  // x one workgroup per CU, nothing at 32 x 2 -- the memory system's rate on random 2 KB rows is the limit either way)
  const long long nblk = (cols + COLS - 1) / COLS;
  const int half = t >> 7, q = t & 127;
  f32x4 v[NL];
  int id_n = -1;
  float w_n = 0.0f;
  auto request = [&](long long blk) {
    const long long col0 = blk * COLS;
    const int last = (int)(cols - col0 < COLS ? cols - col0 : COLS);
    long long src[NL];
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int g = 2 * it + half;
      const long long pos = col0 + (g < last ? g : last - 1);
      src[it] = perm ? (long long)perm[pos] : pos;       // round 6: the columns in the order of a list sorted by texel (see k_fill_perm)
    }
#pragma unroll
    for (int it = 0; it < NL; ++it) v[it] = *reinterpret_cast<const f32x4*>(d_lat + (size_t)src[it] * kLatent + 4 * q);
    const long long mypos = col0 + (t >> 2);
    id_n = -1;
    w_n = 0.0f;
    if (t < NT && mypos < cols) {
      const long long mycol = perm ? perm[mypos] : mypos;
      id_n = tap_row[mycol * 4 + (t & 3)];
      w_n = tap_w[mycol * 4 + (t & 3)];
      if (w_n == 0.0f) id_n = -1;
    }
  };
  if ((long long)blockIdx.x < nblk) request(blockIdx.x);
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
  const int id = id_n;
  const float w = w_n;
  if (t < NT) {
    s_id[t] = id;
    s_fill[t] = 0;
  }
  SCAT_T(0);
#pragma unroll
  for (int it = 0; it < NL; ++it) *reinterpret_cast<f32x4*>(dl + (2 * it + half) * kLatent + 4 * q) = v[it];
  __syncthreads();
  if (blk + gridDim.x < nblk) request(blk + gridDim.x);
  SCAT_T(1);
  // (1) leaders, slots
  int leader = t;
  if (id >= 0) {
    int best = NT;
#pragma unroll 8
    for (int j4 = 0; j4 < NT / 4; ++j4) {
      const int4 o = reinterpret_cast<const int4*>(s_id)[j4];
      int m = o.w == id ? 4 * j4 + 3 : NT;
      m = o.z == id ? 4 * j4 + 2 : m;
      m = o.y == id ? 4 * j4 + 1 : m;
      m = o.x == id ? 4 * j4 : m;
      best = m < best ? m : best;
    }
    leader = best;                                             // <= t: the tap matches itself
  }
  const bool is_leader = id >= 0 && leader == t;
  const unsigned long long ball = __ballot(is_leader);
  if (lane == 0) s_wave_n[wave] = __popcll(ball);
  __syncthreads();
  int before = __popcll(ball & ((1ull << lane) - 1));
  for (int i = 0; i < wave; ++i) before += s_wave_n[i];
  if (is_leader) {
    s_lead_slot[t] = (short)before;
    s_uniq[before] = id;
  }
  const int n_unique = s_wave_n[0] + s_wave_n[1] + s_wave_n[2] + s_wave_n[3];
  __syncthreads();
  const int slot = id >= 0 ? (int)s_lead_slot[leader] : -1;
  if (slot >= 0) atomicAdd(&s_fill[slot], 1);               // taps per texel
  __syncthreads();
  if (wave == 0) {                                           // exclusive prefix over <= NT counts: PER consecutive counts per lane, a wave scan
    constexpr int PER = (NT + 63) / 64;
    int x[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = lane * PER + k;
      x[k] = i < n_unique ? s_fill[i] : 0;
      sum += x[k];
    }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(incl, o);
      if (lane >= o) incl += y;
    }
    int run = incl - sum;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = lane * PER + k;
      if (i < n_unique) {
        s_start[i] = run;
        s_fill[i] = 0;
      }
      run += x[k];
    }
    if (lane == 63) s_start[n_unique] = run;
  }
  __syncthreads();
  if (slot >= 0) {
    const int at = s_start[slot] + atomicAdd(&s_fill[slot], 1);
    s_tcol[at] = (unsigned char)(t >> 2);
    s_tw[at] = w;
  }
  __syncthreads();
  SCAT_T(2);
  // (3) one sum and one atomic per distinct texel and channel: wave `wave` takes texels wave, wave + 4, ...; a lane owns channels
  // [4 lane, +4) and [256 + 4 lane, +4) (16-byte LDS reads, lanes 16 bytes apart); four taps per trip, a trip past the texel's end repeats
  // its last tap with weight 0
  for (int sl = wave; sl < n_unique; sl += 4) {
    const int b = s_start[sl], e = s_start[sl + 1];
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, b0 = a0, b1 = a0;
    for (int i = b; i < e; i += 4) {
      int c[4];
      float wk[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int at = i + k < e ? i + k : e - 1;
        c[k] = (int)s_tcol[at];
        wk[k] = i + k < e ? s_tw[at] : 0.0f;
      }
      f32x4 d0[4], d1[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        d0[k] = *reinterpret_cast<const f32x4*>(dl + c[k] * kLatent + 4 * lane);
        d1[k] = *reinterpret_cast<const f32x4*>(dl + c[k] * kLatent + 256 + 4 * lane);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a0[j] = fmaf(wk[0], d0[0][j], a0[j]);
        a1[j] = fmaf(wk[0], d1[0][j], a1[j]);
        b0[j] = fmaf(wk[1], d0[1][j], b0[j]);
        b1[j] = fmaf(wk[1], d1[1][j], b1[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a0[j] = fmaf(wk[2], d0[2][j], a0[j]);
        a1[j] = fmaf(wk[2], d1[2][j], a1[j]);
        b0[j] = fmaf(wk[3], d0[3][j], b0[j]);
        b1[j] = fmaf(wk[3], d1[3][j], b1[j]);
      }
    }
    a0 += b0;
    a1 += b1;
    float* dst = d_latent_cl + (size_t)s_uniq[sl] * kLatent + 4 * lane;
#ifdef DINER_L512_PROF
    if (no_atomics) {                                        // measurement: the kernel without its atomics (one plain store per wave and texel keeps the sums alive)
      if (a0[0] + a0[1] + a0[2] + a0[3] + a1[0] + a1[1] + a1[2] + a1[3] == 12345.678f) dst[0] = 1.0f;
      continue;
    }
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      atomicAdd(dst + j, a0[j]);
      atomicAdd(dst + 256 + j, a1[j]);
    }
  }
  SCAT_T(3);
#ifdef DINER_L512_PROF
  if (t == 0) atomicAdd(&g_scat_prof[4], 1ull);
#endif
  __syncthreads();                                           // the block and the lists are free for the next trip
  }
}
template <int COLS>
static int scatter_merged_launch(const float* d_lat, const int* tap_row, const float* tap_w, long long cols, float* d_latent_cl, hipStream_t st,
                                 const int* perm) {
  static std::atomic<int> attr_set[64];
  int dev = 0;
  DINER_HIP_OK(hipGetDevice(&dev));
  dev &= 63;
  constexpr int lds = COLS * kLatent * (int)sizeof(float);      // the COLS x 512 block of d_lat
  if (!attr_set[dev].load()) {
    DINER_HIP_OK(hipFuncSetAttribute((const void*)k_scatter_latent_merged<COLS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set[dev].store(1);
  }
  static const int per_cu = [] { const char* e = getenv("DINER_TRAIN_SCATTER_WG_PER_CU"); return e ? atoi(e) : 64 / COLS; }();      // workgroups per CU walking the blocks (what LDS holds); 0: a workgroup per block (no walk)
  static std::atomic<int> cus[64];
  if (per_cu > 0 && !cus[dev].load()) {
    int c = 0;
    DINER_HIP_OK(hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev));
    cus[dev].store(c > 0 ? c : 256);
  }
  const long long nblk = (cols + COLS - 1) / COLS;
  const long long grid = per_cu > 0 && (long long)per_cu * cus[dev].load() < nblk ? (long long)per_cu * cus[dev].load() : nblk;
  hipLaunchKernelGGL(k_scatter_latent_merged<COLS>, dim3((unsigned)grid), dim3(256), lds, st, d_lat, tap_row, tap_w, cols, d_latent_cl, perm);
  DINER_LAUNCH_OK();
  return 0;
}
// round 6: the columns of one object sorted by the texel row of their first tap (counting sort: histogram, exclusive scan in two kernels,
// fill).  A 64 x 64 patch of rays x 40 samples x 4 views names 2.6 M taps on 11.5 k texel rows -- 227 taps per row -- but 32 CONSECUTIVE
// columns (most of one ray in one view) still fall on ~30 distinct texels, so the merged scatter issued 605 k row-atomics per launch
// (1.24 GB of atomic traffic beside the 1.34 GB it reads).  32 columns that are neighbours in the sorted order share their 2 x 2 footprint.
// The order inside a key is the atomics' (as the float atomics' own order: the sums differ in the last bits between runs, as before).
__global__ __launch_bounds__(256) void k_hist_first_tap(const int* __restrict__ tap_row, long long cols, int* __restrict__ hist) {
  for (long long c = blockIdx.x * 256ll + threadIdx.x; c < cols; c += gridDim.x * 256ll) atomicAdd(hist + tap_row[c * 4], 1);
}
constexpr int kScanBlock = 2048;       // elements per workgroup of the scan (8 per thread)
__global__ __launch_bounds__(256) void k_scan_sums(const int* __restrict__ v, int n, int* __restrict__ sums) {
  __shared__ int s_w[4];
  const int base = blockIdx.x * kScanBlock + threadIdx.x * 8;
  int a = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) a += base + i < n ? v[base + i] : 0;
  for (int o = 32; o; o >>= 1) a += __shfl_xor(a, o);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
// v[i] <- sum of v[0 .. i) : the workgroup's base = the sums of the workgroups before it (a few hundred values), then a scan of its 2048
__global__ __launch_bounds__(256) void k_scan_apply(int* __restrict__ v, int n, const int* __restrict__ sums) {
  __shared__ int s_w[4], s_base[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int b = 0;
  for (int i = t; i < (int)blockIdx.x; i += 256) b += sums[i];
  for (int o = 32; o; o >>= 1) b += __shfl_xor(b, o);
  if (lane == 0) s_base[wave] = b;
  const int base = blockIdx.x * kScanBlock + t * 8;
  int x[8], a = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = base + i < n ? v[base + i] : 0; a += x[i]; }
  int incl = a;
  for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int run = s_base[0] + s_base[1] + s_base[2] + s_base[3] + incl - a;
  for (int i = 0; i < wave; ++i) run += s_w[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) { if (base + i < n) v[base + i] = run; run += x[i]; }
}
__global__ __launch_bounds__(256) void k_fill_perm(const int* __restrict__ tap_row, long long cols, int* __restrict__ offs, int* __restrict__ perm) {
  for (long long c = blockIdx.x * 256ll + threadIdx.x; c < cols; c += gridDim.x * 256ll) perm[atomicAdd(offs + tap_row[c * 4], 1)] = (int)c;
}
// perm (cols ints) from the taps; hist: rows ints (destroyed), sums: (rows + 2047) / 2048 ints
static int sort_columns_by_texel(const int* tap_row, long long cols, long long rows, int* hist, int* sums, int* perm, hipStream_t st) {
  DINER_HIP_OK(hipMemsetAsync(hist, 0, (size_t)rows * sizeof(int), st));
  const unsigned nb = (unsigned)((rows + kScanBlock - 1) / kScanBlock);
  hipLaunchKernelGGL(k_hist_first_tap, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, st, tap_row, cols, hist);
  hipLaunchKernelGGL(k_scan_sums, dim3(nb), dim3(256), 0, st, (const int*)hist, (int)rows, sums);
  hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(256), 0, st, hist, (int)rows, (const int*)sums);
  hipLaunchKernelGGL(k_fill_perm, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, st, tap_row, cols, hist, perm);
  DINER_LAUNCH_OK();
  return 0;
}
int scatter_latent_launch(const float* d_lat, const int* tap_row, const float* tap_w, long long cols, float* d_latent_cl, hipStream_t st,
                          const int* perm = nullptr) {
  static const bool merged = [] { const char* e = getenv("DINER_TRAIN_SCATTER_MERGED"); return !(e && *e == '0'); }();
  // same-box A/B, SB 4 step, round 6 (profiles/r06_train_scatter_*): 64 / 32 / 16 columns with the one-block workgroups of the first version 123.6 /
  // 121.9 / 122.2 ms; with sorted columns and the rewritten phases 102.2 / 102.7 / 108.5; 64 columns + one walking workgroup per CU 101.8
  static const int ncols = [] { const char* e = getenv("DINER_TRAIN_SCATTER_COLS"); return e ? atoi(e) : 64; }();
  if (merged && (reinterpret_cast<size_t>(d_lat) & 15) == 0) {
    if (ncols == 16) return scatter_merged_launch<16>(d_lat, tap_row, tap_w, cols, d_latent_cl, st, perm);
    if (ncols == 32) return scatter_merged_launch<32>(d_lat, tap_row, tap_w, cols, d_latent_cl, st, perm);
    return scatter_merged_launch<64>(d_lat, tap_row, tap_w, cols, d_latent_cl, st, perm);
  }
  hipLaunchKernelGGL(k_scatter_latent, dim3((unsigned)((cols + 3) / 4)), dim3(256), 0, st, d_lat, tap_row, tap_w, cols, d_latent_cl);
  DINER_LAUNCH_OK();
  return 0;
}

// channels-last (n, HW, C) -> (n, C, HW) through a 64 x 64 LDS tile: the latent gradient in the layout the image encoder's autograd
// takes it in (a strided torch copy of the permuted view ran at 0.2 TB/s: 79 us for the 8 MB of the 64 x 64 test maps)
__global__ __launch_bounds__(256) void k_cl_to_nchw(const float* __restrict__ src, long long HW, int C, float* __restrict__ dst) {
  __shared__ float tile[64][65];
  const long long p0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const size_t img = (size_t)blockIdx.z * HW * C;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4)
    if (p0 + r < HW && c0 + tx < C) tile[r][tx] = src[img + (size_t)(p0 + r) * C + c0 + tx];
  __syncthreads();
  for (int r = ty; r < 64; r += 4)
    if (c0 + r < C && p0 + tx < HW) dst[img + (size_t)(c0 + r) * HW + p0 + tx] = tile[tx][r];
}

// ---- round 6: the latent rows a training batch touches ----------------------------------------------------------------------------------
// The fused forward gathers its lin_z terms from the latent map projected through lin_z[0..2]; projecting the WHOLE map per object and step
// costs a map's worth of rows whatever the batch (226 k rows at 400 x 300: 1.1 ms per object; 1024 x 1024 maps fell back to the layer-wise
// forward), while a 64 x 64 patch of rays with 40 depth-guided samples touches 2.4 % of the texels (5.5 k of 226 k).  So: mark the texel rows
// the batch's taps name, compact them into a list, project THAT list (gather rows -> product over a device-side row count -> scatter rows).
// Texels a tap could name under another rounding of the last bit carry a weight of ~1e-7: they read whatever finite value the buffer holds
// (the host zero-fills it once), never an unwritten NaN.
__global__ __launch_bounds__(256) void k_mark_rows(const int* __restrict__ tap_row, long long n, int* __restrict__ mark) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) mark[tap_row[i]] = 1;
}
// idx[0 .. count) = the marked rows (order: whatever the atomics give; the products are row-wise); *dense = 1 when more than cap rows are marked
// (the list is then unusable: the caller's dense projection, gated on the flag, takes over)
__global__ __launch_bounds__(256) void k_compact_rows(const int* __restrict__ mark, int nrows, int cap, int* __restrict__ idx, int* __restrict__ count,
                                                      int* __restrict__ dense) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nrows; i += gridDim.x * 256) {
    if (mark[i]) {
      const int at = atomicAdd(count, 1);
      if (at < cap) idx[at] = i;
      else *dense = 1;
    }
  }
}
// dst[r] = src[idx[r]] (gather) or dst[idx[r]] = src[r] (scatter) for r < min(*count, cap), rows of 512 floats; blockIdx.y = plane (strides in floats)
__global__ __launch_bounds__(256) void k_move_rows(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ idx,
                                                   const int* __restrict__ count, int cap, int scatter, size_t src_plane, size_t dst_plane,
                                                   const int* __restrict__ skip) {
  if (skip && *skip != 0) return;
  int n = *count;
  n = n < cap ? n : cap;
  const float* s = src + (size_t)blockIdx.y * src_plane;
  float* d = dst + (size_t)blockIdx.y * dst_plane;
  const int lane = threadIdx.x & 63;
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += gridDim.x * 4) {
    const size_t a = (size_t)(scatter ? r : idx[r]) * kLatent, b = (size_t)(scatter ? idx[r] : r) * kLatent;
    f32x4 v0 = reinterpret_cast<const f32x4*>(s + a)[lane], v1 = reinterpret_cast<const f32x4*>(s + a)[64 + lane];
    if (scatter == 2) {                                      // dst[idx[r]] += src[r] (the list's rows are distinct: no atomics)
      v0 += reinterpret_cast<const f32x4*>(d + b)[lane];
      v1 += reinterpret_cast<const f32x4*>(d + b)[64 + lane];
    }
    reinterpret_cast<f32x4*>(d + b)[lane] = v0;
    reinterpret_cast<f32x4*>(d + b)[64 + lane] = v1;
  }
}
// dst[idx[r]] = 0 for r < *count (rows of 512 floats)
__global__ __launch_bounds__(256) void k_zero_rows(float* __restrict__ dst, const int* __restrict__ idx, const int* __restrict__ count) {
  const int n = *count, lane = threadIdx.x & 63;
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += gridDim.x * 4) {
    f32x4* d = reinterpret_cast<f32x4*>(dst + (size_t)idx[r] * kLatent);
    d[lane] = (f32x4){0.f, 0.f, 0.f, 0.f};
    d[64 + lane] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
}
// seg[s] = rows of list segment s: clamp(*count - s cap, 0, cap)
__global__ void k_segment_counts(const int* __restrict__ count, int cap, int n_seg, int* __restrict__ seg) {
  const int s = threadIdx.x;
  if (s < n_seg) {
    const int left = *count - s * cap;
    seg[s] = left < 0 ? 0 : (left > cap ? cap : left);
  }
}

// relu decisions of (rows, 512) pre-activations as bits in the layout Lin512Args.maskbits names: one thread per dword = 32 features
// 128 s + 16 mo + 4 q + i (s = dword / 4, q = dword % 4; bit 4 mo + i) -- eight 16-byte reads; blockIdx.y = tensor
struct MakeBits { const float* src[10]; unsigned* dst[10]; long long rows[10]; };
__global__ void k_make_bits(MakeBits m, const int* __restrict__ gate) {
  if (gate && *gate == 0) return;
  const float* __restrict__ src = m.src[blockIdx.y];
  unsigned* __restrict__ dst = m.dst[blockIdx.y];
  const long long n = m.rows[blockIdx.y] * 16;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i >> 4;
    const int d = (int)(i & 15), s = d >> 2, q = d & 3;
    const float* x = src + (size_t)row * kHidden + 128 * s + 4 * q;
    unsigned bits = 0;
#pragma unroll
    for (int mo = 0; mo < 8; ++mo) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + 16 * mo);
#pragma unroll
      for (int c = 0; c < 4; ++c) bits |= (v[c] > 0.0f ? 1u : 0u) << (4 * mo + c);
    }
    dst[i] = bits;
  }
}

// y[p][c] = mean_v x[v][p][c]   (combine_interleaved, resnetfc.py:150-152); adjoint: dx[v][p][c] = dy[p][c] / nv
// y = mean_v relu(x[v]) (round 6: the activation operand of block 2's fc_1 weight gradient behind the view mean, see view_shared); PC4 = P x 128
__global__ __launch_bounds__(256) void k_view_mean_relu(const float* __restrict__ x, int nv, long long PC4, float* __restrict__ y) {
  const float inv = 1.0f / (float)nv;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < PC4; i += gridDim.x * 256ll) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int v = 0; v < nv; ++v) {
      const f32x4 t = reinterpret_cast<const f32x4*>(x)[(size_t)v * PC4 + i];
#pragma unroll
      for (int c = 0; c < 4; ++c) s[c] += fmaxf(t[c], 0.0f);
    }
    reinterpret_cast<f32x4*>(y)[i] = s * inv;
  }
}
__global__ void k_view_mean(const float* __restrict__ x, int nv, long long PC, float* __restrict__ y, const int* __restrict__ gate = nullptr) {
  if (gate && *gate == 0) return;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < PC; i += (long long)gridDim.x * blockDim.x) {
    float s = 0.0f;
    for (int v = 0; v < nv; ++v) s += x[(size_t)v * PC + i];
    y[i] = s / (float)nv;
  }
}
// atomic maximum of the bit patterns of |x[0 .. n)| into *out: the scale slot of a dy tensor whose producer was not one of the kernels that
// keep the maximum themselves (the general GEMM behind a product with fewer than 256 rows or odd alignment; ADVICE r4: those slots used
// to stay 0 and the f16x3 consumers staged small gradients unscaled)
__global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
  unsigned m = 0;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
// amax_in / amax_out (or null): the maximum of |dx| is that of |dy| / nv (thread 0 writes it)
__global__ void k_view_bcast(const float* __restrict__ dy, int nv, long long PC, float* __restrict__ dx, const unsigned* __restrict__ amax_in,
                             unsigned* __restrict__ amax_out) {
  const float inv = 1.0f / (float)nv;
  if (amax_out && blockIdx.x == 0 && threadIdx.x == 0) *amax_out = __float_as_uint(fabsf(__uint_as_float(*amax_in) * inv));
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < PC; i += (long long)gridDim.x * blockDim.x) {
    const float gv = dy[i] * inv;
    for (int v = 0; v < nv; ++v) dx[(size_t)v * PC + i] = gv;
  }
}

// round 6: behind the view mean every view's rows of dy are the same row (the broadcast), so block 2's fc_1 data gradient (dy W) is ONE
// product per point: T sits in the rows of view 0 (rows [0, P) of the object, PC = P x 512 floats), this kernel writes view v's rows
// T x [relu decision of H[v]] for v = nv - 1 .. 0 (view 0 in place: the same thread read its four floats before).  bits: the layout of
// k_make_bits (16 dwords per row); the maximum of what is stored goes to amax_out (atomic, slot zeroed by the step).
__global__ __launch_bounds__(256) void k_mask_views(float* __restrict__ dH, const unsigned* __restrict__ bits, int nv, long long P,
                                                    unsigned* __restrict__ amax_out) {
  float m = 0.0f;
  const long long n4 = P * (kHidden / 4);
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += gridDim.x * 256ll) {
    const long long row = i >> 7;
    const int j = (int)(i & 127);                             // features 4 j .. 4 j + 3: dword 4 (j / 32) + j % 4, bits 4 ((j / 4) % 8) + c
    const f32x4 t = *reinterpret_cast<const f32x4*>(dH + (size_t)row * kHidden + 4 * j);
    const int dw = 4 * (j >> 5) + (j & 3), sh = 4 * ((j >> 2) & 7);
    unsigned nb[4];                                           // (the first four views' decisions requested together, in front of the stores)
#pragma unroll
    for (int v = 0; v < 4; ++v) nb[v] = v < nv ? bits[((size_t)v * P + row) * 16 + dw] : 0u;
    auto put = [&](int v, unsigned word) {
      const unsigned nib = word >> sh;
      f32x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        o[c] = (nib >> c) & 1u ? t[c] : 0.0f;
        m = fmaxf(m, fabsf(o[c]));
      }
      *reinterpret_cast<f32x4*>(dH + ((size_t)v * P + row) * kHidden + 4 * j) = o;
    };
    for (int v = nv - 1; v >= 4; --v) put(v, bits[((size_t)v * P + row) * 16 + dw]);
#pragma unroll
    for (int v = 3; v >= 0; --v)
      if (v < nv) put(v, nb[v]);
  }
  if (amax_out) {
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(amax_out, __float_as_uint(m));
  }
}

// db[n] += sum_m dY[m][n]
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ dY, long long M, int N, int ld, float* __restrict__ db) {
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6;
  if (n >= N) return;
  float s = 0.0f;
  for (long long m = (long long)blockIdx.y * 4 + part; m < M; m += (long long)gridDim.y * 4) s += dY[(size_t)m * ld + n];
  atomicAdd(db + n, s);
}

// out = [sigmoid(raw rgb), relu(raw sigma)]  (pixelnerf.py:139-143) and its adjoint
__global__ void k_field_act(const float* __restrict__ raw, long long P, int ld, float* __restrict__ out, const int* __restrict__ gate = nullptr) {
  if (gate && *gate == 0) return;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
    const float* r = raw + (size_t)p * ld;
    f32x4 o;
    o[0] = 1.0f / (1.0f + expf(-r[0]));
    o[1] = 1.0f / (1.0f + expf(-r[1]));
    o[2] = 1.0f / (1.0f + expf(-r[2]));
    o[3] = fmaxf(r[3], 0.0f);
    reinterpret_cast<f32x4*>(out)[p] = o;
  }
}
__global__ void k_field_act_bwd(const float* __restrict__ raw, const float* __restrict__ dout, long long P, int ld,
                                float* __restrict__ draw) {
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
    const float* r = raw + (size_t)p * ld;
    const f32x4 g4 = reinterpret_cast<const f32x4*>(dout)[p];
    float* d = draw + (size_t)p * ld;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float s = 1.0f / (1.0f + expf(-r[c]));
      d[c] = g4[c] * s * (1.0f - s);
    }
    d[3] = r[3] > 0.0f ? g4[3] : 0.0f;
    for (int c = 4; c < ld; ++c) d[c] = 0.0f;
  }
}

// ---- lin_out (512 -> 4) of the training step as skinny fp32 kernels: the general product spends 47 us on the forward and 62 + 22 us on
// the two adjoints of a 5120-row batch whose 10 MB are a few microseconds of memory traffic.  One wave per row at a time, a lane holds
// columns 4 lane .. + 3 and 256 + 4 lane .. + 3 of the row and of the four weight rows; explicit fmaf chains, wave sums by shuffles.
// forward: raw = relu(x) W^T + b and out = [sigmoid(raw rgb), relu(raw sigma)] (resnetfc.py:157-158 + pixelnerf.py:139-143)
__global__ __launch_bounds__(256) void k_lin_out_fwd(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b,
                                                     long long P, float* __restrict__ raw, float* __restrict__ out,
                                                     const int* __restrict__ gate = nullptr) {
  if (gate && *gate == 0) return;
  const int lane = threadIdx.x & 63;
  f32x4 w[4][2];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int h = 0; h < 2; ++h) w[o][h] = *reinterpret_cast<const f32x4*>(W + o * kHidden + 256 * h + 4 * lane);
  const f32x4 bias = *reinterpret_cast<const f32x4*>(b);
  for (long long p = blockIdx.x * 4ll + (threadIdx.x >> 6); p < P; p += gridDim.x * 4ll) {
    f32x4 xv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      xv[h] = *reinterpret_cast<const f32x4*>(x + (size_t)p * kHidden + 256 * h + 4 * lane);
#pragma unroll
      for (int c = 0; c < 4; ++c) xv[h][c] = fmaxf(xv[h][c], 0.0f);
    }
    float s[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float t = 0.0f;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < 4; ++c) t = fmaf(xv[h][c], w[o][h][c], t);
      s[o] = t;
    }
#pragma unroll
    for (int off = 32; off; off >>= 1)
#pragma unroll
      for (int o = 0; o < 4; ++o) s[o] += __shfl_xor(s[o], off);
    if (lane == 0) {
      f32x4 r, a;
#pragma unroll
      for (int o = 0; o < 4; ++o) r[o] = s[o] + bias[o];
      a[0] = 1.0f / (1.0f + expf(-r[0]));
      a[1] = 1.0f / (1.0f + expf(-r[1]));
      a[2] = 1.0f / (1.0f + expf(-r[2]));
      a[3] = fmaxf(r[3], 0.0f);
      reinterpret_cast<f32x4*>(raw)[p] = r;
      reinterpret_cast<f32x4*>(out)[p] = a;
    }
  }
}
// backward: d_raw from (raw, d_out) as k_field_act_bwd computes it; dx = (x > 0) * (d_raw W); dW += d_raw^T relu(x), db += column sums of
// d_raw (atomics: both zeroed by the caller), one pass over x
// amax_out (or null): atomic maximum of the bit patterns of |dx| (the scale of the f16x3 products that consume dx)
__global__ __launch_bounds__(256) void k_lin_out_bwd(const float* __restrict__ x, const float* __restrict__ raw, const float* __restrict__ dout,
                                                     const float* __restrict__ W, long long P, float* __restrict__ dx,
                                                     float* __restrict__ dW, float* __restrict__ db, unsigned* __restrict__ amax_out) {
  __shared__ float red[4][4][kHidden + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned dmax = 0;
  f32x4 w[4][2], gw[4][2];
  float gb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      w[o][h] = *reinterpret_cast<const f32x4*>(W + o * kHidden + 256 * h + 4 * lane);
      gw[o][h] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  for (long long p = blockIdx.x * 4ll + wave; p < P; p += gridDim.x * 4ll) {
    const f32x4 r = reinterpret_cast<const f32x4*>(raw)[p];
    const f32x4 g4 = reinterpret_cast<const f32x4*>(dout)[p];
    float d[4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float sg = 1.0f / (1.0f + expf(-r[c]));
      d[c] = g4[c] * sg * (1.0f - sg);
    }
    d[3] = r[3] > 0.0f ? g4[3] : 0.0f;
#pragma unroll
    for (int o = 0; o < 4; ++o) gb[o] += d[o];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const size_t at = (size_t)p * kHidden + 256 * h + 4 * lane;
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + at);
      f32x4 v;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float t = d[0] * w[0][h][c];
#pragma unroll
        for (int o = 1; o < 4; ++o) t = fmaf(d[o], w[o][h][c], t);
        v[c] = xv[c] > 0.0f ? t : 0.0f;
        dmax = max(dmax, __float_as_uint(v[c]) & 0x7fffffffu);
        const float xr = fmaxf(xv[c], 0.0f);
#pragma unroll
        for (int o = 0; o < 4; ++o) gw[o][h][c] = fmaf(d[o], xr, gw[o][h][c]);
      }
      *reinterpret_cast<f32x4*>(dx + at) = v;
    }
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c = 0; c < 4; ++c) red[wave][o][256 * h + 4 * lane + c] = gw[o][h][c];
    if (lane == 0) red[wave][o][kHidden] = gb[o];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * (kHidden + 1); i += 256) {
    const int o = i / (kHidden + 1), k = i % (kHidden + 1);
    const float t = (red[0][o][k] + red[1][o][k]) + (red[2][o][k] + red[3][o][k]);
    if (k < kHidden) atomicAdd(dW + o * kHidden + k, t);
    else atomicAdd(db + o, t);
  }
  if (amax_out) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dmax = max(dmax, (unsigned)__shfl_xor((int)dmax, o, 64));
    if (lane == 0 && dmax) atomicMax(amax_out, dmax);
  }
}

// Adjoint of the compositing arithmetic (nerf_renderer.py:299-301, :341-360) with respect to the field values:
//   delta_k = z_{k+1} - z_k (last: far - z_K); s_k = relu(sigma_k); a_k = 1 - exp(-delta_k s_k); t_k = 1 - a_k + 1e-10;
//   T_k = prod_{j<k} t_j; w_k = a_k T_k; rgb = sum w c (+ 1 - sum w); depth = sum w z.
// One thread per ray (K <= 256): G_k = g_rgb.c_k + g_depth z_k - [white] sum(g_rgb); dL/da_k = G_k T_k - S_k / t_k with
// S_k = sum_{m>k} G_m w_m; dL/dsigma_k = dL/da_k * delta_k (1 - a_k) * [sigma_k > 0]; dL/dc_k = w_k g_rgb.
constexpr int kCompBwdMaxK = 256;
__global__ void k_composite_bwd(const float* __restrict__ field, const float* __restrict__ z, const float* __restrict__ rays,
                                int NR, int K, int white, const float* __restrict__ g_rgb, const float* __restrict__ g_depth,
                                float* __restrict__ d_field) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= NR) return;
  const float far = rays[(size_t)r * 8 + 7];
  const float gr = g_rgb[3 * r], gg = g_rgb[3 * r + 1], gb = g_rgb[3 * r + 2];
  const float gd = g_depth ? g_depth[r] : 0.0f;
  const float gw = white ? (gr + gg + gb) : 0.0f;
  const float* f = field + (size_t)r * K * 4;
  const float* zr = z + (size_t)r * K;
  float* df = d_field + (size_t)r * K * 4;
  float T = 1.0f;
  for (int k = 0; k < K; ++k) {                    // forward sweep: w_k -> dL/dc_k, stash T_k in the sigma slot
    const float delta = (k + 1 < K ? zr[k + 1] : far) - zr[k];
    const float a = 1.0f - expf(-delta * fmaxf(f[4 * k + 3], 0.0f));
    const float w = a * T;
    df[4 * k + 0] = w * gr;
    df[4 * k + 1] = w * gg;
    df[4 * k + 2] = w * gb;
    df[4 * k + 3] = T;
    T *= 1.0f - a + 1e-10f;
  }
  float S = 0.0f;
  for (int k = K - 1; k >= 0; --k) {               // backward sweep with the suffix sum S_k
    const float delta = (k + 1 < K ? zr[k + 1] : far) - zr[k];
    const float sg = f[4 * k + 3];
    const float a = 1.0f - expf(-delta * fmaxf(sg, 0.0f));
    const float t = 1.0f - a + 1e-10f;
    const float Tk = df[4 * k + 3];
    const float G = gr * f[4 * k] + gg * f[4 * k + 1] + gb * f[4 * k + 2] + gd * zr[k] - gw;
    const float da = G * Tk - S / t;
    df[4 * k + 3] = sg > 0.0f ? da * delta * (1.0f - a) : 0.0f;
    S += G * a * Tk;
  }
}

static int grid1d(long long n, int block = 256, int cap = 8192) {
  const long long b = (n + block - 1) / block;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace train
}  // namespace diner

using namespace diner;
using namespace diner::train;

static int gemm_launch(const float* A, const float* B, float* C, long long M, int N, int K, int lda, int ldb, int ldc,
                       int flags, const float* bias, const float* mask, int k_split, hipStream_t stream,
                       const float* resid = nullptr, float* rowsum = nullptr, const int* gate = nullptr, const int* k_dev = nullptr) {
  DINER_CHECK_ARG(A && B && C, "gemm: null pointer argument");
  DINER_CHECK_ARG(!gate || !(flags & kExact), "gemm: a gated launch runs on the bf16x6 kernel");
  DINER_CHECK_ARG(M > 0 && N > 0 && K > 0 && lda > 0 && ldb > 0 && ldc >= N, "gemm: bad sizes M=%lld N=%d K=%d", M, N, K);
  DINER_CHECK_ARG((flags & ~127) == 0, "gemm: unknown flags 0x%x", flags);
  DINER_CHECK_ARG(k_split >= 1 && (k_split == 1 || (flags & kAtomic)), "gemm: split-K needs the atomic output flag");
  DINER_CHECK_ARG(!((flags & kAtomic) && mask), "gemm: a relu mask cannot be combined with atomic accumulation");
  DINER_CHECK_ARG(!(resid || rowsum) || !(flags & kExact), "gemm: resid / rowsum are epilogues of the bf16x6 kernel");
  DINER_CHECK_ARG(!rowsum || (flags & kTA), "gemm: rowsum needs op(A) stored with the contraction index outermost (kTA)");
  DINER_CHECK_ARG(!resid || !(flags & (kAtomic | kAccum)), "gemm: resid replaces the accumulate flags");
  DINER_CHECK_ARG(!(resid && rowsum), "gemm: resid and rowsum are separate epilogues");
  int chunk = (K + k_split - 1) / k_split;
  chunk = (chunk + XK - 1) / XK * XK;               // (a multiple of every kernel's k-tile)
  static const bool no_xcd = [] { const char* e = getenv("DINER_TRAIN_NO_XCD"); return e && *e == '1'; }();
  if (no_xcd) flags |= kNoXcdOrder;
  DINER_CHECK_ARG(!k_dev || !(flags & kExact), "gemm: a device-side contraction length runs on the bf16x6 kernel");
  GemmArgs g{A, B, C, bias, mask, resid, rowsum, M, N, K, lda, ldb, ldc, flags, chunk, gate, k_dev};
  if (!(flags & kExact)) {
    // split-bf16 on the bf16 matrix pipe (fp32-class products, see k_gemm_bf16x6); ragged and skinny shapes (lin_out: N = 4,
    // its adjoints: K = 4 / M = 4) ride along zero-padded -- a partly empty 128 x 128 tile is still faster than the fp32 kernels
    dim3 grid((N + XN - 1) / XN, (unsigned)((M + XM - 1) / XM), (K + chunk - 1) / chunk);
    if (k_dev) {                                                // chunk walkers (see the kernel's chunk loop)
      static const int walkers = [] { const char* e = getenv("DINER_TRAIN_GEMM_WALKERS"); const int v = e ? atoi(e) : 32; return v < 1 ? 1 : v; }();      // same-box A/B chunk x walkers, SB 4 step: one workgroup per 512-row chunk 102.7 ms, 128 x 64 100.5, 64 x 96 101.0, 256 x 32 100.1, 512 x 24 100.3, 128 x 128 101.6
      if (grid.z > (unsigned)walkers) grid.z = walkers;
    }
    if (resid) hipLaunchKernelGGL(k_gemm_bf16x6<1>, grid, dim3(256), 0, stream, g);
    else if (rowsum) hipLaunchKernelGGL(k_gemm_bf16x6<2>, grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL(k_gemm_bf16x6<0>, grid, dim3(256), 0, stream, g);
  } else if (N >= BN2 && M >= BM2) {
    const dim3 grid((N + BN2 - 1) / BN2, (unsigned)((M + BM2 - 1) / BM2), (K + chunk - 1) / chunk);
    hipLaunchKernelGGL(k_gemm128, grid, dim3(256), 0, stream, g);
  } else {
    const dim3 grid((N + BN - 1) / BN, (unsigned)((M + BM - 1) / BM), (K + chunk - 1) / chunk);
    hipLaunchKernelGGL(k_gemm, grid, dim3(256), 0, stream, g);
  }
  DINER_LAUNCH_OK();
  return 0;
}

extern "C" int diner_gemm_f32(const float* A, const float* B, float* C, long long M, int N, int K, int lda, int ldb, int ldc,
                              int flags, const float* bias, const float* mask, int k_split, void* stream) {
  return gemm_launch(A, B, C, M, N, K, lda, ldb, ldc, flags, bias, mask, k_split, (hipStream_t)stream);
}

extern "C" size_t diner_linear512_pack_bytes(void) { return kL512PackBytes; }

extern "C" int diner_linear512_f32(const float* X, const float* W, float* Y, long long M, int ldx, int ldy, int transpose, int flags,
                                   const float* bias, const float* resid, const float* mask, void* wpack, void* stream) {
  DINER_CHECK_ARG(X && W && Y && wpack && M > 0, "linear512: bad arguments");
  DINER_CHECK_ARG((flags & ~7) == 0, "linear512: unknown flags 0x%x", flags);
  DINER_CHECK_ARG(!(flags & 4) || !transpose, "linear512: the f16x3 arithmetic (flag 4) is the forward product's (transpose = 0)");
  auto al = [](const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; };
  DINER_CHECK_ARG(ldx >= 512 && ldy >= 512 && (ldx & 3) == 0 && (ldy & 3) == 0 && al(X) && al(Y) && al(bias) && al(resid) && al(mask) && al(wpack),
                  "linear512: row strides must be multiples of 4 (>= 512) and pointers 16-byte aligned");
  const bool f16 = (flags & 4) != 0;
  int rc = lin512_pack(W, f16 ? 2 : (transpose != 0 ? 1 : 0), wpack, (hipStream_t)stream);
  if (rc) return rc;
  Lin512Args a{X, wpack, Y, bias, resid, mask, M, ldx, ldy, flags & 3};
  return lin512_launch(a, (hipStream_t)stream, f16 ? 1 : 0);
}

extern "C" size_t diner_wgrad512_scratch_bytes(void) { return wgrad512_part_bytes(); }

extern "C" int diner_wgrad512_f32(const float* dY, const float* X, float* dW, float* db, long long M, int ldy, int ldx, int relu_x,
                                  void* scratch, void* stream) {
  DINER_CHECK_ARG(dY && X && dW && M > 0, "wgrad512: bad arguments");
  DINER_CHECK_ARG(ldy >= 512 && ldx >= 512 && (ldy & 1) == 0 && (ldx & 3) == 0 && (reinterpret_cast<size_t>(dY) & 7) == 0 &&
                  (reinterpret_cast<size_t>(X) & 15) == 0, "wgrad512: ldy even, ldx a multiple of 4 (both >= 512), dY 8-byte / X 16-byte aligned");
  DINER_CHECK_ARG((reinterpret_cast<size_t>(scratch) & 15) == 0, "wgrad512: scratch must be 16-byte aligned");
  return wgrad512_launch(dY, ldy, X, ldx, relu_x != 0, dW, db, M, (hipStream_t)stream, static_cast<float*>(scratch));
}

// gather: also the interpolated latent rows (`lat`: the operand of the layer-wise lin_z products and of their sample-space adjoint; the fused
// forward with the map-space adjoint needs it only in its gated repeat)
static int train_inputs(const DinerScene* scene, const float* xyz, const float* viewdirs, long long P, float freq_factor, float* feat, int* tap_row,
                        float* tap_w, float* lat, void* stream, bool gather);
extern "C" int diner_train_inputs_f32(const DinerScene* scene, const float* xyz, const float* viewdirs, long long P,
                                      float freq_factor, float* feat, int* tap_row, float* tap_w, float* lat, void* stream) {
  return train_inputs(scene, xyz, viewdirs, P, freq_factor, feat, tap_row, tap_w, lat, stream, true);
}
static int train_inputs(const DinerScene* scene, const float* xyz, const float* viewdirs, long long P, float freq_factor, float* feat, int* tap_row,
                        float* tap_w, float* lat, void* stream, bool gather) {
  DINER_CHECK_ARG(scene && xyz && viewdirs && feat && tap_row && tap_w && lat, "train_inputs: null pointer argument");
  DINER_CHECK_ARG(P > 0, "train_inputs: P must be positive");
  SceneDev sd;
  int rc = make_scene_dev(scene, &sd);
  if (rc) return rc;
  DINER_CHECK_ARG(scene->latent_cl && scene->depth && sd.C == kLatent, "train_inputs: latent (512 channels-last) / depth maps missing");
  FieldArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.xyz = xyz;
  fa.viewdirs = viewdirs;
  fa.K = 1;
  fa.P = P;
  fa.freq_factor = freq_factor;
  const long long waves = (P + 15) / 16 * sd.nv;
  hipLaunchKernelGGL(k_train_inputs, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, sd, fa, feat,
                     tap_row, tap_w);
  const long long cols = P * sd.nv;
  if (gather)
    hipLaunchKernelGGL(k_gather_latent, dim3((unsigned)((cols + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)scene->latent_cl, tap_row, tap_w, cols, lat, (const int*)nullptr);
  DINER_LAUNCH_OK();
  return 0;
}

extern "C" int diner_scatter_latent_grad_f32(const float* d_lat, const int* tap_row, const float* tap_w, long long cols,
                                             float* d_latent_cl, void* stream) {
  DINER_CHECK_ARG(d_lat && tap_row && tap_w && d_latent_cl, "scatter_latent_grad: null pointer argument");
  DINER_CHECK_ARG(cols > 0, "scatter_latent_grad: cols must be positive");
  return scatter_latent_launch(d_lat, tap_row, tap_w, cols, d_latent_cl, (hipStream_t)stream);
}

extern "C" int diner_channels_last_to_nchw_f32(const float* src, int n, long long HW, int C, float* dst, void* stream) {
  DINER_CHECK_ARG(src && dst && n > 0 && HW > 0 && C > 0 && n <= 65535 && (C + 63) / 64 <= 65535, "channels_last_to_nchw: bad arguments");
  hipLaunchKernelGGL(k_cl_to_nchw, dim3((unsigned)((HW + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)n), dim3(256), 0,
                     (hipStream_t)stream, src, HW, C, dst);
  DINER_LAUNCH_OK();
  return 0;
}

extern "C" int diner_view_mean_f32(const float* x, int nv, long long PC, float* y, int adjoint, void* stream) {
  DINER_CHECK_ARG(x && y && nv > 0 && PC > 0, "view_mean: bad arguments");
  if (adjoint) hipLaunchKernelGGL(k_view_bcast, dim3(grid1d(PC)), dim3(256), 0, (hipStream_t)stream, x, nv, PC, y, (const unsigned*)nullptr, (unsigned*)nullptr);
  else hipLaunchKernelGGL(k_view_mean, dim3(grid1d(PC)), dim3(256), 0, (hipStream_t)stream, x, nv, PC, y, (const int*)nullptr);
  DINER_LAUNCH_OK();
  return 0;
}

extern "C" int diner_colsum_f32(const float* dY, long long M, int N, int ld, float* db, void* stream) {
  DINER_CHECK_ARG(dY && db && M > 0 && N > 0 && ld >= N, "colsum: bad arguments");
  long long gy = (M + 255) / 256;
  if (gy > 256) gy = 256;
  hipLaunchKernelGGL(k_colsum, dim3((N + 63) / 64, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, dY, M, N, ld, db);
  DINER_LAUNCH_OK();
  return 0;
}

extern "C" int diner_field_act_f32(const float* raw, const float* dout, long long P, int ld, float* out, void* stream) {
  DINER_CHECK_ARG(raw && out && P > 0 && ld >= 4, "field_act: bad arguments");
  if (dout) hipLaunchKernelGGL(k_field_act_bwd, dim3(grid1d(P)), dim3(256), 0, (hipStream_t)stream, raw, dout, P, ld, out);
  else hipLaunchKernelGGL(k_field_act, dim3(grid1d(P)), dim3(256), 0, (hipStream_t)stream, raw, P, ld, out, (const int*)nullptr);
  DINER_LAUNCH_OK();
  return 0;
}

extern "C" int diner_composite_bwd_f32(const float* field, const float* z, const float* rays, int NR, int K, int white_bkgd,
                                       const float* g_rgb, const float* g_depth, float* d_field, void* stream) {
  DINER_CHECK_ARG(field && z && rays && g_rgb && d_field, "composite_bwd: null pointer argument");
  DINER_CHECK_ARG(NR > 0 && K > 0 && K <= kCompBwdMaxK, "composite_bwd: bad sizes NR=%d K=%d (K <= %d)", NR, K, kCompBwdMaxK);
  hipLaunchKernelGGL(k_composite_bwd, dim3((NR + 63) / 64), dim3(64), 0, (hipStream_t)stream, field, z, rays, NR, K,
                     white_bkgd, g_rgb, g_depth, d_field);
  DINER_LAUNCH_OK();
  return 0;
}

// ---- the whole forward / backward of the field as one call each (what diner_amd/train.py does call by call; one entry
// saves ~100 host round trips per step, which matter at the reference's 128-ray training batch) -------------------------
namespace {
// packed-weights slots of the workspace: fc_0 / fc_1 of the 5 blocks + the 3 lin_z, forward and transposed
constexpr int kWPackSlots = 4 * 13;      // forward, transposed (bf16x6); forward, transposed in fp16 hi / lo (f16x3)
// ints of the flag block: [0, 13) range flags of the forward's f16x3 products (per weight slot; also raised when the step's weights do not
// fit, kFlagWBad) -- the weight gradient of a slot runs in f16x3 only if its flag stayed down (its x operand is that product's);
// kFlagWBad: some 16 |w| is no finite fp16 value (set while packing); [kAmax0, +64): bit patterns of max |dy| of the backward's operands
enum { kFlagWBad = 13, kAmax0 = 32, kFlagInts = 128 };
enum { kSlotFc0 = 0, kSlotFc1 = 5, kSlotLinZ = 10 };
struct TrainWs {               // float offsets into the workspace
  size_t feat, tap_row, tap_w, lat, X[5], H[5], x_last, raw, wpack, flags, bX[5], bH[5], saved_total, d_raw, dx, dH, d_lat, wgpart, total;
};      // [0, saved_total): what the forward leaves for the backward; d_raw .. wgpart (offsets from scratch_base): work buffers of either call, nothing in them lives
        // from the forward to the backward (round 5: they may sit in a buffer of their own that the objects of a step share -- `scratch` of
        // the _s entry points; one workspace: they follow the saved part).  bX / bH (round 5): the relu decisions of X[b] / H[b] as bits, 16 dwords per row (Lin512Args.maskbits): what the data gradients read
TrainWs train_ws(long long P, int nv) {
  TrainWs w;
  const size_t cols = (size_t)P * nv;
  size_t o = 0;
  auto take = [&](size_t n) { const size_t at = o; o += (n + 63) / 64 * 64; return at; };
  w.feat = take(cols * kDInPad);
  w.tap_row = take(cols * 4);
  w.tap_w = take(cols * 4);
  w.lat = take(cols * kLatent);
  for (int b = 0; b < 5; ++b) {
    const size_t m = b < 3 ? cols : (size_t)P;
    w.X[b] = take(m * kHidden);
    w.H[b] = take(m * kHidden);
  }
  w.x_last = take((size_t)P * kHidden);
  w.raw = take((size_t)P * 4);
  w.wpack = take(kWPackSlots * (kL512PackBytes / sizeof(float)));      // packed 512 x 512 weights of train_lin512.hip
  w.flags = take(kFlagInts);                                       // flag block (ints), see kFlagWBad
  for (int b = 0; b < 5; ++b) {
    const size_t m = b < 3 ? cols : (size_t)P;
    w.bX[b] = take(m * 16);
    w.bH[b] = take(m * 16);
  }
  w.saved_total = o;
  o = 0;                                                           // the work buffers: offsets from THEIR base (scratch_base)
  w.d_raw = take((size_t)P * 4);
  w.dx = take(cols * kHidden);
  w.dH = take(cols * kHidden);
  w.d_lat = take(cols * kLatent);
  w.wgpart = take(13 * (wgrad512_part_bytes() / sizeof(float)));     // per-chunk partial weight gradients of the 13 512 x 512 layers (train_wgrad512.hip)
  w.total = w.saved_total + o;
  return w;
}
// base pointer for the work buffers' offsets: `scratch` (a buffer of its own, diner_field_train_workspace_split) or the workspace itself
float* scratch_base(float* ws, void* scratch, const TrainWs& w) { return scratch ? (float*)scratch : ws + w.saved_total; }
int check_train_params(const DinerMlpParams* p, bool poscode) { return check_mlp_config(p, "field_train", poscode); }
// DINER_TRAIN_LIN512=0 routes the 512 x 512 layer products back to the general kernel (A/B measurement)
bool use_lin512() {
  static const bool on = [] { const char* e = getenv("DINER_TRAIN_LIN512"); return !(e && *e == '0'); }();
  return on;
}
bool use_lin_out() {            // DINER_TRAIN_LINOUT=0: lin_out and its adjoints back on the general kernel (A/B measurement)
  static const bool on = [] { const char* e = getenv("DINER_TRAIN_LINOUT"); return !(e && *e == '0'); }();
  return on;
}
// The forward products of the 512 x 512 layers in the f16x3 arithmetic of the inference kernels (two fp16 planes per operand, three product
// terms: half the MFMAs of bf16x6, same fp32-class accuracy on activations) with a bf16x6 launch behind each that does nothing unless the
// f16x3 one met an operand beyond the fp16 range.  DINER_TRAIN_FWD_F16X3=0: bf16x6 forward (A/B measurement).
bool use_fwd_f16() {
  static const bool on = [] { const char* e = getenv("DINER_TRAIN_FWD_F16X3"); return !(e && *e == '0'); }();
  return on;
}
// Round 4: the data- and weight-gradient products of the 512 x 512 layers in the f16x3 arithmetic as well (3 MFMAs per fp32 product instead of
// bf16x6's 6, on two thirds of the step's FLOPs).  Loss gradients span many decades, so dy is staged times a power of two taken from its
// maximum (tracked by the epilogue of the product that wrote it): nothing leaves the fp16 range, no repeat is needed for the operand.  What
// can still not fit: the weights (16 |w| beyond fp16, kFlagWBad) and an activation operand of a weight gradient (the forward flag of that
// layer) -- those launches return at once and their bf16x6 twins, issued behind them, do the work.  DINER_TRAIN_BWD_F16X3=0: bf16x6 backward.
bool use_bwd_f16() {
  static const bool on = [] { const char* e = getenv("DINER_TRAIN_BWD_F16X3"); return !(e && *e == '0'); }();
  return on;
}
bool use_wgrad512() {          // DINER_TRAIN_WGRAD512=0: weight gradients back on the general kernel (A/B measurement)
  static const bool on = [] { const char* e = getenv("DINER_TRAIN_WGRAD512"); return !(e && *e == '0'); }();
  return on;
}
bool lin512_ok(const float* x, int ldx, const float* y, int ldy, const float* resid, const float* mask) {
  auto al = [](const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; };
  return use_lin512() && (ldx & 3) == 0 && (ldy & 3) == 0 && al(x) && al(y) && al(resid) && al(mask);
}
void* wpack_slot(float* ws, const TrainWs& w, int slot, bool transposed, bool f16 = false) {
  return reinterpret_cast<char*>(ws + w.wpack) + (size_t)(slot + (transposed ? 13 : 0) + (f16 ? 26 : 0)) * kL512PackBytes;
}
// f16x3 arithmetic of one backward layer (linear_bwd): operand scale source, where the maximum of dx goes, the two flags, the f16 pack of W^T
struct BwdArith {
  const unsigned* amax_dy;
  unsigned* amax_dx;
  const int* wflag;      // kFlagWBad
  const int* xflag;      // the forward flag of the layer (its x operand left the fp16 range, or wflag)
  const void* Wt_f16;
};
// adjoint of y = act(x) W^T + b: dW = dy^T act(x) (split-K atomics into zeroed dW), db = column sums, dx (+)= (dy W) [masked]
int linear_bwd(const float* dy, int ldy, const float* x, int ldx, bool relu_in, const float* W, float* dW, float* db,
               long long M, int N, int K, float* dx, const float* dx_mask, bool dx_accum, hipStream_t st,
               const void* Wt_packed = nullptr, float* wgpart = nullptr, WgReduceJob* defer = nullptr, const BwdArith* f16 = nullptr,
               const unsigned* dx_maskbits = nullptr) {      // dx_maskbits: the decisions of dx_mask as bits, read instead of it by the 512-kernels
  // split-K so that the 16 output tiles of a 512 x 512 weight gradient become 500-1000 workgroups of >= 15 k-tiles each (measured:
  // 128 / 512 / 2048-ray steps 5.70 / 15.6 / 53.1 ms with M / 1024 capped at 32, 5.07 / 14.5 / 51.8 ms with M / 480 capped at 64);
  // round 3: M / 640 -- the row-sum instance of the kernel runs two workgroups per CU, 16 tiles x 32 chunks fill the chip once for the
  // reference batch (128 / 512 / 2048-ray steps 4.37 / 10.56 / 37.6 ms with 480, 4.00 / 10.31 / 37.5 ms with 640);
  // DINER_TRAIN_WGRAD_ROWS / _CAP override (measurement aid)
  static const long long rows_per_chunk = [] { const char* e = getenv("DINER_TRAIN_WGRAD_ROWS"); return e ? atoll(e) : 640LL; }();
  static const long long cap = [] { const char* e = getenv("DINER_TRAIN_WGRAD_CAP"); return e ? atoll(e) : 64LL; }();
  long long split = M / rows_per_chunk;
  split = split < 1 ? 1 : (split > cap ? cap : split);
  // a dx produced by the general GEMM keeps no maximum of its own: one pass over it fills the slot the f16x3 consumers scale by
  auto dx_amax = [&]() {
    if (f16 && f16->amax_dx && dx) hipLaunchKernelGGL(k_absmax, dim3(grid1d(M * (long long)K)), dim3(256), 0, st, dx, M * (long long)K, f16->amax_dx);
  };
  if (N == 512 && K == 512 && M >= 256 && use_wgrad512() && (ldy & 1) == 0 && (ldx & 3) == 0 &&
      (reinterpret_cast<size_t>(dy) & 7) == 0 && (reinterpret_cast<size_t>(x) & 15) == 0 &&
      (reinterpret_cast<size_t>(dW) & 15) == 0 && (reinterpret_cast<size_t>(db) & 15) == 0) {      // (the summing pass stores dW / db as 16-byte vectors)
    // the 512 x 512 layers: persistent feature-sliced kernel (train_wgrad512.hip), bias gradient = row sums of its dy operand
    // with the scratch: partial tiles + one summing pass that OVERWRITES dW / db (no zeroing, no atomics)
    if (!wgpart) {
      DINER_HIP_OK(hipMemsetAsync(dW, 0, (size_t)N * K * sizeof(float), st));
      DINER_HIP_OK(hipMemsetAsync(db, 0, (size_t)N * sizeof(float), st));
    }
    // the data gradient of the layer rides in the same launch (train_512.hip): dx = dy W on k_lin512's bodies, W packed transposed
    static const bool one_launch = [] { const char* e = getenv("DINER_TRAIN_BWD_FUSED"); return !(e && *e == '0'); }();
    const bool dgrad512 = dx && Wt_packed && lin512_ok(dy, ldy, dx, K, nullptr, dx_mask);
    Lin512Args da{dy, Wt_packed, dx, nullptr, nullptr, dx_mask, M, ldy, K, dx_accum ? kL512Accum : 0};
    if (dx_maskbits) { da.mask = nullptr; da.maskbits = dx_maskbits; }
    if (f16 && wgpart && defer && one_launch && (dgrad512 || !dx)) {
      // the f16x3 launch (works unless a flag is up) and its bf16x6 twin (works only then); an accumulating data gradient is safe: exactly
      // one of the two runs, decided by a flag that does not change during the step
      Lin512Args dh = da;
      dh.Wp = f16->Wt_f16;
      dh.amax_in = f16->amax_dy;
      dh.amax_out = f16->amax_dx;
      dh.skip = f16->wflag;
      const WgradArith wa{1, f16->amax_dy, f16->xflag, nullptr};
      int rcf = wgrad512_launch(dy, ldy, x, ldx, relu_in, dW, db, M, st, wgpart, true, defer, dx ? &dh : nullptr, &wa);
      if (rcf) return rcf;
      da.gate = f16->wflag;
      da.amax_out = f16->amax_dx;
      const WgradArith wb{0, nullptr, nullptr, f16->xflag};
      return wgrad512_launch(dy, ldy, x, ldx, relu_in, dW, db, M, st, wgpart, true, defer, dx ? &da : nullptr, &wb);
    }
    if (f16) da.amax_out = f16->amax_dx;          // (bf16x6 here, but a later product may still want the maximum)
    int rcw = wgrad512_launch(dy, ldy, x, ldx, relu_in, dW, db, M, st, wgpart, wgpart != nullptr, wgpart ? defer : nullptr,
                              dgrad512 && one_launch ? &da : nullptr);
    if (rcw) return rcw;
    if (dgrad512) return one_launch ? 0 : lin512_launch(da, st);
    if (dx) {
      int rcd = gemm_launch(dy, W, dx, M, K, N, ldy, K, K, dx_accum ? kAccum : 0, nullptr, dx_mask, 1, st);
      if (rcd) return rcd;
      dx_amax();
    }
    return 0;
  }
  DINER_HIP_OK(hipMemsetAsync(dW, 0, (size_t)N * K * sizeof(float), st));
  DINER_HIP_OK(hipMemsetAsync(db, 0, (size_t)N * sizeof(float), st));
  // few output tiles (lin_in: 512 x 55 = 4 tiles): more row chunks so that the launch still has ~512 workgroups (>= 128 rows each);
  // 32 chunks = 128 workgroups took 95 us for the reference batch's 20480 rows
  const long long n_tiles = (long long)((N + 127) / 128) * ((K + 127) / 128);
  if (n_tiles < 16) {
    long long want = 512 / n_tiles, most = M / 128;
    if (want > 128) want = 128;
    if (want > most) want = most;
    if (split < want) split = want;
  }
  // the bias gradient (column sums of dy) rides on the weight-gradient product: dy^T is its A operand (no k_colsum pass over dy)
  int rc = gemm_launch(dy, x, dW, N, K, M, ldy, ldx, K, kTA | kAtomic | (relu_in ? kReluB : 0), nullptr, nullptr, (int)split, st,
                       nullptr, db);
  if (rc) return rc;
  if (dx && Wt_packed && N == 512 && K == 512 && lin512_ok(dy, ldy, dx, K, nullptr, dx_mask)) {
    // dx = dy W on the feature-sliced kernel (train_lin512.hip): D[k][row] = sum_f W[f][k] dy[row][f], W packed transposed
    Lin512Args a{dy, Wt_packed, dx, nullptr, nullptr, dx_mask, M, ldy, K, dx_accum ? kL512Accum : 0};
    if (dx_maskbits) { a.mask = nullptr; a.maskbits = dx_maskbits; }
    if (f16) a.amax_out = f16->amax_dx;            // (fewer than 256 rows or an odd alignment: bf16x6 here, the consumers may still run f16x3)
    return lin512_launch(a, st);
  }
  if (dx) {
    rc = gemm_launch(dy, W, dx, M, K, N, ldy, K, K, dx_accum ? kAccum : 0, nullptr, dx_mask, 1, st);
    if (!rc) dx_amax();
  }
  return rc;
}
}  // namespace

extern "C" size_t diner_field_train_workspace_bytes(long long P, int nv) {
  if (P <= 0 || nv <= 0) return 0;
  return train_ws(P, nv).total * sizeof(float);
}

// test aid: where the forward keeps the pre-activations inside the workspace (float offsets): X[0..4] (residual stream entering block b;
// rows = P nv for b < 3, P behind the view mean), H[0..4] (fc_0 outputs), x_last (entering lin_out), raw (lin_out's outputs)
extern "C" int diner_field_train_ws_layout(long long P, int nv, long long* float_offsets, int n) {
  DINER_CHECK_ARG(P > 0 && nv > 0 && float_offsets && n >= 12, "field_train_ws_layout: bad arguments (12 offsets)");
  const TrainWs w = train_ws(P, nv);
  for (int b = 0; b < 5; ++b) {
    float_offsets[b] = (long long)w.X[b];
    float_offsets[5 + b] = (long long)w.H[b];
  }
  float_offsets[10] = (long long)w.x_last;
  float_offsets[11] = (long long)w.raw;
  return 0;
}

// The layer-wise forward.  gate != null: the repeat behind the fused forward (below) -- inputs, packed weights and the flag block are in
// place, every launch returns at once unless *gate != 0 (the fused kernels met an activation beyond the fp16 range).
// ws / sc / w: the saved part, the work buffers and the offsets into them -- of one object's workspace, or (ABI v6, the batched step) one
// object's view of the step's workspace (obj_view below; wpack and flags are the step's).  pack: pack the step's weights and clear the flag block
// (the first object of a step; gate == null only).
static int forward_layerwise(const DinerScene* scene, const DinerMlpParams* p, const float* xyz, const float* viewdirs, long long P,
                             float* out, float* ws, float* sc, const TrainWs& w, void* stream, const int* gate, bool pack = true,
                             bool lat_missing = false) {
  int rc = 0;
  hipStream_t st = (hipStream_t)stream;
  const long long cols = P * scene->nv;
  if (gate && lat_missing)      // the repeat behind a fused forward that skipped the gather: the interpolated latent rows, only if the repeat runs
    hipLaunchKernelGGL(k_gather_latent, dim3((unsigned)((cols + 3) / 4)), dim3(256), 0, st, (const float*)scene->latent_cl,
                       (const int*)(ws + w.tap_row), ws + w.tap_w, cols, ws + w.lat, gate);
  if (!gate) {
    rc = diner_train_inputs_f32(scene, xyz, viewdirs, P, p->freq_factor, ws + w.feat, (int*)(ws + w.tap_row), ws + w.tap_w, ws + w.lat, stream);
    if (rc) return rc;
  }
  // the 512 x 512 layers: weights packed once per step (three bf16 planes in the consuming wave's order; one launch for the 13 matrices
  // in both orientations -- the backward call of the step reads the transposed ones from the workspace), products on k_lin512
  if (use_lin512() && !gate && pack) {
    PackMany pm;
    for (int b = 0; b < 5; ++b) { pm.W[kSlotFc0 + b] = p->fc0_w[b]; pm.W[kSlotFc1 + b] = p->fc1_w[b]; }
    for (int b = 0; b < 3; ++b) pm.W[kSlotLinZ + b] = p->lin_z_w[b];
    const bool f16_packs = use_fwd_f16() || use_bwd_f16();
    if (f16_packs) DINER_HIP_OK(hipMemsetAsync(ws + w.flags, 0, 16 * sizeof(int), st));
    if ((rc = lin512_pack_many(pm, 13, ws + w.wpack, st, f16_packs ? 4 : 2, reinterpret_cast<int*>(ws + w.flags) + kFlagWBad))) return rc;
  }
  // seg2_slot >= 0: a second contraction segment in the same product (Lin512Args.X2: + x2 W[seg2_slot]^T + bias2, no relu on x2)
  auto lin = [&](const float* x, int ldx, const float* W, const float* b, float* y, long long M, int N, int K, bool relu,
                 bool accum, const float* resid = nullptr, int slot = -1, const float* resid2 = nullptr, const float* x2 = nullptr,
                 int seg2_slot = -1, const float* bias2 = nullptr) {
    if (slot >= 0 && N == 512 && K == 512 && lin512_ok(x, ldx, y, N, resid, nullptr) && (reinterpret_cast<size_t>(b) & 15) == 0 &&
        (reinterpret_cast<size_t>(resid2) & 15) == 0) {      // (the epilogue reads bias / residuals as 16-byte vectors)
      void* wp = wpack_slot(ws, w, slot, false);
      Lin512Args a{x, wp, y, b, resid, nullptr, M, ldx, N, (relu ? kL512ReluIn : 0) | (accum ? kL512Accum : 0)};
      a.resid2 = resid2;
      if (x2) {
        a.X2 = x2;
        a.Wp2 = wpack_slot(ws, w, seg2_slot, false);
        a.bias2 = bias2;
        a.relu2 = 0;
      }
      if (use_fwd_f16() && !accum) {             // (an accumulating product cannot be run twice: lin_z stays on bf16x6)
        int* flag = reinterpret_cast<int*>(ws + w.flags) + slot;
        Lin512Args h = a;
        h.Wp = wpack_slot(ws, w, slot, false, true);
        if (x2) {
          h.Wp2 = wpack_slot(ws, w, seg2_slot, false, true);
          h.ovf2 = reinterpret_cast<int*>(ws + w.flags) + seg2_slot;      // (the weight gradient of the fused layer looks at its own slot)
        }
        h.ovf = flag;
        h.skip = reinterpret_cast<int*>(ws + w.flags) + kFlagWBad;      // weights beyond the fp16 split: raises the flag and returns
        h.gate = gate;
        int hrc = lin512_launch(h, st, 1);
        if (hrc) return hrc;
        a.gate = flag;                           // the bf16x6 product below runs only if the f16x3 one left the range
        a.gate2 = gate;                          // (repeat mode: the lin_z flags may be up from the map projection of the fused forward)
      } else {
        a.gate = gate;
      }
      return lin512_launch(a, st);
    }
    DINER_CHECK_ARG(!resid2 && !x2, "field_train_forward: second residual / second segment off the 512-kernel path");
    return gemm_launch(x, W, y, M, N, K, ldx, K, N, kTB | (relu ? kReluA : 0) | (accum ? kAccum : 0), b, nullptr, 1, st, resid, nullptr, gate);
  };
  // The lin_z term of block b, Z_b = lat Wz_b^T + bz_b, is a product of its own into a scratch buffer (d_lat is free in the forward) and
  // enters the residual stream through the epilogue of fc_1 of block b - 1, the product that writes X[b] (second residual; block 0: below): no accumulating product is left in the forward, so every 512 x 512 product can run in the f16x3 arithmetic with
  // its gated bf16x6 repeat.  (Without the 512-kernels: lin_z accumulates onto X[b] as before.)
  float* Z = sc + w.d_lat;
  const bool z_sep = use_lin512() && lin512_ok(ws + w.lat, kLatent, Z, kHidden, nullptr, nullptr);
  auto lin_z = [&](int b, float* dst) { return lin(ws + w.lat, kLatent, p->lin_z_w[b], p->lin_z_b[b], dst, cols, kHidden, kLatent, false, false, nullptr, kSlotLinZ + b); };
  // block 0: lin_in (general kernel, plain instance, runs once) writes into the scratch buffer and the lin_z product of block 0 takes it as its
  // residual: X[0] = lat Wz_0^T + bz_0 + lin_in(feat).  (lin_in accumulating onto a Z_0 written first cost 0.88 ms for the 2048-ray batch --
  // it re-reads and re-writes X[0] -- against 0.37 ms plain; the product's residual pass adds 0.14 ms; the kernel's residual instance holds
  // fewer waves per SIMD: 0.77 ms.)
  if ((rc = lin(ws + w.feat, kDInPad, p->lin_in_w, p->lin_in_b, z_sep ? Z : ws + w.X[0], cols, kHidden, kDIn, false, false))) return rc;
  if (z_sep && (rc = lin(ws + w.lat, kLatent, p->lin_z_w[0], p->lin_z_b[0], ws + w.X[0], cols, kHidden, kLatent, false, false, Z, kSlotLinZ + 0))) return rc;
  for (int b = 0; b < 5; ++b) {
    const long long M = b < 3 ? cols : P;
    float* X = ws + w.X[b];
    if (!z_sep && b < 3 && (rc = lin(ws + w.lat, kLatent, p->lin_z_w[b], p->lin_z_b[b], X, M, kHidden, kLatent, false, true, nullptr, kSlotLinZ + b))) return rc;
    if ((rc = lin(X, kHidden, p->fc0_w[b], p->fc0_b[b], ws + w.H[b], M, kHidden, kHidden, true, false, nullptr, kSlotFc0 + b))) return rc;
    // next residual stream: X + fc_1(relu(H)); the view mean comes after block 2
    float* nx = b == 4 ? ws + w.x_last : (b == 2 ? sc + w.dx : ws + w.X[b + 1]);      // (dx doubles as scratch in the forward)
    const bool z_next = z_sep && b + 1 < 3;
    // round 4: the lin_z term of block b + 1 as a SECOND CONTRACTION SEGMENT of this block's fc_1 product (K = 512 + 512: relu(H) W1^T +
    // lat Wz^T + both biases + X in one pass over the rows) -- no Z tensor written and read back (2 of this block's ~8 tensor passes).
    // DINER_TRAIN_FUSE_Z=0: Z_{b+1} as a product of its own into the scratch buffer, entering through the second residual (round 3)
    static const bool fuse_z = [] { const char* e = getenv("DINER_TRAIN_FUSE_Z"); return !(e && *e == '0'); }();
    const bool fused = z_next && fuse_z && use_fwd_f16() && (reinterpret_cast<size_t>(p->lin_z_b[b + 1]) & 15) == 0;
    if (z_next && !fused && (rc = lin_z(b + 1, Z))) return rc;
    // (the residual enters through the product's epilogue: no copy of X)
    if (fused) {
      if ((rc = lin(ws + w.H[b], kHidden, p->fc1_w[b], p->fc1_b[b], nx, M, kHidden, kHidden, true, false, X, kSlotFc1 + b, nullptr,
                    ws + w.lat, kSlotLinZ + b + 1, p->lin_z_b[b + 1]))) return rc;
    } else if ((rc = lin(ws + w.H[b], kHidden, p->fc1_w[b], p->fc1_b[b], nx, M, kHidden, kHidden, true, false, X, kSlotFc1 + b, z_next ? Z : nullptr))) return rc;
    if (b == 2)
      hipLaunchKernelGGL(k_view_mean, dim3(grid1d(P * kHidden)), dim3(256), 0, st, nx, scene->nv, P * kHidden, ws + w.X[3], gate);
  }
  if (use_lin_out() && (reinterpret_cast<size_t>(p->lin_out_w) & 15) == 0 && (reinterpret_cast<size_t>(p->lin_out_b) & 15) == 0 &&
      (reinterpret_cast<size_t>(out) & 15) == 0) {
    hipLaunchKernelGGL(k_lin_out_fwd, dim3(grid1d(P, 4, 1024)), dim3(256), 0, st, ws + w.x_last, p->lin_out_w, p->lin_out_b, P, ws + w.raw, out, gate);
  } else {
    if ((rc = lin(ws + w.x_last, kHidden, p->lin_out_w, p->lin_out_b, ws + w.raw, P, 4, kHidden, true, false))) return rc;
    hipLaunchKernelGGL(k_field_act, dim3(grid1d(P)), dim3(256), 0, st, ws + w.raw, P, 4, out, gate);
  }
  {   // the relu decisions of the ten saved pre-activations as bits, for the backward's data gradients (the fused forward writes them itself)
    MakeBits mb;
    for (int b = 0; b < 5; ++b) {
      const long long m = b < 3 ? cols : P;
      mb.src[2 * b] = ws + w.X[b];      mb.dst[2 * b] = reinterpret_cast<unsigned*>(ws + w.bX[b]);      mb.rows[2 * b] = m;
      mb.src[2 * b + 1] = ws + w.H[b];  mb.dst[2 * b + 1] = reinterpret_cast<unsigned*>(ws + w.bH[b]);  mb.rows[2 * b + 1] = m;
    }
    hipLaunchKernelGGL(k_make_bits, dim3(grid1d(cols * 16, 256, 2048), 10), dim3(256), 0, st, mb, gate);
  }
  DINER_LAUNCH_OK();
  return 0;
}

extern "C" int diner_field_train_forward_s_f32(const DinerScene* scene, const DinerMlpParams* p, const float* xyz, const float* viewdirs,
                                               long long P, float* out, void* workspace, void* scratch, void* stream) {
  DINER_CHECK_ARG(scene && xyz && viewdirs && out && workspace && P > 0, "field_train_forward: bad arguments");
  int rc = check_train_params(p, true);
  if (rc) return rc;
  const TrainWs w = train_ws(P, scene->nv);
  return forward_layerwise(scene, p, xyz, viewdirs, P, out, (float*)workspace, scratch_base((float*)workspace, scratch, w), w, stream, nullptr);
}
extern "C" int diner_field_train_forward_f32(const DinerScene* scene, const DinerMlpParams* p, const float* xyz,
                                             const float* viewdirs, long long P, float* out, void* workspace, void* stream) {
  return diner_field_train_forward_s_f32(scene, p, xyz, viewdirs, P, out, workspace, nullptr, stream);
}

// bytes of the two parts of the training workspace: what the forward keeps for the backward (per object) and the work buffers of either call
// (may be one buffer shared by the objects of a step: calls on one stream use it one after the other)
extern "C" int diner_field_train_workspace_split(long long P, int nv, size_t* saved_bytes, size_t* scratch_bytes) {
  DINER_CHECK_ARG(P > 0 && nv > 0 && saved_bytes && scratch_bytes, "field_train_workspace_split: bad arguments");
  const TrainWs w = train_ws(P, nv);
  *saved_bytes = w.saved_total * sizeof(float);
  *scratch_bytes = (w.total - w.saved_total) * sizeof(float);
  return 0;
}

// ---- training forward on the INFERENCE kernels (round 5, experiment behind DINER_TRAIN_FUSED_FWD=1 of the Python host) ------------------
// The layer-wise forward above reads and writes every activation tensor of the 13 products through HBM (44 KB per per-view row); the
// inference path computes the same network with the activations on chip (k_field_pre_h3n / k_field_post_h3n, f16x3).  Here those kernels
// run in their storing variants (k_train_fwd_pre / k_train_fwd_post, mlp_h3n.hip): the ten pre-activation tensors, the stream entering
// lin_out and lin_out's raw outputs go to the SAME places of the workspace the layer-wise forward uses, so the backward is unchanged.  The
// gather inputs (MLP inputs, tap rows / weights, interpolated latent) and the packed weights of the backward's products are made as before.
// Needs the packed-weights handle of the step's weights and the latent map projected with them: made here into latent_proj_out (below), or
// by the caller (diner_scene_prepare_f32 with THIS handle) when latent_proj_out is null.  An activation beyond the fp16 range raises a flag
// (diner_field_train_fused_overflowed reads it back: a test aid) and the layer-wise forward, enqueued behind the fused kernels and gated on that
// flag, redoes the object -- no host synchronisation.
int field_forward_save(const DinerScene* scene, const DinerMlp* mlp, const float* xyz, const float* viewdirs, long long P, float* out,
                       void* workspace, const SaveActs& sv, int** overflow_flag, hipStream_t stream);
const float* mlp_hoist_bias(const DinerMlp* mlp);
namespace {
__global__ void k_copy_flag(const int* __restrict__ src, int* __restrict__ dst) { *dst = *src; }
enum { kFlagFusedOvf = 14 };
}  // namespace

int field_forward_save_supported(const DinerScene* scene, const DinerMlp* mlp);
// one object of a step: its inputs, (first: the step's packed weights + flag block,) its projected maps, the fused kernels, the gated repeat
static int fused_forward_core(const DinerScene* scene, const DinerMlp* mlp, const DinerMlpParams* p, const float* xyz, const float* viewdirs,
                              long long P, float* out, float* ws, float* sc, const TrainWs& w, float* latent_proj_out, hipStream_t st, bool first,
                              bool gather_lat = true) {
  int rc = field_forward_save_supported(scene, mlp);      // host-known reasons to keep the layer-wise forward: before anything is enqueued
  if (rc) return rc;
  void* stream = (void*)st;
  rc = train_inputs(scene, xyz, viewdirs, P, p->freq_factor, ws + w.feat, (int*)(ws + w.tap_row), ws + w.tap_w, ws + w.lat, stream, gather_lat);
  if (rc) return rc;
  if (first) {   // packed weights of the backward's products + the flag block, as the layer-wise forward leaves them
    PackMany pm;
    for (int b = 0; b < 5; ++b) { pm.W[kSlotFc0 + b] = p->fc0_w[b]; pm.W[kSlotFc1 + b] = p->fc1_w[b]; }
    for (int b = 0; b < 3; ++b) pm.W[kSlotLinZ + b] = p->lin_z_w[b];
    DINER_HIP_OK(hipMemsetAsync(ws + w.flags, 0, 16 * sizeof(int), st));
    if ((rc = lin512_pack_many(pm, 13, ws + w.wpack, st, 4, reinterpret_cast<int*>(ws + w.flags) + kFlagWBad))) return rc;
  }
  // latent_proj_out: the projection of the whole latent map through lin_z[0..2] (what diner_scene_prepare_f32 makes on the exact fp32
  // kernel, 112 TFLOP/s: 3.2 ms for four 264 x 214 maps -- a training step pays it per object, the latent is new every step) as three
  // products of the 512-layer kernel in the arithmetic of the other training products: f16x3, the bf16x6 twin behind a flag
  DinerScene own = *scene;
  if (latent_proj_out) {
    DINER_CHECK_ARG(scene->latent_cl && scene->C == kLatent && scene->Hf > 0 && scene->Wf > 0, "field_train_forward_fused: channels-last latent missing");
    const long long rows = (long long)scene->nv * scene->Hf * scene->Wf;
    DINER_CHECK_ARG(lin512_ok(scene->latent_cl, kLatent, latent_proj_out, kHidden, nullptr, nullptr), "field_train_forward_fused: unaligned latent / projection buffer");
    int* flags = reinterpret_cast<int*>(ws + w.flags);
    // round 6: project only the texel rows this batch touches (see k_mark_rows).  Work buffers in the backward's dH region, free in the
    // forward: [mark rows][idx rows][count, dense][Xc cap x 512][Yc 3 x cap x 512], cap = cols / 4 rows (the region holds cols x 512 floats).
    // More marked rows than cap (a batch spread over most of a small map): `dense` goes up and the whole-map projection below, gated on it, runs.
    const char* e_touched = getenv("DINER_TRAIN_PROJ_TOUCHED");      // (read per call: the tests switch them)
    const char* e_cap = getenv("DINER_TRAIN_PROJ_CAP");              // test aid: a small cap forces the overflow of the list -> the gated dense projection
    const bool touched_only = !(e_touched && *e_touched == '0');
    const long long cap_env = e_cap ? atoll(e_cap) : 0ll;
    const long long cols = P * scene->nv;
    long long cap = (cols - (2 * rows + 64 + 511) / 512 - 2) / 4;      // Xc + 3 planes of Yc behind the two int arrays (rows of 512 floats)
    if (cap_env > 0 && cap_env < cap) cap = cap_env;
    const bool sparse = touched_only && cap >= 64 && rows < (1ll << 31) && cols * 4 < (1ll << 31);
    int* mark = reinterpret_cast<int*>(sc + w.dH);
    int* idx = mark + rows;
    int* cnt = idx + rows;                 // cnt[0] = marked rows, cnt[1] = dense
    float* Xc = reinterpret_cast<float*>(cnt + 64);
    Xc += (64 - ((reinterpret_cast<size_t>(Xc) / sizeof(float)) & 63)) & 63;      // (16-byte alignment and then some)
    float* Yc = Xc + (size_t)(sparse ? cap : 0) * kLatent;
    const int* dense = sparse ? cnt + 1 : nullptr;
    if (sparse) {
      DINER_HIP_OK(hipMemsetAsync(mark, 0, (size_t)rows * sizeof(int), st));
      DINER_HIP_OK(hipMemsetAsync(cnt, 0, 64 * sizeof(int), st));
      hipLaunchKernelGGL(k_mark_rows, dim3(grid1d(cols * 4)), dim3(256), 0, st, (const int*)(ws + w.tap_row), cols * 4, mark);
      hipLaunchKernelGGL(k_compact_rows, dim3(grid1d(rows)), dim3(256), 0, st, mark, (int)rows, (int)cap, idx, cnt, cnt + 1);
      hipLaunchKernelGGL(k_move_rows, dim3(1024, 1), dim3(256), 0, st, (const float*)scene->latent_cl, Xc, idx, cnt, (int)cap, 0, (size_t)0, (size_t)0, dense);
    }
    for (int b = 0; b < 3; ++b) {
      // (the handle's constants: planes 1 and 2 also carry fc_1's bias of the block before, as the per-view kernel expects them)
      int* flag = flags + kSlotLinZ + b;
      if (sparse) {          // the touched rows: X = the gathered rows, Y = compact plane b; row count on the device
        Lin512Args a{Xc, wpack_slot(ws, w, kSlotLinZ + b, false), Yc + (size_t)b * cap * kHidden, mlp_hoist_bias(mlp) + kHidden * b,
                     nullptr, nullptr, cap, kLatent, kHidden, 0};
        a.m_dev = cnt;
        a.skip_silent = dense;
        Lin512Args h = a;
        h.Wp = wpack_slot(ws, w, kSlotLinZ + b, false, true);
        h.ovf = flag;
        h.skip = flags + kFlagWBad;
        if ((rc = lin512_launch(h, st, 1))) return rc;
        a.gate = flag;
        if ((rc = lin512_launch(a, st))) return rc;
      }
      // the whole map: always without the list, else only when the list overflowed (gated on `dense`)
      Lin512Args a{scene->latent_cl, wpack_slot(ws, w, kSlotLinZ + b, false), latent_proj_out + (size_t)b * rows * kHidden, mlp_hoist_bias(mlp) + kHidden * b,
                   nullptr, nullptr, rows, kLatent, kHidden, 0};
      Lin512Args h = a;
      h.Wp = wpack_slot(ws, w, kSlotLinZ + b, false, true);
      h.ovf = flag;
      h.skip = flags + kFlagWBad;
      h.gate = dense;
      if ((rc = lin512_launch(h, st, 1))) return rc;
      a.gate = flag;
      a.gate2 = dense;
      if ((rc = lin512_launch(a, st))) return rc;
    }
    if (sparse)
      hipLaunchKernelGGL(k_move_rows, dim3(1024, 3), dim3(256), 0, st, (const float*)Yc, latent_proj_out, idx, cnt, (int)cap, 1, (size_t)cap * kHidden,
                         (size_t)rows * kHidden, dense);
    own.latent_proj = latent_proj_out;
    own.proj_stamp = diner_mlp_stamp(mlp);
  }
  SaveActs sv;
  for (int b = 0; b < 5; ++b) { sv.X[b] = ws + w.X[b]; sv.H[b] = ws + w.H[b]; }
  sv.x_last = ws + w.x_last;
  sv.raw = ws + w.raw;
  for (int b = 0; b < 5; ++b) { sv.bX[b] = reinterpret_cast<unsigned*>(ws + w.bX[b]); sv.bH[b] = reinterpret_cast<unsigned*>(ws + w.bH[b]); }
  int* ovf = nullptr;
  // hand-over + tile counters of the two kernels: the backward's dx buffer is free in the forward (8 KB per point; 2 KB + flags needed)
  if ((rc = field_forward_save(&own, mlp, xyz, viewdirs, P, out, sc + w.dx, sv, &ovf, st))) return rc;
  hipLaunchKernelGGL(k_copy_flag, dim3(1), dim3(1), 0, st, ovf, reinterpret_cast<int*>(ws + w.flags) + kFlagFusedOvf);
  DINER_LAUNCH_OK();
  // the exact repeat, on the device: the layer-wise forward behind the flag (its ~25 launches return at once when it stayed down)
  return forward_layerwise(scene, p, xyz, viewdirs, P, out, ws, sc, w, stream, reinterpret_cast<const int*>(ws + w.flags) + kFlagFusedOvf, true, !gather_lat);
}

extern "C" int diner_field_train_forward_fused_f32(const DinerScene* scene, const DinerMlp* mlp, const DinerMlpParams* p, const float* xyz,
                                                   const float* viewdirs, long long P, float* out, void* workspace, void* scratch,
                                                   float* latent_proj_out, void* stream) {
  DINER_CHECK_ARG(scene && mlp && xyz && viewdirs && out && workspace && P > 0, "field_train_forward_fused: bad arguments");
  int rc = check_train_params(p, true);
  if (rc) return rc;
  DINER_CHECK_ARG(use_lin512() && (use_fwd_f16() || use_bwd_f16()), "field_train_forward_fused: needs the 512-layer kernels of the backward");
  float* ws = (float*)workspace;
  const TrainWs w = train_ws(P, scene->nv);
  return fused_forward_core(scene, mlp, p, xyz, viewdirs, P, out, ws, scratch_base(ws, scratch, w), w, latent_proj_out, (hipStream_t)stream, true);
}

// 1 when the fused forward that filled `workspace` met an activation beyond the fp16 range (its saved activations are not usable), else 0;
// waits for `stream` (one 4-byte read back)
extern "C" int diner_field_train_fused_overflowed(const void* workspace, long long P, int nv, int* overflowed, void* stream) {
  DINER_CHECK_ARG(workspace && overflowed && P > 0 && nv > 0, "field_train_fused_overflowed: bad arguments");
  const TrainWs w = train_ws(P, nv);
  DINER_HIP_OK(hipMemcpyAsync(overflowed, reinterpret_cast<const int*>((const float*)workspace + w.flags) + kFlagFusedOvf, sizeof(int),
                              hipMemcpyDeviceToHost, (hipStream_t)stream));
  DINER_HIP_OK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

// grads: the same structure as the parameters, device buffers of the parameters' shapes (overwritten);
// d_latent_cl (nv, Hf, Wf, 512) or NULL: overwritten with the gradient of the channels-last feature map
extern "C" int diner_field_train_backward_s_f32(const DinerScene* scene, const DinerMlpParams* p, const DinerMlpParams* grads, long long P,
                                                const float* d_out, void* workspace, void* scratch, float* d_latent_cl, void* stream);
extern "C" int diner_field_train_backward_f32(const DinerScene* scene, const DinerMlpParams* p, const DinerMlpParams* grads,
                                              long long P, const float* d_out, void* workspace, float* d_latent_cl,
                                              void* stream) {
  return diner_field_train_backward_s_f32(scene, p, grads, P, d_out, workspace, nullptr, d_latent_cl, stream);
}
// n_obj objects of P points each, object-major rows in every tensor of the workspace (w = train_ws(n_obj * P, nv): object o's rows of a
// per-view tensor are [o cols, (o + 1) cols), of a post-mean tensor [o P, (o + 1) P)): the layer products run ONCE over all rows; per object
// only the view-mean adjoint and the scatter of the latent gradient into that object's feature-map gradient d_latent_cl[o] (or null)
static int backward_core(const DinerScene* const* scenes, int n_obj, const DinerMlpParams* p, const DinerMlpParams* grads, long long P_obj,
                         const float* d_out, float* ws, float* sc, const TrainWs& w, float* const* d_latent_cl, hipStream_t st,
                         float* map_scratch = nullptr, bool lat_missing = false) {
  int rc = 0;
  const DinerScene* scene = scenes[0];
  const long long P = P_obj * n_obj;
  const long long cols = P * scene->nv, cols_obj = P_obj * scene->nv;
  float* dx = sc + w.dx;
  float* dH = sc + w.dH;
  // weight gradients of the 512 x 512 layers: every layer keeps its partial tiles in its own slot, one launch sums them all at the end
  WgReduceJobs jobs;
  int n_jobs = 0;
  const size_t part_floats = wgrad512_part_bytes() / sizeof(float);
  auto part = [&](int slot) { return sc + w.wgpart + (size_t)slot * part_floats; };
  auto job = [&]() -> WgReduceJob* {
    jobs.job[n_jobs] = WgReduceJob{nullptr, nullptr, nullptr, 0};
    return &jobs.job[n_jobs++];
  };
  auto wt = [&](const float*, int slot) -> const void* {      // W packed transposed by the forward call of the step (k_lin512), or null
    return use_lin512() ? wpack_slot(ws, w, slot, true) : nullptr;
  };
  // f16x3 backward (use_bwd_f16): every dy operand carries the maximum of its magnitudes in an amax slot, written by its producer
  int* flags = reinterpret_cast<int*>(ws + w.flags);
  unsigned* amax = reinterpret_cast<unsigned*>(flags) + kAmax0;
  const bool z_sep = use_lin512() && lin512_ok(ws + w.lat, kLatent, sc + w.d_lat, kHidden, nullptr, nullptr);   // as the forward decided
  const bool bwd16 = use_bwd_f16() && use_fwd_f16() && use_lin512() && use_wgrad512() && z_sep && (reinterpret_cast<size_t>(ws) & 15) == 0;
  if (bwd16) DINER_HIP_OK(hipMemsetAsync(amax, 0, 64 * sizeof(unsigned), st));
  int a_cur = 0, a_next = 1;                                  // slot of the current dx; next free slot
  BwdArith ar_store;
  auto arith = [&](int slot, int a_dy, int a_dx) -> const BwdArith* {
    if (!bwd16) return nullptr;
    ar_store = BwdArith{amax + a_dy, a_dx >= 0 ? amax + a_dx : nullptr, flags + kFlagWBad, flags + slot, wpack_slot(ws, w, slot, true, true)};
    return &ar_store;
  };
  if (use_lin_out() && (reinterpret_cast<size_t>(p->lin_out_w) & 15) == 0 && (reinterpret_cast<size_t>(d_out) & 15) == 0) {
    DINER_HIP_OK(hipMemsetAsync((void*)grads->lin_out_w, 0, (size_t)4 * kHidden * sizeof(float), st));
    DINER_HIP_OK(hipMemsetAsync((void*)grads->lin_out_b, 0, 4 * sizeof(float), st));
    hipLaunchKernelGGL(k_lin_out_bwd, dim3(grid1d(P, 4, 256)), dim3(256), 0, st, ws + w.x_last, ws + w.raw, d_out, p->lin_out_w, P, dx,
                       (float*)grads->lin_out_w, (float*)grads->lin_out_b, bwd16 ? amax + 0 : nullptr);
  } else {
    hipLaunchKernelGGL(k_field_act_bwd, dim3(grid1d(P)), dim3(256), 0, st, ws + w.raw, d_out, P, 4, sc + w.d_raw);
    if ((rc = linear_bwd(sc + w.d_raw, 4, ws + w.x_last, kHidden, true, p->lin_out_w, (float*)grads->lin_out_w,
                         (float*)grads->lin_out_b, P, 4, kHidden, dx, ws + w.x_last, false, st))) return rc;
    // (lin_out off its skinny kernel -- a weight or d_out that is not 16-byte aligned: the general path keeps no maximum of dx)
    if (bwd16) hipLaunchKernelGGL(k_absmax, dim3(grid1d(P * (long long)kHidden)), dim3(256), 0, st, dx, P * (long long)kHidden, amax + 0);
  }
  // round 5: the data gradients of the 512-kernels read the relu decisions as bits (64 B per row instead of 2 KB; DINER_TRAIN_MASKBITS=0: the
  // saved pre-activations themselves, A/B measurement)
  static const bool maskbits_on = [] { const char* e = getenv("DINER_TRAIN_MASKBITS"); return !(e && *e == '0'); }();
  auto bits = [&](size_t off) { return maskbits_on ? reinterpret_cast<const unsigned*>(ws + off) : nullptr; };
  // ---- round 6: the adjoint of the three lin_z terms in MAP space ------------------------------------------------------------------------
  // X_b = in_b + interp(lin_z[b](latent map)) (the forward's hoist), so with D_b = the gradient of X_b scattered through the bilinear taps into
  // map shape:  dWz_b = D_b^T L,  dbz_b = column sums of D_b,  d latent = sum_b D_b Wz_b  -- over the TEXEL rows the batch touches (2.4 % of a
  // 400 x 300 map: 5.5 k rows per object) instead of three data-gradient and three weight-gradient products over the 2.6 M per-view sample
  // rows (20 ms of the 120 ms step) + the gather of the interpolated latent.  Per block and object: zero the touched rows of one map-shaped
  // plane (map_scratch: the forward's projection buffer, free in the backward), scatter dx into it (the merged-tap scatter that served d_lat),
  // then per list segment: gather D and L rows, dWz += D^T L and dbz (general bf16x6 GEMM, split-K with atomics, contraction length on the
  // device), T = D Wz (k_lin512_rows), d_latent_cl[rows] += T.  The list (mark / compact) is rebuilt from the saved tap rows; segments of
  // `cap` rows (what the object's share of the d_lat region holds in three buffers) cover any count without a host decision.
  // DINER_TRAIN_LINZ_MAPSPACE=0: the round-5 products over the sample rows.
  const char* e_ms = getenv("DINER_TRAIN_LINZ_MAPSPACE");
  bool mapspace = map_scratch && !(e_ms && *e_ms == '0') && use_lin512() && cols_obj >= 4096;
  for (int o = 0; o < n_obj && mapspace; ++o)
    mapspace = scenes[o]->latent_cl && scenes[o]->C == kLatent && (long long)scenes[o]->nv * scenes[o]->Hf * scenes[o]->Wf < (1ll << 30) &&
               (!d_latent_cl || !d_latent_cl[o] || (reinterpret_cast<size_t>(d_latent_cl[o]) & 15) == 0);
  struct ObjList { int* idx; int* cnt; int* seg; int* perm; int* hist; float* Dc; float* Lc; float* Tc; long long rows, cap; int n_seg; };
  const char* e_sorted = getenv("DINER_TRAIN_SCATTER_SORTED");      // (read per call, as DINER_TRAIN_LINZ_MAPSPACE: the tests compare the routes in one process)
  const bool sort_cols = !(e_sorted && *e_sorted == '0');
  ObjList ol[64];
  if (mapspace) {
    for (int o = 0; o < n_obj; ++o) {
      ObjList& L = ol[o];
      L.rows = (long long)scenes[o]->nv * scenes[o]->Hf * scenes[o]->Wf;
      float* base = sc + w.d_lat + (size_t)o * cols_obj * kLatent;          // the object's share of the d_lat region (cols_obj x 512 floats), unused in this mode
      int* mark = reinterpret_cast<int*>(base);
      L.idx = mark + L.rows;
      L.cnt = L.idx + L.rows;                                                // [0] marked rows, [1] unused, [2 ..] segment counts
      L.seg = L.cnt + 2;
      L.hist = L.seg + 64;                                                   // [rows] histogram -> offsets, [2048] sums of the scan's workgroups
      L.perm = L.hist + L.rows + 2048;                                       // [cols_obj] the columns sorted by texel (sort_columns_by_texel)
      const long long ints = 3 * L.rows + cols_obj + 64 + 64 + 2048;
      L.cap = (cols_obj - (ints + 511) / 512 - 2) / 3;
      if (L.cap < 64) { mapspace = false; break; }
      L.n_seg = (int)((L.rows + L.cap - 1) / L.cap);
      if (L.n_seg > 60 || L.rows > 2048ll * kScanBlock) { mapspace = false; break; }
      L.Dc = base + ((ints + 511) / 512 + 1) * 512;
      L.Lc = L.Dc + (size_t)L.cap * kLatent;
      L.Tc = L.Lc + (size_t)L.cap * kLatent;
    }
  }
  if (mapspace) {
    for (int o = 0; o < n_obj; ++o) {
      ObjList& L = ol[o];
      int* mark = L.idx - L.rows;
      DINER_HIP_OK(hipMemsetAsync(mark, 0, (size_t)L.rows * sizeof(int), st));
      DINER_HIP_OK(hipMemsetAsync(L.cnt, 0, 128 * sizeof(int), st));
      hipLaunchKernelGGL(k_mark_rows, dim3(grid1d(cols_obj * 4)), dim3(256), 0, st, (const int*)(ws + w.tap_row) + (size_t)o * cols_obj * 4, cols_obj * 4, mark);
      hipLaunchKernelGGL(k_compact_rows, dim3(grid1d(L.rows)), dim3(256), 0, st, mark, (int)L.rows, (int)L.rows, L.idx, L.cnt, L.cnt + 1);
      hipLaunchKernelGGL(k_segment_counts, dim3(1), dim3(64), 0, st, L.cnt, (int)L.cap, L.n_seg, L.seg);
      if (sort_cols && (rc = sort_columns_by_texel((const int*)(ws + w.tap_row) + (size_t)o * cols_obj * 4, cols_obj, L.rows, L.hist, L.hist + L.rows, L.perm, st)))
        return rc;
      if (d_latent_cl && d_latent_cl[o]) DINER_HIP_OK(hipMemsetAsync(d_latent_cl[o], 0, (size_t)L.rows * kLatent * sizeof(float), st));
    }
    for (int b = 0; b < 3; ++b) {
      DINER_HIP_OK(hipMemsetAsync((void*)grads->lin_z_w[b], 0, (size_t)kHidden * kLatent * sizeof(float), st));
      DINER_HIP_OK(hipMemsetAsync((void*)grads->lin_z_b[b], 0, (size_t)kHidden * sizeof(float), st));
    }
    DINER_LAUNCH_OK();
  }
  if (!mapspace && lat_missing)      // the batched forward skipped the interpolated latent rows: the sample-space adjoint reads them
    for (int o = 0; o < n_obj; ++o)
      hipLaunchKernelGGL(k_gather_latent, dim3((unsigned)((cols_obj + 3) / 4)), dim3(256), 0, st, (const float*)scenes[o]->latent_cl,
                         (const int*)(ws + w.tap_row) + (size_t)o * cols_obj * 4, ws + w.tap_w + (size_t)o * cols_obj * 4, cols_obj,
                         ws + w.lat + (size_t)o * cols_obj * kLatent, (const int*)nullptr);
  auto linz_backward_mapspace = [&](int b, const float* dxb) -> int {
    for (int o = 0; o < n_obj; ++o) {
      ObjList& L = ol[o];
      float* D = map_scratch;                                                // one map-shaped plane (rows x 512), dense; only the touched rows are used
      hipLaunchKernelGGL(k_zero_rows, dim3(1024), dim3(256), 0, st, D, L.idx, L.cnt);
      int r = scatter_latent_launch(dxb + (size_t)o * cols_obj * kHidden, (const int*)(ws + w.tap_row) + (size_t)o * cols_obj * 4,
                                    ws + w.tap_w + (size_t)o * cols_obj * 4, cols_obj, D, st, sort_cols ? L.perm : nullptr);
      if (r) return r;
      for (int sgi = 0; sgi < L.n_seg; ++sgi) {
        const int* n_dev = L.seg + sgi;
        const int* ix = L.idx + (size_t)sgi * L.cap;
        hipLaunchKernelGGL(k_move_rows, dim3(1024, 1), dim3(256), 0, st, (const float*)D, L.Dc, ix, n_dev, (int)L.cap, 0, (size_t)0, (size_t)0, (const int*)nullptr);
        hipLaunchKernelGGL(k_move_rows, dim3(1024, 1), dim3(256), 0, st, (const float*)scenes[o]->latent_cl, L.Lc, ix, n_dev, (int)L.cap, 0, (size_t)0, (size_t)0,
                           (const int*)nullptr);
        // dWz_b (512 f x 512 k) += Dc^T Lc over the segment's rows, dbz_b += column sums of Dc: op(A) = Dc^T (kTA), contraction length on the device
        // (contraction chunks of `linz_chunk` rows: the list holds ~11 k of the segment's 200 k rows, so the chunk sets how many workgroups have
        // work -- 512 rows: 1.4 per CU, each waiting for its operand tiles alone, 0.24 ms per product)
        static const int linz_chunk = [] { const char* e = getenv("DINER_TRAIN_LINZ_CHUNK"); const int v = e ? atoi(e) : 256; return v < 16 ? 16 : v; }();
        long long split = (L.cap + linz_chunk - 1) / linz_chunk;
        split = split < 1 ? 1 : (split > 8192 ? 8192 : split);
        if ((r = gemm_launch(L.Dc, L.Lc, (float*)grads->lin_z_w[b], kHidden, kLatent, (int)L.cap, kHidden, kLatent, kLatent, kTA | kAtomic, nullptr, nullptr,
                             (int)split, st, nullptr, (float*)grads->lin_z_b[b], nullptr, n_dev))) return r;
        if (d_latent_cl && d_latent_cl[o]) {
          Lin512Args a{L.Dc, wpack_slot(ws, w, kSlotLinZ + b, true), L.Tc, nullptr, nullptr, nullptr, L.cap, kHidden, kLatent, 0};
          a.m_dev = n_dev;
          if ((r = lin512_launch(a, st))) return r;
          hipLaunchKernelGGL(k_move_rows, dim3(1024, 1), dim3(256), 0, st, (const float*)L.Tc, d_latent_cl[o], ix, n_dev, (int)L.cap, 2, (size_t)0, (size_t)0,
                             (const int*)nullptr);
        }
      }
    }
    DINER_LAUNCH_OK();
    return 0;
  };
  int a_mean = -1;                                            // slot of the view mean's upstream gradient (set at b == 3)
  for (int b = 4; b >= 0; --b) {
    const long long M = b < 3 ? cols : P;
    const float* X = ws + w.X[b];
    const float* H = ws + w.H[b];
    const int a_h = a_next++;                                 // dH = (dx W1) masked
    // round 6: behind the view mean (b == 2) dx is the same row for every view: the product once per point, the views differ in their mask only
    // (k_mask_views) -- a quarter of the rows of this data gradient; the weight gradient keeps all rows.  DINER_TRAIN_VIEW_SHARED=0: as every block
    const char* e_vs = getenv("DINER_TRAIN_VIEW_SHARED");
    const bool view_shared_on = !(e_vs && *e_vs == '0');
    const bool view_shared = b == 2 && view_shared_on && bwd16 && maskbits_on && scene->nv > 1 && P_obj >= 256 &&
                             lin512_ok(dx, kHidden, dH, kHidden, nullptr, nullptr);
    if (view_shared) {
      // the weight gradient: dW = sum over (v, p) of (g[p] / nv)^T relu(H[v][p]) = g^T S with S = mean_v relu(H[v]) -- one pass over H (HBM-bound)
      // and a product over P rows instead of P nv; db = column sums of g.  g (the view mean's upstream gradient) still sits in rows [0, P) of
      // the buffer dH names now, S goes to its rows [P, 2 P); both are overwritten by the data gradient below.  DINER_TRAIN_VIEW_SHARED=2: all rows
      if (a_mean >= 0 && !(e_vs && *e_vs == '2')) {
        float* S = dH + (size_t)P * kHidden;
        for (int o = 0; o < n_obj; ++o)
          hipLaunchKernelGGL(k_view_mean_relu, dim3(grid1d(P_obj * (kHidden / 4))), dim3(256), 0, st, H + (size_t)o * cols_obj * kHidden, scene->nv,
                             P_obj * (kHidden / 4), S + (size_t)o * P_obj * kHidden);
        if ((rc = linear_bwd(dH, kHidden, S, kHidden, false, p->fc1_w[b], (float*)grads->fc1_w[b], (float*)grads->fc1_b[b], P,
                             kHidden, kHidden, nullptr, nullptr, false, st, wt(p->fc1_w[b], kSlotFc1 + b), part(kSlotFc1 + b), job(),
                             arith(kSlotFc1 + b, a_mean, -1), nullptr))) return rc;
      } else
      if ((rc = linear_bwd(dx, kHidden, H, kHidden, true, p->fc1_w[b], (float*)grads->fc1_w[b], (float*)grads->fc1_b[b], M,
                           kHidden, kHidden, nullptr, nullptr, false, st, wt(p->fc1_w[b], kSlotFc1 + b), part(kSlotFc1 + b), job(),
                           arith(kSlotFc1 + b, a_cur, -1), nullptr))) return rc;
      const int a_t = a_next++;                               // the maximum of T (the product's scale bookkeeping; dH's own goes to a_h)
      for (int o = 0; o < n_obj; ++o) {
        const size_t off = (size_t)o * cols_obj * kHidden;
        Lin512Args dh{dx + off, wpack_slot(ws, w, kSlotFc1 + b, true, true), dH + off, nullptr, nullptr, nullptr, P_obj, kHidden, kHidden, 0};
        dh.amax_in = amax + a_cur;
        dh.amax_out = amax + a_t;
        dh.skip = flags + kFlagWBad;
        if ((rc = lin512_launch(dh, st, 1))) return rc;
        Lin512Args da{dx + off, wpack_slot(ws, w, kSlotFc1 + b, true), dH + off, nullptr, nullptr, nullptr, P_obj, kHidden, kHidden, 0};
        da.gate = flags + kFlagWBad;                          // the bf16x6 twin: exactly one of the two runs
        if ((rc = lin512_launch(da, st, 0))) return rc;
        hipLaunchKernelGGL(k_mask_views, dim3(grid1d(P_obj * (kHidden / 4))), dim3(256), 0, st, dH + off,
                           bits(w.bH[b]) + (size_t)o * cols_obj * 16, scene->nv, P_obj, amax + a_h);
      }
      DINER_LAUNCH_OK();
    } else
    if ((rc = linear_bwd(dx, kHidden, H, kHidden, true, p->fc1_w[b], (float*)grads->fc1_w[b], (float*)grads->fc1_b[b], M,
                         kHidden, kHidden, dH, H, false, st, wt(p->fc1_w[b], kSlotFc1 + b), part(kSlotFc1 + b), job(),
                         arith(kSlotFc1 + b, a_cur, a_h), bits(w.bH[b])))) return rc;
    const int a_x = a_next++;                                 // dx += (dH W0) masked
    if ((rc = linear_bwd(dH, kHidden, X, kHidden, true, p->fc0_w[b], (float*)grads->fc0_w[b], (float*)grads->fc0_b[b], M,
                         kHidden, kHidden, dx, X, true, st, wt(p->fc0_w[b], kSlotFc0 + b), part(kSlotFc0 + b), job(),
                         arith(kSlotFc0 + b, a_h, a_x), bits(w.bX[b])))) return rc;
    a_cur = a_x;
    if (b < 3 && mapspace) {
      if ((rc = linz_backward_mapspace(b, dx))) return rc;
    } else if (b < 3 && (rc = linear_bwd(dx, kHidden, ws + w.lat, kLatent, false, p->lin_z_w[b], (float*)grads->lin_z_w[b],
                                         (float*)grads->lin_z_b[b], M, kHidden, kLatent, sc + w.d_lat, nullptr, b < 2, st,
                                         wt(p->lin_z_w[b], kSlotLinZ + b), part(kSlotLinZ + b), job(), arith(kSlotLinZ + b, a_cur, -1)))) return rc;
    if (b == 3) {          // adjoint of the view mean: dH is free here
      const int a_b = a_next++;
      a_mean = a_cur;
      for (int o = 0; o < n_obj; ++o)
        hipLaunchKernelGGL(k_view_bcast, dim3(grid1d(P_obj * kHidden)), dim3(256), 0, st, dx + (size_t)o * P_obj * kHidden, scene->nv, P_obj * kHidden,
                           dH + (size_t)o * cols_obj * kHidden, bwd16 ? amax + a_cur : nullptr, bwd16 ? amax + a_b : nullptr);
      a_cur = a_b;
      float* t = dx; dx = dH; dH = t;
    }
  }
  // lin_in's weight / bias gradient: round 6 on a kernel of its own in the f16x3 arithmetic (dy scaled from its maximum; the encoded inputs are
  // O(1): no range flag needed) -- DINER_TRAIN_WGRAD_IN=0: the general bf16x6 product (A/B measurement)
  static const bool wgrad_in_on = [] { const char* e = getenv("DINER_TRAIN_WGRAD_IN"); return !(e && *e == '0'); }();
  if (bwd16 && wgrad_in_on && cols >= 4096 && (reinterpret_cast<size_t>(dx) & 15) == 0) {
    DINER_HIP_OK(hipMemsetAsync((void*)grads->lin_in_w, 0, (size_t)kHidden * kDIn * sizeof(float), st));
    DINER_HIP_OK(hipMemsetAsync((void*)grads->lin_in_b, 0, (size_t)kHidden * sizeof(float), st));
    if ((rc = wgrad_in_launch(dx, kHidden, ws + w.feat, kDInPad, kDIn, cols, (float*)grads->lin_in_w, (float*)grads->lin_in_b, amax + a_cur, st))) return rc;
  } else if ((rc = linear_bwd(dx, kHidden, ws + w.feat, kDInPad, false, p->lin_in_w, (float*)grads->lin_in_w,
                              (float*)grads->lin_in_b, cols, kHidden, kDIn, nullptr, nullptr, false, st))) return rc;
  {   // (a job the general kernel served -- fewer than 256 rows, odd strides -- stays empty: dropped)
    int n = 0;
    for (int i = 0; i < n_jobs; ++i)
      if (jobs.job[i].part) jobs.job[n++] = jobs.job[i];
    if ((rc = wgrad512_reduce_many(jobs, n, true, st))) return rc;
  }
  for (int o = 0; o < n_obj && !mapspace; ++o) {
    if (!d_latent_cl || !d_latent_cl[o]) continue;
    DINER_HIP_OK(hipMemsetAsync(d_latent_cl[o], 0, (size_t)scenes[o]->nv * scenes[o]->Hf * scenes[o]->Wf * kLatent * sizeof(float), st));
    if ((rc = scatter_latent_launch(sc + w.d_lat + (size_t)o * cols_obj * kLatent, (const int*)(ws + w.tap_row) + (size_t)o * cols_obj * 4,
                                    ws + w.tap_w + (size_t)o * cols_obj * 4, cols_obj, d_latent_cl[o], st))) return rc;
  }
  DINER_LAUNCH_OK();
  return 0;
}

extern "C" int diner_field_train_backward_s_f32(const DinerScene* scene, const DinerMlpParams* p, const DinerMlpParams* grads, long long P,
                                                const float* d_out, void* workspace, void* scratch, float* d_latent_cl, void* stream) {
  DINER_CHECK_ARG(scene && d_out && workspace && P > 0, "field_train_backward: bad arguments");
  int rc = check_train_params(p, false);
  if (rc) return rc;
  rc = check_train_params(grads, false);
  if (rc) return rc;
  float* ws = (float*)workspace;
  const TrainWs w = train_ws(P, scene->nv);
  return backward_core(&scene, 1, p, grads, P, d_out, ws, scratch_base(ws, scratch, w), w, &d_latent_cl, (hipStream_t)stream);
}

// ---- ABI v6: the SB objects of a training step as ONE call pair (VERDICT r5 #1c) ------------------------------------------------------
// The layers are scene-independent: only the inputs / gather (forward) and the view-mean adjoint / scatter (backward) are per object.  The
// forward runs object by object on the fused kernels (each object has its own maps) into one workspace with object-major rows; the backward's
// 13 x (data gradient, weight gradient) products then run once over n_obj x P x nv rows: n_obj times fewer launches, one partial-tile sum,
// the weight gradients of the step summed inside the kernels instead of by autograd.  Also re-packs the persistent handle `mlp` from `p`
// (diner_mlp_update, training subset) -- no host synchronisation anywhere in the step.
namespace {
TrainWs obj_view(const TrainWs& t, long long P, int nv, int o) {
  TrainWs v = t;
  const size_t cols = (size_t)P * nv, oc = (size_t)o * cols, op = (size_t)o * P;
  v.feat += oc * kDInPad;
  v.tap_row += oc * 4;
  v.tap_w += oc * 4;
  v.lat += oc * kLatent;
  for (int b = 0; b < 5; ++b) {
    const size_t m = b < 3 ? oc : op;
    v.X[b] += m * kHidden;
    v.H[b] += m * kHidden;
    v.bX[b] += m * 16;
    v.bH[b] += m * 16;
  }
  v.x_last += op * kHidden;
  v.raw += op * 4;
  v.d_raw += op * 4;
  v.dx += oc * kHidden;
  v.dH += oc * kHidden;
  v.d_lat += oc * kLatent;
  return v;
}
int check_batch(const DinerScene* const* scenes, int n_obj, long long P) {
  DINER_CHECK_ARG(scenes && n_obj > 0 && n_obj <= 64 && P > 0, "field_train_batch: bad arguments (1 <= n_obj <= 64)");
  for (int o = 0; o < n_obj; ++o)
    DINER_CHECK_ARG(scenes[o] && scenes[o]->nv == scenes[0]->nv, "field_train_batch: every object needs the same number of source views");
  return 0;
}
}  // namespace

extern "C" int diner_field_train_batch_workspace_split(long long P, int nv, int n_obj, size_t* saved_bytes, size_t* scratch_bytes) {
  DINER_CHECK_ARG(P > 0 && nv > 0 && n_obj > 0 && saved_bytes && scratch_bytes, "field_train_batch_workspace_split: bad arguments");
  return diner_field_train_workspace_split(P * n_obj, nv, saved_bytes, scratch_bytes);
}

extern "C" int diner_field_train_forward_batch_f32(const DinerScene* const* scenes, int n_obj, DinerMlp* mlp, const DinerMlpParams* p,
                                                   const float* xyz, const float* viewdirs, long long P, float* out, void* saved,
                                                   void* scratch, float* latent_proj_scratch, void* stream) {
  int rc = check_batch(scenes, n_obj, P);
  if (rc) return rc;
  DINER_CHECK_ARG(mlp && xyz && viewdirs && out && saved && scratch && latent_proj_scratch, "field_train_forward_batch: null pointer argument");
  if ((rc = check_train_params(p, true))) return rc;
  DINER_CHECK_ARG(use_lin512() && (use_fwd_f16() || use_bwd_f16()), "field_train_forward_batch: needs the 512-layer kernels of the backward");
  for (int o = 0; o < n_obj; ++o)
    if ((rc = field_forward_save_supported(scenes[o], mlp))) return rc;
  if ((rc = diner_mlp_update(mlp, p, DINER_MLP_UPDATE_TRAIN_ONLY, stream))) return rc;
  float* ws = (float*)saved;
  const int nv = scenes[0]->nv;
  const TrainWs wt = train_ws(P * n_obj, nv);
  for (int o = 0; o < n_obj; ++o) {
    const TrainWs w = obj_view(wt, P, nv, o);
    // (no gather of the interpolated latent rows: the batched backward's map-space lin_z adjoint does not read them, its sample-space
    // fall-back gathers them itself, the gated layer-wise repeat gathers behind its gate)
    if ((rc = fused_forward_core(scenes[o], mlp, p, xyz + (size_t)o * P * 3, viewdirs + (size_t)o * P * 3, P, out + (size_t)o * P * 4, ws,
                                 (float*)scratch, w, latent_proj_scratch, (hipStream_t)stream, o == 0, /*gather_lat=*/false))) return rc;
  }
  return 0;
}

extern "C" int diner_field_train_backward_batch_f32(const DinerScene* const* scenes, int n_obj, const DinerMlpParams* p, const DinerMlpParams* grads,
                                                    long long P, const float* d_out, void* saved, void* scratch, float* const* d_latent_cl,
                                                    float* map_scratch, void* stream) {
  int rc = check_batch(scenes, n_obj, P);
  if (rc) return rc;
  DINER_CHECK_ARG(d_out && saved && scratch, "field_train_backward_batch: null pointer argument");
  if ((rc = check_train_params(p, false))) return rc;
  if ((rc = check_train_params(grads, false))) return rc;
  const TrainWs wt = train_ws(P * n_obj, scenes[0]->nv);
  DINER_CHECK_ARG((reinterpret_cast<size_t>(map_scratch) & 15) == 0, "field_train_backward_batch: map_scratch must be 16-byte aligned");
  return backward_core(scenes, n_obj, p, grads, P, d_out, (float*)saved, (float*)scratch, wt, d_latent_cl, (hipStream_t)stream, map_scratch, /*lat_missing=*/true);
}

#ifdef DINER_L512_PROF
extern "C" int diner_debug_scatter_prof(unsigned long long* out8, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out8, HIP_SYMBOL(diner::train::g_scat_prof), 8 * sizeof(unsigned long long));
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, reset == 2 ? 1ull : 0ull};      // reset == 2: the following launches without their atomics
    hipMemcpyToSymbol(HIP_SYMBOL(diner::train::g_scat_prof), z, sizeof(z));
  }
  return 0;
}
#endif
