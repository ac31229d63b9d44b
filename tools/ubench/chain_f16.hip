// Micro-benchmark (round 5): a stand-alone GEMM CHAIN in the plain-fp16 arithmetic of DINER_PRECISION_F16 -- three
// residual blocks x = x + W1 relu(W0 relu(x) + b0) + b1 of 512 x 512 layers, no gathers, no front end -- in the decomposition of
// k_field_pre_h3n (feature-sliced waves, weights global -> VGPR through a register ring, activations exchanged through LDS in B-operand
// form) and in the candidate shapes with MORE COLUMNS PER WEIGHT FRAGMENT.  Question: does a body with 128 columns per tile beat the 64-column
// body per GEMM and column by >= 1.25x when nothing else is in the way?
//
//   config   waves  features/wave  columns  residual stream x                      regs (acc)     waves/SIMD
//   A 4x64     4        128           64     second fp32 accumulator block           128 + 128         1        = the shipped kernel's shape
//   B 4x128    4        128          128     parked in global scratch as fp16        256               1
//   C 8x64     8         64           64     second fp32 accumulator block           64 + 64           2
//   D 8x128    8         64          128     parked in global scratch as fp16        128               2
//   E 4x96     4        128           96     fp16 in registers (96 VGPRs)            192               1
//
// All configs: barrier, publish relu(.) as fp16 B operands, barrier, GEMM (exposed publishes: the own-chunk trick of the shipped kernel is
// left out everywhere).  Per wave the shader clocks of the GEMMs, the publishes (incl. both barriers) and the residual handling are summed
// and reported per layer / per 64 columns.  Output of the first tile is checked between the configs (same logical weights).
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench/bin/chain_f16 tools/ubench/chain_f16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <utility>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char* lds_ptr;
typedef __attribute__((address_space(3))) h8* lds_h8;
typedef const __attribute__((address_space(1))) char* gptr;

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, ACC, 0, 0, 0)

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {
  unsigned d;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// an accumulator value for the vector ALU: an explicit v_accvgpr_read per value (a plain read lets the register allocator move whole accumulator
// tuples into VGPRs across the GEMMs and spill others to make room -- the shipped kernel's cvt4 does the same)
__device__ __forceinline__ float acc_read(const f32x4& x, int i) {
  int xi;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(xi) : "a"(x[i]));
  return __int_as_float(xi);
}
template <int RT, int NG>
__device__ __forceinline__ void pin_acc(f32x4 (&acc)[RT][NG]) {
#pragma unroll
  for (int mo = 0; mo < RT; ++mo)
#pragma unroll
    for (int g = 0; g < NG; ++g) asm volatile("" : "+a"(acc[mo][g]));
}

constexpr int kLayers = 6, kK = 512, kKT = 16;      // k32 blocks per contraction

// ---- layouts -------------------------------------------------------------------------------------------------------------------------
// weights of one layer, packed per wave in consumption order: [wave NW][t 16][mo RT][lane 64][8 halfs]; lane (i = lane & 15, kq = lane >> 4)
// holds W[feature 16 (RT wave + mo) + i][k], k slots j < 4: 32 t + 4 kq + j, j >= 4: 32 t + 16 + 4 kq + (j - 4)   (the order in which a lane's
// accumulator values become its B-operand slots, see publish)
// B buffer in LDS: [t 16][g NG][lane 64] h8; lane (n = lane & 15, q = lane >> 4): column 16 g + n, k slots as above

template <int NW, int RT, int NG, int XM, int R, int SV = 0, int SA = 0>
struct Cfg {
  static constexpr int sv = SV, sa = SA;      // synthetic side task: SV v_fma_mix_f32 per quarter-step; SA: read-add-write of one accumulator every 4th quarter-step
  static constexpr int nw = NW, rt = RT, ng = NG, xm = XM, ring = R;
  static constexpr int tb = (RT * 16) / 32;                 // k32 blocks a wave publishes (its own features)
  static constexpr int chunk_bytes = 4 * NG * 1024;         // four k32 blocks
  static constexpr size_t lds_bytes = (size_t)kKT * NG * 1024;
  static constexpr int halves = RT / 4;                     // half-steps per k32 block (4 row tiles each)
};

template <class C>
struct ARing {
  h8 a[C::ring][4];
  gptr abase;
  unsigned avoff;
  __device__ __forceinline__ void load1(h8 (&dst)[4], int i) {
    asm volatile("" : "+s"(abase));
    dst[i] = *(const __attribute__((address_space(1))) h8*)(abase + avoff + i * 1024);
    if (i == 3) abase += 4096;
  }
  __device__ __forceinline__ void start(const _Float16* layer, int wave, int lane) {
    constexpr int NH = kKT * C::halves;
    abase = (gptr)(reinterpret_cast<const char*>(layer) + (size_t)wave * kKT * C::rt * 1024);
    avoff = lane * 16;
    static_for<(C::ring - 1 < NH ? C::ring - 1 : NH)>([&](auto H) {
#pragma unroll
      for (int i = 0; i < 4; ++i) load1(a[decltype(H)::value], i);
    });
  }
};

template <class C>
__device__ __forceinline__ lds_h8 bfrag(lds_ptr cb, int tl, int g) { return (lds_h8)(cb + (tl * C::ng + g) * 1024); }

// acc[mo][g] += W[slice][k] . B[k][cols g]
template <class C>
__device__ __forceinline__ void gemm(const _Float16* layer, lds_ptr lbase, int wave, int lane, f32x4 (&acc)[C::rt][C::ng]) {
  constexpr int NG = C::ng, HV = C::halves, NH = kKT * HV, R = C::ring;
  ARing<C> ring;
  ring.start(layer, wave, lane);
  asm volatile("" : "+v"(lbase));
  float sx[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) sx[i] = 0.0f;
  float sc = __int_as_float(lane), sd = 0.0f;
  asm volatile("" : "+v"(sc), "+v"(sd));
  h8 bb[NG];
  lds_ptr cbp[4];
  cbp[0] = lbase;
  asm volatile("" : "+v"(cbp[0]));
#pragma unroll
  for (int g = 0; g < NG; ++g) bb[g] = *bfrag<C>(cbp[0], 0, g);
  static_for<NH * NG>([&](auto Q) {
    constexpr int qi = decltype(Q)::value;
    constexpr int h = qi / NG, g = qi % NG;
    constexpr int t = h / HV, half = h % HV;
    __builtin_amdgcn_sched_barrier(0);
    // weight fragments of half-step h + R - 1: four loads spread over the NG quarter-steps
    if constexpr (h + R - 1 < NH) {
      constexpr int per = (4 + NG - 1) / NG;               // loads per quarter-step (1 for NG >= 4)
#pragma unroll
      for (int i = g * per; i < (g + 1) * per && i < 4; ++i) ring.load1(ring.a[(h + R - 1) % R], i);
    }
    if constexpr (half == HV - 1 && g == 0 && (t & 3) == 3 && t + 1 < kKT) {
      cbp[(t + 1) >> 2] = lbase + ((t + 1) >> 2) * C::chunk_bytes;
      asm volatile("" : "+v"(cbp[(t + 1) >> 2]));
    }
    // B fragment of group g - 1 had its last use for block t in the previous quarter-step: re-read it for block t + 1
    if constexpr (half == HV - 1 && g > 0 && t + 1 < kKT) bb[g - 1] = *bfrag<C>(cbp[(t + 1) >> 2], (t + 1) & 3, g - 1);
    if constexpr (half == 0 && g == 0 && t > 0) bb[NG - 1] = *bfrag<C>(cbp[t >> 2], t & 3, NG - 1);
    if constexpr (C::sv > 0) {
#pragma unroll
      for (int j = 0; j < C::sv; ++j)
        asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(sx[(qi * C::sv + j) & 7]) : "v"(sc), "v"(sd));
    }
    if constexpr (C::sa > 0 && (qi & 3) == 3) {
      constexpr int am = (qi >> 2) % C::rt, ag = ((qi >> 2) / C::rt) % NG;
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = acc_read(acc[am][ag], i) + sx[i];
      acc[am][ag] = v;
      asm volatile("" : "+a"(acc[am][ag]));
    }
    h8 (&ac)[4] = ring.a[h % R];
    const h8 b0 = bb[g];
#pragma unroll
    for (int m = 0; m < 4; ++m) MFMA(acc[4 * half + m][g], ac[m], b0);
#pragma unroll
    for (int m = 0; m < 4; ++m) asm volatile("" : "+a"(acc[4 * half + m][g]));
  });
  if constexpr (C::sv > 0 || C::sa > 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(sx[i]));
  }
}

// relu(acc) -> fp16 B operands of this wave's own k32 blocks (tb blocks from t0 = tb wave)
template <class C>
__device__ __forceinline__ void publish(lds_ptr lbase, int wave, const f32x4 (&acc)[C::rt][C::ng]) {
  asm volatile("" : "+v"(lbase));
#pragma unroll
  for (int tl = 0; tl < C::tb; ++tl) {
    const int t = C::tb * wave + tl;
    lds_ptr cb = lbase + (t >> 2) * C::chunk_bytes;
    asm volatile("" : "+v"(cb));
#pragma unroll
    for (int g = 0; g < C::ng; ++g) {
      u32x4 h;
#pragma unroll
      for (int part = 0; part < 2; ++part) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __int_as_float(max(__float_as_int(acc_read(acc[2 * tl + part][g], i)), 0));
        h[2 * part] = cvt_pk_f16(v[0], v[1]);
        h[2 * part + 1] = cvt_pk_f16(v[2], v[3]);
      }
      // (t & 3 is not a compile-time constant for tb < 4: runtime offset inside the chunk)
      *(lds_h8)(cb + (((t & 3) * C::ng + g) * 1024)) = __builtin_bit_cast(h8, h);
    }
  }
}

struct Marks {
  unsigned long long gemm = 0, pub = 0, res = 0, t;
  __device__ __forceinline__ void start() { t = __builtin_readcyclecounter(); }
  __device__ __forceinline__ void lap(unsigned long long& into) {
    const unsigned long long n = __builtin_readcyclecounter();
    into += n - t;
    t = n;
  }
};

// x0: deterministic tile input; value of (tile, feature, column)
__device__ __forceinline__ float x_init(int lane_base, int f_off, int c_off) {
  const int s = (lane_base + f_off * 131 + c_off * 17) & 1023;
  return (float)s * (1.0f / 1024.0f) - 0.5f;
}

// scalar base + per-lane 32-bit offset + immediate addressing for a run of 1 KB wave accesses (no 64-bit address register per access:
// left to itself the compiler materialises one per load / store of the unrolled code, hoists them out of the tile loop and spills them)
struct Run1K {
  __attribute__((address_space(1))) char* sb;
  unsigned voff;
  int i = 0;
  __device__ __forceinline__ Run1K(void* base, int lane) : sb((__attribute__((address_space(1))) char*)base), voff(lane * 16) {}
  template <class T> __device__ __forceinline__ void store(const T& v) {
    asm volatile("" : "+s"(sb));
    *(__attribute__((address_space(1))) T*)(sb + voff + (i & 3) * 1024) = v;
    if ((++i & 3) == 0) sb += 4096;
  }
  template <class T> __device__ __forceinline__ T load() {
    asm volatile("" : "+s"(sb));
    const T v = *(const __attribute__((address_space(1))) T*)(sb + voff + (i & 3) * 1024);
    if ((++i & 3) == 0) sb += 4096;
    return v;
  }
};

template <class C>
__global__ __launch_bounds__(64 * C::nw, 1) void k_chain(const _Float16* __restrict__ W, const float* __restrict__ bias, int tiles,
                                                         float* __restrict__ out, h8* __restrict__ park, unsigned long long* __restrict__ clk) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int RT = C::rt, NG = C::ng, NW = C::nw;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), q = lane >> 4, n = lane & 15;
  lds_ptr lbase = (lds_ptr)lds + lane * 16;
  const size_t layer_halfs = (size_t)kK * kK;
  Marks mk;
  unsigned long long total0 = __builtin_readcyclecounter();
  // parked residual: [wg][wave][unit RT / 2 * NG][lane] h8 (rows 4q..4q+3 of row tiles 2 u, 2 u + 1)
  h8* mypark = park + ((size_t)(blockIdx.x * NW + wave) * (RT / 2 * NG)) * 64;      // (wave-uniform; the lane's 16 bytes through Run1K)
  const int lane_base = (4 * q * 131 + n * 17) & 1023;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    f32x4 acc[RT][NG];            // XM 0: the hidden block; XM 1 / 2: the one block
    f32x4 xs[C::xm == 0 ? RT : 1][C::xm == 0 ? NG : 1];
    h4 x16[C::xm == 2 ? RT : 1][C::xm == 2 ? NG : 1];
    // ---- tile input
    mk.start();
#pragma unroll
    for (int mo = 0; mo < RT; ++mo)
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = x_init(lane_base + tile * 7919 + wave * (16 * RT * 131), 16 * mo + r, 16 * g);
        if constexpr (C::xm == 0) { xs[mo][g] = v; asm volatile("" : "+a"(xs[mo][g])); } else { acc[mo][g] = v; asm volatile("" : "+a"(acc[mo][g])); }
      }
    auto store_x = [&]() {          // XM 1: park fp16(acc); XM 2: keep fp16(acc) in registers
      Run1K pk(mypark, lane);
#pragma unroll
      for (int mo = 0; mo < RT; mo += 2)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          u32x4 h;
          h[0] = cvt_pk_f16(acc_read(acc[mo][g], 0), acc_read(acc[mo][g], 1));
          h[1] = cvt_pk_f16(acc_read(acc[mo][g], 2), acc_read(acc[mo][g], 3));
          h[2] = cvt_pk_f16(acc_read(acc[mo + 1][g], 0), acc_read(acc[mo + 1][g], 1));
          h[3] = cvt_pk_f16(acc_read(acc[mo + 1][g], 2), acc_read(acc[mo + 1][g], 3));
          if constexpr (C::xm == 1) pk.store(__builtin_bit_cast(h8, h));
          if constexpr (C::xm == 2) {
            const h8 hh = __builtin_bit_cast(h8, h);
            x16[mo][g] = (h4){hh[0], hh[1], hh[2], hh[3]};
            x16[mo + 1][g] = (h4){hh[4], hh[5], hh[6], hh[7]};
          }
        }
    };
    if constexpr (C::xm != 0) store_x();
    mk.lap(mk.res);
    for (int b = 0; b < 3; ++b) {
      const _Float16* W0 = W + (size_t)(2 * b) * layer_halfs;
      const _Float16* W1 = W + (size_t)(2 * b + 1) * layer_halfs;
      const float* b0 = bias + (2 * b) * kK;
      const float* b1 = bias + (2 * b + 1) * kK;
      // ---- relu(x) -> LDS
      __syncthreads();
      if constexpr (C::xm == 0) publish<C>(lbase, wave, xs); else publish<C>(lbase, wave, acc);
      __syncthreads();
      mk.lap(mk.pub);
      // ---- fc_0: h = W0 relu(x) + b0
#pragma unroll
      for (int mo = 0; mo < RT; ++mo) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b0 + 16 * (RT * wave + mo) + 4 * q);
#pragma unroll
        for (int g = 0; g < NG; ++g) { acc[mo][g] = bv; asm volatile("" : "+a"(acc[mo][g])); }
      }
      mk.lap(mk.res);
      gemm<C>(W0, lbase, wave, lane, acc);
      mk.lap(mk.gemm);
      // ---- relu(h) -> LDS
      __syncthreads();
      publish<C>(lbase, wave, acc);
      __syncthreads();
      mk.lap(mk.pub);
      // ---- fc_1: x = x + W1 relu(h) + b1
      if constexpr (C::xm == 0) {
#pragma unroll
        for (int mo = 0; mo < RT; ++mo) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(b1 + 16 * (RT * wave + mo) + 4 * q);
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc_read(xs[mo][g], r) + bv[r];
            xs[mo][g] = v;
            asm volatile("" : "+a"(xs[mo][g]));
          }
        }
        mk.lap(mk.res);
        gemm<C>(W1, lbase, wave, lane, xs);
        mk.lap(mk.gemm);
      } else {
        // accumulators = x (fp16, parked or in registers) + b1
        Run1K pk(mypark, lane);
#pragma unroll
        for (int mo = 0; mo < RT; mo += 2) {
          const f32x4 bv0 = *reinterpret_cast<const f32x4*>(b1 + 16 * (RT * wave + mo) + 4 * q);
          const f32x4 bv1 = *reinterpret_cast<const f32x4*>(b1 + 16 * (RT * wave + mo + 1) + 4 * q);
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            h8 hh;
            if constexpr (C::xm == 1) hh = pk.load<h8>();
            else hh = (h8){x16[mo][g][0], x16[mo][g][1], x16[mo][g][2], x16[mo][g][3], x16[mo + 1][g][0], x16[mo + 1][g][1], x16[mo + 1][g][2], x16[mo + 1][g][3]};
            f32x4 v0, v1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v0[r] = (float)hh[r] + bv0[r];
              v1[r] = (float)hh[4 + r] + bv1[r];
            }
            acc[mo][g] = v0;
            acc[mo + 1][g] = v1;
            asm volatile("" : "+a"(acc[mo][g]), "+a"(acc[mo + 1][g]));
          }
        }
        mk.lap(mk.res);
        gemm<C>(W1, lbase, wave, lane, acc);
        mk.lap(mk.gemm);
        store_x();
        mk.lap(mk.res);
      }
    }
    // ---- tile output: the first tile in full (cross-config check; [wave][mo][g][lane] f32x4), a checksum otherwise
    float cs = 0.f;
    if (tile == 0) {
      Run1K o(out + (size_t)wave * RT * NG * 256, lane);
#pragma unroll
      for (int mo = 0; mo < RT; ++mo)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = C::xm == 0 ? acc_read(xs[mo][g], r) : acc_read(acc[mo][g], r);
          o.store(v);
        }
    } else {
#pragma unroll
      for (int mo = 0; mo < RT; ++mo)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) cs += C::xm == 0 ? acc_read(xs[mo][g], r) : acc_read(acc[mo][g], r);
      out[(size_t)512 * 128 + (size_t)tile * 64 * NW + threadIdx.x] = cs;
    }
    mk.lap(mk.res);
  }
  if (lane == 0) {
    unsigned long long* c = clk + (size_t)(blockIdx.x * NW + wave) * 4;
    c[0] = mk.gemm; c[1] = mk.pub; c[2] = mk.res; c[3] = __builtin_readcyclecounter() - total0;
  }
}

// ---- host ----------------------------------------------------------------------------------------------------------------------------
static std::vector<float> g_W;       // logical weights [layer][out][in]
template <class C>
std::vector<_Float16> pack_weights() {
  std::vector<_Float16> p((size_t)kLayers * kK * kK);
  for (int l = 0; l < kLayers; ++l)
    for (int w = 0; w < C::nw; ++w)
      for (int t = 0; t < kKT; ++t)
        for (int mo = 0; mo < C::rt; ++mo)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int i = lane & 15, kq = lane >> 4;
              const int f = 16 * (C::rt * w + mo) + i;
              const int k = 32 * t + (j < 4 ? 4 * kq + j : 16 + 4 * kq + (j - 4));
              p[(size_t)l * kK * kK + ((((size_t)w * kKT + t) * C::rt + mo) * 64 + lane) * 8 + j] = (_Float16)g_W[((size_t)l * kK + f) * kK + k];
            }
  return p;
}

template <class C>
void run(const char* name, int tiles_per_wg, std::vector<float>* first_out, double clock_ghz) {
  const int grid = 256, tiles = grid * tiles_per_wg;
  std::vector<_Float16> wp = pack_weights<C>();
  std::vector<float> bias((size_t)kLayers * kK);
  for (size_t i = 0; i < bias.size(); ++i) bias[i] = 0.01f * (float)((int)(i % 13) - 6);
  _Float16* dW; float *dB, *dOut; h8* dPark; unsigned long long* dClk;
  HIP_OK(hipMalloc(&dW, wp.size() * 2));
  HIP_OK(hipMalloc(&dB, bias.size() * 4));
  const size_t out_floats = (size_t)512 * 128 + (size_t)(tiles + 1) * 64 * C::nw;
  HIP_OK(hipMalloc(&dOut, out_floats * 4));
  HIP_OK(hipMemset(dOut, 0, out_floats * 4));
  HIP_OK(hipMalloc(&dPark, (size_t)grid * C::nw * (C::rt / 2 * C::ng) * 64 * 16));
  HIP_OK(hipMalloc(&dClk, (size_t)grid * C::nw * 4 * 8));
  HIP_OK(hipMemcpy(dW, wp.data(), wp.size() * 2, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dB, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipFuncSetAttribute((const void*)k_chain<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes));
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_chain<C>), dim3(grid), dim3(64 * C::nw), C::lds_bytes, 0, dW, dB, grid, dOut, dPark, dClk);      // warm-up
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_chain<C>), dim3(grid), dim3(64 * C::nw), C::lds_bytes, 0, dW, dB, tiles, dOut, dPark, dClk);
  HIP_OK(hipEventRecord(e1));
  HIP_OK(hipDeviceSynchronize());
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> clk((size_t)grid * C::nw * 4);
  HIP_OK(hipMemcpy(clk.data(), dClk, clk.size() * 8, hipMemcpyDeviceToHost));
  double s[4] = {0, 0, 0, 0};
  for (size_t i = 0; i < clk.size(); ++i) s[i & 3] += (double)clk[i];
  const double waves = (double)grid * C::nw, cols64 = C::ng / 4.0;
  const double per_gemm = s[0] / waves / tiles_per_wg / kLayers, per_pub = s[1] / waves / tiles_per_wg / kLayers,
               per_res = s[2] / waves / tiles_per_wg / kLayers, per_tile = s[3] / waves / tiles_per_wg;
  const double mfma_clk = (double)(512 / C::nw / 16) * C::ng * kKT * 16.0 * (C::nw / 4.0);      // MFMA clocks per layer per SIMD (waves per SIMD x their MFMAs)
  const double flops = (double)tiles * C::ng * 16 * kLayers * 2.0 * kK * kK;
  std::vector<float> raw((size_t)512 * 128), o((size_t)512 * 128, 0.f);
  HIP_OK(hipMemcpy(raw.data(), dOut, raw.size() * 4, hipMemcpyDeviceToHost));
  for (int w = 0; w < C::nw; ++w)
    for (int mo = 0; mo < C::rt; ++mo)
      for (int g = 0; g < C::ng; ++g)
        for (int lane = 0; lane < 64; ++lane)
          for (int r = 0; r < 4; ++r)
            o[(size_t)(16 * (C::rt * w + mo) + 4 * (lane >> 4) + r) * 128 + 16 * g + (lane & 15)] =
                raw[((((size_t)w * C::rt + mo) * C::ng + g) * 64 + lane) * 4 + r];
  double err = 0, ref = 0;
  if (first_out->empty()) *first_out = o;
  else
    for (int f = 0; f < 512; ++f)
      for (int c = 0; c < 64; ++c) {      // the first 64 columns exist in every config
        err = fmax(err, fabs((double)o[f * 128 + c] - (*first_out)[f * 128 + c]));
        ref = fmax(ref, fabs((double)(*first_out)[f * 128 + c]));
      }
  printf("%-8s waves %d x %3d features x %3d columns, ring %d: %8.3f ms  %7.1f TFLOP/s (%.3f of 2500)  | per layer and wave: gemm %7.0f clk "
         "(MFMA %6.0f per SIMD), publish+barriers %6.0f, residual/bias %6.0f; tile %8.0f clk | per 64 columns: gemm %7.0f, all %7.0f | "
         "check vs first config: max |d| %.3g of %.3g\n",
         name, C::nw, C::rt * 16, C::ng * 16, C::ring, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 2500.0, per_gemm, mfma_clk, per_pub, per_res,
         per_tile, per_gemm / cols64, per_tile / kLayers / cols64, err, ref);
  (void)clock_ghz;
  HIP_OK(hipFree(dW)); HIP_OK(hipFree(dB)); HIP_OK(hipFree(dOut)); HIP_OK(hipFree(dPark)); HIP_OK(hipFree(dClk));
}

int main(int argc, char** argv) {
  const int tiles_per_wg = argc > 1 ? atoi(argv[1]) : 24;
  g_W.resize((size_t)kLayers * kK * kK);
  unsigned s = 12345;
  for (size_t i = 0; i < g_W.size(); ++i) {
    s = s * 1664525u + 1013904223u;
    g_W[i] = ((float)(s >> 8) / 16777216.0f - 0.5f) * 0.08f;
  }
  std::vector<float> first;
#ifndef ONLY
  run<Cfg<4, 8, 4, 0, 4>>("A 4x64", 2 * tiles_per_wg, &first, 0);
  run<Cfg<4, 8, 4, 0, 3>>("A 4x64", 2 * tiles_per_wg, &first, 0);
  run<Cfg<4, 8, 8, 1, 3>>("B 4x128", tiles_per_wg, &first, 0);
  run<Cfg<4, 8, 8, 1, 4>>("B 4x128", tiles_per_wg, &first, 0);
  run<Cfg<8, 4, 4, 0, 3>>("C 8x64", 2 * tiles_per_wg, &first, 0);
  // the same with a synthetic side task per quarter-step (timing only: the outputs of the SA variants differ)
  { std::vector<float> dummy; run<Cfg<8, 4, 4, 0, 3, 2, 0>>("C +2mix", 2 * tiles_per_wg, &dummy, 0); }
  { std::vector<float> dummy; run<Cfg<8, 4, 4, 0, 3, 4, 0>>("C +4mix", 2 * tiles_per_wg, &dummy, 0); }
  { std::vector<float> dummy; run<Cfg<8, 4, 4, 0, 3, 4, 1>>("C +4mix+acc", 2 * tiles_per_wg, &dummy, 0); }
  { std::vector<float> dummy; run<Cfg<8, 4, 4, 0, 3, 0, 1>>("C +acc", 2 * tiles_per_wg, &dummy, 0); }
  { std::vector<float> dummy; run<Cfg<4, 8, 4, 0, 3, 4, 1>>("A +4mix+acc", 2 * tiles_per_wg, &dummy, 0); }
  run<Cfg<8, 4, 8, 1, 3>>("D 8x128", tiles_per_wg, &first, 0);
  run<Cfg<4, 8, 6, 2, 3>>("E 4x96", 2 * tiles_per_wg, &first, 0);
#else
  run<ONLY>("only", tiles_per_wg, &first, 0);
#endif
  return 0;
}
