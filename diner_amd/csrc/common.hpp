// Shared host/device helpers for libdiner_hip.so (gfx950 only; compiled with -ffp-contract=off so that
// every fp32 operation below rounds exactly where the reference's torch ops round).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/diner_hip.h"

namespace diner {

// ---- error plumbing -------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define DINER_CHECK_ARG(cond, ...)                 \
  do {                                             \
    if (!(cond)) {                                 \
      ::diner::set_error(__VA_ARGS__);             \
      return DINER_E_INVALID;                      \
    }                                              \
  } while (0)
#define DINER_HIP_OK(expr)                                                                    \
  do {                                                                                        \
    hipError_t e__ = (expr);                                                                  \
    if (e__ != hipSuccess) {                                                                  \
      ::diner::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return DINER_E_HIP;                                                                     \
    }                                                                                         \
  } while (0)
#define DINER_LAUNCH_OK()                                                                     \
  do {                                                                                        \
    hipError_t e__ = hipGetLastError();                                                       \
    if (e__ != hipSuccess) {                                                                  \
      ::diner::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
      return DINER_E_HIP;                                                                     \
    }                                                                                         \
  } while (0)

int validate_scene(const DinerScene* s);
// ResnetFC / positional-encoding configuration the kernels are built for (host-only check, before any device work);
// poscode = false skips the encoding fields (gradient structs carry none)
int check_mlp_config(const DinerMlpParams* p, const char* who, bool poscode);

constexpr int kMaxViews = 4;      // NV of every shipped config (dtu.py:48, facescape.py:42, multiface.py:46)
constexpr int kStdPad = 100;      // image_encoder.py:190-191
constexpr int kWave = 64;

// Scene constants copied by value into kernel arguments (poses etc. are tiny; keeps them in SGPRs).
struct SceneDev {
  const float* latent_cl;
  const float* depth;
  const float* depth_std;
  const float* normals;
  const float* std_pad_scale;
  float R[kMaxViews][9];
  float t[kMaxViews][3];
  float focal[kMaxViews][2];
  float c[kMaxViews][2];
  float img_w, img_h, feature_padding;
  int nv, C, Hf, Wf, Hs, Ws;
};
// Builds SceneDev from the C-ABI struct (poses / focal / c are host arrays there): no device access, no sync.
int make_scene_dev(const DinerScene* s, SceneDev* out);

// ---- device geometry (bit-compatible with the torch CPU ops of the reference) ---------------------
#ifdef __HIPCC__
// x_c = R x + t : torch.matmul's k-loop is r0*x0, fma(r1,x1,.), fma(r2,x2,.) (probed on the oracle box),
// followed by a separate add of t (pixelnerf.py:92-93, nerf_renderer.py:100-101).
__device__ __forceinline__ float rot_row(const float* r, float x0, float x1, float x2) {
  return __fmaf_rn(r[2], x2, __fmaf_rn(r[1], x1, __fmul_rn(r[0], x0)));
}
__device__ __forceinline__ void world_to_cam(const float* R, const float* t, float x0, float x1, float x2,
                                             float& c0, float& c1, float& c2) {
  c0 = __fadd_rn(rot_row(R + 0, x0, x1, x2), t[0]);
  c1 = __fadd_rn(rot_row(R + 3, x0, x1, x2), t[1]);
  c2 = __fadd_rn(rot_row(R + 6, x0, x1, x2), t[2]);
}
// uv = ((xy / z) * focal + c) / image_shape * 2 - 1     (pixelnerf.py:105-108, nerf_renderer.py:107-110)
__device__ __forceinline__ float project_axis(float x, float z, float focal, float c, float size) {
  float u = __fdiv_rn(x, z);
  u = __fmul_rn(u, focal);
  u = __fadd_rn(u, c);
  u = __fdiv_rn(u, size);
  u = __fmul_rn(u, 2.0f);
  return __fsub_rn(u, 1.0f);
}
// ATen grid_sampler unnormalize, align_corners=False.  The CPU kernel the reference (and the oracle) runs is the vectorised one
// (GridSamplerKernel.cpp, ComputeLocation::unnormalize: (u + 1) * (S / 2) - 0.5), built with -mfma for the AVX2 / AVX-512 dispatch levels,
// where the compiler CONTRACTS the multiply and the subtraction into one fused multiply-add: probed on the pinning host (round 5,
// 200,000 random coordinates on a 576-texel row: the fused form reproduces F.grid_sample's bilinear output to 5e-7, the two-rounding
// form is one ulp of the pixel coordinate -- 6e-5 of a texel at 576 -- off on 0.1 % of them).  One rounding here as well.
__device__ __forceinline__ float unnormalize(float u, int size) {
  return fmaf(__fadd_rn(u, 1.0f), 0.5f * (float)size, -0.5f);
}
__device__ __forceinline__ float clip_border(float p, int size) {
  // min(max(p,0), size-1); a NaN coordinate maps to 0 here (the reference leaves it undefined)
  p = fmaxf(p, 0.0f);
  return fminf(p, (float)(size - 1));
}
// nearest, padding_mode="border"
__device__ __forceinline__ int nearest_border(float u, int size) {
  return (int)rintf(clip_border(unnormalize(u, size), size));
}
// nearest, padding_mode="zeros": returns -1 when out of range (NaN -> -1 as well)
__device__ __forceinline__ int nearest_zeros(float u, int size) {
  float r = rintf(unnormalize(u, size));
  return (r >= 0.0f && r <= (float)(size - 1)) ? (int)r : -1;
}

// sin(x) for |x| < ~2^13 (positional-encoding arguments are |x_c| * 6.28 * 32 + pi/2 < ~1e3): Cody-Waite reduction
// by pi/2 in three fp32 parts + degree-7/6 minimax polynomials on [-pi/4, pi/4].  Max error ~1 ulp over the range --
// the same class as ocml's sinf -- without its register-hungry, branchy Payne-Hanek slow path.
__device__ __forceinline__ float sin_posenc(float x) {
  const float kf = rintf(x * 0.636619747f);                 // round(x * 2/pi)
  const int k = (int)kf;
  float r = __fmaf_rn(kf, -1.57079601287841796875f, x);      // pi/2 = 1.5707960128784 + 3.1391647326e-7 - 5.3903025e-15 ...
  r = __fmaf_rn(kf, -3.1391647326017846353352069855e-7f, r);
  r = __fmaf_rn(kf, -5.3903025299577647655052761875e-15f, r);
  const float r2 = r * r;
  // sin(r) = r + r^3 * S(r^2), cos(r) = 1 - r^2/2 + r^4 * C(r^2)
  float sp = __fmaf_rn(r2, 2.6083159809786593541502952575684e-6f, -1.981069071916863322257995605469e-4f);
  sp = __fmaf_rn(sp, r2, 8.3330785855650901794433593750000e-3f);
  sp = __fmaf_rn(sp, r2, -1.6666659712791442871093750000000e-1f);
  const float sn = __fmaf_rn(r * r2, sp, r);
  float cp = __fmaf_rn(r2, 2.4433157471139430999755859375e-5f, -1.3887316454201936721801757812e-3f);
  cp = __fmaf_rn(cp, r2, 4.1666645556688308715820312500e-2f);
  cp = __fmaf_rn(cp, r2, -0.5f);
  const float cs = __fmaf_rn(cp, r2, 1.0f);
  const float v = (k & 1) ? cs : sn;
  return (k & 2) ? -v : v;
}

// wave-level helpers (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Philox4x32-10 counter-based generator (production noise when no explicit noise tensors are given)
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u32_to_unit(uint32_t x) {  // [0,1) with 24 random bits
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float rng_uniform(uint64_t seed, uint32_t stream_id, uint32_t a, uint32_t b) {
  uint32_t o[4];
  philox4x32_10(a, b, stream_id, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
  return u32_to_unit(o[0]);
}
__device__ __forceinline__ float rng_normal(uint64_t seed, uint32_t stream_id, uint32_t a, uint32_t b) {
  uint32_t o[4];
  philox4x32_10(a, b, stream_id, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
  const float u1 = ((float)(o[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
  const float u2 = u32_to_unit(o[1]);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}
#endif  // __HIPCC__

}  // namespace diner
