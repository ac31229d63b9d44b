// Alpha compositing along rays: one wavefront per ray, wave scan for the transmittance.
// Replaces the tail of NeRFRendererDGS.composite (reference nerf_renderer.py:299-301, :341-360).
#include "common.hpp"

namespace diner {

constexpr int kCompMaxPerLane = 4;   // K <= 256

__global__ __launch_bounds__(256) void k_composite(const float4* __restrict__ field, const float* __restrict__ z,
                                                   const float* __restrict__ rays, int NR, int K, int per_lane,
                                                   int white_bkgd, float* __restrict__ rgb_out,
                                                   float* __restrict__ depth_out, float* __restrict__ weights_out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + wave;
  if (ray >= NR) return;                      // no block-level sync below: waves are independent
  const float far = rays[(size_t)ray * 8 + 7];
  const float* zr = z + (size_t)ray * K;
  const float4* fr = field + (size_t)ray * K;

  float alpha[kCompMaxPerLane], zz[kCompMaxPerLane];
  float4 f[kCompMaxPerLane];
  float prod = 1.0f;
#pragma unroll
  for (int j = 0; j < kCompMaxPerLane; ++j) {
    const int k = lane * per_lane + j;
    alpha[j] = 0.0f; zz[j] = 0.0f; f[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < per_lane && k < K) {
      zz[j] = zr[k];
      const float znext = (k + 1 < K) ? zr[k + 1] : far;                    // :299-301
      const float delta = __fsub_rn(znext, zz[j]);
      f[j] = fr[k];
      const float sig = fmaxf(f[j].w, 0.0f);                                 // relu again (idempotent) :344
      alpha[j] = __fsub_rn(1.0f, expf(-__fmul_rn(delta, sig)));
      prod *= __fadd_rn(__fsub_rn(1.0f, alpha[j]), 1e-10f);                  // :348
    }
  }
  float incl = prod;                          // inclusive multiplicative scan over lanes
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const float up = __shfl_up(incl, o, kWave);
    if (lane >= o) incl *= up;
  }
  float T = __shfl_up(incl, 1, kWave);
  if (lane == 0) T = 1.0f;
  float r = 0.f, g = 0.f, b = 0.f, d = 0.f, wsum = 0.f;
#pragma unroll
  for (int j = 0; j < kCompMaxPerLane; ++j) {
    const int k = lane * per_lane + j;
    if (j < per_lane && k < K) {
      const float w = __fmul_rn(alpha[j], T);                                // :351
      T *= __fadd_rn(__fsub_rn(1.0f, alpha[j]), 1e-10f);
      r += w * f[j].x; g += w * f[j].y; b += w * f[j].z;                     // :355
      d += w * zz[j];                                                        // :356
      wsum += w;
      if (weights_out) weights_out[(size_t)ray * K + k] = w;
    }
  }
  r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); d = wave_sum(d); wsum = wave_sum(wsum);
  if (lane == 0) {
    if (white_bkgd) {                                                        // :357-360
      const float bg = 1.0f - wsum;
      r += bg; g += bg; b += bg;
    }
    rgb_out[(size_t)ray * 3 + 0] = r;
    rgb_out[(size_t)ray * 3 + 1] = g;
    rgb_out[(size_t)ray * 3 + 2] = b;
    depth_out[ray] = d;
  }
}

}  // namespace diner

using namespace diner;

extern "C" int diner_composite_f32(const float* field, const float* z, const float* rays, int NR, int K,
                                   int white_bkgd, float* rgb_out, float* depth_out, float* weights_out,
                                   void* stream) {
  DINER_CHECK_ARG(field && z && rays && rgb_out && depth_out, "composite: null pointer argument");
  DINER_CHECK_ARG(NR > 0 && K > 0 && K <= kWave * kCompMaxPerLane, "composite: bad sizes NR=%d K=%d (K <= %d)", NR, K,
                  kWave * kCompMaxPerLane);
  const int per_lane = (K + kWave - 1) / kWave;
  hipLaunchKernelGGL(k_composite, dim3((NR + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float4*)field, z, rays,
                     NR, K, per_lane, white_bkgd, rgb_out, depth_out, weights_out);
  DINER_LAUNCH_OK();
  return 0;
}
