// Depth-guided ray sampler + stratified fill: one wavefront (64 lanes) per ray.
//
// Replaces NeRFRendererDGS.sample_coarse / sample_depthguided / fill_up_uniform_samples
// (reference nerf_renderer.py:39-63, :65-190, :367-397), weighted_mean_n_std (torch_helpers.py:215-223)
// and the three nearest-neighbour lookups SpatialEncoder.index_depth / index_depth_std / index_normal
// (image_encoder.py:148-223, torch_helpers.py:99-159).
//
// Per ray: 1000 (<=1024) stratified candidates, each projected into the NV source views (3 nearest taps
// per view from the small depth / std / normal maps, which stay L2 resident), erf surface likelihood,
// max over views; then inside the wave: exclusive transmittance product (lane-local product + wave
// scan), top-(K-G) selection by a 31-step radix select on the likelihood bits + ballot-free
// compaction, likelihood-weighted mean/std of the candidate depths, G gaussian samples, bitonic sort,
// stratified fill of the empty slots and the final sort.  All candidate state lives in LDS / registers;
// HBM sees 32 B in and 4K B out per ray (plus the optional explicit noise).
#include "common.hpp"

namespace diner {

constexpr int kMaxCand = 1024;
constexpr int kCandPerLane = kMaxCand / kWave;   // 16
constexpr int kMaxK = 256;
constexpr int kRaysPerBlock = 4;

struct SamplerArgs {
  const float* rays;
  const float* t_base;
  const float* noise_coarse;
  const float* noise_gauss;
  const float* noise_fill;
  float* z_out;
  float* z_unfilled;
  uint64_t seed;
  uint32_t ray_key0;       // index of rays[0] in the caller's ray list: the in-kernel noise of ray i is keyed by ray_key0 + i
  int NR, n_cand, K, G;
  float depth_diff_max;
};

// surface likelihood of one candidate in one source view (nerf_renderer.py:107-128)
__device__ __forceinline__ float view_likelihood(const SceneDev& sc, int v, float px, float py, float pz,
                                                 const float* dcam, float step_size, float ddmax) {
  float xc, yc, zc;
  world_to_cam(sc.R[v], sc.t[v], px, py, pz, xc, yc, zc);
  const float u = project_axis(xc, zc, sc.focal[v][0], sc.c[v][0], sc.img_w);
  const float w = project_axis(yc, zc, sc.focal[v][1], sc.c[v][1], sc.img_h);
  const int Ws = sc.Ws, Hs = sc.Hs;
  const size_t plane = (size_t)Hs * Ws;
  // --- depth_std: nearest on the 100px exponentially padded map, zeros outside (image_encoder.py:185-194)
  const float su = __fmul_rn(u, __fdiv_rn((float)Ws, (float)Ws + 2.0f * kStdPad));
  const float sv = __fmul_rn(w, __fdiv_rn((float)Hs, (float)Hs + 2.0f * kStdPad));
  const int jx = nearest_zeros(su, Ws + 2 * kStdPad);
  const int jy = nearest_zeros(sv, Hs + 2 * kStdPad);
  if (jx < 0 || jy < 0) return 0.0f;                 // std == 0 -> masked (nerf_renderer.py:123)
  const int kx = jx < kStdPad ? kStdPad - jx : (jx > Ws + kStdPad - 1 ? jx - (Ws + kStdPad - 1) : 0);
  const int ky = jy < kStdPad ? kStdPad - jy : (jy > Hs + kStdPad - 1 ? jy - (Hs + kStdPad - 1) : 0);
  const int sx = min(max(jx - kStdPad, 0), Ws - 1);
  const int sy = min(max(jy - kStdPad, 0), Hs - 1);
  const int e = max(max(kx, ky) - 1, 0);
  float sd = sc.depth_std[v * plane + (size_t)sy * Ws + sx];
  if (e > 0) sd = __fmul_rn(sd, sc.std_pad_scale[e]);
  if (sd == 0.0f) return 0.0f;
  // --- depth: nearest / border (image_encoder.py:157-167)
  const int ix = nearest_border(u, Ws), iy = nearest_border(w, Hs);
  const float d = sc.depth[v * plane + (size_t)iy * Ws + ix];
  if (!(fabsf(__fsub_rn(d, zc)) < ddmax)) return 0.0f;                       // :122
  // --- normal: nearest / zeros (image_encoder.py:210-220); dot with the ray direction in this camera
  const int nx = nearest_zeros(u, Ws), ny = nearest_zeros(w, Hs);
  if (nx >= 0 && ny >= 0) {
    const float* np_ = sc.normals + (size_t)v * 3 * plane + (size_t)ny * Ws + nx;
    const float cosd = __fadd_rn(__fadd_rn(__fmul_rn(dcam[0], np_[0]), __fmul_rn(dcam[1], np_[plane])),
                                 __fmul_rn(dcam[2], np_[2 * plane]));        // :119
    if (!(cosd <= 0.0f)) return 0.0f;                                        // :121
  }
  const float den = __fmul_rn(sd, 1.41421356237309515f);                     // sigma * np.sqrt(2)
  const float half = __fdiv_rn(step_size, 2.0f);
  const float a = __fdiv_rn(__fsub_rn(__fadd_rn(zc, half), d), den);
  const float b = __fdiv_rn(__fsub_rn(__fsub_rn(zc, half), d), den);
  const float L = fabsf(__fmul_rn(0.5f, __fsub_rn(erff(a), erff(b))));       // :125-128
  return (L == L) ? L : 0.0f;
}

// in-LDS bitonic sort (ascending) of n2 (power of two <= 256) floats by one wave; block-uniform control flow
__device__ __forceinline__ void bitonic_sort(float* s, int n2, int lane) {
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = lane; t < n2 / 2; t += kWave) {
        const int lo = ((t / j) * 2 * j) + (t % j);
        const int hi = lo + j;
        const bool up = ((lo & k) == 0);
        const float a = s[lo], b = s[hi];
        if ((a > b) == up) { s[lo] = b; s[hi] = a; }
      }
      __syncthreads();
    }
  }
}

// fill_up_uniform_samples on a K-slot LDS row (nerf_renderer.py:367-397); row padded with +inf to n2
__device__ __forceinline__ void fill_and_sort(float* s, int K, int n2, float near, float far, const float* noise_row,
                                              uint64_t seed, int ray, int lane) {
  bitonic_sort(s, n2, lane);                                                 // :377
  int m = 0;
  for (int j = lane; j < K; j += kWave) m += (s[j] == 0.0f);
  m = wave_sum_i(m);                                                          // :382
  if (m > 0) {
    const float step = __fdiv_rn(__fsub_rn(far, near), (float)m);             // :388
    for (int j = lane; j < K; j += kWave) {
      if (s[j] == 0.0f) {
        const float u = noise_row ? noise_row[j] : rng_uniform(seed, 2u, (uint32_t)ray, (uint32_t)j);      // (`ray` here: the noise key of the ray)
        float z = __fadd_rn(near, __fmul_rn((float)j, step));                 // :389
        z = __fadd_rn(z, __fmul_rn(u, step));                                 // :390
        s[j] = z;
      }
    }
  }
  __syncthreads();
  bitonic_sort(s, n2, lane);                                                 // :396
}

__global__ __launch_bounds__(kRaysPerBlock* kWave) void k_sample_depthguided(SceneDev sc, SamplerArgs a) {
  __shared__ float sL[kRaysPerBlock][kMaxCand];
  __shared__ float sZ[kRaysPerBlock][kMaxCand];
  __shared__ float sS[kRaysPerBlock][kMaxK];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray_raw = blockIdx.x * kRaysPerBlock + wave;
  const bool live = ray_raw < a.NR;
  const int ray = live ? ray_raw : a.NR - 1;       // dead waves shadow the last ray, control flow stays uniform
  float* L = sL[wave];
  float* Z = sZ[wave];
  float* S = sS[wave];

  const float* r = a.rays + (size_t)ray * 8;
  const float ox = r[0], oy = r[1], oz = r[2], dx = r[3], dy = r[4], dz = r[5], near = r[6], far = r[7];
  const int n_cand = a.n_cand;
  const float step_size = __fdiv_rn(__fsub_rn(far, near), (float)n_cand);       // nerf_renderer.py:95
  const float jitter = (float)(1.0 / (double)n_cand);                           // :53, applied in fp32 at :57

  float dcam[kMaxViews][3];
#pragma unroll
  for (int v = 0; v < kMaxViews; ++v) {                                         // :102-103
    dcam[v][0] = rot_row(sc.R[v] + 0, dx, dy, dz);
    dcam[v][1] = rot_row(sc.R[v] + 3, dx, dy, dz);
    dcam[v][2] = rot_row(sc.R[v] + 6, dx, dy, dz);
  }

  // ---- candidates: lane owns i = lane + 64 c (coalesced noise reads) -----------------------------
  for (int cidx = 0; cidx < kCandPerLane; ++cidx) {
    const int i = lane + kWave * cidx;
    float lk = 0.0f, z = 0.0f;
    if (i < n_cand) {
      const float un = a.noise_coarse ? a.noise_coarse[(size_t)ray * n_cand + i]
                                      : rng_uniform(a.seed, 0u, a.ray_key0 + (uint32_t)ray, (uint32_t)i);
      const float t = __fadd_rn(a.t_base[i], __fmul_rn(un, jitter));            // :57
      z = __fadd_rn(__fmul_rn(near, __fsub_rn(1.0f, t)), __fmul_rn(far, t));    // :60
      const float px = __fadd_rn(ox, __fmul_rn(z, dx));                         // :96
      const float py = __fadd_rn(oy, __fmul_rn(z, dy));
      const float pz = __fadd_rn(oz, __fmul_rn(z, dz));
      for (int v = 0; v < sc.nv; ++v)
        lk = fmaxf(lk, view_likelihood(sc, v, px, py, pz, dcam[v], step_size, a.depth_diff_max));   // :129
    }
    L[i] = lk;
    Z[i] = z;
  }
  __syncthreads();

  // ---- lane-contiguous view: i = 16 lane + k ------------------------------------------------------
  float lv[kCandPerLane], zv[kCandPerLane];
#pragma unroll
  for (int k = 0; k < kCandPerLane; ++k) {
    lv[k] = L[lane * kCandPerLane + k];
    zv[k] = Z[lane * kCandPerLane + k];
  }
  // exclusive transmittance product  O_i = L_i * prod_{j<i} (1 - L_j)            (:131-132)
  float run = 1.0f;
#pragma unroll
  for (int k = 0; k < kCandPerLane; ++k) run *= (1.0f - lv[k]);
  float incl = run;                          // inclusive wave scan of the per-lane products
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const float up = __shfl_up(incl, o, kWave);
    if (lane >= o) incl *= up;
  }
  float carry = __shfl_up(incl, 1, kWave);
  if (lane == 0) carry = 1.0f;
  float ov[kCandPerLane];
  float osum = 0.0f;
  int any_o = 0;
#pragma unroll
  for (int k = 0; k < kCandPerLane; ++k) {
    ov[k] = lv[k] * carry;
    carry *= (1.0f - lv[k]);
    osum += ov[k];
    any_o |= (ov[k] != 0.0f);
  }
  osum = wave_sum(osum);
  const bool has_surface = wave_sum_i(any_o) > 0;                                // :182
  // weighted mean / std of the candidate depths (torch_helpers.py:215-223)
  float mean = 0.0f, sd = 0.0f;
  if (has_surface) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < kCandPerLane; ++k) acc += zv[k] * (ov[k] / osum);
    mean = wave_sum(acc);
    acc = 0.0f;
#pragma unroll
    for (int k = 0; k < kCandPerLane; ++k) {
      const float dlt = zv[k] - mean;
      acc += (dlt * dlt) * (ov[k] / osum);
    }
    sd = sqrtf(wave_sum(acc));
  }

  // ---- top-(K-G) by likelihood: radix select on the (non-negative) float bit patterns (:172-178) ----
  const int K = a.K, G = a.G, want = K - G;
  uint32_t ub[kCandPerLane];
#pragma unroll
  for (int k = 0; k < kCandPerLane; ++k) ub[k] = __float_as_uint(lv[k]);
  uint32_t T = 0;
  if (want > 0) {
    for (int bit = 30; bit >= 0; --bit) {
      const uint32_t trial = T | (1u << bit);
      int cnt = 0;
#pragma unroll
      for (int k = 0; k < kCandPerLane; ++k) cnt += (ub[k] >= trial);
      if (wave_sum_i(cnt) >= want) T = trial;
    }
  }
  // candidates strictly above T are all taken; ties at T (only if T > 0) fill the remainder in index order
  int n_gt = 0, n_eq = 0;
#pragma unroll
  for (int k = 0; k < kCandPerLane; ++k) {
    n_gt += (ub[k] > T);
    n_eq += (ub[k] == T);
  }
  int pre_gt = n_gt, pre_eq = n_eq;          // inclusive scans over lanes
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const int g1 = __shfl_up(pre_gt, o, kWave), e1 = __shfl_up(pre_eq, o, kWave);
    if (lane >= o) { pre_gt += g1; pre_eq += e1; }
  }
  const int tot_gt = __shfl(pre_gt, kWave - 1, kWave);
  const int eq_take = (T > 0 && want > 0) ? max(want - tot_gt, 0) : 0;
  int off_gt = pre_gt - n_gt;
  int off_eq = pre_eq - n_eq;
  for (int j = lane; j < kMaxK; j += kWave) S[j] = (j < K) ? 0.0f : __builtin_inff();
  __syncthreads();
  if (want > 0) {
#pragma unroll
    for (int k = 0; k < kCandPerLane; ++k) {
      if (ub[k] > T) {
        S[off_gt++] = zv[k];
      } else if (ub[k] == T && T > 0) {
        if (off_eq < eq_take) S[tot_gt + off_eq] = zv[k];
        ++off_eq;
      }
    }
  }
  // gaussian samples into the LAST G slots of every ray (zeros when the ray sees no surface)   (:181-190)
  for (int g = lane; g < G; g += kWave) {
    float zg = 0.0f;
    if (has_surface) {
      const float n = a.noise_gauss ? a.noise_gauss[(size_t)ray * G + g]
                                    : rng_normal(a.seed, 1u, a.ray_key0 + (uint32_t)ray, (uint32_t)g);
      zg = __fadd_rn(__fmul_rn(n, sd), mean);                                    // :188
    }
    S[want + g] = zg;
  }
  __syncthreads();
  if (a.z_unfilled && live)
    for (int j = lane; j < K; j += kWave) a.z_unfilled[(size_t)ray * K + j] = S[j];

  int n2 = 2;
  while (n2 < K) n2 <<= 1;
  fill_and_sort(S, K, n2, near, far, a.noise_fill ? a.noise_fill + (size_t)ray * K : nullptr, a.seed, (int)(a.ray_key0 + (uint32_t)ray), lane);
  if (live)
    for (int j = lane; j < K; j += kWave) a.z_out[(size_t)ray * K + j] = S[j];
}

__global__ __launch_bounds__(kRaysPerBlock* kWave) void k_fill_uniform(const float* z_in, const float* rays, int NR, int K,
                                                                        const float* noise_fill, uint64_t seed,
                                                                        uint32_t ray_key0, float* z_out) {
  __shared__ float sS[kRaysPerBlock][kMaxK];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ray_raw = blockIdx.x * kRaysPerBlock + wave;
  const bool live = ray_raw < NR;
  const int ray = live ? ray_raw : NR - 1;
  float* S = sS[wave];
  for (int j = lane; j < kMaxK; j += kWave) S[j] = (j < K) ? z_in[(size_t)ray * K + j] : __builtin_inff();
  __syncthreads();
  int n2 = 2;
  while (n2 < K) n2 <<= 1;
  fill_and_sort(S, K, n2, rays[(size_t)ray * 8 + 6], rays[(size_t)ray * 8 + 7],
                noise_fill ? noise_fill + (size_t)ray * K : nullptr, seed, (int)(ray_key0 + (uint32_t)ray), lane);
  if (live)
    for (int j = lane; j < K; j += kWave) z_out[(size_t)ray * K + j] = S[j];
}

}  // namespace diner

using namespace diner;

extern "C" int diner_sample_depthguided_f32(const DinerScene* scene, const float* rays, int NR, int n_cand, int K,
                                            int G, float depth_diff_max, const float* t_base,
                                            const float* noise_coarse, const float* noise_gauss,
                                            const float* noise_fill, uint64_t seed, long long ray_index0, float* z_out,
                                            float* z_unfilled, void* stream) {
  DINER_CHECK_ARG(scene && rays && t_base && z_out, "sample_depthguided: null pointer argument");
  DINER_CHECK_ARG(NR > 0, "sample_depthguided: NR must be positive (got %d)", NR);
  DINER_CHECK_ARG(n_cand > 0 && n_cand <= kMaxCand, "sample_depthguided: n_cand=%d outside [1,%d]", n_cand, kMaxCand);
  DINER_CHECK_ARG(K > 0 && K <= kMaxK, "sample_depthguided: n_samples=%d outside [1,%d]", K, kMaxK);
  DINER_CHECK_ARG(G >= 0 && G <= K, "sample_depthguided: need 0 <= n_gaussian <= n_samples (got %d, %d)", G, K);
  DINER_CHECK_ARG(ray_index0 >= 0 && ray_index0 + NR <= (1ll << 32),
                  "sample_depthguided: ray_index0 = %lld outside [0, 2^32 - NR] (the noise key of a ray is a 32-bit index)", ray_index0);
  SceneDev sd;
  int rc = make_scene_dev(scene, &sd);
  if (rc) return rc;
  DINER_CHECK_ARG(scene->depth && scene->depth_std && scene->normals && scene->std_pad_scale,
                  "sample_depthguided: scene depth/std/normal maps missing");
  SamplerArgs a{rays, t_base, noise_coarse, noise_gauss, noise_fill, z_out, z_unfilled, seed, (uint32_t)ray_index0, NR, n_cand, K,
                G, depth_diff_max};
  const int blocks = (NR + kRaysPerBlock - 1) / kRaysPerBlock;
  hipLaunchKernelGGL(k_sample_depthguided, dim3(blocks), dim3(kRaysPerBlock * kWave), 0, (hipStream_t)stream, sd, a);
  DINER_LAUNCH_OK();
  return 0;
}

extern "C" int diner_fill_uniform_f32(const float* z_in, const float* rays, int NR, int K, const float* noise_fill,
                                      uint64_t seed, long long ray_index0, float* z_out, void* stream) {
  DINER_CHECK_ARG(z_in && rays && z_out, "fill_uniform: null pointer argument");
  DINER_CHECK_ARG(NR > 0 && K > 0 && K <= kMaxK, "fill_uniform: bad sizes NR=%d K=%d", NR, K);
  DINER_CHECK_ARG(ray_index0 >= 0 && ray_index0 + NR <= (1ll << 32),
                  "fill_uniform: ray_index0 = %lld outside [0, 2^32 - NR] (the noise key of a ray is a 32-bit index)", ray_index0);
  const int blocks = (NR + kRaysPerBlock - 1) / kRaysPerBlock;
  hipLaunchKernelGGL(k_fill_uniform, dim3(blocks), dim3(kRaysPerBlock * kWave), 0, (hipStream_t)stream, z_in, rays, NR,
                     K, noise_fill, seed, (uint32_t)ray_index0, z_out);
  DINER_LAUNCH_OK();
  return 0;
}
