// Small stand-alone kernels behind the reference's public helper methods:
//   PositionalEncoding.forward                      (positional_encoding.py:33-53)
//   SpatialEncoder.index / index_depth / index_depth_std / index_normal   (image_encoder.py:97-223)
// The fused renderer does not call these (it encodes / gathers in registers); they exist so that the
// drop-in modules keep their stage-level API on the GPU and so that each stage has its own parity test.
#include "common.hpp"

namespace diner {

// out[n][d]             = x[n][d]                               (include_input)
// out[n][off + j*D + d] = sin(phase_j + x[n][d] * freq_j), freq_j = factor*2^(j/2), phase_j = (j&1)*fp32(pi/2)
__global__ void k_posenc(const float* __restrict__ x, long long N, int D, int F, float factor, int include_input,
                         float* __restrict__ out) {
  const int dout = D * (2 * F + (include_input ? 1 : 0));
  const long long total = N * (long long)dout;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long n = idx / dout;
    int o = (int)(idx - n * dout);
    float v;
    if (include_input && o < D) {
      v = x[n * D + o];
    } else {
      if (include_input) o -= D;
      const int j = o / D, d = o - j * D;
      const float freq = __fmul_rn(factor, (float)(1 << (j >> 1)));          // positional_encoding.py:18
      const float phase = (j & 1) ? 1.57079637050628662109375f : 0.0f;        // fp32(pi/2) :30
      const float arg = __fmaf_rn(x[n * D + d], freq, phase);                  // addcmul (fused on the CPU) :46
      v = fabsf(arg) < 8192.0f ? sin_posenc(arg) : sinf(arg);
    }
    out[idx] = v;
  }
}

__global__ void k_index(SceneDev sc, int mode, const float* __restrict__ uv, long long N, float* __restrict__ out) {
  // one thread per (view, point) for the nearest-neighbour lookups; the latent map has its own kernel
  const long long total = (long long)sc.nv * N;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(idx / N);
    const long long n = idx - (long long)v * N;
    const float u = uv[idx * 2 + 0], w = uv[idx * 2 + 1];
    const int Ws = sc.Ws, Hs = sc.Hs;
    const size_t plane = (size_t)Hs * Ws;
    if (mode == 1) {
      const int ix = nearest_border(u, Ws), iy = nearest_border(w, Hs);
      out[(size_t)v * N + n] = sc.depth[v * plane + (size_t)iy * Ws + ix];
    } else if (mode == 2) {
      const float su = __fmul_rn(u, __fdiv_rn((float)Ws, (float)Ws + 2.0f * kStdPad));
      const float sv = __fmul_rn(w, __fdiv_rn((float)Hs, (float)Hs + 2.0f * kStdPad));
      const int jx = nearest_zeros(su, Ws + 2 * kStdPad), jy = nearest_zeros(sv, Hs + 2 * kStdPad);
      float sd = 0.0f;
      if (jx >= 0 && jy >= 0) {
        const int kx = jx < kStdPad ? kStdPad - jx : (jx > Ws + kStdPad - 1 ? jx - (Ws + kStdPad - 1) : 0);
        const int ky = jy < kStdPad ? kStdPad - jy : (jy > Hs + kStdPad - 1 ? jy - (Hs + kStdPad - 1) : 0);
        const int sx = min(max(jx - kStdPad, 0), Ws - 1), sy = min(max(jy - kStdPad, 0), Hs - 1);
        const int e = max(max(kx, ky) - 1, 0);
        sd = sc.depth_std[v * plane + (size_t)sy * Ws + sx];
        if (e > 0) sd = __fmul_rn(sd, sc.std_pad_scale[e]);
      }
      out[(size_t)v * N + n] = sd;
    } else {
      const int nx = nearest_zeros(u, Ws), ny = nearest_zeros(w, Hs);
      for (int ch = 0; ch < 3; ++ch) {
        float val = 0.0f;
        if (nx >= 0 && ny >= 0) val = sc.normals[((size_t)v * 3 + ch) * plane + (size_t)ny * Ws + nx];
        out[((size_t)v * 3 + ch) * N + n] = val;
      }
    }
  }
}

// bilinear / border on the channels-last feature map; one wave per (view, point), lanes over channels.
__global__ __launch_bounds__(256) void k_index_latent(SceneDev sc, const float* __restrict__ uv, long long N,
                                                      float* __restrict__ out) {
  const long long gw = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (gw >= (long long)sc.nv * N) return;
  const int v = (int)(gw / N);
  const long long n = gw - (long long)v * N;
  const int Wf = sc.Wf, Hf = sc.Hf, C = sc.C;
  // uv * (size - 2*feature_padding) / size       (image_encoder.py:113-114)
  const float su = __fmul_rn(uv[gw * 2 + 0],
                             __fdiv_rn(__fsub_rn((float)Wf, __fmul_rn(sc.feature_padding, 2.0f)), (float)Wf));
  const float sv = __fmul_rn(uv[gw * 2 + 1],
                             __fdiv_rn(__fsub_rn((float)Hf, __fmul_rn(sc.feature_padding, 2.0f)), (float)Hf));
  const float px = clip_border(unnormalize(su, Wf), Wf), py = clip_border(unnormalize(sv, Hf), Hf);
  const float x0f = floorf(px), y0f = floorf(py);
  const float wx = px - x0f, wy = py - y0f;
  const int x0 = (int)x0f, y0 = (int)y0f;
  const int x1 = min(x0 + 1, Wf - 1), y1 = min(y0 + 1, Hf - 1);
  const float w00 = (1.0f - wy) * (1.0f - wx), w01 = (1.0f - wy) * wx, w10 = wy * (1.0f - wx), w11 = wy * wx;
  const float* base = sc.latent_cl + (size_t)v * Hf * Wf * C;
  const float* p00 = base + ((size_t)y0 * Wf + x0) * C;
  const float* p01 = base + ((size_t)y0 * Wf + x1) * C;
  const float* p10 = base + ((size_t)y1 * Wf + x0) * C;
  const float* p11 = base + ((size_t)y1 * Wf + x1) * C;
  for (int ch = lane; ch < C; ch += kWave)
    out[((size_t)v * C + ch) * N + n] = p00[ch] * w00 + p01[ch] * w01 + p10[ch] * w10 + p11[ch] * w11;
}

}  // namespace diner

using namespace diner;

extern "C" int diner_posenc_f32(const float* x, long long N, int d_in, int num_freqs, float freq_factor,
                                int include_input, float* out, void* stream) {
  DINER_CHECK_ARG(x && out, "posenc: null pointer argument");
  DINER_CHECK_ARG(N >= 0 && d_in > 0 && num_freqs > 0 && num_freqs <= 30, "posenc: bad sizes");
  if (N == 0) return 0;
  const long long total = N * (long long)d_in * (2 * num_freqs + (include_input ? 1 : 0));
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(k_posenc, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, N, d_in, num_freqs, freq_factor,
                     include_input, out);
  DINER_LAUNCH_OK();
  return 0;
}

extern "C" int diner_index_f32(const DinerScene* scene, int mode, const float* uv, long long N, float* out,
                               void* stream) {
  DINER_CHECK_ARG(scene && uv && out, "index: null pointer argument");
  DINER_CHECK_ARG(mode >= 0 && mode <= 3, "index: mode %d outside [0,3]", mode);
  SceneDev sd;
  int rc = make_scene_dev(scene, &sd);
  if (rc) return rc;
  if (N == 0) return 0;
  if (mode == 0) {
    DINER_CHECK_ARG(scene->latent_cl && sd.C > 0 && sd.Hf > 0 && sd.Wf > 0, "index: latent map missing");
    const long long waves = (long long)sd.nv * N;
    hipLaunchKernelGGL(k_index_latent, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, sd, uv, N,
                       out);
  } else {
    DINER_CHECK_ARG(scene->depth && scene->depth_std && scene->normals && scene->std_pad_scale,
                    "index: depth/std/normal maps missing");
    const long long total = (long long)sd.nv * N;
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(k_index, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sd, mode, uv, N, out);
  }
  DINER_LAUNCH_OK();
  return 0;
}
