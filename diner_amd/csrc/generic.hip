// Generic-shape slow path (round 5): ResnetFC.forward and PixelNeRF.forward for configurations OUTSIDE what the fused field kernels are built
// for (they take d_in 55 / d_latent 512 / d_hidden 512 / 5 blocks / combine 3 / NV 4: every shipped DINER config).  The reference's
// constructors accept anything (resnetfc.py:72-127: d_hidden default 128, any n_blocks / combine_layer, Softplus for beta > 0;
// pixelnerf.py:13-33: any poscode), so the drop-in modules must run them too: here the layers are chained on the general exact-fp32 MFMA
// GEMM of the training path (diner_gemm_f32, kExact) with a view-mean pass at the combine layer -- correct and unhurried (one GEMM launch
// per layer, activations through HBM), never a silent PyTorch fallback.
//   diner_mlp_generic_forward_f32    ResnetFC.forward  (resnetfc.py:129-159)
//   diner_field_inputs_generic_f32   the (NV, P, d_latent + d_in) matrix PixelNeRF.forward hands to the MLP  (pixelnerf.py:84-128)
#include "common.hpp"
#include "field_common.hpp"

extern "C" int diner_gemm_f32(const float* A, const float* B, float* C, long long M, int N, int K, int lda, int ldb, int ldc, int flags,
                              const float* bias, const float* mask, int k_split, void* stream);
extern "C" int diner_view_mean_f32(const float* x, int nv, long long PC, float* y, int adjoint, void* stream);

namespace diner {
namespace {
enum : int { kTB = 2, kReluA = 4, kAccum = 16, kExact = 64 };      // diner_gemm_f32 flags (include/diner_hip.h)

// torch.nn.Softplus(beta) (threshold 20): x for beta x > 20, else log1p(exp(beta x)) / beta
__global__ void k_softplus(const float* __restrict__ x, long long n, float beta, float* __restrict__ y) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
    const float v = x[i], bx = v * beta;
    y[i] = bx > 20.0f ? v : log1pf(expf(bx)) / beta;
  }
}
__global__ void k_fill_zero(float* __restrict__ y, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) y[i] = 0.0f;
}

// One wave per (view, point): the reference's per-view MLP input row [latent C ; poscode(x_c) ; R d ; poscode(depth - z_c)]
// (pixelnerf.py:91-128), any number of views <= 4, latent channels and encoding frequencies.  Geometry in the arithmetic of the fused
// kernels (common.hpp: the reference's rounding points), the encoding as k_posenc, the latent lookup as k_index_latent.
__global__ __launch_bounds__(256) void k_generic_inputs(SceneDev sc, FieldArgs fa, int F, int include_input, float* __restrict__ zx) {
  const long long gw = (blockIdx.x * 256ll + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (gw >= (long long)sc.nv * fa.P) return;
  const int v = (int)(gw / fa.P);
  const long long p = gw - (long long)v * fa.P;
  float px, py, pz, dx, dy, dz;
  load_point(fa, p, px, py, pz, dx, dy, dz);
  float xc[3], vd[3];
  world_to_cam(sc.R[v], sc.t[v], px, py, pz, xc[0], xc[1], xc[2]);
  vd[0] = rot_row(sc.R[v] + 0, dx, dy, dz);
  vd[1] = rot_row(sc.R[v] + 3, dx, dy, dz);
  vd[2] = rot_row(sc.R[v] + 6, dx, dy, dz);
  const float u = project_axis(xc[0], xc[2], sc.focal[v][0], sc.c[v][0], sc.img_w);
  const float w = project_axis(xc[1], xc[2], sc.focal[v][1], sc.c[v][1], sc.img_h);
  const int ix = nearest_border(u, sc.Ws), iy = nearest_border(w, sc.Hs);
  const float dd = __fsub_rn(sc.depth[(size_t)v * sc.Hs * sc.Ws + (size_t)iy * sc.Ws + ix], xc[2]);
  const int C = sc.C, per = 2 * F + (include_input ? 1 : 0), d_in = 3 * per + 3 + per;
  float* row = zx + (size_t)gw * (C + d_in);
  // ---- latent: bilinear / border on the padded map (image_encoder.py:112-123), as k_index_latent
  if (C > 0) {
    const int Wf = sc.Wf, Hf = sc.Hf;
    const float su = __fmul_rn(u, __fdiv_rn(__fsub_rn((float)Wf, __fmul_rn(sc.feature_padding, 2.0f)), (float)Wf));
    const float sv = __fmul_rn(w, __fdiv_rn(__fsub_rn((float)Hf, __fmul_rn(sc.feature_padding, 2.0f)), (float)Hf));
    const float fx = clip_border(unnormalize(su, Wf), Wf), fy = clip_border(unnormalize(sv, Hf), Hf);
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float wx = fx - x0f, wy = fy - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int x1 = min(x0 + 1, Wf - 1), y1 = min(y0 + 1, Hf - 1);
    const float w00 = (1.0f - wy) * (1.0f - wx), w01 = (1.0f - wy) * wx, w10 = wy * (1.0f - wx), w11 = wy * wx;
    const float* base = sc.latent_cl + (size_t)v * Hf * Wf * C;
    const float* p00 = base + ((size_t)y0 * Wf + x0) * C;
    const float* p01 = base + ((size_t)y0 * Wf + x1) * C;
    const float* p10 = base + ((size_t)y1 * Wf + x0) * C;
    const float* p11 = base + ((size_t)y1 * Wf + x1) * C;
    for (int ch = lane; ch < C; ch += kWave) row[ch] = p00[ch] * w00 + p01[ch] * w01 + p10[ch] * w10 + p11[ch] * w11;
  }
  // ---- the encoded inputs (positional_encoding.py:33-53: inputs first, then j-major / d-minor sin(fma(x_d, f_j, phase_j)))
  for (int o = lane; o < d_in; o += kWave) {
    float val;
    int D, oo = o;
    const float* src;
    float one[1] = {dd};
    if (o < 3 * per) { D = 3; src = xc; }
    else if (o < 3 * per + 3) { row[C + o] = vd[o - 3 * per]; continue; }
    else { D = 1; src = one; oo = o - 3 * per - 3; }
    if (include_input && oo < D) {
      val = src[oo];
    } else {
      if (include_input) oo -= D;
      const int j = oo / D, d = oo - j * D;
      const float freq = __fmul_rn(fa.freq_factor, (float)(1 << (j >> 1)));
      const float phase = (j & 1) ? 1.57079637050628662109375f : 0.0f;
      const float arg = __fmaf_rn(src[d], freq, phase);
      val = fabsf(arg) < 8192.0f ? sin_posenc(arg) : sinf(arg);
    }
    row[C + o] = val;
  }
}

int grid1d(long long n) {
  const long long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
size_t align64(size_t n) { return (n + 63) / 64 * 64; }
}  // namespace
}  // namespace diner

using namespace diner;

static int check_generic(const DinerMlpParams* p, int nv, long long B, const char* who) {
  DINER_CHECK_ARG(p && p->lin_out_w && p->lin_out_b && p->fc0_w && p->fc0_b && p->fc1_w && p->fc1_b, "%s: parameter pointers missing", who);
  DINER_CHECK_ARG(p->d_hidden > 0 && p->d_out > 0 && p->n_blocks >= 0 && p->d_in >= 0 && p->d_latent >= 0 && p->d_in + p->d_latent > 0,
                  "%s: bad sizes d_in=%d d_latent=%d d_hidden=%d d_out=%d n_blocks=%d", who, p->d_in, p->d_latent, p->d_hidden, p->d_out, p->n_blocks);
  DINER_CHECK_ARG(p->d_in == 0 || (p->lin_in_w && p->lin_in_b), "%s: lin_in parameters missing", who);
  DINER_CHECK_ARG(p->d_latent == 0 || p->combine_layer <= 0 || p->n_blocks == 0 || (p->lin_z_w && p->lin_z_b), "%s: lin_z parameters missing", who);
  DINER_CHECK_ARG(nv >= 1 && B > 0, "%s: nv >= 1 and B > 0 wanted, got %d, %lld", who, nv, B);
  return 0;
}

// workspace: three (nv B, d_hidden) activation buffers (residual stream, hidden / view mean, activated copy for Softplus)
extern "C" size_t diner_mlp_generic_workspace_bytes(const DinerMlpParams* p, int nv, long long B) {
  if (!p || nv < 1 || B <= 0 || p->d_hidden <= 0) return 0;
  return 3 * align64((size_t)nv * B * p->d_hidden) * sizeof(float);
}

// zx (nv, B, d_latent + d_in) row-major, latent first (resnetfc.py:140-142); out: (B, d_out) when the views are combined inside the
// network (combine_layer < n_blocks), else (nv, B, d_out) -- the reference then never averages (resnetfc.py:149-152)
extern "C" int diner_mlp_generic_forward_f32(const DinerMlpParams* p, float beta, const float* zx, int nv, long long B, float* out,
                                             void* workspace, void* stream) {
  int rc = check_generic(p, nv, B, "mlp_generic_forward");
  if (rc) return rc;
  DINER_CHECK_ARG(zx && out && workspace, "mlp_generic_forward: null pointer argument");
  DINER_CHECK_ARG(beta >= 0.0f, "mlp_generic_forward: beta must be >= 0 (0: ReLU)");
  hipStream_t st = (hipStream_t)stream;
  const int H = p->d_hidden, D = p->d_latent + p->d_in;
  const size_t buf = align64((size_t)nv * B * H);
  float* X = (float*)workspace;
  float* Y = X + buf;
  float* T = Y + buf;
  long long M = (long long)nv * B;
  const bool soft = beta > 0.0f;
  // act(x) as the A operand of a product: ReLU rides on the GEMM's operand staging, Softplus needs a pass of its own
  auto act_operand = [&](const float* x, long long rows, const float** a, int* flag) {
    if (soft) {
      hipLaunchKernelGGL(k_softplus, dim3(grid1d(rows * H)), dim3(256), 0, st, x, rows * H, beta, T);
      *a = T;
      *flag = 0;
    } else {
      *a = x;
      *flag = kReluA;
    }
  };
  if (p->d_in > 0) {
    if ((rc = diner_gemm_f32(zx + p->d_latent, p->lin_in_w, X, M, H, p->d_in, D, p->d_in, H, kTB | kExact, p->lin_in_b, nullptr, 1, stream))) return rc;
  } else {
    hipLaunchKernelGGL(k_fill_zero, dim3(grid1d(M * H)), dim3(256), 0, st, X, M * H);       // x = zeros(d_hidden), broadcast (resnetfc.py:146)
  }
  bool combined = false;
  for (int b = 0; b < p->n_blocks; ++b) {
    if (b == p->combine_layer) {                                  // mean over the views (resnetfc.py:9-14, :149-152)
      if ((rc = diner_view_mean_f32(X, nv, B * (long long)H, Y, 0, stream))) return rc;
      float* t = X; X = Y; Y = t;
      M = B;
      combined = true;
    }
    if (p->d_latent > 0 && b < p->combine_layer)                  // x = x + lin_z[b](z)
      if ((rc = diner_gemm_f32(zx, p->lin_z_w[b], X, M, H, p->d_latent, D, p->d_latent, H, kTB | kAccum | kExact, p->lin_z_b[b], nullptr, 1, stream))) return rc;
    const float* a;
    int fl;
    act_operand(X, M, &a, &fl);                                   // net = fc_0(act(x))
    if ((rc = diner_gemm_f32(a, p->fc0_w[b], Y, M, H, H, H, H, H, kTB | kExact | fl, p->fc0_b[b], nullptr, 1, stream))) return rc;
    act_operand(Y, M, &a, &fl);                                   // x = x + fc_1(act(net))   (size_in == size_out: no shortcut layer)
    if ((rc = diner_gemm_f32(a, p->fc1_w[b], X, M, H, H, H, H, H, kTB | kAccum | kExact | fl, p->fc1_b[b], nullptr, 1, stream))) return rc;
  }
  (void)combined;
  const float* a;
  int fl;
  act_operand(X, M, &a, &fl);                                     // out = lin_out(act(x))
  if ((rc = diner_gemm_f32(a, p->lin_out_w, out, M, p->d_out, H, H, H, p->d_out, kTB | kExact | fl, p->lin_out_b, nullptr, 1, stream))) return rc;
  DINER_LAUNCH_OK();
  return 0;
}

// The per-view MLP inputs of PixelNeRF.forward for any poscode / latent width: zx (nv, P, C + d_in), d_in = 4 (2 F + include_input) + 3.
// Point source: (rays, z) with K samples per ray, or explicit xyz / viewdirs (rays == NULL).
extern "C" int diner_field_inputs_generic_f32(const DinerScene* scene, const float* rays, const float* z, int K, const float* xyz,
                                              const float* viewdirs, long long P, int num_freqs, int include_input, float freq_factor,
                                              float* zx, void* stream) {
  DINER_CHECK_ARG(scene && zx && P > 0, "field_inputs_generic: bad arguments");
  DINER_CHECK_ARG((rays && z && K > 0 && !xyz) || (!rays && xyz && viewdirs), "field_inputs_generic: give (rays, z, K) or (xyz, viewdirs)");
  DINER_CHECK_ARG(num_freqs >= 0 && num_freqs <= 30 && (num_freqs > 0 || include_input), "field_inputs_generic: bad positional encoding");
  SceneDev sd;
  int rc = make_scene_dev(scene, &sd);
  if (rc) return rc;
  DINER_CHECK_ARG(scene->depth && (sd.C == 0 || scene->latent_cl), "field_inputs_generic: depth / latent maps missing");
  FieldArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.rays = rays;
  fa.z = z;
  fa.xyz = xyz;
  fa.viewdirs = viewdirs;
  fa.K = rays ? K : 1;
  fa.P = P;
  fa.freq_factor = freq_factor;
  const long long waves = P * sd.nv;
  hipLaunchKernelGGL(k_generic_inputs, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, sd, fa, num_freqs,
                     include_input ? 1 : 0, zx);
  DINER_LAUNCH_OK();
  return 0;
}
