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
extern "C" int diner_colsum_f32(const float* dY, long long M, int N, int ld, float* db, void* stream);

namespace diner {
namespace {
enum : int { kTA = 1, kTB = 2, kReluA = 4, kReluB = 8, kAccum = 16, kAtomic = 32, kExact = 64 };      // diner_gemm_f32 flags (include/diner_hip.h)

// adjoint of Softplus(beta): dx (+)= dy * sigmoid(beta x)
__global__ void k_softplus_bwd(const float* __restrict__ x, const float* __restrict__ dy, long long n, float beta, int accumulate, float* __restrict__ dx) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
    const float bx = x[i] * beta;
    const float s = bx > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-bx));
    const float v = dy[i] * s;
    dx[i] = accumulate ? dx[i] + v : v;
  }
}
// y (rows, ld) = 0 on the first `cols` columns
__global__ void k_zero_cols(float* __restrict__ y, long long rows, int cols, int ld) {
  const long long n = rows * cols;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) y[(i / cols) * ld + (i % cols)] = 0.0f;
}

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

// adjoint of the latent part of k_generic_inputs: d_latent_cl[v][tap_k][ch] += w_k d_zx[v][p][ch] (float atomics; the geometry is recomputed)
__global__ __launch_bounds__(256) void k_generic_latent_bwd(SceneDev sc, FieldArgs fa, int d_row, const float* __restrict__ d_zx,
                                                            float* __restrict__ d_latent_cl) {
  const long long gw = (blockIdx.x * 256ll + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (gw >= (long long)sc.nv * fa.P) return;
  const int v = (int)(gw / fa.P);
  const long long p = gw - (long long)v * fa.P;
  float px, py, pz, dx, dy, dz;
  load_point(fa, p, px, py, pz, dx, dy, dz);
  float xc[3];
  world_to_cam(sc.R[v], sc.t[v], px, py, pz, xc[0], xc[1], xc[2]);
  const float u = project_axis(xc[0], xc[2], sc.focal[v][0], sc.c[v][0], sc.img_w);
  const float w = project_axis(xc[1], xc[2], sc.focal[v][1], sc.c[v][1], sc.img_h);
  const int C = sc.C, Wf = sc.Wf, Hf = sc.Hf;
  const float su = __fmul_rn(u, __fdiv_rn(__fsub_rn((float)Wf, __fmul_rn(sc.feature_padding, 2.0f)), (float)Wf));
  const float sv = __fmul_rn(w, __fdiv_rn(__fsub_rn((float)Hf, __fmul_rn(sc.feature_padding, 2.0f)), (float)Hf));
  const float fx = clip_border(unnormalize(su, Wf), Wf), fy = clip_border(unnormalize(sv, Hf), Hf);
  const float x0f = floorf(fx), y0f = floorf(fy);
  const float wx = fx - x0f, wy = fy - y0f;
  const int x0 = (int)x0f, y0 = (int)y0f;
  const int x1 = min(x0 + 1, Wf - 1), y1 = min(y0 + 1, Hf - 1);
  const float w00 = (1.0f - wy) * (1.0f - wx), w01 = (1.0f - wy) * wx, w10 = wy * (1.0f - wx), w11 = wy * wx;
  float* base = d_latent_cl + (size_t)v * Hf * Wf * C;
  float* p00 = base + ((size_t)y0 * Wf + x0) * C;
  float* p01 = base + ((size_t)y0 * Wf + x1) * C;
  float* p10 = base + ((size_t)y1 * Wf + x0) * C;
  float* p11 = base + ((size_t)y1 * Wf + x1) * C;
  const float* row = d_zx + (size_t)gw * d_row;
  for (int ch = lane; ch < C; ch += kWave) {
    const float gch = row[ch];
    atomicAdd(p00 + ch, gch * w00);
    atomicAdd(p01 + ch, gch * w01);
    atomicAdd(p10 + ch, gch * w10);
    atomicAdd(p11 + ch, gch * w11);
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


// ---- ABI v6: training through the generic path (resnetfc.py:72-159 under autograd, e.g. ResnetFC's default d_hidden = 128) ----
// The forward keeps what the backward needs -- per block b the stream entering fc_0 (X_b: after the view mean / the lin_z term) and fc_0's
// output H_b, and the stream entering lin_out -- and the backward chains the adjoints on the same exact-fp32 GEMM: data gradients
// (dy W, masked by the saved pre-activation's sign for ReLU, times sigmoid(beta x) for Softplus), weight gradients (dy^T act(x), split-K with
// atomics), bias gradients (column sums), the adjoint of the view mean, and the gradient with respect to zx (latent part through lin_z,
// encoded inputs through lin_in).  Correct and unhurried, like the forward.
namespace diner {
namespace {
struct GenLayout {
  size_t X[64], H[64], x_last, t1, t2, t3, total;      // float offsets: saved X_b / H_b / x_last, then three (nv B, H) temporaries
  long long rows[64];                                   // rows of block b's tensors (nv B before the combine layer, B from it on)
};
GenLayout gen_layout(const DinerMlpParams* p, int nv, long long B) {
  GenLayout L;
  size_t o = 0;
  const int Hd = p->d_hidden;
  long long M = (long long)nv * B;
  for (int b = 0; b < p->n_blocks && b < 64; ++b) {
    if (b == p->combine_layer) M = B;
    L.rows[b] = M;
    L.X[b] = o; o += align64((size_t)M * Hd);
    L.H[b] = o; o += align64((size_t)M * Hd);
  }
  L.x_last = o; o += align64((size_t)M * Hd);
  const size_t big = align64((size_t)nv * B * Hd);
  L.t1 = o; o += big;
  L.t2 = o; o += big;
  L.t3 = o; o += big;
  L.total = o;
  return L;
}
}  // namespace
}  // namespace diner

extern "C" size_t diner_mlp_generic_train_workspace_bytes(const DinerMlpParams* p, int nv, long long B) {
  if (!p || nv < 1 || B <= 0 || p->d_hidden <= 0 || p->n_blocks < 0 || p->n_blocks > 64) return 0;
  return gen_layout(p, nv, B).total * sizeof(float);
}

extern "C" int diner_mlp_generic_train_forward_f32(const DinerMlpParams* p, float beta, const float* zx, int nv, long long B, float* out,
                                                   void* workspace, void* stream) {
  int rc = check_generic(p, nv, B, "mlp_generic_train_forward");
  if (rc) return rc;
  DINER_CHECK_ARG(zx && out && workspace && beta >= 0.0f && p->n_blocks <= 64, "mlp_generic_train_forward: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int H = p->d_hidden, D = p->d_latent + p->d_in;
  const GenLayout L = gen_layout(p, nv, B);
  float* ws = (float*)workspace;
  float* T = ws + L.t1;
  const bool soft = beta > 0.0f;
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
  long long M = (long long)nv * B;
  // the stream entering block 0 (or lin_out when there are no blocks)
  float* cur = p->n_blocks > 0 ? (p->combine_layer == 0 ? ws + L.t2 : ws + L.X[0]) : ws + L.x_last;
  if (p->d_in > 0) {
    if ((rc = diner_gemm_f32(zx + p->d_latent, p->lin_in_w, cur, M, H, p->d_in, D, p->d_in, H, kTB | kExact, p->lin_in_b, nullptr, 1, stream))) return rc;
  } else {
    hipLaunchKernelGGL(k_fill_zero, dim3(grid1d(M * H)), dim3(256), 0, st, cur, M * H);
  }
  for (int b = 0; b < p->n_blocks; ++b) {
    float* X = ws + L.X[b];
    if (b == p->combine_layer) {                                  // `cur` is a temporary with nv B rows: its mean is X_b
      if ((rc = diner_view_mean_f32(cur, nv, B * (long long)H, X, 0, stream))) return rc;
      M = B;
    }
    if (p->d_latent > 0 && b < p->combine_layer)
      if ((rc = diner_gemm_f32(zx, p->lin_z_w[b], X, M, H, p->d_latent, D, p->d_latent, H, kTB | kAccum | kExact, p->lin_z_b[b], nullptr, 1, stream))) return rc;
    const float* a;
    int fl;
    act_operand(X, M, &a, &fl);
    if ((rc = diner_gemm_f32(a, p->fc0_w[b], ws + L.H[b], M, H, H, H, H, H, kTB | kExact | fl, p->fc0_b[b], nullptr, 1, stream))) return rc;
    // next stream = X_b + fc_1(act(H_b)): into the next block's slot (a temporary when that block starts with the view mean)
    float* nx = b + 1 == p->n_blocks ? ws + L.x_last : (b + 1 == p->combine_layer ? ws + L.t2 : ws + L.X[b + 1]);
    DINER_HIP_OK(hipMemcpyAsync(nx, X, (size_t)M * H * sizeof(float), hipMemcpyDeviceToDevice, st));
    act_operand(ws + L.H[b], M, &a, &fl);
    if ((rc = diner_gemm_f32(a, p->fc1_w[b], nx, M, H, H, H, H, H, kTB | kAccum | kExact | fl, p->fc1_b[b], nullptr, 1, stream))) return rc;
    cur = nx;
  }
  const float* a;
  int fl;
  act_operand(ws + L.x_last, M, &a, &fl);
  if ((rc = diner_gemm_f32(a, p->lin_out_w, out, M, p->d_out, H, H, H, p->d_out, kTB | kExact | fl, p->lin_out_b, nullptr, 1, stream))) return rc;
  DINER_LAUNCH_OK();
  return 0;
}

// grads: device buffers of the parameters' shapes (overwritten); d_zx (nv, B, d_latent + d_in) or NULL (overwritten); d_out as `out` of the forward
extern "C" int diner_mlp_generic_backward_f32(const DinerMlpParams* p, const DinerMlpParams* g, float beta, const float* zx, int nv, long long B,
                                              const float* d_out, void* workspace, float* d_zx, void* stream) {
  int rc = check_generic(p, nv, B, "mlp_generic_backward");
  if (rc) return rc;
  if ((rc = check_generic(g, nv, B, "mlp_generic_backward (grads)"))) return rc;
  DINER_CHECK_ARG(zx && d_out && workspace && beta >= 0.0f && p->n_blocks <= 64, "mlp_generic_backward: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int H = p->d_hidden, D = p->d_latent + p->d_in, NO = p->d_out;
  const GenLayout L = gen_layout(p, nv, B);
  float* ws = (float*)workspace;
  float* T = ws + L.t1;          // activated operand (Softplus) / scratch product
  float* G = ws + L.t2;          // gradient of the stream
  float* GH = ws + L.t3;         // gradient of the hidden activation
  const bool soft = beta > 0.0f;
  auto zero = [&](const float* ptr, size_t n) { return hipMemsetAsync((void*)ptr, 0, n * sizeof(float), st); };
  // dW (N, K) = dy^T act(x), db = column sums of dy: split-K over the rows with atomics into zeroed buffers
  auto wgrad = [&](const float* dy, int ldy, int N, const float* x, int ldx, int K, bool act, long long M, const float* dW, const float* db) -> int {
    DINER_HIP_OK(zero(dW, (size_t)N * K));
    DINER_HIP_OK(zero(db, (size_t)N));
    const float* xa = x;
    int fl = 0;
    if (act) {
      if (soft) {
        hipLaunchKernelGGL(k_softplus, dim3(grid1d(M * K)), dim3(256), 0, st, x, M * K, beta, T);      // (act operands are (M, H) contiguous)
        xa = T;
      } else {
        fl = kReluB;
      }
    }
    long long split = M / 640;
    split = split < 1 ? 1 : (split > 64 ? 64 : split);
    int r = diner_gemm_f32(dy, xa, (float*)dW, N, K, M, ldy, ldx, K, kTA | kAtomic | kExact | fl, nullptr, nullptr, (int)split, stream);
    if (r) return r;
    return diner_colsum_f32(dy, M, N, ldy, (float*)db, stream);
  };
  // dx (+)= (dy W) * act'(x): ReLU through the GEMM's mask epilogue, Softplus through a product buffer
  auto dgrad_act = [&](const float* dy, int ldy, int N, const float* W, const float* x, long long M, float* dx, bool accumulate) -> int {
    if (!soft) return diner_gemm_f32(dy, W, dx, M, H, N, ldy, H, H, kExact | (accumulate ? kAccum : 0), nullptr, x, 1, stream);
    int r = diner_gemm_f32(dy, W, T, M, H, N, ldy, H, H, kExact, nullptr, nullptr, 1, stream);
    if (r) return r;
    hipLaunchKernelGGL(k_softplus_bwd, dim3(grid1d(M * H)), dim3(256), 0, st, x, T, M * H, beta, accumulate ? 1 : 0, dx);
    return 0;
  };
  long long M = p->n_blocks > 0 ? L.rows[p->n_blocks - 1] : (long long)nv * B;
  if (p->combine_layer >= p->n_blocks || p->combine_layer < 0) M = (long long)nv * B;
  if (d_zx) hipLaunchKernelGGL(k_zero_cols, dim3(grid1d((long long)nv * B * D)), dim3(256), 0, st, d_zx, (long long)nv * B, D, D);
  // lin_out
  if ((rc = wgrad(d_out, NO, NO, ws + L.x_last, H, H, true, M, g->lin_out_w, g->lin_out_b))) return rc;
  if ((rc = dgrad_act(d_out, NO, NO, p->lin_out_w, ws + L.x_last, M, G, false))) return rc;
  for (int b = p->n_blocks - 1; b >= 0; --b) {
    const float* X = ws + L.X[b];
    const float* Hb = ws + L.H[b];
    M = L.rows[b];
    // in_{b+1} = X_b + fc_1(act(H_b)); G = d in_{b+1}
    if ((rc = wgrad(G, H, H, Hb, H, H, true, M, g->fc1_w[b], g->fc1_b[b]))) return rc;
    if ((rc = dgrad_act(G, H, H, p->fc1_w[b], Hb, M, GH, false))) return rc;
    // H_b = fc_0(act(X_b)); G += (dH W0) * act'(X_b)   (the residual branch's gradient is G itself)
    if ((rc = wgrad(GH, H, H, X, H, H, true, M, g->fc0_w[b], g->fc0_b[b]))) return rc;
    if ((rc = dgrad_act(GH, H, H, p->fc0_w[b], X, M, G, true))) return rc;
    if (p->d_latent > 0 && b < p->combine_layer) {                // X_b = in_b + lin_z_b(z): no activation on either side
      if ((rc = wgrad(G, H, H, zx, D, p->d_latent, false, M, g->lin_z_w[b], g->lin_z_b[b]))) return rc;
      if (d_zx && (rc = diner_gemm_f32(G, p->lin_z_w[b], d_zx, M, p->d_latent, H, H, p->d_latent, D, kExact | kAccum, nullptr, nullptr, 1, stream))) return rc;
    }
    if (b == p->combine_layer) {                                  // X_b = mean over the views of in_b: d in_b = G / nv on every view
      if ((rc = diner_view_mean_f32(G, nv, B * (long long)H, GH, 1, stream))) return rc;
      float* t = G; G = GH; GH = t;
    }
  }
  M = (long long)nv * B;
  if (p->d_in > 0) {
    if ((rc = wgrad(G, H, H, zx + p->d_latent, D, p->d_in, false, M, g->lin_in_w, g->lin_in_b))) return rc;
    if (d_zx && (rc = diner_gemm_f32(G, p->lin_in_w, d_zx + p->d_latent, M, p->d_in, H, H, p->d_in, D, kExact | kAccum, nullptr, nullptr, 1, stream))) return rc;
  }
  DINER_LAUNCH_OK();
  return 0;
}

// adjoint of the latent lookup inside diner_field_inputs_generic_f32: d_latent_cl (nv, Hf, Wf, C) channels-last, OVERWRITTEN, from the first C
// columns of d_zx (nv, P, d_row) -- what diner_mlp_generic_backward_f32 returns (image_encoder.py:97-146 under autograd)
extern "C" int diner_field_inputs_generic_bwd_f32(const DinerScene* scene, const float* xyz, const float* viewdirs, long long P, int d_row,
                                                  const float* d_zx, float* d_latent_cl, void* stream) {
  DINER_CHECK_ARG(scene && xyz && viewdirs && d_zx && d_latent_cl && P > 0, "field_inputs_generic_bwd: bad arguments");
  SceneDev sd;
  int rc = make_scene_dev(scene, &sd);
  if (rc) return rc;
  DINER_CHECK_ARG(sd.C > 0 && d_row >= sd.C && scene->latent_cl, "field_inputs_generic_bwd: latent missing or d_row < C");
  FieldArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.xyz = xyz;
  fa.viewdirs = viewdirs;
  fa.K = 1;
  fa.P = P;
  DINER_HIP_OK(hipMemsetAsync(d_latent_cl, 0, (size_t)sd.nv * sd.Hf * sd.Wf * sd.C * sizeof(float), (hipStream_t)stream));
  const long long waves = P * sd.nv;
  hipLaunchKernelGGL(k_generic_latent_bwd, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, sd, fa, d_row, d_zx, d_latent_cl);
  DINER_LAUNCH_OK();
  return 0;
}
