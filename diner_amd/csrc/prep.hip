// Per-image preparation either side of the renderer, once per image:
//   depth2normal   normal maps from depth maps by central differences      (reference src/util/depth2normal.py:7-87)
//   gen_rays       pixel-centre rays of a pinhole camera, row-major (H, W) (reference src/util/cam_geometry.py:5-48)
//   image output   8-bit quantisation of rgb and the colour-mapped depth (torchvision save_image / torch_cmap as used by
//                  DINER.create_prediction_folder, diner.py:119-133; torch_helpers.py:42-75)
// All are HBM-trivial (a few bytes per pixel); they exist so that encode, ray generation and output preparation stay on
// the device and a sharded rank generates only its own ray range.
#include "common.hpp"

namespace diner {

struct PixelCam {      // fx, fy, cx, cy of one (N,3,3) intrinsics matrix
  float fx, fy, cx, cy;
};
__device__ __forceinline__ PixelCam load_cam(const float* __restrict__ K, int n) {
  const float* k = K + (size_t)n * 9;
  return {k[0], k[4], k[2], k[5]};
}

// back-projected x coordinate of pixel (i, j): ((j + 0.5) - cx) / fx * depth      (depth2normal.py:23-31)
__device__ __forceinline__ float ray_x(const PixelCam& c, int j) { return __fdiv_rn(__fsub_rn((float)j + 0.5f, c.cx), c.fx); }
__device__ __forceinline__ float ray_y(const PixelCam& c, int i) { return __fdiv_rn(__fsub_rn((float)i + 0.5f, c.cy), c.fy); }

struct P3 {
  float x, y, z;
};
__device__ __forceinline__ P3 point_at(const float* __restrict__ d, const PixelCam& c, int H, int W, int i, int j) {
  i = min(max(i, 0), H - 1);          // replicate padding (:32)
  j = min(max(j, 0), W - 1);
  const float z = d[(size_t)i * W + j];
  return {__fmul_rn(ray_x(c, j), z), __fmul_rn(ray_y(c, i), z), z};
}

// un-cleaned normal at (i, j): normalize(cross(down - up, right - left))          (:46-55)
__device__ __forceinline__ P3 raw_normal(const float* __restrict__ d, const PixelCam& c, int H, int W, int i, int j) {
  const P3 dn = point_at(d, c, H, W, i + 1, j), up = point_at(d, c, H, W, i - 1, j);
  const P3 rt = point_at(d, c, H, W, i, j + 1), lf = point_at(d, c, H, W, i, j - 1);
  const float v0 = __fsub_rn(dn.x, up.x), v1 = __fsub_rn(dn.y, up.y), v2 = __fsub_rn(dn.z, up.z);
  const float h0 = __fsub_rn(rt.x, lf.x), h1 = __fsub_rn(rt.y, lf.y), h2 = __fsub_rn(rt.z, lf.z);
  const float c0 = __fsub_rn(__fmul_rn(v1, h2), __fmul_rn(v2, h1));
  const float c1 = __fsub_rn(__fmul_rn(v2, h0), __fmul_rn(v0, h2));
  const float c2 = __fsub_rn(__fmul_rn(v0, h1), __fmul_rn(v1, h0));
  const float nrm = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(c0, c0), __fmul_rn(c1, c1)), __fmul_rn(c2, c2)));
  return {__fdiv_rn(c0, nrm), __fdiv_rn(c1, nrm), __fdiv_rn(c2, nrm)};      // 0/0 = NaN as in the reference
}

// One thread per pixel.  Pixels with a background neighbour (x of the neighbour's point == 0, the reference's test,
// :61-72) take the un-cleaned normal of the pixel shifted AWAY from the hole (:74-78); background pixels get 0 (:79).
__global__ void k_depth2normal(const float* __restrict__ dmap, const float* __restrict__ K, int N, int H, int W,
                               float* __restrict__ out) {
  const long long total = (long long)N * H * W;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(idx / ((long long)H * W));
    const int rem = (int)(idx - (long long)n * H * W);
    const int i = rem / W, j = rem - i * W;
    const float* d = dmap + (size_t)n * H * W;
    const PixelCam c = load_cam(K, n);
    float* o = out + (size_t)n * 3 * H * W + (size_t)i * W + j;
    if (d[(size_t)i * W + j] == 0.0f) {
      o[0] = 0.0f;
      o[(size_t)H * W] = 0.0f;
      o[(size_t)2 * H * W] = 0.0f;
      continue;
    }
    const int dy = (point_at(d, c, H, W, i - 1, j).x == 0.0f ? 1 : 0) - (point_at(d, c, H, W, i + 1, j).x == 0.0f ? 1 : 0);
    const int dx = (point_at(d, c, H, W, i, j - 1).x == 0.0f ? 1 : 0) - (point_at(d, c, H, W, i, j + 1).x == 0.0f ? 1 : 0);
    const int si = min(max(i + dy, 0), H - 1), sj = min(max(j + dx, 0), W - 1);
    const P3 nm = raw_normal(d, c, H, W, si, sj);
    o[0] = nm.x;
    o[(size_t)H * W] = nm.y;
    o[(size_t)2 * H * W] = nm.z;
  }
}

struct RayCam {        // per camera: R (world->cam, row-major), origin = -R^T t, fx, fy, cx, cy, near, far
  float R[9], o[3], fx, fy, cx, cy, zn, zf;
};
constexpr int kMaxRayCams = 16;
struct RayCams {
  RayCam cam[kMaxRayCams];
};

// rays [ray0, ray0 + n) of each camera's row-major (H, W) list: [origin, world direction, near, far]
__global__ void k_gen_rays(RayCams cams, int B, int W, long long ray0, long long n, float* __restrict__ out) {
  const long long total = (long long)B * n;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(idx / n);
    const long long r = ray0 + (idx - (long long)b * n);
    const int i = (int)(r / W), j = (int)(r - (long long)i * W);
    const RayCam& c = cams.cam[b];
    const float x = __fdiv_rn(__fsub_rn((float)j + 0.5f, c.cx), c.fx);       // cam_geometry.py:25-31
    const float y = __fdiv_rn(__fsub_rn((float)i + 0.5f, c.cy), c.fy);
    const float nrm = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), 1.0f));
    const float dx = __fdiv_rn(x, nrm), dy = __fdiv_rn(y, nrm), dz = __fdiv_rn(1.0f, nrm);
    float* o = out + idx * 8;
    o[0] = c.o[0];
    o[1] = c.o[1];
    o[2] = c.o[2];
    // d_w = R^T d_cam (:37-38)
    o[3] = __fmaf_rn(c.R[6], dz, __fmaf_rn(c.R[3], dy, __fmul_rn(c.R[0], dx)));
    o[4] = __fmaf_rn(c.R[7], dz, __fmaf_rn(c.R[4], dy, __fmul_rn(c.R[1], dx)));
    o[5] = __fmaf_rn(c.R[8], dz, __fmaf_rn(c.R[5], dy, __fmul_rn(c.R[2], dx)));
    o[6] = c.zn;
    o[7] = c.zf;
  }
}

// ---- image output -------------------------------------------------------------------------------------------------
// save_image's quantisation: uint8(clamp(v * 255 + 0.5, 0, 255)) in fp32, (3,H,W) planar -> (H,W,3) interleaved
__global__ void k_quantize_rgb(const float* __restrict__ img, long long HW, unsigned char* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = __fadd_rn(__fmul_rn(img[(size_t)c * HW + i], 255.0f), 0.5f);
      v = fminf(fmaxf(v, 0.0f), 255.0f);                                   // NaN -> 0 like clamp_ + the uint8 cast of 0
      out[i * 3 + c] = (unsigned char)(v != v ? 0.0f : v);
    }
  }
}

// order-preserving integer image of a float (for atomicMin / atomicMax)
__device__ __forceinline__ int float_key(float f) {
  const int b = __float_as_int(f);
  return b >= 0 ? b : b ^ 0x7fffffff;
}
__global__ void k_minmax(const float* __restrict__ x, long long n, int* __restrict__ keys) {      // keys: {min, max}
  int lo = 0x7fffffff, hi = (int)0x80000000;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = x[i];
    if (v == v) {                                                            // numpy's min / max would return NaN; see host
      const int k = float_key(v);
      lo = min(lo, k);
      hi = max(hi, k);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, __shfl_xor(lo, o));
    hi = max(hi, __shfl_xor(hi, o));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(keys, lo);
    atomicMax(keys + 1, hi);
  }
}
// matplotlib colormap lookup: index = int(xn * 256) clipped to [0, 255] with xn = (x - vmin) / (vmax - vmin) in
// float64 (torch_cmap converts to float64 first), NaN -> the "bad" colour (0,0,0); lut: 256 x 3 uint8, already quantised
// like save_image does
__global__ void k_colormap(const float* __restrict__ x, long long n, const unsigned char* __restrict__ lut, double vmin,
                           double vmax, unsigned char* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double xn = ((double)x[i] - vmin) / (vmax - vmin);
    unsigned char r = 0, g = 0, b = 0;
    if (xn == xn) {
      int idx = xn >= 1.0 ? 255 : (xn < 0.0 ? 0 : (int)(xn * 256.0));
      idx = min(max(idx, 0), 255);
      r = lut[3 * idx];
      g = lut[3 * idx + 1];
      b = lut[3 * idx + 2];
    }
    out[i * 3] = r;
    out[i * 3 + 1] = g;
    out[i * 3 + 2] = b;
  }
}

}  // namespace diner

using namespace diner;

extern "C" int diner_depth2normal_f32(const float* dmap, const float* K, int N, int H, int W, float* out, void* stream) {
  DINER_CHECK_ARG(dmap && K && out, "depth2normal: null pointer argument");
  DINER_CHECK_ARG(N >= 0 && H > 0 && W > 0, "depth2normal: bad shape N=%d H=%d W=%d", N, H, W);
  if (N == 0) return 0;
  const long long total = (long long)N * H * W;
  const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  hipLaunchKernelGGL(k_depth2normal, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dmap, K, N, H, W, out);
  DINER_LAUNCH_OK();
  return 0;
}

extern "C" int diner_gen_rays_f32(const float* extrinsics, const float* intrinsics, const float* z_near,
                                  const float* z_far, int B, int W, int H, long long ray0, long long n_rays, float* out,
                                  void* stream) {
  DINER_CHECK_ARG(extrinsics && intrinsics && z_near && z_far && (out || B == 0 || n_rays == 0), "gen_rays: null pointer argument");
  DINER_CHECK_ARG(B >= 0 && B <= kMaxRayCams, "gen_rays: %d cameras, at most %d per call", B, kMaxRayCams);
  DINER_CHECK_ARG(W > 0 && H > 0, "gen_rays: bad image size %d x %d", W, H);
  DINER_CHECK_ARG(ray0 >= 0 && n_rays >= 0 && ray0 + n_rays <= (long long)W * H,
                  "gen_rays: ray range [%lld, %lld) outside the %d x %d image", ray0, ray0 + n_rays, W, H);
  if (B == 0 || n_rays == 0) return 0;
  RayCams cams;
  for (int b = 0; b < B; ++b) {
    const float* E = extrinsics + 16 * b;
    const float* Kb = intrinsics + 9 * b;
    RayCam& c = cams.cam[b];
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 3; ++k) c.R[3 * r + k] = E[4 * r + k];
    for (int k = 0; k < 3; ++k)      // origin = -R^T t (:41), same association as the kernels' rot chain
      c.o[k] = -fmaf(E[4 * 2 + k], E[4 * 2 + 3], fmaf(E[4 * 1 + k], E[4 * 1 + 3], E[4 * 0 + k] * E[4 * 0 + 3]));
    c.fx = Kb[0];
    c.fy = Kb[4];
    c.cx = Kb[2];
    c.cy = Kb[5];
    c.zn = z_near[b];
    c.zf = z_far[b];
  }
  const long long total = (long long)B * n_rays;
  const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  hipLaunchKernelGGL(k_gen_rays, dim3(blocks), dim3(256), 0, (hipStream_t)stream, cams, B, W, ray0, n_rays, out);
  DINER_LAUNCH_OK();
  return 0;
}

extern "C" int diner_quantize_rgb_u8(const float* img, int H, int W, unsigned char* out, void* stream) {
  DINER_CHECK_ARG(img && out && H > 0 && W > 0, "quantize_rgb: bad arguments");
  const long long HW = (long long)H * W;
  const int blocks = (int)((HW + 255) / 256 > 8192 ? 8192 : (HW + 255) / 256);
  hipLaunchKernelGGL(k_quantize_rgb, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, HW, out);
  DINER_LAUNCH_OK();
  return 0;
}

extern "C" int diner_minmax_f32(const float* x, long long n, float* out2, void* stream) {
  DINER_CHECK_ARG(x && out2 && n > 0, "minmax: bad arguments");
  const int init[2] = {0x7fffffff, (int)0x80000000};
  DINER_HIP_OK(hipMemcpyAsync(out2, init, sizeof(init), hipMemcpyHostToDevice, (hipStream_t)stream));
  const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(k_minmax, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, (int*)out2);
  DINER_LAUNCH_OK();
  return 0;
}

extern "C" int diner_colormap_u8(const float* x, long long n, const unsigned char* lut_u8, double vmin, double vmax,
                                 unsigned char* out, void* stream) {
  DINER_CHECK_ARG(x && lut_u8 && out && n > 0, "colormap: bad arguments");
  const int blocks = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  hipLaunchKernelGGL(k_colormap, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, lut_u8, vmin, vmax, out);
  DINER_LAUNCH_OK();
  return 0;
}
