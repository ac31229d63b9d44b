// Training path: the kernel entry point and the launchers of the 512 x 512 layer products.  The workgroup bodies live in
// train_lin512.hip (forward / data gradient, three tile shapes) and train_wgrad512.hip (weight / bias gradient), included here: ONE
// kernel runs up to three independent parts -- workgroups [0, n0) the first row range of a product in its shape, [n0, n0 + n1) the
// second, the rest the weight-gradient product of the same layer.  The hardware hands out workgroups in index order, one per CU (LDS and
// registers admit no second one), so a part starts on whichever CU finishes the one before first: no kernel boundary (drain, launch,
// ramp: ~5 us, of a 75 us product at the reference training batch) between the shapes of a launch plan, nor between the data- and the
// weight-gradient product of a layer, and the ragged ends of one part are filled by the next (a 5120-row data gradient occupies 160
// CUs: the weight gradient's workgroups take the other 96 at once).
#include "train_lin512.hip"
#include "train_wgrad512.hip"

namespace diner {
namespace train {

enum : int { kShape64 = 0, kShape32 = 1, kShape32Shared = 2, kShape128 = 3 };      // 64-row tiles; 32-row tiles; 32-row tiles shared by two workgroups;
                                                                                   // 128-row tiles (f16x3 only, round 4: a weight fragment feeds four MFMAs)
struct Run512 {
  Lin512Args part[2];
  int n[2];              // workgroups of the two parts (0: absent)
  int shape[2];
  Wgrad512Args wg;       // weight-gradient part: the remaining workgroups of the grid (none: grid = n[0] + n[1])
};

__device__ __forceinline__ void lin512_part(const Lin512Args& a, int shape, int bid, int nblk) {
  if (shape == kShape64) lin512_body<DINER_L512_RING, 2, 1>(a, bid, nblk);
  else if (shape == kShape32) lin512_body<DINER_L512_RING, 1, 1>(a, bid, nblk);
  else lin512_body<DINER_L512_RING, 1, 2>(a, bid, nblk);
}

// The forward products in the f16x3 arithmetic (lin512_body<.., AR = 1>: half the MFMAs of bf16x6): the same parts, no weight gradient
__device__ __forceinline__ void lin512_part_f16(const Lin512Args& a, int shape, int bid, int nblk) {
  if (shape == kShape128) lin512_body<DINER_L512_RING, 4, 1, 1>(a, bid, nblk);
  else if (shape == kShape64) lin512_body<DINER_L512_RING, 2, 1, 1>(a, bid, nblk);
  else if (shape == kShape32) lin512_body<DINER_L512_RING, 1, 1, 1>(a, bid, nblk);
  else lin512_body<DINER_L512_RING, 1, 2, 1>(a, bid, nblk);
}
__global__ __launch_bounds__(256, 1) void k_fwd512_f16x3(Run512 r) {
  int b = blockIdx.x;
  if (b < r.n[0]) return lin512_part_f16(r.part[0], r.shape[0], b, r.n[0]);
  lin512_part_f16(r.part[1], r.shape[1], b - r.n[0], r.n[1]);
}
// round 5 experiment (DINER_L512_W2=1): the forward products with TWO workgroups per CU (two waves per SIMD, 256 registers each): 64- and
// 32-row tiles only (128 accumulator registers), 66 KB of LDS per workgroup -- one workgroup's epilogue and staging stalls under the other's MFMAs
__device__ __forceinline__ void lin512_part_f16_w2(const Lin512Args& a, int shape, int bid, int nblk) {
  if (shape == kShape64) lin512_body<DINER_L512_RING, 2, 1, 1>(a, bid, nblk);
  else if (shape == kShape32) lin512_body<DINER_L512_RING, 1, 1, 1>(a, bid, nblk);
  else lin512_body<DINER_L512_RING, 1, 2, 1>(a, bid, nblk);
}
__global__ __launch_bounds__(256, 2) void k_fwd512_f16x3_w2(Run512 r) {
  int b = blockIdx.x;
  if (b < r.n[0]) return lin512_part_f16_w2(r.part[0], r.shape[0], b, r.n[0]);
  lin512_part_f16_w2(r.part[1], r.shape[1], b - r.n[0], r.n[1]);
}
// round 4: the backward's launch (data gradient in its shapes + the weight gradient of the same layer) in the f16x3 arithmetic
__global__ __launch_bounds__(256, 1) void k_run512_f16x3(Run512 r) {
  int b = blockIdx.x;
  if (b < r.n[0]) return lin512_part_f16(r.part[0], r.shape[0], b, r.n[0]);
  b -= r.n[0];
  if (b < r.n[1]) return lin512_part_f16(r.part[1], r.shape[1], b, r.n[1]);
  if (r.wg.wide) wgrad512_body_wide(r.wg, b - r.n[1]);
  else wgrad512_body<1>(r.wg, b - r.n[1]);
}

__global__ __launch_bounds__(256, 1) void k_run512(Run512 r) {
  int b = blockIdx.x;
  if (b < r.n[0]) return lin512_part(r.part[0], r.shape[0], b, r.n[0]);
  b -= r.n[0];
  if (b < r.n[1]) return lin512_part(r.part[1], r.shape[1], b, r.n[1]);
  wgrad512_body<0>(r.wg, b - r.n[1]);
}

// round 6: a product over a row list whose length lives on the device (Lin512Args.m_dev: the latent rows a training batch touches): 32-row
// tiles on every CU, AR = 1 (f16x3) or 0 (its bf16x6 twin)
template <int AR>
__global__ __launch_bounds__(256, 1) void k_lin512_rows(Lin512Args a, int nblk) {
  lin512_body<DINER_L512_RING, 1, 1, AR, 4, true>(a, blockIdx.x, nblk);
}
// round 5: the data gradient of an f16x3 launch on eight waves (lin512_body<.., NW = 8>: two waves per SIMD, 64 features and half the staging
// rows per wave; the shared 32-row shape of a plan runs as plain 32-row tiles -- its second workgroups find no tile)
__global__ __launch_bounds__(512, 1) void k_dgrad512_w8(Run512 r) {
  int b = blockIdx.x;
  const int i = b < r.n[0] ? 0 : 1;
  if (i) b -= r.n[0];
  const Lin512Args& a = r.part[i];
  const int shape = r.shape[i], nblk = r.n[i];
  if (shape == kShape128) lin512_body<DINER_L512_RING, 4, 1, 1, 8>(a, b, nblk);
  else if (shape == kShape64) lin512_body<DINER_L512_RING, 2, 1, 1, 8>(a, b, nblk);
  else lin512_body<DINER_L512_RING, 1, 1, 1, 8>(a, b, nblk);
}
// round 5: the weight gradient of an f16x3 launch on eight waves (wgrad512_body_w8), a launch of its own behind the data gradient's
__global__ __launch_bounds__(512, 1) void k_wgrad512_w8(Wgrad512Args a) { wgrad512_body_w8(a, blockIdx.x); }

namespace {
constexpr size_t kLdsBytesRun = kLdsBytesWgrad > kLdsBytes512 ? kLdsBytesWgrad : kLdsBytes512;
static_assert(kLdsBytesRun >= kLdsBytes512F16, "the f16x3 128-row shape needs 128 KB");
int device_cus(int* cus) {                                   // per device: dynamic LDS size of the kernel, CU count
  static std::atomic<int> attr_set[64];
  static std::atomic<int> cu_count[64];
  int dev = 0;
  DINER_HIP_OK(hipGetDevice(&dev));
  dev &= 63;
  if (!attr_set[dev].load()) {
    DINER_HIP_OK(hipFuncSetAttribute((const void*)k_run512, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytesRun));
    DINER_HIP_OK(hipFuncSetAttribute((const void*)k_fwd512_f16x3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes512F16));
    DINER_HIP_OK(hipFuncSetAttribute((const void*)k_fwd512_f16x3_w2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes512F16W2));
    DINER_HIP_OK(hipFuncSetAttribute((const void*)k_run512_f16x3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytesRun));
    DINER_HIP_OK(hipFuncSetAttribute((const void*)k_wgrad512_w8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytesRun));
    DINER_HIP_OK(hipFuncSetAttribute((const void*)k_dgrad512_w8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytesRun));
    DINER_HIP_OK(hipFuncSetAttribute((const void*)k_wgrad_in_f16x3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytesWgradIn));
    DINER_HIP_OK(hipFuncSetAttribute((const void*)k_lin512_rows<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytesRun));
    DINER_HIP_OK(hipFuncSetAttribute((const void*)k_lin512_rows<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytesRun));
    int c = 0;
    DINER_HIP_OK(hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev));
    cu_count[dev].store(c > 1 ? c & ~1 : 256);
    attr_set[dev].store(1);
  }
  *cus = cu_count[dev].load();
  return 0;
}

// The plan of a forward / data-gradient product: rounds of 64-row tiles over all CUs while a whole round is left (a weight fragment feeds
// twice the MFMAs there), and the ragged rest in the shape that costs least -- 32-row tiles, or 32-row tiles shared by two workgroups
// (half the features each).  Costs in units of one round of 32-row tiles, measured on the reference training batch (20480 rows = 2.5
// rounds of 32-row tiles: three rounds as 32-row tiles, 1.85 + 0.55 as one 64-row round + one round of shared tiles).
// DINER_L512_CT = 1 | 2 forces one shape for everything, DINER_L512_HALF = 0 keeps the shared tiles out (measurement aids).
void plan_lin512(const Lin512Args& a, int cus, Run512* r, bool f16 = false) {
  auto part = [&](long long row0, long long rows) {
    Lin512Args b = a;
    b.X += (size_t)row0 * a.ldx;
    b.Y += (size_t)row0 * a.ldy;
    if (a.resid) b.resid += (size_t)row0 * a.ldy;
    if (a.mask) b.mask += (size_t)row0 * a.ldy;
    if (a.maskbits) b.maskbits += (size_t)row0 * 16;
    if (a.resid2) b.resid2 += (size_t)row0 * a.ldy;
    if (a.X2) b.X2 += (size_t)row0 * a.ldx;
    b.M = rows;
    return b;
  };
  auto set = [&](int i, long long row0, long long rows, int shape) {
    r->part[i] = part(row0, rows);
    r->shape[i] = shape;
    const long long units = shape == kShape128 ? (rows + 127) / 128 : shape == kShape64 ? (rows + 63) / 64 : (rows + 31) / 32 * (shape == kShape32Shared ? 2 : 1);
    r->n[i] = (int)(units < cus ? units : cus);
  };
  r->n[0] = r->n[1] = 0;
  r->part[1] = a;
  r->shape[1] = kShape32;
  static const int forced = [] { const char* e = getenv("DINER_L512_CT"); return e ? atoi(e) : 0; }();
  static const bool halves = [] { const char* e = getenv("DINER_L512_HALF"); return !(e && *e == '0'); }();
  if (forced == 1 || forced == 2) return set(0, 0, a.M, forced == 2 ? kShape64 : kShape32);
  // f16x3 at 64-row tiles is bound by the weight stream through the vector-memory path (1 MB per tile for 24.7 k clocks of MFMAs: the
  // path's 64 B/clk; MfmaUtil 0.38, profiles/r04_train_1x4096_pmc_summary.md): whole rounds of 128-row tiles first (DINER_L512_T128=0: off)
  static const bool t128 = [] { const char* e = getenv("DINER_L512_T128"); return !(e && *e == '0'); }();
  if (f16 && t128 && a.M >= 128ll * cus) {
    const long long round128 = 128ll * cus, main128 = a.M / round128 * round128;
    set(0, 0, main128, kShape128);
    if (a.M > main128) {
      const long long rest = a.M - main128;
      int shape_rest = kShape32;
      const double c32r = 1.0, c64r = 1.85;
      auto rounds2 = [&](long long units) { return (double)((units + cus - 1) / cus); };
      const double r64 = rounds2((rest + 63) / 64) * c64r, r32 = rounds2((rest + 31) / 32) * c32r;
      set(1, main128, rest, r64 <= r32 ? kShape64 : shape_rest);
    }
    return;
  }
  const double c32 = 1.0, c64 = 1.85, chalf = 0.55;
  auto rounds = [&](long long units) { return (double)((units + cus - 1) / cus); };
  auto rest_cost = [&](long long rows, int* shape) {         // cheapest 32-row shape for `rows` rows
    const long long t32 = (rows + 31) / 32;
    const double whole = rounds(t32) * c32, shared = rounds(2 * t32) * chalf;
    *shape = halves && shared < whole ? kShape32Shared : kShape32;
    return *shape == kShape32Shared ? shared : whole;
  };
  const long long round64 = 64ll * cus;
  const long long main_rows = a.M / round64 * round64, rest = a.M - main_rows;
  int shape_all = kShape32, shape_rest = kShape32;
  const double all32 = rest_cost(a.M, &shape_all);
  const double all64 = rounds((a.M + 63) / 64) * c64;
  const double split = main_rows && rest ? (double)(main_rows / round64) * c64 + rest_cost(rest, &shape_rest) : 1e30;
  if (split < all32 && split < all64) {
    set(0, 0, main_rows, kShape64);
    set(1, main_rows, rest, shape_rest);
  } else if (all64 <= all32) {
    set(0, 0, a.M, kShape64);
  } else {
    set(0, 0, a.M, shape_all);
  }
}

// The weight-gradient part: row chunks -- 32 (8 tiles x 32 = 256 workgroups) unless a chunk would be shorter than 4 slabs; returns its
// workgroups and the chunks that have rows (the ones the summing pass reads)
int plan_wgrad512(const float* dY, int ldy, const float* X, int ldx, bool relu_x, float* dW, float* db, long long M, float* part,
                  Wgrad512Args* a, int* used, int max_chunks = kWgPlanChunks, bool wide = false) {
  long long n_chunks = max_chunks;
  while (n_chunks > 1 && (M + n_chunks - 1) / n_chunks < 128) n_chunks >>= 1;
  long long rows = (M + n_chunks - 1) / n_chunks;
  rows = (rows + 31) / 32 * 32;
  *a = Wgrad512Args{dY, X, dW, db, part, M, ldy, ldx, relu_x ? 1 : 0, (int)n_chunks, rows, nullptr, nullptr, nullptr, wide ? 1 : 0};
  *used = (int)((M + rows - 1) / rows);
  return (wide ? 4 : 8) * (int)(n_chunks <= 8 ? 8 : n_chunks);   // the block -> (tile, chunk) map needs whole groups of 8 chunks
}
}  // namespace

int wgrad_in_launch(const float* dY, int ldy, const float* F, int ldf, int n_in, long long M, float* dW, float* db, const unsigned* amax_dy,
                    hipStream_t stream) {
  DINER_CHECK_ARG(dY && F && dW && M > 0 && n_in > 0 && n_in <= 64 && ldf >= 64 && ldy >= 512 && (ldy & 3) == 0 &&
                  (reinterpret_cast<size_t>(dY) & 15) == 0 && (reinterpret_cast<size_t>(F) & 3) == 0, "wgrad_in: bad arguments");
  int cus = 0;
  int rc = device_cus(&cus);
  if (rc) return rc;
  long long chunks = cus;
  while (chunks > 1 && (M + chunks - 1) / chunks < 64) chunks >>= 1;
  long long rows = (M + chunks - 1) / chunks;
  rows = (rows + 31) / 32 * 32;
  const WgradInArgs a{dY, F, dW, db, M, rows, ldy, ldf, n_in, amax_dy};
  hipLaunchKernelGGL(k_wgrad_in_f16x3, dim3((unsigned)((M + rows - 1) / rows)), dim3(512), kLdsBytesWgradIn, stream, a);
  DINER_LAUNCH_OK();
  return 0;
}

int lin512_launch(const Lin512Args& a, hipStream_t stream, int arith) {
  int cus = 0;
  int rc = device_cus(&cus);
  if (rc) return rc;
  if (a.m_dev || a.skip_silent) {      // the row count lives on the device: 32-row tiles handed to every CU (a short list fills the chip best that way)
    const long long units = (a.M + 31) / 32;
    const int n = (int)(units < cus ? units : cus);
    if (arith == 1) hipLaunchKernelGGL(k_lin512_rows<1>, dim3(n), dim3(256), kLdsBytesRun, stream, a, n);
    else hipLaunchKernelGGL(k_lin512_rows<0>, dim3(n), dim3(256), kLdsBytesRun, stream, a, n);
    DINER_LAUNCH_OK();
    return 0;
  }
  Run512 r;
  plan_lin512(a, cus, &r, arith == 1);
  memset(&r.wg, 0, sizeof(r.wg));
  static const bool w2 = [] { const char* e = getenv("DINER_L512_W2"); return e && *e == '1'; }();
  if (arith == 1 && w2) {
    plan_lin512(a, 2 * cus, &r, false);      // (no 128-row shape: 256 accumulator registers do not fit two waves per SIMD)
    hipLaunchKernelGGL(k_fwd512_f16x3_w2, dim3(r.n[0] + r.n[1]), dim3(256), kLdsBytes512F16W2, stream, r);
  } else if (arith == 1) hipLaunchKernelGGL(k_fwd512_f16x3, dim3(r.n[0] + r.n[1]), dim3(256), kLdsBytes512F16, stream, r);
  else hipLaunchKernelGGL(k_run512, dim3(r.n[0] + r.n[1]), dim3(256), kLdsBytesRun, stream, r);
  DINER_LAUNCH_OK();
  return 0;
}

// dW (512, 512) += dY^T act(X), db (512) += column sums of dY, over M rows; dW / db zeroed by the caller.  part: null (atomics into dW) or
// wgrad512_part_bytes() of scratch (partial tiles stored per chunk + one reduction pass); with it overwrite = true makes dW / db plain
// outputs (no zeroing by the caller).  dgrad: the data-gradient product of the same layer, run by the same launch (or null).
int wgrad512_launch(const float* dY, int ldy, const float* X, int ldx, bool relu_x, float* dW, float* db, long long M,
                    hipStream_t stream, float* part, bool overwrite, WgReduceJob* defer, const Lin512Args* dgrad, const WgradArith* ar) {
  DINER_CHECK_ARG(part || !overwrite, "wgrad512: overwrite needs the scratch buffer");
  DINER_CHECK_ARG(part || !defer, "wgrad512: a deferred summing pass needs the scratch buffer");
  int cus = 0;
  int rc = device_cus(&cus);
  if (rc) return rc;
  Run512 r;
  r.n[0] = r.n[1] = 0;
  if (dgrad) plan_lin512(*dgrad, cus, &r, ar && ar->arith == 1);
  else {
    memset(r.part, 0, sizeof(r.part));
    r.shape[0] = r.shape[1] = kShape32;
  }
  int used = 0;
  // f16x3: 256 x 256 tiles over up to 64 row chunks (DINER_WGRAD_WIDE=0: the 128 x 256-tile body); the bf16x6 twin of such a launch keeps the
  // same chunks (its partial tiles go where the summing pass of the pair expects them), as 8 tiles x 64 chunks
  static const bool wide_on = [] { const char* e = getenv("DINER_WGRAD_WIDE"); return !(e && *e == '0'); }();
  const bool pair = ar && part && (ar->wg_skip || ar->wg_gate) && wide_on;
  const bool wide = pair && ar->arith == 1;
  const int n_wg = plan_wgrad512(dY, ldy, X, ldx, relu_x, dW, db, M, part, &r.wg, &used, pair ? kWgPlanChunksWide : kWgPlanChunks, wide);
  if (ar) {
    r.wg.amax_dy = ar->amax_dy;
    r.wg.skip = ar->wg_skip;
    r.wg.gate = ar->wg_gate;
  }
  // DINER_TRAIN_BWD_SPLIT=1 (measurement aid): the data-gradient parts and the weight-gradient part as two launches of the same kernel
  static const bool split = [] { const char* e = getenv("DINER_TRAIN_BWD_SPLIT"); return e && *e == '1'; }();
  // DINER_WGRAD_W8=0: the four-wave weight gradient inside the data gradient's launch (round 4; A/B measurement)
  static const bool w8 = [] { const char* e = getenv("DINER_WGRAD_W8"); return !(e && *e == '0'); }();
  if (w8 && wide && n_wg > 0) {
    // DINER_DGRAD_W8=1: the data gradient on eight waves too (round 5 experiment)
    static const bool d8 = [] { const char* e = getenv("DINER_DGRAD_W8"); return e && *e == '1'; }();
    if (r.n[0] + r.n[1] > 0) {
      if (d8) hipLaunchKernelGGL(k_dgrad512_w8, dim3(r.n[0] + r.n[1]), dim3(512), kLdsBytesRun, stream, r);
      else hipLaunchKernelGGL(k_run512_f16x3, dim3(r.n[0] + r.n[1]), dim3(256), kLdsBytesRun, stream, r);
    }
    hipLaunchKernelGGL(k_wgrad512_w8, dim3(n_wg), dim3(512), kLdsBytesRun, stream, r.wg);
  } else if (split && ar && ar->arith == 1 && r.n[0] + r.n[1] > 0 && n_wg > 0) {
    hipLaunchKernelGGL(k_run512_f16x3, dim3(r.n[0] + r.n[1]), dim3(256), kLdsBytesRun, stream, r);
    Run512 w2 = r;
    w2.n[0] = w2.n[1] = 0;
    hipLaunchKernelGGL(k_run512_f16x3, dim3(n_wg), dim3(256), kLdsBytesRun, stream, w2);
  } else if (ar && ar->arith == 1) hipLaunchKernelGGL(k_run512_f16x3, dim3(r.n[0] + r.n[1] + n_wg), dim3(256), kLdsBytesRun, stream, r);
  else hipLaunchKernelGGL(k_run512, dim3(r.n[0] + r.n[1] + n_wg), dim3(256), kLdsBytesRun, stream, r);
  if (part) {
    // chunks that start past M wrote nothing: only the chunks with rows are summed
    if (defer) *defer = WgReduceJob{part, dW, db, used};      // the caller sums (wgrad512_reduce_many)
    else hipLaunchKernelGGL(k_wgrad512_reduce, dim3(256), dim3(256), 0, stream, part, used, overwrite ? 1 : 0, dW, db);
  }
  DINER_LAUNCH_OK();
  return 0;
}

}  // namespace train
}  // namespace diner
