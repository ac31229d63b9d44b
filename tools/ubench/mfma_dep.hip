// Dependent-accumulator issue rate of the fp16 MFMAs on gfx950: N MFMAs round-robin over NACC accumulators, one wave per SIMD.
// usage: ./mfma_dep   -> cycles per MFMA for NACC = 1, 2, 3, 4, 8 (32x32x16) and 1, 2, 4, 8, 16 (16x16x32)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256, 1) void k32(float* out, long long* cyc, int iters) {
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(0.5f + j * 0.01f); }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(256, 1) void k16(float* out, long long* cyc, int iters) {
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(0.5f + j * 0.01f); }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <class K>
void run(const char* name, K kern, int nacc) {
  float* out; long long* cyc; long long h;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 8 * nacc;
  printf("%s NACC=%2d: %.2f s_memtime ticks per MFMA (100 MHz ticks x%.0f), %.3f us per 1000 MFMA -> %.1f cycles at 2.4 GHz\n", name, nacc,
         h / n, 1.0, ms * 1e3 / n * 1000, ms * 1e-3 / n * 2.4e9);
  hipFree(out); hipFree(cyc);
}
int main() {
  run("32x32x16", k32<1>, 1); run("32x32x16", k32<2>, 2); run("32x32x16", k32<3>, 3); run("32x32x16", k32<4>, 4); run("32x32x16", k32<8>, 8);
  run("16x16x32", k16<1>, 1); run("16x16x32", k16<2>, 2); run("16x16x32", k16<4>, 4); run("16x16x32", k16<8>, 8); run("16x16x32", k16<16>, 16);
  return 0;
}
