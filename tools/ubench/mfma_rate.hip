// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 from one wave per SIMD for different accumulator patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 128 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  f32x4 s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int NACC>
void run(const char* name) {
  float* out; hipMalloc(&out, 256 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(256), 0, 0, out, 100, 1.0f, 2.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flop = 256.0 * 4 * iters * 128 * 2048;
  printf("%-10s %8.3f ms  %7.1f TFLOP/s  (%.1f cycles/MFMA at 2.4 GHz)\n", name, ms, flop / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 128.0));
  hipFree(out);
}
int main() { run<1>("1 acc"); run<2>("2 acc"); run<4>("4 acc"); run<8>("8 acc"); return 0; }
