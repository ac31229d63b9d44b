// Micro-benchmark 3: does a straight-line MFMA stream larger than the instruction cache slow down?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int N>   // N MFMAs of straight-line code per loop iteration
__global__ __launch_bounds__(256, 1) void k(float* out, const float* in, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[threadIdx.x + 64 * i] * 1e-3f; b[i] = in[threadIdx.x + 64 * i + 1024] * 1e-3f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < N / 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r & 7], b[(r + i) & 7], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  f32x4 s = acc[0];
  for (int i = 1; i < 8; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int N>
void run() {
  float *out, *in; (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&in, 4096 * 4); (void)hipMemset(in, 0x3c, 4096 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = (20000 * 128) / N;
  hipLaunchKernelGGL(k<N>, dim3(256), dim3(256), 0, 0, out, in, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<N>, dim3(256), dim3(256), 0, 0, out, in, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double n = (double)iters * N;
  printf("%6d MFMAs/iter (%4d KB code): %8.3f ms  %.2f cycles/MFMA at 2.4 GHz\n", N, N * 8 / 1024, ms, ms * 1e-3 * 2.4e9 / n);
}
int main() { run<1024>(); run<4096>(); run<8192>(); run<16384>(); run<32768>(); return 0; }
