// Micro-benchmark 6: does straight-line VALU code larger than the instruction cache slow down?  (one wave per SIMD, like the field kernels)
// N VOP3 instructions (8 bytes each) of straight-line code per loop iteration, 8 independent dependency chains.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__global__ __launch_bounds__(256, 1) void k(float* out, const float* in, int iters, unsigned long long* clk) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = in[threadIdx.x + 64 * i];
  const float a = in[1], b = in[2];
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < N / 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *clk = t1 - t0;
}
template <int N>
void run(int grid) {
  float *out, *in; unsigned long long* clk;
  (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&in, 4096 * 4); (void)hipMemset(in, 0x3c, 4096 * 4); (void)hipMalloc(&clk, 8);
  const int iters = (4000 * 1024) / N;
  hipLaunchKernelGGL(k<N>, dim3(grid), dim3(256), 0, 0, out, in, 4, clk);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL(k<N>, dim3(grid), dim3(256), 0, 0, out, in, iters, clk);
  (void)hipDeviceSynchronize();
  unsigned long long c; (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  printf("grid %4d  %6d VALU/iter (%4d KB code): %.2f shader clocks per instruction\n", grid, N, N * 8 / 1024, (double)c / ((double)iters * N));
}
int main() {
  for (int grid : {1, 256}) { run<1024>(grid); run<4096>(grid); run<8192>(grid); run<16384>(grid); run<32768>(grid); }
  return 0;
}
