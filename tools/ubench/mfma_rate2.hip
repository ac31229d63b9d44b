// Micro-benchmark 2: does D != C (accumulator renaming) or many distinct A/B registers slow v_mfma_f32_16x16x4_f32?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, const float* in, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a[16], b[16];
  for (int i = 0; i < 16; ++i) { a[i] = in[threadIdx.x + 64 * i] * 1e-3f; b[i] = in[threadIdx.x + 64 * i + 1024] * 1e-3f; }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {            // in place, distinct A/B registers
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[(r + i) & 15], acc[i], 0, 0, 0);
    } else {                    // D != C via explicit asm: rotate two register sets
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
          f32x4 t0, t1;
          asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %3, %4\n\tv_mfma_f32_16x16x4_f32 %1, %2, %3, %5"
                       : "=&a"(t0), "=&a"(t1) : "v"(a[r]), "v"(b[(r + i) & 15]), "a"(acc[i]), "a"(acc[i + 1]));
          acc[i] = t0; acc[i + 1] = t1;
        }
    }
  }
  f32x4 s = acc[0];
  for (int i = 1; i < 8; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int MODE>
void run(const char* name) {
  float *out, *in; (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&in, 4096 * 4); (void)hipMemset(in, 0x3c, 4096 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, in, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, in, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double flop = 256.0 * 4 * iters * 128 * 2048;
  printf("%-28s %8.3f ms  %7.1f TFLOP/s  (%.1f cycles/MFMA at 2.4 GHz)\n", name, ms, flop / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 128.0));
}
int main() { run<0>("in-place, 32 A/B regs"); run<1>("D != C (asm)"); return 0; }
