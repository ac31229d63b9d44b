// Micro-benchmark 8: does vector-ALU work issued between MFMAs of the same wave hide under them?  One wave per SIMD; a stream of
// independent v_mfma_f32_16x16x32_f16 (8 accumulators round-robin) with NV VALU instructions (independent v_fma_f32 chains) after
// each MFMA.  Reports shader clocks per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int NV, int KIND>
__global__ __launch_bounds__(256, 1) void k(float* out, const float* in, int iters, unsigned long long* clk) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[threadIdx.x + i]; b[i] = (_Float16)in[threadIdx.x + 8 + i]; }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = in[threadIdx.x + 64 * i];
  const float c = in[1], d = in[2];
  double y[4]; for (int i = 0; i < 4; ++i) y[i] = in[i];
  const double yc = in[5], yd = in[6];
  float z[8]; for (int i = 0; i < 8; ++i) z[i] = in[i + 9];
  const unsigned long long sbase = __builtin_amdgcn_readfirstlane((int)in[3]);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 64; ++r) {
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[r & 7]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(r * NV + j) & 7]) : "v"(c), "v"(d));
        if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&x[2 * ((r * NV + j) & 3)]) : "v"(*(const double*)&x[0]));
        if (KIND == 2) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x[(r * NV + j) & 7]) : "a"(acc[(r + 4) & 7][j & 3]));
        if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(y[(r * NV + j) & 3]) : "v"(yc), "v"(yd));
        if (KIND == 4) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y[(r * NV + j) & 3]) : "v"(yc), "v"(yd));
        if (KIND == 5) asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(y[(r * NV + j) & 3]) : "s"(sbase), "v"(yd));
        if (KIND == 6) asm volatile("v_mov_b64 %0, %1" : "=v"(y[(r * NV + j) & 3]) : "v"(yc));
        if (KIND == 7) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(z[(r * NV + j) & 7]) : "v"(c));
        if (KIND == 8) asm volatile("v_add_u32 %0, %1, %2" : "=v"(x[(r * NV + j) & 7]) : "v"(c), "v"(d));
        if (KIND == 9) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[(r * NV + j) & 7]) : "v"(c), "v"(d));
        if (KIND == 10) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x[(r * NV + j) & 7]) : "v"(c), "v"(d));
        if (KIND == 11) asm volatile("v_pk_fma_f16 %0, %1, %2, %2" : "=v"(x[(r * NV + j) & 7]) : "v"(c), "v"(d));
        if (KIND == 12) asm volatile("v_fma_f64 %0, %1, %2, %2" : "=v"(y[(r * NV + j) & 3]) : "v"(yc), "v"(yd));
        // round 5: what the fp16-tap blend of the plain-fp16 kernels is made of
        if (KIND == 13) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(x[(r * NV + j) & 7]) : "v"(c), "v"(d));
        if (KIND == 14) asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(x[(r * NV + j) & 7]) : "v"(c));
        if (KIND == 15) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(x[(r * NV + j) & 7]) : "v"(c));
        if (KIND == 16) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x[(r * NV + j) & 7]) : "v"(c), "v"(d));
        if (KIND == 17) asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(x[(r * NV + j) & 7]) : "v"(c));
        if (KIND == 18) asm volatile("v_lshl_add_u32 %0, %1, 10, %2" : "=v"(x[(r * NV + j) & 7]) : "v"(c), "v"(d));
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  f32x4 s = acc[0];
  for (int i = 1; i < 8; ++i) s += acc[i];
  float xs = 0.f;
  for (int i = 0; i < 8; ++i) xs += x[i] + z[i];
  for (int i = 0; i < 4; ++i) xs += (float)y[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + xs;
  if (blockIdx.x == 0 && threadIdx.x == 0) *clk = t1 - t0;
}
template <int NV, int KIND>
void run(const char* what) {
  float *out, *in; unsigned long long* clk;
  (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&in, 4096 * 4); (void)hipMemset(in, 0x3c, 4096 * 4); (void)hipMalloc(&clk, 8);
  const int iters = 2000;
  hipLaunchKernelGGL((k<NV, KIND>), dim3(256), dim3(256), 0, 0, out, in, 4, clk);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL((k<NV, KIND>), dim3(256), dim3(256), 0, 0, out, in, iters, clk);
  (void)hipDeviceSynchronize();
  unsigned long long c; (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  printf("%d x %-18s per MFMA: %.2f shader clocks per MFMA\n", NV, what, (double)c / ((double)iters * 64));
}
int main() {
  run<0, 0>("(nothing)");
  run<1, 0>("v_fma_f32"); run<2, 0>("v_fma_f32"); run<3, 0>("v_fma_f32"); run<4, 0>("v_fma_f32"); run<6, 0>("v_fma_f32");
  run<1, 1>("v_pk_mul_f32"); run<2, 1>("v_pk_mul_f32"); run<3, 1>("v_pk_mul_f32");
  run<1, 2>("v_accvgpr_read"); run<2, 2>("v_accvgpr_read"); run<4, 2>("v_accvgpr_read");
  run<1, 3>("v_pk_mul_f32 indep"); run<2, 3>("v_pk_mul_f32 indep");
  run<1, 4>("v_pk_add_f32 indep"); run<2, 4>("v_pk_add_f32 indep");
  run<1, 5>("v_lshl_add_u64"); run<2, 5>("v_lshl_add_u64");
  run<1, 6>("v_mov_b64"); run<2, 6>("v_mov_b64");
  run<1, 7>("v_accvgpr_write"); run<2, 7>("v_accvgpr_write"); run<4, 7>("v_accvgpr_write");
  run<2, 8>("v_add_u32"); run<2, 9>("v_mul_f32"); run<2, 10>("v_cvt_pk_f16_f32"); run<2, 11>("v_pk_fma_f16");
  run<1, 12>("v_fma_f64");
  run<1, 13>("v_fma_mix_f32"); run<2, 13>("v_fma_mix_f32"); run<4, 13>("v_fma_mix_f32");
  run<1, 14>("v_cvt_f32_f16"); run<2, 14>("v_cvt_f32_f16"); run<4, 14>("v_cvt_f32_f16");
  run<1, 15>("v_cvt_f32_f16_sdwa"); run<2, 15>("v_cvt_f32_f16_sdwa"); run<4, 15>("v_cvt_f32_f16_sdwa");
  run<2, 16>("v_fmac_f32"); run<4, 16>("v_fmac_f32");
  run<2, 17>("v_lshrrev_b32"); run<2, 18>("v_lshl_add_u32");
  return 0;
}
