// Micro-benchmark 7: issue cost (shader clocks per instruction, one wave per SIMD) of the VALU instructions the publish step can be
// built from.  4096 instructions of straight-line code per loop iteration, 8 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#define BODY(NAME, ASM, ...)                                                                                      \
  __global__ __launch_bounds__(256, 1) void NAME(float* out, const float* in, int iters, unsigned long long* clk) { \
    float x[8], y[8];                                                                                             \
    for (int i = 0; i < 8; ++i) { x[i] = in[threadIdx.x + 64 * i]; y[i] = in[threadIdx.x + 64 * i + 512]; }     \
    float a = in[1], b = in[2];                                                                                   \
    double d[8]; for (int i = 0; i < 8; ++i) d[i] = in[i];                                                        \
    const unsigned long long t0 = __builtin_readcyclecounter();                                                   \
    for (int it = 0; it < iters; ++it) {                                                                          \
      _Pragma("unroll") for (int r = 0; r < 512; ++r) {                                                           \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : __VA_ARGS__);                            \
      }                                                                                                           \
    }                                                                                                             \
    const unsigned long long t1 = __builtin_readcyclecounter();                                                   \
    float s = 0.f; for (int i = 0; i < 8; ++i) s += x[i] + y[i] + (float)d[i];                                    \
    out[blockIdx.x * 256 + threadIdx.x] = s + a + b;                                                              \
    if (blockIdx.x == 0 && threadIdx.x == 0) *clk = t1 - t0;                                                      \
  }
BODY(k_fma, "v_fma_f32 %0, %0, %1, %2", "+v"(x[i]) : "v"(a), "v"(b))
BODY(k_max, "v_max_i32 %0, %0, %1", "+v"(x[i]) : "v"(a))
BODY(k_pkmul, "v_pk_mul_f32 %0, %0, %1", "+v"(d[i]) : "v"(d[(i + 1) & 7]))
BODY(k_pkadd, "v_pk_add_f32 %0, %0, %1", "+v"(d[i]) : "v"(d[(i + 1) & 7]))
BODY(k_pkfma, "v_pk_fma_f32 %0, %0, %1, %1", "+v"(d[i]) : "v"(d[(i + 1) & 7]))
BODY(k_cvtpk, "v_cvt_pk_f16_f32 %0, %1, %2", "+v"(x[i]) : "v"(a), "v"(b))
BODY(k_cvtrtz, "v_cvt_pkrtz_f16_f32 %0, %1, %2", "+v"(x[i]) : "v"(a), "v"(b))
BODY(k_cvt32, "v_cvt_f32_f16 %0, %1", "+v"(x[i]) : "v"(a))
BODY(k_mixlo, "v_fma_mixlo_f16 %0, %1, %2, 0", "+v"(x[i]) : "v"(a), "v"(b))
BODY(k_mixhi, "v_fma_mixhi_f16 %0, %1, %2, 0", "+v"(x[i]) : "v"(a), "v"(b))
BODY(k_mix, "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]", "+v"(x[i]) : "v"(a), "v"(b))
BODY(k_pkmaxh, "v_pk_max_f16 %0, %0, %1", "+v"(x[i]) : "v"(a))
BODY(k_pkmulh, "v_pk_mul_f16 %0, %0, %1", "+v"(x[i]) : "v"(a))
BODY(k_pkfmah, "v_pk_fma_f16 %0, %0, %1, %1", "+v"(x[i]) : "v"(a))
BODY(k_accrd, "v_accvgpr_read_b32 %0, %1", "=v"(x[i]) : "a"(y[i]))
BODY(k_accwr, "v_accvgpr_write_b32 %0, %1", "=a"(y[i]) : "v"(x[i]))
BODY(k_mov, "v_mov_b32 %0, %1", "=v"(x[i]) : "v"(a))
BODY(k_andor, "v_and_or_b32 %0, %0, %1, %2", "+v"(x[i]) : "v"(a), "v"(b))
BODY(k_perm, "v_perm_b32 %0, %0, %1, %2", "+v"(x[i]) : "v"(a), "v"(b))
template <class K>
void run(const char* name, K kern) {
  float *out, *in; unsigned long long* clk;
  (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&in, 4096 * 4); (void)hipMemset(in, 0x3c, 4096 * 4); (void)hipMalloc(&clk, 8);
  const int iters = 200;
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, in, 4, clk);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, in, iters, clk);
  (void)hipDeviceSynchronize();
  unsigned long long c; (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  printf("%-28s %.2f shader clocks per instruction\n", name, (double)c / ((double)iters * 4096));
}
int main() {
  run("v_fma_f32", k_fma); run("v_max_i32", k_max); run("v_pk_mul_f32", k_pkmul); run("v_pk_add_f32", k_pkadd);
  run("v_pk_fma_f32", k_pkfma); run("v_cvt_pk_f16_f32", k_cvtpk); run("v_cvt_pkrtz_f16_f32", k_cvtrtz);
  run("v_cvt_f32_f16", k_cvt32); run("v_fma_mixlo_f16", k_mixlo); run("v_fma_mixhi_f16", k_mixhi); run("v_fma_mix_f32", k_mix);
  run("v_pk_max_f16", k_pkmaxh); run("v_pk_mul_f16", k_pkmulh); run("v_pk_fma_f16", k_pkfmah);
  run("v_accvgpr_read_b32", k_accrd); run("v_accvgpr_write_b32", k_accwr); run("v_mov_b32", k_mov);
  run("v_and_or_b32", k_andor); run("v_perm_b32", k_perm);
  return 0;
}
