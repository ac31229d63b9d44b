// Does a partial s_waitcnt lgkmcnt(N) guarantee the OLDER ds_read has landed when reads hit different LDS regions
// (below / above 64 KB) or differ in bank-conflict cost?  Prints stale counts per combination.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out, int offA, int strideA, int offB, int strideB, int iters) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < 40960; i += blockDim.x) lds[i] = i;     // 160 KB
  __syncthreads();
  const int lane = threadIdx.x & 63;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    unsigned addrA = offA + lane * strideA + (it & 3) * 16, addrB = offB + lane * strideB;
    unsigned c, bx;
    asm volatile(
        "v_mov_b32 v20, 0xdeadbeef\n"
        "s_nop 4\n"
        "ds_read_b128 v[20:23], %2\n"
        "ds_read_b128 v[24:27], %3\n"
        "s_waitcnt lgkmcnt(1)\n"
        "v_mov_b32 %0, v20\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_mov_b32 %1, v24\n"
        : "=v"(c), "=v"(bx)
        : "v"(addrA), "v"(addrB)
        : "memory", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    if (c != addrA / 4) ++bad;
    if (bx != addrB / 4) bad += 1000;
  }
  out[threadIdx.x] = bad;
}
int main() {
  unsigned* d;
  hipMalloc(&d, 256 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  struct { const char* n; int oa, sa, ob, sb; } cs[] = {
      {"lo(conflict) then lo", 0, 512, 32768, 16},   {"hi(conflict) then lo", 131072, 128, 0, 16},
      {"lo(conflict) then hi", 0, 512, 131072, 16},  {"hi then hi", 131072, 16, 140000 / 16 * 16, 16},
      {"hi(taps-like) then lo", 131072 + 16, 32, 81920, 16}, {"lo then hi(taps-like)", 81920, 16, 131072 + 16, 32}};
  for (auto& c : cs) {
    k<<<1, 256, 163840>>>(d, c.oa, c.sa, c.ob, c.sb, 2000);
    unsigned h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long tot = 0;
    for (unsigned v : h) tot += v;
    printf("%-26s stale/bad = %llu  (%s)\n", c.n, tot, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
