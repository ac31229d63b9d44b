// Does s_waitcnt vmcnt(1) guarantee the OLDER global load (cold, HBM miss) has landed when a younger one hits in L2/L1?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* __restrict__ cold, const unsigned* __restrict__ hot, unsigned* out, int iters, size_t cold_n) {
  const int lane = threadIdx.x & 63;
  unsigned bad = 0;
  size_t idx = ((size_t)blockIdx.x * 9973 + threadIdx.x * 4099) % (cold_n / 4) * 4;
  for (int it = 0; it < iters; ++it) {
    const unsigned* pa = cold + idx;
    const unsigned* pb = hot + lane * 4;
    unsigned c, bx;
    asm volatile(
        "v_mov_b32 v20, 0xdeadbeef\n"
        "s_nop 4\n"
        "global_load_dwordx4 v[20:23], %2, off\n"
        "global_load_dwordx4 v[24:27], %3, off\n"
        "s_waitcnt vmcnt(1)\n"
        "v_mov_b32 %0, v20\n"
        "s_waitcnt vmcnt(0)\n"
        "v_mov_b32 %1, v24\n"
        : "=v"(c), "=v"(bx)
        : "v"(pa), "v"(pb)
        : "memory", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    if (c != (unsigned)idx) ++bad;
    if (bx != (unsigned)(lane * 4)) bad += 100000;
    idx = (idx * 1664525 + 1013904223 + c) % (cold_n / 4) * 4;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = bad;
}
int main() {
  const size_t cold_n = (size_t)1 << 30;   // 4 GB of unsigned: far beyond L2 + MALL
  unsigned *cold, *hot, *out;
  hipMalloc(&cold, cold_n * 4);
  hipMalloc(&hot, 4096);
  hipMalloc(&out, 1024 * 256 * 4);
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = i;
  hipMemcpy(hot, h.data(), 4096, hipMemcpyHostToDevice);
  // cold[i] = i (low 32 bits)
  auto fill = [] __device__(int) {};
  (void)fill;
  {
    std::vector<unsigned> chunk(1 << 24);
    for (size_t base = 0; base < cold_n; base += chunk.size()) {
      for (size_t i = 0; i < chunk.size(); ++i) chunk[i] = (unsigned)(base + i);
      hipMemcpy(cold + base, chunk.data(), chunk.size() * 4, hipMemcpyHostToDevice);
    }
  }
  k<<<1024, 256>>>(cold, hot, out, 2000, cold_n);
  std::vector<unsigned> r(1024 * 256);
  hipMemcpy(r.data(), out, r.size() * 4, hipMemcpyDeviceToHost);
  unsigned long long tot = 0;
  for (unsigned v : r) tot += v;
  printf("cold-then-hot with vmcnt(1): stale/bad = %llu of %llu (%s)\n", tot, 1024ull * 256 * 2000, hipGetErrorString(hipGetLastError()));
  return 0;
}
