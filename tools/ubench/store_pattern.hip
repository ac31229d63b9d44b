// Micro-benchmark (round 5, for the stores of k_train_fwd_pre): what one 1 KB global_store_dwordx4 costs on the CU's vector-memory path as a
// function of how many cache lines it covers.  A persistent workgroup per CU (4 waves) writes "saved activation" tiles of 64 rows x 512 floats
// in one of three lane -> address maps, all 16 bytes per lane:
//   PAT 0  the accumulator layout of save_block (mlp_h3n.hip): lane (q = lane / 16, n = lane % 16) writes row n, bytes [64 mo + 16 q, +16) of
//          the wave's 512-byte slice: 16 rows x 64 bytes per instruction = 16 cache lines
//   PAT 1  two rows x 512 contiguous bytes per instruction (lane / 32 = row, lane % 32 = 16-byte chunk): 8 cache lines
//   PAT 2  one fully contiguous KB per instruction (a tile-blocked layout): 8 cache lines, one row-block
// Every wave writes the same number of bytes in all patterns (8 instructions per 16-row group and wave slice).  Reports GB/s and shader
// clocks per store instruction.    hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o bin/store_pattern && bin/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(256, 1) void k_store(float* __restrict__ dst, long long rows, int iters, unsigned long long* clk) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long n_tiles = rows / 64;
  f32x4 v = {(float)lane, (float)wave, 1.0f, 2.0f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it)
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
      float* base = dst + (size_t)tile * 64 * 512;                    // 64 rows x 512 floats
#pragma unroll
      for (int g = 0; g < 4; ++g) {                                    // four 16-row groups
#pragma unroll
        for (int mo = 0; mo < 8; ++mo) {                               // eight stores per group: the wave's 16 rows x 128 features
          float* p;
          if (PAT == 0) p = base + (size_t)(16 * g + (lane & 15)) * 512 + 128 * wave + 16 * mo + 4 * (lane >> 4);
          else if (PAT == 1) p = base + (size_t)(16 * g + 2 * mo + (lane >> 5)) * 512 + 128 * wave + 4 * (lane & 31);
          else p = base + (size_t)(16 * g) * 512 + (size_t)wave * (16 * 128) + 256 * mo + 4 * lane;      // tile-blocked: 1 KB contiguous
          v[0] += 1.0f;
          *reinterpret_cast<f32x4*>(p) = v;
        }
      }
    }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

// Part 2: the same stores next to matrix work, at the per-view training kernel's ratio (32 stores of 1 KB per wave for ~49 k clocks of MFMAs):
// SPREAD = 0: the 32 stores in one burst in front of the MFMAs (save_block); 1: one store every 96 MFMAs; 2: no stores (the floor).
template <int SPREAD>
__global__ __launch_bounds__(256, 1) void k_mix(float* __restrict__ dst, long long rows, const float* __restrict__ in, float* __restrict__ out,
                                                unsigned long long* clk) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long n_tiles = rows / 64;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[lane + i]; b[i] = (_Float16)in[lane + 8 + i]; }
  f32x4 v = {(float)lane, (float)wave, 1.0f, 2.0f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    float* base = dst + (size_t)tile * 64 * 512;
    auto store = [&](int j) {                                          // store j of the tile's 32 (pattern 0)
      const int g = j >> 3, mo = j & 7;
      v[0] += 1.0f;
      *reinterpret_cast<f32x4*>(base + (size_t)(16 * g + (lane & 15)) * 512 + 128 * wave + 16 * mo + 4 * (lane >> 4)) = v;
    };
    if (SPREAD == 0) {
#pragma unroll
      for (int j = 0; j < 32; ++j) store(j);
    }
#pragma unroll 1
    for (int j = 0; j < 32; ++j) {
      if (SPREAD == 1) store(j);
#pragma unroll
      for (int r = 0; r < 96; ++r) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[r & 7]) : "v"(a), "v"(b));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
  f32x4 sum = acc[0];
  for (int i = 1; i < 8; ++i) sum += acc[i];
  if (sum[0] == 12345.678f) out[threadIdx.x] = sum[1];                // (keeps the accumulators alive)
}

template <int SPREAD>
static void run_mix(float* d, long long rows, int cus, unsigned long long* dclk, const float* din, float* dout) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_mix<SPREAD>, dim3(cus), dim3(256), 0, 0, d, rows / 4, din, dout, dclk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_mix<SPREAD>, dim3(cus), dim3(256), 0, 0, d, rows, din, dout, dclk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_ms = (double)(rows / 64) / cus * 32 * 96 * 16 / 2.1e6;      // at 2.1 GHz, 16 clocks per MFMA
  printf("mix, %s: %.3f ms (the MFMAs alone at 2.1 GHz: %.3f ms)\n", SPREAD == 0 ? "32-store burst in front of 3072 MFMAs" : SPREAD == 1 ? "one store every 96 MFMAs" : "no stores", ms, mfma_ms);
}

// Part 3: as part 2, but the MFMAs CONSUME A STREAM OF LOADS (a weight ring: every 12 MFMAs use a 1 KB fragment requested two steps earlier from
// a 3 MB L2-resident buffer) -- the situation of the per-view kernel: loads queue behind the stores on the in-order vector-memory path.
template <int SPREAD>
__global__ __launch_bounds__(256, 1) void k_ring(float* __restrict__ dst, long long rows, const float* __restrict__ wbuf, float* __restrict__ out,
                                                 unsigned long long* clk) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long n_tiles = rows / 64;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const h8* wp = reinterpret_cast<const h8*>(wbuf) + (size_t)wave * 768 * 64 + lane;      // 768 fragments of 1 KB per wave: 3 MB per workgroup
  h8 b;
  for (int i = 0; i < 8; ++i) b[i] = (_Float16)1.0f;
  f32x4 v = {(float)lane, (float)wave, 1.0f, 2.0f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (SPREAD == 3) {                                                   // bursts, the workgroups started an eighth of a tile apart (8 phase groups)
#pragma unroll 1
    for (int k = 0; k < (int)((blockIdx.x >> 3) & 7) * 32; ++k)
#pragma unroll
      for (int r = 0; r < 12; ++r) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[r & 7]) : "v"(b), "v"(b));
  }
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    float* base = dst + (size_t)tile * 64 * 512;
    auto store = [&](int j) {
      const int g = j >> 3, mo = j & 7;
      v[0] += 1.0f;
      if (SPREAD == 4) *reinterpret_cast<f32x4*>(base + (size_t)(16 * g) * 512 + (size_t)wave * (16 * 128) + 256 * mo + 4 * lane) = v;      // 1 KB contiguous
      else *reinterpret_cast<f32x4*>(base + (size_t)(16 * g + (lane & 15)) * 512 + 128 * wave + 16 * mo + 4 * (lane >> 4)) = v;
    };
    if (SPREAD == 0 || SPREAD == 3 || SPREAD == 4) {
#pragma unroll
      for (int j = 0; j < 32; ++j) store(j);
    }
    h8 w0 = wp[0], w1 = wp[64], w2;
#pragma unroll 1
    for (int st = 0; st < 255; st += 3) {                              // 255 steps of 12 MFMAs, the fragment of step st + 2 requested in step st
      w2 = wp[(size_t)((st + 2) % 768) * 64];
      if (SPREAD == 1 && (st % 24) == 0) store(st / 8);
#pragma unroll
      for (int r = 0; r < 12; ++r) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[r & 7]) : "v"(w0), "v"(b));
      w0 = wp[(size_t)((st + 3) % 768) * 64];
      if (SPREAD == 1 && (st % 24) == 8 * 1 + 1) store(st / 8);
#pragma unroll
      for (int r = 0; r < 12; ++r) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[r & 7]) : "v"(w1), "v"(b));
      w1 = wp[(size_t)((st + 4) % 768) * 64];
      if (SPREAD == 1 && (st % 24) == 8 * 2 + 2) store(st / 8);
#pragma unroll
      for (int r = 0; r < 12; ++r) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[r & 7]) : "v"(w2), "v"(b));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
  f32x4 sum = acc[0];
  for (int i = 1; i < 8; ++i) sum += acc[i];
  if (sum[0] == 12345.678f) out[threadIdx.x] = sum[1];
}

// Part 4: the stores of the four computing waves issued by a FIFTH wave (its vector-memory queue holds nothing else; the hand-over of the data
// through LDS is not modelled), one barrier per tile; the computing waves as in part 3 without stores.
__global__ __launch_bounds__(320, 1) void k_ring_store_wave(float* __restrict__ dst, long long rows, const float* __restrict__ wbuf, float* __restrict__ out) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long n_tiles = rows / 64;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const h8* wp = reinterpret_cast<const h8*>(wbuf) + (size_t)(wave & 3) * 768 * 64 + lane;
  h8 b;
  for (int i = 0; i < 8; ++i) b[i] = (_Float16)1.0f;
  f32x4 v = {(float)lane, (float)wave, 1.0f, 2.0f};
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    float* base = dst + (size_t)tile * 64 * 512;
    __syncthreads();
    if (wave == 4) {
#pragma unroll 4
      for (int j = 0; j < 128; ++j) {                                  // the four waves' 32 stores each, two rows x 512 bytes per instruction
        v[0] += 1.0f;
        *reinterpret_cast<f32x4*>(base + (size_t)(2 * (j >> 2) + (lane >> 5)) * 512 + 128 * (j & 3) + 4 * (lane & 31)) = v;
      }
      continue;
    }
    h8 w0 = wp[0], w1 = wp[64], w2;
#pragma unroll 1
    for (int st = 0; st < 255; st += 3) {
      w2 = wp[(size_t)((st + 2) % 768) * 64];
#pragma unroll
      for (int r = 0; r < 12; ++r) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[r & 7]) : "v"(w0), "v"(b));
      w0 = wp[(size_t)((st + 3) % 768) * 64];
#pragma unroll
      for (int r = 0; r < 12; ++r) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[r & 7]) : "v"(w1), "v"(b));
      w1 = wp[(size_t)((st + 4) % 768) * 64];
#pragma unroll
      for (int r = 0; r < 12; ++r) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[r & 7]) : "v"(w2), "v"(b));
    }
  }
  f32x4 sum = acc[0];
  for (int i = 1; i < 8; ++i) sum += acc[i];
  if (sum[0] == 12345.678f) out[threadIdx.x] = sum[1];
}
static void run_store_wave(float* d, long long rows, int cus, const float* wbuf, float* dout) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_ring_store_wave, dim3(cus), dim3(320), 0, 0, d, rows / 4, wbuf, dout);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_ring_store_wave, dim3(cus), dim3(320), 0, 0, d, rows, wbuf, dout);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("ring, the stores from a fifth wave (one barrier per tile): %.3f ms\n", ms);
}

template <int SPREAD>
static void run_ring(float* d, long long rows, int cus, unsigned long long* dclk, const float* wbuf, float* dout) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_ring<SPREAD>, dim3(cus), dim3(256), 0, 0, d, rows / 4, wbuf, dout, dclk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_ring<SPREAD>, dim3(cus), dim3(256), 0, 0, d, rows, wbuf, dout, dclk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("ring, %s: %.3f ms\n", SPREAD == 0 ? "32-store burst in front of 255 x 12 MFMAs fed by loads" : SPREAD == 1 ? "the stores spread, one every ~8 steps" : SPREAD == 3 ? "bursts, workgroups in 8 phase groups an eighth of a tile apart" : SPREAD == 4 ? "32-store burst, every store one contiguous KB (tile-blocked layout)" : "no stores", ms);
}

template <int PAT>
static void run(float* d, long long rows, int cus, unsigned long long* dclk) {
  const int iters = 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_store<PAT>, dim3(cus), dim3(256), 0, 0, d, rows, 1, dclk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_store<PAT>, dim3(cus), dim3(256), 0, 0, d, rows, iters, dclk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[1024];
  hipMemcpy(h, dclk, cus * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < cus; ++i) mean += (double)h[i];
  mean /= cus;
  const double bytes = (double)rows * 2048.0 * iters;
  const double stores_per_wave = (double)(rows / 64) / cus * 32.0 * iters;
  printf("pattern %d: %.3f ms, %.0f GB/s, %.0f shader clocks per store instruction and wave (%.0f per CU-store with 4 waves in flight)\n", PAT, ms,
         bytes / ms * 1e-6, mean / stores_per_wave, mean / stores_per_wave / 4.0);
}

int main() {
  int cus = 256;
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  cus = pr.multiProcessorCount;
  const long long rows = 655360;                                      // one saved tensor of the shipped training step: 1.34 GB
  float* d;
  unsigned long long* dclk;
  hipMalloc(&d, (size_t)rows * 2048);
  hipMalloc(&dclk, 1024 * sizeof(unsigned long long));
  printf("%d CUs, %lld rows x 2 KB per pass, 4 passes\n", cus, rows);
  run<0>(d, rows, cus, dclk);
  run<1>(d, rows, cus, dclk);
  run<2>(d, rows, cus, dclk);
  run<0>(d, rows, cus, dclk);
  printf("the same stores from 32 and from 8 workgroups only (is ~90 clocks per store the chip's write rate or the CU's?):\n");
  run<0>(d, rows / 8, 32, dclk);
  run<0>(d, rows / 32, 8, dclk);
  run<2>(d, rows / 32, 8, dclk);
  run<1>(d, rows / 32, 8, dclk);
  float *din, *dout;
  hipMalloc(&din, 4096); hipMalloc(&dout, 4096);
  hipMemset(din, 0, 4096);
  run_mix<2>(d, rows, cus, dclk, din, dout);
  run_mix<0>(d, rows, cus, dclk, din, dout);
  run_mix<1>(d, rows, cus, dclk, din, dout);
  run_mix<0>(d, rows, cus, dclk, din, dout);
  run_mix<1>(d, rows, cus, dclk, din, dout);
  float* wbuf;
  hipMalloc(&wbuf, (size_t)4 * 768 * 1024);
  hipMemset(wbuf, 0, (size_t)4 * 768 * 1024);
  run_ring<2>(d, rows, cus, dclk, wbuf, dout);
  run_ring<0>(d, rows, cus, dclk, wbuf, dout);
  run_ring<1>(d, rows, cus, dclk, wbuf, dout);
  run_ring<0>(d, rows, cus, dclk, wbuf, dout);
  run_ring<1>(d, rows, cus, dclk, wbuf, dout);
  run_ring<3>(d, rows, cus, dclk, wbuf, dout);
  run_ring<4>(d, rows, cus, dclk, wbuf, dout);
  run_ring<4>(d, rows, cus, dclk, wbuf, dout);
  run_store_wave(d, rows, cus, wbuf, dout);
  run_store_wave(d, rows, cus, wbuf, dout);
  run_ring<2>(d, rows, cus, dclk, wbuf, dout);
  return 0;
}
