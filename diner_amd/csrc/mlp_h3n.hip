// fp16-operand field kernels (DINER_PRECISION_F16X3: split products, three MFMAs per fp32 product; DINER_PRECISION_F16: plain
// fp16 operands), feature-sliced ("n-split"): wave w owns output features [128 w, 128 w + 128) for ALL 64 columns of the
// workgroup (4 views x 16 points).
//   * every fp32 product a*w is evaluated as a_hi*w_hi + a_hi*w_lo + a_lo*w_hi (a = a_hi + a_lo, w = w_hi + w_lo, fp16 parts)
//     on v_mfma_f32_16x16x32_f16 with fp32 accumulation; the network runs at a power-of-two scale (weights, biases x16,
//     accumulators hold 16x the activations, the fp32 -> (hi, lo) conversion of every B operand folds the exact 1/16 back
//     in) so that the low parts of small weights stay out of the fp16 subnormal range;
//   * weights are wave-private: streamed straight global -> VGPR (16 KB per k32 block per wave, software-prefetched),
//     each A fragment feeds 4 column groups x {hi,lo}: 12 MFMAs per (hi, lo) fragment pair;
//   * activations are exchanged between layers through a 128 KB LDS buffer already in B-operand form (fp16 hi / lo,
//     1/16 scale folded in): every wave converts its 128-feature slice, two barriers per layer;
//   * the view mean is a register sum over the four column groups;
//   * no bias pass on the residual stream in the per-view kernel: the fc_1 biases of blocks 0 and 1 travel in the bias of the next
//     block's projected map (mlp.hip, mlp_pack), block 2's is added by the post kernel to the view mean it takes over.
// Measuring stick: -DDINER_HN_PROF builds book shader clocks per phase of the tile loop (tools/prof_phases.sh); what was learnt
// with it is in profiles/r02_kernel_experiments.md (round 2c) -- in short: no LDS read right in front of its use inside a GEMM,
// no packed-fp32 arithmetic between MFMAs, nothing lane-varying in the front end's control flow.
// (Round 1 also had an LDS-streamed-weights variant of the same arithmetic, mlp_h3.hip; it lost to this one on every
// measurement -- 166.7 k vs 212.5 k rays/s, profiles/r01_v4_* vs r01_v8_* -- and was retired.)
#include <utility>
#include <vector>
#include "field_common.hpp"

namespace diner {
namespace h3n {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr float kScale = 16.0f, kInvScale = 1.0f / 16.0f;
constexpr int kSlice = 8;                 // accumulator row tiles per wave (128 features)
constexpr int kGroups = 4;                // column groups = source views
constexpr int kBHalfs = 16 * kGroups * 2 * 64 * 8;      // B buffer: [t 16][g 4][hl 2][lane 64] h8 = 128 KB
constexpr int kSrcStride = 9;                           // floats per lane in the feature-source exchange (odd: conflict-free reads)
constexpr size_t kTapsBytes = 4 * 16 * 32;              // taps exchange (4 groups x 16 columns x 32 B)
constexpr size_t kFeatTabBytes = 64 * 16;               // input-feature recipe per (q, slot), see FeatRec
constexpr size_t kFeatSrcBytes = 256 * kSrcStride * 4;  // per lane: x_c (3), R d (3), dd, 0
constexpr size_t kLdsBytes = (size_t)kBHalfs * 2 + kTapsBytes + kFeatTabBytes + kFeatSrcBytes;
static_assert(kLdsBytes <= 160 * 1024, "LDS of one CU");
// post kernel: the B buffer + lin_out on the vector ALU: its weights [wave 4][mo 8][q 4][o 4] f32x4 (8 KB) and the per-wave partial
// outputs [wave 4][g 4][pt 16] f32x4 (4 KB)
constexpr size_t kLinOutWBytes = 4 * 8 * 4 * 4 * 16, kLinOutPartBytes = 4 * 4 * 16 * 16;
constexpr size_t kLdsBytesPost = (size_t)kBHalfs * 2 + kLinOutWBytes + kLinOutPartBytes;
#ifndef DINER_HN_LINOUT_VALU
#define DINER_HN_LINOUT_VALU 1
#endif

// LDS operand buffer addressing: a per-lane byte address kept in one register + immediate offsets (the ds offset
// field holds 16 bits, so the 128 KB buffer is reached from two bases 64 KB apart).  The bases are made opaque at
// each use site: otherwise the compiler materialises one address register per fragment, hoists them out of the
// tile loop and spills them.
typedef __attribute__((address_space(3))) char* lds_ptr;
typedef __attribute__((address_space(3))) h8* lds_h8;
constexpr int kChunkBytes = 4 * kGroups * 2 * 1024;      // a chunk = four k32 blocks = 32 fragments of 1 KB
struct LdsB {
  lds_ptr base;            // lane's 16 B slot in fragment 0 of k32 block 0
  __device__ __forceinline__ static LdsB make(h8* b, int lane) { return {(lds_ptr)(reinterpret_cast<char*>(b)) + lane * 16}; }
  __device__ __forceinline__ void opaque() { asm volatile("" : "+v"(base)); }
  // pointer to chunk c (k32 blocks 4c .. 4c+3); c may be a run-time (wave-uniform) value.  Fragments are then reached with
  // immediates (the ds offset field holds 16 bits; a chunk is 32 KB)
  __device__ __forceinline__ lds_ptr chunk(int c) const { return base + c * kChunkBytes; }
  __device__ __forceinline__ static lds_h8 at(lds_ptr cb, int tl, int g, int hl) {       // fragment (block tl of the chunk, g, hl)
    return (lds_h8)(cb + ((tl * kGroups + g) * 2 + hl) * 1024);
  }
};
// Order in which wave w walks the four chunks of a 512-wide contraction: its OWN chunk first (k32 blocks 4w .. 4w+3 are the B
// operands this wave itself publishes -- it can multiply with them straight from its registers, before anybody else has
// published anything), then the others in ascending order.  The weights are packed in that order per wave (k_pack_layer_h3n).
__device__ __forceinline__ int chunk_of(int pos, int wave) { return pos == 0 ? wave : (pos - 1 < wave ? pos - 1 : pos); }

#define DINER_HN_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, ACC, 0, 0, 0)

struct TapRec {            // per column: 4 tap offsets (float4 units into a projected map) + 4 blend weights
  unsigned off[4];
  float w[4];
};

// How lane quarter q makes its slot-th MLP input (input index f = 16 (slot / 4) + 4 q + slot % 4, field_common.hpp input_feature):
// value = sin ? sin_posenc(fma(src, freq, phase)) : src, with src one of the eight per-lane sources {x_c, R d, dd, 0}.
// The records sit in LDS, written once per workgroup; with them the 16 inputs of a lane cost ~30 instructions each instead of the
// ~190 of input_feature's compare / divide / select chains on a lane-varying f (14.7 k of the kernel's 268 k clocks per tile,
// profiles/r02_kernel_experiments.md) -- same operations on the same operands, so the inputs are bit-identical.
struct FeatRec {
  int src;        // index into the lane's source exchange
  float freq;     // freq_factor * 2^k                               (positional_encoding.py:18)
  float phase;    // 0 or fp32(pi / 2)                               (:30)
  int sin;        // 1: encoded, 0: the source itself
};
__device__ __forceinline__ FeatRec feat_recipe(int f, float freq_factor) {
  FeatRec r{7, 0.0f, 0.0f, 0};
  int j = -1;
  if (f < 3) r.src = f;
  else if (f < 39) { j = (f - 3) / 3; r.src = (f - 3) - 3 * j; }
  else if (f < 42) r.src = 3 + (f - 39);
  else if (f == 42) r.src = 6;
  else if (f < kDIn) { j = f - 43; r.src = 6; }
  if (j >= 0) {
    r.freq = __fmul_rn(freq_factor, (float)(1 << (j >> 1)));
    r.phase = (j & 1) ? 1.57079637050628662109375f : 0.0f;
    r.sin = 1;
  }
  return r;
}

// field_common.hpp field_frontend with the table-driven inputs (rays + z or xyz point sources; explicit inputs pass through)
struct MapDims {
  int Wf, Hf, Ws, Hs;      // feature-map and depth-map sizes (SceneDev's, made opaque per tile by the caller)
};
__device__ __forceinline__ void frontend_h3n(const SceneDev& sc, const MapDims& dm, const FieldArgs& a, int v, int q, int lane, long long p,
                                             const FeatRec* __restrict__ tab, float* __restrict__ src, Taps& taps,
                                             float (&feat)[16]) {
  if (a.direct_feat) {
    field_frontend(sc, a, v, q, p, taps, feat);
    return;
  }
  float px, py, pz, dx, dy, dz;
  load_point(a, p, px, py, pz, dx, dy, dz);
  float xc[3], vd[3];
  world_to_cam(sc.R[v], sc.t[v], px, py, pz, xc[0], xc[1], xc[2]);           // pixelnerf.py:91-93
  vd[0] = rot_row(sc.R[v] + 0, dx, dy, dz);                                  // :100
  vd[1] = rot_row(sc.R[v] + 3, dx, dy, dz);
  vd[2] = rot_row(sc.R[v] + 6, dx, dy, dz);
  const float u = project_axis(xc[0], xc[2], sc.focal[v][0], sc.c[v][0], sc.img_w);   // :105-108
  const float w = project_axis(xc[1], xc[2], sc.focal[v][1], sc.c[v][1], sc.img_h);
  const int ix = nearest_border(u, dm.Ws), iy = nearest_border(w, dm.Hs);    // nearest depth tap (:114-116)
  const float dd = __fsub_rn(sc.depth[(size_t)v * dm.Hs * dm.Ws + (size_t)iy * dm.Ws + ix], xc[2]);
  float* mine = src + lane * kSrcStride;             // read back by this lane only: program order, no barrier
  mine[0] = xc[0]; mine[1] = xc[1]; mine[2] = xc[2];
  mine[3] = vd[0]; mine[4] = vd[1]; mine[5] = vd[2];
  mine[6] = dd;    mine[7] = 0.0f;
#pragma unroll
  for (int sl = 0; sl < 16; ++sl) {
    const FeatRec r = tab[q * 16 + sl];
    const float x = mine[r.src];
    const float e = sin_posenc(__fmaf_rn(x, r.freq, r.phase));               // addcmul is fused, positional_encoding.py:46
    feat[sl] = r.sin ? e : x;
  }
  bilinear_taps(dm.Wf, dm.Hf, sc.feature_padding, v, u, w, taps);
}

#ifndef DINER_HN_DYN            // 1: dynamic tile hand-out (atomic counter); 0: static round-robin tile += gridDim.x
#define DINER_HN_DYN 1
#endif
// Tile hand-out.  The workgroups are persistent (one per CU: LDS and registers admit no second one); with a static round-robin the
// launch ends when the slowest CU has done its share.  Here thread 0 asks for the NEXT tile at the top of the current one (an atomic
// on a per-launch counter; the answer is needed ~230 k clocks later) and hands it to the workgroup through LDS at the bottom.
// One queue per XCD (workgroup b runs on XCD b % 8), and a queue holds the tiles of ONE DEPTH RANGE of every ray: the taps of the same
// 16-sample segment of neighbouring rays share texel rows, which then stay in that XCD's L2 (a single global queue was measured at 4x
// the L2 misses and HBM reads of the per-view kernel: 98 KB instead of 24 KB per point).  With S = K / 16 segments per ray the tile of
// (ray, segment) is ray * S + seg.  S = 8 (K = 128): queue = segment = tile % 8.  Other S (round 4; K = 192 has 12 segments, and
// tile % 8 handed an XCD four different segments in turn: L2 hit rate 0.59 instead of 0.95, profiles/r04_cfg5_*): rays are taken
// in groups of R = 8 / gcd(S, 8); the S R (segment, ray phase) slots of a group, ordered segment-major, are dealt to the queues in
// runs of m = S R / 8 -- every queue gets the same number of tiles and at most two neighbouring segments.  Entry e of queue q:
// group e / m, slot q m + e % m, segment = slot / R, ray = group R + slot % R.  (No ray structure -- explicit points, K not a
// multiple of 16 --: S = 8, R = 1, which is tile % 8.)  An XCD that runs dry takes from the others' queues.
struct QueueMap {
  unsigned S, R, m;              // segments per ray (of one pass), rays per group, entries per queue and group
  unsigned per_queue;            // entries per queue (the last group may hold tiles beyond the launch: skipped)
  unsigned passes, per_pass, S_all;   // round 5: the segments of a ray dealt in `passes` passes of S each (S_all = passes * S): entry e of a
                                      // queue belongs to pass e / per_pass -- all rays' near segments first, then the far ones (1: as before)
  __host__ static QueueMap make(long long n_tiles, int K, bool rays) {
    QueueMap q{8, 1, 1, 0, 1, 0, 8};
    // measurement aids (round 5, BASELINE configs[4]: K = 192 -> 12 segments): DINER_QMAP_PASSES = p splits the segments into p passes
    // (12 = 2 x 6: a queue then holds 0.75 segments at a time instead of 1.5), DINER_QMAP_RMUL = k takes k times the rays per group
    static const int env_passes = [] { const char* e = getenv("DINER_QMAP_PASSES"); return e ? atoi(e) : 0; }();
    static const int env_rmul = [] { const char* e = getenv("DINER_QMAP_RMUL"); return e ? atoi(e) : 1; }();
    if (rays && K >= 16 && K % 16 == 0 && K / 16 <= 4096) {
      unsigned S_all = (unsigned)(K / 16);
      // default: more than 8 segments per ray are dealt in passes of the largest power of two <= 8 (and >= 4) that divides them -- K = 192:
      // 3 passes of 4 segments, every queue then holds ONE (segment, ray phase) at a time instead of 1.5 segments: per-view kernel 23.1-23.4
      // -> 22.5-22.8 ms per launch at 1024^2 in f16x3 (passes 2 / 4 / 6 / 12: 22.9-23.2; profiles/r05_qmap_passes_ab.txt); K = 128 (8 segments)
      // stays one pass (2 / 4 / 8 passes measured 0.4-3.7 % slower)
      unsigned def_passes = 1;
      if (S_all > 8)
        for (unsigned sp = 8; sp >= 4; sp >>= 1)
          if (S_all % sp == 0) { def_passes = S_all / sp; break; }
      unsigned passes = env_passes >= 1 && S_all % (unsigned)env_passes == 0 ? (unsigned)env_passes : def_passes;
      q.S_all = S_all;
      q.passes = passes;
      q.S = S_all / passes;
      unsigned g = q.S & (0u - q.S);
      g = g > 8 ? 8 : g;
      q.R = 8 / g * (env_rmul > 1 ? (unsigned)env_rmul : 1u);
      q.m = q.S * q.R / 8;
    }
    const unsigned long long n_rays = ((unsigned long long)n_tiles + q.S_all - 1) / q.S_all;
    q.per_pass = (unsigned)(((n_rays + q.R - 1) / q.R) * q.m);
    q.per_queue = q.per_pass * q.passes;
    return q;
  }
  __device__ __forceinline__ unsigned long long tile(unsigned q, unsigned e) const {
    const unsigned pass = passes > 1 ? e / per_pass : 0;
    e -= pass * per_pass;
    const unsigned group = e / m, slot = q * m + (e - group * m);
    const unsigned seg = slot / R, phase = slot - seg * R;
    return (unsigned long long)(group * R + phase) * S_all + (pass * S + seg);
  }
};
struct Args {
  FieldArgs fa;
  const _Float16* w;       // n-split packed weights: lin_in, then per block b<3: fc_0, fc_1
  const _Float16* w8;      // the same seven layers in the 8-wave kernel's order (k_field_pre_h8; hi plane only), or null
  const _Float16* w8x;     // ... with hi and lo planes (k_field_pre_h8x, the f16x3 arithmetic on eight waves), or null
  const float* b;          // biases x16: lin_in, then per block: fc_0, fc_1  (7 x 512)
  unsigned long long* prof;   // DINER_HN_PROF builds: 32 phase counters (shader clocks summed over waves), else unused
  unsigned* tile_counter;     // 8 counters (one per XCD queue), zeroed per launch: see TileQueue
  QueueMap qmap;              // which tiles a queue holds (h3n_launch_pre fills it in)
};

struct TileQueue {
  unsigned nxt;
  unsigned done;       // queues seen empty (thread 0)
  __device__ __forceinline__ void begin() { done = 0; }
  // thread 0: the next tile of this workgroup's XCD queue (of the others' once it is empty), 0xffffffff when nothing is left
  __device__ __forceinline__ unsigned fetch(unsigned* counters, long long n_tiles, const QueueMap& qm) {
    const unsigned xcd = blockIdx.x & 7;
#pragma nounroll
    for (unsigned s = 0; s < 8; ++s) {
      const unsigned qn = (xcd + s) & 7;
      if (done & (1u << qn)) continue;
      const unsigned taken = gridDim.x > qn ? (gridDim.x - qn + 7) >> 3 : 0;       // entries the workgroups' first requests used up
#pragma nounroll
      for (;;) {
        const unsigned long long e = (unsigned long long)atomicAdd(counters + qn, 1u) + taken;
        if (e >= qm.per_queue) break;
        const unsigned long long cand = qm.tile(qn, (unsigned)e);
        if (cand < (unsigned long long)n_tiles) return (unsigned)cand;
      }
      done |= 1u << qn;
    }
    return 0xffffffffu;
  }
  // the workgroup's first tile: entry blockIdx.x / 8 of queue blockIdx.x % 8 (no atomic), or -- ragged last group -- the next valid one.
  // Thread 0 writes it to *slot; the caller has a barrier in front of the tile loop.
  __device__ __forceinline__ void first(unsigned* counters, long long n_tiles, const QueueMap& qm, unsigned* slot) {
    if (threadIdx.x == 0) {
      unsigned t = 0xffffffffu;
      const unsigned e = blockIdx.x >> 3;
      if (e < qm.per_queue) {
        const unsigned long long cand = qm.tile(blockIdx.x & 7, e);
        if (cand < (unsigned long long)n_tiles) t = (unsigned)cand;
      }
#if DINER_HN_DYN
      if (t == 0xffffffffu) t = fetch(counters, n_tiles, qm);
#else
      t = blockIdx.x < n_tiles ? blockIdx.x : 0xffffffffu;
#endif
      *slot = t;
    }
  }
  __device__ __forceinline__ void request(unsigned* counters, long long n_tiles, const QueueMap& qm) {
#if DINER_HN_DYN
    if (threadIdx.x == 0) nxt = fetch(counters, n_tiles, qm);
#endif
  }
  // the same in two halves around a barrier the caller has anyway: offer() in front of it, take() behind it
  __device__ __forceinline__ void offer(unsigned* slot) {
#if DINER_HN_DYN
    if (threadIdx.x == 0) *slot = nxt;
#endif
  }
  // park(): thread 0 puts the answer into an LDS slot as soon as it has surely arrived (the caller places this behind its first GEMM); the
  // others may read it behind any later barrier, without one of their own.  The caller alternates between two slots: a slot is rewritten
  // two tiles later, behind many barriers.
  __device__ __forceinline__ void park(unsigned* slot) {
#if DINER_HN_DYN
    if (threadIdx.x == 0) *slot = nxt;
#endif
  }
  __device__ __forceinline__ long long initial(const unsigned* slot) {       // what first() left in *slot (behind a barrier)
    const unsigned t = (unsigned)__builtin_amdgcn_readfirstlane((int)*slot);
    return t == 0xffffffffu ? 0x7fffffffffffffffll : (long long)t;
  }
  __device__ __forceinline__ long long take(long long tile, const unsigned* slot) {
#if DINER_HN_DYN
    const unsigned t = (unsigned)__builtin_amdgcn_readfirstlane((int)*slot);
    return t == 0xffffffffu ? 0x7fffffffffffffffll : (long long)t;
#else
    return tile + gridDim.x;
#endif
  }
  __device__ __forceinline__ long long next(long long tile, unsigned* slot) {
#if DINER_HN_DYN
    offer(slot);
    __syncthreads();
    return take(tile, slot);
#else
    return tile + gridDim.x;
#endif
  }
};

// Phase timer of the per-view kernel (DINER_HN_PROF builds only; tools/prof_phases.sh): mark(i) books the shader clocks since the
// previous mark on phase i.  Costs an s_memtime + s_waitcnt lgkmcnt(0) per mark, so the phase sums are slightly pessimistic.
struct Prof {
#ifdef DINER_HN_PROF
  unsigned long long last, acc[20], t0, r0;
  __device__ __forceinline__ void begin() {
#pragma unroll
    for (int i = 0; i < 20; ++i) acc[i] = 0;
    t0 = last = __builtin_readcyclecounter();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  __device__ __forceinline__ void mark(int i) {
    const unsigned long long now = __builtin_readcyclecounter();
    acc[i] += now - last;
    last = now;
  }
  __device__ __forceinline__ void end(unsigned long long* out, int lane) {
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 20; ++i) atomicAdd(out + i, acc[i]);
      atomicAdd(out + 24, __builtin_readcyclecounter() - t0);
      atomicAdd(out + 25, __builtin_amdgcn_s_memrealtime() - r0);
      atomicAdd(out + 26, 1ull);
    }
  }
#else
  __device__ __forceinline__ void begin() {}
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ void end(unsigned long long*, int) {}
#endif
};

// packed weights of one layer with KT k32 blocks: [w 4][t KT][mo 8][hl 2][lane 64][8]
__device__ __forceinline__ const h8* wfrag(const _Float16* layer, int KT, int wave, int t, int mo, int hl, int lane) {
  return reinterpret_cast<const h8*>(layer) + ((((size_t)wave * KT + t) * 8 + mo) * 2 + hl) * 64 + lane;
}

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N-1>)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Side work interleaved with a GEMM, one slice per quarter-step (see gemm): default nothing.
struct NoSide {
  template <int H, int G>
  __device__ __forceinline__ void run() {}
  __device__ __forceinline__ void finish() {}
};

#ifndef DINER_HN_RING
#define DINER_HN_RING 3
#endif
#ifndef DINER_HN_RING0          // A ring of the fc_0 GEMMs (no gather buffers live there)
#define DINER_HN_RING0 DINER_HN_RING
#endif

// acc[mo][g] += W[slice rows][all k] . B[k][cols g]   (B from the LDS exchange buffer, A straight from global).
// Fully unrolled over 2 KT half-steps (k32 block t, row-tile half) of four quarter-steps (column group g): 12 MFMAs
// each (4 row tiles x {hi*hi, lo*hi, hi*lo}), an accumulator revisited 4 MFMAs apart.
//   * A fragments (8 x 1 KB per half-step, wave-private) are requested R-1 half-steps ahead, two per quarter-step,
//     into a ring of R register buffers (an L2 hit takes longer than one half-step's 768 MFMA cycles);
//   * B fragments (hi, lo of one column group, shared) live in one buffer: group g of the next k32 block is re-read
//     right after its last use in the second half (576 MFMA cycles before the next use);
//   * the side task gets a slot per quarter-step, so its VALU / VMEM work is spread between the MFMAs.
// LO = false: plain fp16 operands (hi parts only, one MFMA per product; diner_set_precision(3)).
#ifndef DINER_HN_EARLYA
#define DINER_HN_EARLYA 1
#endif
// The weight ring of one GEMM.  start() issues the first R-1 half-steps; on the per-view kernel's fc_0 GEMMs it is called BEFORE
// the hidden state is published (the barriers in between wait on LDS traffic only, not on vmcnt), so the first fragments arrive
// while the conversion runs -- neutral there (61.58 M vs 61.71 M clocks per wave); in the post kernel the same costs 5 % (the
// loads issued in front of the barrier hold the wave for ~4 k clocks, profiles/r02c_phase_timer_post_kernel.txt), so not there.  (On the fc_1 GEMMs every way of doing the same -- ring started ahead of the gather / no-gather branch, or
// only in the no-gather arm with its own publish -- made the allocator spill 30-130 registers inside the GEMM: not done.)
template <int KT, int R, bool LO>
struct ARing {
  typedef const __attribute__((address_space(1))) char* gptr;      // stays a global (not flat) access through the asm
  h8 a[R][8];                            // half-step ring (static indices after unrolling)
  // scalar base (advanced 8 KB per half-step and kept opaque so the addresses are not all materialised up front)
  // + per-lane 32-bit offset + immediate: no address registers per load
  gptr abase;
  unsigned avoff;
  __device__ __forceinline__ void load_a2(h8 (&dst)[8], int pair) {      // fragments 2 pair, 2 pair + 1 of the half-step at abase
#ifdef DINER_HN_NO_A          // ablation: price the weight stream
    asm volatile("" : "+v"(dst[2 * pair]), "+v"(dst[2 * pair + 1]));
#else
    asm volatile("" : "+s"(abase));
#pragma unroll
    for (int i = 2 * pair; i < 2 * pair + 2; ++i)
      if (LO || (i & 1) == 0) dst[i] = *(const __attribute__((address_space(1))) h8*)(abase + avoff + (i * 1024 - 4096));
    if (pair == 3) abase += 8192;
#endif
  }
  __device__ __forceinline__ void start(const _Float16* __restrict__ layer, int wave, int lane) {
    constexpr int NH = 2 * KT;
    abase = (gptr)(reinterpret_cast<const char*>(layer) + (size_t)wave * KT * 16384 + 4096);
    avoff = lane * 16;
#if defined(DINER_HN_NO_A) || defined(DINER_HN_NO_B)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) a[r][i] = *(reinterpret_cast<const h8*>(layer) + (r * 8 + i) * 64 + lane);
#endif
    static_for<(R - 1 < NH ? R - 1 : NH)>([&](auto H) {
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) load_a2(a[decltype(H)::value], pr);
    });
  }
};

// one IEEE fp32 multiply / add as an opaque single instruction (see GatherSide::blend_step)
__device__ __forceinline__ float mul1(float a, float b) {
  float d;
  asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float mul1s(float a, float s) {      // s: wave-uniform (SGPR or inline constant), no VGPR for it
  float d;
  asm("v_mul_f32 %0, %2, %1" : "=v"(d) : "v"(a), "s"(s));
  return d;
}
__device__ __forceinline__ float add1(float a, float b) {
  float d;
  asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float fma1(float a, float b, float c) {
  float d;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// ---- fp32 accumulator values -> fp16 (hi, lo) B operands, in SINGLE-WIDTH full-rate instructions (tools/ubench/valu_rate: 4.5-4.8 clocks
// each; packed-fp32 arithmetic does not overlap with MFMAs at all and v_fma_mixlo/hi_f16 are half rate):
//   v = max_i32(x, 0) * scale                     relu on the bit pattern, one multiply
//   hi = v_cvt_pk_f16_f32(v0, v1)                 round-to-nearest-even, two values per instruction
//   r  = v_fma_mix_f32(hi.half, -1.0, v)          v - float(hi): exact, the fp16 operand is read in place
//   lo = v_cvt_pk_f16_f32(r0, r1)
// = 4.5 instructions per value with the accumulator read (the compiler's version of the same arithmetic: 5.5).
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {
  unsigned d;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float resid_lo(unsigned h, float v) {       // v - float(low half of h)
  float d;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(v));
  return d;
}
__device__ __forceinline__ float resid_hi(unsigned h, float v) {       // v - float(high half of h)
  float d;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(v));
  return d;
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// four accumulator values (rows 4q .. 4q+3 of one row tile, one column) -> dwords [2 part, 2 part + 1] of the B fragments hi / lo
template <bool LO, int PART>
__device__ __forceinline__ void cvt4(const f32x4& x, float scale, u32x4& h, u32x4& l) {
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // the source stays in the accumulator half of the file: an explicit v_accvgpr_read per value (a plain read lets the register
    // allocator move whole accumulator tuples of the block into VGPRs across the GEMM and spill others to make room)
    int xi;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(xi) : "a"(x[i]));
    v[i] = mul1s(__int_as_float(max(xi, 0)), scale);
  }
  const unsigned h0 = cvt_pk_f16(v[0], v[1]), h1 = cvt_pk_f16(v[2], v[3]);
  h[2 * PART] = h0;
  h[2 * PART + 1] = h1;
  if constexpr (LO) {
    l[2 * PART] = cvt_pk_f16(resid_lo(h0, v[0]), resid_hi(h0, v[1]));
    l[2 * PART + 1] = cvt_pk_f16(resid_lo(h1, v[2]), resid_hi(h1, v[3]));
  }
}

// acc[mo][g] += W[slice rows][all k] . B[k][cols g]   (A straight from global, wave-private; B = the activations in fp16 hi / lo).
// Fully unrolled over 2 KT half-steps (k32 block, row-tile half) of four quarter-steps (column group g): 12 MFMAs
// each (4 row tiles x {hi*hi, lo*hi, hi*lo}), an accumulator revisited 4 MFMAs apart.  A 512-wide contraction is walked in
// the wave's own chunk order (chunk_of): position 0 = the k32 blocks this wave publishes.
//   * A fragments (8 x 1 KB per half-step, wave-private) are requested R-1 half-steps ahead, two per quarter-step,
//     into a ring of R register buffers (an L2 hit takes longer than one half-step's 768 MFMA cycles);
//   * B fragments (hi, lo of one column group, shared through LDS) live in one buffer: group g of the next k32 block is re-read
//     right after its last use in the second half (576 MFMA cycles before the next use);
//   * the side task gets a slot per quarter-step, so its VALU / VMEM work is spread between the MFMAs;
//   * OWN (round 3): the B operands of the wave's own chunk never come back from LDS.  `src` (the accumulator block whose relu
//     is this GEMM's input) is converted by this wave anyway: block 0 in front of the GEMM, block tl + 1 as a side task of block
//     tl's 96 MFMAs -- one f32x4 per quarter-step, written to LDS for the other waves and kept in registers (cv) as this wave's B
//     operands.  A barrier in front of chunk position 1 is the only one the publish needs (everybody's blocks are in LDS then);
//     the conversion of 3/4 of a publish and all of its LDS writes run in the MFMAs' shadow instead of in front of the GEMM.
// LO = false: plain fp16 operands (hi parts only, one MFMA per product; DINER_PRECISION_F16).
template <int KT, int R, bool LO, bool OWN, class Side>
__device__ __forceinline__ void gemm(ARing<KT, R, LO>& ring, LdsB B, int wave, const f32x4 (&src)[kSlice][kGroups], float scale,
                                     f32x4 (&acc)[kSlice][kGroups], Side& side) {
  constexpr int NH = 2 * KT;
  static_assert(!OWN || KT == 16, "own-chunk scheme: 512-wide contractions only");
  h8 bb[kGroups][2];                     // B of the current k32 block: [g][hl]
  B.opaque();
  lds_ptr cbp[4];                        // chunk pointers, made when first needed (wave-uniform offsets on the lane's base)
  auto chunk_ptr = [&](int pos) {
    lds_ptr cb = B.chunk(KT == 16 ? chunk_of(pos, wave) : 0);
    asm volatile("" : "+v"(cb));
    return cb;
  };
  auto load_b = [&](int t, int g) {
#ifdef DINER_HN_NO_B          // ablation: price the LDS operand reads
    asm volatile("" : "+v"(bb[g][0]), "+v"(bb[g][1]));
#else
    bb[g][0] = *LdsB::at(cbp[t >> 2], t & 3, g, 0);
    if (LO) bb[g][1] = *LdsB::at(cbp[t >> 2], t & 3, g, 1);
#endif
  };
  cbp[0] = chunk_ptr(0);
#if defined(DINER_HN_NO_A) || defined(DINER_HN_NO_B)
#pragma unroll
  for (int g = 0; g < kGroups; ++g) bb[g][0] = bb[g][1] = *LdsB::at(cbp[0], 0, g, 0);
#endif
#pragma unroll
  for (int g = 0; g < kGroups; ++g) load_b(0, g);      // (OWN: block 0 of the own chunk, written by this wave just before)
  u32x4 ch, cl;                          // OWN: the fragment pair being converted
  static_for<NH * kGroups>([&](auto Q) {
    constexpr int qi = decltype(Q)::value;
    constexpr int h = qi >> 2, g = qi & 3;
    constexpr int t = h >> 1, half = h & 1;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (h + R - 1 < NH) ring.load_a2(ring.a[(h + R - 1) % R], g);
    if constexpr (OWN && t == 4 && half == 0 && g == 0) __syncthreads();      // every wave's own blocks are in LDS behind this barrier
    if constexpr (half == 1 && g == 1 && (t & 3) == 3 && t + 1 < KT) cbp[(t + 1) >> 2] = chunk_ptr((t + 1) >> 2);
    // previous quarter's group is free (OWN: chunk position 1 belongs to another wave -- not before the barrier)
    if constexpr (half == 1 && g > 0 && t + 1 < KT && !(OWN && t == 3)) load_b(t + 1, g - 1);
    if constexpr (OWN && t == 4 && half == 0 && g == 0) {
#pragma unroll
      for (int g2 = 0; g2 < kGroups; ++g2) load_b(4, g2);
    } else {
      if constexpr (half == 0 && g == 0 && t > 0) load_b(t, kGroups - 1);         // ... and the last one of block t-1
    }
    if constexpr (OWN && t + 1 < 4) {    // convert block t + 1 of the own chunk, one f32x4 per quarter-step: unit u of 8
      constexpr int u = half * 4 + g, gu = u >> 1, part = u & 1;
      cvt4<LO, part>(src[2 * (t + 1) + part][gu], scale, ch, cl);
      if constexpr (part == 1) {         // the pair is complete: to LDS, for the other waves and for this one (in order: no barrier)
        asm volatile("" : "+v"(cbp[0]));
        *LdsB::at(cbp[0], t + 1, gu, 0) = __builtin_bit_cast(h8, ch);
        if constexpr (LO) *LdsB::at(cbp[0], t + 1, gu, 1) = __builtin_bit_cast(h8, cl);
      }
    }
    side.template run<h, g>();
    h8 (&ac)[8] = ring.a[h % R];
    const h8 b0 = bb[g][0], b1 = bb[g][1];
#pragma unroll
    for (int m = 0; m < 4; ++m) DINER_HN_MFMA(acc[4 * half + m][g], ac[2 * m], b0);
    if constexpr (LO) {
#pragma unroll
      for (int m = 0; m < 4; ++m) DINER_HN_MFMA(acc[4 * half + m][g], ac[2 * m + 1], b0);
#pragma unroll
      for (int m = 0; m < 4; ++m) DINER_HN_MFMA(acc[4 * half + m][g], ac[2 * m], b1);
    }
    // Anchor the quarter-step's results here (no code): MFMAs are pure, and without a use in place the optimiser may
    // sink a whole accumulation chain below all of the GEMM's loads (seen in k_field_post_h3n: every fragment spilled).
#pragma unroll
    for (int m = 0; m < 4; ++m) asm volatile("" : "+a"(acc[4 * half + m][g]));
  });
  side.finish();
}
// all B operands from LDS (published before the call): start + run in one go
template <int KT, int R, bool LO, class Side>
__device__ __forceinline__ void gemm(const _Float16* __restrict__ layer, LdsB B, int wave, int lane,
                                     f32x4 (&acc)[kSlice][kGroups], Side& side) {
  ARing<KT, R, LO> ring;
  ring.start(layer, wave, lane);
  gemm<KT, R, LO, false>(ring, B, wave, acc, 0.0f, acc, side);
}

// publish relu(acc) * scale of this wave's 128-feature slice as B operands (its own chunk: k32 blocks 4w .. 4w+3) for all 4 column
// groups, in front of a GEMM (the GEMMs that carry a gather side task, lin_out)
template <bool LO>
__device__ __forceinline__ void publish(LdsB B, int wave, int lane, const f32x4 (&acc)[kSlice][kGroups]) {
  B.opaque();
#ifdef DINER_HN_NO_PUBLISH
  return;
#endif
  lds_ptr cb = B.chunk(wave);
  asm volatile("" : "+v"(cb));
#pragma unroll
  for (int tl = 0; tl < 4; ++tl)
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      u32x4 h, l;
      cvt4<LO, 0>(acc[2 * tl][g], kInvScale, h, l);
      cvt4<LO, 1>(acc[2 * tl + 1][g], kInvScale, h, l);
      *LdsB::at(cb, tl, g, 0) = __builtin_bit_cast(h8, h);
      if constexpr (LO) *LdsB::at(cb, tl, g, 1) = __builtin_bit_cast(h8, l);
    }
}

// relu(src) -> B operands and the GEMM of `layer` on them.  OWN: the own-chunk scheme of gemm (one barrier in front, block 0 of the
// own chunk converted before it so that the wait overlaps with the conversion, one barrier inside the GEMM); otherwise barrier,
// publish, barrier, GEMM.  The weight ring is started first (EARLY) -- the barriers wait on LDS traffic only.
template <int R, bool LO, bool EARLY, bool OWN, bool BIAS_FIRST = true, class Side, class Between>
__device__ __forceinline__ void publish_gemm(const _Float16* __restrict__ layer, LdsB B, int wave, int lane,
                                             const f32x4 (&src)[kSlice][kGroups], f32x4 (&acc)[kSlice][kGroups], Side& side,
                                             Between&& between, Prof& pf, int ph) {
  ARing<16, R, LO> ring;
  if constexpr (EARLY) ring.start(layer, wave, lane);
  if constexpr (OWN) {
    if constexpr (BIAS_FIRST) between();          // bias of the accumulators the GEMM adds into (its loads fly during the conversion)
    u32x4 c0[kGroups][2];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {           // block 0 of the own chunk, converted while the others finish the previous GEMM
      cvt4<LO, 0>(src[0][g], kInvScale, c0[g][0], c0[g][1]);
      cvt4<LO, 1>(src[1][g], kInvScale, c0[g][0], c0[g][1]);
    }
    pf.mark(ph);
    __syncthreads();                              // everybody finished reading the previous B
    pf.mark(ph + 1);
    {
      B.opaque();
      lds_ptr cb = B.chunk(wave);
      asm volatile("" : "+v"(cb));
#pragma unroll
      for (int g = 0; g < kGroups; ++g) {
        *LdsB::at(cb, 0, g, 0) = __builtin_bit_cast(h8, c0[g][0]);
        if constexpr (LO) *LdsB::at(cb, 0, g, 1) = __builtin_bit_cast(h8, c0[g][1]);
      }
    }
    if constexpr (!BIAS_FIRST) between();
    if constexpr (!EARLY) ring.start(layer, wave, lane);
    pf.mark(ph + 2);
    gemm<16, R, LO, true>(ring, B, wave, src, kInvScale, acc, side);
    pf.mark(ph + 3);
  } else {
    __syncthreads();                              // everybody finished reading the previous B
    pf.mark(ph);
    publish<LO>(B, wave, lane, src);
    pf.mark(ph + 1);
    __syncthreads();
    pf.mark(ph + 2);
    between();                                    // bias of the accumulators the GEMM adds into
    if constexpr (!EARLY) ring.start(layer, wave, lane);
    gemm<16, R, LO, false>(ring, B, wave, src, kInvScale, acc, side);
    pf.mark(ph + 3);
  }
}

// Tell the register allocator that a block of accumulators lives in the AGPR half of the file at this point (no code).
__device__ __forceinline__ void pin_acc(f32x4 (&acc)[kSlice][kGroups]) {
#pragma unroll
  for (int mo = 0; mo < kSlice; ++mo)
#pragma unroll
    for (int g = 0; g < kGroups; ++g) asm volatile("" : "+a"(acc[mo][g]));
}

__device__ __forceinline__ void set_bias(f32x4 (&acc)[kSlice][kGroups], const float* __restrict__ bias, int wave, int q) {
#pragma unroll
  for (int mo = 0; mo < kSlice; ++mo) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 128 * wave + 16 * mo + 4 * q);
#pragma unroll
    for (int g = 0; g < kGroups; ++g) acc[mo][g] = bv;
  }
}
__device__ __forceinline__ void add_bias(f32x4 (&acc)[kSlice][kGroups], const float* __restrict__ bias, int wave, int q) {
#pragma unroll
  for (int mo = 0; mo < kSlice; ++mo) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 128 * wave + 16 * mo + 4 * q);
#pragma unroll
    for (int g = 0; g < kGroups; ++g) acc[mo][g] += bv;
  }
}

#ifndef DINER_HN_OWN            // 1: own-chunk scheme on every GEMM without a gather side task (see gemm); 0: exposed publishes
#define DINER_HN_OWN 1
#endif
#ifndef DINER_HN_EARLY1         // weight ring of the fc_1 GEMMs started in front of the conversion as well
#define DINER_HN_EARLY1 0
#endif
#ifndef DINER_HN_EARLYP         // ... and in the post kernel
#define DINER_HN_EARLYP 0
#endif
#ifndef DINER_HN_OWNG           // the same on the two GEMMs that carry the gather side task
#define DINER_HN_OWNG 1
#endif
#ifndef DINER_HN_GDEPTH
#define DINER_HN_GDEPTH 2
#endif
#ifndef DINER_HN_G0DEPTH        // units in flight for block 0's stand-alone gather (no GEMM buffers live there)
#define DINER_HN_G0DEPTH 8
#endif
// Plain-fp16 instances (LO = false): a half-step is 4 x 4 MFMAs = 256 clocks instead of 768, so the same prefetch distances in half-steps
// cover a third of the latency, and the lo planes' registers (48 of the weight ring, 16 of the B buffer) are free: deeper rings there.
#ifndef DINER_HN_RING_F16        // measured (profiles/r04_ab_runs.txt): ring 4 / gather depth 3 is +6.7 % at 800x600, +3.8 % at 1024^2 K=192; 6 / 4 the same, 8 / 4 spills
#define DINER_HN_RING_F16 4
#endif
#ifndef DINER_HN_RING0_F16
#define DINER_HN_RING0_F16 DINER_HN_RING_F16
#endif
#ifndef DINER_HN_GDEPTH_F16
#define DINER_HN_GDEPTH_F16 3
#endif
#ifndef DINER_HN_GDEPTH_H        // GatherSideH (fp16 maps): units between request and blend as a GEMM side task / stand-alone
#define DINER_HN_GDEPTH_H 2
#endif
#ifndef DINER_HN_G0DEPTH_H
#define DINER_HN_G0DEPTH_H 4
#endif

// xs[mo][g] += 16 * interp(lin_z[b](latent)) for this wave's feature slice and all four column groups: 32 units
// (g, mo) of 4 taps each.  As a GEMM side task (SIDE) one unit's taps are requested per half-step, one per quarter-step, and
// blended / added GD - 1 half-steps later, again one tap per quarter-step (the additions commute with the GEMM's accumulation
// into the same registers).  What the side task must not do inside a GEMM (profiles/r02_kernel_experiments.md, round 2c):
//   * read LDS right in front of a use (a tap offset per load, the weights per blend): the wave then parks for the LDS latency
//     with one MFMA in flight, ~100 clocks in every quarter-step -- the column's tap rows and weights of a view are read once,
//     one unit before the view's first unit, and kept in registers (prefetch);
//   * packed-fp32 arithmetic: v_pk_mul_f32 / v_pk_add_f32 (what the compiler makes of float4 expressions) do not overlap with
//     the wave's MFMAs, each costs a whole MFMA slot (+16 clocks, tools/ubench/mfma_valu.hip) -- the blend is written in
//     v_mul_f32 / v_add_f32, which issue in the MFMAs' shadow (two per MFMA are free).
template <int GD, bool SIDE = true>
struct GatherSide {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const float* __restrict__ tz;
  const TapRec* __restrict__ taps_lds;     // [g 4][col 16]
  int wave, q, pt;
  f32x4 (&xs)[kSlice][kGroups];
  f32x4 r[GD][4];
  u32x4 off4[kGroups];                     // this column's tap rows ...
  f32x4 w4[kGroups];                       // ... and blend weights per view
  f32x4 bw, bv;                            // weights and running sum of the unit being blended

  __device__ __forceinline__ GatherSide(const float* tz_, const TapRec* taps_lds_, int wave_, int q_, int pt_,
                                        f32x4 (&xs_)[kSlice][kGroups])
      : tz(tz_), taps_lds(taps_lds_), wave(wave_), q(q_), pt(pt_), xs(xs_) {
    prefetch<0>();
  }
  template <int g>
  __device__ __forceinline__ void prefetch() {
    off4[g] = *reinterpret_cast<const u32x4*>(taps_lds[g * 16 + pt].off);
    w4[g] = *reinterpret_cast<const f32x4*>(taps_lds[g * 16 + pt].w) * kScale;      // the accumulators hold 16 x the activations
  }
  template <int U, int KTAP>       // one of the unit's four taps (the GEMM side task spreads them over the quarter-steps)
  __device__ __forceinline__ void issue_tap() {
    constexpr int g = U >> 3, mo = U & 7;
#ifdef DINER_HN_G_NOLOAD        // ablation: the side task without its loads
    asm volatile("" : "+v"(r[U % GD][KTAP]));
#else
    const char* base = reinterpret_cast<const char*>(tz);          // scalar base + 32-bit lane offset + immediate
    const unsigned lane_off = (32 * wave + q) * 16;
    r[U % GD][KTAP] = *reinterpret_cast<const f32x4*>(base + (off4[g][KTAP] * 2048u + lane_off) + mo * 64);
#endif
  }
  // tap K's share of the blend sum_k t_k (16 w_k)
  template <int U, int K>
  __device__ __forceinline__ void blend_step() {
    constexpr int g = U >> 3, mo = U & 7;
    const f32x4 (&t)[4] = r[U % GD];
#ifdef DINER_HN_G_NOBLEND       // ablation: the loads without the arithmetic
    asm volatile("" :: "v"(t[K]));
    return;
#endif
    if constexpr (K == 0) {
      bw = w4[g];
      // Keep w an opaque register value.  Without this the hipcc 7.2 build of an earlier version of this kernel returned
      // wrong sums when the blend read its weights from LDS inside the GEMM (standalone it was fine;
      // -amdgpu-waitcnt-forcezero, an extra s_waitcnt or this empty asm all cured it; LDS / VMEM return order checked in
      // tools/ubench/{lds,vm}_order; see DESIGN.md "A hazard worth recording").
      asm volatile("" : "+v"(bw));
    }
    // sum_k t_k (16 w_k) as one multiply and three fused multiply-adds per value (the exact-fp32 kernels keep the reference's
    // separate roundings; this arithmetic mode is within 1e-6 of them either way)
    if constexpr (SIDE) {
#pragma unroll
      for (int i = 0; i < 4; ++i) bv[i] = K == 0 ? mul1(t[0][i], bw[0]) : fma1(t[K][i], bw[K], bv[i]);
      if constexpr (K == 3) {
        asm volatile("" : "+a"(xs[mo][g]));      // keep the accumulator file assignment: read, add, write back
        f32x4 acc = xs[mo][g];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = add1(acc[i], bv[i]);
        xs[mo][g] = acc;
        asm volatile("" : "+a"(xs[mo][g]));
      }
    } else {
      if constexpr (K == 0) bv = t[0] * bw[0];
      else bv = __builtin_elementwise_fma(t[K], (f32x4){bw[K], bw[K], bw[K], bw[K]}, bv);
      if constexpr (K == 3) {
        asm volatile("" : "+a"(xs[mo][g]));
        xs[mo][g] += bv;
        asm volatile("" : "+a"(xs[mo][g]));
      }
    }
  }
  // half-step H: unit H's taps are requested and unit V = H - GD + 1 is blended, tap G of either in quarter-step G
  template <int H, int G>
  __device__ __forceinline__ void run() {
#ifndef DINER_HN_NO_GATHER
    static_assert(GD >= 2, "a tap is blended at least one half-step after it was requested");
    constexpr int V = H - GD + 1;
    if constexpr (V >= 0 && V < 32) blend_step<(V >= 0 && V < 32 ? V : 0), G>();
    if constexpr (H < 32) issue_tap<(H < 32 ? H : 0), G>();
    if constexpr (G == 1 && (H & 7) == 7 && H < 31) prefetch<(H < 31 ? (H + 1) >> 3 : 0)>();
#endif
  }
  __device__ __forceinline__ void finish() {
#ifndef DINER_HN_NO_GATHER
    static_for<GD - 1>([&](auto I) {
      constexpr int V = 33 - GD + decltype(I)::value;
      blend_step<V, 0>();
      blend_step<V, 1>();
      blend_step<V, 2>();
      blend_step<V, 3>();
    });
#endif
  }
  // stand-alone (no GEMM to hide under): block 0
  __device__ __forceinline__ void all() {
    static_for<32>([&](auto H) {
      run<decltype(H)::value, 0>();
      run<decltype(H)::value, 1>();
      run<decltype(H)::value, 2>();
      run<decltype(H)::value, 3>();
    });
    finish();
  }
  // the same in two parts (experiment DINER_HN_G0EARLY, round 4): head() = the first GD - 1 units' requests only (no blend reads xs yet),
  // issued in front of the lin_in GEMM; tail() = the rest behind it
  __device__ __forceinline__ void head() {
    static_for<GD - 1>([&](auto H) {
      run<decltype(H)::value, 0>();
      run<decltype(H)::value, 1>();
      run<decltype(H)::value, 2>();
      run<decltype(H)::value, 3>();
    });
  }
  __device__ __forceinline__ void tail() {
    static_for<32 - (GD - 1)>([&](auto I) {
      constexpr int H = GD - 1 + decltype(I)::value;
      run<H, 0>();
      run<H, 1>();
      run<H, 2>();
      run<H, 3>();
    });
    finish();
  }
};

// The same side task for the plain-fp16 instances (round 4): the taps come from the FP16 copy of the projected maps
// (DinerScene.latent_proj_f16, written by k_proj_to_f16 in mlp.hip), whose 512 channels are stored in the order this kernel consumes
// them -- position 128 w + 32 mp + 8 q + 4 (mo & 1) + i holds channel 128 w + 16 mo + 4 q + i (mp = mo / 2) -- so that ONE 16-byte load
// per lane and tap carries the lane's four rows of TWO row tiles: half the load instructions, half the bytes through the vector-memory
// path (which, not the matrix pipe, bounds these instances: at one MFMA per product the weight stream alone needs the path's 64 B/clk,
// profiles/r04_cfg5_f16_*).  16 units (g, mp) of 4 taps; unit U is requested during half-step 2 U (tap G in quarter-step G) and
// blended D units later, row tile 2 mp during the even half-step, 2 mp + 1 during the odd one: the blend is v_fma_mix_f32 (fp16 tap x
// fp32 weight + fp32 sum: full rate, one per value as before).
template <int D, bool SIDE = true>
struct GatherSideH {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const char* __restrict__ tz;             // fp16 map of this block: (NV, Hf, Wf, 512) halves, 1 KB per texel
  const TapRec* __restrict__ taps_lds;     // [g 4][col 16]
  int wave, q, pt;
  f32x4 (&xs)[kSlice][kGroups];
  u32x4 r[D + 1][4];
  u32x4 off4[kGroups];
  f32x4 w4[kGroups];
  f32x4 bw, bv;

  __device__ __forceinline__ GatherSideH(const void* tz_, const TapRec* taps_lds_, int wave_, int q_, int pt_, f32x4 (&xs_)[kSlice][kGroups])
      : tz(reinterpret_cast<const char*>(tz_)), taps_lds(taps_lds_), wave(wave_), q(q_), pt(pt_), xs(xs_) {
    prefetch<0>();
  }
  template <int g>
  __device__ __forceinline__ void prefetch() {
    off4[g] = *reinterpret_cast<const u32x4*>(taps_lds[g * 16 + pt].off);
    w4[g] = *reinterpret_cast<const f32x4*>(taps_lds[g * 16 + pt].w) * kScale;
  }
  template <int U, int KTAP>
  __device__ __forceinline__ void issue_tap() {
    constexpr int g = U >> 2, mp = U & 3;
    const unsigned lane_off = (unsigned)(wave * 256 + q * 16);
    r[U % (D + 1)][KTAP] = *reinterpret_cast<const u32x4*>(tz + (off4[g][KTAP] * 1024u + lane_off) + mp * 64);
  }
  static __device__ __forceinline__ float mix_lo(unsigned h, float w, float c) {      // float(low half of h) * w + c
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(w), "v"(c));
    return d;
  }
  static __device__ __forceinline__ float mix_hi(unsigned h, float w, float c) {      // float(high half of h) * w + c
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(w), "v"(c));
    return d;
  }
  // tap K of unit U into row tile 2 mp + HF: the blend is summed on its own and added to the accumulator in ONE step behind tap 3 (the
  // GEMM this rides on accumulates into the same registers between the quarter-steps)
  template <int U, int HF, int K>
  __device__ __forceinline__ void blend_step() {
    constexpr int g = U >> 2, mo = 2 * (U & 3) + HF;
    const u32x4& t = r[U % (D + 1)][K];
    if constexpr (K == 0) {
      bw = w4[g];
      asm volatile("" : "+v"(bw));           // (see GatherSide::blend_step)
      bv = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
    bv[0] = mix_lo(t[2 * HF], bw[K], bv[0]);
    bv[1] = mix_hi(t[2 * HF], bw[K], bv[1]);
    bv[2] = mix_lo(t[2 * HF + 1], bw[K], bv[2]);
    bv[3] = mix_hi(t[2 * HF + 1], bw[K], bv[3]);
    if constexpr (K == 3) {
      asm volatile("" : "+a"(xs[mo][g]));      // keep the accumulator file assignment: read, add, write back
      f32x4 acc = xs[mo][g];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = add1(acc[i], bv[i]);
      xs[mo][g] = acc;
      asm volatile("" : "+a"(xs[mo][g]));
    }
  }
  template <int H, int G>
  __device__ __forceinline__ void run() {
    constexpr int U = H >> 1, V = U - D;
    if constexpr (V >= 0 && V < 16) blend_step<(V >= 0 && V < 16 ? V : 0), (H & 1), G>();
    if constexpr ((H & 1) == 0 && U < 16) issue_tap<(U < 16 ? U : 0), G>();
    if constexpr (G == 1 && (H & 7) == 7 && H < 31) prefetch<(H < 31 ? (H + 1) >> 3 : 0)>();
  }
  __device__ __forceinline__ void finish() {
    static_for<D>([&](auto I) {
      constexpr int V = 16 - D + decltype(I)::value;
      static_for<2>([&](auto HF) {
        blend_step<V, decltype(HF)::value, 0>();
        blend_step<V, decltype(HF)::value, 1>();
        blend_step<V, decltype(HF)::value, 2>();
        blend_step<V, decltype(HF)::value, 3>();
      });
    });
  }
  __device__ __forceinline__ void all() {
    static_for<32>([&](auto H) {
      run<decltype(H)::value, 0>();
      run<decltype(H)::value, 1>();
      run<decltype(H)::value, 2>();
      run<decltype(H)::value, 3>();
    });
    finish();
  }
};

// one accumulator block (this wave's 128 features x the 64 columns of the tile) -> a saved activation tensor (see SaveActs), x 1/16;
// column group g is view g (per-view kernel: rows g P + p) or the g-th 16-point tile of the workgroup's 64 points (post kernel: rows p)
// bits: the relu decisions of the same values, one dword per lane and row -- the lane's 32 features 16 mo + 4 q + i of the wave's slice at bit
// 4 mo + i, dword 4 wave + q of the row's 16 (the layout the training data gradients read, Lin512Args.maskbits)
template <bool PER_VIEW>
__device__ __forceinline__ void save_block(float* __restrict__ dst, unsigned* __restrict__ bits, long long P, long long tile, int wave, int lane,
                                           const f32x4 (&acc)[kSlice][kGroups]) {
  static_assert(kSlice == 8, "one dword of decisions per lane");
  const int q = lane >> 4, n = lane & 15;
#pragma unroll
  for (int g = 0; g < kGroups; ++g) {
    const long long p = PER_VIEW ? tile * kPtsPerWave + n : (tile * 4 + g) * kPtsPerWave + n;
    if (p >= P) continue;
    const size_t r = (PER_VIEW ? (size_t)g * P : (size_t)0) + (size_t)p;
    float* row = dst + r * kHidden + 128 * wave + 4 * q;
    unsigned m = 0;
#pragma unroll
    for (int mo = 0; mo < kSlice; ++mo) {
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int xi;
        asm("v_accvgpr_read_b32 %0, %1" : "=v"(xi) : "a"(acc[mo][g][i]));
        v[i] = __int_as_float(xi) * kInvScale;
        m |= (v[i] > 0.0f ? 1u : 0u) << (4 * mo + i);
      }
      *reinterpret_cast<f32x4*>(row + 16 * mo) = v;
    }
    if (bits) bits[r * 16 + 4 * wave + q] = m;
  }
}

template <bool LO, bool SAVE>
__device__ __forceinline__ void field_pre_body(const SceneDev& sc, const Args& a, const SaveActs& sv) {
  constexpr int kRing = LO ? DINER_HN_RING : DINER_HN_RING_F16, kRing0 = LO ? DINER_HN_RING0 : DINER_HN_RING0_F16;
  constexpr int kGDepth = LO ? DINER_HN_GDEPTH : DINER_HN_GDEPTH_F16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  h8* B = reinterpret_cast<h8*>(smem);
  TapRec* taps_lds = reinterpret_cast<TapRec*>(reinterpret_cast<char*>(smem) + (size_t)kBHalfs * 2);
  FeatRec* feat_tab = reinterpret_cast<FeatRec*>(reinterpret_cast<char*>(taps_lds) + kTapsBytes);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, pt = lane & 15;
  float* feat_src = reinterpret_cast<float*>(reinterpret_cast<char*>(feat_tab) + kFeatTabBytes) + wave * 64 * kSrcStride;
  const FieldArgs& fa = a.fa;
  if (threadIdx.x < 64) {      // recipe of (q, slot) = (threadIdx.x / 16, threadIdx.x % 16)
    const int sl = threadIdx.x & 15;
    feat_tab[threadIdx.x] = feat_recipe(16 * (sl >> 2) + 4 * (threadIdx.x >> 4) + (sl & 3), fa.freq_factor);
  }
  const long long n_tiles = (fa.P + kPtsPerWave - 1) / kPtsPerWave;
  __shared__ unsigned s_tile;
  TileQueue tq;
  tq.begin();
  tq.first(a.tile_counter, n_tiles, a.qmap, &s_tile);
  __syncthreads();
  const LdsB Bl = LdsB::make(B, lane);
  const _Float16* w_in = a.w;                                   // [4][2][8][2][64][8]  = 4 * 2 * 16 KB
  const _Float16* w_blk = a.w + (size_t)4 * 2 * 8192;           // then 6 layers of 4 * 16 * 16 KB
  constexpr size_t kLayerHalfs = (size_t)4 * 16 * 8192;

  Prof pf;
  pf.begin();
  for (long long tile = tq.initial(&s_tile); tile < n_tiles; tile = tq.next(tile, &s_tile)) {
    tq.request(a.tile_counter, n_tiles, a.qmap);
    long long p = tile * kPtsPerWave + pt;
    if (p >= fa.P) p = fa.P - 1;
    Taps taps;
    float feat[16];
    // The map sizes as per-tile opaque scalars: otherwise float(size) / 2, float(size - 1), ... are hoisted out of the tile loop,
    // do not survive the GEMMs in registers and come back from scratch in every tile (seven reloads in front of the projection).
    MapDims dims{sc.Wf, sc.Hf, sc.Ws, sc.Hs};
    asm volatile("" : "+s"(dims.Wf), "+s"(dims.Hf), "+s"(dims.Ws), "+s"(dims.Hs));
    frontend_h3n(sc, dims, fa, /*view=*/wave, q, lane, p, feat_tab, feat_src, taps, feat);
    pf.mark(0);
    __syncthreads();                              // previous tile's readers of B / taps are done
    pf.mark(1);
    {   // publish lin_in B operands (scale 1) for column group `wave` and this column's taps
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        h8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = feat[8 * t + j];
          const _Float16 hh = (_Float16)v;
          h[j] = hh;
          l[j] = (_Float16)(v - (float)hh);
        }
        B[((t * kGroups + wave) * 2 + 0) * 64 + lane] = h;
        if constexpr (LO) B[((t * kGroups + wave) * 2 + 1) * 64 + lane] = l;
      }
      if (q == 0) {
        TapRec r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          r.off[k] = (unsigned)(taps.off[k] >> 9);          // float offset -> units of 512 floats (one texel row)
          r.w[k] = taps.w[k];
        }
        taps_lds[wave * 16 + pt] = r;
      }
    }
    pf.mark(2);
    __syncthreads();
    pf.mark(3);
    f32x4 xs[kSlice][kGroups], ns[kSlice][kGroups];
    set_bias(xs, a.b, wave, q);
#if defined(DINER_HN_G0EARLY)
    if constexpr (LO) {      // experiment: block 0's first tap requests in front of the lin_in GEMM (measured, not kept: profiles/r04_ab_runs.txt)
      NoSide none;
      GatherSide<DINER_HN_G0DEPTH, false> g0(fa.tz, taps_lds, wave, q, pt, xs);
      g0.head();
      gemm<2, 2, LO>(w_in, Bl, wave, lane, xs, none);
      pf.mark(4);
      g0.tail();
      pf.mark(5);
    } else
#endif
    {
      NoSide none;
      gemm<2, 2, LO>(w_in, Bl, wave, lane, xs, none);
      pf.mark(4);
      if constexpr (LO) {
        GatherSide<DINER_HN_G0DEPTH, false> g0(fa.tz, taps_lds, wave, q, pt, xs);   // lin_z[0]: nothing long enough to hide under yet
        g0.all();
      } else {
        GatherSideH<DINER_HN_G0DEPTH_H, false> g0(fa.tz16, taps_lds, wave, q, pt, xs);
        g0.all();
      }
      pf.mark(5);
    }
#pragma nounroll
    for (int b = 0; b < 2; ++b) {
      const float* bias = a.b + kHidden * (1 + 2 * b);
      if constexpr (SAVE) save_block<true>(sv.X[b], sv.bX[b], fa.P, tile, wave, lane, xs);
      {
        NoSide none;
        publish_gemm<kRing0, LO, DINER_HN_EARLYA != 0, DINER_HN_OWN != 0>(
            w_blk + (size_t)(2 * b) * kLayerHalfs, Bl, wave, lane, xs, ns, none, [&] { set_bias(ns, bias, wave, q); }, pf, 6);
      }
      if constexpr (SAVE) save_block<true>(sv.H[b], sv.bH[b], fa.P, tile, wave, lane, ns);
      // the next block's lin_z contribution rides on the fc_1 GEMM (additions into xs commute); this block's fc_1 bias comes with it
      // (folded into the projected map's bias when the weights are packed, mlp.hip)
      const _Float16* w1 = w_blk + (size_t)(2 * b + 1) * kLayerHalfs;
#if DINER_HN_OWNG
      if constexpr (LO) {
        GatherSide<kGDepth> gs(fa.tz + (size_t)(b + 1) * fa.tz_stride, taps_lds, wave, q, pt, xs);
        publish_gemm<kRing, LO, DINER_HN_EARLY1 != 0, true>(w1, Bl, wave, lane, ns, xs, gs, [&] { pin_acc(xs); }, pf, 10);
      } else {
        GatherSideH<DINER_HN_GDEPTH_H> gs(reinterpret_cast<const _Float16*>(fa.tz16) + (size_t)(b + 1) * fa.tz_stride, taps_lds, wave, q, pt, xs);
        publish_gemm<kRing, LO, DINER_HN_EARLY1 != 0, true>(w1, Bl, wave, lane, ns, xs, gs, [&] { pin_acc(xs); }, pf, 10);
      }
#else
      __syncthreads();
      pf.mark(10);
      publish<LO>(Bl, wave, lane, ns);
      pf.mark(11);
      __syncthreads();
      pf.mark(12);
      pin_acc(xs);
      GatherSide<kGDepth> gs(fa.tz + (size_t)(b + 1) * fa.tz_stride, taps_lds, wave, q, pt, xs);
      gemm<16, kRing, LO>(w1, Bl, wave, lane, xs, gs);
      pf.mark(13);
#endif
    }
    {   // block 2: no gather left (and its fc_1 bias is added by the post kernel)
      const float* bias = a.b + kHidden * 5;
      NoSide none;
      if constexpr (SAVE) save_block<true>(sv.X[2], sv.bX[2], fa.P, tile, wave, lane, xs);
      publish_gemm<kRing0, LO, DINER_HN_EARLYA != 0, DINER_HN_OWN != 0>(
          w_blk + (size_t)4 * kLayerHalfs, Bl, wave, lane, xs, ns, none, [&] { set_bias(ns, bias, wave, q); }, pf, 6);
      if constexpr (SAVE) save_block<true>(sv.H[2], sv.bH[2], fa.P, tile, wave, lane, ns);
      publish_gemm<kRing, LO, DINER_HN_EARLY1 != 0, DINER_HN_OWN != 0>(w_blk + (size_t)5 * kLayerHalfs, Bl, wave, lane, ns, xs, none,
                                                                [&] { pin_acc(xs); }, pf, 10);
    }
    // view mean = mean over the four column groups; hand-over at scale 1 in accumulator layout (row tile 8 w + mo)
    f32x4* out = reinterpret_cast<f32x4*>(fa.xpre) + (size_t)tile * (kTiles * 64) + lane;
#pragma unroll
    for (int mo = 0; mo < kSlice; ++mo)
      out[(8 * wave + mo) * 64] = (((xs[mo][0] + xs[mo][1]) + xs[mo][2]) + xs[mo][3]) * (0.25f * kInvScale);
    pf.mark(14);
  }
  pf.end(a.prof, lane);
}
template <bool LO>
__global__ __launch_bounds__(256, 1) void k_field_pre_h3n(SceneDev sc, Args a) { field_pre_body<LO, false>(sc, a, SaveActs{}); }
// the same kernel storing the pre-activations of blocks 0-2 (training forward, DINER_TRAIN_FUSED_FWD)
__global__ __launch_bounds__(256, 1) void k_train_fwd_pre(SceneDev sc, Args a, SaveActs sv) { field_pre_body<true, true>(sc, a, sv); }

// =====================================================================================================================================
// Round 5: the plain-fp16 per-view kernel with EIGHT waves per workgroup (two per SIMD): k_field_pre_h8.
//
// What the GEMM-chain micro-benchmark showed (tools/ubench/chain_f16.hip, profiles/r05_chain_f16_ubench.txt): with ONE wave per SIMD a clean
// 512-wide GEMM of the 64-column body takes 10.0-10.2 k clocks for 8.2 k clocks of MFMAs, and a 128-column body (half the weight bytes per
// column) takes 2 x 10.6-10.8 k -- the weight stream is NOT what bounds a plain-fp16 GEMM; one wave cannot issue v_mfma_f32_16x16x32_f16
// back to back (18-20 clocks apiece, tools/ubench/mfma_dep.hip).  The same 64 columns on eight waves of 64 features each -- two waves per
// SIMD whose MFMAs interleave -- run the GEMM in 8.35 k clocks (0.98 of the matrix pipe), and everything one wave does between MFMAs
// (operand conversion, tap blends, waiting for a load) is covered by the other wave's MFMAs instead of by hand-placed side tasks.
//   * wave w (0..7) owns features [64 w, 64 w + 64) of all 64 columns (4 views x 16 points): 4 row tiles x 4 column groups = 16 accumulators
//     (64 registers) for the residual stream, 16 for the hidden activation; 256 registers per wave
//   * weights [wave 8][k32 block][row tile 4][lane 64][8 halfs] (hi plane only), global -> VGPR through a register ring as before;
//     a fragment feeds 4 MFMAs (one per column group) -- the weight bytes per column are those of the 4-wave kernel
//   * B operands: one 64 KB LDS buffer [k32 16][g 4][lane 64] x 8 halfs; barrier, publish, barrier per layer (the own-chunk scheme of the
//     4-wave kernel is not needed to hide the conversion: the SIMD's other wave is multiplying)
//   * front end: wave w serves view w & 3 and computes the 8 inputs of k32 block w >> 2 per lane
//   * taps: the fp16 copy of the projected maps in the same channel order (a wave pair = one wave of the 4-wave kernel), 8 units (g, mp) per wave
//   * the hand-over to the post kernel is unchanged (row tile 4 w + mo)
namespace w8 {
constexpr int kS8 = 4;                                    // row tiles per wave
constexpr int kB8Bytes = 16 * kGroups * 1024;             // 64 KB
constexpr size_t kFeatSrcBytes8 = 512 * kSrcStride * 4;
constexpr size_t kLdsBytes8 = (size_t)2 * kB8Bytes + kTapsBytes + kFeatTabBytes + kFeatSrcBytes8;      // two B buffers (see the kernel)
static_assert(kLdsBytes8 <= 160 * 1024, "LDS of one CU");
constexpr size_t kLinInHalfs8 = (size_t)8 * 2 * 4 * 512, kLayerHalfs8 = (size_t)8 * 16 * 4 * 512;
#ifndef DINER_H8_RING
#define DINER_H8_RING 3
#endif
#ifndef DINER_H8_RING0            // ... of the GEMMs without a side task (no tap buffers live)
#define DINER_H8_RING0 4
#endif
#ifndef DINER_H8_GDEPTH          // units between a tap request and its blend: as a GEMM side task / stand-alone (block 0)
#define DINER_H8_GDEPTH 1
#endif
#ifndef DINER_H8_G0DEPTH
#define DINER_H8_G0DEPTH 4
#endif
#ifndef DINER_H8_FLAGS           // 1: the 512-wide GEMMs wait for the data they need next (per-wave publish flags) instead of a barrier per layer
#define DINER_H8_FLAGS 0
#endif
#ifndef DINER_H8_TAPS_A          // 1: the tap buffers of the side task in AGPRs
#define DINER_H8_TAPS_A 0
#endif

__device__ __forceinline__ lds_h8 bfrag8(lds_ptr base, int t, int g) { return (lds_h8)(base + (t * kGroups + g) * 1024); }

// The weight ring of one GEMM: BUFFER loads -- a descriptor over the wave's slice of the layer (scalars), the lane's 16 bytes as the one
// vector offset of the whole GEMM, the fragment as a constant scalar offset: no vector-ALU instruction per load (the flat form the compiler
// makes of `scalar base + lane offset + immediate` costs a v_lshl_add_u64 per fragment -- with two waves per SIMD the vector ALU's slots
// between the MFMAs are what the side tasks live on)
template <int KT, int R>
struct ARing8 {
  h8 a[R][4];
  __amdgpu_buffer_rsrc_t rs;
  unsigned avoff;
  template <int H>
  __device__ __forceinline__ void load1(h8 (&dst)[4], int i) {      // fragment i of k32 block H
    dst[i] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rs, avoff, (H * 4 + i) * 1024, 0));
  }
  __device__ __forceinline__ void start(const _Float16* __restrict__ layer, int wave, int lane) {
    rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(layer) + (size_t)wave * KT * 4096), 0, KT * 4096, 0x00020000);
    avoff = lane * 16;
    static_for<(R - 1 < KT ? R - 1 : KT)>([&](auto H) {
#pragma unroll
      for (int i = 0; i < 4; ++i) load1<decltype(H)::value>(a[decltype(H)::value], i);
    });
  }
};

struct NoSide8 {
  template <int T, int G>
  __device__ __forceinline__ void run() {}
  __device__ __forceinline__ void finish() {}
};

// acc[mo][g] += W[64 w + 16 mo ..][k] . B[k][16 g ..]: 16 MFMAs per k32 block (one weight fragment per row tile, one B fragment per
// column group); the fragments of block t + R - 1 are requested one per quarter-step, B fragment g of block t + 1 is re-read right
// after its last use for block t
// ACC_A: the accumulator block lives in the AGPR half of the file (the residual stream xs); false: in arch VGPRs (the hidden block ns, dead
// while the gather-carrying GEMM runs -- with both blocks pinned to AGPRs the arch half is 128 registers and the ring + taps spill)
// FL (round 5, 512-wide contractions): NO barrier in front of the GEMM.  Wave w walks the k32 blocks in ITS OWN order -- step s is block
// (2 w + s) & 15: its own two blocks first (it published them itself), then the next wave's, ... -- and, one step before it first reads
// the blocks of wave j, waits until that wave has published this layer's operands: `flags[j] >= need` (an LDS word per wave, stored with
// release semantics behind the wave's publish).  The weights are packed in the same rotated order (k_pack_layer_h8).  A wave that is
// done with its GEMM publishes at once into the other B buffer; nobody waits for the slowest wave any more, only for the data it needs
// next.  (Two buffers are enough: to FINISH layer L + 1 a wave needs every wave's layer-L operands, which a wave publishes after its
// own layer-L GEMM -- so nobody can still be reading the buffer a finished layer-(L+1) wave writes into.)
template <int KT, int R, bool ACC_A, bool FL = false, class Side>
__device__ __forceinline__ void gemm8(const _Float16* __restrict__ layer, lds_ptr Bb, int wave, int lane, f32x4 (&acc)[kS8][kGroups], Side& side,
                                      const unsigned* flags = nullptr, unsigned need = 0) {
  static_assert(!FL || KT == 16, "flag-synchronised walk: 512-wide contractions");
  ARing8<KT, R> ring;
  ring.start(layer, wave, lane);
  asm volatile("" : "+v"(Bb));
  h8 bb[kGroups];
  // FL: the lane's pointer to block step s (fragment offsets then are immediates below 4 KB); otherwise Bb + immediates
  auto step_ptr = [&](int s) -> lds_ptr {
    if constexpr (FL) {
      const int blk = (2 * wave + s) & 15;          // (scalar)
      lds_ptr pp = Bb + blk * (kGroups * 1024);
      asm volatile("" : "+v"(pp));
      return pp;
    } else {
      return Bb + s * (kGroups * 1024);
    }
  };
  lds_ptr pcur = step_ptr(0), pnext = KT > 1 ? step_ptr(1) : pcur;
#pragma unroll
  for (int g = 0; g < kGroups; ++g) bb[g] = *(lds_h8)(pcur + g * 1024);
  static_for<KT * kGroups>([&](auto Q) {
    constexpr int qi = decltype(Q)::value;
    constexpr int t = qi >> 2, g = qi & 3;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (t + R - 1 < KT) ring.template load1<(t + R - 1 < KT ? t + R - 1 : 0)>(ring.a[(t + R - 1) % R], g);
    if constexpr (g == 0 && t > 0) {
      bb[kGroups - 1] = *(lds_h8)(pcur + (kGroups - 1) * 1024);
      if constexpr (t + 1 < KT) pnext = step_ptr(t + 1);
    }
    if constexpr (FL && g == 1 && (t & 1) == 1 && t + 1 < KT) {
      // the blocks of step t + 1, t + 2 belong to wave (w + (t + 1) / 2) & 7: its publish of this layer must have happened
      const int j = (wave + (t + 1) / 2) & 7;
      while (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(flags + j, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) - (int)need < 0)
        __builtin_amdgcn_s_sleep(1);
    }
    if constexpr (g > 0 && t + 1 < KT) bb[g - 1] = *(lds_h8)(pnext + (g - 1) * 1024);
    side.template run<t, g>();
    h8 (&ac)[4] = ring.a[t % R];
    const h8 b0 = bb[g];
#pragma unroll
    for (int m = 0; m < kS8; ++m) DINER_HN_MFMA(acc[m][g], ac[m], b0);
#pragma unroll
    for (int m = 0; m < kS8; ++m) {
      if constexpr (ACC_A) asm volatile("" : "+a"(acc[m][g]));
      else asm volatile("" : "+v"(acc[m][g]));
    }
    if constexpr (g == kGroups - 1) pcur = pnext;
  });
  side.finish();
}

// two row tiles' values of one column group (rows 4q .. 4q+3 each) -> one B fragment: relu on the bit pattern, x 1/16, round to fp16
template <bool ACC_A>
__device__ __forceinline__ u32x4 cvt_frag8(const f32x4& x0, const f32x4& x1) {
  u32x4 h;
  if constexpr (ACC_A) {
    u32x4 l;
    cvt4<false, 0>(x0, kInvScale, h, l);
    cvt4<false, 1>(x1, kInvScale, h, l);
  } else {
    float v[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = mul1s(__int_as_float(max(__float_as_int(x0[i]), 0)), kInvScale);
      v[4 + i] = mul1s(__int_as_float(max(__float_as_int(x1[i]), 0)), kInvScale);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = cvt_pk_f16(v[2 * i], v[2 * i + 1]);
  }
  return h;
}

// relu(acc) / 16 -> fp16 B operands of this wave's two k32 blocks (2 w, 2 w + 1), all four column groups
template <bool ACC_A>
__device__ __forceinline__ void publish8(lds_ptr Bb, int wave, const f32x4 (&acc)[kS8][kGroups]) {
  lds_ptr mine = Bb + wave * (2 * kGroups * 1024);
  asm volatile("" : "+v"(mine));
#pragma unroll
  for (int tl = 0; tl < 2; ++tl)
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      *bfrag8(mine, tl, g) = __builtin_bit_cast(h8, cvt_frag8<ACC_A>(acc[2 * tl][g], acc[2 * tl + 1][g]));
    }
}

__device__ __forceinline__ void set_bias8(f32x4 (&acc)[kS8][kGroups], const float* __restrict__ bias, int wave, int q) {
#pragma unroll
  for (int mo = 0; mo < kS8; ++mo) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 64 * wave + 16 * mo + 4 * q);
#pragma unroll
    for (int g = 0; g < kGroups; ++g) acc[mo][g] = bv;
  }
}
__device__ __forceinline__ void pin_acc8(f32x4 (&acc)[kS8][kGroups]) {
#pragma unroll
  for (int mo = 0; mo < kS8; ++mo)
#pragma unroll
    for (int g = 0; g < kGroups; ++g) asm volatile("" : "+a"(acc[mo][g]));
}

// xs[mo][g] += 16 * interp(projected map) from the fp16 maps (see GatherSideH): 8 units U = (g, mp) of 4 taps; ONE 16-byte load per lane and
// tap carries the lane's four rows of the wave's row tiles 2 mp and 2 mp + 1.  Steps T (the k32 blocks of the GEMM this rides on, or the
// steps of the stand-alone loop) of four quarters G: unit U is requested during step 2 U (tap G in quarter G) and blended D units
// later, row tile 2 mp during step 2 (U + D), row tile 2 mp + 1 during the next one.  A column's tap rows (`off4`, needed when group g's
// units are requested: steps 4 g, 4 g + 2) and blend weights (`w4`, needed from step 4 g + 2 D on) are read from LDS one step ahead of
// their first use into ONE slot each (the previous group's last use lies behind by then).
template <int D>
struct Gather8 {
  static constexpr int kSteps = 2 * (8 + D);             // steps until the last unit is blended
  const char* __restrict__ tz;
  const TapRec* __restrict__ taps_lds;
  int pt;
  unsigned lane_off;
  f32x4 (&xs)[kS8][kGroups];
  u32x4 r[D + 1][4];
  u32x4 off4;
  f32x4 w4, bw, bv;
  __device__ __forceinline__ Gather8(const void* tz_, const TapRec* taps_lds_, int wave, int q, int pt_, f32x4 (&xs_)[kS8][kGroups])
      : tz(reinterpret_cast<const char*>(tz_)), taps_lds(taps_lds_), pt(pt_),
        lane_off((unsigned)((wave >> 1) * 256 + (wave & 1) * 128 + q * 16)), xs(xs_) {
    off4 = *reinterpret_cast<const u32x4*>(taps_lds[pt].off);
  }
  template <int U, int KTAP>
  __device__ __forceinline__ void issue_tap() {
    constexpr int mp = U & 1;
#ifdef DINER_H8_G_NOLOAD        // ablation: the side task without its loads
    asm volatile("" : "+v"(r[U % (D + 1)][KTAP]));
#else
#ifdef DINER_H8_TAPS_SAME       // ablation: every tap from texel row 0..3 (always cached): prices the taps' latency
    r[U % (D + 1)][KTAP] = *reinterpret_cast<const u32x4*>(tz + ((off4[KTAP] & 3u) * 1024u + lane_off) + mp * 64);
#else
    r[U % (D + 1)][KTAP] = *reinterpret_cast<const u32x4*>(tz + (off4[KTAP] * 1024u + lane_off) + mp * 64);
#endif
#if DINER_H8_TAPS_A
    // the tap lands in the AGPR half of the file (the hidden block's 64 registers are dead while this GEMM runs; the arch half holds the
    // weight ring and the B fragments): read back one dword at a time where it is blended
    asm volatile("" : "+a"(r[U % (D + 1)][KTAP]));
#endif
#endif
  }
  template <int U, int HF, int K>
  __device__ __forceinline__ void blend_step() {
    constexpr int g = U >> 1, mo = 2 * (U & 1) + HF;
#if DINER_H8_TAPS_A
    u32x4 t;
    {
      const u32x4& ta = r[U % (D + 1)][K];
      int t0, t1;
      asm("v_accvgpr_read_b32 %0, %1" : "=v"(t0) : "a"(ta[2 * HF]));
      asm("v_accvgpr_read_b32 %0, %1" : "=v"(t1) : "a"(ta[2 * HF + 1]));
      t[2 * HF] = (unsigned)t0;
      t[2 * HF + 1] = (unsigned)t1;
    }
#else
    const u32x4& t = r[U % (D + 1)][K];
#endif
#ifdef DINER_H8_NO_BLEND        // ablation: the loads without the arithmetic
    asm volatile("" :: "v"(t[2 * HF]), "v"(t[2 * HF + 1]));
    return;
#endif
    if constexpr (K == 0) {
      bw = w4;
      asm volatile("" : "+v"(bw));           // (see GatherSide::blend_step)
      bv = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
    bv[0] = GatherSideH<1>::mix_lo(t[2 * HF], bw[K], bv[0]);
    bv[1] = GatherSideH<1>::mix_hi(t[2 * HF], bw[K], bv[1]);
    bv[2] = GatherSideH<1>::mix_lo(t[2 * HF + 1], bw[K], bv[2]);
    bv[3] = GatherSideH<1>::mix_hi(t[2 * HF + 1], bw[K], bv[3]);
    if constexpr (K == 3) {
#ifdef DINER_H8_NO_ACCUM        // ablation: the blend without the accumulator update
      asm volatile("" :: "v"(bv));
      return;
#endif
      asm volatile("" : "+a"(xs[mo][g]));      // keep the accumulator file assignment: read, add, write back
      f32x4 acc = xs[mo][g];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = add1(acc[i], bv[i]);
      xs[mo][g] = acc;
      asm volatile("" : "+a"(xs[mo][g]));
    }
  }
  template <int T, int G>
  __device__ __forceinline__ void run() {
    constexpr int U = T >> 1, V = U - D;
    if constexpr (V >= 0 && V < 8) blend_step<(V >= 0 && V < 8 ? V : 0), (T & 1), G>();
    if constexpr ((T & 1) == 0 && U < 8) issue_tap<(U < 8 ? U : 0), G>();
    if constexpr (G == 1) {
      // tap rows of group (T + 1) / 4, whose first unit is requested in step T + 1 (this step, 4 g' - 1, is odd: it requests nothing)
      if constexpr ((T & 3) == 3 && (T + 1) / 4 < kGroups) off4 = *reinterpret_cast<const u32x4*>(taps_lds[((T + 1) / 4) * 16 + pt].off);
      // blend weights of group gw, first used in step 4 gw + 2 D = T + 1 (the previous group's weights were copied in quarter 0 of this step)
      if constexpr (T + 1 >= 2 * D && ((T + 1 - 2 * D) & 3) == 0 && (T + 1 - 2 * D) / 4 < kGroups)
        w4 = *reinterpret_cast<const f32x4*>(taps_lds[((T + 1 - 2 * D) / 4) * 16 + pt].w) * kScale;
    }
  }
  // behind a 16-block GEMM: the steps that are left
  __device__ __forceinline__ void finish() {
    static_for<(kSteps > 16 ? kSteps - 16 : 0)>([&](auto I) {
      constexpr int T = 16 + decltype(I)::value;
      run<T, 0>(); run<T, 1>(); run<T, 2>(); run<T, 3>();
    });
  }
  __device__ __forceinline__ void all() {              // stand-alone (block 0: no GEMM long enough in front of it)
    static_for<kSteps>([&](auto I) {
      constexpr int T = decltype(I)::value;
      run<T, 0>(); run<T, 1>(); run<T, 2>(); run<T, 3>();
    });
  }
};

__global__ __launch_bounds__(512, 1) void k_field_pre_h8(SceneDev sc, Args a) {
  constexpr int R = DINER_H8_RING, R0 = DINER_H8_RING0;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  h8* B = reinterpret_cast<h8*>(smem);
  TapRec* taps_lds = reinterpret_cast<TapRec*>(reinterpret_cast<char*>(smem) + (size_t)2 * kB8Bytes);
  FeatRec* feat_tab = reinterpret_cast<FeatRec*>(reinterpret_cast<char*>(taps_lds) + kTapsBytes);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, pt = lane & 15;
  const int view = wave & 3, tin = wave >> 2;          // front end: this wave's view and which k32 block of the 64 padded inputs it makes
  float* feat_src = reinterpret_cast<float*>(reinterpret_cast<char*>(feat_tab) + kFeatTabBytes) + wave * 64 * kSrcStride;
  const FieldArgs& fa = a.fa;
  if (threadIdx.x < 64) {      // recipe of (q, slot) = (threadIdx.x / 16, threadIdx.x % 16)
    const int sl = threadIdx.x & 15;
    feat_tab[threadIdx.x] = feat_recipe(16 * (sl >> 2) + 4 * (threadIdx.x >> 4) + (sl & 3), fa.freq_factor);
  }
  const long long n_tiles = (fa.P + kPtsPerWave - 1) / kPtsPerWave;
  __shared__ unsigned s_flag[8];             // per wave: number of publishes done (DINER_H8_FLAGS)
  if (threadIdx.x < 8) s_flag[threadIdx.x] = 0;
  unsigned seq = 0;
  __shared__ unsigned s_tile;
  TileQueue tq;
  tq.begin();
  tq.first(a.tile_counter, n_tiles, a.qmap, &s_tile);
  __syncthreads();
  // TWO B buffers: a layer reads one and publishes its result into the other, so a wave converts its slice as soon as its own GEMM is done
  // -- while slower waves are still multiplying -- and ONE barrier per layer (everybody has published) is left.  A tile runs seven layers:
  // its last GEMM reads the buffer its lin_in operands were written to, so the next tile starts in the other one.
  lds_ptr Brd = (lds_ptr)(reinterpret_cast<char*>(B)) + lane * 16, Bwr = Brd + kB8Bytes;
  const _Float16* w_in = a.w8;
  const _Float16* w_blk = a.w8 + kLinInHalfs8;
  const _Float16* tz16 = reinterpret_cast<const _Float16*>(fa.tz16);

  Prof pf;
  pf.begin();
  for (long long tile = tq.initial(&s_tile); tile < n_tiles; tile = tq.next(tile, &s_tile)) {
    tq.request(a.tile_counter, n_tiles, a.qmap);
    long long p = tile * kPtsPerWave + pt;
    if (p >= fa.P) p = fa.P - 1;
    MapDims dims{sc.Wf, sc.Hf, sc.Ws, sc.Hs};
    asm volatile("" : "+s"(dims.Wf), "+s"(dims.Hf), "+s"(dims.Ws), "+s"(dims.Hs));
    h8 fin;                                       // this lane's 8 inputs of k32 block `tin`, as B operands
    Taps taps;
    {
      float px, py, pz, dx, dy, dz;
      load_point(fa, p, px, py, pz, dx, dy, dz);
      float xc[3], vd[3];
      world_to_cam(sc.R[view], sc.t[view], px, py, pz, xc[0], xc[1], xc[2]);           // pixelnerf.py:91-93
      vd[0] = rot_row(sc.R[view] + 0, dx, dy, dz);                                      // :100
      vd[1] = rot_row(sc.R[view] + 3, dx, dy, dz);
      vd[2] = rot_row(sc.R[view] + 6, dx, dy, dz);
      const float u = project_axis(xc[0], xc[2], sc.focal[view][0], sc.c[view][0], sc.img_w);   // :105-108
      const float w = project_axis(xc[1], xc[2], sc.focal[view][1], sc.c[view][1], sc.img_h);
      const int ix = nearest_border(u, dims.Ws), iy = nearest_border(w, dims.Hs);    // nearest depth tap (:114-116)
      const float dd = __fsub_rn(sc.depth[(size_t)view * dims.Hs * dims.Ws + (size_t)iy * dims.Ws + ix], xc[2]);
      float* mine = feat_src + lane * kSrcStride;        // read back by this lane only: program order, no barrier
      mine[0] = xc[0]; mine[1] = xc[1]; mine[2] = xc[2];
      mine[3] = vd[0]; mine[4] = vd[1]; mine[5] = vd[2];
      mine[6] = dd;    mine[7] = 0.0f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const FeatRec r = feat_tab[q * 16 + 8 * tin + j];
        const float x = mine[r.src];
        const float e = sin_posenc(__fmaf_rn(x, r.freq, r.phase));               // addcmul is fused, positional_encoding.py:46
        fin[j] = (_Float16)(r.sin ? e : x);
      }
      bilinear_taps(dims.Wf, dims.Hf, sc.feature_padding, view, u, w, taps);
    }
    pf.mark(0);
    // (no barrier here: this buffer was last read two layers before the previous tile ended, the taps by its second gather-carrying
    // GEMM -- every wave has passed at least two barriers since)
    pf.mark(1);
    *bfrag8(Brd, tin, view) = fin;
    if (tin == 0 && q == 0) {
      TapRec r;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        r.off[k] = (unsigned)(taps.off[k] >> 9);          // float offset -> units of 512 floats (one texel row)
        r.w[k] = taps.w[k];
      }
      taps_lds[view * 16 + pt] = r;
    }
    pf.mark(2);
    __syncthreads();
    pf.mark(3);
    f32x4 xs[kS8][kGroups], ns[kS8][kGroups];
    set_bias8(xs, a.b, wave, q);
    pin_acc8(xs);
    {
      NoSide8 none;
      gemm8<2, 2, true>(w_in, Brd, wave, lane, xs, none);
      pf.mark(4);
      Gather8<DINER_H8_G0DEPTH> g0(tz16, taps_lds, wave, q, pt, xs);      // lin_z[0]: nothing long enough to hide under yet
      g0.all();
      pf.mark(5);
    }
    // one residual block: x += fc_1(relu(fc_0(relu(x)))) (+ the next block's projected taps riding on the fc_1 GEMM).  The hidden block
    // lives inside the lambda: dead behind its publish, its 64 registers are free while the gather-carrying GEMM runs
    auto published = [&]() {                       // this wave's operands of the next layer are in LDS: tell the others
      ++seq;
#if DINER_H8_FLAGS
      if (lane == 0) __hip_atomic_store(&s_flag[wave], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
      __syncthreads();
#endif
    };
    auto block = [&](int b, auto&& side) {
      const float* bias = a.b + kHidden * (1 + 2 * b);
      pf.mark(6);
      publish8<true>(Bwr, wave, xs);                // (the other buffer: nobody reads it now)
      pf.mark(7);
      published();
      pf.mark(8);
      {
        f32x4 ns[kS8][kGroups];
        set_bias8(ns, bias, wave, q);
        NoSide8 none;
        gemm8<16, R0, true, DINER_H8_FLAGS != 0>(w_blk + (size_t)(2 * b) * kLayerHalfs8, Bwr, wave, lane, ns, none, s_flag, seq);
        pf.mark(9);
        pf.mark(10);
        publish8<true>(Brd, wave, ns);
        pf.mark(11);
      }
      published();
      pf.mark(12);
      pin_acc8(xs);
      constexpr int R1 = std::is_same<std::decay_t<decltype(side)>, NoSide8>::value ? R0 : R;
      gemm8<16, R1, true, DINER_H8_FLAGS != 0>(w_blk + (size_t)(2 * b + 1) * kLayerHalfs8, Brd, wave, lane, xs, side, s_flag, seq);
      pf.mark(13);
    };
#pragma nounroll
    for (int b = 0; b < 2; ++b) {
      // the next block's lin_z contribution rides on the fc_1 GEMM (additions into xs commute); this block's fc_1 bias comes with it
      // (folded into the projected map's bias when the weights are packed, mlp.hip)
#ifdef DINER_H8_NO_GATHER       // ablation
      NoSide8 gs;
#else
      Gather8<DINER_H8_GDEPTH> gs(tz16 + (size_t)(b + 1) * fa.tz_stride, taps_lds, wave, q, pt, xs);
#endif
      block(b, gs);
    }
    {   // block 2: no gather left (and its fc_1 bias is added by the post kernel)
      NoSide8 none;
      block(2, none);
    }
    // view mean = mean over the four column groups; hand-over at scale 1 in accumulator layout (row tile 4 w + mo)
    f32x4* out = reinterpret_cast<f32x4*>(fa.xpre) + (size_t)tile * (kTiles * 64) + lane;
#pragma unroll
    for (int mo = 0; mo < kS8; ++mo)
      out[(4 * wave + mo) * 64] = (((xs[mo][0] + xs[mo][1]) + xs[mo][2]) + xs[mo][3]) * (0.25f * kInvScale);
    pf.mark(14);
    { const lds_ptr t = Brd; Brd = Bwr; Bwr = t; }      // the last GEMM read Brd: the next tile's inputs go to the other buffer
  }
  pf.end(a.prof, lane);
}

// -------------------------------------------------------------------------------------------------------------------------------------
// The same decomposition for the f16x3 arithmetic (hi and lo planes, three MFMAs per product): k_field_pre_h8x.
//   * weights [wave 8][k32 block][row tile 4][hi | lo][lane 64][8 halfs]: 8 KB per wave and block, two buffer loads per quarter-step,
//     ring of R blocks (R = 2: a block is 48 MFMAs = 768 clocks of this wave, about twice that of wall time with two waves per SIMD)
//   * B operands [k32 16][g 4][hi | lo][lane 64]: 128 KB, ONE buffer (two would not fit): barrier, publish, barrier per layer
//   * taps: fp32 projected maps (the parity-grade mode keeps fp32 taps), 16 units (g, mo) of 4 taps per wave; the tap buffers live in the
//     AGPR half of the file (the hidden block's 64 registers are dead while the gather-carrying GEMM runs; the arch half holds ring + B)
constexpr size_t kLinInHalfs8x = (size_t)8 * 2 * 4 * 2 * 512, kLayerHalfs8x = (size_t)8 * 16 * 4 * 2 * 512;
constexpr size_t kLdsBytes8x = (size_t)2 * kB8Bytes + kTapsBytes + kFeatTabBytes + kFeatSrcBytes8;      // (one B buffer of hi + lo planes)
#ifndef DINER_H8X_RING
#define DINER_H8X_RING 2
#endif
#ifndef DINER_H8X_GDEPTH
#define DINER_H8X_GDEPTH 1
#endif
#ifndef DINER_H8X_G0DEPTH
#define DINER_H8X_G0DEPTH 4
#endif
#ifndef DINER_H8X_TAPS_A
#define DINER_H8X_TAPS_A 0
#endif

__device__ __forceinline__ lds_h8 bfrag8x(lds_ptr base, int t, int g, int hl) { return (lds_h8)(base + ((t * kGroups + g) * 2 + hl) * 1024); }

template <int KT, int R>
struct ARing8X {
  h8 a[R][8];                            // [row tile m][hi | lo] = a[.][2 m + hl]
  __amdgpu_buffer_rsrc_t rs;
  unsigned avoff;
  template <int H>
  __device__ __forceinline__ void load2(h8 (&dst)[8], int pair) {      // fragments 2 pair, 2 pair + 1 (hi, lo of row tile `pair`) of block H
#pragma unroll
    for (int i = 2 * pair; i < 2 * pair + 2; ++i)
      dst[i] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rs, avoff, (H * 8 + i) * 1024, 0));
  }
  __device__ __forceinline__ void start(const _Float16* __restrict__ layer, int wave, int lane) {
    rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(layer) + (size_t)wave * KT * 8192), 0, KT * 8192, 0x00020000);
    avoff = lane * 16;
    static_for<(R - 1 < KT ? R - 1 : KT)>([&](auto H) {
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) load2<decltype(H)::value>(a[decltype(H)::value], pr);
    });
  }
};

// acc[mo][g] += W . B in three products (hi hi, lo hi, hi lo): 48 MFMAs per k32 block
template <int KT, int R, class Side>
__device__ __forceinline__ void gemm8x(const _Float16* __restrict__ layer, lds_ptr Bb, int wave, int lane, f32x4 (&acc)[kS8][kGroups], Side& side) {
  ARing8X<KT, R> ring;
  ring.start(layer, wave, lane);
  asm volatile("" : "+v"(Bb));
  h8 bb[kGroups][2];
#pragma unroll
  for (int g = 0; g < kGroups; ++g) {
    bb[g][0] = *bfrag8x(Bb, 0, g, 0);
    bb[g][1] = *bfrag8x(Bb, 0, g, 1);
  }
  static_for<KT * kGroups>([&](auto Q) {
    constexpr int qi = decltype(Q)::value;
    constexpr int t = qi >> 2, g = qi & 3;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (t + R - 1 < KT) ring.template load2<(t + R - 1 < KT ? t + R - 1 : 0)>(ring.a[(t + R - 1) % R], g);
    if constexpr (g > 0 && t + 1 < KT) {
      bb[g - 1][0] = *bfrag8x(Bb, t + 1, g - 1, 0);
      bb[g - 1][1] = *bfrag8x(Bb, t + 1, g - 1, 1);
    }
    if constexpr (g == 0 && t > 0) {
      bb[kGroups - 1][0] = *bfrag8x(Bb, t, kGroups - 1, 0);
      bb[kGroups - 1][1] = *bfrag8x(Bb, t, kGroups - 1, 1);
    }
    side.template run<t, g>();
    h8 (&ac)[8] = ring.a[t % R];
    const h8 b0 = bb[g][0], b1 = bb[g][1];
#pragma unroll
    for (int m = 0; m < kS8; ++m) DINER_HN_MFMA(acc[m][g], ac[2 * m], b0);
#pragma unroll
    for (int m = 0; m < kS8; ++m) DINER_HN_MFMA(acc[m][g], ac[2 * m + 1], b0);
#pragma unroll
    for (int m = 0; m < kS8; ++m) DINER_HN_MFMA(acc[m][g], ac[2 * m], b1);
#pragma unroll
    for (int m = 0; m < kS8; ++m) asm volatile("" : "+a"(acc[m][g]));
  });
  side.finish();
}

__device__ __forceinline__ void publish8x(lds_ptr Bb, int wave, const f32x4 (&acc)[kS8][kGroups]) {
  lds_ptr mine = Bb + wave * (2 * kGroups * 2 * 1024);
  asm volatile("" : "+v"(mine));
#pragma unroll
  for (int tl = 0; tl < 2; ++tl)
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      u32x4 h, l;
      cvt4<true, 0>(acc[2 * tl][g], kInvScale, h, l);
      cvt4<true, 1>(acc[2 * tl + 1][g], kInvScale, h, l);
      *bfrag8x(mine, tl, g, 0) = __builtin_bit_cast(h8, h);
      *bfrag8x(mine, tl, g, 1) = __builtin_bit_cast(h8, l);
    }
}

// xs[mo][g] += 16 * interp(projected map), fp32 maps: 16 units U = (g, mo) of 4 taps (one f32x4 per lane and tap: rows 4q .. 4q+3 of the
// wave's row tile mo).  Unit U is requested during step U (tap G in quarter G) and blended D units later.  Tap buffers in AGPRs (TA).
template <int D, bool TA>
struct Gather8F {
  static constexpr int kSteps = 16 + D;
  const float* __restrict__ tz;
  const TapRec* __restrict__ taps_lds;
  int pt;
  unsigned lane_off;
  f32x4 (&xs)[kS8][kGroups];
  f32x4 r[D + 1][4];
  u32x4 off4;
  f32x4 w4, bw, bv;
  __device__ __forceinline__ Gather8F(const float* tz_, const TapRec* taps_lds_, int wave, int q, int pt_, f32x4 (&xs_)[kS8][kGroups])
      : tz(tz_), taps_lds(taps_lds_), pt(pt_), lane_off((unsigned)((16 * wave + q) * 16)), xs(xs_) {
    off4 = *reinterpret_cast<const u32x4*>(taps_lds[pt].off);
  }
  template <int U, int KTAP>
  __device__ __forceinline__ void issue_tap() {
    constexpr int mo = U & 3;
    const char* base = reinterpret_cast<const char*>(tz);
#ifdef DINER_H8X_G_NOLOAD
    asm volatile("" : "+v"(r[U % (D + 1)][KTAP]));
#else
    r[U % (D + 1)][KTAP] = *reinterpret_cast<const f32x4*>(base + (off4[KTAP] * 2048u + lane_off) + mo * 64);
#endif
    if constexpr (TA) asm volatile("" : "+a"(r[U % (D + 1)][KTAP]));
  }
  template <int U, int K>
  __device__ __forceinline__ void blend_step() {
    constexpr int g = U >> 2, mo = U & 3;
    f32x4 t;
    if constexpr (TA) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int ti;
        asm("v_accvgpr_read_b32 %0, %1" : "=v"(ti) : "a"(r[U % (D + 1)][K][i]));
        t[i] = __int_as_float(ti);
      }
    } else {
      t = r[U % (D + 1)][K];
    }
#ifdef DINER_H8X_NO_BLEND
    asm volatile("" :: "v"(t));
    return;
#endif
    if constexpr (K == 0) {
      bw = w4;
      asm volatile("" : "+v"(bw));           // (see GatherSide::blend_step)
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = K == 0 ? mul1(t[i], bw[0]) : fma1(t[i], bw[K], bv[i]);
    if constexpr (K == 3) {
      asm volatile("" : "+a"(xs[mo][g]));
      f32x4 acc = xs[mo][g];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = add1(acc[i], bv[i]);
      xs[mo][g] = acc;
      asm volatile("" : "+a"(xs[mo][g]));
    }
  }
  template <int T, int G>
  __device__ __forceinline__ void run() {
    constexpr int V = T - D;
    if constexpr (V >= 0 && V < 16) blend_step<(V >= 0 && V < 16 ? V : 0), G>();
    if constexpr (T < 16) issue_tap<(T < 16 ? T : 0), G>();
    if constexpr (G == 1) {
      // tap rows of the next group: its first unit is requested in step T + 1 = 4 g'.  This step's requests (quarters 2, 3 still to come) use
      // the old rows: read into a spare and swap behind quarter 3 -- done by reading in quarter 3 instead (below)
    }
    if constexpr (G == 3) {
      if constexpr ((T & 3) == 3 && (T + 1) / 4 < kGroups) off4 = *reinterpret_cast<const u32x4*>(taps_lds[((T + 1) / 4) * 16 + pt].off);
    }
    if constexpr (G == 1) {
      // blend weights of group gw, first used in step 4 gw + D = T + 1 (this step's blend copied the previous group's in quarter 0)
      if constexpr (T + 1 >= D && ((T + 1 - D) & 3) == 0 && (T + 1 - D) / 4 < kGroups)
        w4 = *reinterpret_cast<const f32x4*>(taps_lds[((T + 1 - D) / 4) * 16 + pt].w) * kScale;
    }
  }
  __device__ __forceinline__ void finish() {
    static_for<D>([&](auto I) {
      constexpr int T = 16 + decltype(I)::value;
      run<T, 0>(); run<T, 1>(); run<T, 2>(); run<T, 3>();
    });
  }
  __device__ __forceinline__ void all() {
    static_for<kSteps>([&](auto I) {
      constexpr int T = decltype(I)::value;
      run<T, 0>(); run<T, 1>(); run<T, 2>(); run<T, 3>();
    });
  }
};

__global__ __launch_bounds__(512, 1) void k_field_pre_h8x(SceneDev sc, Args a) {
  constexpr int R = DINER_H8X_RING;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  h8* B = reinterpret_cast<h8*>(smem);
  TapRec* taps_lds = reinterpret_cast<TapRec*>(reinterpret_cast<char*>(smem) + (size_t)2 * kB8Bytes);
  FeatRec* feat_tab = reinterpret_cast<FeatRec*>(reinterpret_cast<char*>(taps_lds) + kTapsBytes);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, pt = lane & 15;
  const int view = wave & 3, tin = wave >> 2;
  float* feat_src = reinterpret_cast<float*>(reinterpret_cast<char*>(feat_tab) + kFeatTabBytes) + wave * 64 * kSrcStride;
  const FieldArgs& fa = a.fa;
  if (threadIdx.x < 64) {
    const int sl = threadIdx.x & 15;
    feat_tab[threadIdx.x] = feat_recipe(16 * (sl >> 2) + 4 * (threadIdx.x >> 4) + (sl & 3), fa.freq_factor);
  }
  const long long n_tiles = (fa.P + kPtsPerWave - 1) / kPtsPerWave;
  __shared__ unsigned s_tile;
  TileQueue tq;
  tq.begin();
  tq.first(a.tile_counter, n_tiles, a.qmap, &s_tile);
  __syncthreads();
  lds_ptr Bb = (lds_ptr)(reinterpret_cast<char*>(B)) + lane * 16;
  const _Float16* w_in = a.w8x;
  const _Float16* w_blk = a.w8x + kLinInHalfs8x;

  Prof pf;
  pf.begin();
  for (long long tile = tq.initial(&s_tile); tile < n_tiles; tile = tq.next(tile, &s_tile)) {
    tq.request(a.tile_counter, n_tiles, a.qmap);
    long long p = tile * kPtsPerWave + pt;
    if (p >= fa.P) p = fa.P - 1;
    MapDims dims{sc.Wf, sc.Hf, sc.Ws, sc.Hs};
    asm volatile("" : "+s"(dims.Wf), "+s"(dims.Hf), "+s"(dims.Ws), "+s"(dims.Hs));
    h8 fin_h, fin_l;
    Taps taps;
    {
      float px, py, pz, dx, dy, dz;
      load_point(fa, p, px, py, pz, dx, dy, dz);
      float xc[3], vd[3];
      world_to_cam(sc.R[view], sc.t[view], px, py, pz, xc[0], xc[1], xc[2]);
      vd[0] = rot_row(sc.R[view] + 0, dx, dy, dz);
      vd[1] = rot_row(sc.R[view] + 3, dx, dy, dz);
      vd[2] = rot_row(sc.R[view] + 6, dx, dy, dz);
      const float u = project_axis(xc[0], xc[2], sc.focal[view][0], sc.c[view][0], sc.img_w);
      const float w = project_axis(xc[1], xc[2], sc.focal[view][1], sc.c[view][1], sc.img_h);
      const int ix = nearest_border(u, dims.Ws), iy = nearest_border(w, dims.Hs);
      const float dd = __fsub_rn(sc.depth[(size_t)view * dims.Hs * dims.Ws + (size_t)iy * dims.Ws + ix], xc[2]);
      float* mine = feat_src + lane * kSrcStride;
      mine[0] = xc[0]; mine[1] = xc[1]; mine[2] = xc[2];
      mine[3] = vd[0]; mine[4] = vd[1]; mine[5] = vd[2];
      mine[6] = dd;    mine[7] = 0.0f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const FeatRec r = feat_tab[q * 16 + 8 * tin + j];
        const float x = mine[r.src];
        const float e = sin_posenc(__fmaf_rn(x, r.freq, r.phase));
        const float v = r.sin ? e : x;
        const _Float16 hh = (_Float16)v;
        fin_h[j] = hh;
        fin_l[j] = (_Float16)(v - (float)hh);
      }
      bilinear_taps(dims.Wf, dims.Hf, sc.feature_padding, view, u, w, taps);
    }
    pf.mark(0);
    __syncthreads();                              // previous tile's readers of B / taps are done
    pf.mark(1);
    *bfrag8x(Bb, tin, view, 0) = fin_h;
    *bfrag8x(Bb, tin, view, 1) = fin_l;
    if (tin == 0 && q == 0) {
      TapRec r;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        r.off[k] = (unsigned)(taps.off[k] >> 9);
        r.w[k] = taps.w[k];
      }
      taps_lds[view * 16 + pt] = r;
    }
    pf.mark(2);
    __syncthreads();
    pf.mark(3);
    f32x4 xs[kS8][kGroups];
    set_bias8(xs, a.b, wave, q);
    pin_acc8(xs);
    {
      NoSide8 none;
      gemm8x<2, 2>(w_in, Bb, wave, lane, xs, none);
      pf.mark(4);
      Gather8F<DINER_H8X_G0DEPTH, false> g0(fa.tz, taps_lds, wave, q, pt, xs);
      g0.all();
      pf.mark(5);
    }
    auto block = [&](int b, auto&& side) {
      const float* bias = a.b + kHidden * (1 + 2 * b);
      __syncthreads();                            // everybody finished reading the previous B
      pf.mark(6);
      publish8x(Bb, wave, xs);
      pf.mark(7);
      __syncthreads();
      pf.mark(8);
      {
        f32x4 ns[kS8][kGroups];
        set_bias8(ns, bias, wave, q);
        NoSide8 none;
        gemm8x<16, R>(w_blk + (size_t)(2 * b) * kLayerHalfs8x, Bb, wave, lane, ns, none);
        pf.mark(9);
        __syncthreads();
        pf.mark(10);
        publish8x(Bb, wave, ns);
        pf.mark(11);
      }
      __syncthreads();
      pf.mark(12);
      pin_acc8(xs);
      gemm8x<16, R>(w_blk + (size_t)(2 * b + 1) * kLayerHalfs8x, Bb, wave, lane, xs, side);
      pf.mark(13);
    };
#pragma nounroll
    for (int b = 0; b < 2; ++b) {
#ifdef DINER_H8X_NO_GATHER      // ablation
      NoSide8 gs;
#else
      Gather8F<DINER_H8X_GDEPTH, DINER_H8X_TAPS_A != 0> gs(fa.tz + (size_t)(b + 1) * fa.tz_stride, taps_lds, wave, q, pt, xs);
#endif
      block(b, gs);
    }
    {
      NoSide8 none;
      block(2, none);
    }
    f32x4* out = reinterpret_cast<f32x4*>(fa.xpre) + (size_t)tile * (kTiles * 64) + lane;
#pragma unroll
    for (int mo = 0; mo < kS8; ++mo)
      out[(4 * wave + mo) * 64] = (((xs[mo][0] + xs[mo][1]) + xs[mo][2]) + xs[mo][3]) * (0.25f * kInvScale);
    pf.mark(14);
  }
  pf.end(a.prof, lane);
}

// layer packing for k_field_pre_h8x: [w 8][t KT][mo 4][hl 2][lane 64][8]: hi / lo of W * scale
__global__ void k_pack_layer_h8x(const float* __restrict__ W, int rows, int cols, int KT, float scale, _Float16* __restrict__ dst) {
  const long long total = (long long)8 * KT * 4096;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63, hl = (i >> 9) & 1, mo = (i >> 10) & 3;
    const int wt = (int)(i >> 12), t = wt % KT, w = wt / KT;
    const int row = 64 * w + 16 * mo + (lane & 15);
    const int col = 32 * t + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
    const float x = (row < rows && col < cols) ? W[(size_t)row * cols + col] * scale : 0.0f;
    const _Float16 h = (_Float16)x;
    dst[i] = hl ? (_Float16)(x - (float)h) : h;
  }
}

// layer packing for k_field_pre_h8: [w 8][t KT][mo 4][lane 64][8] = W[64 w + 16 mo + (lane & 15)][32 t + 16 (j >> 2) + 4 (lane >> 4) + (j & 3)] * scale
// (512-wide layers, DINER_H8_FLAGS: position s of wave w's stream is k32 block (2 w + s) & 15 -- the wave's own blocks first, see gemm8)
__global__ void k_pack_layer_h8(const float* __restrict__ W, int rows, int cols, int KT, float scale, _Float16* __restrict__ dst) {
  const long long total = (long long)8 * KT * 2048;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63, mo = (i >> 9) & 3;
    const int wt = (int)(i >> 11), ts = wt % KT, w = wt / KT;
    const int t = (KT == 16 && DINER_H8_FLAGS != 0) ? ((2 * w + ts) & 15) : ts;
    const int row = 64 * w + 16 * mo + (lane & 15);
    const int col = 32 * t + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
    dst[i] = (_Float16)((row < rows && col < cols) ? W[(size_t)row * cols + col] * scale : 0.0f);
  }
}
}  // namespace w8

struct PostArgsN {
  PostArgs pa;
  const _Float16* w;        // n-split packed fc_0 / fc_1 of blocks 3, 4 (4 layers of 4 * 16 * 16 KB)
  const _Float16* w8;       // the same four layers in the eight-wave kernels' order (hi plane; k_field_post_h8), or null
  const _Float16* w_out;    // lin_out fragments [t 16][hl 2][lane 64][8] (rows >= 4 zero), x16; behind them (32 KB on) the fp32 pack
                            // [wave 4][mo 8][q 4][o 4][j 4] = Wout[o][128 wave + 16 mo + 4 q + j] / 16 of the vector-ALU lin_out
  unsigned long long* prof; // DINER_HN_PROF builds: phase counters, else unused
  unsigned* tile_counter;   // see TileQueue
  QueueMap qmap;            // the default map (tile % 8): the post kernel's tiles are 64 consecutive points, no taps
};

// The lane index, opaque to the optimiser: what is derived from it is derived at the point of use.  (Lane-derived values hoisted out
// of the post kernel's tile loop have to live across GEMMs that use all 512 registers: they come back from scratch.)
__device__ __forceinline__ int lane_here() {
  int l = threadIdx.x & 63;
  asm volatile("" : "+v"(l));
  return l;
}

// Blocks 3-4 + lin_out + output activations on the view-averaged hidden state, same feature-sliced scheme: a
// workgroup takes 64 points (four 16-point tiles = the four column groups), wave w owns features [128 w, 128 w + 128).
template <bool LO, bool SAVE>
__device__ __forceinline__ void field_post_body(const PostArgsN& a, const SaveActs& sv) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  h8* B = reinterpret_cast<h8*>(smem);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int q = lane >> 4, pt = lane & 15;
  const PostArgs& pa = a.pa;
  const LdsB Bl = LdsB::make(B, lane);
  const long long n_t16 = (pa.P + kPtsPerWave - 1) / kPtsPerWave;
  const long long n_tiles = (n_t16 + 3) / 4;
  constexpr size_t kLayerHalfs = (size_t)4 * 16 * 8192;

  Prof pf;
  pf.begin();
  __shared__ unsigned s_tile2[2];
  int par = 0;                                    // slot of the current tile's answer (TileQueue::park)
  TileQueue tq;
  tq.begin();
  f32x4 xs[kSlice][kGroups], ns[kSlice][kGroups];
#if DINER_HN_LINOUT_VALU
  typedef __attribute__((address_space(3))) f32x4* lds_f4;
  const lds_f4 lo_w = (lds_f4)((lds_ptr)(reinterpret_cast<char*>(smem)) + (size_t)kBHalfs * 2);
  const lds_f4 lo_part = (lds_f4)((lds_ptr)(reinterpret_cast<char*>(smem)) + (size_t)kBHalfs * 2 + kLinOutWBytes);
  {
    const f32x4* gw = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.w_out) + 32768);
    for (int i = threadIdx.x; i < (int)(kLinOutWBytes / 16); i += 256) lo_w[i] = gw[i];
    __syncthreads();
  }
#endif
  // The hand-over of a tile (2 KB per point, written by the per-view kernel in accumulator layout) is REQUESTED while the previous
  // tile's lin_out runs, straight into the residual block, which is dead from lin_out's publish on (round 2's attempt at this made the
  // allocator spill the block; with the accumulator accesses pinned it does not).  A tile starts by waiting for it: x16 + block 2's
  // fc_1 bias, which the per-view kernel leaves to this one.
  auto request_handover = [&](long long t) {
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      long long t16 = t * 4 + g;
      if (t16 >= n_t16) t16 = n_t16 - 1;
      const f32x4* xp = reinterpret_cast<const f32x4*>(pa.xpre);
      asm volatile("" : "+s"(xp));                 // per-tile address arithmetic: hoisted, the 64-bit lane addresses get spilled
      const f32x4* in = xp + (size_t)t16 * (kTiles * 64) + lane_here();
#pragma unroll
      for (int mo = 0; mo < kSlice; ++mo) xs[mo][g] = in[(8 * wave + mo) * 64];
    }
  };
  long long tile = blockIdx.x;
  if (tile < n_tiles) request_handover(tile);
  while (tile < n_tiles) {
    long long tile_next_v = n_tiles;
    // lane-derived quantities are re-derived per tile from an opaque copy: hoisted out of the tile loop they do not survive the GEMMs in
    // registers (the kernel uses all 512) and come back from scratch
    const int q = lane_here() >> 4;
    tq.request(a.tile_counter, n_tiles, a.qmap);
    const float* bpost = pa.b_post;
    asm volatile("" : "+s"(bpost));                // (as above)
#pragma unroll
    for (int mo = 0; mo < kSlice; ++mo) {
      const f32x4 b2 = *reinterpret_cast<const f32x4*>(bpost + 4 * kHidden + 16 + 128 * wave + 16 * mo + 4 * q);
#pragma unroll
      for (int g = 0; g < kGroups; ++g) xs[mo][g] = xs[mo][g] * kScale + b2;
    }
    NoSide none;
    pin_acc(xs);
    pf.mark(8);
#pragma nounroll
    for (int b = 0; b < 2; ++b) {
      const float* bias = bpost + 2 * kHidden * b;
      if constexpr (SAVE) save_block<false>(sv.X[3 + b], sv.bX[3 + b], pa.P, tile, wave, lane_here(), xs);
      publish_gemm<(LO ? DINER_HN_RING0 : DINER_HN_RING0_F16), LO, DINER_HN_EARLYP != 0, DINER_HN_OWN != 0>(a.w + (size_t)(2 * b) * kLayerHalfs, Bl, wave, lane, xs, ns, none, [&] {
        set_bias(ns, bias, wave, lane_here() >> 4);
        pin_acc(xs);                              // the residual stream stays in registers across the fc_0 GEMM
      }, pf, 0);
      if (b == 0) tq.park(&s_tile2[par]);           // (the request went out at the top of the tile)
      if constexpr (SAVE) save_block<false>(sv.H[3 + b], sv.bH[3 + b], pa.P, tile, wave, lane_here(), ns);
      pin_acc(xs);
      publish_gemm<(LO ? DINER_HN_RING : DINER_HN_RING_F16), LO, DINER_HN_EARLYP != 0, DINER_HN_OWN != 0, false>(a.w + (size_t)(2 * b + 1) * kLayerHalfs, Bl, wave, lane, ns, xs, none,
                                      [&] { add_bias(xs, bias + kHidden, wave, lane_here() >> 4); }, pf, 4);
    }
#if DINER_HN_LINOUT_VALU
    pin_acc(xs);                                  // (else the block is copied to vector registers here and back for the reads below)
    if constexpr (SAVE) save_block<false>(sv.x_last, nullptr, pa.P, tile, wave, lane_here(), xs);
    // ---- lin_out on relu(x), fp32 on the vector ALU straight from the accumulators: a lane holds 4 features x 4 columns of each of its 8
    // row tiles; 512 fused multiply-adds give its share of the four outputs of its four columns, two shuffles sum the four feature
    // quarters, 4 KB of LDS the four waves.  (Until round 3 the block was published to LDS as fp16 hi / lo like a hidden layer and
    // multiplied with padded 16 x 16 x 32 MFMAs: 11 k + 8 k clocks per tile, 12 % of the kernel, for 0.4 % of its FLOPs; and an x
    // beyond the fp16 range -- finite in fp32 -- sent the launch to the exact-fp32 pass for nothing.)
    f32x4 res;
    int lane_o = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_o));                 // the section's lane quantities are derived here, not kept alive across the GEMMs
    const int q_o = lane_o >> 4, pt = lane_o & 15;
    // Range check of the launch, on the raw bits of the 128 values a lane holds of x (one integer max3 per pair, twice): the largest as
    // signed integers is the largest positive value (+inf / a NaN with a clear sign bit above all), the largest as unsigned integers is a
    // NaN with the sign bit set if there is one (then -inf).  An fp16 operand that left the range anywhere in the launch shows here: it
    // turns the products of its column into NaN / inf of either sign, relu (an integer maximum in every operand conversion) keeps the
    // positive ones and the next product spreads them to all features of the column, and they stay in the residual stream, which is
    // only ever added to; a residual value beyond the range (the operands are x / 16: 65504 * 16 here) stays beyond it.  (Until round 3 the test was lin_out's own result being non-finite, which relied on relu letting the NaNs through:
    // those with the sign bit set became zeros.)
    bool wave_bad = false;
    {
      int m_pos = 0;
      unsigned m_neg = 0;
      f32x4 po[kGroups];
#pragma unroll
      for (int g = 0; g < kGroups; ++g) po[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // the weights of row tile mo + 1 are read while mo is multiplied, no further ahead: left alone the scheduler hoists all 32 reads
      // (128 registers) and shuffles the sums through the free accumulator registers (1500 instructions for 900)
      f32x4 wn[4];
#pragma unroll
      for (int o = 0; o < 4; ++o) wn[o] = lo_w[((wave * 8 + 0) * 4 + q_o) * 4 + o];
#pragma unroll
      for (int mo = 0; mo < kSlice; ++mo) {
        __builtin_amdgcn_sched_barrier(0);
        f32x4 wv[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) wv[o] = wn[o];
        if (mo + 1 < kSlice) {
#pragma unroll
          for (int o = 0; o < 4; ++o) wn[o] = lo_w[((wave * 8 + mo + 1) * 4 + q_o) * 4 + o];
        }
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
          float v[4];
          int xi[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {             // (explicit accumulator reads, see cvt4)
            asm("v_accvgpr_read_b32 %0, %1" : "=v"(xi[jj]) : "a"(xs[mo][g][jj]));
            v[jj] = __int_as_float(max(xi[jj], 0));
          }
          m_pos = max(max(m_pos, xi[0]), xi[1]);
          m_pos = max(max(m_pos, xi[2]), xi[3]);
          m_neg = max(max(m_neg, (unsigned)xi[0]), (unsigned)xi[1]);
          m_neg = max(max(m_neg, (unsigned)xi[2]), (unsigned)xi[3]);
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            float t = po[g][o];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) t = fmaf(wv[o][jj], v[jj], t);
            po[g][o] = t;
          }
        }
      }
      // x is dead from here on: the next tile's hand-over is requested now (its index was parked in LDS behind the first GEMM), ahead of
      // the sums over lanes and waves and of the barrier, whose wait the loads then fill (the top of a tile waited 5.7 k clocks for them)
      const long long tile_nx = tq.take(tile, &s_tile2[par]);
      if (tile_nx < n_tiles) request_handover(tile_nx);
#pragma unroll
      for (int g = 0; g < kGroups; ++g) {
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          float t = po[g][o];                       // (ds_bpermute by hand: __shfl_xor takes the lane index from mbcnt, hoisted and spilled)
          t += __int_as_float(__builtin_amdgcn_ds_bpermute((lane_o ^ 16) << 2, __float_as_int(t)));
          t += __int_as_float(__builtin_amdgcn_ds_bpermute((lane_o ^ 32) << 2, __float_as_int(t)));
          po[g][o] = t;
        }
        if (q_o == 0) lo_part[(wave * kGroups + g) * 16 + pt] = po[g];
      }
      // x is held x16: an operand conversion overflows from 65504 * 16 on (0x497fe000); -inf and the sign-bit NaNs from 0xff800000 on
      wave_bad = __any(m_pos >= 0x497fe000 || m_neg >= 0xff800000u);
      __syncthreads();
      pf.mark(9);
      // wave w finishes column group w (its 16 points): the four waves' shares
      const lds_f4 lp = lo_part + wave * 16 + pt;
      res = (lp[0 * kGroups * 16] + lp[1 * kGroups * 16]) + (lp[2 * kGroups * 16] + lp[3 * kGroups * 16]);
      pf.mark(10);
      tile_next_v = tile_nx;
    }
    {
#else
    const int q_o = q, pt = lane & 15;
    const bool wave_bad = false;
    // ---- lin_out on relu(x): wave w produces the four outputs of column group w (its 16 points)
    // (requesting its 32 weight fragments before the publish moves 4 k clocks from here into the publish and the next tile's
    // hand-over load: measured, no net gain)
    __syncthreads();
    publish<LO>(Bl, wave, lane, xs);
    __syncthreads();
    tile_next_v = tq.take(tile, &s_tile2[par]);
    if (tile_next_v < n_tiles) request_handover(tile_next_v);
    pf.mark(9);
    {
      typedef const __attribute__((address_space(1))) h8* gh8;
      gh8 wo = (gh8)(reinterpret_cast<const h8*>(a.w_out) + lane);
      asm volatile("" : "+v"(wo));                // loop-invariant otherwise: 32 hoisted (and spilled) addresses
      LdsB Bo = Bl;                               // column group `wave`: two fragments = 2 KB further on
      Bo.base += wave * 2048;
      Bo.opaque();
      f32x4 o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        if ((t & 3) == 0) __builtin_amdgcn_sched_barrier(0);      // keep the operand loads from being hoisted in one burst
        const h8 ah = wo[(t * 2 + 0) * 64], al = wo[(t * 2 + 1) * 64];
        const h8 bh = *LdsB::at(Bo.chunk(t >> 2), t & 3, 0, 0), bl = *LdsB::at(Bo.chunk(t >> 2), t & 3, 0, 1);
        DINER_HN_MFMA(o[t & 3], ah, bh);
        if constexpr (LO) {
          DINER_HN_MFMA(o[(t + 1) & 3], al, bh);
          DINER_HN_MFMA(o[(t + 2) & 3], ah, bl);
        }
      }
      f32x4 res = ((o[0] + o[1]) + (o[2] + o[3])) * kInvScale;
      asm volatile("" : "+v"(res));
      pf.mark(10);
#endif
      res += *reinterpret_cast<const f32x4*>(bpost + 4 * kHidden + 4 * q_o);     // lin_out bias kept at scale 1
      const long long t16 = tile * 4 + wave;
      const long long p = t16 * kPtsPerWave + pt;
      // the residual-stream range test of ANY wave raises the flag, whether or not this wave's column group holds points of the launch (a
      // ragged last tile repeats its last 16-point group: the repeats are real columns) -- detection does not rest on an overflow having
      // spread to the features of a wave that stores
      if (pa.overflow && wave_bad && lane_here() == 0) *pa.overflow = 1;
      if (t16 < n_t16 && q_o == 0 && p < pa.P) {
        // A hidden activation beyond the fp16 range turns into inf in a B operand and reaches every raw output of the
        // point as inf / NaN (so does a non-finite input): raise the flag that un-gates the exact-fp32 pass (mlp.hip).
        const float probe = (res[0] - res[0]) + (res[1] - res[1]) + (res[2] - res[2]) + (res[3] - res[3]);   // 0 or NaN
        if (pa.overflow && (probe != 0.0f || wave_bad)) *pa.overflow = 1;
        if constexpr (SAVE) reinterpret_cast<f32x4*>(sv.raw)[p] = res;      // lin_out's outputs in front of the activations
        if (!pa.raw) {
          res[0] = 1.0f / (1.0f + expf(-res[0]));
          res[1] = 1.0f / (1.0f + expf(-res[1]));
          res[2] = 1.0f / (1.0f + expf(-res[2]));
          res[3] = fmaxf(res[3], 0.0f);
        }
        reinterpret_cast<f32x4*>(pa.out)[p] = res;
      }
    }
    pf.mark(11);
    tile = tile_next_v;
    par ^= 1;
  }
  pf.end(a.prof, lane);
}
template <bool LO>
__global__ __launch_bounds__(256, 1) void k_field_post_h3n(PostArgsN a) { field_post_body<LO, false>(a, SaveActs{}); }
// the same kernel storing the pre-activations of blocks 3-4, the stream entering lin_out and lin_out's raw outputs (training forward)
__global__ __launch_bounds__(256, 1) void k_train_fwd_post(PostArgsN a, SaveActs sv) { field_post_body<true, true>(a, sv); }

// Round 5: the post kernel of the plain-fp16 mode on eight waves (two per SIMD), the scheme of k_field_pre_h8: wave w owns features
// [64 w, 64 w + 64) of the 64 points of a tile (four 16-point tiles = the four column groups); two B buffers, one barrier per layer; lin_out
// on the vector ALU straight from the accumulators as in k_field_post_h3n (the same fp32 weight pack: wave pair = one wave of that kernel),
// the eight waves' shares summed through 8 KB of LDS, waves 0..3 finish column group w.
namespace w8 {
constexpr size_t kLinOutPartBytes8 = 8 * 4 * 16 * 16;
constexpr size_t kLdsBytesPost8 = (size_t)2 * kB8Bytes + kLinOutWBytes + kLinOutPartBytes8;
static_assert(kLdsBytesPost8 <= 160 * 1024, "LDS of one CU");

__global__ __launch_bounds__(512, 1) void k_field_post_h8(PostArgsN a) {
  constexpr int R0 = DINER_H8_RING0;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const PostArgs& pa = a.pa;
  const long long n_t16 = (pa.P + kPtsPerWave - 1) / kPtsPerWave;
  const long long n_tiles = (n_t16 + 3) / 4;
  lds_ptr Brd = (lds_ptr)(reinterpret_cast<char*>(smem)) + lane * 16, Bwr = Brd + kB8Bytes;
  typedef __attribute__((address_space(3))) f32x4* lds_f4;
  const lds_f4 lo_w = (lds_f4)((lds_ptr)(reinterpret_cast<char*>(smem)) + (size_t)2 * kB8Bytes);
  const lds_f4 lo_part = (lds_f4)((lds_ptr)(reinterpret_cast<char*>(smem)) + (size_t)2 * kB8Bytes + kLinOutWBytes);
  {
    const f32x4* gw = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.w_out) + 32768);
    for (int i = threadIdx.x; i < (int)(kLinOutWBytes / 16); i += 512) lo_w[i] = gw[i];
  }
  Prof pf;
  pf.begin();
  __shared__ unsigned s_tile2[2];
  int par = 0;
  TileQueue tq;
  tq.begin();
  __syncthreads();
  f32x4 xs[kS8][kGroups];
  auto request_handover = [&](long long t) {
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      long long t16 = t * 4 + g;
      if (t16 >= n_t16) t16 = n_t16 - 1;
      const f32x4* xp = reinterpret_cast<const f32x4*>(pa.xpre);
      asm volatile("" : "+s"(xp));
      const f32x4* in = xp + (size_t)t16 * (kTiles * 64) + lane_here();
#pragma unroll
      for (int mo = 0; mo < kS8; ++mo) xs[mo][g] = in[(4 * wave + mo) * 64];
    }
  };
  long long tile = blockIdx.x;
  if (tile < n_tiles) request_handover(tile);
  while (tile < n_tiles) {
    long long tile_next_v = n_tiles;
    const int q = lane_here() >> 4;
    tq.request(a.tile_counter, n_tiles, a.qmap);
    const float* bpost = pa.b_post;
    asm volatile("" : "+s"(bpost));
#pragma unroll
    for (int mo = 0; mo < kS8; ++mo) {
      const f32x4 b2 = *reinterpret_cast<const f32x4*>(bpost + 4 * kHidden + 16 + 64 * wave + 16 * mo + 4 * q);
#pragma unroll
      for (int g = 0; g < kGroups; ++g) xs[mo][g] = xs[mo][g] * kScale + b2;
    }
    pin_acc8(xs);
    pf.mark(8);
#pragma nounroll
    for (int b = 0; b < 2; ++b) {
      const float* bias = bpost + 2 * kHidden * b;
      pf.mark(0);
      publish8<true>(Bwr, wave, xs);
      pf.mark(1);
      __syncthreads();
      pf.mark(2);
      {
        f32x4 ns[kS8][kGroups];
        set_bias8(ns, bias, wave, q);
        NoSide8 none;
        gemm8<16, R0, true>(a.w8 + (size_t)(2 * b) * kLayerHalfs8, Bwr, wave, lane, ns, none);
        pf.mark(3);
        if (b == 0) tq.park(&s_tile2[par]);         // (the request went out at the top of the tile)
        publish8<true>(Brd, wave, ns);
        pf.mark(5);
      }
      __syncthreads();
      pf.mark(6);
      pin_acc8(xs);
      {     // + fc_1's bias: into the accumulators the GEMM adds to
#pragma unroll
        for (int mo = 0; mo < kS8; ++mo) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + kHidden + 64 * wave + 16 * mo + 4 * q);
#pragma unroll
          for (int g = 0; g < kGroups; ++g) xs[mo][g] += bv;
        }
        pin_acc8(xs);
        NoSide8 none;
        gemm8<16, R0, true>(a.w8 + (size_t)(2 * b + 1) * kLayerHalfs8, Brd, wave, lane, xs, none);
      }
      pf.mark(7);
    }
    pin_acc8(xs);
    // ---- lin_out on relu(x), fp32 on the vector ALU straight from the accumulators (see k_field_post_h3n), + the range check of the launch
    f32x4 res = (f32x4){0.f, 0.f, 0.f, 0.f};
    int lane_o = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_o));
    const int q_o = lane_o >> 4, pt = lane_o & 15;
    bool wave_bad = false;
    {
      int m_pos = 0;
      unsigned m_neg = 0;
      f32x4 po[kGroups];
#pragma unroll
      for (int g = 0; g < kGroups; ++g) po[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int wrow = (wave >> 1) * 8 + (wave & 1) * 4;        // this wave's first row tile in the four-wave kernel's pack
      f32x4 wn[4];
#pragma unroll
      for (int o = 0; o < 4; ++o) wn[o] = lo_w[((wrow + 0) * 4 + q_o) * 4 + o];
#pragma unroll
      for (int mo = 0; mo < kS8; ++mo) {
        __builtin_amdgcn_sched_barrier(0);
        f32x4 wv[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) wv[o] = wn[o];
        if (mo + 1 < kS8) {
#pragma unroll
          for (int o = 0; o < 4; ++o) wn[o] = lo_w[((wrow + mo + 1) * 4 + q_o) * 4 + o];
        }
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
          float v[4];
          int xi[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            asm("v_accvgpr_read_b32 %0, %1" : "=v"(xi[jj]) : "a"(xs[mo][g][jj]));
            v[jj] = __int_as_float(max(xi[jj], 0));
          }
          m_pos = max(max(m_pos, xi[0]), xi[1]);
          m_pos = max(max(m_pos, xi[2]), xi[3]);
          m_neg = max(max(m_neg, (unsigned)xi[0]), (unsigned)xi[1]);
          m_neg = max(max(m_neg, (unsigned)xi[2]), (unsigned)xi[3]);
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            float t = po[g][o];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) t = fmaf(wv[o][jj], v[jj], t);
            po[g][o] = t;
          }
        }
      }
      // x is dead from here on: the next tile's hand-over is requested now (its index was parked in LDS behind the first GEMM)
      const long long tile_nx = tq.take(tile, &s_tile2[par]);
      if (tile_nx < n_tiles) request_handover(tile_nx);
#pragma unroll
      for (int g = 0; g < kGroups; ++g) {
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          float t = po[g][o];
          t += __int_as_float(__builtin_amdgcn_ds_bpermute((lane_o ^ 16) << 2, __float_as_int(t)));
          t += __int_as_float(__builtin_amdgcn_ds_bpermute((lane_o ^ 32) << 2, __float_as_int(t)));
          po[g][o] = t;
        }
        if (q_o == 0) lo_part[(wave * kGroups + g) * 16 + pt] = po[g];
      }
      wave_bad = __any(m_pos >= 0x497fe000 || m_neg >= 0xff800000u);
      __syncthreads();
      pf.mark(9);
      if (wave < kGroups) {      // wave w finishes column group w (its 16 points): the eight waves' shares
        const lds_f4 lp = lo_part + wave * 16 + pt;
        res = ((lp[0 * kGroups * 16] + lp[1 * kGroups * 16]) + (lp[2 * kGroups * 16] + lp[3 * kGroups * 16])) +
              ((lp[4 * kGroups * 16] + lp[5 * kGroups * 16]) + (lp[6 * kGroups * 16] + lp[7 * kGroups * 16]));
      }
      pf.mark(10);
      tile_next_v = tile_nx;
    }
    if (pa.overflow && wave_bad && lane_here() == 0) *pa.overflow = 1;
    if (wave < kGroups) {
      res += *reinterpret_cast<const f32x4*>(bpost + 4 * kHidden + 4 * q_o);     // lin_out bias kept at scale 1
      const long long t16 = tile * 4 + wave;
      const long long p = t16 * kPtsPerWave + pt;
      if (t16 < n_t16 && q_o == 0 && p < pa.P) {
        const float probe = (res[0] - res[0]) + (res[1] - res[1]) + (res[2] - res[2]) + (res[3] - res[3]);   // 0 or NaN
        if (pa.overflow && (probe != 0.0f || wave_bad)) *pa.overflow = 1;
        if (!pa.raw) {
          res[0] = 1.0f / (1.0f + expf(-res[0]));
          res[1] = 1.0f / (1.0f + expf(-res[1]));
          res[2] = 1.0f / (1.0f + expf(-res[2]));
          res[3] = fmaxf(res[3], 0.0f);
        }
        reinterpret_cast<f32x4*>(pa.out)[p] = res;
      }
    }
    pf.mark(11);
    // (the partial sums of this tile are read before the next tile's can be written: two barriers of the next tile's first block lie between)
    tile = tile_next_v;
    par ^= 1;
  }
  pf.end(a.prof, lane);
}
}  // namespace w8

// layer packing: [w 4][ts KT][mo 8][hl 2][lane 64][8]: W[128 w + 16 mo + (lane&15)][32 t(w, ts) + 16 (j>>2) + 4 (lane>>4) + (j&3)] * scale
__global__ void k_pack_layer_h3n(const float* __restrict__ W, int rows, int cols, int KT, float scale,
                                 _Float16* __restrict__ dst) {
  const long long total = (long long)4 * KT * 8192;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63, hl = (i >> 9) & 1, mo = (i >> 10) & 7;
    const int wt = (int)(i >> 13), ts = wt % KT, w = wt / KT;
    // position ts in wave w's weight stream -> k32 block of the contraction: 512-wide layers are walked own chunk first (chunk_of)
    const int t = KT == 16 ? 4 * ((ts >> 2) == 0 ? w : ((ts >> 2) - 1 < w ? (ts >> 2) - 1 : (ts >> 2))) + (ts & 3) : ts;
    const int row = 128 * w + 16 * mo + (lane & 15);
    const int col = 32 * t + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
    const float x = (row < rows && col < cols) ? W[(size_t)row * cols + col] * scale : 0.0f;
    const _Float16 h = (_Float16)x;
    dst[i] = hl ? (_Float16)(x - (float)h) : h;
  }
}

// lin_out fragments [t 16][hl 2][lane 64][8]: Wout[lane&15][32 t + 16 (j>>2) + 4 (lane>>4) + (j&3)] * scale, rows >= d_out zero
__global__ void k_pack_lin_out_h3n(const float* __restrict__ W, int rows, int cols, float scale, _Float16* __restrict__ dst) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 16384; i += gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63, hl = (i >> 9) & 1, t = i >> 10;
    const int row = lane & 15, col = 32 * t + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
    const float w = (row < rows && col < cols) ? W[(size_t)row * cols + col] * scale : 0.0f;
    const _Float16 h = (_Float16)w;
    dst[i] = hl ? (_Float16)(w - (float)h) : h;
  }
}
// lin_out for the vector ALU: [wave 4][mo 8][q 4][o 4][j 4] = Wout[o][128 wave + 16 mo + 4 q + j] * scale (the accumulators carry x16)
__global__ void k_pack_lin_out_valu(const float* __restrict__ W, int rows, int cols, float scale, float* __restrict__ dst) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2048; i += gridDim.x * blockDim.x) {
    const int j = i & 3, o = (i >> 2) & 3, q = (i >> 4) & 3, mo = (i >> 6) & 7, w = i >> 9;
    const int col = 128 * w + 16 * mo + 4 * q + j;
    dst[i] = (o < rows && col < cols) ? W[(size_t)o * cols + col] * scale : 0.0f;
  }
}
__global__ void k_scale_pad(const float* __restrict__ src, int n, int n_pad, float scale, float* __restrict__ dst) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += gridDim.x * blockDim.x)
    dst[i] = i < n ? src[i] * scale : 0.0f;
}

}  // namespace h3n

// w: lin_in + 6 per-view + 4 post layers (n-split fragments); w_out: lin_out fragments; b_pre: 7 x 512 (x16); b_post: 4 x 512
// (x16) + the lin_out bias at scale 1 (padded to 16).  The caller frees whatever was allocated when this fails.
static size_t h3n_halfs4() { return (size_t)4 * 2 * 8192 + (size_t)(6 + 4) * 4 * 16 * 8192; }      // lin_in, 6 per-view layers, 4 post layers
int h3n_alloc(float** w_out, float** w_lin_out, float** b_pre, float** b_post) {
  using namespace h3n;
  const size_t halfs4 = h3n_halfs4();
  const size_t halfs8 = w8::kLinInHalfs8 + 6 * w8::kLayerHalfs8;                    // the per-view layers in the 8-wave kernels' order: hi plane,
  const size_t halfs8x = w8::kLinInHalfs8x + 6 * w8::kLayerHalfs8x;                 // ... hi + lo planes,
  const size_t halfs = halfs4 + halfs8 + halfs8x + 4 * w8::kLayerHalfs8;             // and the four post layers (hi plane)
  DINER_HIP_OK(hipMalloc(w_out, halfs * sizeof(_Float16)));
  DINER_HIP_OK(hipMalloc(w_lin_out, (size_t)16384 * sizeof(_Float16) + kLinOutWBytes));      // MFMA fragments + the fp32 pack of the vector-ALU lin_out
  DINER_HIP_OK(hipMalloc(b_pre, 7 * kHidden * sizeof(float)));
  DINER_HIP_OK(hipMalloc(b_post, (5 * kHidden + 16) * sizeof(float)));
  return 0;
}
// train_only: the four-wave layouts, the lin_out packs and the biases (what k_train_fwd_pre / k_train_fwd_post read); the eight-wave
// layouts keep their old contents
int h3n_pack(const DinerMlpParams* p, hipStream_t stream, float* w_out_, float* w_lin_out_, float* b_pre_, float* b_post_, bool train_only) {
  using namespace h3n;
  float** w_out = &w_out_;
  float** w_lin_out = &w_lin_out_;
  float** b_pre = &b_pre_;
  float** b_post = &b_post_;
  const size_t halfs4 = h3n_halfs4();
  const size_t halfs8 = w8::kLinInHalfs8 + 6 * w8::kLayerHalfs8;
  const size_t halfs8x = w8::kLinInHalfs8x + 6 * w8::kLayerHalfs8x;
  auto bias = [&](const float* b, int n, int n_pad, float scale, float* dst) {
    hipLaunchKernelGGL(k_scale_pad, dim3(4), dim3(256), 0, stream, b, n, n_pad, scale, dst);
  };
  bias(p->lin_in_b, kHidden, kHidden, kScale, *b_pre);
  for (int b = 0; b < 3; ++b) {
    bias(p->fc0_b[b], kHidden, kHidden, kScale, *b_pre + kHidden * (1 + 2 * b));
    // the fc_1 biases are not added by the per-view kernel: those of blocks 0 and 1 travel in the next block's projected map (see
    // mlp_pack), block 2's is added by the post kernel to the view mean it takes over (the mean of x + b is mean(x) + b)
    bias(p->fc1_b[b], kHidden, kHidden, 0.0f, *b_pre + kHidden * (2 + 2 * b));
  }
  for (int b = 3; b < 5; ++b) {
    bias(p->fc0_b[b], kHidden, kHidden, kScale, *b_post + 2 * kHidden * (b - 3));
    bias(p->fc1_b[b], kHidden, kHidden, kScale, *b_post + 2 * kHidden * (b - 3) + kHidden);
  }
  bias(p->lin_out_b, 4, 16, 1.0f, *b_post + 4 * kHidden);
  bias(p->fc1_b[2], kHidden, kHidden, kScale, *b_post + 4 * kHidden + 16);
  hipLaunchKernelGGL(k_pack_lin_out_h3n, dim3(64), dim3(256), 0, stream, p->lin_out_w, 4, kHidden, kScale,
                     (_Float16*)*w_lin_out);
  hipLaunchKernelGGL(k_pack_lin_out_valu, dim3(8), dim3(256), 0, stream, p->lin_out_w, 4, kHidden, kInvScale,
                     reinterpret_cast<float*>(reinterpret_cast<char*>(*w_lin_out) + 32768));
  _Float16* wp = (_Float16*)*w_out;
  hipLaunchKernelGGL(k_pack_layer_h3n, dim3(256), dim3(256), 0, stream, p->lin_in_w, kHidden, kDIn, 2, kScale, wp);
  wp += (size_t)4 * 2 * 8192;
  for (int b = 0; b < 3; ++b) {
    hipLaunchKernelGGL(k_pack_layer_h3n, dim3(512), dim3(256), 0, stream, p->fc0_w[b], kHidden, kHidden, 16, kScale, wp);
    wp += (size_t)4 * 16 * 8192;
    hipLaunchKernelGGL(k_pack_layer_h3n, dim3(512), dim3(256), 0, stream, p->fc1_w[b], kHidden, kHidden, 16, kScale, wp);
    wp += (size_t)4 * 16 * 8192;
  }
  for (int b = 3; b < 5; ++b) {
    hipLaunchKernelGGL(k_pack_layer_h3n, dim3(512), dim3(256), 0, stream, p->fc0_w[b], kHidden, kHidden, 16, kScale, wp);
    wp += (size_t)4 * 16 * 8192;
    hipLaunchKernelGGL(k_pack_layer_h3n, dim3(512), dim3(256), 0, stream, p->fc1_w[b], kHidden, kHidden, 16, kScale, wp);
    wp += (size_t)4 * 16 * 8192;
  }
  if (train_only) {
    DINER_LAUNCH_OK();
    return 0;
  }
  {
    _Float16* w8p = (_Float16*)*w_out + halfs4;
    hipLaunchKernelGGL(w8::k_pack_layer_h8, dim3(64), dim3(256), 0, stream, p->lin_in_w, kHidden, kDIn, 2, kScale, w8p);
    w8p += w8::kLinInHalfs8;
    for (int b = 0; b < 3; ++b) {
      hipLaunchKernelGGL(w8::k_pack_layer_h8, dim3(512), dim3(256), 0, stream, p->fc0_w[b], kHidden, kHidden, 16, kScale, w8p);
      w8p += w8::kLayerHalfs8;
      hipLaunchKernelGGL(w8::k_pack_layer_h8, dim3(512), dim3(256), 0, stream, p->fc1_w[b], kHidden, kHidden, 16, kScale, w8p);
      w8p += w8::kLayerHalfs8;
    }
  }
  {
    _Float16* wx = (_Float16*)*w_out + halfs4 + halfs8;
    hipLaunchKernelGGL(w8::k_pack_layer_h8x, dim3(64), dim3(256), 0, stream, p->lin_in_w, kHidden, kDIn, 2, kScale, wx);
    wx += w8::kLinInHalfs8x;
    for (int b = 0; b < 3; ++b) {
      hipLaunchKernelGGL(w8::k_pack_layer_h8x, dim3(512), dim3(256), 0, stream, p->fc0_w[b], kHidden, kHidden, 16, kScale, wx);
      wx += w8::kLayerHalfs8x;
      hipLaunchKernelGGL(w8::k_pack_layer_h8x, dim3(512), dim3(256), 0, stream, p->fc1_w[b], kHidden, kHidden, 16, kScale, wx);
      wx += w8::kLayerHalfs8x;
    }
  }
  {
    _Float16* wq = (_Float16*)*w_out + halfs4 + halfs8 + halfs8x;
    for (int b = 3; b < 5; ++b) {
      hipLaunchKernelGGL(w8::k_pack_layer_h8, dim3(512), dim3(256), 0, stream, p->fc0_w[b], kHidden, kHidden, 16, kScale, wq);
      wq += w8::kLayerHalfs8;
      hipLaunchKernelGGL(w8::k_pack_layer_h8, dim3(512), dim3(256), 0, stream, p->fc1_w[b], kHidden, kHidden, 16, kScale, wq);
      wq += w8::kLayerHalfs8;
    }
  }
  DINER_LAUNCH_OK();
  return 0;
}
int h3n_set_attributes() {
  DINER_HIP_OK(hipFuncSetAttribute((const void*)h3n::w8::k_field_post_h8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h3n::w8::kLdsBytesPost8));
  DINER_HIP_OK(hipFuncSetAttribute((const void*)h3n::w8::k_field_pre_h8x, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h3n::w8::kLdsBytes8x));
  DINER_HIP_OK(hipFuncSetAttribute((const void*)h3n::w8::k_field_pre_h8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h3n::w8::kLdsBytes8));
  DINER_HIP_OK(hipFuncSetAttribute((const void*)h3n::k_train_fwd_pre, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h3n::kLdsBytes));
  DINER_HIP_OK(hipFuncSetAttribute((const void*)h3n::k_train_fwd_post, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h3n::kLdsBytesPost));
  DINER_HIP_OK(hipFuncSetAttribute((const void*)h3n::k_field_pre_h3n<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)h3n::kLdsBytes));
  DINER_HIP_OK(hipFuncSetAttribute((const void*)h3n::k_field_post_h3n<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)h3n::kLdsBytes));
  DINER_HIP_OK(hipFuncSetAttribute((const void*)h3n::k_field_pre_h3n<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)h3n::kLdsBytes));
  DINER_HIP_OK(hipFuncSetAttribute((const void*)h3n::k_field_post_h3n<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)h3n::kLdsBytes));
  return 0;
}
// split = true: f16x3 split products (hi and lo parts, three MFMAs per product); false: plain fp16 operands
void h3n_launch_pre(const SceneDev& sc, const FieldArgs& fa, const float* w, const float* b, int grid, bool split,
                    unsigned* tile_counter, hipStream_t stream, const SaveActs* sv) {
  const _Float16* w8p = (const _Float16*)w + ((size_t)4 * 2 * 8192 + (size_t)(6 + 4) * 4 * 16 * 8192);
  // DINER_F16_W8=0: the plain-fp16 mode on the 4-wave kernel (A/B measurement aid)
  static const bool use_w8 = [] { const char* e = getenv("DINER_F16_W8"); return !(e && *e == '0'); }();
  const _Float16* w8xp = w8p + (h3n::w8::kLinInHalfs8 + 6 * h3n::w8::kLayerHalfs8);
  // DINER_F16X3_W8=1: the f16x3 mode on the eight-wave kernel (round 5 experiment; default off until measured)
  static const bool use_w8x = [] { const char* e = getenv("DINER_F16X3_W8"); return e && *e == '1'; }();
  h3n::Args a{fa, (const _Float16*)w, w8p, w8xp, b, nullptr, tile_counter,
              h3n::QueueMap::make((fa.P + kPtsPerWave - 1) / kPtsPerWave, fa.K, fa.rays != nullptr && fa.xyz == nullptr && fa.direct_feat == nullptr)};
#ifdef DINER_HN_PROF
  static unsigned long long* prof = nullptr;
  if (!prof) hipMalloc(&prof, 32 * sizeof(unsigned long long));
  hipMemsetAsync(prof, 0, 32 * sizeof(unsigned long long), stream);
  a.prof = prof;
#endif
  if (sv) {          // training forward: the f16x3 kernel storing the pre-activations
    hipLaunchKernelGGL(h3n::k_train_fwd_pre, dim3(grid), dim3(256), h3n::kLdsBytes, stream, sc, a, *sv);
  } else if (split && use_w8x) hipLaunchKernelGGL(h3n::w8::k_field_pre_h8x, dim3(grid), dim3(512), h3n::w8::kLdsBytes8x, stream, sc, a);
  else if (split) hipLaunchKernelGGL(h3n::k_field_pre_h3n<true>, dim3(grid), dim3(256), h3n::kLdsBytes, stream, sc, a);
  else if (use_w8) hipLaunchKernelGGL(h3n::w8::k_field_pre_h8, dim3(grid), dim3(512), h3n::w8::kLdsBytes8, stream, sc, a);
  else hipLaunchKernelGGL(h3n::k_field_pre_h3n<false>, dim3(grid), dim3(256), h3n::kLdsBytes, stream, sc, a);
#ifdef DINER_HN_PROF
  unsigned long long h[32];
  static const char* fnames[4] = {"  load_point", "  project + depth tap", "  features", "  bilinear taps"};
  hipStreamSynchronize(stream);
  hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
  static const char* names[15] = {"frontend", "sync", "lin_in publish", "sync", "lin_in gemm", "gather 0", "fc_0 sync A", "fc_0 publish",
                                  "fc_0 sync B", "fc_0 gemm", "fc_1 sync A", "fc_1 publish", "fc_1 sync B", "fc_1 gemm(+gather)", "store"};
  const double waves = (double)h[26], tot = (double)h[24];
  fprintf(stderr, "[h3n prof] waves %.0f  clocks/wave %.0f  shader MHz %.0f  (P=%lld)\n", waves, tot / waves,
          tot / ((double)h[25] / 100.0), fa.P);
  for (int i = 0; i < 15; ++i) fprintf(stderr, "[h3n prof]   %-20s %6.2f %%  %10.0f clk/wave\n", names[i], 100.0 * h[i] / tot, h[i] / waves);
  for (int i = 15; i < 19; ++i)
    if (h[i]) fprintf(stderr, "[h3n prof]   %-20s %6.2f %%  %10.0f clk/wave\n", fnames[i - 15], 100.0 * h[i] / tot, h[i] / waves);
#endif
}

// w: the n-split pack (post layers follow the per-view ones); w_lin_out: the lin_out fragments
void h3n_launch_post(const PostArgs& pa, const float* w, const float* w_lin_out, int grid, bool split, unsigned* tile_counter,
                     hipStream_t stream, const SaveActs* sv) {
  const _Float16* wn = (const _Float16*)w + (size_t)4 * 2 * 8192 + (size_t)6 * 4 * 16 * 8192;
  const _Float16* wo = (const _Float16*)w_lin_out;
  const long long n_t16 = (pa.P + kPtsPerWave - 1) / kPtsPerWave;
  const _Float16* w8post = (const _Float16*)w + ((size_t)4 * 2 * 8192 + (size_t)(6 + 4) * 4 * 16 * 8192) + (h3n::w8::kLinInHalfs8 + 6 * h3n::w8::kLayerHalfs8) +
                           (h3n::w8::kLinInHalfs8x + 6 * h3n::w8::kLayerHalfs8x);
  // DINER_F16_W8=0 / DINER_F16_POST_W8=0: the plain-fp16 post kernel on four waves (A/B measurement aids)
  static const bool use_w8 = [] { const char* e = getenv("DINER_F16_W8"); const char* f = getenv("DINER_F16_POST_W8"); return !(e && *e == '0') && !(f && *f == '0'); }();
  h3n::PostArgsN a{pa, wn, w8post, wo, nullptr, tile_counter, h3n::QueueMap::make((n_t16 + 3) / 4, 0, false)};
#ifdef DINER_HN_PROF
  static unsigned long long* prof = nullptr;
  if (!prof) hipMalloc(&prof, 32 * sizeof(unsigned long long));
  hipMemsetAsync(prof, 0, 32 * sizeof(unsigned long long), stream);
  a.prof = prof;
#endif
  if (sv) {
    hipLaunchKernelGGL(h3n::k_train_fwd_post, dim3(grid), dim3(256), h3n::kLdsBytesPost, stream, a, *sv);
  } else if (split) hipLaunchKernelGGL(h3n::k_field_post_h3n<true>, dim3(grid), dim3(256), h3n::kLdsBytesPost, stream, a);
  else if (use_w8) hipLaunchKernelGGL(h3n::w8::k_field_post_h8, dim3(grid), dim3(512), h3n::w8::kLdsBytesPost8, stream, a);
  else hipLaunchKernelGGL(h3n::k_field_post_h3n<false>, dim3(grid), dim3(256), h3n::kLdsBytesPost, stream, a);
#ifdef DINER_HN_PROF
  unsigned long long h[32];
  hipStreamSynchronize(stream);
  hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
  static const char* names[12] = {"fc_0 sync A", "fc_0 publish", "fc_0 sync B", "fc_0 gemm", "fc_1 sync A", "fc_1 publish", "fc_1 sync B",
                                  "fc_1 gemm", "hand-over load", "lin_out publish", "lin_out MFMAs", "epilogue"};
  const double waves = (double)h[26], tot = (double)h[24];
  fprintf(stderr, "[h3n prof post] waves %.0f  clocks/wave %.0f  shader MHz %.0f  (P=%lld)\n", waves, tot / waves,
          tot / ((double)h[25] / 100.0), pa.P);
  for (int i = 0; i < 12; ++i) fprintf(stderr, "[h3n prof post]   %-20s %6.2f %%  %10.0f clk/wave\n", names[i], 100.0 * h[i] / tot, h[i] / waves);
#endif
}

}  // namespace diner
