// Interface of train_512.hip (the 512 x 512 layer products of the training path: train_lin512.hip + train_wgrad512.hip) for train.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace diner {
namespace train {

enum : int { kL512ReluIn = 1, kL512Accum = 2 };
constexpr size_t kL512PackBytes = (size_t)512 * 512 * 3 * 2;      // one packed weight matrix: three bf16 planes

struct Lin512Args {
  const float* X;        // (M, ldx): rows of 512 contraction values
  const void* Wp;        // packed weights (lin512_pack)
  float* Y;              // (M, ldy), 512 outputs per row
  const float* bias;     // 512 or null
  const float* resid;    // (M, ldy) or null
  const float* mask;     // (M, ldy) or null: Y = 0 where mask <= 0
  long long M;
  int ldx, ldy, flags;
  int* ovf;              // f16x3 arithmetic: raised when a staged operand leaves the fp16 range (null: not reported)
  const int* gate;       // null, or: the launch does nothing unless *gate != 0 (the bf16x6 fall-back behind an f16x3 product)
  const float* resid2;   // (M, ldy) or null: a second residual (the next block's lin_z term, which then needs no accumulating product)
  // round 4 -- the f16x3 arithmetic on operands that are not activations (the data gradient dx = dy W: loss gradients span many decades):
  const unsigned* amax_in;   // f16x3: bit pattern of max |X| over the whole operand (a producer's amax_out): X is staged times the power of two
                             // that brings that maximum into [2^14, 2^15) -- nothing leaves the fp16 range, and entries more than 2^39 below the
                             // maximum lose their low part (far below fp32 round-off of any sum the maximum takes part in); null: no scaling
  unsigned* amax_out;        // null, or: atomic maximum of the bit patterns of |Y| as stored (zeroed by the caller): the next product's amax_in
  const int* skip;           // null, or: the launch does nothing when *skip != 0 (an f16x3 product whose weights do not fit 16 w in fp16)
  // round 4 -- a second contraction segment: Y = act(X) Wp^T + act2(X2) Wp2^T + ... (K = 512 + 512 in one pass over the rows; the forward
  // uses it for fc_1 of block b together with lin_z of block b + 1, whose sum is the next residual stream: no Z tensor written and read back)
  const float* X2;           // (M, ldx) or null; same row stride as X
  const void* Wp2;           // packed weights of the second segment (same pack mode as Wp)
  const float* bias2;        // 512 or null
  int relu2;                 // relu on the second segment's operand
  int* ovf2;                 // f16x3: a second flag raised together with ovf (the weight gradient of the fused layer looks at its own slot)
  const unsigned* maskbits;  // null, or (instead of mask) the relu decisions of the saved pre-activation as bits: 16 dwords per row (round 5;
                             // written by the forward -- save_block of mlp_h3n.hip / k_make_bits of train.hip: feature f of the row sits in
                             // dword 4 (f / 128) + (f % 16) / 4 at bit 4 ((f % 128) / 16) + f % 4); Y = 0 where the bit is down
  // round 6 -- a product over a row list whose length only the device knows (the latent rows a training batch touches, train.hip):
  const int* m_dev;          // null, or: the number of rows is min(*m_dev, M) (M = the capacity the launch was planned for)
  const int* skip_silent;    // null, or: the launch does nothing when *skip_silent != 0 (unlike `skip`, no flag is raised)
  const int* gate2;          // null, or a second condition like gate (round 5: the layer-wise forward as the repeat behind the fused training forward --
                             // its bf16x6 twins run only if the fused kernels left the range AND their own f16x3 product did)
};

// W (512, 512) row-major fp32 -> packed planes; transpose = 0: y = x W^T (W as nn.Linear stores it), 1: y = x W
int lin512_pack(const float* W, int mode, void* dst, hipStream_t stream);
// n <= 13 weight matrices in one launch: W[i] -> base + (13 m + i) * kL512PackBytes for the pack modes m < modes (0 forward, 1 transposed,
// 2 forward / 3 transposed as fp16 hi / lo of 16 W for the f16x3 arithmetic).  wbad (modes > 2; zeroed by the caller): set to 1 when some
// 16 |w| is not a finite fp16 value -- the f16x3 products of the step must then be skipped and their bf16x6 twins run (Lin512Args.skip / .gate)
struct PackMany { const float* W[13]; };
int lin512_pack_many(const PackMany& w, int n, void* base, hipStream_t stream, int modes = 2, int* wbad = nullptr);
int lin512_launch(const Lin512Args& a, hipStream_t stream, int arith = 0);      // arith 1: f16x3 (Wp packed with mode 2; a.ovf reports operands out of range)
// train_wgrad512.hip: dW (512, 512) += dY^T act(X), db (512, or null) += column sums of dY over M rows.  part = null: atomics into dW / db
// (both zeroed by the caller); part = wgrad512_part_bytes() of scratch: per-chunk partial tiles + one summing pass, which overwrites
// dW / db instead of adding to them when overwrite is set
// defer: with `part`, leave the summing pass to the caller -- *defer describes it; wgrad512_reduce_many runs up to 13 of them in one launch
struct WgReduceJob { const float* part; float* dW; float* db; int n_chunks; };
struct WgReduceJobs { WgReduceJob job[13]; };
// arith 1: both parts in the f16x3 arithmetic -- dY staged times the power of two from *amax_dy (see Lin512Args.amax_in), X as it is (an
// activation: in range when the forward product that consumed it raised no flag); the launch's weight-gradient part does nothing when
// *wg_skip != 0 (arith 1) / unless *wg_gate != 0 (arith 0): the caller issues both launches and exactly one of them works
struct WgradArith { int arith; const unsigned* amax_dy; const int* wg_skip; const int* wg_gate; };
int wgrad512_launch(const float* dY, int ldy, const float* X, int ldx, bool relu_x, float* dW, float* db, long long M,
                    hipStream_t stream, float* part = nullptr, bool overwrite = false, WgReduceJob* defer = nullptr,
                    const Lin512Args* dgrad = nullptr,       // dgrad: the data-gradient product of the same layer in the same launch
                    const WgradArith* ar = nullptr);
// round 6: lin_in's weight / bias gradient on a kernel of its own (train_wgrad512.hip: k_wgrad_in_f16x3): dW (512, n_in <= 64) += dY^T F, db (512) +=
// column sums of dY over M rows; dW / db zeroed by the caller; dY staged times the power of two from *amax_dy (null: unscaled)
int wgrad_in_launch(const float* dY, int ldy, const float* F, int ldf, int n_in, long long M, float* dW, float* db, const unsigned* amax_dy,
                    hipStream_t stream);
int wgrad512_reduce_many(const WgReduceJobs& jobs, int n, bool overwrite, hipStream_t stream);
size_t wgrad512_part_bytes();

}  // namespace train
}  // namespace diner
