// Interface of train_lin512.hip (the 512 x 512 layer products of the training path) for train.hip.
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
};

// W (512, 512) row-major fp32 -> packed planes; transpose = 0: y = x W^T (W as nn.Linear stores it), 1: y = x W
int lin512_pack(const float* W, int transpose, void* dst, hipStream_t stream);
int lin512_launch(const Lin512Args& a, hipStream_t stream);

}  // namespace train
}  // namespace diner
