// gemv.h -- the skinny product of a cached decode step: Y[M, N] = X[M, K] . W[N, K]^T with M <= kGemvMaxRows = 16 (gemv.hip).
#pragma once
#include "common.h"

namespace tamd {

constexpr int kGemvMaxRows = 16;     // rows of X these kernels take
constexpr int kGemvValuRows = 4;     // ... of which the VALU kernel takes up to this many, the MFMA kernel the rest

struct GemvArgs {
  const void* X;
  const void* W;
  void* Y;
  const void* bias;
  const void* R;
  int64_t M, N, K, ldx, ldw, ldy, ldr;
};
// epilogue: TAMD_EPI_NONE / TAMD_EPI_BIAS / TAMD_EPI_RESIDUAL (bias optional) with the GEMM kernels' roundings
int gemv_run(const GemvArgs& g, int epilogue, int dtype, hipStream_t stream);
// LlamaMLP's inner product for M <= kGemvMaxRows (modeling_llama.py:174-176): W = [gate rows (I) ; up rows (I)] [2I, K],
// ACT[M, I] = round(round(silu(round(g))) * round(u)) -- the bits of the plain product followed by tamd_swiglu_fwd -- and, when GU
// is not null, GU[M, 2I] = the rounded gate | up themselves.  Uses g.Y = ACT (ldy), g.R = GU (ldr; may be null), g.N = I.
int gemv_swiglu_run(const GemvArgs& g, int dtype, hipStream_t stream);

}  // namespace tamd
