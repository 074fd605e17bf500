// gemv.hip -- Y[M, N] = X[M, K] . W[N, K]^T for M <= 8: the projections of a cached decode step (one new token per sequence:
// models/llama/modeling_llama.py:254-256, 280, 174-176 and the lm_head, :480, with `hidden_states` of shape [batch, 1, hidden]).
//
// Such a product reads every weight once and does 2*M flops per weight element: it is bound by HBM, not by the matrix pipe, and
// a 256 x 256 MFMA tile serves it badly twice over -- N / 256 workgroups (16 for a 4096-row projection) cannot pull 8 TB/s, and
// 255 of the tile's 256 rows of A are padding.  This kernel streams W instead: one wave owns R consecutive weight rows, its 64
// lanes walk K in 16-byte pieces (1 KiB per row and step, coalesced), multiply against the M input rows (16-byte loads that
// hit L1 / L2: X is a few KiB) with fp32 accumulators on the VALU, and a xor-butterfly adds the lanes up at the end.  No LDS,
// no MFMA.  R = 2 or 4 by N so that every CU holds several waves; N / (4 R) workgroups of 4 waves.
// HBM traffic = the algorithmic bytes: N*K*2 (+ M*(K + N)*2).
// Roundings as in the GEMM kernels' epilogues: round(acc [+ bias]), then -- residual -- round(that + R).  The summation order
// differs from the MFMA kernels' (per-lane partial sums over K, then the butterfly): results agree to fp32 summation error.
#include "gemv.h"

namespace tamd {

template <typename T, int MB, int R, int EPI>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs g) {
  typedef typename elem<T>::raw raw;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n0 = ((int64_t)blockIdx.x * 4 + wave) * R;
  if (n0 >= g.N) return;
  const T* __restrict__ W = reinterpret_cast<const T*>(g.W);
  const T* __restrict__ X = reinterpret_cast<const T*>(g.X);
  const T* wrow[R];
  const T* xrow[MB];
#pragma unroll
  for (int r = 0; r < R; ++r) wrow[r] = W + (n0 + r < g.N ? n0 + r : g.N - 1) * g.ldw;  // (rows past N: re-read the last one)
#pragma unroll
  for (int m = 0; m < MB; ++m) xrow[m] = X + (m < g.M ? m : g.M - 1) * g.ldx;
  float acc[MB][R];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int r = 0; r < R; ++r) acc[m][r] = 0.f;
#pragma unroll 2
  for (int64_t k = (int64_t)lane * 8; k < g.K; k += 512) {
    u32x4 wq[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wq[r] = ld16(wrow[r] + k);
    float xv[MB][8];
#pragma unroll
    for (int m = 0; m < MB; ++m) unpack16<T>(ld16(xrow[m] + k), xv[m]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float wv[8];
      unpack16<T>(wq[r], wv);
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[m][r] = fmaf(wv[e], xv[m][e], acc[m][r]);
    }
  }
  // every lane ends up with every total; lane m*R + r stores output (m, n0 + r)
  float mine = 0.f;
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float t = wave_sum(acc[m][r]);
      if (lane == m * R + r) mine = t;
    }
  const int m = lane / R, r = lane % R;
  const int64_t n = n0 + r;
  if (lane < MB * R && m < g.M && n < g.N) {
    float v = mine;
    if ((EPI == TAMD_EPI_BIAS || EPI == TAMD_EPI_RESIDUAL) && g.bias != nullptr)
      v += elem<T>::to_f32(reinterpret_cast<const raw*>(g.bias)[n]);
    if (EPI == TAMD_EPI_RESIDUAL)
      v = round_through<T>(v) + elem<T>::to_f32(reinterpret_cast<const raw*>(g.R)[m * g.ldr + n]);
    reinterpret_cast<raw*>(g.Y)[m * g.ldy + n] = elem<T>::from_f32(v);
  }
}

template <typename T, int MB, int R>
static int gemv_launch_epi(const GemvArgs& g, int epilogue, hipStream_t s) {
  dim3 grid((unsigned)ceil_div(g.N, 4 * R)), block(256);
  switch (epilogue) {
    case TAMD_EPI_NONE: hipLaunchKernelGGL((gemv_kernel<T, MB, R, TAMD_EPI_NONE>), grid, block, 0, s, g); break;
    case TAMD_EPI_BIAS: hipLaunchKernelGGL((gemv_kernel<T, MB, R, TAMD_EPI_BIAS>), grid, block, 0, s, g); break;
    case TAMD_EPI_RESIDUAL: hipLaunchKernelGGL((gemv_kernel<T, MB, R, TAMD_EPI_RESIDUAL>), grid, block, 0, s, g); break;
    default: return TAMD_E_ARG;
  }
  return launch_status();
}

template <typename T, int MB>
static int gemv_launch_rows(const GemvArgs& g, int epilogue, hipStream_t s) {
  // enough waves for every CU to hold several: 256 CUs x 4 SIMDs; 2 rows per wave below 16384 weight rows
  if (g.N >= 16384) return gemv_launch_epi<T, MB, 4>(g, epilogue, s);
  return gemv_launch_epi<T, MB, 2>(g, epilogue, s);
}

template <typename T>
static int gemv_launch(const GemvArgs& g, int epilogue, hipStream_t s) {
  if (g.M <= 1) return gemv_launch_rows<T, 1>(g, epilogue, s);
  if (g.M <= 2) return gemv_launch_rows<T, 2>(g, epilogue, s);
  if (g.M <= 4) return gemv_launch_rows<T, 4>(g, epilogue, s);
  return gemv_launch_rows<T, 8>(g, epilogue, s);
}

// gate|up product + SiLU(gate) * up: a wave owns R FEATURES, i.e. weight rows i (gate) and I + i (up); the expression is
// swiglu_fwd_kernel's (elementwise.hip) on the rounded gate / up, so the result is the two-launch path's bit for bit
template <typename T, int MB, int R>
__global__ __launch_bounds__(256) void gemv_swiglu_kernel(GemvArgs g) {
  typedef typename elem<T>::raw raw;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t inter = g.N;
  const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * R;
  if (i0 >= inter) return;
  const T* __restrict__ W = reinterpret_cast<const T*>(g.W);
  const T* __restrict__ X = reinterpret_cast<const T*>(g.X);
  const T* wrow[2 * R];
  const T* xrow[MB];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t i = i0 + r < inter ? i0 + r : inter - 1;
    wrow[r] = W + i * g.ldw;
    wrow[R + r] = W + (inter + i) * g.ldw;
  }
#pragma unroll
  for (int m = 0; m < MB; ++m) xrow[m] = X + (m < g.M ? m : g.M - 1) * g.ldx;
  float acc[MB][2 * R];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int r = 0; r < 2 * R; ++r) acc[m][r] = 0.f;
#pragma unroll 2
  for (int64_t k = (int64_t)lane * 8; k < g.K; k += 512) {
    u32x4 wq[2 * R];
#pragma unroll
    for (int r = 0; r < 2 * R; ++r) wq[r] = ld16(wrow[r] + k);
    float xv[MB][8];
#pragma unroll
    for (int m = 0; m < MB; ++m) unpack16<T>(ld16(xrow[m] + k), xv[m]);
#pragma unroll
    for (int r = 0; r < 2 * R; ++r) {
      float wv[8];
      unpack16<T>(wq[r], wv);
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[m][r] = fmaf(wv[e], xv[m][e], acc[m][r]);
    }
  }
  float gate = 0.f, up = 0.f;
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float tg = wave_sum(acc[m][r]), tu = wave_sum(acc[m][R + r]);
      if (lane == m * R + r) gate = tg, up = tu;
    }
  const int m = lane / R, r = lane % R;
  const int64_t i = i0 + r;
  if (lane < MB * R && m < g.M && i < inter) {
    const float gr = round_through<T>(gate), ur = round_through<T>(up);
    const float sg = 1.f / (1.f + __expf(-gr));  // = silu_f of elementwise.hip
    reinterpret_cast<raw*>(g.Y)[m * g.ldy + i] = elem<T>::from_f32(round_through<T>(gr * sg) * ur);
    if (g.R != nullptr) {
      raw* gu = reinterpret_cast<raw*>(const_cast<void*>(g.R));
      gu[m * g.ldr + i] = elem<T>::from_f32(gr);
      gu[m * g.ldr + inter + i] = elem<T>::from_f32(ur);
    }
  }
}

template <typename T, int MB>
static int gemv_swiglu_launch_rows(const GemvArgs& g, hipStream_t s) {
  constexpr int R = 2;  // (2 features = 4 weight rows per wave)
  dim3 grid((unsigned)ceil_div(g.N, 4 * R)), block(256);
  hipLaunchKernelGGL((gemv_swiglu_kernel<T, MB, R>), grid, block, 0, s, g);
  return launch_status();
}
template <typename T>
static int gemv_swiglu_launch(const GemvArgs& g, hipStream_t s) {
  if (g.M <= 1) return gemv_swiglu_launch_rows<T, 1>(g, s);
  if (g.M <= 2) return gemv_swiglu_launch_rows<T, 2>(g, s);
  if (g.M <= 4) return gemv_swiglu_launch_rows<T, 4>(g, s);
  return gemv_swiglu_launch_rows<T, 8>(g, s);
}
int gemv_swiglu_run(const GemvArgs& g, int dtype, hipStream_t stream) {
  if (g.M < 1 || g.M > kGemvMaxRows || (g.K % 8) != 0) return TAMD_E_SHAPE;
  TAMD_DISPATCH_HALF(dtype, return (gemv_swiglu_launch<T>(g, stream)));
  return TAMD_E_DTYPE;
}

int gemv_run(const GemvArgs& g, int epilogue, int dtype, hipStream_t stream) {
  if (g.M < 1 || g.M > kGemvMaxRows || (g.K % 8) != 0) return TAMD_E_SHAPE;
  TAMD_DISPATCH_HALF(dtype, return (gemv_launch<T>(g, epilogue, stream)));
  return TAMD_E_DTYPE;
}

}  // namespace tamd
