// gemv.hip -- Y[M, N] = X[M, K] . W[N, K]^T for M <= 16: the projections of a cached decode step (one new token per sequence:
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

#include <stdlib.h>

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
  return gemv_launch_rows<T, 4>(g, epilogue, s);  // (kGemvValuRows)
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
    const float sg = fast_sigmoid(gr);  // = silu_f of elementwise.hip
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
  return gemv_swiglu_launch_rows<T, 4>(g, s);  // (kGemvValuRows)
}
// ---- 5 .. 16 rows: the same stream through the matrix pipe.  With 8 rows the VALU kernel above spends 352 instructions per
// 4 weight loads and ran the gate|up product at 2.2 TB/s (profiles/r04m_decode_kernel_stats.csv); one v_mfma_f32_16x16x32 eats
// 16 weight rows x 32 k (1 KiB) against up to 16 input rows.  A workgroup owns 16 weight rows (SWIGLU: 16 features = 16 gate +
// 16 up rows); its four waves split K and meet in LDS (fixed order).  Lane l loads W[row0 + (l & 15)][k + 8 (l >> 4) .. + 7] --
// 64 contiguous bytes per row and instruction, whole 128-byte lines over two k-steps -- and X[l & 15][the same k] (zero for
// rows past M).  Accumulator register r of lane l is y[m = l & 15][n = row0 + 4 (l >> 4) + r].
// RB blocks of 16 weight rows per workgroup share every X fragment: with 16 live rows X is as many bytes per k-step as one
// block of W, and re-read from L2 by every workgroup -- one block per workgroup ran the 128256-row lm_head at 3.75 TB/s with 16
// rows against 6.09 with one (profiles/r04r_gemv_bench.jsonl); large N takes RB = 2 / 4, small N keeps one block for the grid.
template <typename T, int EPI, bool SWIGLU, int RB>
__global__ __launch_bounds__(256) void gemv_mfma_kernel(GemvArgs g) {
  typedef typename elem<T>::raw raw;
  constexpr int NA = SWIGLU ? 2 * RB : RB;  // accumulators (gate blocks, then up blocks)
  __shared__ float red[4][NA][64][4];
  const int lane = threadIdx.x & 63, wave = wave_id_uniform();
  const int l15 = lane & 15, g4 = lane >> 4;
  const int64_t n0 = (int64_t)blockIdx.x * 16 * RB;
  const T* __restrict__ W = reinterpret_cast<const T*>(g.W);
  const T* __restrict__ X = reinterpret_cast<const T*>(g.X);
  const T* wp[NA];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int64_t row = n0 + rb * 16 + l15 < g.N ? n0 + rb * 16 + l15 : g.N - 1;
    wp[rb] = W + row * g.ldw + g4 * 8;
    if (SWIGLU) wp[RB + rb] = W + (g.N + row) * g.ldw + g4 * 8;
  }
  const bool xlive = l15 < g.M;
  const T* xr = X + (xlive ? l15 : 0) * g.ldx + g4 * 8;
  // this wave's K range: a quarter, in whole 32-element steps
  const int64_t steps = (g.K + 31) / 32, per = (steps + 3) / 4;
  const int64_t s0 = wave * per, s1 = (s0 + per < steps) ? s0 + per : steps;
  f32x4 acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // U k-steps per round: every load of the round is issued before its first MFMA (8 .. 32 KiB per wave in flight: with one
  // workgroup per CU -- a 4096-row projection is 256 of them -- that is what keeps HBM busy; four steps per round with the
  // loads left to the compiler ran o_proj / down_proj at 2.2 TB/s, profiles/r04q_decode_b8_kernel_stats.csv)
  constexpr int U = NA >= 4 ? 4 : 8;
  const u32x4 z = {0u, 0u, 0u, 0u};
  for (int64_t st = s0; st < s1; st += U) {
    u32x4 a[NA][U], xb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t k = (st + u) * 32;
      const bool in = st + u < s1 && k + g4 * 8 < g.K;  // (K % 8 == 0: a lane's 8 elements are inside or outside together)
#pragma unroll
      for (int i = 0; i < NA; ++i) a[i][u] = in ? ld16(wp[i] + k) : z;
      xb[u] = (in && xlive) ? ld16(xr + k) : z;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < NA; ++i) acc[i] = mfma16<T>(a[i][u], xb[u], acc[i]);
  }
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][i][lane][r] = acc[i][r];
  block_sync();
  if (wave != 0 || !xlive) return;
  const int m = l15;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t n = n0 + rb * 16 + 4 * g4 + r;
      if (n >= g.N) continue;
      float v = ((red[0][rb][lane][r] + red[1][rb][lane][r]) + red[2][rb][lane][r]) + red[3][rb][lane][r];
      if (SWIGLU) {
        constexpr int UP = SWIGLU ? RB : 0;
        const float u = ((red[0][UP + rb][lane][r] + red[1][UP + rb][lane][r]) + red[2][UP + rb][lane][r]) + red[3][UP + rb][lane][r];
        const float gr = round_through<T>(v), ur = round_through<T>(u);
        const float sg = fast_sigmoid(gr);  // = silu_f of elementwise.hip
        reinterpret_cast<raw*>(g.Y)[m * g.ldy + n] = elem<T>::from_f32(round_through<T>(gr * sg) * ur);
        if (g.R != nullptr) {
          raw* gu = reinterpret_cast<raw*>(const_cast<void*>(g.R));
          gu[m * g.ldr + n] = elem<T>::from_f32(gr);
          gu[m * g.ldr + g.N + n] = elem<T>::from_f32(ur);
        }
      } else {
        if ((EPI == TAMD_EPI_BIAS || EPI == TAMD_EPI_RESIDUAL) && g.bias != nullptr)
          v += elem<T>::to_f32(reinterpret_cast<const raw*>(g.bias)[n]);
        if (EPI == TAMD_EPI_RESIDUAL)
          v = round_through<T>(v) + elem<T>::to_f32(reinterpret_cast<const raw*>(g.R)[m * g.ldr + n]);
        reinterpret_cast<raw*>(g.Y)[m * g.ldy + n] = elem<T>::from_f32(v);
      }
    }
}

template <typename T, int RB>
static int gemv_mfma_launch_rb(const GemvArgs& g, int epilogue, bool swiglu, hipStream_t s) {
  dim3 grid((unsigned)ceil_div(g.N, 16 * RB)), block(256);
  if (swiglu) {
    hipLaunchKernelGGL((gemv_mfma_kernel<T, TAMD_EPI_NONE, true, (RB > 2 ? 2 : RB)>), grid, block, 0, s, g);
    return launch_status();
  }
  switch (epilogue) {
    case TAMD_EPI_NONE: hipLaunchKernelGGL((gemv_mfma_kernel<T, TAMD_EPI_NONE, false, RB>), grid, block, 0, s, g); break;
    case TAMD_EPI_BIAS: hipLaunchKernelGGL((gemv_mfma_kernel<T, TAMD_EPI_BIAS, false, RB>), grid, block, 0, s, g); break;
    case TAMD_EPI_RESIDUAL: hipLaunchKernelGGL((gemv_mfma_kernel<T, TAMD_EPI_RESIDUAL, false, RB>), grid, block, 0, s, g); break;
    default: return TAMD_E_ARG;
  }
  return launch_status();
}
template <typename T>
static int gemv_mfma_launch(const GemvArgs& g, int epilogue, bool swiglu, hipStream_t s) {
  // row blocks (SwiGLU: blocks of features, two weight rows each) per workgroup: as many as leave >= 512 workgroups
  const int64_t wgs1 = ceil_div(g.N, 16);
  if (wgs1 >= 4 * 512 && !swiglu) return gemv_mfma_launch_rb<T, 4>(g, epilogue, swiglu, s);
  if (wgs1 >= 2 * 512) return gemv_mfma_launch_rb<T, 2>(g, epilogue, swiglu, s);
  return gemv_mfma_launch_rb<T, 1>(g, epilogue, swiglu, s);
}
// rows up to which the VALU kernel runs (libtamd_diag.so: TAMD_GEMV_VALU_ROWS in the environment, A/B of the two kernels)
static int gemv_valu_rows() {
#ifdef TAMD_DIAG
  static const int v = [] {
    const char* e = getenv("TAMD_GEMV_VALU_ROWS");
    const int n = e ? atoi(e) : kGemvValuRows;
    return n < 0 ? 0 : (n > 4 ? 4 : n);
  }();
  return v;
#else
  return kGemvValuRows;
#endif
}

// Which kernel (profiles/r04s_gemv_bench*.jsonl, cold weights): one row -- the VALU kernel everywhere (q|k|v 12.8 vs 12.9 us,
// lm_head 174 vs 193); 2 .. 4 rows -- the VALU kernel for 16384 weight rows and more (gate|up 46 vs 50 us, lm_head 181 vs 199 at
// 4 rows), the MFMA kernel below (down_proj 28.6 vs 35.2, o_proj 10.7 vs 14.1); 5 .. 16 rows -- the MFMA kernel.
static bool gemv_use_valu(int64_t m, int64_t weight_rows) {
  return m <= gemv_valu_rows() && (m == 1 || weight_rows >= 16384);
}

int gemv_swiglu_run(const GemvArgs& g, int dtype, hipStream_t stream) {
  if (g.M < 1 || g.M > kGemvMaxRows || (g.K % 8) != 0) return TAMD_E_SHAPE;
  if (!gemv_use_valu(g.M, 2 * g.N)) {
    TAMD_DISPATCH_HALF(dtype, return (gemv_mfma_launch<T>(g, TAMD_EPI_NONE, true, stream)));
  }
  TAMD_DISPATCH_HALF(dtype, return (gemv_swiglu_launch<T>(g, stream)));
  return TAMD_E_DTYPE;
}

int gemv_run(const GemvArgs& g, int epilogue, int dtype, hipStream_t stream) {
  if (g.M < 1 || g.M > kGemvMaxRows || (g.K % 8) != 0) return TAMD_E_SHAPE;
  if (!gemv_use_valu(g.M, g.N)) {
    TAMD_DISPATCH_HALF(dtype, return (gemv_mfma_launch<T>(g, epilogue, false, stream)));
  }
  TAMD_DISPATCH_HALF(dtype, return (gemv_launch<T>(g, epilogue, stream)));
  return TAMD_E_DTYPE;
}

}  // namespace tamd
