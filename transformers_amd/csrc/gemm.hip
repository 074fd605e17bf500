// gemm.hip -- bf16/f16 MFMA GEMM for gfx950 with fused epilogues.
//
//   C[M,N] = epilogue( A[M,K] . B[N,K]^T )        fp32 accumulation on v_mfma_f32_32x32x16_{bf16,f16}
//
// replaces every nn.Linear on the hot path (reference call sites, src/transformers/):
//   models/llama/modeling_llama.py:254-256 (q/k/v), :280 (o_proj), :174-176 (gate/up/down),
//   models/bert/modeling_bert.py:175-177, :289, :334, :347, pytorch_utils.py:117-121 (GPT-2 Conv1D),
//   models/clip/modeling_clip.py:304-333, :346-350, models/llava/modeling_llava.py:102-106,
// and, through the two layout flags, both backward products of a linear layer
//   dX[M,K'] = dY[M,N'] . W[N',K']      -> TAMD_GEMM_B_KN  (B stored [K,N])
//   dW[N',K'] = dY[M,N']^T . X[M,K']    -> TAMD_GEMM_A_KM | TAMD_GEMM_B_KN (both stored k-major)
// so no operand is ever transposed through HBM.
//
// Structure (cdna_hip_programming.md §5, "glds, 2 LDS buffers, BK=64" tier):
//   * 256x256 output tile per 512-thread workgroup (8 waves as 2(M) x 4(N), 128x64 per wave,
//     8 accumulators of 32x32 = 128 acc registers), BK = 64, one barrier per K tile;
//   * both operand tiles stream HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip),
//     double buffered (2 x 64 KiB);
//   * LDS images are lane-linear (a glds requirement), so the bank-conflict swizzle is applied to
//     the per-lane SOURCE address and undone on the fragment read (guide rule 21):
//       k-contiguous operand  [256 rows][64 k]: 16-B slot' = slot ^ ((row>>1)&7)   -> ds_read_b128 conflict-free
//       k-major operand       [64 k][256 cols]: 16-B slot' = slot ^ ((k&3)<<2)     -> ds_read_b64_tr_b16 conflict-free
//   * the MFMA is issued "swapped" (A-operand = B/W fragment, B-operand = A/X fragment) so each lane
//     ends up with 4 consecutive output columns of one output row; the epilogue rounds to the storage
//     dtype, stages the wave's 128x64 tile in LDS and writes full 128-byte row segments;
//   * workgroup ids are remapped XCD-aware (8 XCDs, private L2s): each XCD owns a contiguous chunk
//     of a grouped (8 M-tiles wide) tile order, so concurrently resident tiles share A/B panels in L2.
#include <stdlib.h>

#include "common.h"

namespace tamd {

constexpr int kBM = 256, kBN = 256, kBK = 64;
constexpr int kGemmThreads = 512;
constexpr int kTileBytes = 256 * 64 * 2;             // one operand tile, 32 KiB
constexpr int kBufBytes = 2 * kTileBytes;            // A tile + B tile
constexpr int kStageRowBytes = 64 * 2 + 16;          // epilogue staging row (64 cols + 16 B pad)
constexpr int kStageWaveBytes = 128 * kStageRowBytes;
constexpr int kGemmSmem = (2 * kBufBytes > 8 * kStageWaveBytes) ? 2 * kBufBytes : 8 * kStageWaveBytes;

__device__ __attribute__((aligned(16))) static const unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

struct GemmArgs {
  const void* A;
  const void* B;
  void* C;
  const void* bias;
  const void* R;
  int64_t M, N, K, lda, ldb, ldc, ldr;
  int tiles_m, tiles_n;
  unsigned long long* trace;  // diagnostic: per-phase shader-clock stamps of workgroup 0 (tamd_gemm_trace)
};

// ---- operand tile loaders -------------------------------------------------------------------
// K-contiguous operand: global [rows, K] (row stride ld); LDS [256][64] with slot swizzle.
template <typename T>
__device__ __forceinline__ void issue_tile_rowmajor(const T* __restrict__ G, int64_t ld, int64_t row0, int64_t nrows,
                                                    int64_t k0, int64_t K, char* smem, unsigned tile_off, int wave,
                                                    int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 8 + (lane >> 3);
    const int p = lane & 7;
    const int s = p ^ ((r >> 1) & 7);
    const int64_t gr = row0 + r, gk = k0 + s * 8;
    const void* src = (gr < nrows && gk < K) ? (const void*)(G + gr * ld + gk) : (const void*)g_zero16;
    glds16(src, smem, tile_off + (unsigned)(wave * 4 + i) * 1024u);
  }
}
// K-major operand: global [K, cols] (row stride ld); LDS [64][256] with slot swizzle.
template <typename T>
__device__ __forceinline__ void issue_tile_kmajor(const T* __restrict__ G, int64_t ld, int64_t col0, int64_t ncols,
                                                  int64_t k0, int64_t K, char* smem, unsigned tile_off, int wave,
                                                  int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int kr = (wave * 4 + i) * 2 + (lane >> 5);
    const int p = lane & 31;
    const int s = p ^ ((kr & 3) << 2);
    const int64_t gk = k0 + kr, gc = col0 + s * 8;
    const void* src = (gk < K && gc < ncols) ? (const void*)(G + gk * ld + gc) : (const void*)g_zero16;
    glds16(src, smem, tile_off + (unsigned)(wave * 4 + i) * 1024u);
  }
}

// ---- fragment reads: 8 k-values (16 B) of one row/col of the tile for one lane ----------------
// row-major tile: element (row, k) ; lane needs k = ks*16 + hi*8 .. +7
__device__ __forceinline__ u32x4 frag_rowmajor(const char* smem, unsigned tile_off, int row, int ks, int hi) {
  const int slot = (ks * 2 + hi) ^ ((row >> 1) & 7);
  return lds_read16(smem, tile_off + (unsigned)row * 128u + (unsigned)slot * 16u);
}
// k-major tile: two transposing 8-byte reads; `col32` = first column of the 32-wide MFMA tile
__device__ __forceinline__ u32x4 frag_kmajor(const char* smem, unsigned tile_off, int col32, int ks, int lane) {
  const int hi = lane >> 5;
  const int kq = (lane & 15) >> 2;  // row within the 4-row block
  const int col = col32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int slot = (col >> 3) ^ (kq << 2);  // (k & 3) == kq : the other k terms are multiples of 4
  const unsigned inner = (unsigned)(col & 7) * 2u;
  const int kbase = ks * 16 + hi * 8 + kq;
  const u32x2 lo = lds_read8_tr16(smem, tile_off + (unsigned)kbase * 512u + (unsigned)slot * 16u + inner);
  const u32x2 hi2 = lds_read8_tr16(smem, tile_off + (unsigned)(kbase + 4) * 512u + (unsigned)slot * 16u + inner);
  return u32x4{lo[0], lo[1], hi2[0], hi2[1]};
}

template <int ACT>
__device__ __forceinline__ float gemm_act(float x) {
  if (ACT == TAMD_ACT_GELU_ERF) return x * 0.5f * (1.f + erff(x * 0.70710678118654752440f));
  if (ACT == TAMD_ACT_GELU_TANH) return 0.5f * x * (1.f + tanhf(0.79788456080286535588f * (x + 0.044715f * x * x * x)));
  if (ACT == TAMD_ACT_QUICK_GELU) return x / (1.f + __expf(-1.702f * x));
  if (ACT == TAMD_ACT_SILU) return x / (1.f + __expf(-x));
  return x;
}

template <typename T, bool A_KM, bool B_KN, int EPI, int ACT>
__global__ __launch_bounds__(kGemmThreads) void gemm_kernel(GemmArgs g) {
  TAMD_DYN_SMEM(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 2, wn = wave & 3;
  const int hi = lane >> 5, l31 = lane & 31;

  // ---- XCD-aware, grouped tile order
  const int nwg = g.tiles_m * g.tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, in_xcd = bid >> 3;
  const int q = nwg >> 3, rr = nwg & 7;
  const int logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + in_xcd;
  constexpr int GROUP_M = 8;
  const int group_size = GROUP_M * g.tiles_n;
  const int grp = logical / group_size;
  const int first_m = grp * GROUP_M;
  const int gm = (g.tiles_m - first_m < GROUP_M) ? (g.tiles_m - first_m) : GROUP_M;
  const int tile_m = first_m + (logical % group_size) % gm;
  const int tile_n = (logical % group_size) / gm;
  const int64_t m0 = (int64_t)tile_m * kBM, n0 = (int64_t)tile_n * kBN;

  const T* A = reinterpret_cast<const T*>(g.A);
  const T* B = reinterpret_cast<const T*>(g.B);

  f32x16 acc[2][4];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  const int nk = (int)((g.K + kBK - 1) / kBK);
  auto issue = [&](int t, int buf) {
    const int64_t k0 = (int64_t)t * kBK;
    const unsigned a_off = (unsigned)buf * kBufBytes, b_off = a_off + kTileBytes;
    if (A_KM)
      issue_tile_kmajor<T>(A, g.lda, m0, g.M, k0, g.K, smem, a_off, wave, lane);
    else
      issue_tile_rowmajor<T>(A, g.lda, m0, g.M, k0, g.K, smem, a_off, wave, lane);
    if (B_KN)
      issue_tile_kmajor<T>(B, g.ldb, n0, g.N, k0, g.K, smem, b_off, wave, lane);
    else
      issue_tile_rowmajor<T>(B, g.ldb, n0, g.N, k0, g.K, smem, b_off, wave, lane);
  };

  issue(0, 0);
  wait_vmcnt0();
  block_sync();
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    if (t + 1 < nk) issue(t + 1, cur ^ 1);
    const unsigned a_off = (unsigned)cur * kBufBytes, b_off = a_off + kTileBytes;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 xa[4], wb[2];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        if (B_KN)
          wb[ni] = frag_kmajor(smem, b_off, wn * 64 + ni * 32, ks, lane);
        else
          wb[ni] = frag_rowmajor(smem, b_off, wn * 64 + ni * 32 + l31, ks, hi);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        if (A_KM)
          xa[mi] = frag_kmajor(smem, a_off, wm * 128 + mi * 32, ks, lane);
        else
          xa[mi] = frag_rowmajor(smem, a_off, wm * 128 + mi * 32 + l31, ks, hi);
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = mfma32<T>(wb[ni], xa[mi], acc[ni][mi]);
    }
    wait_vmcnt0();
    block_sync();
  }

  // ---- epilogue: round -> stage the wave's 128(m) x 64(n) tile in LDS -> full-row global stores
  // acc[ni][mi][r] = D[n = ni*32 + (r&3) + 8*(r>>2) + 4*hi][m = mi*32 + l31]
  const unsigned st_off = (unsigned)wave * kStageWaveBytes;
  const T* bias = reinterpret_cast<const T*>(g.bias);
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int nl = ni * 32 + 8 * qd + 4 * hi;  // first of 4 consecutive local columns
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (EPI == TAMD_EPI_BIAS || EPI == TAMD_EPI_BIAS_ACT || (EPI == TAMD_EPI_RESIDUAL && bias != nullptr)) {
        const int64_t gn = n0 + wn * 64 + nl;
        if (gn < g.N) {  // N % 8 == 0 and nl % 4 == 0: the 4 columns are valid together
          const u32x2 bq = ld8(bias + gn);
          bv[0] = elem<T>::to_f32((typename elem<T>::raw)(bq[0] & 0xffffu));
          bv[1] = elem<T>::to_f32((typename elem<T>::raw)(bq[0] >> 16));
          bv[2] = elem<T>::to_f32((typename elem<T>::raw)(bq[1] & 0xffffu));
          bv[3] = elem<T>::to_f32((typename elem<T>::raw)(bq[1] >> 16));
        }
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = acc[ni][mi][qd * 4 + e] + bv[e];
          if (EPI == TAMD_EPI_BIAS_ACT) x = gemm_act<ACT>(round_through<T>(x));
          v[e] = x;
        }
        const u32x2 pk = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
        lds_write8(smem, st_off + (unsigned)(mi * 32 + l31) * kStageRowBytes + (unsigned)nl * 2u, pk);
      }
    }
  }
  // wave-private staging region: the wave's own LDS writes are ordered before its reads below
  wave_lockstep_point();
  T* C = reinterpret_cast<T*>(g.C);
  const T* R = reinterpret_cast<const T*>(g.R);
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int row = it * 8 + (lane >> 3), slot = lane & 7;
    const int64_t gm_ = m0 + wm * 128 + row, gn = n0 + wn * 64 + slot * 8;
    u32x4 v = lds_read16(smem, st_off + (unsigned)row * kStageRowBytes + (unsigned)slot * 16u);
    if (gm_ < g.M && gn < g.N) {
      if (EPI == TAMD_EPI_RESIDUAL || EPI == TAMD_EPI_ACCUM) {
        const T* rp = (EPI == TAMD_EPI_ACCUM) ? (C + gm_ * g.ldc + gn) : (R + gm_ * g.ldr + gn);
        float a[8], b[8];
        unpack16<T>(v, a);
        unpack16<T>(ld16(rp), b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += b[e];
        v = pack16<T>(a);
      }
      st16(C + gm_ * g.ldc + gn, v);
    }
  }
}

template <typename T, bool A_KM, bool B_KN>
static int gemm_launch_epi(const GemmArgs& g, int epilogue, int act, hipStream_t s) {
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n)), block(kGemmThreads);
#define TAMD_G(E_, A_)                                                                                          \
  hipLaunchKernelGGL((gemm_kernel<T, A_KM, B_KN, E_, A_>), grid, block, (size_t)kGemmSmem, s, g); \
  return launch_status();
  switch (epilogue) {
    case TAMD_EPI_NONE: TAMD_G(TAMD_EPI_NONE, TAMD_ACT_NONE)
    case TAMD_EPI_BIAS: TAMD_G(TAMD_EPI_BIAS, TAMD_ACT_NONE)
    case TAMD_EPI_RESIDUAL: TAMD_G(TAMD_EPI_RESIDUAL, TAMD_ACT_NONE)
    case TAMD_EPI_ACCUM: TAMD_G(TAMD_EPI_ACCUM, TAMD_ACT_NONE)
    case TAMD_EPI_BIAS_ACT:
      switch (act) {
        case TAMD_ACT_GELU_ERF: TAMD_G(TAMD_EPI_BIAS_ACT, TAMD_ACT_GELU_ERF)
        case TAMD_ACT_GELU_TANH: TAMD_G(TAMD_EPI_BIAS_ACT, TAMD_ACT_GELU_TANH)
        case TAMD_ACT_QUICK_GELU: TAMD_G(TAMD_EPI_BIAS_ACT, TAMD_ACT_QUICK_GELU)
        case TAMD_ACT_SILU: TAMD_G(TAMD_EPI_BIAS_ACT, TAMD_ACT_SILU)
        default: return TAMD_E_ARG;
      }
    default: return TAMD_E_ARG;
  }
#undef TAMD_G
}

template <typename T>
static int gemm_launch(const GemmArgs& g, int flags, int epilogue, int act, hipStream_t s) {
  const bool akm = flags & TAMD_GEMM_A_KM, bkn = flags & TAMD_GEMM_B_KN;
  if (!akm && !bkn) return gemm_launch_epi<T, false, false>(g, epilogue, act, s);
  if (!akm && bkn) return gemm_launch_epi<T, false, true>(g, epilogue, act, s);
  if (akm && bkn) return gemm_launch_epi<T, true, true>(g, epilogue, act, s);
  return gemm_launch_epi<T, true, false>(g, epilogue, act, s);
}

}  // namespace tamd

// =====================================================================================================
// v2: ping-pong schedule on a 4-stage LDS ring  (selected by default; TAMD_GEMM=v1 keeps the kernel above)
// =====================================================================================================
// Same tile, same fragment layouts, same epilogue as v1 -- what changes is WHO waits for WHAT, WHEN:
//   * the K loop advances in sub-tiles of BK=32; the 128 KiB of LDS hold a ring of 4 stages
//     (stage = A[256][32] 16 KiB + B[256][32] 16 KiB), so loads run three sub-tiles ahead of the math;
//   * the 8 waves form two groups (waves 0-3 own output rows 0-127, waves 4-7 rows 128-255); waves w and
//     w+4 share a SIMD.  The groups run one phase apart:
//         phase 2j   : group 0 LOADs  fragments of sub-tile j   | group 1 COMPUTEs sub-tile j-1
//         phase 2j+1 : group 0 COMPUTEs sub-tile j (16 MFMA)    | group 1 LOADs  fragments of sub-tile j
//     so on every SIMD one wave feeds the matrix pipe from registers while its partner does the LDS reads
//     and issues the next direct-to-LDS loads (cdna_hip_programming.md T3/T5: role split + s_setprio);
//   * a LOAD phase = 12 ds_read (ds_read_b128 or tr16 pairs) + 4 global_load_lds (this wave's share of
//     sub-tile j+3 into the stage sub-tile j-1 just vacated) + ONE counted wait: vmcnt(8) retires the share
//     issued two LOAD phases ago (sub-tile j+1) and leaves the two newest batches in flight across the
//     barrier; raw s_barrier, never __syncthreads() (which would drain vmcnt to 0);
//   * hazards: sub-tile j is read by group 0 in phase 2j and group 1 in phase 2j+1; its stage is rewritten by
//     loads issued in phases 2j+2 / 2j+3 (after the barrier that ends phase 2j+1, by which time every reader
//     has passed lgkmcnt(0)); the data is first read in phase 2j+8, after both issuers' vmcnt waits
//     (end of phases 2j+6 / 2j+7) and the barrier that ends phase 2j+7.
// LDS images per stage: row-major operand [256][32] (64-byte rows): 16-B slot' = slot ^ ((row>>2)&3);
//                       k-major operand   [32][256]              : 16-B slot' = slot ^ ((k&3)<<2).
namespace tamd {

constexpr int kSubK = 32;
constexpr int kStageOperand = 256 * kSubK * 2;  // 16 KiB
constexpr int kStageBytes = 2 * kStageOperand;  // A + B
constexpr int kRing = 4;

// source address of this lane's 16 bytes of wave-instruction `inst` (0..15) of one operand stage
template <typename T, bool KMAJOR>
__device__ __forceinline__ const void* pp_src(const T* __restrict__ G, int64_t ld, int64_t rc0, int64_t nrc,
                                              int64_t k0, int64_t K, int inst, int lane) {
  if (KMAJOR) {
    const int kr = inst * 2 + (lane >> 5);
    const int p = lane & 31;
    const int s = p ^ ((kr & 3) << 2);
    const int64_t gk = k0 + kr, gc = rc0 + s * 8;
    return (gk < K && gc < nrc) ? (const void*)(G + gk * ld + gc) : (const void*)g_zero16;
  }
  const int r = inst * 16 + (lane >> 2);
  const int p = lane & 3;
  const int s = p ^ ((r >> 2) & 3);
  const int64_t gr = rc0 + r, gk = k0 + s * 8;
  return (gr < nrc && gk < K) ? (const void*)(G + gr * ld + gk) : (const void*)g_zero16;
}

template <typename T, bool KMAJOR, int AUX = 0>
__device__ __forceinline__ void pp_issue(const T* __restrict__ G, int64_t ld, int64_t rc0, int64_t nrc, int64_t k0,
                                         int64_t K, char* smem, unsigned off, int wave, int lane, int i0 = 0,
                                         int i1 = 2) {
#pragma unroll
  for (int i = i0; i < i1; ++i) {
    const int inst = wave * 2 + i;  // 16 wave-instructions per operand stage, 2 per wave
    glds16<AUX>(pp_src<T, KMAJOR>(G, ld, rc0, nrc, k0, K, inst, lane), smem, off + (unsigned)inst * 1024u);
  }
}

template <bool KMAJOR>
__device__ __forceinline__ u32x4 pp_frag(const char* smem, unsigned off, int rc32, int ks, int lane) {
  const int hi = lane >> 5;
  if (KMAJOR) {
    const int kq = (lane & 15) >> 2;
    const int col = rc32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int slot = (col >> 3) ^ (kq << 2);
    const unsigned inner = (unsigned)(col & 7) * 2u;
    const int kbase = ks * 16 + hi * 8 + kq;
    const u32x2 lo = lds_read8_tr16(smem, off + (unsigned)kbase * 512u + (unsigned)slot * 16u + inner);
    const u32x2 h2 = lds_read8_tr16(smem, off + (unsigned)(kbase + 4) * 512u + (unsigned)slot * 16u + inner);
    return u32x4{lo[0], lo[1], h2[0], h2[1]};
  } else {
    const int row = rc32 + (lane & 31);
    const int slot = (ks * 2 + hi) ^ ((row >> 2) & 3);
    return lds_read16(smem, off + (unsigned)row * 64u + (unsigned)slot * 16u);
  }
}

// byte offset of a lane's fragment inside an operand stage (first of the two reads when k-major), and the read
template <bool KMAJOR>
__device__ __forceinline__ unsigned pp_frag_off(int rc32, int ks, int lane) {
  const int hi = lane >> 5;
  if (KMAJOR) {
    const int kq = (lane & 15) >> 2;
    const int col = rc32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int slot = (col >> 3) ^ (kq << 2);
    return (unsigned)(ks * 16 + hi * 8 + kq) * 512u + (unsigned)slot * 16u + (unsigned)(col & 7) * 2u;
  }
  const int row = rc32 + (lane & 31);
  return (unsigned)row * 64u + (unsigned)(((ks * 2 + hi) ^ ((row >> 2) & 3)) * 16);
}
template <bool KMAJOR>
__device__ __forceinline__ u32x4 pp_frag_at(const char* smem, unsigned off) {
  if (KMAJOR) {
    const u32x2 lo = lds_read8_tr16(smem, off);
    const u32x2 h2 = lds_read8_tr16(smem, off + 4u * 512u);
    return u32x4{lo[0], lo[1], h2[0], h2[1]};
  }
  return lds_read16(smem, off);
}

template <typename T, bool A_KM, bool B_KN, int EPI, int ACT, bool TRACE = false, int VAR = 0>
__global__ __launch_bounds__(kGemmThreads) void gemm_pp_kernel(GemmArgs g) {
  TAMD_DYN_SMEM(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 2, wn = wave & 3;  // wm = ping-pong group
  const int hi = lane >> 5, l31 = lane & 31;

  const int nwg = g.tiles_m * g.tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, in_xcd = bid >> 3;
  const int q = nwg >> 3, rr = nwg & 7;
  const int logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + in_xcd;
  constexpr int GROUP_M = 8;
  const int group_size = GROUP_M * g.tiles_n;
  const int grp = logical / group_size;
  const int first_m = grp * GROUP_M;
  const int gm = (g.tiles_m - first_m < GROUP_M) ? (g.tiles_m - first_m) : GROUP_M;
  const int tile_m = first_m + (logical % group_size) % gm;
  const int tile_n = (logical % group_size) / gm;
  const int64_t m0 = (int64_t)tile_m * kBM, n0 = (int64_t)tile_n * kBN;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* B = reinterpret_cast<const T*>(g.B);

  f32x16 acc[2][4];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  const int nsub = (int)((g.K + kSubK - 1) / kSubK);
  constexpr int AUX = (VAR & 32) ? 2 : ((VAR & 64) ? 1 : ((VAR & 128) ? 16 : 0));
  // K-sweep rotation (sum over k is order independent): spreads the instantaneous k-window -- and with it the
  // L2 / fabric channels being hit -- across XCDs (VAR&2048) and/or across the M-tiles of a patch (VAR&4096)
  int joff = 0;
  if (VAR & 2048) joff += xcd * (nsub >> 3);
  if (VAR & 4096) joff += (tile_m & 7) * (nsub >> 3);
  auto ksub = [&](int j) { return (VAR & (2048 | 4096)) ? (int64_t)((j + joff) % nsub) * kSubK : (int64_t)j * kSubK; };
  auto issue = [&](int j) {  // this wave's share (2 A + 2 B pieces) of sub-tile j into stage j % 4
    const unsigned st = (unsigned)(j & (kRing - 1)) * kStageBytes;
    const int64_t k0 = (j < nsub) ? ksub(j) : (int64_t)j * kSubK;
    pp_issue<T, A_KM, AUX>(A, g.lda, m0, g.M, k0, g.K, smem, st, wave, lane);
    pp_issue<T, B_KN, AUX>(B, g.ldb, n0, g.N, k0, g.K, smem, st + kStageOperand, wave, lane);
  };
  auto issue_half = [&](int j, int half) {  // one A piece + one B piece
    const unsigned st = (unsigned)(j & (kRing - 1)) * kStageBytes;
    const int64_t k0 = (int64_t)j * kSubK;
    pp_issue<T, A_KM, AUX>(A, g.lda, m0, g.M, k0, g.K, smem, st, wave, lane, half, half + 1);
    pp_issue<T, B_KN, AUX>(B, g.ldb, n0, g.N, k0, g.K, smem, st + kStageOperand, wave, lane, half, half + 1);
  };

  // VAR&256: register-staged loads (global_load -> VGPR -> ds_write one LOAD phase later) instead of LDS-DMA
  u32x4 stg[4];
  auto rload = [&](int j) {
    const int64_t k0 = (int64_t)j * kSubK;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      stg[i] = ld16(pp_src<T, A_KM>(A, g.lda, m0, g.M, k0, g.K, wave * 2 + i, lane));
      stg[2 + i] = ld16(pp_src<T, B_KN>(B, g.ldb, n0, g.N, k0, g.K, wave * 2 + i, lane));
    }
  };
  auto rstore = [&](int j) {
    const unsigned st = (unsigned)(j & (kRing - 1)) * kStageBytes;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      lds_write16(smem, st + (unsigned)(wave * 2 + i) * 1024u + (unsigned)lane * 16u, stg[i]);
      lds_write16(smem, st + kStageOperand + (unsigned)(wave * 2 + i) * 1024u + (unsigned)lane * 16u, stg[2 + i]);
    }
  };
  if (VAR & 256) {
    rload(0);
    rstore(0);
    rload(1);
    rstore(1);
    rload(2);
    wait_lgkmcnt0();
    raw_barrier();
  } else {
    // prologue: sub-tiles 0..2 (sub-tiles past the end read the zero page: counts stay uniform)
    issue(0);
    issue(1);
    issue(2);
    wait_vmcnt<0>();
    raw_barrier();
  }
  if (wm == 1) raw_barrier();  // stagger: group 1 runs one phase behind group 0

  u32x4 xa[2][4], wb[2][2];
  // TRACE: 8 stamps per sub-tile for the first 32 sub-tiles of every wave of workgroup 0
  const bool tr = TRACE && g.trace != nullptr && blockIdx.x == 0 && lane == 0;
#define TAMD_STAMP(i_)                                                          \
  if (TRACE && tr && j < 32) g.trace[((size_t)wave * 32 + j) * 8 + (i_)] = device_clock();
  for (int j = 0; j < nsub; ++j) {
    // ---------------- LOAD phase
    TAMD_STAMP(0)
    const unsigned st = (unsigned)(j & (kRing - 1)) * kStageBytes;
    if (VAR & 4) setprio_hi();
    if (VAR & 256) {
      rstore(j + 2);  // loaded one LOAD phase ago
      rload(j + 3);
    }
    if ((VAR & 1) && !(VAR & 8)) issue(j + 3);
    if (!(VAR & 16) || j == 0) {  // VAR&16: ablation -- fragments are read once, the LDS read traffic disappears
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          wb[ks][ni] = pp_frag<B_KN>(smem, st + kStageOperand, wn * 64 + ni * 32, ks, lane);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) xa[ks][mi] = pp_frag<A_KM>(smem, st, wm * 128 + mi * 32, ks, lane);
      }
    }
    TAMD_STAMP(1)
    if (VAR & 1024) issue_half(j + 3, 0);
    if (!(VAR & 1) && !(VAR & 8) && !(VAR & 256) && !(VAR & 1024)) issue(j + 3);
    TAMD_STAMP(2)
    if (VAR & 1024) wait_vmcnt<6>();
    if (VAR & 16384) wait_vmcnt<20>();  // ablation: deeper in-flight window (results are garbage)
    if (!(VAR & 256) && !(VAR & 1024) && !(VAR & 16384)) wait_vmcnt<8>();   // retires this wave's share of sub-tile j+1; j+2, j+3 stay in flight
    TAMD_STAMP(3)
    wait_lgkmcnt0();   // fragments are in registers: the stage may be recycled after the next barrier
    TAMD_STAMP(4)
    if (VAR & 4) setprio_lo();
    sched_fence();
    if (!(VAR & 8192)) raw_barrier();
    TAMD_STAMP(5)
    // ---------------- COMPUTE phase (registers only)
    if (!(VAR & 6)) setprio_hi();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if ((VAR & 1024) && ks == 1) {  // second half of this wave's loads rides inside the MFMA stream
        sched_fence();
        issue_half(j + 3, 1);
        sched_fence();
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          if (VAR & 512) {  // ablation: no matrix work (keep the fragments observable)
            acc[ni][mi][0] += u32_as_f32(wb[ks][ni][0] ^ xa[ks][mi][0]);
          } else {
            acc[ni][mi] = mfma32<T>(wb[ks][ni], xa[ks][mi], acc[ni][mi]);
          }
        }
    }
    if (!(VAR & 6)) setprio_lo();
    sched_fence();
    TAMD_STAMP(6)
    if (!(VAR & 8192)) raw_barrier();
    TAMD_STAMP(7)
  }
#undef TAMD_STAMP
  if (wm == 0) raw_barrier();
  wait_vmcnt<0>();  // trailing (zero-page) loads must land before the epilogue reuses the LDS
  raw_barrier();

  // ---- epilogue (identical to v1)
  const unsigned st_off = (unsigned)wave * kStageWaveBytes;
  const T* bias = reinterpret_cast<const T*>(g.bias);
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int nl = ni * 32 + 8 * qd + 4 * hi;
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (EPI == TAMD_EPI_BIAS || EPI == TAMD_EPI_BIAS_ACT || (EPI == TAMD_EPI_RESIDUAL && bias != nullptr)) {
        const int64_t gn = n0 + wn * 64 + nl;
        if (gn < g.N) {
          const u32x2 bq = ld8(bias + gn);
          bv[0] = elem<T>::to_f32((typename elem<T>::raw)(bq[0] & 0xffffu));
          bv[1] = elem<T>::to_f32((typename elem<T>::raw)(bq[0] >> 16));
          bv[2] = elem<T>::to_f32((typename elem<T>::raw)(bq[1] & 0xffffu));
          bv[3] = elem<T>::to_f32((typename elem<T>::raw)(bq[1] >> 16));
        }
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = acc[ni][mi][qd * 4 + e] + bv[e];
          if (EPI == TAMD_EPI_BIAS_ACT) x = gemm_act<ACT>(round_through<T>(x));
          v[e] = x;
        }
        const u32x2 pk = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
        lds_write8(smem, st_off + (unsigned)(mi * 32 + l31) * kStageRowBytes + (unsigned)nl * 2u, pk);
      }
    }
  }
  wave_lockstep_point();
  T* C = reinterpret_cast<T*>(g.C);
  const T* R = reinterpret_cast<const T*>(g.R);
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int row = it * 8 + (lane >> 3), slot = lane & 7;
    const int64_t gm_ = m0 + wm * 128 + row, gn = n0 + wn * 64 + slot * 8;
    u32x4 v = lds_read16(smem, st_off + (unsigned)row * kStageRowBytes + (unsigned)slot * 16u);
    if (gm_ < g.M && gn < g.N) {
      if (EPI == TAMD_EPI_RESIDUAL || EPI == TAMD_EPI_ACCUM) {
        const T* rp = (EPI == TAMD_EPI_ACCUM) ? (C + gm_ * g.ldc + gn) : (R + gm_ * g.ldr + gn);
        float a[8], b[8];
        unpack16<T>(v, a);
        unpack16<T>(ld16(rp), b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += b[e];
        v = pack16<T>(a);
      }
      st16(C + gm_ * g.ldc + gn, v);
    }
  }
}

template <typename T, bool A_KM, bool B_KN>
static int gemm_pp_launch_epi(const GemmArgs& g, int epilogue, int act, hipStream_t s) {
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n)), block(kGemmThreads);
#define TAMD_G(E_, A_)                                                                                          \
  hipLaunchKernelGGL((gemm_pp_kernel<T, A_KM, B_KN, E_, A_, false, 2>), grid, block, (size_t)kGemmSmem, s, g); \
  return launch_status();
  static const int var = [] {
    const char* e = getenv("TAMD_GEMM_VAR");
    return e ? atoi(e) : 2;  // 2 = no s_setprio around the MFMA phase (measured +4 % on MI355X, profiles/r01_gemm_variants.md)
  }();
#define TAMD_GV(V_)                                                                                              \
  hipLaunchKernelGGL((gemm_pp_kernel<T, A_KM, B_KN, TAMD_EPI_NONE, TAMD_ACT_NONE, false, V_>), grid, block,      \
                     (size_t)kGemmSmem, s, g);                                                                   \
  return launch_status();
  if (epilogue == TAMD_EPI_NONE && var != 0 && var != 2) {
    switch (var) {
      case 1: TAMD_GV(1)
      case 3: TAMD_GV(3)
      case 4: TAMD_GV(4)
      case 5: TAMD_GV(5)
      case 10: TAMD_GV(10)  // ablation: no global->LDS traffic in the loop
      case 18: TAMD_GV(18)  // ablation: no LDS fragment reads in the loop
      case 26: TAMD_GV(26)  // ablation: MFMA + barriers only
      case 34: TAMD_GV(34)   // nt loads
      case 66: TAMD_GV(66)   // sc0 loads
      case 130: TAMD_GV(130) // sc1 loads
      case 258: TAMD_GV(258) // register-staged loads
      case 1026: TAMD_GV(1026) // loads split between LOAD and COMPUTE phases
      case 2050: TAMD_GV(2050) // K sweep rotated per XCD
      case 4098: TAMD_GV(4098) // K sweep rotated per M-tile
      case 6146: TAMD_GV(6146) // both
      case 514: TAMD_GV(514) // ablation: loads + LDS reads, no MFMA
      case 530: TAMD_GV(530) // ablation: loads only
      case 8722: TAMD_GV(8722) // ablation: loads only, no barriers in the loop
      case 25106: TAMD_GV(25106) // ablation: loads only, no barriers, 24 pieces in flight per wave
      case 16914: TAMD_GV(16914) // ablation: loads only, barriers, 24 pieces in flight
      default: break;
    }
  }
#undef TAMD_GV
  switch (epilogue) {
    case TAMD_EPI_NONE: TAMD_G(TAMD_EPI_NONE, TAMD_ACT_NONE)
    case TAMD_EPI_BIAS: TAMD_G(TAMD_EPI_BIAS, TAMD_ACT_NONE)
    case TAMD_EPI_RESIDUAL: TAMD_G(TAMD_EPI_RESIDUAL, TAMD_ACT_NONE)
    case TAMD_EPI_ACCUM: TAMD_G(TAMD_EPI_ACCUM, TAMD_ACT_NONE)
    case TAMD_EPI_BIAS_ACT:
      switch (act) {
        case TAMD_ACT_GELU_ERF: TAMD_G(TAMD_EPI_BIAS_ACT, TAMD_ACT_GELU_ERF)
        case TAMD_ACT_GELU_TANH: TAMD_G(TAMD_EPI_BIAS_ACT, TAMD_ACT_GELU_TANH)
        case TAMD_ACT_QUICK_GELU: TAMD_G(TAMD_EPI_BIAS_ACT, TAMD_ACT_QUICK_GELU)
        case TAMD_ACT_SILU: TAMD_G(TAMD_EPI_BIAS_ACT, TAMD_ACT_SILU)
        default: return TAMD_E_ARG;
      }
    default: return TAMD_E_ARG;
  }
#undef TAMD_G
}

template <typename T>
static int gemm_pp_launch(const GemmArgs& g, int flags, int epilogue, int act, hipStream_t s) {
  const bool akm = flags & TAMD_GEMM_A_KM, bkn = flags & TAMD_GEMM_B_KN;
  if (!akm && !bkn) return gemm_pp_launch_epi<T, false, false>(g, epilogue, act, s);
  if (!akm && bkn) return gemm_pp_launch_epi<T, false, true>(g, epilogue, act, s);
  if (akm && bkn) return gemm_pp_launch_epi<T, true, true>(g, epilogue, act, s);
  return gemm_pp_launch_epi<T, true, false>(g, epilogue, act, s);
}

}  // namespace tamd

// =====================================================================================================
// v3: one wave per SIMD -- 4 waves x (128 x 128) with all 512 registers  (TAMD_GEMM=v3)
// =====================================================================================================
// What the v2 measurements said (profiles/r01_gemm_variants.md): the loop is bound by the global->LDS feed and
// by the LDS port, not by MFMA scheduling.  This variant halves the number of waves and doubles the per-wave
// tile: 256 accumulator registers per wave (the unified 512-entry file of gfx950, one wave per SIMD), so a
// BK=32 step reads 64 KiB of fragments per CU instead of 96 KiB and every wave issues 8 instead of 4 LDS-DMA
// pieces per step with nobody else on its SIMD: reads for the next k-step and the next sub-tile's loads are
// interleaved BETWEEN the MFMAs of the current k-step (register double buffer, sched_group_barrier pattern),
// one barrier per BK=32 step.  Same LDS ring, images, swizzles and epilogue as v2.
namespace tamd {

constexpr int kW4Threads = 256;

template <typename T, bool A_KM, bool B_KN, int EPI, int ACT>
__global__ __launch_bounds__(kW4Threads, 1) void gemm_w4_kernel(GemmArgs g) {
  TAMD_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id_uniform();  // SGPR: LDS-DMA destinations (M0) become scalar arithmetic
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;

  const int nwg = g.tiles_m * g.tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, in_xcd = bid >> 3;
  const int q = nwg >> 3, rr = nwg & 7;
  const int logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + in_xcd;
  constexpr int GROUP_M = 8;
  const int group_size = GROUP_M * g.tiles_n;
  const int grp = logical / group_size;
  const int first_m = grp * GROUP_M;
  const int gm = (g.tiles_m - first_m < GROUP_M) ? (g.tiles_m - first_m) : GROUP_M;
  const int tile_m = first_m + (logical % group_size) % gm;
  const int tile_n = (logical % group_size) / gm;
  const int64_t m0 = (int64_t)tile_m * kBM, n0 = (int64_t)tile_n * kBN;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* B = reinterpret_cast<const T*>(g.B);

  f32x16 acc[4][4];  // [ni][mi]
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  const int nsub = (int)(g.K / kSubK);  // this kernel requires K % 32 == 0 (host dispatch)
  // ---- per-lane source pointers of this wave's 4 A pieces and 4 B pieces; they advance by a constant per
  // sub-tile.  Rows/columns outside the matrix point at the zero page with a zero increment.
  const char* srcp[8];
  int inc[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int inst = wave * 4 + i;
    const void* pa = pp_src<T, A_KM>(A, g.lda, m0, g.M, 0, g.K, inst, lane);
    const void* pb = pp_src<T, B_KN>(B, g.ldb, n0, g.N, 0, g.K, inst, lane);
    srcp[i] = (const char*)pa;
    srcp[4 + i] = (const char*)pb;
    inc[i] = (pa == (const void*)g_zero16) ? 0 : (int)((A_KM ? (int64_t)kSubK * g.lda : (int64_t)kSubK) * 2);
    inc[4 + i] = (pb == (const void*)g_zero16) ? 0 : (int)((B_KN ? (int64_t)kSubK * g.ldb : (int64_t)kSubK) * 2);
  }
  const unsigned piece0 = (unsigned)wave * 4096u;  // this wave's first piece inside an operand stage
  auto issue_part = [&](int stage, int part) {     // part 0: the 4 A pieces, part 1: the 4 B pieces
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = part * 4 + i;
      glds16(srcp[p], smem, (unsigned)stage * kStageBytes + (unsigned)part * kStageOperand + piece0 + (unsigned)i * 1024u);
      srcp[p] += inc[p];
    }
  };
  // ---- loop-invariant fragment offsets inside a stage (swizzles depend on the lane only)
  unsigned offx[2][4], offw[2][4];  // [ks][mi / ni]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      offx[ks][t] = pp_frag_off<A_KM>(wm * 128 + t * 32, ks, lane);
      offw[ks][t] = kStageOperand + pp_frag_off<B_KN>(wn * 128 + t * 32, ks, lane);
    }
  u32x4 fx[2][4], fw[2][4];  // [buffer][mi / ni]
  auto read_frags = [&](int stage, int ks, int buf) {
    const unsigned st = (unsigned)stage * kStageBytes;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) fw[buf][ni] = pp_frag_at<B_KN>(smem, st + offw[ks][ni]);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) fx[buf][mi] = pp_frag_at<A_KM>(smem, st + offx[ks][mi]);
  };
  auto mma = [&](int buf) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = mfma32<T>(fw[buf][ni], fx[buf][mi], acc[ni][mi]);
  };
  // interleave for one k-step: 16 MFMA with the 8 fragment reads (16 LDS instructions when k-major) and the
  // 4 LDS-DMA pieces tucked into the MFMA shadows
  auto pattern = [&]() {
    constexpr int DSN = ((A_KM ? 2 : 1) * 4 + (B_KN ? 2 : 1) * 4) / 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);    // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, DSN, 0);  // DS read
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);    // MFMA
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // VMEM read (LDS-DMA)
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);    // VALU (pointer increments)
    }
  };

  // prologue: sub-tiles 0..2
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (j == nsub) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        srcp[p] = (const char*)g_zero16;
        inc[p] = 0;
      }
    }
    issue_part(j, 0);
    issue_part(j, 1);
  }
  wait_vmcnt<0>();
  raw_barrier();
  read_frags(0, 0, 0);
  // main loop, unrolled over the 4 ring stages so every LDS address is base register + immediate
  for (int j0 = 0; j0 < nsub; j0 += kRing) {
#pragma unroll
    for (int u = 0; u < kRing; ++u) {
      const int j = j0 + u;
      if (j < nsub) {
        if (j + 3 == nsub) {  // past the last sub-tile: keep the load counts uniform but read the zero page
#pragma unroll
          for (int p = 0; p < 8; ++p) {
            srcp[p] = (const char*)g_zero16;
            inc[p] = 0;
          }
        }
        // k-step 0 of sub-tile j (buffer 0) | fetch k-step 1 fragments, first half of sub-tile j+3's loads
        sched_fence();
        read_frags(u, 1, 1);
        issue_part((u + 3) & 3, 0);
        mma(0);
        pattern();
        sched_fence();
        // hand-off: sub-tile j+1 has landed for everybody; everybody's reads of sub-tile j are in registers
        wait_vmcnt<12>();  // own pieces of sub-tile j+1; sub-tile j+2 (8) and half of j+3 (4) stay in flight
        wait_lgkmcnt0();
        raw_barrier();
        sched_fence();
        // k-step 1 (buffer 1) | fetch k-step 0 of sub-tile j+1, second half of sub-tile j+3's loads
        read_frags((u + 1) & 3, 0, 0);
        issue_part((u + 3) & 3, 1);
        mma(1);
        pattern();
        sched_fence();
      }
    }
  }
  wait_vmcnt<0>();
  wait_lgkmcnt0();
  raw_barrier();

  // ---- epilogue: two passes of 64 rows through the wave's staging region (128 cols -> 272-byte rows)
  constexpr int kRowB = 128 * 2 + 16;
  const unsigned st_off = (unsigned)wave * (64u * kRowB);
  const T* bias = reinterpret_cast<const T*>(g.bias);
  T* C = reinterpret_cast<T*>(g.C);
  const T* R = reinterpret_cast<const T*>(g.R);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int nl = ni * 32 + 8 * qd + 4 * hi;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (EPI == TAMD_EPI_BIAS || EPI == TAMD_EPI_BIAS_ACT || (EPI == TAMD_EPI_RESIDUAL && bias != nullptr)) {
          const int64_t gn = n0 + wn * 128 + nl;
          if (gn < g.N) {
            const u32x2 bq = ld8(bias + gn);
            bv[0] = elem<T>::to_f32((typename elem<T>::raw)(bq[0] & 0xffffu));
            bv[1] = elem<T>::to_f32((typename elem<T>::raw)(bq[0] >> 16));
            bv[2] = elem<T>::to_f32((typename elem<T>::raw)(bq[1] & 0xffffu));
            bv[3] = elem<T>::to_f32((typename elem<T>::raw)(bq[1] >> 16));
          }
        }
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
          const int mi = half * 2 + m2;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = acc[ni][mi][qd * 4 + e] + bv[e];
            if (EPI == TAMD_EPI_BIAS_ACT) x = gemm_act<ACT>(round_through<T>(x));
            v[e] = x;
          }
          const u32x2 pk = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
          lds_write8(smem, st_off + (unsigned)(m2 * 32 + l31) * kRowB + (unsigned)nl * 2u, pk);
        }
      }
    }
    wave_lockstep_point();
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int row = it * 4 + (lane >> 4), slot = lane & 15;
      const int64_t gm_ = m0 + wm * 128 + half * 64 + row, gn = n0 + wn * 128 + slot * 8;
      u32x4 v = lds_read16(smem, st_off + (unsigned)row * kRowB + (unsigned)slot * 16u);
      if (gm_ < g.M && gn < g.N) {
        if (EPI == TAMD_EPI_RESIDUAL || EPI == TAMD_EPI_ACCUM) {
          const T* rp = (EPI == TAMD_EPI_ACCUM) ? (C + gm_ * g.ldc + gn) : (R + gm_ * g.ldr + gn);
          float a[8], b[8];
          unpack16<T>(v, a);
          unpack16<T>(ld16(rp), b);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += b[e];
          v = pack16<T>(a);
        }
        st16(C + gm_ * g.ldc + gn, v);
      }
    }
    wave_lockstep_point();
  }
}

template <typename T, bool A_KM, bool B_KN>
static int gemm_w4_launch_epi(const GemmArgs& g, int epilogue, int act, hipStream_t s) {
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n)), block(kW4Threads);
#define TAMD_G(E_, A_)                                                                                \
  hipLaunchKernelGGL((gemm_w4_kernel<T, A_KM, B_KN, E_, A_>), grid, block, (size_t)kGemmSmem, s, g); \
  return launch_status();
  switch (epilogue) {
    case TAMD_EPI_NONE: TAMD_G(TAMD_EPI_NONE, TAMD_ACT_NONE)
    case TAMD_EPI_BIAS: TAMD_G(TAMD_EPI_BIAS, TAMD_ACT_NONE)
    case TAMD_EPI_RESIDUAL: TAMD_G(TAMD_EPI_RESIDUAL, TAMD_ACT_NONE)
    case TAMD_EPI_ACCUM: TAMD_G(TAMD_EPI_ACCUM, TAMD_ACT_NONE)
    case TAMD_EPI_BIAS_ACT:
      switch (act) {
        case TAMD_ACT_GELU_ERF: TAMD_G(TAMD_EPI_BIAS_ACT, TAMD_ACT_GELU_ERF)
        case TAMD_ACT_GELU_TANH: TAMD_G(TAMD_EPI_BIAS_ACT, TAMD_ACT_GELU_TANH)
        case TAMD_ACT_QUICK_GELU: TAMD_G(TAMD_EPI_BIAS_ACT, TAMD_ACT_QUICK_GELU)
        case TAMD_ACT_SILU: TAMD_G(TAMD_EPI_BIAS_ACT, TAMD_ACT_SILU)
        default: return TAMD_E_ARG;
      }
    default: return TAMD_E_ARG;
  }
#undef TAMD_G
}

template <typename T>
static int gemm_w4_launch(const GemmArgs& g, int flags, int epilogue, int act, hipStream_t s) {
  const bool akm = flags & TAMD_GEMM_A_KM, bkn = flags & TAMD_GEMM_B_KN;
  if (!akm && !bkn) return gemm_w4_launch_epi<T, false, false>(g, epilogue, act, s);
  if (!akm && bkn) return gemm_w4_launch_epi<T, false, true>(g, epilogue, act, s);
  if (akm && bkn) return gemm_w4_launch_epi<T, true, true>(g, epilogue, act, s);
  return gemm_w4_launch_epi<T, true, false>(g, epilogue, act, s);
}

}  // namespace tamd

using namespace tamd;

// Diagnostic: C = A.B^T (row-major operands, no epilogue) with the ping-pong kernel while workgroup 0 writes
// 8 shader-clock stamps per sub-tile and wave into `trace` (8 waves x 32 sub-tiles x 8 u64).  See tools/.
extern "C" int tamd_gemm_trace(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, void* trace,
                               tamd_stream_t stream) {
  if (!A || !B || !C || !trace) return TAMD_E_NULL;
  if ((K % 8) || (N % 8)) return TAMD_E_SHAPE;
  GemmArgs g;
  g.A = A;
  g.B = B;
  g.C = C;
  g.bias = nullptr;
  g.R = nullptr;
  g.M = M;
  g.N = N;
  g.K = K;
  g.lda = K;
  g.ldb = K;
  g.ldc = N;
  g.ldr = 0;
  g.tiles_m = (int)ceil_div(M, kBM);
  g.tiles_n = (int)ceil_div(N, kBN);
  g.trace = reinterpret_cast<unsigned long long*>(trace);
  hipLaunchKernelGGL((gemm_pp_kernel<bf16_t, false, false, TAMD_EPI_NONE, TAMD_ACT_NONE, true>),
                     dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(kGemmThreads), (size_t)kGemmSmem,
                     TAMD_STREAM(stream), g);
  return launch_status();
}

extern "C" int tamd_gemm(const void* A, const void* B, void* C, const void* bias, const void* R, int64_t M, int64_t N,
                         int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int flags, int epilogue,
                         int act, int dtype, tamd_stream_t stream) {
  if (!A || !B || !C) return TAMD_E_NULL;
  if (M <= 0 || N <= 0 || K <= 0) return TAMD_E_SHAPE;
  if ((K % 8) || (N % 8) || (lda % 8) || (ldb % 8) || (ldc % 8)) return TAMD_E_SHAPE;
  if ((flags & TAMD_GEMM_A_KM) && (M % 8)) return TAMD_E_SHAPE;
  if (!aligned16(A) || !aligned16(B) || !aligned16(C)) return TAMD_E_ALIGN;
  if ((epilogue == TAMD_EPI_BIAS || epilogue == TAMD_EPI_BIAS_ACT) && !bias) return TAMD_E_NULL;
  if (bias && (reinterpret_cast<uintptr_t>(bias) & 7u)) return TAMD_E_ALIGN;
  if (epilogue == TAMD_EPI_RESIDUAL && (!R || (ldr % 8) || !aligned16(R))) return R ? TAMD_E_ALIGN : TAMD_E_NULL;
  GemmArgs g;
  g.A = A;
  g.B = B;
  g.C = C;
  g.bias = bias;
  g.R = R;
  g.M = M;
  g.N = N;
  g.K = K;
  g.lda = lda;
  g.ldb = ldb;
  g.ldc = ldc;
  g.ldr = ldr;
  g.tiles_m = (int)ceil_div(M, kBM);
  g.tiles_n = (int)ceil_div(N, kBN);
  g.trace = nullptr;
  // kernel variant: v2 (ping-pong ring, default) or v1 (double-buffered K tiles); read once
  static const int variant = [] {
    const char* e = getenv("TAMD_GEMM");
    return (e && e[0] == 'v' && e[1] >= '1' && e[1] <= '3') ? e[1] - '0' : 0;  // 0 = auto
  }();
  if (variant == 1) {
    TAMD_DISPATCH_HALF(dtype, return (gemm_launch<T>(g, flags, epilogue, act, TAMD_STREAM(stream))));
  } else if ((variant == 3 || (variant == 0 && flags == 0)) && K % kSubK == 0) {
    // auto: row-major operands run best on the one-wave-per-SIMD kernel, k-major ones on the ping-pong kernel
    TAMD_DISPATCH_HALF(dtype, return (gemm_w4_launch<T>(g, flags, epilogue, act, TAMD_STREAM(stream))));
  } else {
    TAMD_DISPATCH_HALF(dtype, return (gemm_pp_launch<T>(g, flags, epilogue, act, TAMD_STREAM(stream))));
  }
  return TAMD_E_DTYPE;
}


