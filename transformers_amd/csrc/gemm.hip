// gemm.hip -- bf16/f16 MFMA GEMM for gfx950 with fused epilogues.
//
//   C[M,N] = epilogue( A[M,K] . B[N,K]^T )        fp32 accumulation on v_mfma_f32_16x16x32_{bf16,f16} (gemm_fl_kernel) or
//                                                 v_mfma_f32_32x32x16_{bf16,f16} (gemm_pp_kernel, gemm_sm_kernel)
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
// Common to the kernels below (each kernel's own comment has its tile, ring and schedule):
//   * operands stream L2 -> LDS with LDS-DMA (global_load_lds_dwordx4 / buffer_load ... lds: no VGPR round trip) into a ring
//     of stages, ahead of the math, retired by COUNTED s_waitcnt vmcnt(N) and raw s_barrier (never __syncthreads(), which
//     would drain vmcnt);
//   * LDS images are lane-linear (an LDS-DMA requirement), so the bank-conflict swizzle is applied to the per-lane SOURCE
//     address and undone on the fragment read (cdna guide rule 21):
//       row-major operand stage, 64-deep  [rows][64 k] (128-byte rows) : chunk' = chunk ^ ((row>>1)&7)    ds_read_b128
//       row-major operand stage, 32-deep  [rows][32 k] (64-byte rows)  : slot'  = slot ^ ((row>>2)&3)     ds_read_b128
//       k-major operand stage             [k][256 cols]               : slot'  = slot ^ f(k)             ds_read_b64_tr_b16
//     (conflict-free in the LDS bank model of tests/hipemu and by SQ_LDS_BANK_CONFLICT on MI355X);
//   * the MFMA is issued "swapped" (A-operand = B/W fragment, B-operand = A/X fragment) so each lane ends up
//     with 4 consecutive output columns of one output row; the epilogue rounds to the storage dtype, stages the
//     wave's tile in LDS and writes full row segments (bias / activation / residual / accumulate fused there);
//   * workgroup ids are remapped XCD-aware (8 XCDs, private L2s): each XCD owns a contiguous chunk of a grouped
//     (8 M-tiles wide) tile order, so concurrently resident tiles share A/B panels in L2.
// Three kernels (profiles/r01_gemm_variants.md, r02_gemm_variants.md have the measurements that led to them):
//   gemm_fl_kernel  256 x 256 tile, 4 waves x (128 x 128) on v_mfma_f32_16x16x32, one wave per SIMD, 512 registers, 64-deep
//                   "full-line" stages in 5 half-slots of 32 KiB (all 160 KiB of LDS): every layout when K % 64 == 0 (all
//                   Llama / BERT / CLIP / GPT-2 products, forward and backward)
//   gemm_pp_kernel  256 x 256 tile, 8 waves in 2 groups one phase apart (ping-pong) on v_mfma_f32_32x32x16, 32-deep stages
//                   in a 4-stage ring: any K (ragged token counts in dW, odd hidden sizes)
//   gemm_sm_kernel  128 x 128 tile, 4 waves x (64 x 64) on v_mfma_f32_32x32x16, two workgroups per CU: forward products whose
//                   256 x 256 grid cannot spread over the GPU (a CLIP tower's 577 tokens)
#include <limits.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "gemv.h"

namespace tamd {

constexpr int kBM = 256, kBN = 256;
constexpr int kGemmThreads = 512;
constexpr int kStageRowBytes = 64 * 2 + 16;          // epilogue staging row (64 cols + 16 B pad)
constexpr int kStageWaveBytes = 128 * kStageRowBytes;
constexpr int kSubK = 32;                       // K per ring stage
constexpr int kStageOperand = 256 * kSubK * 2;  // 16 KiB
constexpr int kStageBytes = 2 * kStageOperand;  // A + B
constexpr int kRing = 4;
constexpr int kGemmSmem = (kRing * kStageBytes > 8 * kStageWaveBytes) ? kRing * kStageBytes : 8 * kStageWaveBytes;

__device__ __attribute__((aligned(16))) static const unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

// Diagnostic build only (tamd_gemm_set_clock_buffer, include/tamd_diag.h): every workgroup of the 4-wave kernels
// stores {shader-clock ticks, 100 MHz real-time ticks} of its K loop (prologue included, epilogue excluded) at
// clock[2 * workgroup]: ticks ratio = the clock the kernel ran at, 64 MFMAs x 32 cycles per stage / ticks = how busy
// the matrix pipe was.  tools/gemm_clock.py
#ifdef TAMD_DIAG
#define TAMD_CLOCK_BEGIN \
  const unsigned long long clk_t0 = g.trace ? device_clock() : 0ull, clk_r0 = g.trace ? device_realtime() : 0ull;
#define TAMD_CLOCK_END                                          \
  if (g.trace != nullptr && threadIdx.x == 0) {                 \
    g.trace[2 * (size_t)vbid] = device_clock() - clk_t0;  \
    g.trace[2 * (size_t)vbid + 1] = device_realtime() - clk_r0; \
  }
// ... and a timeline (tamd_gemm_set_timeline_buffer): {real-time stamp at kernel entry, at exit (after the way out), XCC id}
// at timeline[3 * workgroup]: how long the grid's tail is, whether the XCDs finish together.  tools/gemm_timeline.py
#define TAMD_TIMELINE_BEGIN const unsigned long long tl_r0 = g.timeline ? device_realtime() : 0ull;
#define TAMD_TIMELINE_END                                                                            \
  if (g.timeline != nullptr && (threadIdx.x & 63) == 0 && wave_id_uniform() == 0) {                  \
    g.timeline[3 * (size_t)blockIdx.x] = tl_r0;                                                      \
    g.timeline[3 * (size_t)blockIdx.x + 1] = device_realtime();                                      \
    g.timeline[3 * (size_t)blockIdx.x + 2] = (unsigned long long)device_xcc_id();                     \
  }
#else
#define TAMD_CLOCK_BEGIN
#define TAMD_CLOCK_END
#define TAMD_TIMELINE_BEGIN
#define TAMD_TIMELINE_END
#endif

struct GemmArgs {
  const void* A;
  const void* B;
  void* C;
  const void* bias;
  const void* R;
  int64_t M, N, K, lda, ldb, ldc, ldr;
  int tiles_m, tiles_n;
  unsigned long long* trace;  // diagnostic: per-phase shader-clock stamps of workgroup 0 (tamd_gemm_trace)
  unsigned long long* timeline;  // diagnostic: entry / exit real-time stamps and XCC id per workgroup (gemm_fl_kernel)
  // split-K (gemm_fl_kernel with EPI = kEpiSplitK): workgroup id = tile * splits + split; split s reduces stages
  // [s*stages_per_split, ...) and writes an fp32 partial tile to ws[s][M][N]; splitk_reduce_kernel sums and rounds
  float* ws;
  int splits, stages_per_split;
  // SwiGLU epilogue (kEpiSwiGLU): B = [gate rows (n_half) ; up rows (n_half)], C2 = act [M, n_half] (ldc2), C = the
  // [M, 2*n_half] gate|up output or nullptr when nothing will be differentiated
  void* C2;
  int64_t ldc2, n_half;
  // SwiGLU-backward epilogue (kEpiSwiGLUBwd): N = n_half = I, R = the saved gate|up [M, 2I] (ldr), C = d_gate|d_up [M, 2I] (ldc)
  // column-scale epilogue (kEpiColScale): C[m, n] = round((acc + bias[n]) * (n < scale_cols ? col_scale : 1)) -- the query
  // columns of a q|k|v projection leave carrying the attention kernels' scale*log2(e) (tamd_attn_params.q_prescaled)
  int64_t scale_cols;
  float col_scale;
  // segmented output (tamd_gemm_seg; dW layout, plain / accumulate): rows [0, seg_row1) of C live at C, [seg_row1, seg_row2) at
  // C_seg1, [seg_row2, M) at C_seg2 (each with leading dimension ldc; seg_row* multiples of 256, 0 = no such segment) -- the
  // gradient of a fused q|k|v / gate|up weight written straight into the member parameters' separate gradient buffers
  void* C_seg1;
  void* C_seg2;
  int64_t seg_row1, seg_row2;
};
// the output base (as if row 0 were the matrix's first row) of the tile whose first row is m0
template <typename T>
__device__ __forceinline__ void* gemm_seg_base(const GemmArgs& g, int64_t m0) {
  if (g.seg_row2 > 0 && m0 >= g.seg_row2) return reinterpret_cast<T*>(g.C_seg2) - g.seg_row2 * g.ldc;
  if (g.seg_row1 > 0 && m0 >= g.seg_row1) return reinterpret_cast<T*>(g.C_seg1) - g.seg_row1 * g.ldc;
  return g.C;
}
constexpr int kEpiSplitK = 100;
constexpr int kEpiSwiGLU = 101;
constexpr int kEpiColScale = 104;
constexpr int kEpiSwiGLUBwd = 105;

// the LlamaMLP inner product (models/llama/modeling_llama.py:174-176; same expression as swiglu_fwd_kernel in
// elementwise.hip, so the fused epilogue and the stand-alone kernel agree bit for bit)
__device__ __forceinline__ float gemm_silu(float x) { return x * fast_sigmoid(x); }

template <int ACT>
__device__ __forceinline__ float gemm_act(float x) {
  if (ACT == TAMD_ACT_GELU_ERF) return gelu_erf_f(x);  // (tamd_device.h: one definition with the activation kernels)
  if (ACT == TAMD_ACT_GELU_TANH) return 0.5f * x * (1.f + tanhf(0.79788456080286535588f * (x + 0.044715f * x * x * x)));
  if (ACT == TAMD_ACT_QUICK_GELU) return x * fast_sigmoid(1.702f * x);  // (= act_fwd_f of elementwise.hip)
  if (ACT == TAMD_ACT_SILU) return x * fast_sigmoid(x);
  return x;
}


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

template <typename T, bool KMAJOR>
__device__ __forceinline__ void pp_issue(const T* __restrict__ G, int64_t ld, int64_t rc0, int64_t nrc, int64_t k0,
                                         int64_t K, char* smem, unsigned off, int wave, int lane, int i0 = 0,
                                         int i1 = 2) {
#pragma unroll
  for (int i = i0; i < i1; ++i) {
    const int inst = wave * 2 + i;  // 16 wave-instructions per operand stage, 2 per wave
    glds16(pp_src<T, KMAJOR>(G, ld, rc0, nrc, k0, K, inst, lane), smem, off + (unsigned)inst * 1024u);
  }
}

// byte offset of a lane's fragment inside an operand stage (first of the two reads when k-major), and the read:
// 8 k-values (16 B) of row/column rc32 + (lane & 31), k-step ks of the sub-tile
template <bool KMAJOR>
__device__ __forceinline__ unsigned pp_frag_off(int rc32, int ks, int lane) {
  const int hi = lane >> 5;
  if (KMAJOR) {
    const int kq = (lane & 15) >> 2;
    const int col = rc32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int slot = (col >> 3) ^ (kq << 2);  // (k & 3) == kq : the other k terms are multiples of 4
    return (unsigned)(ks * 16 + hi * 8 + kq) * 512u + (unsigned)slot * 16u + (unsigned)(col & 7) * 2u;
  }
  const int row = rc32 + (lane & 31);
  return (unsigned)row * 64u + (unsigned)(((ks * 2 + hi) ^ ((row >> 2) & 3)) * 16);
}
template <bool KMAJOR>
__device__ __forceinline__ u32x4 pp_frag_at(const char* smem, unsigned off) {
  if (KMAJOR) {  // two transposing 8-byte reads, k rows +0..3 and +4..7
    const u32x2 lo = lds_read8_tr16(smem, off);
    const u32x2 h2 = lds_read8_tr16(smem, off + 4u * 512u);
    return u32x4{lo[0], lo[1], h2[0], h2[1]};
  }
  return lds_read16(smem, off);
}

// XCD-aware grouped tile order: workgroup id -> (tile_m, tile_n)
__device__ __forceinline__ void gemm_tile_of_block(const GemmArgs& g, int bid, int* tile_m, int* tile_n) {
  const int nwg = g.tiles_m * g.tiles_n;
  const int xcd = bid & 7, in_xcd = bid >> 3;
  const int q = nwg >> 3, rr = nwg & 7;
  const int logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + in_xcd;
  constexpr int GROUP_M = 8;
  const int group_size = GROUP_M * g.tiles_n;
  const int grp = logical / group_size;
  const int first_m = grp * GROUP_M;
  const int gm = (g.tiles_m - first_m < GROUP_M) ? (g.tiles_m - first_m) : GROUP_M;
  *tile_m = first_m + (logical % group_size) % gm;
  *tile_n = (logical % group_size) / gm;
}

// Epilogue of one wave over a (HALVES*64) x (NCOLS) piece of C at (row0, col0).  Per 64 rows: `stage(half, bias4)` rounds
// the accumulators (+bias, +activation) into this wave's private LDS region (row pitch NCOLS*2 + 16 bytes; the
// accumulator layout is the caller's business: stage32 / stage16 below), then full-row 16-byte stores (+residual / +C)
// leave from there.
template <typename T, int EPI, int ACT, int NCOLS, int HALVES, typename StageFn>
__device__ __forceinline__ void gemm_epilogue_rows(const GemmArgs& g, char* smem, unsigned st_off, int64_t row0, int64_t col0,
                                                   int lane, StageFn stage) {
  constexpr int ROWB = NCOLS * 2 + 16;   // staged row + 16 B pad
  constexpr int SLOTS = NCOLS / 8;       // 16-byte slots per row
  constexpr int RPI = 64 / SLOTS;        // rows per wave instruction on the way out
  // Addressing (round 5): the piece's first element is a wave-uniform buffer base, a lane's position a 32-bit byte offset
  // that steps by RPI rows per instruction, and the matrix edge is the hardware's range check (rows past M: beyond the
  // buffer's size; a lane whose 8 columns lie past N gets an offset outside it once) -- per store: one ds_read, one v_add,
  // one buffer_store.  (Before: per store a 64-bit row * ldc product, a 64-bit compare and an exec-masked branch -- ~13
  // instructions, 32 stores per wave, with the matrix pipe idle: tools/gemm_isa.sh counted 1180 instructions in the plain way
  // out, ~700 now.)
  const int64_t rows_left = g.M - row0;  // (<= 0: the tile's second wave row lies wholly past a ragged M -- an empty buffer)
  const unsigned nrows = rows_left <= 0 ? 0u : (unsigned)(rows_left < HALVES * 64 ? rows_left : HALVES * 64);
  const int lrow = lane / SLOTS, slot = lane % SLOTS;
  const bool col_ok = col0 + slot * 8 < g.N;
  T* Cw = reinterpret_cast<T*>(g.C) + row0 * g.ldc + col0;
  // (a lane past N keeps its out-of-range offset: its step is 0, so the 32-bit offset cannot wrap back into the buffer)
  const unsigned c_bytes = nrows * (unsigned)g.ldc * 2u, c_step = col_ok ? (unsigned)RPI * (unsigned)g.ldc * 2u : 0u;
  const unsigned c_off0 = col_ok ? ((unsigned)lrow * (unsigned)g.ldc + (unsigned)slot * 8u) * 2u : 0xfffffff0u;
  // operands the way out reads from memory (residual / previous C): all of a half's loads are issued BEFORE the accumulators
  // are rounded and staged, so their latency overlaps that work and 16 KiB per wave are in flight instead of 4 (a way out
  // that loads inside its row loop is latency-bound at ~2.7 TB/s)
  constexpr int NR = (EPI == TAMD_EPI_RESIDUAL || EPI == TAMD_EPI_ACCUM) ? 1 : 0;
  constexpr int NIT = 64 / RPI;
  const T* Rw = (EPI == TAMD_EPI_ACCUM) ? Cw : (reinterpret_cast<const T*>(g.R) + row0 * g.ldr + col0);
  const unsigned ldr_ = (EPI == TAMD_EPI_ACCUM) ? (unsigned)g.ldc : (unsigned)g.ldr;
  const unsigned r_bytes = nrows * ldr_ * 2u, r_step = col_ok ? (unsigned)RPI * ldr_ * 2u : 0u;
  const unsigned r_off0 = col_ok ? ((unsigned)lrow * ldr_ + (unsigned)slot * 8u) * 2u : 0xfffffff0u;
#pragma unroll
  for (int half = 0; half < HALVES; ++half) {
    u32x4 pre[NR > 0 ? NIT : 1];
    if (NR > 0) {
      unsigned ro = r_off0 + (unsigned)half * (unsigned)NIT * r_step;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        pre[it] = buf_load16_rng(Rw, r_bytes, ro);  // (outside the matrix: zeros)
        ro += r_step;
      }
      sched_fence();
    }
    stage(half);
    wave_lockstep_point();  // wave-private region: this wave's writes are ordered before its reads
    unsigned co = c_off0 + (unsigned)half * (unsigned)NIT * c_step;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      u32x4 v = lds_read16(smem, st_off + (unsigned)(it * RPI + lrow) * ROWB + (unsigned)slot * 16u);
      if (EPI == TAMD_EPI_RESIDUAL || EPI == TAMD_EPI_ACCUM) {
        float a[8], b[8];
        unpack16<T>(v, a);
        unpack16<T>(pre[NR == 1 ? it : 0], b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += b[e];
        v = pack16<T>(a);
      }
      buf_store16_rng(Cw, c_bytes, co, v);  // (outside the matrix: nothing is stored)
      co += c_step;
    }
    wave_lockstep_point();
  }
}

// bias of 4 consecutive columns starting at global column gn (zero when the epilogue has none)
template <typename T, int EPI>
__device__ __forceinline__ void gemm_bias4(const GemmArgs& g, int64_t gn, float* bv) {
  bv[0] = bv[1] = bv[2] = bv[3] = 0.f;
  const T* bias = reinterpret_cast<const T*>(g.bias);
  if (EPI == TAMD_EPI_BIAS || EPI == TAMD_EPI_BIAS_ACT || ((EPI == TAMD_EPI_RESIDUAL || EPI == kEpiColScale) && bias != nullptr)) {
    if (gn < g.N) {  // N % 8 == 0 and gn % 4 == 0: the 4 columns are valid together
      const u32x2 bq = ld8(bias + gn);
      bv[0] = elem<T>::to_f32((typename elem<T>::raw)(bq[0] & 0xffffu));
      bv[1] = elem<T>::to_f32((typename elem<T>::raw)(bq[0] >> 16));
      bv[2] = elem<T>::to_f32((typename elem<T>::raw)(bq[1] & 0xffffu));
      bv[3] = elem<T>::to_f32((typename elem<T>::raw)(bq[1] >> 16));
    }
  }
}
// 4 accumulator values of one output row -> rounded (+bias, +activation) -> 8 staged bytes
// (sc: the column-scale epilogue's factor of these 4 columns, applied BEFORE the one rounding)
template <typename T, int EPI, int ACT>
__device__ __forceinline__ u32x2 gemm_round4(float a0, float a1, float a2, float a3, const float* bv, float sc = 1.f) {
  // (epilogues that cannot carry a bias skip the add: 128 packed adds of zero per wave in every plain way out)
  constexpr bool MAY_BIAS = (EPI == TAMD_EPI_BIAS || EPI == TAMD_EPI_BIAS_ACT || EPI == TAMD_EPI_RESIDUAL || EPI == kEpiColScale);
  float v[4] = {a0, a1, a2, a3};
  if (MAY_BIAS) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += bv[e];
  }
  if (EPI == TAMD_EPI_BIAS_ACT) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gemm_act<ACT>(round_through<T>(v[e]));
  }
  if (EPI == kEpiColScale) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= sc;
  }
  return u32x2{pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
}
// factor of the 4 consecutive columns starting at gn (scale_cols % 4 == 0: they are scaled together)
template <int EPI>
__device__ __forceinline__ float gemm_colscale4(const GemmArgs& g, int64_t gn) {
  return (EPI == kEpiColScale && gn < g.scale_cols) ? g.col_scale : 1.f;
}
// 32x32x16 accumulators (gemm_pp_kernel): acc[ni][mi][r] = D[n = ni*32 + (r&3) + 8*(r>>2) + 4*hi][m = mi*32 + l31]
template <typename T, int EPI, int ACT, int NI, int MI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[NI][MI], char* smem, unsigned st_off,
                                              int64_t row0, int64_t col0, int lane) {
  constexpr int ROWB = NI * 32 * 2 + 16;
  const int hi = lane >> 5, l31 = lane & 31;
  auto stage_as = [&](auto epi_tag, int half) {
    constexpr int E = decltype(epi_tag)::value;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int nl = ni * 32 + 8 * qd + 4 * hi;  // first of 4 consecutive local columns
        float bv[4];
        gemm_bias4<T, E>(g, col0 + nl, bv);
        const float sc = gemm_colscale4<E>(g, col0 + nl);
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
          const int mi = half * 2 + m2;
          lds_write8(smem, st_off + (unsigned)(m2 * 32 + l31) * ROWB + (unsigned)nl * 2u,
                     gemm_round4<T, E, ACT>(acc[ni][mi][qd * 4 + 0], acc[ni][mi][qd * 4 + 1], acc[ni][mi][qd * 4 + 2],
                                            acc[ni][mi][qd * 4 + 3], bv, sc));
        }
      }
    }
  };
  gemm_epilogue_rows<T, EPI, ACT, NI * 32, MI / 2>(g, smem, st_off, row0, col0, lane,
                                                   [&](int half) { stage_as(std::integral_constant<int, EPI>{}, half); });
}

// 16x16x32 accumulators (gemm_fl_kernel): acc[nb][mb][r] = D[n = nb*16 + 4*(lane>>4) + r][m = mb*16 + (lane&15)],
// 8 x 8 blocks = a 128 x 128 piece of C
template <typename T, int EPI, int ACT>
__device__ __forceinline__ void gemm_epilogue16(const GemmArgs& g, f32x4 (&acc)[8][8], char* smem, unsigned st_off,
                                                int64_t row0, int64_t col0, int lane) {
  constexpr int ROWB = 128 * 2 + 16;
  const int g4 = lane >> 4, l15 = lane & 15;
  auto stage_as = [&](auto epi_tag, int half) {
    constexpr int E = decltype(epi_tag)::value;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const int nl = nb * 16 + 4 * g4;  // first of 4 consecutive local columns
      float bv[4];
      gemm_bias4<T, E>(g, col0 + nl, bv);
      const float sc = gemm_colscale4<E>(g, col0 + nl);
#pragma unroll
      for (int m4 = 0; m4 < 4; ++m4) {
        const f32x4 a = acc[nb][half * 4 + m4];
        lds_write8(smem, st_off + (unsigned)(m4 * 16 + l15) * ROWB + (unsigned)nl * 2u,
                   gemm_round4<T, E, ACT>(a[0], a[1], a[2], a[3], bv, sc));
      }
      // (one column block at a time: left alone the scheduler reads all 128 accumulators of the half out of the AGPRs first --
      // 128 live VGPRs beside the 64 of a half's residual rows: the f16 residual kernel spilled 3 of them)
      if (E == TAMD_EPI_RESIDUAL || E == TAMD_EPI_ACCUM) sched_fence();
    }
  };
  gemm_epilogue_rows<T, EPI, ACT, 128, 2>(g, smem, st_off, row0, col0, lane,
                                          [&](int half) { stage_as(std::integral_constant<int, EPI>{}, half); });
}

// SwiGLU epilogue of one wave of gemm_fl_kernel<..., kEpiSwiGLU>.  The tile's 256 B-rows are 8 blocks of 32 weight rows,
// alternately from the gate and the up half of the fused [2*I, K] weight (block b: feature block b>>1, part b&1), so
// of a wave's eight 16-column accumulator blocks nb = 4p + 2*part + sub (p = feature block, sub = its 16-column half)
// the gate and up blocks of one feature (nb, nb + 2) have identical lane layouts:
//     act = round(round(silu(round(g))) * round(u))          lane-local, the roundings of the unfused path
// Per 64 output rows: stage [gate 64 | up 64] and [act 64] in this wave's LDS region, then full 128-byte row segments
// to C[m, f0..] (gate), C[m, I + f0..] (up) -- streaming stores (saved for the backward only), skipped when C == nullptr --
// and C2[m, f0..] (act).
template <typename T>
__device__ __forceinline__ void gemm_epilogue_swiglu(const GemmArgs& g, f32x4 (&acc)[8][8], char* smem, unsigned st_off,
                                                     int64_t row0, int64_t f0, int lane) {
  constexpr int GU_ROWB = 128 * 2 + 16, ACT_ROWB = 64 * 2 + 16;
  constexpr unsigned ACT_OFF = 64u * GU_ROWB;
  const int g4 = lane >> 4, l15 = lane & 15;
  T* C = reinterpret_cast<T*>(g.C);
  const int64_t I = g.n_half;
  const int64_t rows_left = g.M - row0;  // (<= 0: this wave's rows lie wholly past a ragged M -- empty buffers)
  const unsigned nrows = rows_left <= 0 ? 0u : (unsigned)(rows_left < 128 ? rows_left : 128);
  // gate | up rows: lane -> (row lane >> 4, slot lane & 15: slots 0..7 the gate features f0.., 8..15 the up features at + I)
  T* Cw = C != nullptr ? C + row0 * g.ldc + f0 : nullptr;
  const bool gu_ok = f0 + (lane & 7) * 8 < I;
  const unsigned gu_bytes = nrows * (unsigned)g.ldc * 2u, gu_step = gu_ok ? 4u * (unsigned)g.ldc * 2u : 0u;
  const unsigned gu_off0 = gu_ok ? ((unsigned)(lane >> 4) * (unsigned)g.ldc + (unsigned)((lane & 8) ? I : 0) + (unsigned)(lane & 7) * 8u) * 2u
                                 : 0xfffffff0u;
  // act rows: lane -> (row lane >> 3, slot lane & 7)
  T* C2w = reinterpret_cast<T*>(g.C2) + row0 * g.ldc2 + f0;
  const bool act_ok = f0 + (lane & 7) * 8 < I;
  const unsigned act_bytes = nrows * (unsigned)g.ldc2 * 2u, act_step = act_ok ? 8u * (unsigned)g.ldc2 * 2u : 0u;
  const unsigned act_off0 = act_ok ? ((unsigned)(lane >> 3) * (unsigned)g.ldc2 + (unsigned)(lane & 7) * 8u) * 2u : 0xfffffff0u;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const int nl = p * 32 + sub * 16 + 4 * g4;  // first of 4 consecutive local features
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) {
          const int mb = half * 4 + m4;
          // one vector conversion per PAIR and rounding point; the packed registers are what gets staged, their halves read
          // back as fp32 are the rounded values (a round_through per element is a convert + shift each: the epilogue runs
          // with the matrix pipe idle, so its instruction count is launch time -- 3705 -> ~2100 instructions per wave)
          const f32x4 ag = acc[4 * p + sub][mb], au = acc[4 * p + 2 + sub][mb];
          const u32x2 gp = {pack2<T>(ag[0], ag[1]), pack2<T>(ag[2], ag[3])};
          const u32x2 up = {pack2<T>(au[0], au[1]), pack2<T>(au[2], au[3])};
          float gg[4], uu[4], sl[4];
          unpack2<T>(gp[0], gg[0], gg[1]);
          unpack2<T>(gp[1], gg[2], gg[3]);
          unpack2<T>(up[0], uu[0], uu[1]);
          unpack2<T>(up[1], uu[2], uu[3]);
#pragma unroll
          for (int e = 0; e < 4; ++e) sl[e] = gemm_silu(gg[e]);
          const u32x2 sp = {pack2<T>(sl[0], sl[1]), pack2<T>(sl[2], sl[3])};
          unpack2<T>(sp[0], sl[0], sl[1]);
          unpack2<T>(sp[1], sl[2], sl[3]);
          const unsigned r = (unsigned)(m4 * 16 + l15);
          lds_write8(smem, st_off + r * GU_ROWB + (unsigned)nl * 2u, gp);
          lds_write8(smem, st_off + r * GU_ROWB + (unsigned)(64 + nl) * 2u, up);
          lds_write8(smem, st_off + ACT_OFF + r * ACT_ROWB + (unsigned)nl * 2u,
                     u32x2{pack2<T>(sl[0] * uu[0], sl[1] * uu[1]), pack2<T>(sl[2] * uu[2], sl[3] * uu[3])});
        }
      }
    }
    wave_lockstep_point();
    // (addressing as in gemm_epilogue_rows: wave-uniform buffer base, 32-bit lane offsets, the hardware's range check for the
    // matrix edge)
    if (C != nullptr) {
      unsigned go = gu_off0 + (unsigned)half * 16u * gu_step;
#pragma unroll
      for (int it = 0; it < 16; ++it) {  // 64 rows x 16 slots of 16 B: 4 rows per wave instruction
        const u32x4 v = lds_read16(smem, st_off + (unsigned)(it * 4 + (lane >> 4)) * GU_ROWB + (unsigned)(lane & 15) * 16u);
        buf_store16_rng_nt(Cw, gu_bytes, go, v);
        go += gu_step;
      }
    }
    unsigned ao = act_off0 + (unsigned)half * 8u * act_step;
#pragma unroll
    for (int it = 0; it < 8; ++it) {  // 64 rows x 8 slots: 8 rows per wave instruction
      const u32x4 v = lds_read16(smem, st_off + ACT_OFF + (unsigned)(it * 8 + (lane >> 3)) * ACT_ROWB + (unsigned)(lane & 7) * 16u);
      buf_store16_rng(C2w, act_bytes, ao, v);
      ao += act_step;
    }
    wave_lockstep_point();
  }
}

// Backward of the SiLU*up product as the way out of the down projection's dX GEMM (gemm_fl_kernel<..., kEpiSwiGLUBwd>;
// models/llama/modeling_llama.py:174-176 differentiated): the product is d_act[M, I] = dY . W_down, and instead of storing it
// (0.94 GB at Llama-3-8B 8 x 4096) for swiglu_bwd_kernel to read back beside gate and up, the wave that holds a 128 x 128 piece
// of it reads the same piece of gate and of up (R = the saved [M, 2I] gate|up matrix) and stores
//     d_up   = round(d * round(silu(g)))          d_gate = round(round(d * u) * silu'(g))           d = round(acc)
// to C = d_gate|d_up [M, 2I] -- the expressions and roundings of swiglu_bwd_kernel (elementwise.hip) on the rounded d_act, so
// the two paths agree bit for bit.  Per 64 rows the accumulators are rounded and staged in the wave's LDS region; then per wave
// instruction 4 rows x 16 slots of 8 features, in quarters of 32 rows whose gate / up loads run one quarter ahead.
__device__ __forceinline__ float gemm_dsilu(float x) {  // = dsilu_f of elementwise.hip
  const float s = fast_sigmoid(x);
  return s * (1.f + x * (1.f - s));
}
// the SiLU*up backward of one packed pair (element 2i in the low half of the word, 2i + 1 in the high half): the scalar
// expressions of swiglu_bwd_kernel per element, written on two-element vectors so that the multiplies and adds of a pair are ONE
// packed instruction each and a word is unpacked and re-packed as a unit (left to itself the vectoriser paired element i of
// word 0 with element i of word 1 and paid four shuffles per two words to put the halves back: 18 -> 14 instructions per element)
template <typename T>
__device__ __forceinline__ void gemm_swiglu_bwd_pair(unsigned dw, unsigned gw, unsigned uw, unsigned* dgw, unsigned* duw) {
  float d0, d1, g0, g1, u0, u1, a0, a1, b0, b1;
  unpack2<T>(dw, d0, d1);
  unpack2<T>(gw, g0, g1);
  unpack2<T>(uw, u0, u1);
  const f32x2 d = {d0, d1}, g = {g0, g1}, u = {u0, u1};
  const f32x2 t = g * -1.44269504088896340736f;  // fast_sigmoid (tamd_device.h), two at a time
  const f32x2 den = 1.f + f32x2{fast_exp2(t.x), fast_exp2(t.y)};
  const f32x2 sig = {fast_rcp(den.x), fast_rcp(den.y)};
  const f32x2 sl = g * sig;                      // gemm_silu
  unpack2<T>(pack2<T>(sl.x, sl.y), a0, a1);      // round(silu(g))
  const f32x2 rs = {a0, a1};
  const f32x2 du = d * rs;
  const f32x2 tu = d * u;
  unpack2<T>(pack2<T>(tu.x, tu.y), b0, b1);      // round(d * u)
  const f32x2 rt = {b0, b1};
  const f32x2 ds = sig * (1.f + g * (1.f - sig));  // gemm_dsilu
  const f32x2 dg = rt * ds;
  *dgw = pack2<T>(dg.x, dg.y);
  *duw = pack2<T>(du.x, du.y);
}
// (plain functions with everything passed by value, not lambdas: through a by-reference closure hipcc lost the wave-uniformity
// of the buffer bases and wrapped every load and store in a waterfall loop -- 259 v_readfirstlane, 65 exec-masked loops)
constexpr int kSwbQIT = 8;  // a quarter = 32 rows = 8 wave instructions of 4 rows x 16 slots
template <typename T>
__device__ __forceinline__ void gemm_swb_request(u32x4 (&pg)[kSwbQIT], u32x4 (&pu)[kSwbQIT], const T* Gw, const T* Uw,
                                                 unsigned r_bytes, unsigned ro, unsigned r_step) {
#pragma unroll
  for (int it = 0; it < kSwbQIT; ++it) {
    pg[it] = buf_load16_rng(Gw, r_bytes, ro);  // (outside the matrix: zeros)
    pu[it] = buf_load16_rng(Uw, r_bytes, ro);
    ro += r_step;
  }
}
template <typename T>
__device__ __forceinline__ void gemm_swb_finish(const u32x4 (&pg)[kSwbQIT], const u32x4 (&pu)[kSwbQIT], const char* smem,
                                                unsigned lds_off, T* DGw, T* DUw, unsigned c_bytes, unsigned co,
                                                unsigned c_step) {
  constexpr int ROWB = 128 * 2 + 16;
#pragma unroll
  for (int it = 0; it < kSwbQIT; ++it) {
    const u32x4 v = lds_read16(smem, lds_off + (unsigned)(it * 4) * ROWB);
    u32x4 dgv, duv;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      unsigned a, b;
      gemm_swiglu_bwd_pair<T>(v[w], pg[it][w], pu[it][w], &a, &b);
      dgv[w] = a;
      duv[w] = b;
    }
    buf_store16_rng(DGw, c_bytes, co, dgv);  // (outside the matrix: nothing is stored)
    buf_store16_rng(DUw, c_bytes, co, duv);
    co += c_step;
  }
}
template <typename T>
__device__ __forceinline__ void gemm_swb_stage(f32x4 (&acc)[8][8], int half, char* smem, unsigned st_off, int g4, int l15) {
  constexpr int ROWB = 128 * 2 + 16;
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
    const int nl = nb * 16 + 4 * g4;  // first of 4 consecutive local columns
#pragma unroll
    for (int m4 = 0; m4 < 4; ++m4) {
      const f32x4 a = acc[nb][half * 4 + m4];
      lds_write8(smem, st_off + (unsigned)(m4 * 16 + l15) * ROWB + (unsigned)nl * 2u,
                 u32x2{pack2<T>(a[0], a[1]), pack2<T>(a[2], a[3])});
    }
    sched_fence();  // (one column block at a time: see gemm_epilogue16)
  }
}
template <typename T>
__device__ __forceinline__ void gemm_epilogue_swiglu_bwd(const GemmArgs& g, f32x4 (&acc)[8][8], char* smem, unsigned st_off,
                                                         int64_t row0, int64_t col0, int lane) {
  constexpr int ROWB = 128 * 2 + 16;
  constexpr int QIT = kSwbQIT;
  const int g4 = lane >> 4, l15 = lane & 15;
  const int64_t I = g.n_half;
  const int64_t rows_left = g.M - row0;  // (<= 0: this wave's rows lie wholly past a ragged M -- empty buffers)
  const unsigned nrows = rows_left <= 0 ? 0u : (unsigned)(rows_left < 128 ? rows_left : 128);
  const int lrow = lane >> 4, slot = lane & 15;
  const bool col_ok = col0 + slot * 8 < I;
  // (addressing as in gemm_epilogue_rows: wave-uniform buffer bases -- the up halves are the gate bases + I columns with the same
  // byte count, so a row past M is out of range for both -- 32-bit lane offsets, the hardware's range check for the matrix edge)
  const T* Gw = reinterpret_cast<const T*>(g.R) + row0 * g.ldr + col0;
  const T* Uw = Gw + I;
  T* DGw = reinterpret_cast<T*>(g.C) + row0 * g.ldc + col0;
  T* DUw = DGw + I;
  const unsigned r_bytes = nrows * (unsigned)g.ldr * 2u, r_step = col_ok ? 4u * (unsigned)g.ldr * 2u : 0u;
  const unsigned r_off0 = col_ok ? ((unsigned)lrow * (unsigned)g.ldr + (unsigned)slot * 8u) * 2u : 0xfffffff0u;
  const unsigned c_bytes = nrows * (unsigned)g.ldc * 2u, c_step = col_ok ? 4u * (unsigned)g.ldc * 2u : 0u;
  const unsigned c_off0 = col_ok ? ((unsigned)lrow * (unsigned)g.ldc + (unsigned)slot * 8u) * 2u : 0xfffffff0u;
  const unsigned lds_q0 = st_off + (unsigned)lrow * ROWB + (unsigned)slot * 16u, lds_q1 = lds_q0 + 32u * ROWB;
  // The 128 rows go out as four quarters, the gate / up loads of quarter q + 1 requested BEFORE quarter q is computed and stored
  // (two register sets of 16 loads: 16 KiB per wave always in flight under ~900 VALU instructions).  Measured LEVEL with the first
  // version (all 32 loads of a half issued and waited for together, 18 instructions per element): the launch costs the plain
  // product + 0.47-0.50 ms either way (profiles/r06o_, r06r_gemm_swiglu_bwd_ab.jsonl) = the way out's extra 2.8 GB at ~6 TB/s --
  // the 256 workgroups of a round reach their ways out together, so it is HBM time with no K loop beside it, not
  // load latency or arithmetic.  Kept for the smaller code (4040 against 4600 instructions per wave).
  u32x4 pga[QIT], pua[QIT], pgb[QIT], pub[QIT];
  gemm_swb_request<T>(pga, pua, Gw, Uw, r_bytes, r_off0, r_step);
  sched_fence();
  gemm_swb_stage<T>(acc, 0, smem, st_off, g4, l15);  // d_act rounded as the plain way out stores it
  wave_lockstep_point();  // wave-private region: this wave's writes are ordered before its reads
  gemm_swb_request<T>(pgb, pub, Gw, Uw, r_bytes, r_off0 + (unsigned)QIT * r_step, r_step);
  sched_fence();
  gemm_swb_finish<T>(pga, pua, smem, lds_q0, DGw, DUw, c_bytes, c_off0, c_step);
  sched_fence();
  gemm_swb_request<T>(pga, pua, Gw, Uw, r_bytes, r_off0 + (unsigned)(2 * QIT) * r_step, r_step);
  sched_fence();
  gemm_swb_finish<T>(pgb, pub, smem, lds_q1, DGw, DUw, c_bytes, c_off0 + (unsigned)QIT * c_step, c_step);
  wave_lockstep_point();
  gemm_swb_stage<T>(acc, 1, smem, st_off, g4, l15);
  wave_lockstep_point();
  gemm_swb_request<T>(pgb, pub, Gw, Uw, r_bytes, r_off0 + (unsigned)(3 * QIT) * r_step, r_step);
  sched_fence();
  gemm_swb_finish<T>(pga, pua, smem, lds_q0, DGw, DUw, c_bytes, c_off0 + (unsigned)(2 * QIT) * c_step, c_step);
  sched_fence();
  gemm_swb_finish<T>(pgb, pub, smem, lds_q1, DGw, DUw, c_bytes, c_off0 + (unsigned)(3 * QIT) * c_step, c_step);
  wave_lockstep_point();
}

// ============================================================================================ ping-pong kernel
// 8 waves in two groups (waves 0-3 own output rows 0-127, waves 4-7 rows 128-255; waves w and w+4 share a SIMD)
// that run one phase apart:
//     phase 2j   : group 0 LOADs  fragments of sub-tile j   | group 1 COMPUTEs sub-tile j-1
//     phase 2j+1 : group 0 COMPUTEs sub-tile j (16 MFMA)    | group 1 LOADs  fragments of sub-tile j
// so on every SIMD one wave feeds the matrix pipe from registers while its partner reads LDS and issues the next
// LDS-DMA loads.  LOAD phase = 12 fragment reads + this wave's 4 pieces of sub-tile j+3 (into the stage sub-tile
// j-1 vacated) + vmcnt(8): retires the share issued two LOAD phases ago (sub-tile j+1), leaves the two newest
// batches in flight across the barrier.  Hazards: sub-tile j is read by group 0 in phase 2j and group 1 in phase
// 2j+1; its stage is rewritten by loads issued in phases 2j+2 / 2j+3 (after the barrier that ends phase 2j+1, by
// which every reader has passed lgkmcnt(0)); the data is first read in phase 2j+8, after both issuers' vmcnt waits
// (end of phases 2j+6 / 2j+7) and the barrier that ends phase 2j+7.  No s_setprio: measured -4 % here.
// TRACE: diagnostic build that stamps the shader clock at the phase boundaries of workgroup 0 (tamd_gemm_trace).
template <typename T, bool A_KM, bool B_KN, int EPI, int ACT, bool TRACE = false>
__global__ __launch_bounds__(kGemmThreads) void gemm_pp_kernel(GemmArgs g) {
  TAMD_DYN_SMEM(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 2, wn = wave & 3;  // wm = ping-pong group
  int tile_m, tile_n;
  gemm_tile_of_block(g, blockIdx.x, &tile_m, &tile_n);
  const int64_t m0 = (int64_t)tile_m * kBM, n0 = (int64_t)tile_n * kBN;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* B = reinterpret_cast<const T*>(g.B);

  f32x16 acc[2][4];  // [ni][mi]
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  const int nsub = (int)((g.K + kSubK - 1) / kSubK);
  auto issue = [&](int j) {  // this wave's share (2 A + 2 B pieces) of sub-tile j into stage j % 4
    const unsigned st = (unsigned)(j & (kRing - 1)) * kStageBytes;
    const int64_t k0 = (int64_t)j * kSubK;  // sub-tiles past the end read the zero page: counts stay uniform
    pp_issue<T, A_KM>(A, g.lda, m0, g.M, k0, g.K, smem, st, wave, lane);
    pp_issue<T, B_KN>(B, g.ldb, n0, g.N, k0, g.K, smem, st + kStageOperand, wave, lane);
  };
  issue(0);
  issue(1);
  issue(2);
  wait_vmcnt<0>();
  raw_barrier();
  if (wm == 1) raw_barrier();  // stagger: group 1 runs one phase behind group 0

  const bool tr = TRACE && g.trace != nullptr && blockIdx.x == 0 && lane == 0;
#define TAMD_STAMP(i_) \
  if (TRACE && tr && j < 32) g.trace[((size_t)wave * 32 + j) * 8 + (i_)] = device_clock();
  u32x4 xa[2][4], wb[2][2];
  for (int j = 0; j < nsub; ++j) {
    // ---------------- LOAD phase
    TAMD_STAMP(0)
    const unsigned st = (unsigned)(j & (kRing - 1)) * kStageBytes;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        wb[ks][ni] = pp_frag_at<B_KN>(smem, st + kStageOperand + pp_frag_off<B_KN>(wn * 64 + ni * 32, ks, lane));
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) xa[ks][mi] = pp_frag_at<A_KM>(smem, st + pp_frag_off<A_KM>(wm * 128 + mi * 32, ks, lane));
    }
    TAMD_STAMP(1)
    issue(j + 3);
    TAMD_STAMP(2)
    wait_vmcnt<8>();  // retires this wave's share of sub-tile j+1; sub-tiles j+2, j+3 stay in flight
    TAMD_STAMP(3)
    wait_lgkmcnt0();  // fragments are in registers: the stage may be recycled after the next barrier
    TAMD_STAMP(4)
    sched_fence();
    raw_barrier();
    TAMD_STAMP(5)
    // ---------------- COMPUTE phase (registers only)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = mfma32<T>(wb[ks][ni], xa[ks][mi], acc[ni][mi]);
    sched_fence();
    TAMD_STAMP(6)
    raw_barrier();
    TAMD_STAMP(7)
  }
#undef TAMD_STAMP
  if (wm == 0) raw_barrier();
  wait_vmcnt<0>();  // trailing (zero-page) loads must land before the epilogue reuses the LDS
  raw_barrier();
  gemm_epilogue<T, EPI, ACT, 2, 4>(g, acc, smem, (unsigned)wave * kStageWaveBytes, m0 + wm * 128, n0 + wn * 64, lane);
}

// ============================================================================================ full-line feed
// 4 waves x (128 x 128): 256 accumulator registers per wave (the unified 512-entry file of gfx950), one wave per
// SIMD, on v_mfma_f32_16x16x32: 8 x 8 accumulator blocks of 16 x 16.  Why the small MFMA: this GEMM is bounded by the
// power budget, not by its schedule -- the K loop of the 32x32x16 version issued an MFMA on 90-94 % of the cycles of a
// clock the part had lowered to 1.45-1.65 GHz (profiles/r02_gemm_variants.md) -- and a register-only stream of 16x16x32
// MFMAs on random operands sustains 2.06 PFLOP/s at 2.04 GHz where 32x32x16 sustains 1.80 at 1.76 (the 32-deep dot
// product halves the accumulator updates per flop; tools/mfma_power.py, profiles/r02g_mfma_power.jsonl).
// A ring stage is 64 k deep so one row-major operand row of a stage is exactly one 128-byte L2 line and
// one LDS-DMA wave-instruction covers 8 rows x 128 B = 8 whole lines (k-major stages are 512-byte k-rows: whole
// lines too).  32-deep stages request every line in two 64-byte halves, one k-step apart: rocprofv3 shows 2.0x
// the TCP_TCC_READ_REQ of hipBLASLt's MT256x256x64 kernel for the same bytes (profiles/r01_gemm_pmc.txt).
// LDS: 5 half-slots of 32 KiB = all 160 KiB.  Operand-stage A_j lives in slot (2j) % 5, B_j in (2j+1) % 5.
//   row-major operand stage: row r is 128 B at r*128, logical 16-byte chunk c at slot c ^ ((r>>1)&7);
//   k-major operand stage  : [64 k][256 cols], 512-byte k-rows, chunk slot' = slot ^ (((k&3)<<2) | (((k>>3)&1)<<1))
// (applied on the LDS-DMA source address, undone on the fragment read; both fragment reads are conflict-free: a 16-row
// x 4-chunk ds_read_b128, a 2 x (4 k-rows x 16 columns) ds_read_b64_tr_b16 per 32 lanes).
// Schedule of stage s (2 k-steps of 32; fragments double-buffered one k-step ahead, 16 + 16 registers x 4):
//     k-step 0 : 64 MFMA | 16 fragment reads (second half of stage s) | A_{s+2}[0:8]  (into the slot B_{s-1} vacated)
//     hand-off : vmcnt(8) retires this wave's B_{s+1} (A_{s+1} is older), lgkmcnt(0), barrier
//     k-step 1 : 64 MFMA | 16 fragment reads of stage s+1 | B_{s+2}[0:8]  (into the slot A_s vacated)
// The reads sit behind MFMA pairs 0..15 of a k-step (every read has at least 32 MFMAs to land), the pieces behind the odd
// pairs 17..31; 8-16 pieces per wave (32-64 KiB per CU) are in flight across every barrier, one barrier per 64 k.
// Rows past M / N are clamped to the last valid row (their products land in rows / columns the epilogue never
// stores); requires K % 64 == 0 (host dispatch).
constexpr int kFlThreads = 256;
constexpr int kXK = 64;
// compile-time loop: f(IntC<B>{}), ..., f(IntC<E-1>{}) -- every iteration its own instantiation, `if constexpr` on the index
// (a 64-slot schedule table is more than `#pragma unroll` folds: the unroller gave up and indexed the accumulators at run time)
// acc += A . B with the accumulator named as an AGPR tuple in the instruction itself (`+a`: ONE register tuple, read and
// written in place).  The three-barrier loop issues its MFMAs this way: through the builtin the register allocator rotated the
// accumulators of that loop through VGPRs (s_nop 6 + v_accvgpr_read / _write groups behind every other MFMA pair, a 450-move
// permutation on the back edge).  The compiler does not know this is an MFMA: the caller keeps the result away from other
// instructions for the pipeline's depth (the K loop only feeds accumulators back into MFMAs; `mfma_drain` before the way out).
template <typename T>
__device__ __forceinline__ void mfma16_inplace(f32x4& c, const u32x4& a, const u32x4& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (sizeof(T) == 2 && __is_same(T, bf16_t))
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
#else
  c = mfma16<T>(a, b, c);
#endif
}
// counted LDS wait (the fragment reads of the three-barrier loop are issued behind the compiler's back; LDS operations return in
// order, so "at most N outstanding" = "everything but the N youngest reads has arrived")
template <int N>
__device__ __forceinline__ void wait_lgkmcnt_le() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
#endif
}
__device__ __forceinline__ void mfma_drain() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#endif
}
template <int N>
struct IntC {
  static constexpr int value = N;
};
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(IntC<B>{});
    static_for<B + 1, E>(f);
  }
}
constexpr unsigned kXHalf = 256u * kXK * 2u;  // one operand of one stage: 32 KiB
constexpr int kXSlots = 5;
constexpr int kXSmem = kXSlots * (int)kXHalf;  // 163840 = the whole LDS of a CU

// byte offset (without the k-step / block immediates) of a lane's 16x16x32 fragment inside a k-major operand stage:
// two ds_read_b64_tr_b16 (k rows +0..3 and +4..7 of the lane group's 8) of the 16 columns starting at c16
__device__ __forceinline__ unsigned fl_frag_off_km(int c16, int lane) {
  const int g4 = lane >> 4, kq = (lane & 15) >> 2;
  const int col = c16 + 4 * (lane & 3);
  const int slot = (col >> 3) ^ ((kq << 2) | ((g4 & 1) << 1));  // (k&3) == kq, ((k>>3)&1) == g4&1 for every k-step
  return (unsigned)(8 * g4 + kq) * 512u + (unsigned)slot * 16u + (unsigned)(col & 7) * 2u;
}

// DBG (diagnostic instantiations, built only with -DTAMD_DIAG into libtamd_diag.so; TAMD_GEMM_DBG=n): wrong results by
// design -- 1 no LDS-DMA after the prologue, 2 no LDS fragment reads, 4 no vmcnt wait at the hand-off, 8 no barrier; correct,
// bit-identical results -- 32 the early piece placement (below) in every layout, 128 the late placement (pieces behind
// the odd MFMA pairs 17..31: the schedule of round 2) in every layout.
// Piece placement (EARLY): the 8 LDS-DMA pieces of a k-step go out behind its MFMA pairs 2, 5, .. 23, the 16 fragment
// reads of the next k-step on the pairs between them -- 23 MFMAs (~400 cycles) more on average for a piece to land
// before the hand-off waits for it than behind the odd pairs 17..31 (hipBLASLt's gfx950 kernel gives its operands
// 84-182 MFMAs).  Measured on MI355X (profiles/r03a_gemm_stagger_ab.jsonl, r03b_gemm_persist_ab.jsonl): +1.0 ... +2.6 %
// on the five forward shapes of Llama-3-8B and +0.7 ... +2.9 % on their dX products (lm_head dX, 2004 stages: +10 %), but
// -3 ... -4 % on the long dW products -- so it is the product schedule whenever A is row-major (forward and dX), and the
// late placement stays for dW (both operands k-major).  The same A/B buried two other differences to
// hipBLASLt's loop: a staggered, wrapping K start per workgroup (13 configurations: -1 ... +2 %, no pattern) and a
// second barrier per k-step (+-0.5 %).
// (Round 3 also measured a persistent walk -- one workgroup per CU, the XCD's workgroups starting every dispatch round
// together through an arrival counter, optionally re-aligned every 64 stages inside a tile -- against the drift of the
// long-K products: it took the L2 hit rate of the gate|up dX / dW from 74 / 62 % to the 81 % of the 8 x 4 patch and their
// fabric traffic to the patch floor (11.2 GB), and bought nothing: 1492 vs 1498 TFLOP/s with aligned rounds, -22 % with the
// hand-shakes.  The L2 misses of the long-K products are not what bounds them.  profiles/r03b_gemm_persist_ab.jsonl,
// r03b_gemm_persist_pmc.txt; the code: profiles/r03b_gemm_persist.patch.)
// (the kernel body: `vbid` = the workgroup's index inside ITS product -- blockIdx.x for gemm_fl_kernel, the index behind the
// product's first workgroup for gemm_fl_group_kernel)
template <typename T, bool A_KM, bool B_KN, int EPI, int ACT, int DBG>
// `pstride` (PERSIST instantiations, DBG bit 32768): the workgroup walks the tiles vbid, vbid + pstride, ... and requests stage 0 of
// its NEXT tile before it starts the way out of the current one, so that the first-stage latency of a tile (and the dispatch of a
// fresh workgroup) hides under the way out: the two-point fit of profiles/r06d_gemm_vs_hipblaslt_pmc.md prices prologue + way out
// at ~4 stage-times per tile -- 6 % of a 64-stage tile.  The way out then stages above the first LDS buffer (offset 64 KiB).
// Measured (profiles/r06m_gemm_piece_ab.jsonl; bit-identical; CPU model: three tiles per workgroup under adversarial LDS-DMA
// timing): forward q|k|v +0.9 %, o_proj -1.4 %, gate|up +0.2 %, down -0.2 %; dW o_proj +1.2 %, gate|up +0.5 % -- level, like the
// persistent walks of rounds 2 and 3.  Under the board's power cap (profiles/r06j_gemm_power.jsonl) an idle gap between tiles is
// not lost time: the cycles it frees come back as clock.  Diagnostic library only (tamd_gemm_set_dbg(32768)).
__device__ __forceinline__ void gemm_fl_body(const GemmArgs& g, const int vbid, const int pstride = 0) {
  TAMD_DYN_SMEM(smem);
  const int lane = threadIdx.x & 63;
  const int wave = wave_id_uniform();
  const int wm = wave >> 1, wn = wave & 1;
  const int g4 = lane >> 4, l15 = lane & 15;
  TAMD_TIMELINE_BEGIN
  TAMD_CLOCK_BEGIN
  constexpr bool PERSIST = (DBG & 32768) != 0;
  static_assert(!PERSIST || (EPI != kEpiSplitK && EPI != kEpiSwiGLU && A_KM == B_KN), "persistent walk: three-barrier layouts, plain ways out");
  int tile_m, tile_n;
  const int split = (EPI == kEpiSplitK) ? (int)((unsigned)vbid % (unsigned)g.splits) : 0;
  gemm_tile_of_block(g, (EPI == kEpiSplitK) ? (int)((unsigned)vbid / (unsigned)g.splits) : vbid, &tile_m, &tile_n);
  int64_t m0 = (int64_t)tile_m * kBM, n0 = (int64_t)tile_n * kBN;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* B = reinterpret_cast<const T*>(g.B);

  f32x4 acc[8][8];  // [nb][mb]
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) acc[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nst_all = (int)(g.K / kXK);
  const int st0 = (EPI == kEpiSplitK) ? split * g.stages_per_split : 0;  // first stage of this workgroup's K range
  const int nst = (EPI == kEpiSplitK) ? (nst_all - st0 < g.stages_per_split ? nst_all - st0 : g.stages_per_split) : nst_all;
  // per-lane source offsets: this wave's 8 A pieces ([0..7]) and 8 B pieces ([8..15]) of an operand stage.
  //   row-major operand: piece i = rows (wave*8+i)*8 .. +7, lane -> (row = lane>>3, physical chunk = lane&7);
  //                      next stage = +128 B
  //   k-major operand  : piece i = k-rows (wave*8+i)*2 .. +1, lane -> (k = lane>>5, physical chunk = lane&31);
  //                      next stage = +64 rows
  // Rows / column chunks outside the matrix are clamped to the last valid one.
  // Feed: buffer_load ... lds (tamd_device.h glds16_buf).  Wave-uniform operand base stepped per stage, loop-invariant
  // 32-bit lane offsets, one M0 per 4 pieces through the shared immediate.
  unsigned voff[16];  // byte offset from the operand base + 4096 - 1024*(piece & 3)
  int64_t kinc_a, kinc_b;  // bytes per stage; 0 once parked
  const char *base_a, *base_b;
  auto tile_sources = [&]() __attribute__((always_inline)) {  // bases and lane offsets of the tile at (m0, n0)
  kinc_a = A_KM ? (int64_t)kXK * g.lda * 2 : kXK * 2;
  kinc_b = B_KN ? (int64_t)kXK * g.ldb * 2 : kXK * 2;
  // operand bases (tile origin - 4096 B so that no lane offset goes negative after the immediate is taken out)
  base_a = (const char*)(A_KM ? A + m0 : A + m0 * g.lda) - 4096 + (int64_t)st0 * kinc_a;
  const int64_t nb0 = (EPI == kEpiSwiGLU) ? 0 : n0;  // SwiGLU: per-lane offsets address the whole fused weight
  base_b = (const char*)(B_KN ? B + n0 : B + nb0 * g.ldb) - 4096 + (int64_t)st0 * kinc_b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (wave * 8 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    const int kr = (wave * 8 + i) * 2 + (lane >> 5);
    const int col = ((lane & 31) ^ (((kr & 3) << 2) | (((kr >> 3) & 1) << 1))) * 8;
    const int64_t gca = (m0 + col < g.M) ? m0 + col : g.M - 8;
    const int64_t ra = (m0 + row < g.M) ? m0 + row : g.M - 1;
    const int64_t gcb = (n0 + col < g.N) ? n0 + col : g.N - 8;
    int64_t rb = (n0 + row < g.N) ? n0 + row : g.N - 1;
    if (EPI == kEpiSwiGLU) {  // tile row -> row of the fused [gate ; up] weight (see gemm_epilogue_swiglu)
      int64_t feat = (n0 >> 1) + ((row >> 6) << 5) + (row & 31);
      if (feat >= g.n_half) feat = g.n_half - 1;
      rb = ((row >> 5) & 1) * g.n_half + feat;
    }
    const int64_t oa = A_KM ? (int64_t)kr * g.lda + (gca - m0) : (ra - m0) * g.lda + c * 8;
    const int64_t ob = B_KN ? (int64_t)kr * g.ldb + (gcb - n0) : (rb - nb0) * g.ldb + c * 8;
    voff[i] = (unsigned)(oa * 2 + 4096 - (i & 3) * 1024);
    voff[8 + i] = (unsigned)(ob * 2 + 4096 - (i & 3) * 1024);
  }
  };
  tile_sources();
  // past the last stage: keep the load counts uniform and re-read the last valid stage (idempotent)
  auto park = [&]() __attribute__((always_inline)) {
    base_a -= kinc_a;
    base_b -= kinc_b;
    kinc_a = 0;
    kinc_b = 0;
  };
  const unsigned piece0 = (unsigned)wave * 8192u;  // this wave's first piece inside an operand stage
  bool dma_on = true;
  auto issue = [&](int p, int slot) __attribute__((always_inline)) {  // piece p (0..15) of this wave into half-slot `slot`
    if ((DBG & 1) && !dma_on) return;
    const unsigned dst = (unsigned)slot * kXHalf + piece0 + (unsigned)((p & 7) >> 2) * 4096u;  // + immediate
    const char* base = (p < 8) ? base_a : base_b;
    switch (p & 3) {
      case 0: glds16_buf<0>(base, voff[p], smem, dst); break;
      case 1: glds16_buf<1024>(base, voff[p], smem, dst); break;
      case 2: glds16_buf<2048>(base, voff[p], smem, dst); break;
      default: glds16_buf<3072>(base, voff[p], smem, dst); break;
    }
    if (p == 7) base_a += kinc_a;  // every piece of the operand stage is out: step to the next stage
    if (p == 15) base_b += kinc_b;
  };
  // fragment offsets inside a half-slot (block t = 16 rows / columns of this wave's 128, k-step q = 32 of the stage's 64)
  //   row-major: one register per k-step, the block is the immediate t*2048;
  //   k-major  : one register per block, the k-step is the immediate q*16384
  constexpr int NOX = A_KM ? 8 : 2, NOW = B_KN ? 8 : 2;
  unsigned offx[NOX], offw[NOW];
  {
    const unsigned sw = (unsigned)(l15 >> 1) & 7u;
#pragma unroll
    for (int j = 0; j < NOX; ++j)
      offx[j] = A_KM ? fl_frag_off_km(wm * 128 + j * 16, lane)
                     : (unsigned)(wm * 128 + l15) * 128u + (((unsigned)(j * 4 + g4) ^ sw) * 16u);
#pragma unroll
    for (int j = 0; j < NOW; ++j)
      offw[j] = B_KN ? fl_frag_off_km(wn * 128 + j * 16, lane)
                     : (unsigned)(wn * 128 + l15) * 128u + (((unsigned)(j * 4 + g4) ^ sw) * 16u);
  }
  // k-major fragments: two transposing 8-byte reads (k rows +0..3 and +4..7), issued untracked (tamd_device.h:
  // the compiler would drain vmcnt in front of each); every k-step therefore opens with an explicit lgkmcnt(0)
  auto frag_km = [&](unsigned off, int imm) __attribute__((always_inline)) -> u32x4 {
    const u32x2 lo = lds_read8_tr16_untracked(smem, off, imm);
    const u32x2 h2 = lds_read8_tr16_untracked(smem, off, imm + 4 * 512);
    return u32x4{lo[0], lo[1], h2[0], h2[1]};
  };
  auto frag_a = [&](int slot, int q, int t) __attribute__((always_inline)) -> u32x4 {
    if (A_KM) return frag_km((unsigned)slot * kXHalf + offx[A_KM ? t : 0], q * 16384);
    return lds_read16(smem, (unsigned)slot * kXHalf + offx[A_KM ? 0 : q] + (unsigned)t * 2048u);
  };
  auto frag_b = [&](int slot, int q, int t) __attribute__((always_inline)) -> u32x4 {
    if (B_KN) return frag_km((unsigned)slot * kXHalf + offw[B_KN ? t : 0], q * 16384);
    return lds_read16(smem, (unsigned)slot * kXHalf + offw[B_KN ? 0 : q] + (unsigned)t * 2048u);
  };
  auto kstep_open = [&]() __attribute__((always_inline)) {  // fragments of this k-step (read during the previous one) are in registers
    if (A_KM || B_KN) wait_lgkmcnt0();
    sched_fence();
  };
  u32x4 fx[2][8], fw[2][8];  // [buffer][mb / nb]
  // fragment read number r (0..15) of k-step q of stage slots (sa, sb) into buffer buf, in the order the MFMAs of the
  // next k-step want them: w0 x0 x1 .. x7 w1 .. w7
  auto rd1 = [&](int sa, int sb, int q, int buf, int r) __attribute__((always_inline)) {
    if ((DBG & 2) && !dma_on) return;
    if (r == 0)
      fw[buf][0] = frag_b(sb, q, 0);
    else if (r <= 8)
      fx[buf][r - 1] = frag_a(sa, q, r - 1);
    else
      fw[buf][r - 8] = frag_b(sb, q, r - 8);
  };
  // one k-step: 64 MFMAs from fragment buffer `buf` issued as 32 pairs, acc[nb][mb] with mb innermost (the W fragment
  // stays on the MFMA's A port for 8 instructions); in the gaps between the pairs the 16 fragment reads of the NEXT
  // k-step (stage slots ra / rb, k-step rq, into the other buffer) and the LDS-DMA pieces pb .. pb+7 into half-slot
  // ps -- one small group of feed instructions per gap, pinned with sched_barrier(0) so the matrix pipe (one wave per
  // SIMD: nobody else fills it) never waits behind a clump of LDS / LDS-DMA issues.  EARLY: of the pairs 0..23 every
  // third carries a piece, the other two a fragment read (the last read 9 pairs ahead of the hand-off's lgkmcnt(0));
  // otherwise reads behind the pairs 0..15, pieces behind the odd pairs 17..31.
  constexpr bool EARLY = (!A_KM || (DBG & 32)) && !(DBG & 128);
  // diagnostic placements (round 5, after the PMC side-by-side with hipBLASLt's kernel showed our waves parked at the hand-off
  // twice as long as theirs, profiles/r05a_gemm_vs_hipblaslt_pmc.md): 1 = pieces behind the even pairs 0..14, reads behind the odd
  // pairs 1..15 and 16..23; 2 = all 8 pieces behind the pairs 0..7, reads behind 8..23; 3 = as 2, and the hand-off is split --
  // only the write-after-read half (lgkmcnt(0) + barrier) stays at the k-step boundary, the wait for the landed stage
  // (vmcnt(16) + barrier) moves behind the 8 pieces of k-step 1, in front of the first read of the new stage: the operand
  // with the short lead (B) gets a whole stage (128-144 MFMAs instead of 82-124)
  // Measured (profiles/r05b_gemm_piece_ab.jsonl, r05c_gemm_piece_ab.jsonl; interleaved, two MI355X boxes): placement 1 within
  // +-0.2 % of the product schedule on the five forward shapes over five rounds (the first box's +0...2.6 % over three rounds
  // was noise), 2 -1 ... -2.5 %, 3 -2 ... -4 %: back-to-back LDS-DMA issues and a barrier inside the MFMA stream cost more
  // than the longer lead buys.  The product schedule stays EARLY; the three stay selectable in the diagnostic library.
  // 4 (round 6, DBG 1024) = the loop STRUCTURE of hipBLASLt's MT256x256x64_MI16x16x1 kernel, read off its disassembly
  // (profiles/r06_hipblaslt_loop.md): LDS double-buffered by whole stages (A_s, B_s in half-slots 2(s&1), 2(s&1)+1), three
  // barriers per stage, each in the middle of an MFMA run behind a wait that is long satisfied --
  //   k-step 0: the 8 A fragments of k-step 1 behind MFMAs 1,3..15 | lgkmcnt(0) 21, BARRIER 22: A_s is read by everybody |
  //             A_{s+2} pieces 0..4 behind 23,26..35 with the 8 B fragments of k-step 1 behind 25,28,31,34,37,39,41,43 |
  //             lgkmcnt(0) 51, BARRIER 52: B_s is read | A_{s+2} pieces 5..7 behind 53,56,59 | B_{s+2} piece 0 behind 62
  //   k-step 1: B_{s+2} pieces 1..4 behind 1,22,24,26 | vmcnt(13) 28, BARRIER 29: stage s+1 has landed for everybody |
  //             the 16 fragments of k-step 0 of stage s+1 behind 30..60, B_{s+2} pieces 5..7 behind 33,37,61 | lgkmcnt(0) 63
  // Same MFMA order, same summation order: bit-identical to the product schedule.
  // Measured on MI355X against the one-barrier schedule, interleaved, bit-identical (profiles/r06b_gemm_piece_ab.jsonl): forward
  // q|k|v +2.9 %, o_proj +2.3 %, gate|up +1.8 %, down -0.1 %; dW o_proj +1.9 %, gate|up +2.8 %; dX -1.2 ... +1.0 %.  It is the
  // product schedule of the forward layout (both operands row-major) and of dW (both k-major) since round 6; dX (row-major A,
  // k-major B) keeps the one-barrier ring.  DBG 1024 forces it in every layout, DBG 2048 forces the one-barrier ring.
  constexpr int PLACE = (DBG & 64) ? 1 : ((DBG & 256) ? 2 : ((DBG & 512) ? 3 : (((DBG & 1024) || (A_KM == B_KN && !(DBG & (2048 | 32 | 128 | 15)))) ? 4 : 0)));
  auto kstep = [&](int buf, int ra, int rb, int rq, int pb, int ps) __attribute__((always_inline)) {
    kstep_open();
#pragma unroll
    for (int p = 0; p < 32; ++p) {
      const int nb = p >> 2, mb = (p & 3) * 2;
      acc[nb][mb] = mfma16<T>(fw[buf][nb], fx[buf][mb], acc[nb][mb]);
      acc[nb][mb + 1] = mfma16<T>(fw[buf][nb], fx[buf][mb + 1], acc[nb][mb + 1]);
      sched_fence();
      if (PLACE == 1) {
        if (p < 16 && (p & 1) == 0) issue(pb + (p >> 1), ps);
        if (p < 16 && (p & 1) == 1) rd1(ra, rb, rq, buf ^ 1, p >> 1);
        if (p >= 16 && p < 24) rd1(ra, rb, rq, buf ^ 1, p - 8);
      } else if (PLACE >= 2) {
        if (p < 8) issue(pb + p, ps);
        if (PLACE == 3 && buf == 1 && p == 7) {  // the landed stage s+1: everybody's pieces (all but the 16 newest loads)
          wait_vmcnt<16>();
          raw_barrier();
        }
        if (p >= 8 && p < 24) rd1(ra, rb, rq, buf ^ 1, p - 8);
      } else if (EARLY) {
        if (p < 24 && p % 3 != 2) rd1(ra, rb, rq, buf ^ 1, p - p / 3);
        if (p < 24 && p % 3 == 2) issue(pb + p / 3, ps);
      } else {
        if (p < 16) rd1(ra, rb, rq, buf ^ 1, p);
        if (p >= 16 && (p & 1)) issue(pb + ((p - 17) >> 1), ps);
      }
      sched_fence();
    }
  };
  // ---- PLACE 4 (their instruction positions rounded to our MFMA pairs: a slot = behind MFMA 2p + 2)
  // Fragment reads behind the compiler's back in every layout (`=v` pins the fragments to the VGPR half: with tracked reads the
  // allocator put fragments into AGPRs and accumulators into VGPRs -- a copy in front of every MFMA, 56 spilled registers)
  auto frag_a4 = [&](int slot, int q, int t) __attribute__((always_inline)) -> u32x4 {
    if (A_KM) return frag_km((unsigned)slot * kXHalf + offx[A_KM ? t : 0], q * 16384);
    return lds_read16_untracked(smem, (unsigned)slot * kXHalf + offx[A_KM ? 0 : q], t * 2048);
  };
  auto frag_b4 = [&](int slot, int q, int t) __attribute__((always_inline)) -> u32x4 {
    if (B_KN) return frag_km((unsigned)slot * kXHalf + offw[B_KN ? t : 0], q * 16384);
    return lds_read16_untracked(smem, (unsigned)slot * kXHalf + offw[B_KN ? 0 : q], t * 2048);
  };
  // k-step 0 of the stage in LDS buffer b (half-slots 2b, 2b+1): fragments of its k-step 1 into register buffer 1, the pieces of
  // stage s+2 into the same LDS buffer as the two barriers release its halves
  auto kstep3_0 = [&](int b) __attribute__((always_inline)) {
    const int sa = 2 * b, sb = 2 * b + 1;
    wait_lgkmcnt0();  // (the fragments of this k-step: read behind the compiler's back)
    sched_fence();
#pragma unroll
    for (int p = 0; p < 32; ++p) {
      const int nb = p >> 2, mb = (p & 3) * 2;
      mfma16_inplace<T>(acc[nb][mb], fw[0][nb], fx[0][mb]);
      mfma16_inplace<T>(acc[nb][mb + 1], fw[0][nb], fx[0][mb + 1]);
      sched_fence();
      if (p < 8) fx[1][p] = frag_a4(sa, 1, p);
      if (p == 10) {  // A_s is read by everybody
        wait_lgkmcnt0();
        raw_barrier();
      }
      if (p == 11) issue(0, sa);
      if (p == 12) fw[1][0] = frag_b4(sb, 1, 0);
      if (p == 12) issue(1, sa);
      if (p == 13) fw[1][1] = frag_b4(sb, 1, 1);
      if (p == 14) issue(2, sa);
      if (p == 15) fw[1][2] = frag_b4(sb, 1, 2);
      if (p == 15) issue(3, sa);
      if (p == 16) fw[1][3] = frag_b4(sb, 1, 3);
      if (p == 17) issue(4, sa);
      if (p >= 18 && p <= 21) fw[1][p - 14] = frag_b4(sb, 1, p - 14);
      if (p == 25) {  // B_s is read by everybody
        wait_lgkmcnt0();
        raw_barrier();
      }
      if (p == 26) issue(5, sa);
      if (p == 27) issue(6, sa);
      if (p == 29) issue(7, sa);
      if (p == 30) issue(8, sb);
      sched_fence();
    }
  };
  // k-step 1: the rest of B_{s+2}; once stage s+1 has landed (LDS buffer b ^ 1) the fragments of its k-step 0 into register buffer 0
  auto kstep3_1 = [&](int b) __attribute__((always_inline)) {
    const int sb = 2 * b + 1, na = 2 * (b ^ 1), nbs = 2 * (b ^ 1) + 1;
    wait_lgkmcnt0();  // (the fragments of this k-step: read behind the compiler's back)
    sched_fence();
#pragma unroll
    for (int p = 0; p < 32; ++p) {
      const int nb = p >> 2, mb = (p & 3) * 2;
      mfma16_inplace<T>(acc[nb][mb], fw[1][nb], fx[1][mb]);
      mfma16_inplace<T>(acc[nb][mb + 1], fw[1][nb], fx[1][mb + 1]);
      sched_fence();
      if (p == 0) issue(9, sb);
      if (p >= 10 && p <= 12) issue(p, sb);
      if (p == 13) {  // everything but this stage's 8 + 5 pieces: stage s+1 is in LDS, for everybody
        wait_vmcnt<13>();
        raw_barrier();
      }
      if (p == 14) fx[0][0] = frag_a4(na, 0, 0);
      if (p == 15) fx[0][1] = frag_a4(na, 0, 1);
      if (p == 15) fx[0][2] = frag_a4(na, 0, 2);
      if (p == 16) fx[0][3] = frag_a4(na, 0, 3);
      if (p == 16) issue(13, sb);
      if (p == 17) fx[0][4] = frag_a4(na, 0, 4);
      if (p == 18) issue(14, sb);
      if (p == 19) fx[0][5] = frag_a4(na, 0, 5);
      if (p == 19) fx[0][6] = frag_a4(na, 0, 6);
      if (p == 20) fx[0][7] = frag_a4(na, 0, 7);
      if (p == 20) fw[0][0] = frag_b4(nbs, 0, 0);
      if (p == 21) fw[0][1] = frag_b4(nbs, 0, 1);
      if (p == 22) fw[0][2] = frag_b4(nbs, 0, 2);
      if (p == 24) fw[0][3] = frag_b4(nbs, 0, 3);
      if (p == 25) fw[0][4] = frag_b4(nbs, 0, 4);
      if (p == 26) fw[0][5] = frag_b4(nbs, 0, 5);
      if (p == 28) fw[0][6] = frag_b4(nbs, 0, 6);
      if (p == 29) fw[0][7] = frag_b4(nbs, 0, 7);
      if (p == 30) issue(15, sb);
      sched_fence();
    }
  };
  // ---- the placements of this structure that were measured (profiles/r06b_, r06g_gemm_piece_ab.jsonl; interleaved, bit-identical):
  //   FINE   : the vendor table at its own granularity -- ONE MFMA per gap, at most one feed instruction behind it.  THE PRODUCT
  //            PLACEMENT: against the pair-rounded table forward +0.9 / +0.5 / +0.3 / 0.0 % (q|k|v, o_proj, gate|up, down), dW
  //            o_proj +1.9 %, gate|up +2.2 %
  //   pairs  : the same table rounded to our MFMA pairs (kstep3_*; DBG 4096): what first showed the structure pays
  //   SPREAD : pairs, at most ONE memory instruction per gap, B_{s+2} in the first 13 gaps of k-step 1 (every other one) so that
  //            the landed-data wait is vmcnt(16) and the 16 fragment reads of stage s+1 have gaps 14..29 to themselves (DBG 8192):
  //            -0.5 ... -4.4 % -- an emptier gap is not what the loop was missing
  constexpr bool SPREAD = (DBG & 8192) != 0, FINE = !(DBG & 4096) && !SPREAD;
  // COUNTED (DBG 16384): the stage opens with lgkmcnt(4) instead of lgkmcnt(0) -- the four youngest reads (B fragments 4..7 of this
  // k-step, requested behind MFMAs 51..60 of the previous stage) are first used by MFMA 33 and are covered by the lgkmcnt(0)
  // in front of barrier (1), behind MFMA 21; B fragment 3 was requested 15 MFMAs ago.  (k-major fragments are two reads each.)
  constexpr bool COUNTED = (DBG & 16384) != 0;
  auto kfine_0 = [&](int b) __attribute__((always_inline)) {
    const int sa = 2 * b, sb = 2 * b + 1;
    if (COUNTED)
      wait_lgkmcnt_le<B_KN ? 8 : 4>();
    else
      wait_lgkmcnt0();
    sched_fence();
    static_for<1, 65>([&](auto I) __attribute__((always_inline)) {
      constexpr int i = decltype(I)::value;
      mfma16_inplace<T>(acc[(i - 1) >> 3][(i - 1) & 7], fw[0][(i - 1) >> 3], fx[0][(i - 1) & 7]);
      sched_fence();
#ifdef TAMD_B3_TABLE  // (tools/b3_search.py: a generated table, force-included; the product has the vendor table below)
      TAMD_B3_K0_ACTIONS
#else
      if constexpr (i <= 15 && (i & 1)) fx[1][(i - 1) >> 1] = frag_a4(sa, 1, (i - 1) >> 1);
      if constexpr (i == 21) wait_lgkmcnt0();
      if constexpr (i == 22) raw_barrier();
      if constexpr (i >= 23 && i <= 35 && (i - 23) % 3 == 0) issue((i - 23) / 3, sa);
      if constexpr (i >= 25 && i <= 37 && (i - 25) % 3 == 0) fw[1][(i - 25) / 3] = frag_b4(sb, 1, (i - 25) / 3);
      if constexpr (i == 39 || i == 41 || i == 43) fw[1][5 + (i - 39) / 2] = frag_b4(sb, 1, 5 + (i - 39) / 2);
      if constexpr (i == 51) wait_lgkmcnt0();
      if constexpr (i == 52) raw_barrier();
      if constexpr (i == 53 || i == 56 || i == 59) issue(5 + (i - 53) / 3, sa);
      if constexpr (i == 62) issue(8, sb);
#endif
      sched_fence();
    });
  };
  auto kfine_1 = [&](int b) __attribute__((always_inline)) {
    const int sb = 2 * b + 1, na = 2 * (b ^ 1), nbs = 2 * (b ^ 1) + 1;
    wait_lgkmcnt0();
    sched_fence();
    static_for<1, 65>([&](auto I) __attribute__((always_inline)) {
      constexpr int i = decltype(I)::value;
      mfma16_inplace<T>(acc[(i - 1) >> 3][(i - 1) & 7], fw[1][(i - 1) >> 3], fx[1][(i - 1) & 7]);
      sched_fence();
#ifdef TAMD_B3_TABLE
      TAMD_B3_K1_ACTIONS
#else
      if constexpr (i == 1) issue(9, sb);
      if constexpr (i == 22 || i == 24 || i == 26) issue(10 + (i - 22) / 2, sb);
      if constexpr (i == 28) wait_vmcnt<13>();
      if constexpr (i == 29) raw_barrier();
      if constexpr (i == 30 || i == 31 || i == 32) fx[0][i - 30] = frag_a4(na, 0, i - 30);
      if constexpr (i == 33) issue(13, sb);
      if constexpr (i == 34 || i == 35) fx[0][i - 31] = frag_a4(na, 0, i - 31);
      if constexpr (i == 37) issue(14, sb);
      if constexpr (i == 39 || i == 40 || i == 41) fx[0][i - 34] = frag_a4(na, 0, i - 34);
      if constexpr (i == 42 || i == 43) fw[0][i - 42] = frag_b4(nbs, 0, i - 42);
      if constexpr (i == 46) fw[0][2] = frag_b4(nbs, 0, 2);
      if constexpr (i == 49) fw[0][3] = frag_b4(nbs, 0, 3);
      if constexpr (i == 51) fw[0][4] = frag_b4(nbs, 0, 4);
      if constexpr (i == 54) fw[0][5] = frag_b4(nbs, 0, 5);
      if constexpr (i == 57) fw[0][6] = frag_b4(nbs, 0, 6);
      if constexpr (i == 60) fw[0][7] = frag_b4(nbs, 0, 7);
      if constexpr (i == 61) issue(15, sb);
#endif
      sched_fence();
    });
  };
  auto kspread_0 = [&](int b) __attribute__((always_inline)) {
    const int sa = 2 * b, sb = 2 * b + 1;
    wait_lgkmcnt0();
    sched_fence();
#pragma unroll
    for (int p = 0; p < 32; ++p) {
      const int nb = p >> 2, mb = (p & 3) * 2;
      mfma16_inplace<T>(acc[nb][mb], fw[0][nb], fx[0][mb]);
      mfma16_inplace<T>(acc[nb][mb + 1], fw[0][nb], fx[0][mb + 1]);
      sched_fence();
      if (p < 8) fx[1][p] = frag_a4(sa, 1, p);
      if (p == 10) {
        wait_lgkmcnt0();
        raw_barrier();
      }
      if (p >= 11 && p <= 19 && (p & 1)) issue((p - 11) >> 1, sa);              // 11 13 15 17 19: A pieces 0..4
      if (p >= 12 && p <= 20 && !(p & 1)) fw[1][(p - 12) >> 1] = frag_b4(sb, 1, (p - 12) >> 1);  // 12 .. 20: B fragments 0..4
      if (p >= 21 && p <= 23) fw[1][p - 16] = frag_b4(sb, 1, p - 16);             // 21 22 23: B fragments 5..7
      if (p == 26) {
        wait_lgkmcnt0();
        raw_barrier();
      }
      if (p >= 27 && p <= 29) issue(p - 22, sa);  // A pieces 5..7
      if (p == 30) issue(8, sb);
      sched_fence();
    }
  };
  auto kspread_1 = [&](int b) __attribute__((always_inline)) {
    const int sb = 2 * b + 1, na = 2 * (b ^ 1), nbs = 2 * (b ^ 1) + 1;
    wait_lgkmcnt0();
    sched_fence();
#pragma unroll
    for (int p = 0; p < 32; ++p) {
      const int nb = p >> 2, mb = (p & 3) * 2;
      mfma16_inplace<T>(acc[nb][mb], fw[1][nb], fx[1][mb]);
      mfma16_inplace<T>(acc[nb][mb + 1], fw[1][nb], fx[1][mb + 1]);
      sched_fence();
      if (p <= 12 && !(p & 1)) issue(9 + (p >> 1), sb);  // 0 2 .. 12: B pieces 1..7
      if (p == 13) {  // all of this stage's 16 pieces may be in flight: everything older (stage s+1) has landed
        wait_vmcnt<16>();
        raw_barrier();
      }
      if (p >= 14 && p <= 21) fx[0][p - 14] = frag_a4(na, 0, p - 14);
      if (p >= 22 && p <= 29) fw[0][p - 22] = frag_b4(nbs, 0, p - 22);
      sched_fence();
    }
  };
  int vb = vbid;     // PERSIST: the tile this workgroup is on
  bool pre = false;  // PERSIST: stage 0 of the current tile was requested during the previous tile's way out
  for (;;) {
  if (PERSIST) {
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) acc[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // prologue: A_0 B_0 A_1 B_1 into half-slots 0..3
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (j == nst) park();
    if (!(PERSIST && j == 0 && pre)) {
#pragma unroll
      for (int p = 0; p < 16; ++p) issue(p, 2 * j + (p >> 3));
    }
  }
  if (PLACE == 4) {
    wait_vmcnt<16>();  // stage 0 has landed; stage 1 stays in flight (the first vmcnt(13) + barrier covers it)
    raw_barrier();
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      fx[0][t] = frag_a4(0, 0, t);
      fw[0][t] = frag_b4(1, 0, t);
    }
    wait_lgkmcnt0();  // (the loop's first stage may open with a counted wait that assumes the loop's own read order)
    const int nst2 = nst & ~1;
    for (int s0 = 0; s0 < nst2; s0 += 2) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        sched_fence();
        if (s0 + b + 2 == nst) park();
        if (FINE) {
          kfine_0(b);
          kfine_1(b);
        } else if (SPREAD) {
          kspread_0(b);
          kspread_1(b);
        } else {
          kstep3_0(b);
          kstep3_1(b);
        }
      }
    }
    if (nst & 1) {  // the last stage of an odd count (its parity is even: LDS buffer 0); stage s+2 is parked since s = nst - 2
      sched_fence();
      if (FINE) {
        kfine_0(0);
        kfine_1(0);
      } else if (SPREAD) {
        kspread_0(0);
        kspread_1(0);
      } else {
        kstep3_0(0);
        kstep3_1(0);
      }
    }
  } else {
  wait_vmcnt<0>();
  raw_barrier();
#pragma unroll
  for (int r = 0; r < 16; ++r) rd1(0, 1, 0, 0, r);
  dma_on = false;
  for (int s0 = 0; s0 < nst; s0 += kXSlots) {
#pragma unroll
    for (int u = 0; u < kXSlots; ++u) {
      const int s = s0 + u;
      if (s < nst) {
        const int sa = (2 * u) % kXSlots, sb = (2 * u + 1) % kXSlots;        // A_s, B_s
        const int sa1 = (2 * u + 2) % kXSlots, sb1 = (2 * u + 3) % kXSlots;  // A_{s+1}, B_{s+1}
        const int sa2 = (2 * u + 4) % kXSlots;                               // A_{s+2} (= slot of B_{s-1})
        const int sb2 = sa;                                                  // B_{s+2} (= slot of A_s)
        sched_fence();
        if (s + 2 == nst) park();
        kstep(0, sa, sb, 1, 0, sa2);  // k-step 0 | second-half fragments of stage s | A_{s+2}
        // hand-off: stage s+1 has landed for everybody; everybody's reads of stage s are in registers
        if (!(DBG & 4) && PLACE != 3) wait_vmcnt<8>();  // own B_{s+1} (and the older A_{s+1}); the 8 newest (A_{s+2}) stay in flight
        wait_lgkmcnt0();
        if (!(DBG & 8)) raw_barrier();
        sched_fence();
        kstep(1, sa1, sb1, 0, 8, sb2);  // k-step 1 | first-half fragments of stage s+1 | B_{s+2} into the slot A_s vacated
      }
    }
  }
  }  // PLACE != 4
  if (PLACE == 4) mfma_drain();
  wait_vmcnt<0>();
  wait_lgkmcnt0();
  raw_barrier();
  TAMD_CLOCK_END
  // (the epilogue takes its lane index from v_mbcnt: kept from the kernel entry it is spilled around the K loop, and a
  // kernel with scratch is throttled in how many of its waves a CU runs)
  // (checked per instantiation with tools/gemm_isa.sh and tests/test_isa_lint.py: since the buffer-addressed way out of round 5 no
  // full-line instantiation has scratch -- the k-major residual / accumulate ones used to spill 5-17 registers)
  const int elane = lane_id_mbcnt();
  // PERSIST: the way out is for the tile at (em0, en0); the sources move on to the next tile first and its stage 0 is requested
  // into LDS buffer 0 (half-slots 0, 1), which every wave has finished reading (the barrier above); the way out stages above it
  const int64_t em0 = m0, en0 = n0;
  constexpr unsigned kStageBase = PERSIST ? 2u * kXHalf : 0u;
  if (PERSIST) {
    const int nxt = vb + pstride;
    pre = nxt < g.tiles_m * g.tiles_n;
    if (pre) {
      gemm_tile_of_block(g, nxt, &tile_m, &tile_n);
      m0 = (int64_t)tile_m * kBM;
      n0 = (int64_t)tile_n * kBN;
      tile_sources();
#pragma unroll
      for (int p = 0; p < 16; ++p) issue(p, p >> 3);
    }
    vb = nxt;
  }
  if (EPI == kEpiSplitK) {  // fp32 partial tile: lane = output row, 4 consecutive columns per accumulator block
    float* ws = g.ws + (int64_t)split * g.M * g.N;
    const int l15 = elane & 15, g4 = elane >> 4;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      const int64_t m = em0 + wm * 128 + mb * 16 + l15;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        const int64_t n = en0 + wn * 128 + nb * 16 + 4 * g4;
        if (m < g.M && n < g.N)
          st16(ws + m * g.N + n, u32x4{f32_as_u32(acc[nb][mb][0]), f32_as_u32(acc[nb][mb][1]), f32_as_u32(acc[nb][mb][2]),
                                       f32_as_u32(acc[nb][mb][3])});
      }
    }
  } else if (EPI == kEpiSwiGLU) {
    gemm_epilogue_swiglu<T>(g, acc, smem, (unsigned)wave * (64u * (128 * 2 + 16) + 64u * (64 * 2 + 16)), em0 + wm * 128,
                            (en0 >> 1) + wn * 64, elane);
  } else if (EPI == kEpiSwiGLUBwd) {
    gemm_epilogue_swiglu_bwd<T>(g, acc, smem, kStageBase + (unsigned)wave * (64u * (4 * 32 * 2 + 16)), em0 + wm * 128, en0 + wn * 128,
                                elane);
  } else {
    constexpr int E2 = (EPI == kEpiSplitK || EPI == kEpiSwiGLU || EPI == kEpiSwiGLUBwd) ? TAMD_EPI_NONE : EPI;
    if (A_KM && B_KN && (EPI == TAMD_EPI_NONE || EPI == TAMD_EPI_ACCUM)) {  // dW: the tile's segment (wave-uniform selects;
      GemmArgs gs = g;                                                        // ONE epilogue instance: a second one spills)
      gs.C = gemm_seg_base<T>(g, em0);
      gemm_epilogue16<T, E2, ACT>(gs, acc, smem, kStageBase + (unsigned)wave * (64u * (4 * 32 * 2 + 16)), em0 + wm * 128, en0 + wn * 128,
                                  elane);
    } else {
      gemm_epilogue16<T, E2, ACT>(g, acc, smem, kStageBase + (unsigned)wave * (64u * (4 * 32 * 2 + 16)), em0 + wm * 128, en0 + wn * 128,
                                  elane);
    }
  }
  if (!PERSIST || !pre) break;
  wait_lgkmcnt0();  // every wave is done with its staging area before stage 1 of the next tile lands in half-slots 2, 3
  raw_barrier();
  }  // for (;;)
  TAMD_TIMELINE_END
}
template <typename T, bool A_KM, bool B_KN, int EPI, int ACT, int DBG = 0>
__global__ __launch_bounds__(kFlThreads, 1) void gemm_fl_kernel(GemmArgs g) {
  gemm_fl_body<T, A_KM, B_KN, EPI, ACT, DBG>(g, (int)blockIdx.x);
}
// the persistent walk: one workgroup per CU (grid = the device's CU count, a multiple of 8 so that blockIdx & 7 stays the XCD)
template <typename T, bool A_KM, bool B_KN, int EPI, int ACT, int DBG = 0>
__global__ __launch_bounds__(kFlThreads, 1) void gemm_fl_persist_kernel(GemmArgs g) {
  gemm_fl_body<T, A_KM, B_KN, EPI, ACT, DBG | 32768>(g, (int)blockIdx.x, (int)gridDim.x);
}

// ============================================================================================ grouped launch
// Several independent products of ONE layout in a single launch (tamd_gemm_group): the four weight gradients of a BERT layer
// (dW = dY^T . X, outputs of 768 .. 3072 rows and columns over 16384 tokens) are 9 .. 36 tiles each -- launched one by one
// every one of them splits K 7 .. 16 ways to reach the 256 CUs, 16 .. 37 stages per workgroup behind a prologue and in front
// of a 256 KiB fp32 partial tile (256 MB of partials per layer, 335 us for 232 GFLOP: profiles/r04c_bert_kernel_stats.csv).
// Together they are 108 tiles: two splits fill the GPU with 128-stage workgroups and a quarter of the partial traffic.
constexpr int kGroupMax = 4;
struct GemmGroupArgs {
  GemmArgs p[kGroupMax];
  int start[kGroupMax];   // first workgroup of problem i (unused problems: INT_MAX).  Direct launches pad every problem to a
  int blocks[kGroupMax];  // multiple of 8 workgroups (blockIdx & 7 stays the XCD for gemm_tile_of_block); the rest exit
};
template <typename T, bool A_KM, bool B_KN, int EPI>
__global__ __launch_bounds__(kFlThreads, 1) void gemm_fl_group_kernel(GemmGroupArgs grp) {
  const int bid = (int)blockIdx.x;
  int i = 0;
#pragma unroll
  for (int j = 1; j < kGroupMax; ++j) i = (bid >= grp.start[j]) ? j : i;
  const int local = bid - grp.start[i];
  if (local >= grp.blocks[i]) return;
  gemm_fl_body<T, A_KM, B_KN, EPI, TAMD_ACT_NONE, 0>(grp.p[i], local);  // (wave-uniform index into the kernel arguments)
}

// ============================================================================================ small tile
// 128 x 128 tile for grids the 256 x 256 tile cannot spread over the GPU: a CLIP-L tower runs its 577 tokens through
// projections of 1024 .. 4096 features -- 12 .. 48 workgroups of 256 x 256 on 256 CUs, and the bias / activation
// epilogues rule split-K out (8 ms of a 29 ms LLaVA-1.5-7B forward for 0.9 TFLOP, profiles/r03d_llava_kernel_stats.csv).
// Forward layout only (A [M,K] and B [N,K] row-major), K % 64 == 0.  4 waves x (64 x 64) on v_mfma_f32_32x32x16 ("swapped",
// accumulator layout of gemm_epilogue), 64-deep stages of 16 KiB per operand, double-buffered: 64 KiB, so two workgroups
// share a CU and one's operand latency, hand-off and epilogue hide behind the other's MFMAs -- a kernel for problems
// whose K loops are too short to amortise a deep pipeline.  Per stage and wave: 8 LDS-DMA pieces (per-lane row
// pointers are loop invariants, rows past M / N clamped to the last valid one), 16 fragment reads, 16 MFMAs; one barrier
// per stage.  LDS image of an operand stage = gemm_fl_kernel's row-major one: row r at r*128, 16-byte chunk c at slot
// c ^ ((r>>1)&7) (swizzle on the LDS-DMA source address, undone on the fragment read, conflict-free both ways).
constexpr int kSmTile = 128;
constexpr int kSmThreads = 256;
constexpr unsigned kSmOperand = (unsigned)kSmTile * kXK * 2u;  // 16 KiB
constexpr unsigned kSmStage = 2u * kSmOperand;                 // A + B
constexpr int kSmSmem = 2 * (int)kSmStage;                     // 64 KiB (covers the epilogue staging: 4 x 64 x 144 B)
constexpr unsigned kSmStageWave = 64u * (64u * 2u + 16u);
// grids of at most this many 256 x 256 tiles take the small tile (host dispatch): ahead of the 256 x 256 kernel up to 128
// tiles (37 vs 42 us at 128, 21 vs 37 at 32), behind it from 192 on (57 vs 46 us) -- profiles/r03q_gemm_sm_ab.jsonl
constexpr int kSmMaxBigTiles = 128;

template <typename T, int EPI, int ACT>
__global__ __launch_bounds__(kSmThreads, 2) void gemm_sm_kernel(GemmArgs g) {
  TAMD_DYN_SMEM(smem);
  const int lane = threadIdx.x & 63;
  const int wave = wave_id_uniform();
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;
  int tile_m, tile_n;
  gemm_tile_of_block(g, (int)blockIdx.x, &tile_m, &tile_n);
  const int64_t m0 = (int64_t)tile_m * kSmTile, n0 = (int64_t)tile_n * kSmTile;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* B = reinterpret_cast<const T*>(g.B);

  f32x16 acc[2][2];  // [ni][mi]
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  // this wave's 4 + 4 pieces of an operand stage: piece i = rows (wave*4+i)*8 .. +7, lane -> (row = lane>>3, LDS slot =
  // lane&7 holding chunk slot ^ ((row>>1)&7)); the stage-to-stage step is +128 bytes
  const char* pa[4];
  const char* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    const int64_t ra = (m0 + r < g.M) ? m0 + r : g.M - 1, rb = (n0 + r < g.N) ? n0 + r : g.N - 1;
    pa[i] = reinterpret_cast<const char*>(A + ra * g.lda + c * 8);
    pb[i] = reinterpret_cast<const char*>(B + rb * g.ldb + c * 8);
  }
  auto issue = [&](int st, unsigned buf_off) {
    const int64_t kb = (int64_t)st * (kXK * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(pa[i] + kb, smem, buf_off + (unsigned)(wave * 4 + i) * 1024u);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(pb[i] + kb, smem, buf_off + kSmOperand + (unsigned)(wave * 4 + i) * 1024u);
  };
  // fragment offsets of this lane inside an operand stage (rows wm/wn * 64 + blk * 32 + l31; k-step ks adds chunk 2*ks)
  unsigned fa[2], fb[2];
  const unsigned swz = (unsigned)((l31 >> 1) & 7);  // ((row >> 1) & 7) of every row this lane reads (block bases are multiples of 32)
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    fa[blk] = (unsigned)(wm * 64 + blk * 32 + l31) * 128u;
    fb[blk] = kSmOperand + (unsigned)(wn * 64 + blk * 32 + l31) * 128u;
  }
  const int nst = (int)(g.K / kXK);
  issue(0, 0u);
  for (int st = 0; st < nst; ++st) {
    const unsigned cur = (unsigned)(st & 1) * kSmStage;
    wait_vmcnt0();   // this wave's pieces of stage st have landed ...
    raw_barrier();   // ... everyone's have, and everyone is done reading the other buffer (stage st-1)
    if (st + 1 < nst) issue(st + 1, cur ^ kSmStage);
    // fragments of k-step ks+1 are requested before the MFMAs of k-step ks issue (two register sets)
    u32x4 xa[2][2], wb[2][2];
    auto frags = [&](int ks, int set) {
      const unsigned ch = (((unsigned)(ks * 2 + hi)) ^ swz) * 16u;
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        xa[set][blk] = lds_read16(smem, cur + fa[blk] + ch);
        wb[set][blk] = lds_read16(smem, cur + fb[blk] + ch);
      }
    };
    frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks + 1 < 4) frags(ks + 1, (ks + 1) & 1);
      sched_fence();
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = mfma32<T>(wb[ks & 1][ni], xa[ks & 1][mi], acc[ni][mi]);
      sched_fence();
    }
  }
  block_sync();  // every wave is past its fragment reads: the staging regions overlay the operand buffers
  gemm_epilogue<T, EPI, ACT, 2, 2>(g, acc, smem, (unsigned)wave * kSmStageWave, m0 + wm * 64, n0 + wn * 64, lane);
}

// out[m][n] = epilogue(sum_s ws[s][m][n]): 4 columns per thread (16-byte reads, 8-byte stores).  The reduction applies the
// product's epilogue with the roundings of the unsplit kernel -- MODE = TAMD_EPI_NONE round(acc); TAMD_EPI_BIAS
// round(acc + bias); TAMD_EPI_RESIDUAL round(round(acc [+ bias]) + R); TAMD_EPI_ACCUM round(round(acc) + C_old) -- so a
// bias / residual product on a grid that cannot fill the GPU (CLIP fc2: 12 tiles over K = 4096; o_proj / down_proj of a
// short prompt) splits like a plain one (round 3 sent those to the 128 x 128 kernel, or copied the residual into C first and
// accumulated onto it: 41 us against hipBLASLt's 19 for CLIP fc2, profiles/r04b_gemm_tw_ab.jsonl).
// one 4-column vector (row m, columns n..n+3) of the reduction
template <typename T, int MODE>
__device__ __forceinline__ void splitk_reduce_vec(const float* __restrict__ ws, T* __restrict__ out, const T* __restrict__ bias,
                                                  const T* __restrict__ res, int64_t M, int64_t N, int64_t m, int64_t n,
                                                  int splits) {
  typedef typename elem<T>::raw raw;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  for (int sidx = 0; sidx < splits; ++sidx) {
    const u32x4 v = ld16(ws + ((int64_t)sidx * M + m) * N + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] += u32_as_f32(v[e]);
  }
  if ((MODE == TAMD_EPI_BIAS || MODE == TAMD_EPI_RESIDUAL) && bias != nullptr) {
    const u32x2 bq = ld8(bias + n);
    a[0] += elem<T>::to_f32((raw)(bq[0] & 0xffffu));
    a[1] += elem<T>::to_f32((raw)(bq[0] >> 16));
    a[2] += elem<T>::to_f32((raw)(bq[1] & 0xffffu));
    a[3] += elem<T>::to_f32((raw)(bq[1] >> 16));
  }
  if (MODE == TAMD_EPI_ACCUM || MODE == TAMD_EPI_RESIDUAL) {
    const u32x2 o = ld8(MODE == TAMD_EPI_ACCUM ? (const T*)out : res);
    a[0] = round_through<T>(a[0]) + elem<T>::to_f32((raw)(o[0] & 0xffffu));
    a[1] = round_through<T>(a[1]) + elem<T>::to_f32((raw)(o[0] >> 16));
    a[2] = round_through<T>(a[2]) + elem<T>::to_f32((raw)(o[1] & 0xffffu));
    a[3] = round_through<T>(a[3]) + elem<T>::to_f32((raw)(o[1] >> 16));
  }
  st8(out, u32x2{pack2<T>(a[0], a[1]), pack2<T>(a[2], a[3])});
}
template <typename T, int MODE>
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, T* __restrict__ C, const T* __restrict__ bias,
                                     const T* __restrict__ R, int64_t M, int64_t N, int64_t ldc, int64_t ldr, int splits,
                                     T* C_seg1 = nullptr, T* C_seg2 = nullptr, int64_t seg_row1 = 0, int64_t seg_row2 = 0) {
  const int64_t nvec = M * (N / 4);
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nvec; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = idx / (N / 4), n = (idx % (N / 4)) * 4;
    T* out = C + m * ldc + n;
    if (seg_row2 > 0 && m >= seg_row2)  // (segmented output: tamd_gemm_seg)
      out = C_seg2 + (m - seg_row2) * ldc + n;
    else if (seg_row1 > 0 && m >= seg_row1)
      out = C_seg1 + (m - seg_row1) * ldc + n;
    splitk_reduce_vec<T, MODE>(ws, out, bias, R + m * ldr + n, M, N, m, n, splits);
  }
}
// the reductions of a grouped launch (tamd_gemm_group) in one: plain or accumulating
struct ReduceGroupArgs {
  const float* ws[kGroupMax];
  void* C[kGroupMax];
  int64_t M[kGroupMax], N[kGroupMax], ldc[kGroupMax];
  int splits[kGroupMax];
  int start[kGroupMax];   // first block of problem i (unused: INT_MAX)
  int blocks[kGroupMax];
};
template <typename T, int MODE>
__global__ void splitk_reduce_group_kernel(ReduceGroupArgs r) {
  const int bid = (int)blockIdx.x;
  int i = 0;
#pragma unroll
  for (int j = 1; j < kGroupMax; ++j) i = (bid >= r.start[j]) ? j : i;
  const int64_t M = r.M[i], N = r.N[i], ldc = r.ldc[i];
  const float* ws = r.ws[i];
  T* C = reinterpret_cast<T*>(r.C[i]);
  const int splits = r.splits[i];
  const int64_t nvec = M * (N / 4), stride = (int64_t)r.blocks[i] * blockDim.x;
  for (int64_t idx = (int64_t)(bid - r.start[i]) * blockDim.x + threadIdx.x; idx < nvec; idx += stride) {
    const int64_t m = idx / (N / 4), n = (idx % (N / 4)) * 4;
    splitk_reduce_vec<T, MODE>(ws, C + m * ldc + n, nullptr, nullptr, M, N, m, n, splits);
  }
}

// ============================================================================================ host dispatch
#ifndef TAMD_GEMM_KERNELS_ONLY  // (tools/gemm_isa.sh instantiates single kernels for a look at their ISA)
#ifdef TAMD_DIAG
// ablation selector of the diagnostic build (tamd_gemm_set_dbg, include/tamd_diag.h; TAMD_GEMM_DBG in the environment
// sets the initial value): see the DBG template parameter of gemm_fl_kernel
static int g_gemm_dbg = -1;
static int gemm_diag_dbg() {
  if (g_gemm_dbg < 0) {
    const char* e = getenv("TAMD_GEMM_DBG");
    g_gemm_dbg = e ? atoi(e) : 0;
  }
  return g_gemm_dbg;
}
#endif
#define TAMD_EPI_SWITCH(LAUNCH)                                           \
  switch (epilogue) {                                                     \
    case TAMD_EPI_NONE: LAUNCH(TAMD_EPI_NONE, TAMD_ACT_NONE)              \
    case TAMD_EPI_BIAS: LAUNCH(TAMD_EPI_BIAS, TAMD_ACT_NONE)              \
    case TAMD_EPI_RESIDUAL: LAUNCH(TAMD_EPI_RESIDUAL, TAMD_ACT_NONE)      \
    case TAMD_EPI_ACCUM: LAUNCH(TAMD_EPI_ACCUM, TAMD_ACT_NONE)            \
    case kEpiColScale: LAUNCH(kEpiColScale, TAMD_ACT_NONE)                \
    case TAMD_EPI_BIAS_ACT:                                               \
      switch (act) {                                                      \
        case TAMD_ACT_GELU_ERF: LAUNCH(TAMD_EPI_BIAS_ACT, TAMD_ACT_GELU_ERF)     \
        case TAMD_ACT_GELU_TANH: LAUNCH(TAMD_EPI_BIAS_ACT, TAMD_ACT_GELU_TANH)   \
        case TAMD_ACT_QUICK_GELU: LAUNCH(TAMD_EPI_BIAS_ACT, TAMD_ACT_QUICK_GELU) \
        case TAMD_ACT_SILU: LAUNCH(TAMD_EPI_BIAS_ACT, TAMD_ACT_SILU)      \
        default: return TAMD_E_ARG;                                       \
      }                                                                   \
    default: return TAMD_E_ARG;                                           \
  }

template <typename T, bool A_KM, bool B_KN>
static int gemm_pp_launch_epi(const GemmArgs& g, int epilogue, int act, hipStream_t s) {
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n)), block(kGemmThreads);
#define TAMD_G(E_, A_)                                                                                \
  hipLaunchKernelGGL((gemm_pp_kernel<T, A_KM, B_KN, E_, A_>), grid, block, (size_t)kGemmSmem, s, g); \
  return launch_status();
  TAMD_EPI_SWITCH(TAMD_G)
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

constexpr int kPersistGrid = 256;  // one workgroup per CU of an MI355X (the split-K policy below counts the same 256)
template <typename T, bool A_KM, bool B_KN>
static int gemm_fl_launch_epi(const GemmArgs& g, int epilogue, int act, hipStream_t s) {
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n)), block(kFlThreads);
#ifdef TAMD_DIAG  // ablation / A-B instantiations: libtamd_diag.so only, never the product library
  const int dbg = gemm_diag_dbg();
  if ((dbg == 32 || dbg == 128) && epilogue == TAMD_EPI_NONE) {  // the other piece placement (correct results)
    if (dbg == 32)
      hipLaunchKernelGGL((gemm_fl_kernel<T, A_KM, B_KN, TAMD_EPI_NONE, TAMD_ACT_NONE, 32>), grid, block, (size_t)kXSmem, s, g);
    else
      hipLaunchKernelGGL((gemm_fl_kernel<T, A_KM, B_KN, TAMD_EPI_NONE, TAMD_ACT_NONE, 128>), grid, block, (size_t)kXSmem, s, g);
    return launch_status();
  }
  if constexpr (A_KM == B_KN) {
    if (dbg == 32768 && (epilogue == TAMD_EPI_NONE || epilogue == TAMD_EPI_RESIDUAL)) {  // the persistent walk (prefetch under the way out), forced
      const int tiles = g.tiles_m * g.tiles_n;
      static const int want = [] {  // (TAMD_PERSIST_GRID: the CPU model's test walks several tiles per workgroup on a 6-tile product)
        const char* e = getenv("TAMD_PERSIST_GRID");
        return e ? atoi(e) : kPersistGrid;
      }();
      dim3 pgrid((unsigned)(tiles < want ? tiles : want));
      if (epilogue == TAMD_EPI_NONE)
        hipLaunchKernelGGL((gemm_fl_persist_kernel<T, A_KM, B_KN, TAMD_EPI_NONE, TAMD_ACT_NONE>), pgrid, block, (size_t)kXSmem, s, g);
      else
        hipLaunchKernelGGL((gemm_fl_persist_kernel<T, A_KM, B_KN, TAMD_EPI_RESIDUAL, TAMD_ACT_NONE>), pgrid, block, (size_t)kXSmem, s, g);
      return launch_status();
    }
  }
  if (dbg == 1024 + 16384 && epilogue == TAMD_EPI_NONE) {  // the vendor table with a counted wait at the stage boundary
    hipLaunchKernelGGL((gemm_fl_kernel<T, A_KM, B_KN, TAMD_EPI_NONE, TAMD_ACT_NONE, 1024 + 16384>), grid, block, (size_t)kXSmem, s, g);
    return launch_status();
  }
  if ((dbg == 1024 + 4096 || dbg == 1024 + 8192) && epilogue == TAMD_EPI_NONE) {  // placements of the three-barrier loop: pairs / SPREAD
    if (dbg == 1024 + 4096)
      hipLaunchKernelGGL((gemm_fl_kernel<T, A_KM, B_KN, TAMD_EPI_NONE, TAMD_ACT_NONE, 1024 + 4096>), grid, block, (size_t)kXSmem, s, g);
    else
      hipLaunchKernelGGL((gemm_fl_kernel<T, A_KM, B_KN, TAMD_EPI_NONE, TAMD_ACT_NONE, 1024 + 8192>), grid, block, (size_t)kXSmem, s, g);
    return launch_status();
  }
  if ((dbg == 1024 || dbg == 2048) && epilogue == TAMD_EPI_NONE) {  // round 6: the three-barrier loop / the one-barrier ring, forced
    if (dbg == 1024)
      hipLaunchKernelGGL((gemm_fl_kernel<T, A_KM, B_KN, TAMD_EPI_NONE, TAMD_ACT_NONE, 1024>), grid, block, (size_t)kXSmem, s, g);
    else
      hipLaunchKernelGGL((gemm_fl_kernel<T, A_KM, B_KN, TAMD_EPI_NONE, TAMD_ACT_NONE, 2048>), grid, block, (size_t)kXSmem, s, g);
    return launch_status();
  }
  if constexpr (!A_KM) {  // round-5 placements (forward and dX layouts; correct, bit-identical results)
    if ((dbg == 64 || dbg == 256 || dbg == 512) && epilogue == TAMD_EPI_NONE) {
      if (dbg == 64)
        hipLaunchKernelGGL((gemm_fl_kernel<T, A_KM, B_KN, TAMD_EPI_NONE, TAMD_ACT_NONE, 64>), grid, block, (size_t)kXSmem, s, g);
      else if (dbg == 256)
        hipLaunchKernelGGL((gemm_fl_kernel<T, A_KM, B_KN, TAMD_EPI_NONE, TAMD_ACT_NONE, 256>), grid, block, (size_t)kXSmem, s, g);
      else
        hipLaunchKernelGGL((gemm_fl_kernel<T, A_KM, B_KN, TAMD_EPI_NONE, TAMD_ACT_NONE, 512>), grid, block, (size_t)kXSmem, s, g);
      return launch_status();
    }
  }
  if (dbg && epilogue == TAMD_EPI_NONE && !A_KM && !B_KN) {
#define TAMD_GD(N_)                                                                                             \
  hipLaunchKernelGGL((gemm_fl_kernel<T, false, false, TAMD_EPI_NONE, TAMD_ACT_NONE, N_>), grid, block, (size_t)kXSmem, \
                     s, g);                                                                                     \
  return launch_status();
    switch (dbg) {
      case 1: TAMD_GD(1)
      case 2: TAMD_GD(2)
      case 4: TAMD_GD(4)
      case 8: TAMD_GD(8)
      case 12: TAMD_GD(12)
      case 15: TAMD_GD(15)
      default: break;
    }
#undef TAMD_GD
  }
#endif
#define TAMD_G(E_, A_)                                                                              \
  hipLaunchKernelGGL((gemm_fl_kernel<T, A_KM, B_KN, E_, A_>), grid, block, (size_t)kXSmem, s, g); \
  return launch_status();
  TAMD_EPI_SWITCH(TAMD_G)
#undef TAMD_G
}

// split-K: partial tiles into the fp32 workspace, then the reduction, which applies the epilogue (plain / bias / residual /
// accumulate)
template <typename T, bool A_KM, bool B_KN>
static int gemm_fl_splitk_launch2(const GemmArgs& g, int epilogue, hipStream_t s) {
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n * g.splits)), block(kFlThreads);
  hipLaunchKernelGGL((gemm_fl_kernel<T, A_KM, B_KN, kEpiSplitK, TAMD_ACT_NONE>), grid, block, (size_t)kXSmem, s, g);
  const int64_t nvec = g.M * (g.N / 4);
  int64_t blocks = ceil_div(nvec, 256);
  if (blocks > 4096) blocks = 4096;
#define TAMD_RK(MODE_)                                                                                                   \
  hipLaunchKernelGGL((splitk_reduce_kernel<T, MODE_>), dim3((unsigned)blocks), dim3(256), 0, s, g.ws, (T*)g.C,          \
                     (const T*)g.bias, (const T*)g.R, g.M, g.N, g.ldc, g.ldr, g.splits, (T*)g.C_seg1, (T*)g.C_seg2,     \
                     g.seg_row1, g.seg_row2)
  switch (epilogue) {
    case TAMD_EPI_ACCUM: TAMD_RK(TAMD_EPI_ACCUM); break;
    case TAMD_EPI_BIAS: TAMD_RK(TAMD_EPI_BIAS); break;
    case TAMD_EPI_RESIDUAL: TAMD_RK(TAMD_EPI_RESIDUAL); break;
    default: TAMD_RK(TAMD_EPI_NONE); break;
  }
#undef TAMD_RK
  return launch_status();
}

template <typename T>
static int gemm_fl_splitk_launch(const GemmArgs& g, int flags, int epilogue, hipStream_t s) {
  const bool akm = flags & TAMD_GEMM_A_KM, bkn = flags & TAMD_GEMM_B_KN;
  if (!akm && !bkn) return gemm_fl_splitk_launch2<T, false, false>(g, epilogue, s);
  if (!akm && bkn) return gemm_fl_splitk_launch2<T, false, true>(g, epilogue, s);
  if (akm && bkn) return gemm_fl_splitk_launch2<T, true, true>(g, epilogue, s);
  return gemm_fl_splitk_launch2<T, true, false>(g, epilogue, s);
}

template <typename T>
static int gemm_fl_launch(const GemmArgs& g, int flags, int epilogue, int act, hipStream_t s) {
  const bool akm = flags & TAMD_GEMM_A_KM, bkn = flags & TAMD_GEMM_B_KN;
  if (!akm && !bkn) return gemm_fl_launch_epi<T, false, false>(g, epilogue, act, s);
  if (!akm && bkn) return gemm_fl_launch_epi<T, false, true>(g, epilogue, act, s);
  if (akm && bkn) return gemm_fl_launch_epi<T, true, true>(g, epilogue, act, s);
  return gemm_fl_launch_epi<T, true, false>(g, epilogue, act, s);
}

// the 128 x 128 kernel (row-major operands, K % 64 == 0): tiles_m / tiles_n re-counted for its tile
template <typename T>
static int gemm_sm_launch(GemmArgs g, int epilogue, int act, hipStream_t s) {
  g.tiles_m = (int)ceil_div(g.M, kSmTile);
  g.tiles_n = (int)ceil_div(g.N, kSmTile);
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n)), block(kSmThreads);
#define TAMD_G(E_, A_)                                                                    \
  hipLaunchKernelGGL((gemm_sm_kernel<T, E_, A_>), grid, block, (size_t)kSmSmem, s, g); \
  return launch_status();
  TAMD_EPI_SWITCH(TAMD_G)
#undef TAMD_G
}

}  // namespace tamd

using namespace tamd;

#ifdef TAMD_DIAG
extern "C" int tamd_gemm_set_dbg(int dbg) {
  g_gemm_dbg = dbg;
  return TAMD_OK;
}
static unsigned long long* g_gemm_clock = nullptr;
extern "C" int tamd_gemm_set_clock_buffer(void* buf) {
  g_gemm_clock = reinterpret_cast<unsigned long long*>(buf);
  return TAMD_OK;
}
static unsigned long long* g_gemm_timeline = nullptr;
extern "C" int tamd_gemm_set_timeline_buffer(void* buf) {  // 3 x uint64 per workgroup of the next gemm_fl_kernel launches
  g_gemm_timeline = reinterpret_cast<unsigned long long*>(buf);
  return TAMD_OK;
}
#endif

static int gemm_fill_args(GemmArgs* g, const void* A, const void* B, void* C, const void* bias, const void* R,
                          int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr) {
  g->A = A;
  g->B = B;
  g->C = C;
  g->bias = bias;
  g->R = R;
  g->M = M;
  g->N = N;
  g->K = K;
  g->lda = lda;
  g->ldb = ldb;
  g->ldc = ldc;
  g->ldr = ldr;
  g->tiles_m = (int)ceil_div(M, kBM);
  g->tiles_n = (int)ceil_div(N, kBN);
#ifdef TAMD_DIAG
  g->trace = g_gemm_clock;
  g->timeline = g_gemm_timeline;
#else
  g->trace = nullptr;
  g->timeline = nullptr;
#endif
  g->ws = nullptr;
  g->splits = 1;
  g->stages_per_split = 0;
  g->C2 = nullptr;
  g->ldc2 = 0;
  g->n_half = 0;
  g->scale_cols = 0;
  g->col_scale = 1.f;
  g->C_seg1 = g->C_seg2 = nullptr;
  g->seg_row1 = g->seg_row2 = 0;
  return TAMD_OK;
}

#ifdef TAMD_DIAG
// Diagnostic: C = A.B^T (bf16 row-major operands, no epilogue) on the ping-pong kernel while workgroup 0 writes
// 8 shader-clock stamps per sub-tile and wave into `trace` (8 waves x 32 sub-tiles x 8 u64): tools/gemm_phase_trace.py
extern "C" int tamd_gemm_trace(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, void* trace,
                               tamd_stream_t stream) {
  if (!A || !B || !C || !trace) return TAMD_E_NULL;
  if ((K % 8) || (N % 8)) return TAMD_E_SHAPE;
  GemmArgs g;
  gemm_fill_args(&g, A, B, C, nullptr, nullptr, M, N, K, K, K, N, 0);
  g.trace = reinterpret_cast<unsigned long long*>(trace);
  hipLaunchKernelGGL((gemm_pp_kernel<bf16_t, false, false, TAMD_EPI_NONE, TAMD_ACT_NONE, true>),
                     dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(kGemmThreads), (size_t)kGemmSmem,
                     TAMD_STREAM(stream), g);
  return launch_status();
}
#endif  // TAMD_DIAG

// Split-K policy.  (1) A 256x256 tile grid that cannot fill the 256 CUs (weight gradients of narrow layers: dW of a
// 768x3072 BERT projection is 36 tiles over K = tokens) is cut along K so that tiles x splits ~ one workgroup per CU,
// at least 8 stages (512 k) per split.  (2) A grid of a few rounds whose last round is mostly empty (dW of the Llama-3-8B
// q|k|v projection: 384 tiles = 1.5 rounds of 256 CUs, 25 % of the machine idle) is cut in 2..4 when the rounds saved
// outweigh writing and re-reading the fp32 partial tiles:
//     cost(s) = ceil(tiles * s / 256) / s * stages * 1.4 us  +  (s > 1) * s * M * N * 8 B / 4 TB/s   (s = 1..4)
//     (forward products -- both operands row-major -- of fewer than 64 stages are not split: the 128 x 128 kernel spreads them
//     over the GPU without partial tiles, 20-37 us against 25-50 for K = 2048 over 8 .. 128 tiles; from 64 stages on
//     split-K wins, 58 vs 65 us at K = 4096 and 105 vs 158 at 11008: profiles/r03q_gemm_sm_ab.jsonl)
static int gemm_choose_splits(int64_t M, int64_t N, int64_t K, int flags, int epilogue, int* stages_per_split) {
  *stages_per_split = 0;
  if (K % kXK != 0 || (N % 4) != 0 || epilogue == TAMD_EPI_BIAS_ACT || epilogue < TAMD_EPI_NONE || epilogue > TAMD_EPI_ACCUM) return 1;
  if ((flags & (TAMD_GEMM_A_KM | TAMD_GEMM_B_KN)) == 0 && M <= kGemvMaxRows && epilogue <= TAMD_EPI_RESIDUAL &&
      !(M > kGemvValuRows && N >= 65536))
    return 1;  // gemv.hip
  const int64_t tiles = ceil_div(M, kBM) * ceil_div(N, kBN), nst = K / kXK;
  if (nst < 32) return 1;
  if ((flags & (TAMD_GEMM_A_KM | TAMD_GEMM_B_KN)) == 0 && tiles <= kSmMaxBigTiles && nst < 64) return 1;
  int64_t s = 1;
  if (tiles <= 128) {
    s = 256 / tiles;
    if (s > nst / 8) s = nst / 8;
    if (s > 16) s = 16;
  } else if (tiles < 2048 && nst >= 128) {  // integer nanoseconds (ops.gemm_workspace_bytes mirrors this exactly)
    int64_t best = 0;
    for (int64_t c = 1; c <= 4; ++c) {
      const int64_t cost = ceil_div(tiles * c, 256) * nst * 1400 / c + (c > 1 ? c * M * N / 500 : 0);
      if (c == 1 || cost * 100 < best * 97) best = cost, s = c;  // a split has to win by 3 %
    }
  }
  if (s < 2) return 1;
  const int64_t sps = ceil_div(nst, s);
  *stages_per_split = (int)sps;
  return (int)ceil_div(nst, sps);  // no empty split
}

extern "C" size_t tamd_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int flags, int epilogue) {
  int sps;
  const int splits = gemm_choose_splits(M, N, K, flags, epilogue, &sps);
  return splits > 1 ? (size_t)splits * (size_t)M * (size_t)N * sizeof(float) : 0;
}

extern "C" int tamd_gemm(const void* A, const void* B, void* C, const void* bias, const void* R, int64_t M, int64_t N,
                         int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int flags, int epilogue,
                         int act, int dtype, tamd_stream_t stream) {
  return tamd_gemm_ws(A, B, C, bias, R, M, N, K, lda, ldb, ldc, ldr, flags, epilogue, act, dtype, nullptr, 0, stream);
}

// argument checks shared by the plain-GEMM entry points
static int gemm_check(const void* A, const void* B, const void* C, const void* bias, const void* R, int64_t M, int64_t N,
                      int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int flags, int epilogue) {
  if (!A || !B || !C) return TAMD_E_NULL;
  if (M <= 0 || N <= 0 || K <= 0) return TAMD_E_SHAPE;
  // 16-byte accesses run along the contiguous dimension of each operand: K for a row-major operand, M (A) / N (B)
  // for a k-major one.  With both operands k-major K is a row count and may be anything (dW = dY^T.X over a
  // ragged number of tokens).
  const bool k_contig = !(flags & TAMD_GEMM_A_KM) || !(flags & TAMD_GEMM_B_KN);
  if ((k_contig && (K % 8)) || (N % 8) || (lda % 8) || (ldb % 8) || (ldc % 8)) return TAMD_E_SHAPE;
  if ((flags & TAMD_GEMM_A_KM) && (M % 8)) return TAMD_E_SHAPE;
  if (!aligned16(A) || !aligned16(B) || !aligned16(C)) return TAMD_E_ALIGN;
  if ((epilogue == TAMD_EPI_BIAS || epilogue == TAMD_EPI_BIAS_ACT) && !bias) return TAMD_E_NULL;
  if (bias && (reinterpret_cast<uintptr_t>(bias) & 7u)) return TAMD_E_ALIGN;
  if (epilogue == TAMD_EPI_RESIDUAL && (!R || (ldr % 8) || !aligned16(R))) return R ? TAMD_E_ALIGN : TAMD_E_NULL;
  return TAMD_OK;
}

// kernel selection for a filled GemmArgs: split-K (when a workspace allows it), the 128 x 128 tile for small row-major
// grids, the full-line kernel whenever K % 64 == 0, else the ping-pong kernel.  TAMD_GEMM=pp in the environment (diagnostic
// build only) or a
// TAMD_GEMM_SCHED_* hint in `flags` forces one (A/B measurements, tests).
static int gemm_run(GemmArgs& g, int flags, int epilogue, int act, int dtype, void* workspace, size_t workspace_bytes,
                    tamd_stream_t stream) {
  const int64_t M = g.M, N = g.N, K = g.K;
#ifdef TAMD_DIAG  // (libtamd_diag.so only: the product library reads no schedule from the environment)
  static const int forced = [] {
    const char* e = getenv("TAMD_GEMM");
    return !e ? 0 : (e[0] == 'p' ? 1 : (e[0] == 'x' ? 3 : (e[0] == 's' ? 2 : 0)));
  }();
#else
  constexpr int forced = 0;
#endif
  const int sched = (flags >> 8) & 7 ? (flags >> 8) & 7 : forced;  // per-call hint wins over the environment
  flags &= 0xff;
  // a cached decode step's projections (M = batch <= 8, row-major weight): streamed by the VALU kernel of gemv.hip -- bound by
  // HBM, where N / 256 MFMA tiles with one live row each are not.  A schedule hint keeps the product on the tile kernels.
  // (except the widest products at 5+ rows -- an lm_head of 128256 rows: 1 GB of weights is 501 tile columns, the 256 x 256
  // kernel streams it at 5.4 TB/s, 188 us, where the MFMA streaming kernel re-reads X per workgroup: 207-225 us,
  // profiles/r04s_gemv_bench.jsonl)
  if (sched == 0 && flags == 0 && M <= kGemvMaxRows && !(M > kGemvValuRows && N >= 65536 && K % kXK == 0) &&
      (epilogue == TAMD_EPI_NONE || epilogue == TAMD_EPI_BIAS || epilogue == TAMD_EPI_RESIDUAL)) {
    const GemvArgs v{g.A, g.B, g.C, g.bias, g.R, M, N, K, g.lda, g.ldb, g.ldc, g.ldr};
    return gemv_run(v, epilogue, dtype, TAMD_STREAM(stream));
  }
  if (sched != 1 && workspace != nullptr) {
    int sps;
    const int splits = gemm_choose_splits(M, N, K, flags, epilogue, &sps);
    if (splits > 1 && workspace_bytes >= (size_t)splits * (size_t)M * (size_t)N * sizeof(float) && aligned16(workspace)) {
      g.ws = reinterpret_cast<float*>(workspace);
      g.splits = splits;
      g.stages_per_split = sps;
      TAMD_DISPATCH_HALF(dtype, return (gemm_fl_splitk_launch<T>(g, flags, epilogue, TAMD_STREAM(stream))));
    }
  }
  // 128 x 128 tiles when the 256 x 256 grid would leave most of the GPU idle (and split-K, above, did not take the
  // problem: bias / activation / residual epilogues, short K): row-major operands only.  Threshold from the A/B of the two
  // kernels over tile counts, profiles/r03q_gemm_sm_ab.jsonl
  if (K % kXK == 0 && flags == 0 && (sched == 2 || (sched == 0 && (int64_t)g.tiles_m * g.tiles_n <= kSmMaxBigTiles))) {
    TAMD_DISPATCH_HALF(dtype, return (gemm_sm_launch<T>(g, epilogue, act, TAMD_STREAM(stream))));
  }
  if (K % kXK == 0 && sched != 1) {
    TAMD_DISPATCH_HALF(dtype, return (gemm_fl_launch<T>(g, flags, epilogue, act, TAMD_STREAM(stream))));
  } else {
    TAMD_DISPATCH_HALF(dtype, return (gemm_pp_launch<T>(g, flags, epilogue, act, TAMD_STREAM(stream))));
  }
  return TAMD_E_DTYPE;
}

extern "C" int tamd_gemm_ws(const void* A, const void* B, void* C, const void* bias, const void* R, int64_t M,
                            int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int flags,
                            int epilogue, int act, int dtype, void* workspace, size_t workspace_bytes,
                            tamd_stream_t stream) {
  const int st = gemm_check(A, B, C, bias, R, M, N, K, lda, ldb, ldc, ldr, flags, epilogue);
  if (st != TAMD_OK) return st;
  if (epilogue < TAMD_EPI_NONE || epilogue > TAMD_EPI_ACCUM) return TAMD_E_ARG;
  GemmArgs g;
  gemm_fill_args(&g, A, B, C, bias, R, M, N, K, lda, ldb, ldc, ldr);
  return gemm_run(g, flags, epilogue, act, dtype, workspace, workspace_bytes, stream);
}

// dW = dY^T . X of a FUSED projection (weight rows = [q | k | v] or [gate | up]) with every member's rows stored into its own
// buffer: the members' gradients are separate tensors (DDP's bucket views; `.grad` of separate parameters), not slices of
// one.  Both operands k-major; plain or accumulating epilogue; segments are whole 256-row tiles.
extern "C" int tamd_gemm_seg(const void* A, const void* B, void* const* C_segs, const int64_t* seg_rows, int nseg, int64_t N,
                             int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int epilogue, int dtype, void* workspace,
                             size_t workspace_bytes, tamd_stream_t stream) {
  if (!C_segs || !seg_rows) return TAMD_E_NULL;
  if (nseg < 1 || nseg > 3 || (epilogue != TAMD_EPI_NONE && epilogue != TAMD_EPI_ACCUM)) return TAMD_E_ARG;
  int64_t M = 0;
  for (int i = 0; i < nseg; ++i) {
    if (!C_segs[i]) return TAMD_E_NULL;
    if (seg_rows[i] <= 0 || (i + 1 < nseg && seg_rows[i] % kBM != 0)) return TAMD_E_SHAPE;  // (the last may be ragged)
    if (!aligned16(C_segs[i])) return TAMD_E_ALIGN;
    M += seg_rows[i];
  }
  const int flags = TAMD_GEMM_A_KM | TAMD_GEMM_B_KN;
  const int st = gemm_check(A, B, C_segs[0], nullptr, nullptr, M, N, K, lda, ldb, ldc, 0, flags, epilogue);
  if (st != TAMD_OK) return st;
  if (K % kXK != 0) return TAMD_E_SHAPE;  // (the full-line kernel and its split-K: the ping-pong kernel has no segments)
  GemmArgs g;
  gemm_fill_args(&g, A, B, C_segs[0], nullptr, nullptr, M, N, K, lda, ldb, ldc, 0);
  if (nseg > 1) {
    g.C_seg1 = C_segs[1];
    g.seg_row1 = seg_rows[0];
  }
  if (nseg > 2) {
    g.C_seg2 = C_segs[2];
    g.seg_row2 = seg_rows[0] + seg_rows[1];
  }
  // only gemm_fl_kernel and its split-K honour the segments: always pin the schedule (a TAMD_GEMM=pp in the environment must not
  // send a segmented product to the ping-pong kernel, which would store past the first segment); split-K stays possible
  return gemm_run(g, flags | TAMD_GEMM_SCHED_FL, epilogue, TAMD_ACT_NONE, dtype, workspace, workspace_bytes, stream);
}

// ---- grouped launch (kernels: gemm_fl_group_kernel, splitk_reduce_group_kernel)
// The plan: one stage count L per workgroup for every product -- the smallest L (>= 16 stages) with
// sum_p tiles_p * ceil(stages_p / L) <= 256 workgroups: one dispatch round, equal K ranges.  Products too many for one round
// (more than 256 tiles) run unsplit.  `allow_split` false (no workspace): unsplit as well.
struct GroupPlan {
  int splits[kGroupMax], sps[kGroupMax];
  size_t ws_floats[kGroupMax];  // partial planes of product p
  bool any_split;
};
static int gemm_group_plan(const tamd_gemm_problem* pr, int count, int flags, bool allow_split, GroupPlan* plan) {
  if (!pr) return TAMD_E_NULL;
  if (count < 1 || count > kGroupMax) return TAMD_E_ARG;
  int64_t nst[kGroupMax], tiles[kGroupMax], max_nst = 0, sum_tiles = 0;
  for (int i = 0; i < count; ++i) {
    const tamd_gemm_problem& q = pr[i];
    const int st = gemm_check(q.a, q.b, q.c, nullptr, nullptr, q.m, q.n, q.k, q.lda, q.ldb, q.ldc, 0, flags, TAMD_EPI_NONE);
    if (st != TAMD_OK) return st;
    if (q.k % kXK != 0) return TAMD_E_SHAPE;  // (the full-line kernel)
    nst[i] = q.k / kXK;
    tiles[i] = ceil_div(q.m, kBM) * ceil_div(q.n, kBN);
    if (nst[i] > max_nst) max_nst = nst[i];
    sum_tiles += tiles[i];
  }
  int64_t L = max_nst;
  if (allow_split && sum_tiles <= 256) {
    for (int64_t c = 16; c < max_nst; ++c) {
      int64_t wgs = 0;
      for (int i = 0; i < count; ++i) wgs += tiles[i] * ceil_div(nst[i], c);
      if (wgs <= 256) {
        L = c;
        break;
      }
    }
  }
  plan->any_split = false;
  for (int i = 0; i < count; ++i) {
    const int64_t s0 = ceil_div(nst[i], L), sps = ceil_div(nst[i], s0);
    plan->sps[i] = (int)sps;
    plan->splits[i] = (int)ceil_div(nst[i], sps);  // no empty split
    plan->any_split = plan->any_split || plan->splits[i] > 1;
  }
  for (int i = 0; i < count; ++i)
    plan->ws_floats[i] = plan->any_split ? (size_t)plan->splits[i] * (size_t)pr[i].m * (size_t)pr[i].n : 0;
  return TAMD_OK;
}

extern "C" size_t tamd_gemm_group_workspace_bytes(const tamd_gemm_problem* problems, int count, int flags) {
  GroupPlan plan;
  if (gemm_group_plan(problems, count, flags & 0xff, true, &plan) != TAMD_OK) return 0;
  size_t total = 0;
  for (int i = 0; i < count; ++i) total += plan.ws_floats[i];
  return total * sizeof(float);
}

template <typename T, bool A_KM, bool B_KN>
static int gemm_group_launch(const tamd_gemm_problem* pr, int count, const GroupPlan& plan, int epilogue, float* ws,
                             hipStream_t s) {
  GemmGroupArgs grp;
  ReduceGroupArgs red;
  int next = 0, rnext = 0;
  for (int i = 0; i < kGroupMax; ++i) {
    grp.start[i] = red.start[i] = INT_MAX;
    grp.blocks[i] = red.blocks[i] = 0;
    if (i >= count) {
      grp.p[i] = grp.p[0];
      red.ws[i] = nullptr, red.C[i] = nullptr, red.M[i] = red.N[i] = red.ldc[i] = 0, red.splits[i] = 0;
      continue;
    }
    const tamd_gemm_problem& q = pr[i];
    GemmArgs& g = grp.p[i];
    gemm_fill_args(&g, q.a, q.b, q.c, nullptr, nullptr, q.m, q.n, q.k, q.lda, q.ldb, q.ldc, 0);
    grp.start[i] = next;
    if (plan.any_split) {
      g.ws = ws;
      g.splits = plan.splits[i];
      g.stages_per_split = plan.sps[i];
      grp.blocks[i] = g.tiles_m * g.tiles_n * g.splits;
      next += grp.blocks[i];
      const int64_t nvec = q.m * (q.n / 4);
      int64_t blocks = ceil_div(nvec, 256);
      if (blocks > 1024) blocks = 1024;
      red.ws[i] = ws, red.C[i] = q.c, red.M[i] = q.m, red.N[i] = q.n, red.ldc[i] = q.ldc, red.splits[i] = g.splits;
      red.start[i] = rnext, red.blocks[i] = (int)blocks;
      rnext += (int)blocks;
      ws += plan.ws_floats[i];
    } else {
      grp.blocks[i] = g.tiles_m * g.tiles_n;
      next += (grp.blocks[i] + 7) & ~7;  // (blockIdx & 7 stays the XCD of the product's own workgroup index)
    }
  }
  dim3 grid((unsigned)next), block(kFlThreads);
  if (plan.any_split) {
    hipLaunchKernelGGL((gemm_fl_group_kernel<T, A_KM, B_KN, kEpiSplitK>), grid, block, (size_t)kXSmem, s, grp);
    if (epilogue == TAMD_EPI_ACCUM)
      hipLaunchKernelGGL((splitk_reduce_group_kernel<T, TAMD_EPI_ACCUM>), dim3((unsigned)rnext), dim3(256), 0, s, red);
    else
      hipLaunchKernelGGL((splitk_reduce_group_kernel<T, TAMD_EPI_NONE>), dim3((unsigned)rnext), dim3(256), 0, s, red);
  } else if (epilogue == TAMD_EPI_ACCUM) {
    hipLaunchKernelGGL((gemm_fl_group_kernel<T, A_KM, B_KN, TAMD_EPI_ACCUM>), grid, block, (size_t)kXSmem, s, grp);
  } else {
    hipLaunchKernelGGL((gemm_fl_group_kernel<T, A_KM, B_KN, TAMD_EPI_NONE>), grid, block, (size_t)kXSmem, s, grp);
  }
  return launch_status();
}

// Up to 4 independent products of one layout in ONE launch (+ one reduction launch when K is split):
//   C_p = A_p . B_p (plain) or C_p += A_p . B_p (TAMD_EPI_ACCUM), K_p % 64 == 0.
// Layouts: the weight-gradient one (TAMD_GEMM_A_KM | TAMD_GEMM_B_KN: the dW products of one layer's backward) and the
// row-major forward one (0).  `workspace` of tamd_gemm_group_workspace_bytes (smaller / null: no K split).
extern "C" int tamd_gemm_group(const tamd_gemm_problem* problems, int count, int flags, int epilogue, int dtype,
                               void* workspace, size_t workspace_bytes, tamd_stream_t stream) {
  flags &= 0xff;
  if (epilogue != TAMD_EPI_NONE && epilogue != TAMD_EPI_ACCUM) return TAMD_E_ARG;
  const bool akm = flags & TAMD_GEMM_A_KM, bkn = flags & TAMD_GEMM_B_KN;
  if (akm != bkn) return TAMD_E_ARG;
  GroupPlan plan;
  int st = gemm_group_plan(problems, count, flags, workspace != nullptr && aligned16(workspace), &plan);
  if (st != TAMD_OK) return st;
  if (plan.any_split) {
    size_t need = 0;
    for (int i = 0; i < count; ++i) need += plan.ws_floats[i];
    if (need * sizeof(float) > workspace_bytes) {
      st = gemm_group_plan(problems, count, flags, false, &plan);
      if (st != TAMD_OK) return st;
    }
  }
  float* ws = reinterpret_cast<float*>(workspace);
  if (akm) {
    TAMD_DISPATCH_HALF(dtype, return (gemm_group_launch<T, true, true>(problems, count, plan, epilogue, ws, TAMD_STREAM(stream))));
  } else {
    TAMD_DISPATCH_HALF(dtype, return (gemm_group_launch<T, false, false>(problems, count, plan, epilogue, ws, TAMD_STREAM(stream))));
  }
  return TAMD_E_DTYPE;
}

// A projection whose leading columns leave scaled (the query columns of a fused q|k|v projection, models/bert/
// modeling_bert.py:175-177, carrying the attention kernels' scale*log2(e): tamd_attn_params.q_prescaled):
//   C[m, n] = round((A . B^T [+ bias])[m, n] * (n < scale_cols ? col_scale : 1))      -- ONE rounding, like the reference's q
extern "C" int tamd_gemm_colscale(const void* A, const void* B, void* C, const void* bias, int64_t M, int64_t N, int64_t K,
                                  int64_t lda, int64_t ldb, int64_t ldc, int flags, int64_t scale_cols, float col_scale,
                                  int dtype, tamd_stream_t stream) {
  const int st = gemm_check(A, B, C, bias, nullptr, M, N, K, lda, ldb, ldc, 0, flags, TAMD_EPI_NONE);
  if (st != TAMD_OK) return st;
  if (scale_cols < 0 || scale_cols > N || (scale_cols % 4)) return TAMD_E_SHAPE;
  GemmArgs g;
  gemm_fill_args(&g, A, B, C, bias, nullptr, M, N, K, lda, ldb, ldc, 0);
  g.scale_cols = scale_cols;
  g.col_scale = col_scale;
  return gemm_run(g, flags, kEpiColScale, TAMD_ACT_NONE, dtype, nullptr, 0, stream);
}

// gate|up projection of LlamaMLP with the SiLU*up product in the GEMM epilogue (models/llama/modeling_llama.py:174-176):
//   GU[M, 2I] = X[M,K] . Wgu[2I,K]^T  (gate columns | up columns; skipped when GU == NULL),  ACT[M, I] = silu(gate) * up
// with the roundings of the unfused path (tamd_gemm then tamd_swiglu_fwd): results are bit-identical to it.
extern "C" int tamd_gemm_swiglu(const void* X, const void* Wgu, void* GU, void* ACT, int64_t M, int64_t I, int64_t K,
                                int64_t ldx, int64_t ldw, int64_t ldgu, int64_t ldact, int dtype, tamd_stream_t stream) {
  if (!X || !Wgu || !ACT) return TAMD_E_NULL;
  if (M <= 0 || I <= 0 || K <= 0) return TAMD_E_SHAPE;
  if ((K % kXK) || (I % 8) || (ldx % 8) || (ldw % 8) || (ldact % 8) || (GU && (ldgu % 8))) return TAMD_E_SHAPE;
  if (!aligned16(X) || !aligned16(Wgu) || !aligned16(ACT) || (GU && !aligned16(GU))) return TAMD_E_ALIGN;
  if ((int64_t)2 * I * ldw * 2 >= ((int64_t)1 << 31)) return TAMD_E_SHAPE;  // 32-bit buffer offsets over the fused weight
  if (M <= kGemvMaxRows) {  // a decode step: the weight-streaming kernel (gemv.hip), same bits
    const GemvArgs v{X, Wgu, ACT, nullptr, GU, M, I, K, ldx, ldw, ldact, ldgu};
    return gemv_swiglu_run(v, dtype, TAMD_STREAM(stream));
  }
  GemmArgs g;
  gemm_fill_args(&g, X, Wgu, GU, nullptr, nullptr, M, 2 * I, K, ldx, ldw, ldgu, 0);
  g.tiles_n = (int)ceil_div(I, kBN / 2);  // 128 features (gate + up columns) per 256-wide tile
  g.C2 = ACT;
  g.ldc2 = ldact;
  g.n_half = I;
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n)), block(kFlThreads);
  TAMD_DISPATCH_HALF(dtype, {
    hipLaunchKernelGGL((gemm_fl_kernel<T, false, false, kEpiSwiGLU, TAMD_ACT_NONE>), grid, block, (size_t)kXSmem,
                       TAMD_STREAM(stream), g);
    return launch_status();
  });
  return TAMD_E_DTYPE;
}

// dX of the down projection of LlamaMLP with the SiLU*up backward as its way out (models/llama/modeling_llama.py:174-176
// differentiated):   d_act = dY[M,K] . Wd[K,I]  (Wd = down_proj.weight, [hidden, I] row-major: the k-major B operand),
//   dGU[M, 2I] = [ round(round(d_act * up) * silu'(gate)) | round(d_act * round(silu(gate))) ]      GU = the forward's gate|up
// with the roundings of the unfused path (tamd_gemm with TAMD_GEMM_B_KN, then tamd_swiglu_bwd): results are bit-identical to it,
// d_act never reaches memory.
extern "C" int tamd_gemm_swiglu_bwd(const void* dY, const void* Wd, const void* GU, void* dGU, int64_t M, int64_t I, int64_t K,
                                    int64_t lddy, int64_t ldw, int64_t ldgu, int64_t lddgu, int dtype, tamd_stream_t stream) {
  if (!dY || !Wd || !GU || !dGU) return TAMD_E_NULL;
  if (M <= 0 || I <= 0 || K <= 0) return TAMD_E_SHAPE;
  if ((K % kXK) || (I % 8) || (lddy % 8) || (ldw % 8) || (ldgu % 8) || (lddgu % 8)) return TAMD_E_SHAPE;
  if (!aligned16(dY) || !aligned16(Wd) || !aligned16(GU) || !aligned16(dGU)) return TAMD_E_ALIGN;
  if ((int64_t)128 * ldgu * 2 >= ((int64_t)1 << 31) || (int64_t)128 * lddgu * 2 >= ((int64_t)1 << 31)) return TAMD_E_SHAPE;  // 32-bit buffer offsets
  GemmArgs g;
  gemm_fill_args(&g, dY, Wd, dGU, nullptr, GU, M, I, K, lddy, ldw, lddgu, ldgu);
  g.n_half = I;
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n)), block(kFlThreads);
  TAMD_DISPATCH_HALF(dtype, {
    hipLaunchKernelGGL((gemm_fl_kernel<T, false, true, kEpiSwiGLUBwd, TAMD_ACT_NONE>), grid, block, (size_t)kXSmem,
                       TAMD_STREAM(stream), g);
    return launch_status();
  });
  return TAMD_E_DTYPE;
}

#else
}  // namespace tamd
#endif  // TAMD_GEMM_KERNELS_ONLY
