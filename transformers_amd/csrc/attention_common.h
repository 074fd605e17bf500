// attention_common.h -- shared device code of the attention translation units (attention.hip: forward, delta, dQ;
// attention_bwd_dkdv.hip: dK/dV).  Both are compiled with -amdgpu-mfma-vgpr-form (accumulators in VGPRs: no AGPR
// copies in the softmax arithmetic); the dK/dV unit additionally turns SLP packing off (build.py).
#pragma once
#include <stdlib.h>

#include "common.h"
#include "dropout.h"

namespace tamd {

// a compile-time integer as a function argument (generic lambdas: `decltype(c)::value`)
template <int N>
struct IntC {
  static constexpr int value = N;
};

constexpr int kAttnThreads = 256;
constexpr int kQB = 128;   // query rows per workgroup (32 per wave)
constexpr int kKB = 64;    // keys per tile

__device__ __attribute__((aligned(16))) static const unsigned int g_zero16a[4] = {0u, 0u, 0u, 0u};

struct AttnArgs {
  const void* q;
  const void* k;
  const void* v;
  void* o;
  float* lse;
  const uint8_t* key_valid;
  const int* q_start;  // packed sequences: first visible key of each query, or null
  int batch, heads_q, heads_kv, seq_q, seq_k;
  int64_t qsb, qss, qsh, ksb, kss, ksh, vsb, vss, vsh, osb, oss, osh;
  float scale_log2;  // scale * log2(e)
  unsigned drop_thr;  // keep iff the element's 16-bit hash field >= drop_thr (0 = no dropout); dropout.h AttnDrop
  float drop_scale;   // 65536 / (65536 - drop_thr): 1 / (1 - p) of the quantised p (attention.hip make_args)
  unsigned seed_lo, seed_hi;  // the halves of attn_seed_mix(dropout seed) (dropout.h)
  const unsigned long long* seed_dev;  // non-null: the (unmixed) seed lives in device memory and replaces seed_lo / seed_hi
                                       // (tamd_attn_params.dropout_seed_dev: graph-replay-safe dropout)
  int nqt;           // query tiles per (b, h)
  int xcd_map;       // 1: (b,kv-head) groups pinned to XCDs
  int q_prescaled;   // 1: q already carries scale*log2(e) (include/tamd.h): no operand is scaled and re-rounded here
  unsigned long long* trace;  // diagnostic build: per-phase shader-clock sums of workgroup 0 (tamd_attn_set_trace), else null
  // split-KV forward (tamd_attn_decode: few query rows over a long key range): workgroup = (query tile, head, batch, split);
  // split s visits key tiles [s * tiles_per_split, ...) and writes its normalised fp32 output rows and their log-sum-exp to
  // o_part [splits][B][Sq][Hq][D] / lse_part [splits][B][Hq][Sq]; attn_combine_kernel merges them
  float* o_part;
  float* lse_part;
  int kv_splits, tiles_per_split;
};

// Diagnostic build only: phase i of the forward tile loop ends here (s_memtime stamps of workgroup 0, summed per wave)
#ifdef TAMD_DIAG
#define TAMD_ATTN_PHASE(i_)                              \
  if (tr) {                                              \
    asm volatile("" ::: "memory");                       \
    const unsigned long long now_ = device_clock();      \
    asm volatile("" ::: "memory");                       \
    ph[i_] += now_ - tlast;                              \
    tlast = now_;                                        \
  }
#else
#define TAMD_ATTN_PHASE(i_)
#endif

// the dropout context of a kernel: thresholds, the mixed seed (from the host, or mixed here from the device-resident word:
// a wave-uniform scalar load and a dozen scalar instructions per workgroup), block-index geometry
__device__ __forceinline__ AttnDrop attn_drop_ctx(const AttnArgs& a) {
  unsigned s0 = a.seed_lo, s1 = a.seed_hi;
  if (a.seed_dev != nullptr) {
    const unsigned long long mixed = attn_seed_mix(*a.seed_dev);
    s0 = (unsigned)mixed;
    s1 = (unsigned)(mixed >> 32);
  }
  return AttnDrop{a.drop_thr << 16, s0, s1, a.drop_scale, ((unsigned long long)a.seq_q + 1) >> 1,
                  ((unsigned long long)a.seq_k + 1) >> 1};
}

// One swizzle serves both read patterns of a [rows][D] tile (rows = keys or queries):
//   ds_read_b128 of 16 distinct rows at one logical slot  -> needs a bijection of the row bits onto slots,
//   ds_read_b64_tr_b16 of 4 consecutive rows x 64 B       -> needs the low row bits on the 64-byte window bits.
template <int D>
__device__ __forceinline__ int row_swz(int row) {
  return D == 128 ? (((row & 3) << 2) | ((row >> 2) & 3)) : ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
}

// one [64][D] tile: global rows `key0 + r` (stride `stride` elements) -> LDS at tile_off, swizzled by SWZ
template <typename T, int D>
__device__ __forceinline__ void issue_kv_tile(const T* __restrict__ base, int64_t stride, int key0, int nkeys,
                                              char* smem, unsigned tile_off, int wave, int lane) {
  constexpr int ROWB = D * 2;            // bytes per row
  constexpr int SLOTS = ROWB / 16;       // 16-byte slots per row
  constexpr int RPI = 1024 / ROWB;       // rows per wave instruction
  constexpr int NI = (kKB * ROWB) / 1024 / 4;  // instructions per wave (4 waves)
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int inst = wave * NI + i;
    const int r = inst * RPI + lane / SLOTS;
    const int p = lane % SLOTS;
    const int s = p ^ row_swz<D>(r);
    const int key = key0 + r;
    const void* src = (key < nkeys) ? (const void*)(base + (int64_t)key * stride + s * 8) : (const void*)g_zero16a;
    glds16(src, smem, tile_off + (unsigned)inst * 1024u);
  }
}

// The same tile through buffer-addressed LDS-DMA with loop-invariant lane offsets: the tile's first row goes into the
// wave-uniform base (one 64-bit scalar add per tile), nothing per-lane is recomputed -- issue_kv_tile spends ~14 VALU /
// SALU instructions per piece on 64-bit addresses, the bounds test and the zero page (110 per tile and wave, half of
// what the softmax costs).  Full tiles only (every row < nkeys) and 32-bit offsets (64 rows x stride x 2 B < 2^31); the
// callers take issue_kv_tile for the ragged last tile.
template <int D>
struct TileFeed {
  static constexpr int NI = (kKB * D * 2) / 1024 / 4;  // pieces per wave
  unsigned voff[NI];
  __device__ __forceinline__ void init(int64_t stride, int wave, int lane) {
    constexpr int ROWB = D * 2, SLOTS = ROWB / 16, RPI = 1024 / ROWB;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int inst = wave * NI + i;
      const int r = inst * RPI + lane / SLOTS;
      const int s = (lane % SLOTS) ^ row_swz<D>(r);
      voff[i] = (unsigned)(((int64_t)r * stride + s * 8) * 2);
    }
  }
  static __host__ __device__ __forceinline__ bool usable(int64_t stride) { return stride > 0 && stride * (kKB * 2) < ((int64_t)1 << 31); }
  template <typename T>
  __device__ __forceinline__ void issue_one(const T* __restrict__ first_row, char* smem, unsigned tile_off, int wave,
                                            int i) const {  // piece i < NI of this wave
    glds16_buf<0>(first_row, voff[i], smem, tile_off + (unsigned)(wave * NI + i) * 1024u);
  }
  // ... range-checked: `rows` (wave-uniform, 0 .. 64) of the tile exist, the others arrive as zeros -- the ragged last tile of a
  // sequence and "no next tile" (rows = 0: the buffer nobody reads is zero-filled) cost no branch and no second code path
  template <typename T>
  __device__ __forceinline__ void issue_one_rng(const T* __restrict__ first_row, int rows, int64_t stride, char* smem,
                                                unsigned tile_off, int wave, int i) const {
    glds16_buf_rng(first_row, (unsigned)rows * (unsigned)stride * 2u, voff[i], smem, tile_off + (unsigned)(wave * NI + i) * 1024u);
  }
  template <typename T>
  __device__ __forceinline__ void issue(const T* __restrict__ first_row, char* smem, unsigned tile_off, int wave) const {
#pragma unroll
    for (int i = 0; i < NI; ++i) issue_one(first_row, smem, tile_off, wave, i);
  }
};

// Loop-invariant LDS byte offsets of one lane inside a swizzled [64][D] tile (the swizzle terms depend on the
// lane only), so every tile read in the attention loops is `tile base + offset register + immediate`:
//   row[ks]   : ds_read_b128 of row l31 (add 32*ROWB for the second 32-row sub-tile), k-step ks
//   tr[dt][t] : ds_read_b64_tr_b16 of rows 4*hi+kq (+8*t), this lane's 4 columns of d-tile dt; MFMA step j adds
//               ((j>>1)*32 + (j&1)*16) * ROWB.  The row order delivered matches the C-layout registers
//               8*(j&1)+jj of sub-tile j>>1 (key/query = sub*32 + 16*(j&1) + 8*(jj>>2) + 4*hi + (jj&3)).
template <int D>
struct TileOffsets {
  unsigned row[D / 16];
  unsigned tr[D / 32][2];
  __device__ __forceinline__ void init(int lane) {
    constexpr int ROWB = D * 2;
    const int hi = lane >> 5, l31 = lane & 31, kq = (lane & 15) >> 2;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks)
      row[ks] = (unsigned)l31 * ROWB + (unsigned)(((ks * 2 + hi) ^ row_swz<D>(l31)) * 16);
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt) {
      const int col = dt * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const int r = 4 * hi + kq + 8 * t2;
        tr[dt][t2] = (unsigned)r * ROWB + (unsigned)(((col >> 3) ^ row_swz<D>(r)) * 16) + (unsigned)(col & 7) * 2u;
      }
    }
  }
  // MFMA A operand [32 d][16 rows] for step j (transposing reads)
  __device__ __forceinline__ u32x4 read_tr(const char* smem, unsigned tile_off, int dt, int j) const {
    const unsigned rb = (unsigned)(((j >> 1) * 32 + (j & 1) * 16) * (D * 2));
    const u32x2 lo = lds_read8_tr16(smem, tile_off + tr[dt][0] + rb);
    const u32x2 h2 = lds_read8_tr16(smem, tile_off + tr[dt][1] + rb);
    return u32x4{lo[0], lo[1], h2[0], h2[1]};
  }
  __device__ __forceinline__ u32x4 read_row(const char* smem, unsigned tile_off, int sub, int ks) const {
    return lds_read16(smem, tile_off + row[ks] + (unsigned)(sub * 32 * D * 2));
  }
};

// s_waitcnt lgkmcnt(N) that fragment `f` depends on: the MFMA consuming it cannot be scheduled above the wait, and the
// request of a later fragment (volatile asm too) cannot sink below it.  For LDS reads issued untracked (tamd_device.h)
// N = the number of reads issued after the one that fills `f` (LDS reads of a wave return in order).
template <int N>
__device__ __forceinline__ void wait_frag(u32x4& f) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N) : "memory");
#else
  (void)f;
#endif
}
// The two fragments are "produced" here: whatever computes them cannot be sunk below this point, nothing that reads them hoisted
// above it (an empty asm: no instruction)
__device__ __forceinline__ void pin_frags(u32x4& f0, u32x4& f1) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(f0), "+v"(f1));
#else
  (void)f0;
  (void)f1;
#endif
}
// ... for a PAIR of fragments (one s_waitcnt in front of two MFMAs: N counts the reads issued after the pair's second fragment)
template <int N>
__device__ __forceinline__ void wait_frag2(u32x4& f0, u32x4& f1) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f0), "+v"(f1) : "n"(N) : "memory");
#else
  (void)f0;
  (void)f1;
#endif
}
template <int CAP>
__device__ __forceinline__ void constexpr_wait_frag2(int n, u32x4& f0, u32x4& f1) {
  static_assert(CAP <= 15, "lgkmcnt is a 4-bit counter");
  if (n >= CAP) return wait_frag2<CAP>(f0, f1);
#define TAMD_WF(N_) \
  if (N_ < CAP && n == N_) return wait_frag2<(N_ < CAP ? N_ : 0)>(f0, f1);
  TAMD_WF(14) TAMD_WF(13) TAMD_WF(12) TAMD_WF(11) TAMD_WF(10) TAMD_WF(9) TAMD_WF(8) TAMD_WF(7) TAMD_WF(6) TAMD_WF(5) TAMD_WF(4)
  TAMD_WF(3) TAMD_WF(2) TAMD_WF(1)
#undef TAMD_WF
  wait_frag2<0>(f0, f1);
}
// wait_frag<min(n, CAP)> for a compile-time-foldable n (unrolled loop index arithmetic)
template <int CAP>
__device__ __forceinline__ void constexpr_wait_frag(int n, u32x4& f) {
  static_assert(CAP <= 15, "lgkmcnt is a 4-bit counter");
  if (n >= CAP) return wait_frag<CAP>(f);
#define TAMD_WF(N_) \
  if (N_ < CAP && n == N_) return wait_frag<(N_ < CAP ? N_ : 0)>(f);
  TAMD_WF(14) TAMD_WF(13) TAMD_WF(12) TAMD_WF(11) TAMD_WF(10) TAMD_WF(9) TAMD_WF(8) TAMD_WF(7) TAMD_WF(6) TAMD_WF(5) TAMD_WF(4)
  TAMD_WF(3) TAMD_WF(2) TAMD_WF(1)
#undef TAMD_WF
  wait_frag<0>(f);
}

struct AttnBwdArgs {
  AttnArgs f;
  const void* dout;
  void* dq;
  void* dk;
  void* dv;
  float* delta;
  float scale;
  // optional: the transposed rotary embedding applied to dq / dk on their way out (head_dim 128); null = none
  const void* rope_cos;
  const void* rope_sin;
  int rope_cos_batch;
};


// stage a wave's [32 rows][D] fp32-accumulator tile (C layout: lane column = row, registers = d) through LDS
// and write it as full rows:  dst[(row0 + r) * stride + d].
// cosp != null (D = 128): the rows are gradients of rotated query / key heads and leave through the transposed rotary
// embedding (the backward of apply_rotary_pos_emb, models/llama/modeling_llama.py:130-160):
//     dx[d] = round(round(dy[d]*cos[d]) + round(dy[d+64]*sin[d])),  dx[d+64] = round(round(dy[d+64]*cos[d]) - round(dy[d]*sin[d]))
// on the rounded staged values, with rope_kernel's roundings (elementwise.hip, conj): bit-identical to storing dy and
// running tamd_rope_inplace(conj) afterwards.  cos / sin rows: crow0 + r.
template <typename T, int D>
__device__ __forceinline__ void store_rows_via_lds(const f32x16* acc, float mul, char* smem, unsigned st_off, T* dst,
                                                   int64_t stride, int row0, int nrows, int lane,
                                                   const T* cosp = nullptr, const T* sinp = nullptr, int64_t crow0 = 0) {
  constexpr int ROWB = D * 2, OROWB = ROWB + 16, DT = D / 32;
  const int hi = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int d0 = dt * 32 + 8 * qd + 4 * hi;
      const u32x2 pk = {pack2<T>(acc[dt][qd * 4 + 0] * mul, acc[dt][qd * 4 + 1] * mul),
                        pack2<T>(acc[dt][qd * 4 + 2] * mul, acc[dt][qd * 4 + 3] * mul)};
      lds_write8(smem, st_off + (unsigned)l31 * OROWB + (unsigned)d0 * 2u, pk);
    }
  wave_lockstep_point();
  constexpr int SLOTS = ROWB / 16, RPI = 64 / SLOTS;
#pragma unroll
  for (int it = 0; it < 32 / RPI; ++it) {
    const int row = it * RPI + lane / SLOTS, slot = lane % SLOTS;
    u32x4 v = lds_read16(smem, st_off + (unsigned)row * OROWB + (unsigned)slot * 16u);
    if (D == 128 && cosp != nullptr) {  // (wave-uniform)
      const u32x4 vp = lds_read16(smem, st_off + (unsigned)row * OROWB + (unsigned)(slot ^ (SLOTS / 2)) * 16u);
      if (row0 + row < nrows) {
        float x[8], xp[8], cs[8], sn[8], o[8];
        unpack16<T>(v, x);
        unpack16<T>(vp, xp);
        unpack16<T>(ld16(cosp + (crow0 + row) * D + slot * 8), cs);
        unpack16<T>(ld16(sinp + (crow0 + row) * D + slot * 8), sn);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          o[e] = round_through<T>(x[e] * cs[e]) + round_through<T>((slot < SLOTS / 2 ? xp[e] : -xp[e]) * sn[e]);
        v = pack16<T>(o);
      }
    }
    if (row0 + row < nrows) st16(dst + (int64_t)(row0 + row) * stride + slot * 8, v);
  }
}

template <typename T>
__device__ __forceinline__ void pack_c_to_b(const float* p, u32x4* out2) {
  // 16 C-layout registers of one 32-row sub-tile -> two B operands (registers 0..7, 8..15)
#pragma unroll
  for (int st = 0; st < 2; ++st)
    out2[st] = u32x4{pack2<T>(p[8 * st + 0], p[8 * st + 1]), pack2<T>(p[8 * st + 2], p[8 * st + 3]),
                     pack2<T>(p[8 * st + 4], p[8 * st + 5]), pack2<T>(p[8 * st + 6], p[8 * st + 7])};
}

// A resident MFMA operand multiplied by a constant and rounded again (once per workgroup): with the fragment of Q (forward,
// dQ kernel) or of K (dK/dV kernel) carrying scale*log2(e), S leaves the matrix pipe in the exp2 domain, and with the
// accumulator chain STARTED from -m (the forward's running row maximum) or -lse*log2(e) (the backward kernels) -- the
// MFMA's srcC operand, free -- the softmax numerator is ONE instruction per element, exp2(S''), instead of fma + exp2.
// The attention loops are bound by instruction issue (4-5 instructions hide beside an MFMA), so an
// instruction less per element is time.  Cost: one more rounding of the operand to the storage dtype (relative 2^-9 in
// bf16; the reference's own bf16 path rounds S itself to bf16).
// `apply` (wave-uniform): false when the producer delivered the operand pre-scaled (AttnArgs::q_prescaled)
template <typename T, bool IN_AGPR = false>  // IN_AGPR: the one-wave-per-SIMD kernels keep their resident operands in AGPRs
__device__ __forceinline__ u32x4 scale_frag(u32x4 f, float c, bool apply = true) {
  u32x4 r = f;
  if (apply) {
    float x[8];
    unpack16<T>(f, x);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] *= c;
    r = pack16<T>(x);
  }
#if defined(__HIP_DEVICE_COMPILE__)
  // one 128-bit value from here on, born in the register file its MFMAs read it from (left as four dwords, or pinned in
  // VGPRs, the dK/dV kernel's tuples were re-assembled by 28 v_accvgpr_mov per tile in front of the MFMAs)
  if (IN_AGPR)
    asm volatile("" : "+a"(r));
  else
    asm volatile("" : "+v"(r));
#endif
  return r;
}
// 16 equal accumulator registers (a per-lane constant as the srcC tuple of the first MFMA of a chain); opaque to the
// compiler so that it stays ONE resident tuple instead of 16 moves in front of every chain
__device__ __forceinline__ f32x16 splat16(float v) {
  f32x16 t;
#pragma unroll
  for (int r = 0; r < 16; ++r) t[r] = v;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(t));
#endif
  return t;
}

// first visible key of query row `qrow` (0 when sequences are not packed) and its wave-wide max / min
__device__ __forceinline__ int packed_klo(const AttnArgs& a, int b, int qrow) {
  return (a.q_start != nullptr && qrow < a.seq_q) ? a.q_start[(int64_t)b * a.seq_q + qrow] : 0;
}

constexpr int kKVB = 128;  // keys per workgroup in the dK/dV kernel (32 per wave)
constexpr int kQT = 64;    // query rows per Q/dO tile

// dK/dV launch (attention_bwd_dkdv.hip); dtype / head_dim already validated by the caller
int attn_bwd_dkdv_launch(const AttnBwdArgs& g, int dtype, int head_dim, bool causal, hipStream_t s);

}  // namespace tamd
