// attention_fwd64.hip -- forward attention with 64 query rows per wave (one wave per SIMD, the whole register file).
// DIAGNOSTIC LIBRARY ONLY (build.py DIAG_SOURCES; selected with tamd_attn_set_fwd64, include/tamd_diag.h): measured level
// with the product kernel at the product shape, so not promoted -- see "Measured" below.
//
// The same arithmetic per query row, in the same order, as attn_fwd_kernel (attention_fwd_kernel.inc): outputs and LSE
// are bit-identical (tests/test_kernels.py::test_attention_fwd64_matches_fwd on the CPU model; tools/attn_fwd64_ab.py on
// the silicon).  Restructured so that the matrix pipe and the VALU of a SIMD are driven by ONE instruction stream that
// interleaves them instead of by two waves that take turns.
//
// A workgroup = 4 waves = 256 query rows of one (batch, head); a wave carries two 32-row query blocks A and B.  Per
// 64-key tile t, two phases of 32 MFMAs:
//     X  S(t+1) = K(t+1) . Q^T  for both blocks -- every K fragment read from LDS once and used twice --
//        interleaved with the online softmax of S(t): block A complete, block B up to its first 16 keys;
//     Y  O += V(t)^T . P(t)     for both blocks -- every V fragment read once and used twice -- interleaved with the rest
//        of block B's exponentials (their P fragments are consumed from the 9th MFMA of the phase on);
// S is double-buffered in registers, K/V tiles sit in a 4-deep LDS ring (128 KiB) loaded three tiles ahead, and the loop is
// unrolled by four so that both S copies and the ring slot are statically addressed.  Register files are assigned by hand
// (tamd_device.h mfma32_s / mfma32_o: inline-asm MFMAs): S in VGPRs (the softmax reads it), O, Q and the K fragments in AGPRs -- with the builtin
// the compiler puts every accumulator of a >256-register kernel into AGPRs and spills (347 registers in the first try).
// Every softmax value is pinned behind "its" MFMA (an empty volatile asm on its result): without that the compiler
// sinks the arithmetic to where P is consumed.
// Scope: head_dim 128, no padding mask, no dropout, no packed sequences, seq_k % 64 == 0, K and V rows the same distance
// apart -- tamd_attn_fwd takes attn_fwd_kernel otherwise.
//
// Measured (MI355X, bf16, random data, five boxes; profiles/r03f .. r03m_attn_fwd64_ab.jsonl; TFLOP/s):
//                                   attn_fwd_kernel   fwd64 (best variant)
//     Llama-3-8B causal   8x4096        961 .. 1039       976 .. 1042      (+0 .. +3 %)
//     bidirectional       8x4096       1046 .. 1106      1092 .. 1175      (+4 .. +7 %)
//     causal 2x8192                    1030 .. 1083      1064 .. 1134      (+3 .. +5 %)
//     causal 16x2048, MHA 4x4096, 1088-token prompt:      -3 .. -17 %  (256-row workgroups: diagonal waste, fewer workgroups)
//   Ablations (wrong results, timing only): without the softmax arithmetic 1550-1730, without the tile loads +5-15 %,
//   without both 1600-1990 (80 % of the 2.5 PFLOP/s figure): the MFMA + LDS-read stream is fine, the loop is bound by
//   instruction ISSUE.  tools/probes/mfma_filler_probe.hip (profiles/r03k_mfma_filler_probe.jsonl): with one wave per SIMD
//   FOUR VALU instructions hide beside a 32x32x16 MFMA (five with the accumulator in AGPRs), every further one costs ~5
//   cycles; this loop has ~7.2 beside each MFMA (276 softmax VALU, 48 LDS reads, 17-35 waits, 8 LDS-DMA pieces + their M0
//   writes, ~40 SALU, ~20 pad nops per 64 MFMAs), for which the probe predicts ~1250 TFLOP/s before barriers.  WHERE the
//   softmax instructions sit does not matter once every gap is over budget: dense slices, a 3-stage pipeline of the
//   exponentials (no instruction reads a result of its own gap) and one value per gap over both phases measured within
//   3 % of each other (patches: profiles/r03i_attn_fwd64_pipe_uniform.patch, r03j_attn_fwd64_gen1.patch).  What is left is instruction COUNT:
//   the minimum for this arithmetic is ~256 VALU + 48 LDS reads per 64 MFMAs = 4.75 per gap -- MI355X_MICROARCH.md quotes
//   1.25-1.40 PFLOP/s for exactly that stream.  DESIGN.md section 7.
#include "attention_common.h"

namespace tamd {

constexpr int kQB64 = 256;  // query rows per workgroup (64 per wave)

// MFMA kind K (0: S = a.b, 1: S += a.b, 2: O += a.b) behind `s_waitcnt lgkmcnt(min(n, CAP))`, n foldable
template <typename T, int CAP, int K>
__device__ __forceinline__ void mfma_after_wait(int n, f32x16& d, const u32x4& a, const u32x4& b) {
  static_assert(CAP <= 8, "extend the chain");
#define TAMD_MW(N_)                                                     \
  if ((n >= CAP ? CAP : n) == N_) {                                     \
    if (K == 0) mfma32_s0_w<N_>((const T*)nullptr, d, a, b);            \
    else if (K == 1) mfma32_s_w<N_>((const T*)nullptr, d, a, b);        \
    else mfma32_o_w<N_>((const T*)nullptr, d, a, b);                    \
    return;                                                             \
  }
  TAMD_MW(8) TAMD_MW(7) TAMD_MW(6) TAMD_MW(5) TAMD_MW(4) TAMD_MW(3) TAMD_MW(2) TAMD_MW(1) TAMD_MW(0)
#undef TAMD_MW
}

// The kernel (second generation; the first -- dynamic ring slots, one barrier at the end of a tile, per-fragment waits, and
// its schedule variants -- is profiles/r03j_attn_fwd64_gen1.patch).  What the measurements of the first one said: without
// the softmax arithmetic the loop runs at 1550-1730 TFLOP/s, with it at 1090-1175, and WHERE its instructions sit changes
// nothing -- the loop is bound by instruction ISSUE, ~475 instructions beside the 64 MFMAs of a tile.  So this version
// removes instructions and exposed latency instead of moving them: the tile loop is unrolled by four so that the LDS ring slot is static (slot offsets ride in the ds_read
// immediates: no address adds); fragments are waited for in pairs; the barrier
// sits BETWEEN the phases, where the first V fragments of phase Y (requested behind the last K fragments of phase X)
// are already in flight, and the first K fragments of the next tile's phase X are requested behind the last V fragments
// of phase Y -- no phase starts by waiting for LDS; the eight pieces of tile t+3 go out in phase Y behind the MFMAs that
// have no softmax work.
constexpr int kG2NoDma = 1, kG2NoSm = 2;  // ablations (wrong results)
constexpr int kG2Split = 4;  // the V pieces of tile t+2 in phase X (odd gaps 1-7), the K pieces of tile t+3 in phase Y (odd gaps 25-31)
template <typename T, bool CAUSAL, int VAR>
__global__ __launch_bounds__(kAttnThreads, 1) void attn_fwd64_kernel(AttnArgs a) {
  constexpr bool DMA = (VAR & kG2NoDma) == 0, SM = (VAR & kG2NoSm) == 0, SPLIT = (VAR & kG2Split) != 0;
  constexpr int D = 128, ROWB = D * 2, TILEB = kKB * ROWB, KS = D / 16, DT = D / 32, OROWB = ROWB + 16;
  constexpr int NBUF = 4, LA = 3;
  TAMD_DYN_SMEM(smem);
  const int lane = threadIdx.x & 63;
  const int wave = wave_id_uniform();
  const int hi = lane >> 5, l31 = lane & 31;

  const int group = a.heads_q / a.heads_kv;
  const int nqt = (a.seq_q + kQB64 - 1) / kQB64;
  int b, h, qt;
  {
    const int bid = blockIdx.x;
    const int per_grp = nqt * group;
    int g, within;
    if (a.xcd_map) {
      const int xcd = bid & 7, j = bid >> 3;
      g = (j / per_grp) * 8 + xcd;
      within = j % per_grp;
    } else {
      g = bid / per_grp;
      within = bid % per_grp;
    }
    b = g / a.heads_kv;
    const int hkv_ = g % a.heads_kv;
    h = hkv_ * group + within % group;
    qt = nqt - 1 - within / group;  // heavy (late) causal tiles first
  }
  const int hkv = h / group;
  const int q0 = qt * kQB64;
  const int off = a.seq_k - a.seq_q;
  const T* Q = reinterpret_cast<const T*>(a.q) + (int64_t)b * a.qsb + (int64_t)h * a.qsh;
  const T* K = reinterpret_cast<const T*>(a.k) + (int64_t)b * a.ksb + (int64_t)hkv * a.ksh;
  const T* V = reinterpret_cast<const T*>(a.v) + (int64_t)b * a.vsb + (int64_t)hkv * a.vsh;

  const int qw0 = q0 + wave * 64;
  u32x4 qf[2][KS];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int qrow = qw0 + blk * 32 + l31;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      qf[blk][ks] = (qrow < a.seq_q) ? ld16(Q + (int64_t)qrow * a.qss + ks * 16 + hi * 8) : u32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) to_agpr(qf[blk][ks]);
  f32x16 oacc[2][DT];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[blk][dt][r] = 0.f;
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) to_agpr(oacc[blk][dt]);
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

  int kend = a.seq_k;
  if (CAUSAL) {
    const int lim = q0 + kQB64 + off;
    kend = lim < kend ? lim : kend;
    if (kend < 0) kend = 0;
  }
  const int nkt = (kend + kKB - 1) / kKB;
  int tw = nkt - 1;  // the last tile this wave computes (attn_fwd64_kernel)
  if (CAUSAL) {
    const int last = qw0 + 63 + off;
    const int twv = last < 0 ? -1 : last / kKB;
    tw = twv < tw ? twv : tw;
  }

  // LDS: the four K tiles of the ring, then the four V tiles (64 KiB each): every slot of a kind is within reach of the
  // 16-bit ds_read immediate from ONE per-lane address (absolute, the array base folded in)
  unsigned rowK[KS], trV[DT][2];
  {
    TileOffsets<D> toff;
    toff.init(lane);
    const unsigned lds0 = lds_base_u32(smem);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) rowK[ks] = lds0 + toff.row[ks];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) trV[dt][t2] = lds0 + (unsigned)(NBUF * TILEB) + toff.tr[dt][t2];
  }
  constexpr float kDeferThr = 6.f;
  TileFeed<D> feed;
  feed.init(a.kss, wave, lane);
  constexpr int NI = TileFeed<D>::NI, NP = 2 * NI;
  auto issue_piece = [&](int t, int slot, int n) __attribute__((always_inline)) {  // piece n of tile t into ring slot `slot`
    const unsigned k_off = (unsigned)slot * TILEB, v_off = (unsigned)(NBUF + slot) * TILEB;
    if (n < NI)
      feed.issue_one(K + (int64_t)t * kKB * a.kss, smem, k_off, wave, n);
    else
      feed.issue_one(V + (int64_t)t * kKB * a.vss, smem, v_off, wave, n - NI);
  };
  // K fragment i = (k-step i >> 1, sub-tile i & 1) / V fragment i = (16-key step i / DT, d-tile i % DT) of ring slot SL
  auto kreq = [&](auto slc, int i) __attribute__((always_inline)) -> u32x4 {
    constexpr int SL = decltype(slc)::value;
    return lds_read16_abs_agpr(rowK[i >> 1], SL * TILEB + (i & 1) * 32 * ROWB);
  };
  auto vreq = [&](auto slc, int i) __attribute__((always_inline)) -> u32x4 {
    constexpr int SL = decltype(slc)::value;
    const int dt = i % DT, j = i / DT;
    const int imm = SL * TILEB + ((j >> 1) * 32 + (j & 1) * 16) * ROWB;
    const u32x2 lo = lds_read8_tr16_abs(trV[dt][0], imm);
    const u32x2 h2 = lds_read8_tr16_abs(trV[dt][1], imm);
    return u32x4{lo[0], lo[1], h2[0], h2[1]};
  };

  // prologue: tiles 0 .. 2 (clamped) landed
  if (nkt > 0) {
#pragma unroll
    for (int tt = 0; tt < LA; ++tt) {
      const int tc = tt < nkt ? tt : nkt - 1;
#pragma unroll
      for (int n = 0; n < NP; ++n)
        if (!(SPLIT && tt == LA - 1 && n >= NI)) issue_piece(tc, tt, n);  // (SPLIT: V of tile 2 goes out in phase X of tile 0)
    }
  }
  wait_vmcnt<0>();
  raw_barrier();

  f32x16 s0[2][2], s1[2][2];
  u32x4 pf[2][4];
  if (!SM) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int j = 0; j < 4; ++j) pf[blk][j] = u32x4{0, 0, 0, 0};
  }
  constexpr int NF = 2 * KS, NV = DT * 4, AH = 4;  // fragments per tile; requested ahead
  u32x4 kr[AH + 1], vr[AH + 1];                    // fragment rings (K in AGPRs, V in VGPRs), carried across the phases

  auto qk_step = [&](f32x16 (&sn)[2][2], const u32x4& kfrag, int i, int blk, int nwait) __attribute__((always_inline)) {
    const int sub = i & 1, ks = i >> 1;
    if (nwait >= 0) {
      if (ks == 0)
        mfma_after_wait<T, 8, 0>(nwait, sn[blk][sub], kfrag, qf[blk][ks]);
      else
        mfma_after_wait<T, 8, 1>(nwait, sn[blk][sub], kfrag, qf[blk][ks]);
    } else if (ks == 0) {
      mfma32_s0<T>(sn[blk][sub], kfrag, qf[blk][ks]);
    } else {
      mfma32_s<T>(sn[blk][sub], kfrag, qf[blk][ks]);
    }
  };

  // one key tile in ring slot SL = t & 3: sc = S(t), sn = S(t+1); on entry the first AH fragments of K(t+1) are requested
  auto tile = [&](auto slc, int t, f32x16 (&sc)[2][2], f32x16 (&sn)[2][2]) __attribute__((always_inline)) {
    constexpr int SL = decltype(slc)::value;
    const int tp = (t + LA < nkt) ? t + LA : nkt - 1;  // the tile loaded during this one (clamped: attn_fwd64_kernel)
    const int tpv = (t + 2 < nkt) ? t + 2 : nkt - 1;   // SPLIT: the tile whose V pieces go out in phase X
    const int kt0 = t * kKB;
    if (SM && CAUSAL && (kt0 + kKB - 1 > qw0 + off)) {  // diagonal tile (wave-uniform)
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const int lim = qw0 + blk * 32 + l31 + off - kt0 - 4 * hi;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            sc[blk][sub][r] = (sub * 32 + (r & 3) + 8 * (r >> 2) <= lim) ? sc[blk][sub][r] : -INFINITY;
      }
    }
    float mx[2] = {0.f, 0.f}, mref[2] = {0.f, 0.f}, psum[2] = {0.f, 0.f};
    auto sm_max = [&](int blk, int sub) __attribute__((always_inline)) {  // (two chains; the maximum is exact in any order)
      float m = sub == 0 ? sc[blk][0][0] : mx[blk], m2 = sc[blk][sub][8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        m = fmaxf(m, sc[blk][sub][r]);
        m2 = fmaxf(m2, sc[blk][sub][8 + r]);
      }
      m = fmaxf(m, m2);
      pin_here(m);
      mx[blk] = m;
    };
    auto sm_fin = [&](int blk) __attribute__((always_inline)) {
      const float m = fmaxf(mx[blk], swap32_f32(mx[blk]));
      const float m_tile = m * a.scale_log2;
      if (ballot64(m_tile - m_run[blk] > kDeferThr) != 0ull) {
        const float m_new = fmaxf(m_run[blk], m_tile);
        const float alpha = (m_new == -INFINITY) ? 1.f : fast_exp2(m_run[blk] - m_new);
        m_run[blk] = m_new;
        l_run[blk] *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) agpr_scale(oacc[blk][dt], alpha);
      }
      mref[blk] = (m_run[blk] == -INFINITY) ? 0.f : m_run[blk];
    };
    auto sm_exp = [&](int blk, int q) __attribute__((always_inline)) {
      const int sub = q >> 3, r = 2 * (q & 7);
      const float p0 = fast_exp2(__builtin_fmaf(sc[blk][sub][r], a.scale_log2, -mref[blk]));
      const float p1 = fast_exp2(__builtin_fmaf(sc[blk][sub][r + 1], a.scale_log2, -mref[blk]));
      psum[blk] += p0;
      psum[blk] += p1;
      unsigned w = pack2<T>(p0, p1);
      pin_here(w, psum[blk]);
      pf[blk][sub * 2 + (r >> 3)][(r & 7) >> 1] = w;
      if (q == 15) l_run[blk] += psum[blk];
    };
    // ---- phase X: S(t+1) from K(t+1) in slot SL+1; softmax slices as in attn_fwd64_kernel; the last four steps request
    // the first V(t) fragments.  Fragments are waited for in pairs (at even i: i and i+1).
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      if (i + AH < NF) kr[(i + AH) % (AH + 1)] = kreq(IntC<(SL + 1) & 3>{}, i + AH);
      if (i >= NF - AH) vr[i - (NF - AH)] = vreq(IntC<SL>{}, i - (NF - AH));
      const int last = i + AH < NF - 1 ? i + AH : NF - 1;  // the latest K fragment requested so far
      const int after = (last - (i + 1) > 0 ? last - (i + 1) : 0) + (i >= NF - AH ? 2 * (i - (NF - AH) + 1) : 0);
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const int m = 2 * i + blk;
        qk_step(sn, kr[i % (AH + 1)], i, blk, (blk == 0 && (i & 1) == 0) ? after : -1);
        if (SM) {
          if (m < 2) sm_max(0, m);
          else if (m == 2) sm_fin(0);
          else if (m < 19) sm_exp(0, m - 3);
          else if (m < 21) sm_max(1, m - 19);
          else if (m == 21) sm_fin(1);
          else if (m < 26) sm_exp(1, m - 22);
        }
        if (DMA && SPLIT && m <= 7 && (m & 1)) issue_piece(tpv, (SL + 2) & 3, NI + (m >> 1));
        sched_fence();
      }
    }
    // ---- between the phases: tile t+2 has landed (its pieces went out during tile t-1); every wave is done with slot SL-1
    // (SPLIT: K of tile t+2 has landed; its V pieces, issued in this phase X, may stay in flight: they are read after the
    // next barrier between phases, which leaves only the then-newest four in flight)
    if (SPLIT)
      wait_vmcnt<NI>();
    else
      wait_vmcnt<0>();
    raw_barrier();
    // ---- phase Y: O += V(t)^T P(t); the rest of block B's exponentials behind MFMAs 0-11; the pieces of tile t+3 into
    // slot SL-1 behind the odd MFMAs 13 .. 27; the last four steps request the first K(t+2) fragments
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (i + AH < NV) vr[(i + AH) % (AH + 1)] = vreq(IntC<SL>{}, i + AH);
      if (i >= NV - AH) kr[i - (NV - AH)] = kreq(IntC<(SL + 2) & 3>{}, i - (NV - AH));
      const int last = i + AH < NV - 1 ? i + AH : NV - 1;
      const int after = 2 * (last - (i + 1) > 0 ? last - (i + 1) : 0) + (i >= NV - AH ? i - (NV - AH) + 1 : 0);
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const int m = 2 * i + blk;
        if (blk == 0 && (i & 1) == 0)
          mfma_after_wait<T, 8, 2>(after, oacc[blk][i % DT], vr[i % (AH + 1)], pf[blk][i / DT]);
        else
          mfma32_o<T>(oacc[blk][i % DT], vr[i % (AH + 1)], pf[blk][i / DT]);
        if (SM && m < 12) sm_exp(1, m + 4);
        if (DMA && !SPLIT && m >= 13 && m <= 27 && (m & 1)) issue_piece(tp, (SL + 3) & 3, (m - 13) >> 1);
        if (DMA && SPLIT && m >= 25 && (m & 1)) issue_piece(tp, (SL + 3) & 3, (m - 25) >> 1);
        sched_fence();
      }
    }
  };

  if (tw >= 0) {  // S(0) from slot 0, then the first K(1) fragments for phase X of tile 0
    u32x4 k0[AH + 1];
#pragma unroll
    for (int i = 0; i < AH; ++i) k0[i] = kreq(IntC<0>{}, i);
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      if (i + AH < NF) k0[(i + AH) % (AH + 1)] = kreq(IntC<0>{}, i + AH);
      constexpr_wait_frag_agpr<AH>(NF - 1 - i, k0[i % (AH + 1)]);
      qk_step(s0, k0[i % (AH + 1)], i, 0, -1);
      qk_step(s0, k0[i % (AH + 1)], i, 1, -1);
    }
#pragma unroll
    for (int i = 0; i < AH; ++i) kr[i] = kreq(IntC<1>{}, i);
    nop_states<16>();  // (S(0) is read by the VALU a few instructions into tile 0)
  }
  for (int t = 0; t <= tw; t += 4) {
    tile(IntC<0>{}, t, s0, s1);
    if (t + 1 <= tw) tile(IntC<1>{}, t + 1, s1, s0);
    if (t + 2 <= tw) tile(IntC<2>{}, t + 2, s0, s1);
    if (t + 3 <= tw) tile(IntC<3>{}, t + 3, s1, s0);
  }
  // past its last tile the wave only loads, in the sequence of a computing wave: (SPLIT: the V pieces of tile t+2,) wait
  // for its pieces of tile t+2, barrier, the pieces of tile t+3 (SPLIT: their K half)
  for (int t = tw + 1; t < nkt; ++t) {
    const int tp = (t + LA < nkt) ? t + LA : nkt - 1;
    if (DMA && SPLIT) {
      const int tpv = (t + 2 < nkt) ? t + 2 : nkt - 1;
#pragma unroll
      for (int n = NI; n < NP; ++n) issue_piece(tpv, (t + 2) & 3, n);
      wait_vmcnt<NI>();
    } else {
      wait_vmcnt<0>();
    }
    raw_barrier();
    if (DMA) {
#pragma unroll
      for (int n = 0; n < (SPLIT ? NI : NP); ++n) issue_piece(tp, (t + LA) & 3, n);
    }
  }
  wait_vmcnt<0>();  // (nothing may land in the ring once it holds the O tiles; the K fragments requested last are dropped)
  wait_lgkmcnt0();
  raw_barrier();
  nop_states<16>();
  // ---- finalise both blocks: normalise, LSE, stage O through LDS, row-wise stores
  T* O = reinterpret_cast<T*>(a.o) + (int64_t)b * a.osb + (int64_t)h * a.osh;
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int qb0 = qw0 + blk * 32, qrow = qb0 + l31;
    float l = l_run[blk];
    l += swap32_f32(l);
    const float inv_l = (l > 0.f) ? 1.f / l : 0.f;
    if (a.lse != nullptr && hi == 0 && qrow < a.seq_q) {
      const float lse = (l > 0.f) ? (m_run[blk] + fast_log2(l)) * 0.69314718055994530942f : INFINITY;
      a.lse[((int64_t)b * a.heads_q + h) * a.seq_q + qrow] = lse;
    }
    const unsigned st_off = (unsigned)(wave * 2 + blk) * (32u * OROWB);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int d0 = dt * 32 + 8 * qd + 4 * hi;
        const u32x2 pk = {pack2<T>(oacc[blk][dt][qd * 4 + 0] * inv_l, oacc[blk][dt][qd * 4 + 1] * inv_l),
                          pack2<T>(oacc[blk][dt][qd * 4 + 2] * inv_l, oacc[blk][dt][qd * 4 + 3] * inv_l)};
        lds_write8(smem, st_off + (unsigned)l31 * OROWB + (unsigned)d0 * 2u, pk);
      }
    wave_lockstep_point();
    constexpr int SLOTS = ROWB / 16, RPI = 64 / SLOTS;
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) {
      const int row = it * RPI + lane / SLOTS, slot = lane % SLOTS;
      const int qr = qb0 + row;
      const u32x4 v = lds_read16(smem, st_off + (unsigned)row * OROWB + (unsigned)slot * 16u);
      if (qr < a.seq_q) st16(O + (int64_t)qr * a.oss + slot * 8, v);
    }
  }
}

template <typename T, int VAR>
static int fwd64_launch_t(const AttnArgs& a, bool causal, hipStream_t s) {
  const int nqt64 = (a.seq_q + kQB64 - 1) / kQB64;
  dim3 grid((unsigned)(nqt64 * a.heads_q * a.batch)), block(kAttnThreads);
  const size_t smem = (size_t)4 * 2 * kKB * 128 * 2;
  if (causal)
    hipLaunchKernelGGL((attn_fwd64_kernel<T, true, VAR>), grid, block, smem, s, a);
  else
    hipLaunchKernelGGL((attn_fwd64_kernel<T, false, VAR>), grid, block, smem, s, a);
  return launch_status();
}

bool attn_fwd64_applies(const AttnArgs& a, int head_dim) {
  return head_dim == 128 && a.key_valid == nullptr && a.drop_thr == 0 && a.q_start == nullptr && a.seq_k % kKB == 0 &&
         a.kss == a.vss && TileFeed<128>::usable(a.kss);
}

// `variant` (tamd_attn_set_fwd64): 1 the kernel; 2 with the LDS-DMA split over the phases; 5 / 6 / 7 ablations (WRONG
// results, timing only): no tile loads / no softmax arithmetic / neither.  fp16: variant 1 only.
int attn_fwd64_launch(const AttnArgs& a, bool causal, int dtype, int variant, hipStream_t s) {
  if (variant > 1 && dtype == TAMD_BF16) {
    switch (variant) {
      case 2: return fwd64_launch_t<bf16_t, kG2Split>(a, causal, s);
      case 5: return fwd64_launch_t<bf16_t, kG2NoDma>(a, causal, s);
      case 6: return fwd64_launch_t<bf16_t, kG2NoSm>(a, causal, s);
      case 7: return fwd64_launch_t<bf16_t, kG2NoDma | kG2NoSm>(a, causal, s);
      default: return TAMD_E_ARG;
    }
  }
  TAMD_DISPATCH_HALF(dtype, return (fwd64_launch_t<T, 0>(a, causal, s)));
  return TAMD_E_DTYPE;
}

}  // namespace tamd
