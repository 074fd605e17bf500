// attention_bwd_dkdv.hip -- dK / dV of the flash attention backward (kernel 3 of attention_bwd.inc's plan).
//
// Contract (SURVEY.md section 8a "Backward contract"; the reference differentiates eager_attention_forward,
// models/llama/modeling_llama.py:191-213, by autograd):
//   P = exp(scale*S - lse);  dV = P^T dO;  dP = dO V^T;  dS = P * (dP - delta);  dK = scale * dS^T Q
//   (GQA: summed over the query heads of the group; dropout: P and dP are multiplied by keep/(1-p)).
//
// One workgroup = 128 keys of one (batch, kv-head), 4 waves x 32 keys, ONE wave per SIMD (dK/dV accumulators 128
// registers + K/V fragments of the wave's keys 64 registers, resident for the whole kernel).  A wave alone on its
// SIMD has nobody to hide its latencies behind and only ~5 free issue slots per MFMA, so the loop over (64-row
// Q/dO tile, query head of the group) is built as a software pipeline and trimmed for instruction count:
//   * tiles are walked from the LAST q-tile down, heads innermost: all workgroups of a (batch, kv-head) group then
//     request the same Q/dO tile at about the same time and 31 of 32 CUs of the XCD hit in L2 (walking up from each
//     workgroup's own first visible tile streamed Q/dO from MALL/HBM 16 times over: 8.8 GB per launch);
//   * Q/dO tiles and the lse*log2e / delta rows stream global -> LDS by LDS-DMA only (16-byte pieces for the tiles,
//     4-byte pieces for the statistics), two tiles ahead of the compute, one barrier per tile.  (Fetching the
//     statistics with ordinary loads made the s_waitcnt for their data drain every LDS-DMA piece issued before
//     them -- vmcnt retires in order -- so no tile load ever overlapped compute.)
//   * the 64 MFMAs of a tile are issued as 16 groups of 4; the LDS fragments of group n+1 are requested before
//     the MFMAs of group n issue, into the other half of a two-group register ring;
//   * every LDS read of the loop is issued untracked (tamd_device.h) and tied to counted s_waitcnt lgkmcnt(N)
//     through register dependencies: the compiler drains vmcnt in front of every ds_read_b64_tr_b16 it can see
//     while LDS-DMA is in flight, and its own lgkmcnt(0) for one tracked read would drain the reads issued ahead;
//   * the S and dP accumulator chains START from the statistics: K is multiplied by scale*log2(e) once per workgroup
//     (scale_frag, attention_common.h) and the dQ kernel stores -lse*log2(e) and -delta, whose LDS rows are read (16
//     bytes = 4 consecutive query rows = 4 C-layout registers) straight into the accumulator tuples; the MFMAs then
//     deliver S'' = log2 p and dP - delta, and an element costs exp2 + mul + its share of two conversions: 3 VALU
//     instructions instead of 5 (the dropout variants fold the lse only: (dP*keep - delta) is not linear in dP);
//   * softmax-backward arithmetic of sub-tile s (4 chunks of 4 query rows) is interleaved with the MFMAs of the
//     next phase (sched_group_barrier), branch-free (the mask test costs 2 VALU per element on every tile, a
//     branch would cut the interleave), packed-f32 VALU off (-fno-slp-vectorize: an anti-lever beside MFMAs);
//   * the loop is unrolled over the two LDS buffers so every LDS offset is an immediate; per-lane global offsets of
//     the LDS-DMA pieces are loop invariants; (head, tile) counters instead of divisions.
// Measured on MI355X at the Llama-3-8B shape (tools/attn_bench.py, tools/attn_dkdv_dbg.py): whole backward 5.98 ->
// 4.16 ms (460 -> 661 TFLOP/s); ablations of the final kernel: -0.57 ms without the softmax arithmetic, -0.44
// without LDS fragment reads, -0.61 without MFMA: issue-slot-bound, not pipe-bound.
// S = Q.K^T is computed un-swapped so a lane owns a key column: the P and dS registers are directly the MFMA B
// operands of dV^T[d][key] += dO^T[d][q].P[q][key] and dK^T[d][key] += Q^T[d][q].dS[q][key].
#include "attention_common.h"

namespace tamd {

// s_waitcnt lgkmcnt(N) that the four fragments depend on: MFMAs consuming them cannot be scheduled above it
template <int N>
__device__ __forceinline__ void wait_frags(u32x4& f0, u32x4& f1, u32x4& f2, u32x4& f3) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "n"(N) : "memory");
#else
  (void)f0;
  (void)f1;
  (void)f2;
  (void)f3;
#endif
}

// DBG (diagnostic instantiations, built only with -DTAMD_DIAG into libtamd_diag.so; TAMD_DKDV_DBG=n, wrong results): 1 no softmax arithmetic, 2 no LDS fragment reads,
// 4 no MFMA, 8 no tile loads after the prologue, 16 no barrier; 32 (correct results) every tile through the loop body WITH the mask test
// order-only dependency: the registers are "produced" here, after every earlier volatile asm (the waits)
__device__ __forceinline__ void after_wait(u32x4& x0, u32x4& x1) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(x0), "+v"(x1)::"memory");
#else
  (void)x0;
  (void)x1;
#endif
}

// PACKED (include/tamd.h q_start plane 1): key k is seen by queries up to k_end[k], the last token of its sequence --
// a per-lane scalar here (a lane owns a key), so the packed mask is one more compare per element and q-tiles past
// the block's last sequence are never visited.  A separate instantiation keeps it out of the common path.
__device__ __forceinline__ void after_wait1(u32x4& x0) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(x0)::"memory");
#else
  (void)x0;
#endif
}
// ... for two accumulator tuples filled by untracked LDS reads (the statistics rows that start the S / dP chains)
__device__ __forceinline__ void after_wait_acc(f32x16& x0, f32x16& x1) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(x0), "+v"(x1)::"memory");
#else
  (void)x0;
  (void)x1;
#endif
}

template <typename T, int D, bool CAUSAL, bool HAS_MASK, bool DROP, int DBG = 0, bool PACKED = false>
__global__ __launch_bounds__(kAttnThreads, 1) void attn_bwd_dkdv_kernel(AttnBwdArgs g, int nkvt) {
  const AttnArgs& a = g.f;
  constexpr int ROWB = D * 2, TILEB = kQT * ROWB, KS = D / 16, DT = D / 32, OROWB = ROWB + 16;
  constexpr int BUFB = 2 * TILEB + 2 * kQT * 4;  // Q tile + dO tile + lse2[64] + delta[64]
  constexpr int NGA = KS / 2;                    // MFMA groups of one S/dP sub-tile (2 k-steps x {S, dP} each)
  constexpr int NGC = DT;                        // MFMA groups of one half of dV/dK (2 d-tiles x {dV, dK} each)
  static_assert(KS % 2 == 0 && DT % 2 == 0, "head_dim must be a multiple of 64");
  TAMD_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id_uniform();
  const int hi = lane >> 5, l31 = lane & 31;
  const int group = a.heads_q / a.heads_kv;
  int b, hkv, kvt;
  {
    const int bid = blockIdx.x;
    int gi, within;
    if (a.xcd_map) {
      const int xcd = bid & 7, j = bid >> 3;
      gi = (j / nkvt) * 8 + xcd;
      within = j % nkvt;
    } else {
      gi = bid / nkvt;
      within = bid % nkvt;
    }
    b = gi / a.heads_kv;
    hkv = gi % a.heads_kv;
    kvt = within;  // early key blocks see the most queries under a causal mask: they come first
    if (PACKED && nkvt <= 64) {
      // ... which is no longer true under bound planes: the blocks at the start of EVERY sequence / chunk are the long ones, and
      // with the natural order a long block of the last (batch, head) group of an XCD starts when the others are done (list
      // scheduling of 8 groups x 32 blocks on 32 CUs: 1.10x the balanced time for two chunks of 2048, 1.00x for the plain
      // causal mask).  Longest first inside the group: block of rank `within` in descending number of visible q-tiles (ties:
      // the earlier block) -- every wave derives the same permutation from the k_end plane (nkvt uniform loads).
      const int* ke = a.q_start + (int64_t)a.batch * a.seq_q + (int64_t)b * a.seq_k;
      auto visible_tiles = [&](int j) {
        const int last = j * kKVB + kKVB - 1 < a.seq_k ? j * kKVB + kKVB - 1 : a.seq_k - 1;
        return ke[last] / kQT - (j * kKVB) / kQT;
      };
      const int j = lane < nkvt ? lane : nkvt - 1;
      const int wj = visible_tiles(j);
      int rank = 0;
      for (int i0 = 0; i0 < nkvt; i0 += 8) {  // eight independent loads per trip (one wait, not eight)
        int wi[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wi[u] = visible_tiles(i0 + u < nkvt ? i0 + u : nkvt - 1);
#pragma unroll
        for (int u = 0; u < 8; ++u) rank += (i0 + u < nkvt && (wi[u] > wj || (wi[u] == wj && i0 + u < j))) ? 1 : 0;
      }
      const unsigned long long hit = ballot64(lane < nkvt && rank == within);
      kvt = hit != 0ull ? (int)__builtin_ctzll(hit) : within;
    }
  }
  const int k0 = kvt * kKVB, kw0 = k0 + wave * 32, krow = kw0 + l31;
  const int off = a.seq_k - a.seq_q;
  const T* K = reinterpret_cast<const T*>(a.k) + (int64_t)b * a.ksb + (int64_t)hkv * a.ksh;
  const T* V = reinterpret_cast<const T*>(a.v) + (int64_t)b * a.vsb + (int64_t)hkv * a.vsh;

  // K / V fragments of this lane's key (MFMA B operands), resident in registers for the whole kernel
  u32x4 kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const bool ok = krow < a.seq_k;
    kf[ks] = ok ? ld16(K + (int64_t)krow * a.kss + ks * 16 + hi * 8) : u32x4{0, 0, 0, 0};
    vf[ks] = ok ? ld16(V + (int64_t)krow * a.vss + ks * 16 + hi * 8) : u32x4{0, 0, 0, 0};
  }
  if (!(DBG & 1)) {
#pragma unroll
    // S leaves the MFMAs in the exp2 domain: the factor rides on K here, unless the Q tiles already carry it (q_prescaled)
    for (int ks = 0; ks < KS; ++ks) kf[ks] = scale_frag<T, true>(kf[ks], a.scale_log2, !a.q_prescaled);
  }
  constexpr bool FOLD_DELTA = !DROP;  // dP's chain starts from -delta (dropout: (dP*keep - delta) is not linear in dP)
  bool key_ok = krow < a.seq_k;
  if (HAS_MASK && a.key_valid != nullptr)
    key_ok = key_ok && a.key_valid[(int64_t)b * a.seq_k + (krow < a.seq_k ? krow : 0)] != 0;
  const AttnDrop drop = attn_drop_ctx(a);
  // lane part of a block index: this lane's key pair + the 2 * hi query pairs its rows sit above the chunk's first one
  // (rows / keys past the end index blocks of other rows: their probabilities are zero anyway)
  const unsigned long long drop_lane = (unsigned long long)(krow >> 1) + (unsigned long long)(2 * hi) * drop.csk;
  const bool drop_kodd = (krow & 1) != 0;  // this lane's key: the first or the second word of each hash
  // ... with the query pair this lane hashes for its key pair: the chunk's first (even key) or second (odd key)
  const unsigned long long drop_lane_j = drop_lane + (drop_kodd ? drop.csk : 0ull);
  // PACKED: last query that sees this lane's key, and (wave-uniform) the last one that sees any key of the block
  const int* k_end = PACKED ? a.q_start + (int64_t)a.batch * a.seq_q + (int64_t)b * a.seq_k : nullptr;
  const int kend = PACKED ? k_end[krow < a.seq_k ? krow : a.seq_k - 1] : 0x3fffffff;
  const int kend_blk = PACKED ? k_end[k0 + kKVB - 1 < a.seq_k ? k0 + kKVB - 1 : a.seq_k - 1] : 0x3fffffff;

  TileOffsets<D> toff;
  toff.init(lane);
  f32x16 dkacc[DT], dvacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dkacc[dt][r] = 0.f;
      dvacc[dt][r] = 0.f;
    }

  // query tiles that can see any key of this block
  int qt_first = 0;
  if (CAUSAL) {
    const int qmin = k0 - off;  // first query row that sees key k0
    qt_first = qmin > 0 ? qmin / kQT : 0;
  }
  int nqt64 = (a.seq_q + kQT - 1) / kQT;
  if (PACKED) {  // q-tiles after the last query that can see a key of this block are never visited
    const int lastq = kend_blk - off;
    const int hi_tiles = lastq >= 0 ? lastq / kQT + 1 : 0;
    nqt64 = hi_tiles < nqt64 ? hi_tiles : nqt64;
  }
  const int per_head = nqt64 > qt_first ? nqt64 - qt_first : 0;
  const int niter = per_head * group;

  // loop-invariant byte offsets of this lane inside a [64][D] global tile, one per LDS-DMA piece of this wave (the
  // generic issue_kv_tile recomputes rows, swizzles and 64-bit products per piece and per tile: ~150 instructions
  // a tile, all of them exposed on a one-wave-per-SIMD kernel)
  constexpr int SLOTS = ROWB / 16, RPI = 1024 / ROWB, NI = (kQT * ROWB) / 1024 / 4;
  const int lane_row = lane / SLOTS;
  unsigned pq[NI], po[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int r = (wave * NI + i) * RPI + lane_row;
    const unsigned sl = (unsigned)((lane % SLOTS) ^ row_swz<D>(r)) * 16u;
    pq[i] = (unsigned)r * (unsigned)a.qss * 2u + sl;
    po[i] = (unsigned)r * (unsigned)a.oss * 2u + sl;
  }
  // Order of the tiles: q-tiles from the LAST one down to the first visible one, the query heads of the group innermost.
  // Every workgroup of a (batch, kv-head) group then walks the same (tile, head) sequence from the same starting point, so
  // the 32 CUs of an XCD request a Q/dO tile at about the same time and 31 of them hit in L2; walking up from each
  // workgroup's own first visible tile (the previous order) streamed the tiles from MALL/HBM 16 times over: 8.8 GB per
  // launch, a 1.05 ms floor at the Llama-3-8B shape.
  // Round 5 -- the tile feed, rebuilt.  Until then the 9 LDS-DMA pieces of a tile went out in one clump behind the hand-off
  // barrier, from per-lane 64-bit pointers: ~150 instructions per tile (64-bit products of (batch, head, tile) per tile, the
  // lane offsets parked in AGPRs and copied back, a zero page select per piece) with no MFMA running beside them -- a tenth
  // of the kernel (profiles/r03d_dkdv_ablation.txt).  Now:
  //   * a tile's source is three wave-uniform 64-bit bases (Q, dO, the statistics rows) that STEP from tile to tile (one
  //     select + one 64-bit add each), pieces are buffer-addressed (base in SGPRs + a loop-invariant 32-bit lane offset);
  //   * the rows a tile does not have (the ragged last q-tile, the padding tile of an odd count) are cut off by the buffer's
  //     size -- the hardware range check delivers zeros (tamd_device.h glds16_buf_rng), which is what such rows must
  //     contribute: Q = dO = 0 and statistics 0 give S'' = 0, p = 1, dP - delta = 0, dS = 0 -- no branch, no zero page;
  //   * tile it+1 is requested DURING tile it, one piece behind each of its first nine MFMA groups (an LDS-DMA issue holds
  //     the wave 60-185 cycles: behind four queued MFMAs the matrix pipe works through that time), into the buffer the
  //     hand-off of tile it-1 released; it has landed at the hand-off of tile it (vmcnt(0) + barrier), like before.
  // (elements) the next head of the group: + step_*_h; past the group's last head also + wrap_*: back to its first head, one tile down
  const int64_t step_q_h = a.qsh, wrap_q = -(int64_t)kQT * a.qss - (int64_t)group * a.qsh;
  const int64_t step_o_h = a.osh, wrap_o = -(int64_t)kQT * a.oss - (int64_t)group * a.osh;
  const int64_t step_s_h = a.seq_q, wrap_s = -(int64_t)kQT - (int64_t)group * a.seq_q;
  int iss_h = 0, iss_qt = nqt64 - 1;  // (head, q-tile) of the next tile to issue: no division in the loop
  const T* iss_q = reinterpret_cast<const T*>(a.q) + (int64_t)b * a.qsb + (int64_t)(hkv * group) * a.qsh + (int64_t)iss_qt * kQT * a.qss;
  const T* iss_o = reinterpret_cast<const T*>(g.dout) + (int64_t)b * a.osb + (int64_t)(hkv * group) * a.osh + (int64_t)iss_qt * kQT * a.oss;
  const float* iss_s = (const float*)g.delta + ((int64_t)b * a.heads_q + hkv * group) * a.seq_q + (int64_t)iss_qt * kQT;  // -delta rows
  const int64_t lse_plane = (int64_t)a.batch * a.heads_q * a.seq_q;  // -lse*log2(e): the second plane (elements)
  const unsigned pstat = (unsigned)lane * 4u;
  // the tile being issued: its bases and how many of its 64 rows exist (0: a padding tile)
  const T* src_q = iss_q;
  const T* src_o = iss_o;
  const float* src_s = iss_s;
  int src_rows = 0;
  auto issue_begin = [&](int it) {  // (wave-uniform scalar work: once per tile)
    src_q = iss_q;
    src_o = iss_o;
    src_s = iss_s;
    const int left = a.seq_q - iss_qt * kQT;
    src_rows = it < niter ? (left < kQT ? left : kQT) : 0;
    const bool wrap = ++iss_h == group;
    if (wrap) {
      iss_h = 0;
      --iss_qt;
    }
    // (mask arithmetic, not a select between two kernel-argument values: hipcc turns that into a select between their
    // ADDRESSES -- a scratch copy of the steps and a flat load per tile)
    const int64_t wm = -(int64_t)wrap;
    iss_q += step_q_h + (wm & wrap_q);
    iss_o += step_o_h + (wm & wrap_o);
    iss_s += step_s_h + (wm & wrap_s);
  };
  constexpr int NPIECE = 2 * NI + 1;  // Q pieces, dO pieces, one piece of statistics
  auto issue_piece = [&](int buf, int n) {
    const unsigned q_off = (unsigned)buf * BUFB, do_off = q_off + TILEB, st_off = do_off + TILEB;
    if (n < NI)
      glds16_buf_rng(src_q, (unsigned)src_rows * (unsigned)a.qss * 2u, pq[n], smem, q_off + (unsigned)(wave * NI + n) * 1024u);
    else if (n < 2 * NI)
      glds16_buf_rng(src_o, (unsigned)src_rows * (unsigned)a.oss * 2u, po[n - NI], smem, do_off + (unsigned)(wave * NI + n - NI) * 1024u);
    else  // waves 0 / 2: -lse*log2(e) of the tile's 64 rows, waves 1 / 3: -delta (twice the same bytes: no branch); the plane
          // goes into the base, not into a scalar offset: that would count towards the range
      glds4_buf_rng(src_s + ((wave & 1) ? 0 : lse_plane), (unsigned)src_rows * 4u, pstat, 0u, smem,
                    st_off + (unsigned)(wave & 1) * (kQT * 4));
  };
  auto issue = [&](int it, int buf) {  // the whole tile at once (prologue)
    issue_begin(it);
#pragma unroll
    for (int n = 0; n < NPIECE; ++n) issue_piece(buf, n);
  };

  // absolute LDS addresses of this lane's fragment reads (untracked reads take base + 16-bit immediate)
  const unsigned lds0 = lds_base_u32(smem);
  unsigned rowaddr[KS], traddr[DT][2];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) rowaddr[ks] = lds0 + toff.row[ks];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    traddr[dt][0] = lds0 + toff.tr[dt][0];
    traddr[dt][1] = lds0 + toff.tr[dt][1];
  }
  const unsigned stataddr[2] = {lds0 + (unsigned)hi * 16u + (unsigned)(2 * TILEB),
                                lds0 + (unsigned)hi * 16u + (unsigned)(BUFB + 2 * TILEB)};
  auto mm = [&](u32x4 x, u32x4 y, f32x16 c) -> f32x16 {
    if (DBG & 4) {
      c[0] += u32_as_f32(x[0] ^ y[0]);
      return c;
    }
    return mfma32<T>(x, y, c);
  };
  u32x4 fa[4], fb[4];  // two-group fragment ring: even groups use fa, odd groups fb (16 groups per tile: even)
  // S/dP group ga (0..NGA-1) of sub-tile `sub`: fragments Q(ks0) dO(ks0) Q(ks0+1) dO(ks0+1), ks0 = 2*ga
  auto load_rows = [&](u32x4 (&f)[4], unsigned q_off, unsigned do_off, int sub, int ga) {
    if (DBG & 2) return;
    const int si = sub * 32 * ROWB;
    f[0] = lds_read16_abs(rowaddr[2 * ga], (int)q_off + si);
    f[1] = lds_read16_abs(rowaddr[2 * ga], (int)do_off + si);
    f[2] = lds_read16_abs(rowaddr[2 * ga + 1], (int)q_off + si);
    f[3] = lds_read16_abs(rowaddr[2 * ga + 1], (int)do_off + si);
  };
  // dV/dK group (d-tile pair dtp, MFMA step j): fragments dO^T(dt0,j) Q^T(dt0,j) dO^T(dt1,j) Q^T(dt1,j)
  // (the loop is unrolled over the two LDS buffers, so tile offsets are constants and ride in the offset field)
  auto tr_frag = [&](unsigned tile_off, int dt, int j) -> u32x4 {
    const int imm = ((j >> 1) * 32 + (j & 1) * 16) * ROWB + (int)tile_off;
    const u32x2 lo = lds_read8_tr16_abs(traddr[dt][0], imm);
    const u32x2 h2 = lds_read8_tr16_abs(traddr[dt][1], imm);
    return u32x4{lo[0], lo[1], h2[0], h2[1]};
  };
  auto load_tr = [&](u32x4 (&f)[4], unsigned q_off, unsigned do_off, int dtp, int j) {
    if (DBG & 2) return;
    f[0] = tr_frag(do_off, 2 * dtp, j);
    f[1] = tr_frag(q_off, 2 * dtp, j);
    f[2] = tr_frag(do_off, 2 * dtp + 1, j);
    f[3] = tr_frag(q_off, 2 * dtp + 1, j);
  };

  // the S / dP accumulators of sub-tile `sub` of the tile in LDS buffer `buf`, loaded with the negated statistics of their
  // 16 query rows (C layout: registers 4j .. 4j+3 = rows 8j + 4*hi .. +3 = one 16-byte read)
  auto load_stats = [&](f32x16& s_, f32x16& dp_, int buf, int sub) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32x4 l4 = {0u, 0u, 0u, 0u}, d4 = {0u, 0u, 0u, 0u};
      if (!(DBG & 1)) {
        l4 = lds_read16_abs(stataddr[buf], (sub * 32 + 8 * j) * 4);
        if (FOLD_DELTA) d4 = lds_read16_abs(stataddr[buf], (sub * 32 + 8 * j) * 4 + kQT * 4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s_[4 * j + e] = u32_as_f32(l4[e]);
        dp_[4 * j + e] = u32_as_f32(d4[e]);
      }
    }
  };
  constexpr int NSTAT = (DBG & 1) ? 0 : (FOLD_DELTA ? 8 : 4);  // LDS reads of one load_stats

  if (niter > 0) issue(0, 0);
  wait_vmcnt0();
  block_sync();
  // Straight-line loop body (no per-wave skip of fully masked tiles: the causal mask zeroes them, and the two
  // waves it concerns lose one tile per head; any divergent path through the body makes the compiler shuffle the
  // 128 accumulator registers at every back edge).  Group 0's fragments are requested one tile ahead, into fa.
  f32x16 sn, dpn;  // S / dP of sub-tile 0 of the NEXT tile: statistics requested one tile ahead, like group 0's fragments
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    sn[r] = 0.f;
    dpn[r] = 0.f;
  }
  // FINE (every variant without dropout): the tile loop as a hand-placed stream -- see tile_pair_fine below
  constexpr bool FINE = !DROP;
  constexpr int NGRP = 2 * NGA + 2 * NGC;  // MFMA groups of a tile
  // fragment j of group m of the tile in LDS buffer `buf` -> slot j of the ring half m & 1 (one b128 read, or two transposing
  // reads): groups 0 .. 2*NGA-1 are S / dP of the two sub-tiles (Q(ks0) dO(ks0) Q(ks0+1) dO(ks0+1)), the others dV / dK
  // (dO^T(dt0,j) Q^T(dt0,j) dO^T(dt1,j) Q^T(dt1,j))
  auto req = [&](int buf, int m, int j) {
    if (DBG & 2) return;
    const unsigned q_off = (unsigned)buf * BUFB, do_off = q_off + TILEB;
    u32x4(&f)[4] = (m & 1) ? fb : fa;
    if (m < 2 * NGA) {
      const int sub = m / NGA, ga = m % NGA;
      f[j] = lds_read16_abs(rowaddr[2 * ga + (j >> 1)], (int)((j & 1) ? do_off : q_off) + sub * 32 * ROWB);
    } else {
      const int c = m - 2 * NGA, half = c / NGC, gc = c % NGC;
      f[j] = tr_frag((j & 1) ? q_off : do_off, 2 * (gc >> 1) + (j >> 1), half * 2 + (gc & 1));
    }
  };
  if (niter > 0) {
    if (FINE) {  // what the tail of a tile leaves behind for the next one
      load_stats(sn, dpn, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) req(0, 0, j);
#pragma unroll
      for (int j = 0; j < 3; ++j) req(0, 1, j);
    } else {
      load_rows(fa, 0u, (unsigned)TILEB, 0, 0);
      load_stats(sn, dpn, 0, 0);
    }
  }
  const int niter2 = (niter + 1) & ~1;  // even: an odd count gets one all-zero padding tile
  int cmp_h = 0, cmp_qt = nqt64 - 1;    // (head, q-tile) of the tile being computed
  // Two copies of the loop body.  PLAIN: the tile needs no mask for any key of this wave (every key of the wave valid,
  // and, causal, the tile's first query row sees the wave's last key): no mask test -- 2 VALU per element, 64 of the
  // ~420 instructions beside the 64 MFMAs of a tile, on a loop that is bound by instruction issue (4-5 instructions hide
  // beside an MFMA: profiles/r03k_mfma_filler_probe.jsonl).  Tiles are walked from the last query tile down, so a
  // wave's plain tiles come first: two straight-line loops, no branch inside a body (the count is wave-uniform; waves
  // of one workgroup switch loops at different tiles, their barriers still pair up one per tile).
  auto tile_pair = [&](auto plain_c, int it0) __attribute__((always_inline)) {
   constexpr bool PLAIN = decltype(plain_c)::value != 0;
#pragma unroll
   for (int cur = 0; cur < 2; ++cur) {
    const int it = it0 + cur;
    const unsigned q_off = (unsigned)cur * BUFB, do_off = q_off + TILEB, st_off = do_off + TILEB;
    const unsigned nq_off = (unsigned)(cur ^ 1) * BUFB, ndo_off = nq_off + TILEB;
    const int qt0 = it < niter ? cmp_qt * kQT : nqt64 * kQT;
    const int hcur = hkv * group + (it < niter ? cmp_h : 0);
    if (++cmp_h == group) {
      cmp_h = 0;
      --cmp_qt;
    }
    // hand-off, placed before the last MFMA group of the tile: by then every fragment of this tile is in
    // registers, so the buffer can take tile it+2 at once; tile it+1 (requested one tile ago) must have landed
    auto hand_off = [&]() {
      wait_vmcnt0();
      wait_lgkmcnt0();
      if (!(DBG & 16)) raw_barrier();
      sched_fence();
    };
    // tile it+1 goes out behind the first NPIECE MFMA groups of this tile (the last tile of the loop requests nothing)
    // (no branch: behind the last tile of the loop issue_begin finds no rows, the pieces fetch nothing and zeros land in the
    // buffer nobody reads any more)
    if (!(DBG & 8)) issue_begin(it + 1);
    // ... strictly BEHIND the group's MFMAs (the scheduler otherwise puts the piece behind the first one: one queued MFMA to
    // cover a 60-185 cycle issue instead of four)
    auto feed_piece = [&](int gidx) {  // one piece behind each of the groups 0 .. NPIECE-1
      if (DBG & 8) return;
      sched_fence();
      if (gidx < NPIECE) issue_piece(cur ^ 1, gidx);
    };
    // key visible to local query row r of this tile iff mask_lim <= r (padding / out-of-range keys: never)
    const int mask_lim = key_ok ? (CAUSAL ? krow - (qt0 + off) : -0x40000000) : 0x40000000;
    const int mask_hi = kend - qt0;  // PACKED: last local query row of this tile that belongs to the key's sequence
    // dropout: block index of (this tile's head, its first query pair, key pair 0) -- wave-uniform
    const unsigned long long drop_tile = DROP ? drop.row_base((unsigned long long)b * a.heads_q + hcur, (unsigned long long)qt0) : 0ull;
    sched_fence();

    f32x16 s[2], dp[2];
    u32x4 pf[4], dsf[4];
    // softmax backward of 4 consecutive query rows (chunk qd) of sub-tile `sub`: C-layout registers qd*4 .. +3 of
    // P and dS, rounded and packed at once into B operand sub*2 + (qd>>1), dwords (qd&1)*2 .. +1
    // lse / delta of 4 consecutive query rows of chunk qd (16 bytes each; lane part of the address = 16*hi)
    // (only the dropout variants still read -delta beside the arithmetic; NSR = reads per chunk)
    constexpr int NSR = ((DBG & 1) || FOLD_DELTA) ? 0 : 1;
    auto stat_reads = [&](int sub, int qd, u32x4& d4) {
      if (NSR == 0) return;
      const int imm = (sub * 32 + 8 * qd) * 4;  // (st_off of the second buffer does not fit the 16-bit offset field)
      d4 = lds_read16_abs(stataddr[cur], imm + kQT * 4);
    };
    auto softmax_chunk = [&](int sub, int qd, const u32x4& d4) {
      if (DBG & 1) {
        const int op = sub * 2 + (qd >> 1), w = (qd & 1) * 2;
        pf[op][w] = f32_as_u32(s[sub][qd * 4]);
        pf[op][w + 1] = f32_as_u32(s[sub][qd * 4 + 1]);
        dsf[op][w] = f32_as_u32(dp[sub][qd * 4]);
        dsf[op][w + 1] = f32_as_u32(dp[sub][qd * 4 + 1]);
        return;
      }
      float p[4], ds[4];
      const int ql = sub * 32 + 8 * qd + 4 * hi;  // 4 consecutive query rows (r&3)
      bool keep4[4] = {true, true, true, true};
      if (DROP) {  // rows ql + 2j, ql + 2j + 1 share a hash; this lane's key picks the word, the row its half
        // block of query pair (qt0 + ql) / 2 + j: wave-uniform tile base and chunk term, the lane's part added once.
        // The block's two keys sit in neighbouring lanes (lane ^ 1), which need the same 64 bits: the even key's lane hashes
        // the chunk's first query pair (j = 0), the odd key's lane the second, and the words cross the pair through DPP moves
        // (round 6; the forward kernel has the transposed arrangement) -- one hash per chunk and lane instead of two.
        unsigned w0, w1;
        drop.words(drop_tile + (unsigned long long)(sub * 16 + 4 * qd) * drop.csk + drop_lane_j, w0, w1);
        // even key: its word of j = 0 is its own w0, of j = 1 the neighbour's w0; odd key: the neighbour's w1 and its own w1
        const unsigned x0 = pair_swap_u32(w0), x1 = pair_swap_u32(w1);
        const unsigned wa = drop_kodd ? x1 : w0;  // j = 0
        const unsigned wb = drop_kodd ? w1 : x0;  // j = 1
        keep4[0] = drop.kept_lo(wa);
        keep4[1] = drop.kept_hi(wa);
        keep4[2] = drop.kept_lo(wb);
        keep4[3] = drop.kept_hi(wb);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = qd * 4 + e;
        float x = s[sub][r];
        // branch-free: a branch here would cut the MFMA / VALU interleave into pieces; the test costs 2 VALU per
        // element on every tile (mask_lim is -2^30 on tiles that need no mask)
        if (PACKED && !PLAIN)
          x = (mask_lim <= ql + e && ql + e <= mask_hi) ? x : -INFINITY;
        else if (!PLAIN)
          x = (mask_lim <= ql + e) ? x : -INFINITY;
        const float pe = fast_exp2(x);  // x = S*scale*log2(e) - lse*log2(e): the chain started from -lse*log2(e)
        // dV uses the dropped probabilities; their factor 1 / (1 - p) multiplies the finished dV rows, dP's rides in the fma
        p[e] = (DROP && !keep4[e]) ? 0.f : pe;
        ds[e] = FOLD_DELTA ? pe * dp[sub][r]
                           : pe * fmaf((DROP && !keep4[e]) ? 0.f : dp[sub][r], drop.scale, u32_as_f32(d4[e]));  // (d4 = -delta)
      }
      const int op = sub * 2 + (qd >> 1), w = (qd & 1) * 2;
      pf[op][w] = pack2<T>(p[0], p[1]);
      pf[op][w + 1] = pack2<T>(p[2], p[3]);
      dsf[op][w] = pack2<T>(ds[0], ds[1]);
      dsf[op][w + 1] = pack2<T>(ds[2], ds[3]);
    };

    // one MFMA, then a slice of the chunk's arithmetic, four times: the wave has the SIMD to itself, so the VALU
    // work must sit inside the MFMA shadows of its own instruction stream
    auto mfma_valu_interleave = [&]() {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x402, 14, 0);  // VALU or TRANS, in the scheduler's own order (a fixed exp slot
                                                             // put every v_exp next to its consumer: 46 trans-use hazard NOPs per
                                                             // tile).  (This grouped body serves the dropout variants only since
                                                             // round 5 -- the others run tile_pair_fine -- and their ~46 hash +
                                                             // softmax instructions per group overflow any per-gap cap: "at most
                                                             // 4 per gap" measured 1.5 % behind "up to 14" at bert-base with
                                                             // dropout, profiles/r05n_attn_variants_ab.jsonl.)
      }
    };
    // ---- phases A0, A1: S and dP of the two 32-row sub-tiles; A1 carries the softmax backward of sub-tile 0
    s[0] = sn;  // (requested behind the previous tile's hand-off)
    dp[0] = dpn;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int ga = 0; ga < NGA; ++ga) {
        const int gidx = sub * NGA + ga;  // group index inside the tile (parity picks the ring half)
        u32x4(&fc)[4] = (gidx & 1) ? fb : fa;
        u32x4(&fn)[4] = (gidx & 1) ? fa : fb;
        constexpr int NCH_A = 4 / NGA;  // softmax chunks carried by one A1 group
        u32x4 d4[NCH_A];
        // every LDS read of the loop is untracked; issue order = completion order:
        //   (dropout: -delta of this group's chunks) | next group's fragments | (first group of the tile: the statistics
        //   that start sub-tile 1's chains) | wait: all but the newest batch
        if (sub == 1) {
#pragma unroll
          for (int c = 0; c < NCH_A; ++c) stat_reads(0, ga * NCH_A + c, d4[c]);
        }
        if (ga + 1 < NGA)
          load_rows(fn, q_off, do_off, sub, ga + 1);
        else if (sub == 0)
          load_rows(fn, q_off, do_off, 1, 0);
        else
          load_tr(fn, q_off, do_off, 0, 0);
        if (sub == 0 && ga == 0) load_stats(s[1], dp[1], cur, 1);
        sched_fence();
        if (sub == 1 && ga == NGA - 1)
          wait_frags<8>(fc[0], fc[1], fc[2], fc[3]);
        else if (sub == 0 && ga == 0)
          wait_frags<4 + NSTAT>(fc[0], fc[1], fc[2], fc[3]);  // (everything older has landed: sub-tile 0's statistics too)
        else
          wait_frags<4>(fc[0], fc[1], fc[2], fc[3]);
        if (sub == 0 && ga == 0) after_wait_acc(s[0], dp[0]);
        if (sub == 0 && ga == 1) after_wait_acc(s[1], dp[1]);  // (NGA >= 2: covered by this group's wait)
        if (sub == 1 && NSR) {
#pragma unroll
          for (int c = 0; c < NCH_A; ++c) after_wait1(d4[c]);
        }
        sched_fence();
        s[sub] = mm(fc[0], kf[2 * ga], s[sub]);
        dp[sub] = mm(fc[1], vf[2 * ga], dp[sub]);
        s[sub] = mm(fc[2], kf[2 * ga + 1], s[sub]);
        dp[sub] = mm(fc[3], vf[2 * ga + 1], dp[sub]);
        if (sub == 1) {  // softmax backward of sub-tile 0 in the shadow of these MFMAs (NGA = 2: two chunks)
#pragma unroll
          for (int c = 0; c < NCH_A; ++c) softmax_chunk(0, ga * NCH_A + c, d4[c]);
          mfma_valu_interleave();
        }
        feed_piece(gidx);
        sched_fence();
      }
    }
    // ---- phases C0, C1: dV += dO^T P, dK += Q^T dS over the two halves of the tile's query rows; C0 carries
    // the softmax backward of sub-tile 1; the hand-off sits before the last group
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int gc = 0; gc < NGC; ++gc) {
        const int gidx = 2 * NGA + half * NGC + gc;
        const int dtp = gc >> 1, j = half * 2 + (gc & 1);
        u32x4(&fc)[4] = (gidx & 1) ? fb : fa;
        u32x4(&fn)[4] = (gidx & 1) ? fa : fb;
        const bool last = (half == 1 && gc == NGC - 1);
        constexpr int NCH_C = 4 / NGC;
        u32x4 d4[NCH_C];
        if (half == 0) {
#pragma unroll
          for (int c = 0; c < NCH_C; ++c) stat_reads(1, gc * NCH_C + c, d4[c]);
        }
        if (last) {
          hand_off();  // waits lgkmcnt(0): this group's fragments are in registers
          load_rows(fn, nq_off, ndo_off, 0, 0);  // group 0 of tile it+1 (stale but harmless after the last tile)
          load_stats(sn, dpn, cur ^ 1, 0);       // ... and the statistics that start its first S / dP chains
          sched_fence();
          wait_frags<15>(fc[0], fc[1], fc[2], fc[3]);  // dependency only (nothing left to wait for)
        } else {
          const int gn = gc + 1 < NGC ? gc + 1 : 0, hn = gc + 1 < NGC ? half : 1;
          load_tr(fn, q_off, do_off, gn >> 1, hn * 2 + (gn & 1));
          sched_fence();
          wait_frags<8>(fc[0], fc[1], fc[2], fc[3]);  // all but the 8 reads just issued
        }
        if (half == 0 && NSR) {
#pragma unroll
          for (int c = 0; c < NCH_C; ++c) after_wait1(d4[c]);
        }
        sched_fence();
        dvacc[2 * dtp] = mm(fc[0], pf[j], dvacc[2 * dtp]);
        dkacc[2 * dtp] = mm(fc[1], dsf[j], dkacc[2 * dtp]);
        dvacc[2 * dtp + 1] = mm(fc[2], pf[j], dvacc[2 * dtp + 1]);
        dkacc[2 * dtp + 1] = mm(fc[3], dsf[j], dkacc[2 * dtp + 1]);
        if (half == 0) {  // softmax backward of sub-tile 1 in the shadow of these MFMAs
#pragma unroll
          for (int c = 0; c < NCH_C; ++c) softmax_chunk(1, gc * NCH_C + c, d4[c]);
          mfma_valu_interleave();
        }
        feed_piece(gidx);
        sched_fence();
      }
    }
   }
  };
  // ---- The tile loop as a hand-placed stream (round 5; every variant without dropout).
  // One wave per SIMD issues in order: while it queues four MFMAs back to back nothing else of it issues, and whatever stands
  // between two groups of MFMAs runs with the matrix pipe idle -- the counters of the grouped loop above said exactly that
  // (profiles/r05f_attn_pmc.md: pipe busy 51 % of the time, MFMA and VALU co-executing 7 % of it; 4000 cycles per tile = 2048 of
  // MFMA + ~1950 of everything else, nothing hidden).  Here every MFMA is followed by ITS share of the rest, pinned by a
  // scheduling fence per gap:
  //   gap (m, k) = behind MFMA k of group m:  the LDS read(s) of fragment (m+2, k-1) (k = 0: of (m+1, 3)) -- 4 .. 7 MFMAs
  //   ahead of its use, into the ring slot the MFMA before just read; three instructions of softmax-backward arithmetic
  //   (groups that carry a chunk); behind a group's last gap a piece of the next tile (groups 0 .. 8).
  // One counted wait per group: fragment (m, 3) is the youngest of group m, requested four gaps ago; behind it went only
  // the three fragments (m+1, 0..2) (+ six statistics reads in group 0) -- lgkmcnt(3 R) covers the group.  The tail of a tile
  // may not read the next tile before the hand-off (start of the last group): the fragments (G, 0..2) wait for gap (G-1, 0),
  // so the hand-off's lgkmcnt(0) finds nothing younger than three gaps.
  auto tile_pair_fine = [&](auto plain_c, int it0) __attribute__((always_inline)) {
   constexpr bool PLAIN = decltype(plain_c)::value != 0;
   constexpr int G = NGRP;
#pragma unroll
   for (int cur = 0; cur < 2; ++cur) {
    const int it = it0 + cur;
    const int qt0 = it < niter ? cmp_qt * kQT : nqt64 * kQT;
    if (++cmp_h == group) {
      cmp_h = 0;
      --cmp_qt;
    }
    if (!(DBG & 8)) issue_begin(it + 1);
    const int mask_lim = key_ok ? (CAUSAL ? krow - (qt0 + off) : -0x40000000) : 0x40000000;
    const int mask_hi = kend - qt0;  // PACKED: last local query row of this tile that belongs to the key's sequence
    sched_fence();
    f32x16 s[2], dp[2];
    u32x4 pf[4], dsf[4];
    s[0] = sn;  // (requested behind the previous tile's hand-off)
    dp[0] = dpn;
    constexpr int NCH = 4 / NGA;  // softmax chunks (4 query rows each) a group carries (NGA == NGC)
    static_assert(NGA == NGC, "head_dim 64 / 128");
    float pe[NCH][4], dse[NCH][4];
    // slice k of the softmax backward of chunk qd of sub-tile sub (4 consecutive query rows: C-layout registers qd*4 .. +3):
    // three instructions per slice for a tile that needs no mask
    auto slice = [&](int sub, int qd, int c, int k) {
      if (DBG & 1) {
        if (k == 3) {
          const int op = sub * 2 + (qd >> 1), w = (qd & 1) * 2;
          pf[op][w] = f32_as_u32(s[sub][qd * 4]);
          pf[op][w + 1] = f32_as_u32(s[sub][qd * 4 + 1]);
          dsf[op][w] = f32_as_u32(dp[sub][qd * 4]);
          dsf[op][w + 1] = f32_as_u32(dp[sub][qd * 4 + 1]);
        }
        return;
      }
      const int ql = sub * 32 + 8 * qd + 4 * hi;
      auto ex = [&](int e) {
        float x = s[sub][qd * 4 + e];
        if (PACKED && !PLAIN)
          x = (mask_lim <= ql + e && ql + e <= mask_hi) ? x : -INFINITY;
        else if (!PLAIN)
          x = (mask_lim <= ql + e) ? x : -INFINITY;
        pe[c][e] = fast_exp2(x);  // x = S*scale*log2(e) - lse*log2(e): the chain started from -lse*log2(e)
      };
      auto ml = [&](int e) { dse[c][e] = pe[c][e] * dp[sub][qd * 4 + e]; };  // (the dP chain started from -delta)
      const int op = sub * 2 + (qd >> 1), w = (qd & 1) * 2;
      if (k == 0) {
        ex(0);
        ex(1);
        ex(2);
      } else if (k == 1) {
        ex(3);
        ml(0);
        ml(1);
      } else if (k == 2) {
        ml(2);
        ml(3);
        pf[op][w] = pack2<T>(pe[c][0], pe[c][1]);
      } else {
        pf[op][w + 1] = pack2<T>(pe[c][2], pe[c][3]);
        dsf[op][w] = pack2<T>(dse[c][0], dse[c][1]);
        dsf[op][w + 1] = pack2<T>(dse[c][2], dse[c][3]);
      }
    };
    constexpr int RA = 1, RC = 2;  // LDS reads per fragment: one b128 row read, two transposing reads
    auto nreads = [&](int m) -> int { return (DBG & 2) ? 0 : (((m % G) < 2 * NGA) ? RA : RC); };
#pragma unroll
    for (int m = 0; m < G; ++m) {
      u32x4(&fc)[4] = (m & 1) ? fb : fa;
      const bool aph = m < 2 * NGA;
      const int sub = m / NGA, ga = m % NGA;                                      // S / dP groups
      const int cidx = m - 2 * NGA, half = cidx / NGC, gc = cidx % NGC;           // dV / dK groups
      const int dtp = gc >> 1, jc = half * 2 + (gc & 1);
      // ---- this group's fragments have landed
      if (m == G - 1) {  // hand-off: every fragment of this tile is in registers; tile it+1 (requested during this tile) has landed
        wait_vmcnt0();
        wait_lgkmcnt0();
        if (!(DBG & 16)) raw_barrier();
        sched_fence();
        wait_frags<15>(fc[0], fc[1], fc[2], fc[3]);  // dependency only
      } else {
        const int behind = 3 * nreads(m + 1) + ((m == 1 && !(DBG & 1)) ? 6 : 0);  // (group 0's gaps 1 .. 3 also carry statistics reads)
        if (behind >= 9)
          wait_frags<9>(fc[0], fc[1], fc[2], fc[3]);
        else if (behind >= 6)
          wait_frags<6>(fc[0], fc[1], fc[2], fc[3]);
        else if (behind >= 3)
          wait_frags<3>(fc[0], fc[1], fc[2], fc[3]);
        else
          wait_frags<0>(fc[0], fc[1], fc[2], fc[3]);
      }
      if (m == 0) after_wait_acc(s[0], dp[0]);
      if (m == 2) after_wait_acc(s[1], dp[1]);  // (requested in the gaps of groups 0 and 1: older than everything group 2 waits for)
      sched_fence();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // ---- MFMA k of the group
        if (aph) {
          if (k & 1)
            dp[sub] = mm(fc[k], vf[2 * ga + (k >> 1)], dp[sub]);
          else
            s[sub] = mm(fc[k], kf[2 * ga + (k >> 1)], s[sub]);
        } else {
          if (k & 1)
            dkacc[2 * dtp + (k >> 1)] = mm(fc[k], dsf[jc], dkacc[2 * dtp + (k >> 1)]);
          else
            dvacc[2 * dtp + (k >> 1)] = mm(fc[k], pf[jc], dvacc[2 * dtp + (k >> 1)]);
        }
        // ---- its gap
        // statistics that start sub-tile 1's chains: two reads in each of the gaps (0, 1), (0, 2), (0, 3), (1, 0), BEFORE the
        // gap's fragment request (the wait counts above rely on that order)
        {
          const int jst = (m == 0) ? k - 1 : (m == 1 && k == 0 ? 3 : -1);
          if (jst >= 0 && !(DBG & 1)) {
            const u32x4 l4 = lds_read16_abs(stataddr[cur], (32 + 8 * jst) * 4);
            const u32x4 d4 = lds_read16_abs(stataddr[cur], (32 + 8 * jst) * 4 + kQT * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              s[1][4 * jst + e] = u32_as_f32(l4[e]);
              dp[1][4 * jst + e] = u32_as_f32(d4[e]);
            }
          }
          if (m == 0 && k == 0 && (DBG & 1)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              s[1][r] = 0.f;
              dp[1][r] = 0.f;
            }
          }
        }
        // the fragment request of the gap
        if (m == G - 1 && k == 0) {  // behind the hand-off: the next tile's first statistics, its group 0, and (below) (G, 3)
          load_stats(sn, dpn, cur ^ 1, 0);
#pragma unroll
          for (int j = 0; j < 3; ++j) req(cur ^ 1, 0, j);
        }
        {
          const int mt = k == 0 ? m + 1 : m + 2, jt = k == 0 ? 3 : k - 1;  // target fragment of gap (m, k)
          const bool held = (m == G - 2 && k >= 1);                         // (G, 0..2): not before the hand-off
          if (!held) {
            if (mt >= G)
              req(cur ^ 1, mt - G, jt);
            else
              req(cur, mt, jt);
          }
        }
        // softmax backward: sub-tile 0 beside S / dP of sub-tile 1, sub-tile 1 beside the first half of dV / dK
        if (aph && sub == 1) {
#pragma unroll
          for (int c = 0; c < NCH; ++c) slice(0, ga * NCH + c, c, k);
        }
        if (!aph && half == 0) {
#pragma unroll
          for (int c = 0; c < NCH; ++c) slice(1, gc * NCH + c, c, k);
        }
        // a piece of the next tile in the first gap of the groups 0 .. NPIECE-1 (measured level with / 0.4-0.9 % ahead of "behind
        // the group's last MFMA", profiles/r05h_, r05i_attn_variants_ab.jsonl)
        if (k == 0 && m < NPIECE && !(DBG & 8)) issue_piece(cur ^ 1, m);
        sched_fence();
      }
    }
   }
  };
  int it_plain = 0;  // (even) leading tiles of this wave that take the body without the mask test
  int it_first = 0;  // PACKED: (even) tiles in front of them whose last rows lie behind the end of a key's sequence / window
  if (!(DBG & 32) && ballot64(key_ok) == ~0ull) {
    int n = niter2;
    if (CAUSAL) {
      const int need = kw0 + 31 - off;                       // first query row that sees the wave's last key
      const int qmin = need > 0 ? (need + kQT - 1) / kQT : 0;  // first query tile all of whose rows see it
      const int tiles = nqt64 > qmin ? nqt64 - qmin : 0;
      n = tiles * group < niter2 ? tiles * group : niter2;
    }
    it_plain = n & ~1;
    if (PACKED) {
      // Round 6 (end): until then EVERY tile of a packed / windowed launch carried the two-sided test (4 VALU per element on a
      // loop bound by instruction issue: 1.6-1.8x the plain kernel's time per visible pair, profiles/r06ad_attn_window_kernels.txt).
      // A tile whose 64 query rows all lie at or before the sequence end of EVERY key of this wave (k_end is non-decreasing:
      // the wave's minimum) and behind the causal diagonal needs no test at all -- the walk goes from the last visible
      // q-tile down, so the tiles are: a few with the upper test, the plain ones, the diagonal ones.
      const int kmin = -(int)wave_max(-(float)kend);
      const int qt_ok = (kmin + 1 - off) / kQT - 1;  // last q-tile all of whose rows see every key of the wave
      int lead = (nqt64 - 1 - qt_ok) * group;       // tiles walked before it
      lead = lead < 0 ? 0 : lead;
      it_first = (lead + 1) & ~1;
      it_first = it_first < it_plain ? it_first : it_plain;
    }
  }
  int it0 = 0;
  if constexpr (PACKED) {
    // one loop, a wave-uniform choice of the body per tile pair (two copies of the body, as in the other variants)
    constexpr bool TWO = FINE || !(DROP && D > 64);  // (two bodies of the dropout variant at head_dim 128 do not fit the registers)
    for (; it0 < niter2; it0 += 2) {
      const bool plain = TWO && it0 >= it_first && it0 < it_plain;
      if constexpr (FINE) {
        if (plain)
          tile_pair_fine(IntC<1>{}, it0);
        else
          tile_pair_fine(IntC<0>{}, it0);
      } else {
        if (plain)
          tile_pair(IntC<(TWO ? 1 : 0)>{}, it0);
        else
          tile_pair(IntC<0>{}, it0);
      }
    }
  } else if constexpr (FINE) {
    for (; it0 < it_plain; it0 += 2) tile_pair_fine(IntC<1>{}, it0);
    for (; it0 < niter2; it0 += 2) tile_pair_fine(IntC<0>{}, it0);
  } else {
    if (!(DROP && D > 64)) {  // (two bodies of the dropout variant at head_dim 128 do not fit the registers)
      for (; it0 < it_plain; it0 += 2) tile_pair(IntC<1>{}, it0);
    }
    for (; it0 < niter2; it0 += 2) tile_pair(IntC<0>{}, it0);
  }
  // the last hand-off left nothing in flight; every wave is past its LDS reads only after a barrier
  wait_vmcnt0();
  block_sync();
  T* dK = reinterpret_cast<T*>(g.dk) + (int64_t)b * a.ksb + (int64_t)hkv * a.ksh;
  T* dV = reinterpret_cast<T*>(g.dv) + (int64_t)b * a.vsb + (int64_t)hkv * a.vsh;
  const unsigned st = (unsigned)wave * (32u * OROWB);
  // dK = scale * dS^T Q; with pre-scaled queries Q' = Q * scale * log2(e): dK = dS^T Q' / log2(e)
  const float dk_mul = a.q_prescaled ? g.scale / a.scale_log2 : g.scale;
  store_rows_via_lds<T, D>(dkacc, dk_mul, smem, st, dK, a.kss, kw0, a.seq_k, lane, reinterpret_cast<const T*>(g.rope_cos),
                           reinterpret_cast<const T*>(g.rope_sin),
                           (g.rope_cos_batch == 1 ? 0 : (int64_t)b * a.seq_k) + kw0);  // (rotary: key position = row index)
  wave_lockstep_point();
  store_rows_via_lds<T, D>(dvacc, DROP ? drop.scale : 1.f, smem, st, dV, a.vss, kw0, a.seq_k, lane);  // (dropout: P's 1 / (1 - p))
}

template <typename T, int D>
static int dkdv_launch(const AttnBwdArgs& g, bool causal, hipStream_t s) {
  const AttnArgs& a = g.f;
  const bool mask = a.key_valid != nullptr;
  const bool drop = a.drop_thr != 0;
  const int nkvt = (int)ceil_div(a.seq_k, kKVB);
  const size_t smem = (size_t)2 * (2 * kQT * D * 2 + 2 * kQT * 4);
  dim3 grid((unsigned)(nkvt * a.heads_kv * a.batch)), block(kAttnThreads);
#define TAMD_KV(C_, M_, D_) hipLaunchKernelGGL((attn_bwd_dkdv_kernel<T, D, C_, M_, D_>), grid, block, smem, s, g, nkvt)
  if (a.q_start != nullptr) {  // packed sequences (causal only, checked by the caller)
    if (drop)
      hipLaunchKernelGGL((attn_bwd_dkdv_kernel<T, D, true, true, true, 0, true>), grid, block, smem, s, g, nkvt);
    else
      hipLaunchKernelGGL((attn_bwd_dkdv_kernel<T, D, true, true, false, 0, true>), grid, block, smem, s, g, nkvt);
    return launch_status();
  }
  if (drop) {  // the dropout variants always carry the padding-mask code
    if (causal)
      TAMD_KV(true, true, true);
    else
      TAMD_KV(false, true, true);
  } else if (causal) {
    if (mask) {
      TAMD_KV(true, true, false);
      return launch_status();
    }
#ifdef TAMD_DIAG  // ablation instantiations (wrong results by design): libtamd_diag.so only
    static const int dbg = [] {
      const char* e = getenv("TAMD_DKDV_DBG");
      return e ? atoi(e) : 0;
    }();
#define TAMD_KVD(N_)                                                                                          \
  if (dbg == N_) {                                                                                            \
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<T, D, true, false, false, N_>), grid, block, smem, s, g, nkvt);  \
    return launch_status();                                                                                   \
  }
    TAMD_KVD(1) TAMD_KVD(2) TAMD_KVD(4) TAMD_KVD(8) TAMD_KVD(16) TAMD_KVD(3) TAMD_KVD(7) TAMD_KVD(6) TAMD_KVD(32)
#undef TAMD_KVD
#endif
    TAMD_KV(true, false, false);
  } else {
    if (mask)
      TAMD_KV(false, true, false);
    else
      TAMD_KV(false, false, false);
  }
#undef TAMD_KV
  return launch_status();
}

int attn_bwd_dkdv_launch(const AttnBwdArgs& g, int dtype, int head_dim, bool causal, hipStream_t s) {
  if (head_dim == 128) {
    TAMD_DISPATCH_HALF(dtype, return (dkdv_launch<T, 128>(g, causal, s)));
  } else {
    TAMD_DISPATCH_HALF(dtype, return (dkdv_launch<T, 64>(g, causal, s)));
  }
  return TAMD_E_DTYPE;
}

}  // namespace tamd
