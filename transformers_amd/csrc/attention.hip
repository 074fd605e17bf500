// attention.hip -- flash-style scaled-dot-product attention for gfx950: forward kernel, C-ABI entry points; the
// backward's delta / dQ kernels come from attention_bwd.inc, the dK/dV kernel is attention_bwd_dkdv.hip, shared
// device code (tile loads, swizzles, dropout hash, packed-sequence bounds) is attention_common.h.
//
// Replaces, behind AttentionInterface (src/transformers/modeling_utils.py:5092-5130):
//   eager_attention_forward + repeat_kv   models/llama/modeling_llama.py:179-213 (fp32 softmax, GQA)
//   eager_attention_forward               models/bert/modeling_bert.py:111-136, models/clip/modeling_clip.py:259-277,
//                                         models/gpt2/modeling_gpt2.py:54-72
//   sdpa_attention_forward                integrations/sdpa_attention.py:84-170
// Never materialises the [S,S] score matrix; GQA is native (no repeat_kv); causal tiles above the
// diagonal are skipped; output is written directly in the [B,S,H,D] layout the callers reshape.
//
// Forward structure (per workgroup = 4 waves = 128 query rows of one (batch, head); 64-key K/V tiles):
//   * K and V tiles stream global -> LDS with global_load_lds_dwordx4, double buffered; the LDS image is
//     lane-linear so the bank swizzle lives on the source address (cdna_hip_programming.md rule 21):
//       [64][D] tile, 16-byte slot' = slot ^ row_swz(row): conflict-free for BOTH the ds_read_b128 row
//       fragments (K in QK^T) and the ds_read_b64_tr_b16 transposing reads (V in PV, K/Q/dO in backward)
//   * S^T = K.Q^T is computed "swapped" (MFMA A=K fragment, B=Q fragment held in registers), so a lane owns
//     one query column: the softmax row max / sum are lane-local plus ONE v_permlane32_swap, and the
//     exponentiated P registers are, without any data movement, the B operand of O^T += V^T.P^T
//     (the key order of the P registers is matched by the addresses of the transposing V reads);
//   * fp32 online softmax in the exp2 domain (scale*log2e folded), fp32 accumulators, P rounded to the
//     storage dtype before PV exactly as the reference rounds softmax(...).to(query.dtype);
//   * the O tile is normalised, rounded, staged through LDS and written as full rows.
#include "attention_common.h"

namespace tamd {

template <typename T, int D, bool CAUSAL, bool HAS_MASK, bool DROP>
// (two workgroups per CU where the registers allow it: the dropout variants at D = 128 need more than 256)
__global__ __launch_bounds__(kAttnThreads, (DROP && D > 64) ? 1 : 2) void attn_fwd_kernel(AttnArgs a) {
  constexpr int ROWB = D * 2;
  constexpr int TILEB = kKB * ROWB;       // one K or V tile
  constexpr int KS = D / 16;              // QK^T k-steps
  constexpr int DT = D / 32;              // output d-tiles
  constexpr int OROWB = ROWB + 16;        // padded staging row
  TAMD_DYN_SMEM(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id_uniform();
  const int hi = lane >> 5, l31 = lane & 31;

  // ---- work decode: (b, h, query tile); heavy (late) causal tiles first, K/V-sharing blocks on one XCD
  const int group = a.heads_q / a.heads_kv;
  int b, h, qt;
  {
    const int bid = blockIdx.x;
    const int per_grp = a.nqt * group;
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
    const int hkv = g % a.heads_kv;
    h = hkv * group + within % group;
    qt = a.nqt - 1 - within / group;
  }
  const int hkv = h / group;
  const int q0 = qt * kQB;
  const int off = a.seq_k - a.seq_q;  // causal: key k visible to query s iff k <= s + off
  const T* Q = reinterpret_cast<const T*>(a.q) + (int64_t)b * a.qsb + (int64_t)h * a.qsh;
  const T* K = reinterpret_cast<const T*>(a.k) + (int64_t)b * a.ksb + (int64_t)hkv * a.ksh;
  const T* V = reinterpret_cast<const T*>(a.v) + (int64_t)b * a.vsb + (int64_t)hkv * a.vsh;

  // ---- Q fragments (MFMA B operand): Q[q = qw0 + l31][d = ks*16 + hi*8 .. +7]
  const int qw0 = q0 + wave * 32;
  const int qrow = qw0 + l31;
  u32x4 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    qf[ks] = (qrow < a.seq_q) ? ld16(Q + (int64_t)qrow * a.qss + ks * 16 + hi * 8) : u32x4{0, 0, 0, 0};

  // packed sequences: this query's first visible key; wave-uniform bounds for tile skipping / mask-free tiles
  const int klo = packed_klo(a, b, qrow);
  const int klo_max = (int)wave_max((float)klo), klo_min = -(int)wave_max(-(float)klo);

  f32x16 oacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;  // running max (log2 domain, scaled) and row sum of this lane's key half

  int kend = a.seq_k;
  if (CAUSAL) {
    const int lim = q0 + kQB - 1 + off + 1;  // keys visible to the last query row of the block
    kend = lim < kend ? lim : kend;
    if (kend < 0) kend = 0;
  }
  const int nkt = (kend + kKB - 1) / kKB;

  TileOffsets<D> toff;
  toff.init(lane);
  const unsigned lds0 = lds_base_u32(smem);
  const DropCtx drop = {a.drop_thr, a.seed_lo, a.seed_hi, a.drop_scale};
  const unsigned long long drop_base = (((unsigned long long)b * a.heads_q + h) * a.seq_q + (qrow < a.seq_q ? qrow : 0)) *
                                       (unsigned long long)a.seq_k;
  constexpr float kDeferThr = 6.f;  // skip the O rescale while the row max grows by < 2^6 (cdna guide T13)

  TileFeed<D> feed;  // (one set of lane offsets: the fast path wants K and V rows the same distance apart)
  feed.init(a.kss, wave, lane);
  const bool fast_feed = TileFeed<D>::usable(a.kss) && a.kss == a.vss;
  // tile t -> LDS buffer buf.  A full tile goes by buffer-addressed LDS-DMA (TileFeed), and inside the tile loop its
  // pieces are issued one by one behind the K.Q^T MFMAs (an LDS-DMA instruction holds its wave for ~90 cycles: 720 per
  // tile when the eight were issued in a row; behind an MFMA the matrix pipe works through that time)
  auto tile_is_fast = [&](int t) { return fast_feed && (t + 1) * kKB <= a.seq_k; };  // (wave-uniform)
  auto issue_piece = [&](int t, int buf, int n) {  // piece n < 2 * NI of a fast tile: K pieces, then V pieces
    constexpr int NI = TileFeed<D>::NI;
    const unsigned k_off = (unsigned)buf * 2u * TILEB, v_off = k_off + TILEB;
    if (n < NI)
      feed.issue_one(K + (int64_t)t * kKB * a.kss, smem, k_off, wave, n);
    else
      feed.issue_one(V + (int64_t)t * kKB * a.vss, smem, v_off, wave, n - NI);
  };
  auto issue = [&](int t, int buf) {
    const unsigned k_off = (unsigned)buf * 2u * TILEB, v_off = k_off + TILEB;
    if (tile_is_fast(t)) {
#pragma unroll
      for (int n = 0; n < 2 * TileFeed<D>::NI; ++n) issue_piece(t, buf, n);
    } else {
      issue_kv_tile<T, D>(K, a.kss, t * kKB, a.seq_k, smem, k_off, wave, lane);
      issue_kv_tile<T, D>(V, a.vss, t * kKB, a.seq_k, smem, v_off, wave, lane);
    }
  };
  // packed sequences: q_start is non-decreasing along a row, so no row of this block sees a key before the first
  // visible key of its first row: whole K/V tiles below it are neither loaded nor visited (buffers alternate from t0)
  const int t0 = (q0 < a.seq_q ? packed_klo(a, b, q0) : 0) / kKB;
  if (nkt > t0) issue(t0, t0 & 1);
  wait_vmcnt0();
  block_sync();

#ifdef TAMD_DIAG
  const bool tr = a.trace != nullptr && blockIdx.x == 0;
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = tr ? device_clock() : 0ull;
#endif
  for (int t = t0; t < nkt; ++t) {
    const int cur = t & 1;
    // the next tile: a ragged one (the last) is issued here in one go, a full one behind this tile's K.Q^T MFMAs
    const bool next_fast = t + 1 < nkt && tile_is_fast(t + 1);
    if (t + 1 < nkt && !next_fast) issue(t + 1, cur ^ 1);
    TAMD_ATTN_PHASE(0)
    const unsigned k_off = (unsigned)cur * 2u * TILEB, v_off = k_off + TILEB;
    const int kt0 = t * kKB;
    // wave-uniform skip: every key of the tile is above the diagonal for all 32 rows of this wave
    // (packed: ... or below the first visible key of all 32 rows)
    const bool wave_active = (!CAUSAL || (kt0 <= qw0 + 31 + off)) && (kt0 + kKB > klo_min);
    if (wave_active) {
      // ---- S^T tile [64 keys][32 q] = K . Q^T   (two 32-key sub-tiles, MFMAs alternate between them)
      // K fragments through a 5-deep register ring, requested four MFMAs ahead: every read of the loop is issued
      // untracked (the compiler would put vmcnt(0) -- the next tile's LDS-DMA -- in front of the transposing V reads and
      // serialise each K read with its MFMA) and waited for with a counted lgkmcnt tied to the fragment register.
      const unsigned kb = lds0 + k_off, vb = lds0 + v_off;
      f32x16 s[2];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
      {
        constexpr int NM = 2 * KS;                 // MFMAs
        constexpr int KA = (DROP || (HAS_MASK && D > 64)) ? 2 : (NM < 8 ? NM : 8);  // K fragments requested ahead (LDS latency under load: several MFMAs)
        u32x4 kr[KA + 1];
#pragma unroll
        for (int i = 0; i < KA; ++i) kr[i] = lds_read16_abs(kb + toff.row[i >> 1], (i & 1) * 32 * ROWB);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          if (i + KA < NM) kr[(i + KA) % (KA + 1)] = lds_read16_abs(kb + toff.row[(i + KA) >> 1], ((i + KA) & 1) * 32 * ROWB);
          constexpr_wait_frag<KA>(NM - 1 - i, kr[i % (KA + 1)]);
          s[i & 1] = mfma32<T>(kr[i % (KA + 1)], qf[i >> 1], s[i & 1]);
          if ((i & 1) && (i >> 1) < 2 * TileFeed<D>::NI && next_fast) issue_piece(t + 1, cur ^ 1, i >> 1);
        }
      }
      TAMD_ATTN_PHASE(1)
      // the first VA V fragments are requested now and land under the softmax arithmetic; P.V requests the others VA
      // steps ahead (LDS latency under load -- two workgroups' reads and the tile loads -- is several MFMAs)
      constexpr int NV = DT * 4;
      constexpr int VA = (DROP || (HAS_MASK && D > 64)) ? 2 : (NV < 7 ? NV : 7);  // (no registers to spare in the dropout / padding-mask variants)
      u32x4 vr[VA + 1];
      auto vreq = [&](int i) -> u32x4 {  // step i = (key block i / DT, d-tile i % DT): two transposing reads
        const int dt = i % DT, j = i / DT;
        const int rb = ((j >> 1) * 32 + (j & 1) * 16) * ROWB;
        const u32x2 lo = lds_read8_tr16_abs(vb + toff.tr[dt][0], rb);
        const u32x2 h2 = lds_read8_tr16_abs(vb + toff.tr[dt][1], rb);
        return u32x4{lo[0], lo[1], h2[0], h2[1]};
      };
#pragma unroll
      for (int i = 0; i < VA; ++i) vr[i] = vreq(i);
      // ---- mask (diagonal / ragged / padded tiles only: wave-uniform branch, the common tile has no mask code)
      const bool need_mask = HAS_MASK || (kt0 + kKB > a.seq_k) || (CAUSAL && (kt0 + kKB - 1 > qw0 + off)) ||
                             (kt0 < klo_max);
      if (need_mask) {
        unsigned long long vmask = ~0ull;
        if (HAS_MASK) {
          const int kp = kt0 + lane;
          vmask = ballot64(kp < a.seq_k && (a.key_valid == nullptr || a.key_valid[(int64_t)b * a.seq_k + kp] != 0));
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kl = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;  // key index inside the tile
            const int kp = kt0 + kl;
            bool vis = kp < a.seq_k && kp >= klo;
            if (CAUSAL) vis = vis && (kp <= qrow + off);
            if (HAS_MASK) vis = vis && ((vmask >> kl) & 1ull);
            s[sub][r] = vis ? s[sub][r] : -INFINITY;
          }
      }
      // ---- online softmax in the exp2 domain: p = exp2(s*c - m), c = scale*log2(e) > 0
      float mx = s[0][0];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[sub][r]);
      mx = fmaxf(mx, swap32_f32(mx));  // the other half-wave holds the other 32 keys of this query
      const float m_tile = mx * a.scale_log2;
      // deferred rescale: keep the old reference max while no row of the wave outgrows it by 2^thr
      if (ballot64(m_tile - m_run > kDeferThr) != 0ull) {
        const float m_new = fmaxf(m_run, m_tile);
        const float alpha = (m_new == -INFINITY) ? 1.f : fast_exp2(m_run - m_new);  // m_run = -inf -> 0
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
      }
      const float m_ref = (m_run == -INFINITY) ? 0.f : m_run;  // row fully masked so far: every p is exp2(-inf) = 0
      float psum = 0.f;
      u32x4 pf[4];  // P^T as MFMA B operand: step j covers registers 8*(j&1).. of sub-tile j>>1
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          // the reference rounds softmax probabilities to the storage dtype before P.V; the row sum uses the
          // unrounded fp32 values like softmax(dtype=float32) does.
          p[r] = fast_exp2(__builtin_fmaf(s[sub][r], a.scale_log2, -m_ref));
          psum += p[r];
          // dropout acts on the probabilities that multiply V, not on the normaliser (softmax, then dropout)
          if (DROP) p[r] *= drop.factor(drop_base, kt0 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi);
        }
#pragma unroll
        for (int st = 0; st < 2; ++st)
          pf[sub * 2 + st] = u32x4{pack2<T>(p[8 * st + 0], p[8 * st + 1]), pack2<T>(p[8 * st + 2], p[8 * st + 3]),
                                   pack2<T>(p[8 * st + 4], p[8 * st + 5]), pack2<T>(p[8 * st + 6], p[8 * st + 7])};
      }
      l_run += psum;
      TAMD_ATTN_PHASE(2)
      // ---- O^T[d][q] += V^T[d][key] . P^T[key][q]
      // step (sub, st): P registers r = 8*st + j  <->  key = sub*32 + 16*st + 8*(j>>2) + 4*hi + (j&3)
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (i + VA < NV) vr[(i + VA) % (VA + 1)] = vreq(i + VA);
        constexpr_wait_frag<2 * VA>(2 * (NV - 1 - i), vr[i % (VA + 1)]);
        oacc[i % DT] = mfma32<T>(vr[i % (VA + 1)], pf[i / DT], oacc[i % DT]);  // consecutive MFMAs: different accumulators
      }
      TAMD_ATTN_PHASE(3)
    } else if (next_fast) {  // this wave skips the tile (all its rows are above the diagonal): its share of the loads still goes out
#pragma unroll
      for (int n = 0; n < 2 * TileFeed<D>::NI; ++n) issue_piece(t + 1, cur ^ 1, n);
    }
    wait_vmcnt0();
    TAMD_ATTN_PHASE(4)
    block_sync();
    TAMD_ATTN_PHASE(5)
  }
#ifdef TAMD_DIAG
  if (tr && lane == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a.trace[wave * 8 + i] = ph[i];
  }
#endif

  // ---- finalise: l over both half-waves, normalise, LSE, stage O through LDS, row-wise stores
  l_run += swap32_f32(l_run);
  const float inv_l = (l_run > 0.f) ? 1.f / l_run : 0.f;
  if (a.lse != nullptr && hi == 0 && qrow < a.seq_q) {
    const float lse = (l_run > 0.f) ? (m_run + fast_log2(l_run)) * 0.69314718055994530942f : INFINITY;
    a.lse[((int64_t)b * a.heads_q + h) * a.seq_q + qrow] = lse;
  }
  const unsigned st_off = (unsigned)wave * (32u * OROWB);
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int d0 = dt * 32 + 8 * qd + 4 * hi;
      const u32x2 pk = {pack2<T>(oacc[dt][qd * 4 + 0] * inv_l, oacc[dt][qd * 4 + 1] * inv_l),
                        pack2<T>(oacc[dt][qd * 4 + 2] * inv_l, oacc[dt][qd * 4 + 3] * inv_l)};
      lds_write8(smem, st_off + (unsigned)l31 * OROWB + (unsigned)d0 * 2u, pk);
    }
  wave_lockstep_point();
  T* O = reinterpret_cast<T*>(a.o) + (int64_t)b * a.osb + (int64_t)h * a.osh;
  constexpr int SLOTS = ROWB / 16;      // 16 (D=128) or 8 (D=64) lanes per row
  constexpr int RPI = 64 / SLOTS;       // rows per wave instruction
#pragma unroll
  for (int it = 0; it < 32 / RPI; ++it) {
    const int row = it * RPI + lane / SLOTS, slot = lane % SLOTS;
    const int qr = qw0 + row;
    const u32x4 v = lds_read16(smem, st_off + (unsigned)row * OROWB + (unsigned)slot * 16u);
    if (qr < a.seq_q) st16(O + (int64_t)qr * a.oss + slot * 8, v);
  }
}

}  // namespace tamd

#ifdef TAMD_DIAG
#include "attention_fwd64.inc"  // experimental 64-rows-per-wave forward: diagnostic library only
#endif
#include "attention_bwd.inc"

using namespace tamd;

namespace {

#ifdef TAMD_DIAG
unsigned long long* g_attn_trace = nullptr;
int g_attn_fwd64 = 0;  // tamd_attn_set_fwd64: route eligible forwards to attn_fwd64_kernel
int g_attn_fwd64_launches = 0;
#endif

template <typename T, int D>
int attn_fwd_launch(const AttnArgs& a, bool causal, hipStream_t s) {
  const size_t smem = (size_t)4 * kKB * D * 2;  // 2 buffers x (K + V); also covers the O staging (4 x 32 x (2D+16))
  dim3 grid((unsigned)(a.nqt * a.heads_q * a.batch)), block(kAttnThreads);
  const bool mask = a.key_valid != nullptr;
  const bool drop = a.drop_thr != 0;
#ifdef TAMD_DIAG
  if (g_attn_fwd64 && D == 128 && !mask && !drop && a.q_start == nullptr && a.seq_k % kKB == 0 && a.kss == a.vss &&
      TileFeed<128>::usable(a.kss)) {
    const int nqt64 = (a.seq_q + kQB64 - 1) / kQB64;
    dim3 grid64((unsigned)(nqt64 * a.heads_q * a.batch));
    ++g_attn_fwd64_launches;
    const size_t smem64 = (size_t)3 * 2 * kKB * 128 * 2;  // 3 x (K + V) = 96 KiB (covers the O staging: 8 x 32 x 272 B)
    if (causal)
      hipLaunchKernelGGL((attn_fwd64_kernel<T, true>), grid64, block, smem64, s, a);
    else
      hipLaunchKernelGGL((attn_fwd64_kernel<T, false>), grid64, block, smem64, s, a);
    return launch_status();
  }
#endif
#define TAMD_AF(C_, M_, D_) hipLaunchKernelGGL((attn_fwd_kernel<T, D, C_, M_, D_>), grid, block, smem, s, a)
  if (drop) {  // the dropout variants always carry the padding-mask code (rare path: keep the instantiation count down)
    if (causal)
      TAMD_AF(true, true, true);
    else
      TAMD_AF(false, true, true);
  } else if (causal) {
    if (mask)
      TAMD_AF(true, true, false);
    else
      TAMD_AF(true, false, false);
  } else {
    if (mask)
      TAMD_AF(false, true, false);
    else
      TAMD_AF(false, false, false);
  }
#undef TAMD_AF
  return launch_status();
}

int attn_check(const tamd_attn_params* p) {
  if (!p || !p->q || !p->k || !p->v || !p->o) return TAMD_E_NULL;
  if (p->head_dim != 64 && p->head_dim != 128) return TAMD_E_SHAPE;
  if (p->batch <= 0 || p->heads_q <= 0 || p->heads_kv <= 0 || p->seq_q <= 0 || p->seq_k <= 0) return TAMD_E_SHAPE;
  if (p->heads_q % p->heads_kv != 0) return TAMD_E_SHAPE;
  if (p->dtype != TAMD_BF16 && p->dtype != TAMD_F16) return TAMD_E_DTYPE;
  if (!(p->dropout_p >= 0.f && p->dropout_p < 1.f)) return TAMD_E_ARG;
  if (p->q_start != nullptr && (!p->causal || p->seq_q != p->seq_k)) return TAMD_E_ARG;
  const int64_t strides[] = {p->q_stride_b, p->q_stride_s, p->q_stride_h, p->k_stride_b, p->k_stride_s, p->k_stride_h,
                             p->v_stride_b, p->v_stride_s, p->v_stride_h, p->o_stride_b, p->o_stride_s, p->o_stride_h};
  for (int64_t st : strides)
    if (st % 8 != 0) return TAMD_E_ALIGN;
  if (!aligned16(p->q) || !aligned16(p->k) || !aligned16(p->v) || !aligned16(p->o)) return TAMD_E_ALIGN;
  return TAMD_OK;
}

AttnArgs make_args(const tamd_attn_params* p) {
  AttnArgs a;
  a.q = p->q;
  a.k = p->k;
  a.v = p->v;
  a.o = p->o;
  a.lse = p->lse;
  a.key_valid = p->key_valid;
  a.q_start = p->q_start;
  a.batch = (int)p->batch;
  a.heads_q = (int)p->heads_q;
  a.heads_kv = (int)p->heads_kv;
  a.seq_q = (int)p->seq_q;
  a.seq_k = (int)p->seq_k;
  a.qsb = p->q_stride_b;
  a.qss = p->q_stride_s;
  a.qsh = p->q_stride_h;
  a.ksb = p->k_stride_b;
  a.kss = p->k_stride_s;
  a.ksh = p->k_stride_h;
  a.vsb = p->v_stride_b;
  a.vss = p->v_stride_s;
  a.vsh = p->v_stride_h;
  a.osb = p->o_stride_b;
  a.oss = p->o_stride_s;
  a.osh = p->o_stride_h;
  a.scale_log2 = p->scale * 1.44269504088896340736f;
  const double pd = p->dropout_p;
  a.drop_thr = (pd > 0.0) ? (unsigned)(pd >= 1.0 ? 4294967295.0 : pd * 4294967296.0) : 0u;
  a.drop_scale = (pd > 0.0 && pd < 1.0) ? (float)(1.0 / (1.0 - pd)) : 1.f;
  a.seed_lo = (unsigned)p->dropout_seed;
  a.seed_hi = (unsigned)(p->dropout_seed >> 32);
  a.nqt = (int)ceil_div(p->seq_q, kQB);
  a.xcd_map = ((p->batch * p->heads_kv) % 8 == 0) ? 1 : 0;
#ifdef TAMD_DIAG
  a.trace = g_attn_trace;
#else
  a.trace = nullptr;
#endif
  return a;
}

}  // namespace

#ifdef TAMD_DIAG
// per-phase shader-clock sums of workgroup 0 of the next tamd_attn_fwd launches: trace[wave * 8 + phase] (uint64[32]),
// phases: 0 tile-load issue, 1 K.Q^T, 2 mask + softmax, 3 P.V, 4 vmcnt wait, 5 barrier.  tools/attn_phases.py
extern "C" int tamd_attn_set_trace(void* buf) {
  g_attn_trace = reinterpret_cast<unsigned long long*>(buf);
  return TAMD_OK;
}
// 1: tamd_attn_fwd takes the experimental 64-rows-per-wave kernel (attention_fwd64.inc) wherever it applies
extern "C" int tamd_attn_set_fwd64(int on) {
  g_attn_fwd64 = on;
  return g_attn_fwd64_launches;  // (how many forwards have taken the experimental kernel so far)
}
#endif

extern "C" uint32_t tamd_dropout_hash(uint64_t seed, uint64_t index) {
  return dropout_hash((unsigned)seed, (unsigned)(seed >> 32), (unsigned)index, (unsigned)(index >> 32));
}

extern "C" int tamd_attn_fwd(const struct tamd_attn_params* p, tamd_stream_t stream) {
  const int chk = attn_check(p);
  if (chk != TAMD_OK) return chk;
  const AttnArgs a = make_args(p);
  hipStream_t s = TAMD_STREAM(stream);
  if (p->head_dim == 128) {
    TAMD_DISPATCH_HALF(p->dtype, return (attn_fwd_launch<T, 128>(a, p->causal != 0, s)));
  } else {
    TAMD_DISPATCH_HALF(p->dtype, return (attn_fwd_launch<T, 64>(a, p->causal != 0, s)));
  }
  return TAMD_E_DTYPE;
}

extern "C" int tamd_attn_bwd(const struct tamd_attn_bwd_params* p, tamd_stream_t stream) {
  if (!p) return TAMD_E_NULL;
  const int chk = attn_check(&p->fwd);
  if (chk != TAMD_OK) return chk;
  if (!p->dout || !p->dq || !p->dk || !p->dv || !p->delta || !p->fwd.lse) return TAMD_E_NULL;
  if ((p->rope_cos != nullptr) != (p->rope_sin != nullptr)) return TAMD_E_NULL;
  if (p->rope_cos != nullptr) {  // rotary on the way out: heads of 128, positions = row indices (no KV offset)
    if (p->fwd.head_dim != 128 || p->fwd.seq_q != p->fwd.seq_k || (p->rope_cos_batch != 1 && p->rope_cos_batch != p->fwd.batch))
      return TAMD_E_ARG;
    if (!aligned16(p->rope_cos) || !aligned16(p->rope_sin)) return TAMD_E_ALIGN;
  }
  if (!aligned16(p->dout) || !aligned16(p->dq) || !aligned16(p->dk) || !aligned16(p->dv)) return TAMD_E_ALIGN;
  const AttnArgs a = make_args(&p->fwd);
  hipStream_t s = TAMD_STREAM(stream);
  if (p->fwd.head_dim == 128) {
    TAMD_DISPATCH_HALF(p->fwd.dtype, return (attn_bwd_launch<T, 128>(a, p, p->fwd.causal != 0, s)));
  } else {
    TAMD_DISPATCH_HALF(p->fwd.dtype, return (attn_bwd_launch<T, 64>(a, p, p->fwd.causal != 0, s)));
  }
  return TAMD_E_DTYPE;
}
