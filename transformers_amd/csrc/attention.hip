// attention.hip -- flash-style scaled-dot-product attention for gfx950: forward kernel (text: attention_fwd_kernel.inc), C-ABI entry points; the
// backward's dQ kernel (which also computes delta) comes from attention_bwd.inc, the dK/dV kernel is attention_bwd_dkdv.hip, shared
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

#include "attention_fwd_kernel.inc"

}  // namespace tamd

#include "attention_bwd.inc"

using namespace tamd;

namespace {

#ifdef TAMD_DIAG
unsigned long long* g_attn_trace = nullptr;
#endif

template <typename T, int D>
int attn_fwd_launch(const AttnArgs& a, bool causal, hipStream_t s) {
  const size_t smem = (size_t)4 * kKB * D * 2;  // 2 buffers x (K + V); also covers the O staging (4 x 32 x (2D+16))
  dim3 grid((unsigned)(a.nqt * a.heads_q * a.batch)), block(kAttnThreads);
  const bool mask = a.key_valid != nullptr;
  const bool drop = a.drop_thr != 0;
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

// split-KV forward + merge (tamd_attn_decode).  No dropout, no packed sequences (checked by the caller).
template <typename T, int D>
int attn_decode_launch(const AttnArgs& a, bool causal, hipStream_t s) {
  const size_t smem = (size_t)4 * kKB * D * 2;
  dim3 grid((unsigned)(a.nqt * a.heads_q * a.batch * a.kv_splits)), block(kAttnThreads);
  if (causal)
    hipLaunchKernelGGL((attn_fwd_kernel<T, D, true, true, false, true>), grid, block, smem, s, a);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<T, D, false, true, false, true>), grid, block, smem, s, a);
  const int rows = a.batch * a.seq_q * a.heads_q;
  hipLaunchKernelGGL((attn_combine_kernel<T, D>), dim3((unsigned)rows), dim3(256), 0, s, a);  // one workgroup per row
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
  // the tile loaders address a 64-row tile with 32-bit byte offsets from a scalar base (buffer-addressed LDS-DMA) and cut ragged
  // tiles off with the buffer's size: rows must follow each other upwards, at most 2^24 elements apart (64 rows x 2^24 x 2 B = 2^31)
  // (a one-row operand's row stride means nothing -- torch leaves anything there: make_args replaces it)
  const int64_t row_strides[] = {p->q_stride_s, p->k_stride_s, p->v_stride_s, p->o_stride_s};
  const int64_t row_counts[] = {p->seq_q, p->seq_k, p->seq_k, p->seq_q};
  for (int i = 0; i < 4; ++i)
    if (row_counts[i] > 1 && (row_strides[i] < p->head_dim || row_strides[i] > ((int64_t)1 << 24))) return TAMD_E_ARG;
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
  a.qss = p->seq_q > 1 ? p->q_stride_s : p->head_dim;  // (one row: any positive stride addresses it; see attn_check)
  a.qsh = p->q_stride_h;
  a.ksb = p->k_stride_b;
  a.kss = p->seq_k > 1 ? p->k_stride_s : p->head_dim;
  a.ksh = p->k_stride_h;
  a.vsb = p->v_stride_b;
  a.vss = p->seq_k > 1 ? p->v_stride_s : p->head_dim;
  a.vsh = p->v_stride_h;
  a.osb = p->o_stride_b;
  a.oss = p->seq_q > 1 ? p->o_stride_s : p->head_dim;
  a.osh = p->o_stride_h;
  a.scale_log2 = p->scale * 1.44269504088896340736f;
  const double pd = p->dropout_p;
  // 16-bit keep threshold (dropout.h: one hash decides a 2 x 2 block of probabilities); a p below 2^-16 is no dropout
  a.drop_thr = (pd > 0.0) ? (unsigned)(pd >= 1.0 ? 65535.0 : pd * 65536.0) : 0u;
  // the keep-scale is the inverse of the EFFECTIVE keep probability (65536 - thr) / 65536 of the quantised threshold, so that
  // E[keep * scale] == 1 exactly (1 / (1 - p) differs from it by <= 2^-16 relative: p = 0.1 -> thr 6553, p_eff 0.09999)
  a.drop_scale = (a.drop_thr != 0u && pd < 1.0) ? (float)(65536.0 / (65536.0 - (double)a.drop_thr)) : 1.f;
  const unsigned long long mixed = attn_seed_mix(p->dropout_seed);  // (dropout.h: the kernels' block mix takes a mixed seed)
  a.seed_lo = (unsigned)mixed;
  a.seed_hi = (unsigned)(mixed >> 32);
  a.seed_dev = reinterpret_cast<const unsigned long long*>(p->dropout_seed_dev);
  a.nqt = (int)ceil_div(p->seq_q, kQB);
  a.xcd_map = ((p->batch * p->heads_kv) % 8 == 0) ? 1 : 0;
  a.q_prescaled = p->q_prescaled != 0;
  a.o_part = nullptr;
  a.lse_part = nullptr;
  a.kv_splits = 1;
  a.tiles_per_split = 0;
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
#endif

extern "C" uint32_t tamd_dropout_hash(uint64_t seed, uint64_t index) {
  return dropout_hash((unsigned)seed, (unsigned)(seed >> 32), (unsigned)index, (unsigned)(index >> 32));
}
extern "C" uint32_t tamd_attn_dropout_field(uint64_t seed, uint64_t batch_head, uint64_t seq_q, uint64_t seq_k, uint64_t q,
                                            uint64_t k) {
  const unsigned long long mixed = attn_seed_mix(seed);
  return attn_dropout_field((unsigned)mixed, (unsigned)(mixed >> 32), batch_head, seq_q, seq_k, q, k);
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

// ---- decode: few query rows (one new token per sequence, or a short speculative / chunked block) over a long key range
// (cache_utils.py:1730, :1822: the KV cache the reference attends over in `generate`; models/llama/modeling_llama.py:243-281).
// The training kernel gives such a call batch x heads workgroups that each walk the whole cache; here
//   * one new token per sequence and grouped-query attention: the `group` query heads of a KV head become the ROWS of one
//     query tile (a re-striding of q and o, no copy), so K / V are read once per KV head instead of once per query head;
//   * the key range is split over enough workgroups to fill the GPU (split-KV), each writing a normalised fp32 partial row
//     and its log-sum-exp into the workspace, merged by attn_combine_kernel.
struct DecodePlan {
  tamd_attn_params p;  // the (possibly re-strided) problem
  int splits, tiles_per_split;
};
static DecodePlan decode_plan(const tamd_attn_params* p0) {
  DecodePlan d;
  d.p = *p0;
  tamd_attn_params& p = d.p;
  const int64_t group = p.heads_q / p.heads_kv;
  if (p.seq_q == 1 && group > 1) {  // rows = the query heads of a KV head; one "head" per KV head; every key visible
    p.seq_q = group;
    p.q_stride_s = p0->q_stride_h;
    p.o_stride_s = p0->o_stride_h;
    p.q_stride_h = group * p0->q_stride_h;
    p.o_stride_h = group * p0->o_stride_h;
    p.heads_q = p.heads_kv;
    p.causal = 0;
  }
  int64_t kend = p.seq_k;  // (causal: the last query row sees every key of the range; rows before it fewer)
  const int64_t nkt = ceil_div(kend, kKB), nqt = ceil_div(p.seq_q, kQB);
  const int64_t base = nqt * p.heads_q * p.batch;
  int64_t splits = ceil_div(768, base);  // ~3 workgroups per CU
  if (splits > nkt) splits = nkt;
  if (splits > 256) splits = 256;
  if (splits < 1) splits = 1;
  d.tiles_per_split = (int)ceil_div(nkt, splits);
  d.splits = (int)ceil_div(nkt, d.tiles_per_split);  // no empty split
  return d;
}
extern "C" size_t tamd_attn_decode_workspace_bytes(const struct tamd_attn_params* p) {
  if (!p || p->heads_kv <= 0 || p->heads_q % p->heads_kv != 0 || p->seq_q <= 0 || p->seq_k <= 0) return 0;
  const DecodePlan d = decode_plan(p);
  const size_t rows = (size_t)d.splits * (size_t)d.p.batch * (size_t)d.p.seq_q * (size_t)d.p.heads_q;
  return rows * ((size_t)d.p.head_dim + 1) * sizeof(float);
}
extern "C" int tamd_attn_decode(const struct tamd_attn_params* p0, void* workspace, size_t workspace_bytes,
                                tamd_stream_t stream) {
  const int chk = attn_check(p0);
  if (chk != TAMD_OK) return chk;
  if (p0->dropout_p != 0.f || p0->q_start != nullptr) return TAMD_E_ARG;
  if (!workspace) return TAMD_E_NULL;
  if (workspace_bytes < tamd_attn_decode_workspace_bytes(p0) || !aligned16(workspace)) return TAMD_E_WORKSPACE;
  const DecodePlan d = decode_plan(p0);
  AttnArgs a = make_args(&d.p);
  a.xcd_map = 0;
  a.kv_splits = d.splits;
  a.tiles_per_split = d.tiles_per_split;
  const size_t rows = (size_t)d.splits * (size_t)d.p.batch * (size_t)d.p.seq_q * (size_t)d.p.heads_q;
  a.o_part = reinterpret_cast<float*>(workspace);
  a.lse_part = a.o_part + rows * (size_t)d.p.head_dim;
  hipStream_t s = TAMD_STREAM(stream);
  if (d.p.head_dim == 128) {
    TAMD_DISPATCH_HALF(d.p.dtype, return (attn_decode_launch<T, 128>(a, d.p.causal != 0, s)));
  } else {
    TAMD_DISPATCH_HALF(d.p.dtype, return (attn_decode_launch<T, 64>(a, d.p.causal != 0, s)));
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
