/* tamd.h -- C ABI of libtamd.so: the MI355X (gfx950) transformer-block kernels.
 *
 * This is the drop-in boundary ("B3"): plain pointers,
 * sizes, strides, scalars and a hipStream_t -- no torch types.  The product's
 * host side is compiled: transformers_amd/csrc/torch_binding.cpp
 * (libtamd_torch.so) resolves these symbols with dlopen/dlsym at run time and
 * exposes them as torch.ops.tamd.* (TORCH_LIBRARY); ctypes (transformers_amd/
 * _cabi.py) is only how the tests and tools call an entry point directly.  Any
 * other host (C++, a `kernels`-style Hub package) can bind the same symbols.
 *
 * Contract for every entry point:
 *   - returns 0 on success; <0 = TAMD_E_* argument error (nothing launched);
 *     >0 = the hipError_t of a failed launch;
 *   - never allocates, frees, synchronises or owns memory: all buffers
 *     (including workspaces, sized by the *_workspace_bytes helpers) belong
 *     to the caller and must stay alive until the stream has passed the call;
 *   - launches asynchronously on `stream` (the caller's current stream);
 *   - re-entrant, no global mutable state (autograd worker threads call in);
 *   - device pointers must be 16-byte aligned and row strides multiples of
 *     8 elements unless a parameter says otherwise.
 *
 * Each declaration cites the reference call site it replaces
 * (paths relative to /root/reference/src/transformers).
 */
#ifndef TAMD_H_
#define TAMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TAMD_ABI_VERSION 11

typedef void* tamd_stream_t; /* hipStream_t */

enum tamd_dtype { TAMD_BF16 = 0, TAMD_F16 = 1, TAMD_F32 = 2 };

enum tamd_error {
  TAMD_OK = 0,
  TAMD_E_DTYPE = -1,     /* unsupported dtype for this op */
  TAMD_E_SHAPE = -2,     /* unsupported / inconsistent shape */
  TAMD_E_ALIGN = -3,     /* pointer or stride alignment */
  TAMD_E_NULL = -4,      /* required pointer is NULL */
  TAMD_E_WORKSPACE = -5, /* workspace too small */
  TAMD_E_ARG = -6        /* other invalid argument */
};

enum tamd_act { /* activations.py:58-66 (gelu_new), :69-89 (gelu), :92-103 (silu), :116-123 (quick_gelu) */
  TAMD_ACT_NONE = 0,
  TAMD_ACT_GELU_ERF = 1,
  TAMD_ACT_GELU_TANH = 2,
  TAMD_ACT_QUICK_GELU = 3,
  TAMD_ACT_SILU = 4
};

int tamd_abi_version(void);
const char* tamd_error_string(int code);

/* ------------------------------------------------------------------ norms */

/* LlamaRMSNorm.forward, models/llama/modeling_llama.py:62-67, optionally fused
 * with the residual add of LlamaDecoderLayer.forward :317/:323.
 *   h   = residual ? round(x + residual) : x            (h_out written iff residual)
 *   y   = w * round(h * rsqrt(mean(h^2) + eps))          (fp32 statistics)
 * rstd [rows] fp32 is saved for the backward.  x,residual,y,h_out: [rows, cols]
 * contiguous; w: [cols]. */
int tamd_rmsnorm_fwd(const void* x, const void* residual, const void* w, void* y, void* h_out, float* rstd,
                     int64_t rows, int64_t cols, float eps, int dtype, tamd_stream_t stream);

size_t tamd_norm_bwd_workspace_bytes(int64_t rows, int64_t cols);

/* Backward of the above (autograd of modeling_llama.py:62-67):
 *   dx = (dres ? dres : 0) + rstd * (g - xhat * mean(g * xhat)),  g = dy * w,  xhat = h * rstd
 *   dw = sum_rows dy * round(xhat)
 * h is the tensor that was normalised (h_out, or x when there was no residual).
 * workspace: tamd_norm_bwd_workspace_bytes(rows, cols) bytes of scratch. */
int tamd_rmsnorm_bwd(const void* dy, const void* h, const void* w, const float* rstd, const void* dres, void* dx,
                     void* dw, void* workspace, size_t workspace_bytes, int64_t rows, int64_t cols, int dtype,
                     tamd_stream_t stream);

/* nn.LayerNorm as used by BertSelfOutput/BertOutput (models/bert/modeling_bert.py:289-293, :347-351),
 * BertEmbeddings (:106), GPT2Block (models/gpt2/modeling_gpt2.py:252-254), CLIPEncoderLayer
 * (models/clip/modeling_clip.py:358-360).  Optional fused residual add:
 *   h = residual ? round(x + residual) : x ;  y = (h - mean) * rstd * w + b
 * mean, rstd: [rows] fp32 saved for backward; b may be NULL. */
int tamd_layernorm_fwd(const void* x, const void* residual, const void* w, const void* b, void* y, void* h_out,
                       float* mean, float* rstd, int64_t rows, int64_t cols, float eps, int dtype,
                       tamd_stream_t stream);

/* dx = (dres?dres:0) + rstd*(g - mean(g) - xhat*mean(g*xhat)); dw = sum dy*xhat; db = sum dy (db may be NULL).
 * ABI 8: dcolsum (nullable, [cols], needs dres == NULL) additionally receives the column sums of dx as stored -- in a post-LN
 * block, LayerNorm(dense(x) + residual) (modeling_bert.py:289-293, :347-351), that is the bias gradient of `dense`, which
 * otherwise costs a tamd_colsum pass over dx. */
int tamd_layernorm_bwd(const void* dy, const void* h, const void* w, const float* mean, const float* rstd,
                       const void* dres, void* dx, void* dw, void* db, void* dcolsum, void* workspace,
                       size_t workspace_bytes, int64_t rows, int64_t cols, int dtype, tamd_stream_t stream);

/* Post-LN block of BERT in train mode (models/bert/modeling_bert.py:289-293, :347-351):
 *   h = dropout(x, p) + residual;  y = LayerNorm(h)
 * The keep mask is the counter-based hash of (seed, row * cols + col) (tamd_dropout_hash), so the backward regenerates
 * it; kept elements are scaled by 1/(1-p) and rounded to the storage type before the residual is added, as the
 * reference's bf16 ops do.  Backward: dx = d loss / d h (the gradient of the residual input), dx_drop = the gradient of
 * x (dx masked and scaled); dres as in tamd_layernorm_bwd; dcolsum (ABI 8, nullable, needs dres == NULL): the column
 * sums of dx_drop = the bias gradient of the dense layer that produced x.  seed_dev (ABI 8, nullable): the seed as one
 * 64-bit word in device memory, replacing `seed` (graph-replay-safe dropout: see tamd_attn_params.dropout_seed_dev). */
int tamd_layernorm_dropout_fwd(const void* x, const void* residual, const void* w, const void* b, void* y, void* h_out,
                               float* mean, float* rstd, int64_t rows, int64_t cols, float eps, float dropout_p,
                               uint64_t seed, const uint64_t* seed_dev, int dtype, tamd_stream_t stream);
int tamd_layernorm_dropout_bwd(const void* dy, const void* h, const void* w, const float* mean, const float* rstd,
                               const void* dres, void* dx, void* dx_drop, void* dw, void* db, void* dcolsum,
                               void* workspace, size_t workspace_bytes, int64_t rows, int64_t cols, float dropout_p,
                               uint64_t seed, const uint64_t* seed_dev, int dtype, tamd_stream_t stream);

/* ------------------------------------------------------------------ rotary */

/* apply_rotary_pos_emb + rotate_half, models/llama/modeling_llama.py:130-160, applied IN PLACE to
 * `nheads` consecutive heads of every token row of a [tokens, row_stride] buffer (the fused QKV
 * projection output: q heads then k heads), with the reference's intermediate roundings:
 *   out = round(round(x*cos) + round(rotate_half(x)*sin)).
 * cos/sin: [cos_batch, seq, head_dim] in `dtype` (cos_batch is 1 or tokens/seq).
 * conj != 0 applies the transposed rotation (the backward, SURVEY §8a "Rope").
 * ABI 7: the first `q_heads` heads (the query heads) are multiplied by `q_scale` BEFORE the final rounding,
 *   out = round((round(x*cos) + round(rotate_half(x)*sin)) * q_scale)
 * -- one rounding, like the reference's own; with q_scale = softmax scale * log2(e) they are the pre-scaled queries of
 * tamd_attn_params.q_prescaled.  q_heads = 0 (or q_scale = 1): the reference's bits.  Forward rotation only. */
int tamd_rope_inplace(void* x, const void* cos, const void* sin, int64_t tokens, int64_t seq, int64_t row_stride,
                      int64_t nheads, int64_t head_dim, int64_t cos_batch, int conj, int64_t q_heads, float q_scale,
                      int dtype, tamd_stream_t stream);

/* ------------------------------------------------------------------ embedding */

/* nn.Embedding forward (models/llama/modeling_llama.py:381; BERT :99-104): out[t,:] = table[ids[t],:].
 * ids int64; bit-exact row copy.  Returns TAMD_E_ARG-free: out-of-range ids are clamped to row 0 and
 * flagged in *oob_flag (int32 device word, may be NULL). */
int tamd_embedding_fwd(const int64_t* ids, const void* table, void* out, int64_t ntokens, int64_t vocab,
                       int64_t dim, int32_t* oob_flag, int dtype, tamd_stream_t stream);

/* embedding_dense_backward: dtable[v,:] = sum_{t: ids[t]==v} dout[t,:] (fp32 accumulation), rows of
 * dtable not referenced are left untouched (caller zero-fills).  sorted_ids/perm: ids sorted ascending
 * and the permutation that sorts them (int64).  padding_idx < 0 = none (BERT padding_idx=0 gets no grad,
 * models/bert/modeling_bert.py:58).  Two passes over 32-token segments of the sorted order (long runs of one id are
 * split across waves and joined deterministically); workspace: tamd_embedding_bwd_workspace_bytes() bytes of fp32. */
size_t tamd_embedding_bwd_workspace_bytes(int64_t ntokens, int64_t dim);
int tamd_embedding_bwd(const int64_t* sorted_ids, const int64_t* perm, const void* dout, void* dtable,
                       void* workspace, size_t workspace_bytes, int64_t ntokens, int64_t vocab, int64_t dim,
                       int64_t padding_idx, int dtype, tamd_stream_t stream);

/* BertEmbeddings.forward, models/bert/modeling_bert.py:68-108: word + token_type + position gathers,
 * adds (each rounded like the reference's three bf16 adds), LayerNorm.  pre_ln (nullable) receives the
 * summed embeddings for the backward. */
int tamd_bert_embeddings_fwd(const int64_t* input_ids, const int64_t* token_type_ids, const int64_t* position_ids,
                             const void* word, const void* type, const void* pos, const void* ln_w,
                             const void* ln_b, void* out, void* pre_ln, float* mean, float* rstd, int64_t ntokens,
                             int64_t dim, int64_t vocab, int64_t type_vocab, int64_t max_pos, float eps,
                             int dtype, tamd_stream_t stream);

/* ------------------------------------------------------------------ MLP element-wise */

/* LlamaMLP.forward inner product, models/llama/modeling_llama.py:174-176:
 *   act[t, j] = round(round(silu(gate[t,j])) * up[t,j]);  gate/up are column blocks of one
 * [tokens, ld] buffer (the fused gate|up projection output) or two buffers. */
int tamd_swiglu_fwd(const void* gate, const void* up, void* act, int64_t tokens, int64_t inter, int64_t ld_gate_up,
                    int64_t ld_act, int dtype, tamd_stream_t stream);
/* d_gate = dact*up*silu'(gate), d_up = dact*silu(gate); also re-materialises act when act_out != NULL. */
int tamd_swiglu_bwd(const void* gate, const void* up, const void* dact, void* dgate, void* dup, void* act_out,
                    int64_t tokens, int64_t inter, int64_t ld_gate_up, int64_t ld_act, int dtype,
                    tamd_stream_t stream);

/* y = act(x [+ bias]) for BertIntermediate (models/bert/modeling_bert.py:334-337), GPT2MLP
 * (models/gpt2/modeling_gpt2.py:238-243), CLIPMLP (models/clip/modeling_clip.py:346-350). bias may be NULL. */
int tamd_bias_act_fwd(const void* x, const void* bias, void* y, int64_t rows, int64_t cols, int act, int dtype,
                      tamd_stream_t stream);
/* dx = dy * act'(x [+ bias]).  ABI 8: dbias (nullable, [cols]) additionally receives the column sums of dx as stored -- the
 * bias gradient of the dense layer whose output x is -- in the same pass; it needs a workspace of
 * tamd_colsum_workspace_bytes(rows, cols) bytes (workspace / workspace_bytes are ignored when dbias == NULL). */
int tamd_bias_act_bwd(const void* x, const void* bias, const void* dy, void* dx, void* dbias, void* workspace,
                      size_t workspace_bytes, int64_t rows, int64_t cols, int act, int dtype, tamd_stream_t stream);

/* out = round(a + b) elementwise (residual adds, modeling_llama.py:317,323), n elements. */
int tamd_add(const void* a, const void* b, void* out, int64_t n, int dtype, tamd_stream_t stream);

/* column sums of a [rows, cols] matrix with fp32 accumulation (bias gradients db = sum_rows dY). */
size_t tamd_colsum_workspace_bytes(int64_t rows, int64_t cols);
int tamd_colsum(const void* x, void* out, void* workspace, size_t workspace_bytes, int64_t rows, int64_t cols,
                int64_t ld, int dtype, tamd_stream_t stream);

/* out[c, r] = in[r, c]: [rows, cols] (row stride ld_in) -> [cols, rows] (row stride ld_out). */
int tamd_transpose(const void* in, void* out, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out, int dtype,
                   tamd_stream_t stream);

/* ------------------------------------------------------------------ loss */

/* ForCausalLMLoss / fixed_cross_entropy, loss/loss_utils.py:32-71, on already-shifted labels:
 *   lse[t] = logsumexp(float(logits[t,:]));  row_loss[t] = label[t]==ignore ? 0 : lse[t] - logits[t,label[t]]
 * logits: [tokens, vocab] (row stride ld) in `dtype`, upcast in-register (never materialised in fp32).
 * lse, row_loss: [tokens] fp32.  The caller reduces row_loss (deterministic) and divides by the number of
 * non-ignored labels (reduction="mean") or by num_items_in_batch (reduction="sum", loss_utils.py:39-45). */
int tamd_cross_entropy_fwd(const void* logits, const int64_t* labels, float* lse, float* row_loss, int64_t tokens,
                           int64_t vocab, int64_t ld, int64_t ignore_index, int dtype, tamd_stream_t stream);
/* dlogits[t,j] = (exp(logits[t,j]-lse[t]) - [j==label[t]]) * (*gscale)   (0 for ignored rows);
 * gscale: device fp32 scalar = upstream grad / normaliser.
 * dlogits is a [tokens, ld] buffer (the row stride of `logits`): columns vocab .. ld-1 of every row are row padding and
 * are set to ZERO, so that the buffer is a valid zero-padded operand of the following dX / dW GEMMs when vocab is not a
 * multiple of 8 (bert-base: vocab 30522 in rows of 30528; models/bert/modeling_bert.py:466-497). */
int tamd_cross_entropy_bwd(const void* logits, const int64_t* labels, const float* lse, const float* gscale,
                           void* dlogits, int64_t tokens, int64_t vocab, int64_t ld, int64_t ignore_index,
                           int dtype, tamd_stream_t stream);

/* ------------------------------------------------------------------ GEMM (MFMA) */

enum tamd_gemm_flags {
  TAMD_GEMM_A_KM = 1, /* A stored [K, M] (k-major rows) instead of [M, K] */
  TAMD_GEMM_B_KN = 2, /* B stored [K, N] instead of [N, K]               */
  /* diagnostic schedule hints (A/B measurements, tests); 0 = library default (full-line kernel when K % 64 == 0 --
   * the small tile for small row-major grids -- else ping-pong).  A hint that does not apply to K is ignored. */
  TAMD_GEMM_SCHED_PP = 1 << 8, /* 8-wave ping-pong kernel, 32-deep stages (every layout, any K)                */
  TAMD_GEMM_SCHED_SM = 2 << 8, /* 128 x 128 tile, two workgroups per CU (row-major operands, K % 64 == 0): the  */
                               /* default for grids of few 256 x 256 tiles that split-K does not take           */
  TAMD_GEMM_SCHED_FL = 3 << 8  /* one wave per SIMD, 64-deep full-line stages (every layout, K % 64 == 0)      */
};
enum tamd_gemm_epilogue {
  TAMD_EPI_NONE = 0,
  TAMD_EPI_BIAS = 1,      /* C = round(acc + bias[n])                        nn.Linear with bias           */
  TAMD_EPI_RESIDUAL = 2,  /* C = round(round(acc [+bias]) + R[m,n])          o_proj/down_proj + residual   */
  TAMD_EPI_BIAS_ACT = 3,  /* C = round(act(round(acc + bias)))               BertIntermediate              */
  TAMD_EPI_ACCUM = 4      /* C = round(acc + C_old[m,n])  (gradient accumulation into an existing .grad)   */
};

/* C[M,N] = A[M,K] . B[N,K]^T  with fp32 MFMA accumulation: nn.Linear forward
 * (modeling_llama.py:254-256,280; :174-176; modeling_bert.py:175-177,289,334,347; pytorch_utils.py:117-121)
 * and, through the layout flags, its two backward products dX = dY.W and dW = dY^T.X.
 * bf16/f16 only.  Requirements: N % 8 == 0, lda/ldb/ldc/ldr % 8 == 0; K % 8 == 0 unless both operands are
 * k-major (then K is a row count and unrestricted: dW over a ragged token count); M % 8 == 0 with TAMD_GEMM_A_KM.
 * bias: [N] or NULL; R: [M, ldr] or NULL (for TAMD_EPI_ACCUM, R is ignored and C is read).
 * Round 4: row-major products with M <= 16 and a plain / bias / residual epilogue -- the projections of a cached decode step,
 * one new token per sequence (modeling_llama.py:254-256, 280, 174-176 with hidden_states [batch, 1, hidden]) -- are bound by
 * the weight bytes, not by the matrix pipe: without a schedule hint they run on the weight-streaming kernels of csrc/gemv.hip
 * (same roundings; fp32 summation order differs from the MFMA tiles'), except the widest ones (N >= 65536) at 5+ rows. */
int tamd_gemm(const void* A, const void* B, void* C, const void* bias, const void* R, int64_t M, int64_t N,
              int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int flags, int epilogue, int act,
              int dtype, tamd_stream_t stream);
/* Same product with an optional fp32 workspace that allows split-K: tile grids too small to fill the GPU (weight
 * gradients of narrow layers) are cut along K, partial tiles go to the workspace and a second kernel reduces and rounds
 * them.  tamd_gemm_workspace_bytes() returns the size that enables it (0: split-K would not be used; every epilogue
 * but TAMD_EPI_BIAS_ACT splits: the reduction applies bias / residual / accumulate with the unsplit kernel's roundings).
 * With workspace == NULL this is tamd_gemm.  Results differ from the unsplit kernel only by fp32 summation order. */
size_t tamd_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int flags, int epilogue);
int tamd_gemm_ws(const void* A, const void* B, void* C, const void* bias, const void* R, int64_t M, int64_t N, int64_t K,
                 int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int flags, int epilogue, int act, int dtype,
                 void* workspace, size_t workspace_bytes, tamd_stream_t stream);

/* ABI 8.  The weight gradient dW[M, N] = dY[K, M]^T . X[K, N] (both operands k-major: TAMD_GEMM_A_KM | TAMD_GEMM_B_KN) of a
 * FUSED projection -- weight rows [q | k | v] (modeling_llama.py:254-256) or [gate | up] (:174-176) -- with the rows of each
 * member stored into that member's own buffer: nseg <= 3 segments, C_segs[i] = [seg_rows[i], ldc], every segment but the last
 * a multiple of 256 rows.  The members' gradients are separate tensors (under DistributedDataParallel: non-adjacent views of
 * the all-reduce buckets, trainer.py:712-737); this writes them without a copy.  K % 64 == 0; epilogue TAMD_EPI_NONE or
 * TAMD_EPI_ACCUM; workspace as tamd_gemm_ws (tamd_gemm_workspace_bytes(M, N, K, flags 3, epilogue), may be NULL). */
int tamd_gemm_seg(const void* A, const void* B, void* const* C_segs, const int64_t* seg_rows, int nseg, int64_t N, int64_t K,
                  int64_t lda, int64_t ldb, int64_t ldc, int epilogue, int dtype, void* workspace, size_t workspace_bytes,
                  tamd_stream_t stream);

/* ABI 8.  Up to 4 independent products of ONE layout in a single launch (+ one reduction launch when K is split):
 *   C_p = A_p . B_p   (TAMD_EPI_NONE)    or    C_p += A_p . B_p   (TAMD_EPI_ACCUM),      K_p % 64 == 0
 * `flags`: TAMD_GEMM_A_KM | TAMD_GEMM_B_KN (A_p = [K, M], B_p = [K, N]: the weight gradients dW = dY^T . X of the dense layers of
 * one transformer layer's backward -- what autograd computes one `mm` at a time for BertSelfAttention.query/key/value,
 * BertSelfOutput.dense, BertIntermediate.dense and BertOutput.dense, models/bert/modeling_bert.py:131-133, 288, 333, 347) or 0
 * (row-major A [M, K] and B [N, K]).  Such gradients are 9 .. 36 output tiles each over 16384 tokens: alone each one has to split
 * K 7 .. 16 ways to reach the GPU's 256 CUs; together two splits do.  Every product is split into equal K ranges of one common
 * length chosen so that the whole group is one round of workgroups; the fp32 partial tiles go to `workspace`
 * (tamd_gemm_group_workspace_bytes; smaller or NULL: no K split). */
typedef struct tamd_gemm_problem {
  const void* a;
  const void* b;
  void* c;
  int64_t m, n, k;
  int64_t lda, ldb, ldc;
} tamd_gemm_problem;
size_t tamd_gemm_group_workspace_bytes(const tamd_gemm_problem* problems, int count, int flags);
int tamd_gemm_group(const tamd_gemm_problem* problems, int count, int flags, int epilogue, int dtype, void* workspace,
                    size_t workspace_bytes, tamd_stream_t stream);

/* ABI 8.  A projection whose leading columns leave scaled: the query columns of a fused q|k|v projection
 * (models/bert/modeling_bert.py:175-177, models/clip/modeling_clip.py:304-318) carrying the attention kernels'
 * scale*log2(e) BEFORE their one rounding (tamd_attn_params.q_prescaled; the counterpart of tamd_rope_inplace's q_scale
 * for models without a rotary kernel):
 *   C[m, n] = round((A . B^T [+ bias])[m, n] * (n < scale_cols ? col_scale : 1))        bias nullable; scale_cols % 4 == 0 */
int tamd_gemm_colscale(const void* A, const void* B, void* C, const void* bias, int64_t M, int64_t N, int64_t K,
                       int64_t lda, int64_t ldb, int64_t ldc, int flags, int64_t scale_cols, float col_scale, int dtype,
                       tamd_stream_t stream);

/* The gate|up projection of LlamaMLP with the SiLU*up product in the GEMM epilogue
 * (models/llama/modeling_llama.py:174-176: down_proj(act_fn(gate_proj(x)) * up_proj(x))):
 *   GU[M, 2I]  = X[M, K] . Wgu[2I, K]^T      Wgu = [gate_proj.weight ; up_proj.weight] (rows), GU = gate | up columns
 *   ACT[M, I]  = silu(gate) * up              rounded exactly as tamd_gemm followed by tamd_swiglu_fwd
 * GU may be NULL (inference: the projection outputs are never written to HBM).  K % 64 == 0, I % 8 == 0. */
int tamd_gemm_swiglu(const void* X, const void* Wgu, void* GU, void* ACT, int64_t M, int64_t I, int64_t K, int64_t ldx,
                     int64_t ldw, int64_t ldgu, int64_t ldact, int dtype, tamd_stream_t stream);

/* ABI 11.  The backward of the same product as the way out of the down projection's dX GEMM (the derivative of
 * models/llama/modeling_llama.py:174-176 with respect to gate_proj(x) and up_proj(x)):
 *   d_act[M, I] = dY[M, K] . Wd[K, I]          Wd = down_proj.weight ([hidden = K, I] row-major), dY = the MLP output's gradient
 *   dGU[M, 2I]  = [ d_act * up * silu'(gate) | d_act * silu(gate) ]      GU = the forward's gate | up columns (tamd_gemm_swiglu)
 * rounded exactly as tamd_gemm(TAMD_GEMM_B_KN) followed by tamd_swiglu_bwd -- d_act never reaches memory (0.94 GB written and
 * read back per Llama-3-8B layer at 8 x 4096 otherwise).  K % 64 == 0, I % 8 == 0; dGU must not alias GU. */
int tamd_gemm_swiglu_bwd(const void* dY, const void* Wd, const void* GU, void* dGU, int64_t M, int64_t I, int64_t K,
                         int64_t lddy, int64_t ldw, int64_t ldgu, int64_t lddgu, int dtype, tamd_stream_t stream);

/* ------------------------------------------------------------------ attention (MFMA, flash-style) */

/* Scaled-dot-product attention replacing eager_attention_forward / repeat_kv
 * (models/llama/modeling_llama.py:179-213), the BERT/CLIP/GPT-2 eager variants
 * (models/bert/modeling_bert.py:111-136, models/clip/modeling_clip.py:259-277,
 * models/gpt2/modeling_gpt2.py:54-72) and integrations/sdpa_attention.py:158-167.
 *
 *   O[b,s,h,:] = softmax_k( scale * Q[b,s,h,:] . K[b,k,h/g,:]  masked ) . V[b,k,h/g,:]
 *
 * q/k/v/o are addressed by element strides (batch, seq, head); head_dim contiguous.  The rows of an operand with more than
 * one row must follow each other upwards, head_dim <= stride_s <= 2^24 elements (the kernels address a 64-row tile with 32-bit
 * byte offsets from a scalar base and cut a ragged tile off with the buffer's size: TAMD_E_ARG otherwise; ABI 9 of round 5).
 * causal: key k visible to query s iff k <= s + (seq_k - seq_q)  (masking_utils.py:76-81).
 * key_valid: optional [batch, seq_k] uint8 padding mask (1 = attend); NULL = no padding.
 * lse [batch, heads_q, seq_q] fp32 (natural-log sum-exp of the scaled scores) is written for
 * the backward; may be NULL for inference.  head_dim in {64, 128}; bf16/f16. */
/* 32-bit mixing function of the dropout masks (exported so hosts/tests can rebuild them). */
uint32_t tamd_dropout_hash(uint64_t seed, uint64_t index);
/* ABI 7: the 16-bit field that decides attention-dropout element (batch_head = b*heads_q + h, q, k): one hash decides a
 * 2 x 2 block of the probability matrix (csrc/dropout.h) -- the element is kept iff field >= floor(dropout_p * 65536). */
uint32_t tamd_attn_dropout_field(uint64_t seed, uint64_t batch_head, uint64_t seq_q, uint64_t seq_k, uint64_t q, uint64_t k);
struct tamd_attn_params {
  const void* q;
  const void* k;
  const void* v;
  void* o;
  float* lse;
  const uint8_t* key_valid;
  int64_t batch, heads_q, heads_kv, seq_q, seq_k, head_dim;
  int64_t q_stride_b, q_stride_s, q_stride_h;
  int64_t k_stride_b, k_stride_s, k_stride_h;
  int64_t v_stride_b, v_stride_s, v_stride_h;
  int64_t o_stride_b, o_stride_s, o_stride_h;
  float scale;
  int causal;
  int dtype;
  /* attention dropout (nn.functional.dropout on the probabilities, modeling_llama.py:209, modeling_bert.py:131):
   * element (b, h, q, k) is kept iff tamd_attn_dropout_field(seed, b*heads_q+h, seq_q, seq_k, q, k) >= floor(dropout_p*2^16)
   * and scaled by 1/(1-p); the same counter-based mask is regenerated in the backward.  0 (or p < 2^-16) disables. */
  float dropout_p;
  uint64_t dropout_seed;
  /* packed sequences (several sequences in one batch row, position_ids restarting: masking_utils.py:728-757,
   * packed_sequence_mask_function :182-188).  int32 [2, batch, seq] or NULL; requires causal = 1 and seq_q == seq_k:
   *   plane 0  q_start[b, q] = index of the first token of query q's sequence
   *   plane 1  k_end[b, k]   = index of the last token of key k's sequence
   * key k is visible to query q iff q_start[b,q] <= k <= q (equivalently k <= q <= k_end[b,k]).
   * The kernels only rely on both planes being non-decreasing along the row with q_start[q] <= q and k_end[k] >= k, so the
   * same planes carry the causal sliding window (masking_utils.py:92-101: q_start = max(0, q - w + 1), k_end = min(seq - 1,
   * k + w - 1)), chunked attention (:104-113: chunk numbers as sequence ids) and any intersection of these (elementwise
   * max of plane 0, min of plane 1). */
  const int32_t* q_start;
  /* ABI 7.  The kernels take the scores in the exp2 domain: they multiply their resident operand by scale*log2(e) and
   * round it again to the storage dtype (0).  A producer that rounds q once anyway can apply the factor before its own
   * rounding (tamd_rope_inplace's q_scale): q_prescaled = 1 says q already carries scale*log2(e) -- same speed, no second
   * rounding.  `scale` stays the softmax scale of the UNSCALED q; the backward's dq / dk are gradients with respect to
   * the unscaled q and to k. */
  int32_t q_prescaled;
  /* ABI 8.  Non-NULL: the dropout seed is read from DEVICE memory (one 64-bit word) instead of `dropout_seed`, by the forward
   * and both backward kernels -- so a training step captured in a HIP graph draws a fresh mask on every replay when the word is
   * written by a captured RNG kernel (torch: `torch.empty(1, dtype=int64, device=...).random_()`, whose philox offset
   * advances per replay).  The mask of seed value s is the same whichever way s arrives. */
  const uint64_t* dropout_seed_dev;
};
int tamd_attn_fwd(const struct tamd_attn_params* p, tamd_stream_t stream);

/* ABI 8.  The same attention for DECODE shapes -- a few query rows (one new token per sequence, or a short speculative /
 * chunked block) over a long key range: the KV cache of `generate` (cache_utils.py:1730, :1822; modeling_llama.py:243-281).
 * Same parameter block and result as tamd_attn_fwd (dropout_p must be 0, q_start NULL), different schedule: with one query row
 * and grouped-query attention the query heads of a KV head share one pass over its keys, and the key range is split over
 * enough workgroups to fill the GPU (split-KV; fp32 partial rows in `workspace`, merged by a second kernel).
 * workspace: tamd_attn_decode_workspace_bytes(p) bytes, 16-byte aligned. */
size_t tamd_attn_decode_workspace_bytes(const struct tamd_attn_params* p);
int tamd_attn_decode(const struct tamd_attn_params* p, void* workspace, size_t workspace_bytes, tamd_stream_t stream);

/* Backward (SURVEY §8a "Attention"): delta = rowsum(dO*O); dV = P^T dO; dS = P*(dP - delta);
 * dQ = scale dS K; dK = scale dS^T Q; GQA sums dK/dV over the query heads of a group. */
struct tamd_attn_bwd_params {
  struct tamd_attn_params fwd; /* q,k,v,o,lse as in the forward */
  const void* dout;            /* strides = o strides */
  void* dq;                    /* strides = q strides */
  void* dk;                    /* strides = k strides */
  void* dv;                    /* strides = v strides */
  float* delta;                /* workspace, 2 x [batch, heads_q, seq_q] fp32: -rowsum(dO*O), then -lse*log2(e) */
  /* optional (ABI 5): q and k were rotated by apply_rotary_pos_emb before the attention -- dq and dk leave through the
   * transposed rotation (bit-identical to tamd_rope_inplace(conj) on the stored gradients).  cos / sin
   * [rope_cos_batch, seq, 128] in the storage dtype, rope_cos_batch 1 or batch; head_dim 128, seq_q == seq_k;
   * NULL, NULL, 0 = gradients of q, k as given. */
  const void* rope_cos;
  const void* rope_sin;
  int64_t rope_cos_batch;
};
int tamd_attn_bwd(const struct tamd_attn_bwd_params* p, tamd_stream_t stream);

/* ------------------------------------------------------------------ optimizer */

/* One torch.optim.AdamW step on one tensor (the optimizer Trainer builds by default, trainer.py:1783-1799; update rule
 * torch/optim/adam.py `_single_tensor_adam`, decoupled weight decay), fused-kernel semantics: fp32 arithmetic, each
 * stored tensor rounded once.
 *   p *= 1 - lr*wd;  m += (1-b1)*(g*grad_scale - m);  v = b2*v + (1-b2)*(g*grad_scale)^2;
 *   p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps)          step >= 1 is the NEW step count.
 * p, g in `dtype`; m, v in `state_dtype` (= dtype, or TAMD_F32 master-precision moments).  n % 8 == 0 for 16-bit
 * dtype with 16-bit states, n % 4 == 0 otherwise; 16-byte aligned pointers. */
int tamd_adamw_step(void* p, const void* g, void* m, void* v, int64_t n, double lr, double beta1, double beta2,
                    double eps, double weight_decay, int64_t step, double grad_scale, int dtype, int state_dtype,
                    tamd_stream_t stream);

/* ---- multi-tensor step: global gradient-norm clip + AdamW over a whole parameter set (ABI 10) ----
 * Replaces, per optimizer step of `Trainer` (trainer.py:2538-2548 `accelerator.clip_grad_norm_(model.parameters(),
 * args.max_grad_norm)` = torch.nn.utils.clip_grad_norm_, default max_grad_norm 1.0: training_args.py:856; then
 * `self.optimizer.step()`, trainer.py:1783-1799 torch.optim.AdamW): per-tensor norms + a norm of norms + a scaling pass
 * over every gradient + one optimizer launch per parameter.
 *
 * The parameter set of one (dtype, state_dtype) lives in ONE table of int64 words in DEVICE memory, n tensors:
 *     [0,n) p pointers | [n,2n) g pointers | [2n,3n) m | [3n,4n) v | [4n,5n) element counts |
 *     [5n,6n] first chunk of tensor i = sum_{j<i} ceil(numel_j / TAMD_MT_CHUNK); word 6n = total chunks
 * (6n + 1 words).  tamd_mt_sumsq / tamd_mt_scale read only the g, numel and chunk columns.
 * One workgroup per chunk; `total_chunks` is word 6n, passed by the host that built the table.  Pointers need no
 * alignment (a tensor whose pointers are not all 16-byte aligned runs one element per lane). */
#define TAMD_MT_CHUNK 65536
/* partials[c] = sum of squares (fp32) of gradient chunk c, c in [0, total_chunks): fixed order, deterministic. */
int tamd_mt_sumsq(const int64_t* table, int n_tensors, int64_t total_chunks, float* partials, int dtype,
                  tamd_stream_t stream);
/* out[0] = norm = sqrt(sum(partials[0..count))) (summed in double), out[1] = clip coefficient
 * min(1, max_norm / (norm + 1e-6)) -- torch.nn.utils.clip_grad_norm_'s -- or 1 when max_norm <= 0.  Device memory: no
 * host synchronisation. */
int tamd_mt_norm_finish(const float* partials, int64_t count, float* out, double max_norm, tamd_stream_t stream);
/* g *= *coef in place for every tensor of the table (the side effect of clip_grad_norm_; skipped when *coef == 1). */
int tamd_mt_scale(const int64_t* table, int n_tensors, int64_t total_chunks, const float* coef, int dtype,
                  tamd_stream_t stream);
/* tamd_adamw_step on every tensor of the table in one launch; the gradient is scaled by grad_scale * (*grad_scale_dev)
 * (grad_scale_dev NULL: grad_scale alone) in registers -- the clipped gradient is never written. */
int tamd_mt_adamw_step(const int64_t* table, int n_tensors, int64_t total_chunks, double lr, double beta1, double beta2,
                       double eps, double weight_decay, int64_t step, double grad_scale, const float* grad_scale_dev,
                       int dtype, int state_dtype, tamd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TAMD_H_ */
