// elementwise.hip -- the HBM-bound streaming kernels of the transformer block for gfx950:
// rotary embedding, embedding gather / scatter-add, SwiGLU and GeLU-family activations,
// residual add, 2-D transpose and the (fp32-in-register) cross-entropy.
//
// All of them move 16 bytes per lane per access (cdna_hip_programming.md G13), keep the
// reference's intermediate roundings where the reference computes in the storage dtype,
// and never synchronise or allocate.
#include "common.h"
#include "colsum.h"

namespace tamd {

// =============================================================== rotary
// apply_rotary_pos_emb / rotate_half, models/llama/modeling_llama.py:130-160
//   out[i]       = round( round(x[i]      *cos[i])      + round(-x[i+D/2]*sin[i]) )
//   out[i + D/2] = round( round(x[i + D/2]*cos[i+D/2])  + round( x[i]    *sin[i+D/2]) )
// conj: the transposed rotation  (dq = dq'*cos + R^T(dq'*sin))  -- SURVEY §8a backward contract.
template <typename T>
__global__ void rope_kernel(T* __restrict__ x, const T* __restrict__ cosp, const T* __restrict__ sinp,
                            int64_t tokens, int64_t seq, int64_t row_stride, int nheads, int head_dim,
                            int64_t cos_batch, int conj, int q_heads, float q_scale) {
  constexpr int VE = vec16<T>::N;
  const int half = head_dim / 2;
  const int vec_per_head = half / VE;  // vectors in the first half of a head
  const int64_t total = tokens * nheads * vec_per_head;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int vi = (int)(idx % vec_per_head);
    const int64_t th = idx / vec_per_head;
    const int hd = (int)(th % nheads);
    const int64_t tok = th / nheads;
    const int64_t crow = (cos_batch == 1) ? (tok % seq) : tok;
    T* p1 = x + tok * row_stride + (int64_t)hd * head_dim + vi * VE;
    T* p2 = p1 + half;
    const T* c1 = cosp + crow * head_dim + vi * VE;
    const T* s1 = sinp + crow * head_dim + vi * VE;
    // query heads can leave multiplied by scale*log2(e), applied to the fp32 sum of the two rounded products BEFORE the one
    // rounding to the storage dtype (include/tamd.h: the pre-scaled q of the attention kernels); 1.0 is exact
    const float hs = (hd < q_heads) ? q_scale : 1.f;
    float a[VE], b[VE], ca[VE], cb[VE], sa[VE], sb[VE], oa[VE], ob[VE];
    unpack16<T>(ld16(p1), a);
    unpack16<T>(ld16(p2), b);
    unpack16<T>(ld16(c1), ca);
    unpack16<T>(ld16(c1 + half), cb);
    unpack16<T>(ld16(s1), sa);
    unpack16<T>(ld16(s1 + half), sb);
#pragma unroll
    for (int i = 0; i < VE; ++i) {
      if (!conj) {
        oa[i] = (round_through<T>(a[i] * ca[i]) + round_through<T>(-b[i] * sa[i])) * hs;
        ob[i] = (round_through<T>(b[i] * cb[i]) + round_through<T>(a[i] * sb[i])) * hs;
      } else {
        // y = x*cos + R(x)*sin with R(x)=[-x2, x1]  =>  dx1 = dy1*cos1 + dy2*sin2 ; dx2 = dy2*cos2 - dy1*sin1
        oa[i] = round_through<T>(a[i] * ca[i]) + round_through<T>(b[i] * sb[i]);
        ob[i] = round_through<T>(b[i] * cb[i]) + round_through<T>(-a[i] * sa[i]);
      }
    }
    st16(p1, pack16<T>(oa));
    st16(p2, pack16<T>(ob));
  }
}

// =============================================================== embedding
// One wave copies one row: lane-strided 16-byte chunks (bit-exact gather).
template <typename T>
__global__ void embedding_fwd_kernel(const int64_t* __restrict__ ids, const T* __restrict__ table,
                                     T* __restrict__ out, int64_t ntokens, int64_t vocab, int dim,
                                     int32_t* __restrict__ oob) {
  constexpr int VE = vec16<T>::N;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int64_t t = (int64_t)blockIdx.x * wpb + wave; t < ntokens; t += (int64_t)gridDim.x * wpb) {
    int64_t id = ids[t];
    if (id < 0 || id >= vocab) {
      if (oob != nullptr && lane == 0) atomicOr(reinterpret_cast<int*>(oob), 1);
      id = 0;
    }
    const T* src = table + id * dim;
    T* dst = out + t * dim;
    for (int c = lane * VE; c < dim; c += 64 * VE) st16(dst + c, ld16(src + c));
  }
}

// Scatter-add with fp32 accumulation over runs of equal (sorted) ids, in two passes so that a long run (every BERT
// token has token_type 0; padding ids; frequent tokens) is not serialised on one wave:
//   pass 1  one wave per segment of kEmbSeg consecutive sorted tokens.  Runs that lie inside the segment are summed
//           and stored.  The piece of a run that reaches the segment from the left (head) or leaves it to the right
//           (tail; a run covering the whole segment counts as tail) goes to the fp32 workspace ws[seg][0|1][dim].
//   pass 2  one wave per run that crosses a segment boundary (found at the segment where it starts): tail of the
//           first segment + tails of fully covered segments + head of the last one -> dtable.  Deterministic.
constexpr int kEmbSeg = 32;

template <typename T>
__global__ void embedding_bwd_seg_kernel(const int64_t* __restrict__ sorted_ids, const int64_t* __restrict__ perm,
                                         const T* __restrict__ dout, T* __restrict__ dtable, float* __restrict__ ws,
                                         int64_t ntokens, int64_t vocab, int dim, int64_t padding_idx) {
  constexpr int VE = vec16<T>::N;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int64_t nseg = (ntokens + kEmbSeg - 1) / kEmbSeg;
  for (int64_t sg = (int64_t)blockIdx.x * wpb + wave; sg < nseg; sg += (int64_t)gridDim.x * wpb) {
    const int64_t t0 = sg * kEmbSeg, t1 = (t0 + kEmbSeg < ntokens) ? t0 + kEmbSeg : ntokens;
    for (int c = lane * VE; c < dim; c += 64 * VE) {
      int64_t j = t0;
      while (j < t1) {
        const int64_t id = sorted_ids[j];
        int64_t e = j + 1;
        while (e < t1 && sorted_ids[e] == id) ++e;
        float acc[VE];
#pragma unroll
        for (int i = 0; i < VE; ++i) acc[i] = 0.f;
        for (int64_t q = j; q < e; ++q) {
          float v[VE];
          unpack16<T>(ld16(dout + perm[q] * dim + c), v);
#pragma unroll
          for (int i = 0; i < VE; ++i) acc[i] += v[i];
        }
        const bool from_left = (j == t0) && t0 > 0 && sorted_ids[t0 - 1] == id;
        const bool to_right = (e == t1) && t1 < ntokens && sorted_ids[t1] == id;
        const bool valid = id >= 0 && id < vocab && id != padding_idx;
        if (to_right || from_left) {  // a piece of a longer run: fp32 partial (tail wins when the run covers the segment)
          float* slot = ws + ((sg * 2 + (to_right ? 1 : 0)) * (int64_t)dim + c);
#pragma unroll
          for (int i = 0; i < VE; i += 4)
            st16(slot + i, u32x4{f32_as_u32(acc[i]), f32_as_u32(acc[i + 1]), f32_as_u32(acc[i + 2]), f32_as_u32(acc[i + 3])});
        } else if (valid) {
          st16(dtable + id * dim + c, pack16<T>(acc));
        }
        j = e;
      }
    }
  }
}

// Pass 2: one WORKGROUP (16 waves) per run that crosses a segment boundary, found at the segment where the run starts.  The
// run's end comes from a binary search in the sorted ids; its pieces -- tail of the first segment, tails of the fully covered
// ones, head of the last -- are split over the waves (wave w takes pieces w, w + 16, ...: independent loads, 8 in flight)
// and combined through LDS in wave order: deterministic.  (The first version gave the whole run to ONE wave, piece after
// piece: BERT's token_type_ids are all 0 -- one run of 16384 tokens, 512 dependent round trips, 669 us of a 25 ms
// bert-base step, profiles/r04a_bert_kernel_stats.csv.)
constexpr int kEmbJoinWaves = 16;
template <typename T>
__global__ __launch_bounds__(kEmbJoinWaves * 64) void embedding_bwd_join_kernel(const int64_t* __restrict__ sorted_ids,
                                                                                 T* __restrict__ dtable,
                                                                                 const float* __restrict__ ws, int64_t ntokens,
                                                                                 int64_t vocab, int dim, int64_t padding_idx) {
  __shared__ __attribute__((aligned(16))) float part[kEmbJoinWaves][64][4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t nseg = (ntokens + kEmbSeg - 1) / kEmbSeg;
  for (int64_t sg = blockIdx.x; sg + 1 < nseg; sg += gridDim.x) {  // (block-uniform control flow throughout)
    const int64_t t1 = (sg + 1) * kEmbSeg;  // first token of the next segment (exists: sg + 1 < nseg)
    const int64_t id = sorted_ids[t1 - 1];
    if (sorted_ids[t1] != id) continue;  // no run leaves this segment to the right
    // the run must START in this segment (else an earlier segment owns it): its first token is inside [t0, t1)
    const int64_t t0 = sg * kEmbSeg;
    if (sorted_ids[t0] == id && t0 > 0 && sorted_ids[t0 - 1] == id) continue;
    if (id < 0 || id >= vocab || id == padding_idx) continue;
    int64_t lo = t1, hi = ntokens;  // first token after the run: the first index in [t1, ntokens] whose id differs
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (sorted_ids[mid] == id)
        lo = mid + 1;
      else
        hi = mid;
    }
    const int64_t kl = (lo - 1) / kEmbSeg;  // the segment the run ends in (> sg)
    const int64_t npieces = kl - sg + 1;
    // piece i: i = 0 the tail of segment sg, 0 < i < npieces - 1 the tail of segment sg + i (fully covered: pass 1 filed
    // the whole segment as tail), i = npieces - 1 the head of segment kl
    auto piece = [&](int64_t i) -> const float* {
      return ws + ((sg + i) * 2 + (i == npieces - 1 ? 0 : 1)) * (int64_t)dim;
    };
    for (int c = lane * 4; c < ((dim + 255) / 256) * 256; c += 256) {  // (every wave walks every chunk: the barriers are uniform)
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      if (c < dim) {
#pragma unroll 8
        for (int64_t i = wave; i < npieces; i += kEmbJoinWaves) {
          const u32x4 v = ld16(piece(i) + c);
          a0 += u32_as_f32(v[0]);
          a1 += u32_as_f32(v[1]);
          a2 += u32_as_f32(v[2]);
          a3 += u32_as_f32(v[3]);
        }
      }
      part[wave][lane][0] = a0;
      part[wave][lane][1] = a1;
      part[wave][lane][2] = a2;
      part[wave][lane][3] = a3;
      block_sync();
      if (wave == 0 && c < dim) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int w = 0; w < kEmbJoinWaves; ++w) {
          s0 += part[w][lane][0];
          s1 += part[w][lane][1];
          s2 += part[w][lane][2];
          s3 += part[w][lane][3];
        }
        if constexpr (sizeof(T) == 4)
          st16(dtable + id * dim + c, u32x4{f32_as_u32(s0), f32_as_u32(s1), f32_as_u32(s2), f32_as_u32(s3)});
        else
          st8(dtable + id * dim + c, u32x2{pack2<T>(s0, s1), pack2<T>(s2, s3)});
      }
      block_sync();
    }
  }
}

// BertEmbeddings.forward, models/bert/modeling_bert.py:68-108 (gathers + 2 adds + LayerNorm), one wave per token.
template <typename T, int NCH>
__global__ void bert_embeddings_kernel(const int64_t* __restrict__ input_ids, const int64_t* __restrict__ type_ids,
                                       const int64_t* __restrict__ pos_ids, const T* __restrict__ word,
                                       const T* __restrict__ type, const T* __restrict__ pos,
                                       const T* __restrict__ ln_w, const T* __restrict__ ln_b, T* __restrict__ out,
                                       T* __restrict__ pre_ln, float* __restrict__ mean_out,
                                       float* __restrict__ rstd_out, int64_t ntokens, int dim, int64_t vocab,
                                       int64_t type_vocab, int64_t max_pos, float eps) {
  constexpr int VE = vec16<T>::N;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
  if (t >= ntokens) return;
  int64_t wi = input_ids[t], ti = type_ids[t], pi = pos_ids[t];
  wi = (wi < 0 || wi >= vocab) ? 0 : wi;
  ti = (ti < 0 || ti >= type_vocab) ? 0 : ti;
  pi = (pi < 0 || pi >= max_pos) ? 0 : pi;
  float v[NCH][VE];
  float s1 = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * VE;
    if (col < dim) {
      float a[VE], b[VE], p[VE];
      unpack16<T>(ld16(word + wi * dim + col), a);
      unpack16<T>(ld16(type + ti * dim + col), b);
      unpack16<T>(ld16(pos + pi * dim + col), p);
#pragma unroll
      for (int i = 0; i < VE; ++i) {
        v[c][i] = round_through<T>(round_through<T>(a[i] + b[i]) + p[i]);
        s1 += v[c][i];
      }
      if (pre_ln != nullptr) st16(pre_ln + t * dim + col, pack16<T>(v[c]));
    } else {
#pragma unroll
      for (int i = 0; i < VE; ++i) v[c][i] = 0.f;
    }
  }
  const float mu = wave_sum(s1) / (float)dim;
  float s2 = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * VE;
    if (col < dim) {
#pragma unroll
      for (int i = 0; i < VE; ++i) {
        const float d = v[c][i] - mu;
        s2 += d * d;
      }
    }
  }
  const float rs = rsqrtf(wave_sum(s2) / (float)dim + eps);
  if (lane == 0) {
    mean_out[t] = mu;
    rstd_out[t] = rs;
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * VE;
    if (col < dim) {
      float wv[VE], bv[VE], o[VE];
      unpack16<T>(ld16(ln_w + col), wv);
      unpack16<T>(ld16(ln_b + col), bv);
#pragma unroll
      for (int i = 0; i < VE; ++i) o[i] = (v[c][i] - mu) * rs * wv[i] + bv[i];
      st16(out + t * dim + col, pack16<T>(o));
    }
  }
}

// =============================================================== activations
__device__ __forceinline__ float sigmoid_f(float x) { return fast_sigmoid(x); }  // (tamd_device.h)
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
__device__ __forceinline__ float dsilu_f(float x) {
  const float s = sigmoid_f(x);
  return s * (1.f + x * (1.f - s));
}

template <int ACT>
__device__ __forceinline__ float act_f(float x) {
  if (ACT == TAMD_ACT_GELU_ERF) return gelu_erf_f(x);  // activations.py:69-89 (tamd_device.h)
  if (ACT == TAMD_ACT_GELU_TANH)  // activations.py:58-66
    return 0.5f * x * (1.f + tanhf(0.79788456080286535588f * (x + 0.044715f * x * x * x)));
  if (ACT == TAMD_ACT_QUICK_GELU) return x * sigmoid_f(1.702f * x);  // activations.py:116-123
  if (ACT == TAMD_ACT_SILU) return silu_f(x);                        // activations.py:92-103
  return x;
}
template <int ACT>
__device__ __forceinline__ float dact_f(float x) {
  if (ACT == TAMD_ACT_GELU_ERF) return dgelu_erf_f(x);
  if (ACT == TAMD_ACT_GELU_TANH) {
    const float k = 0.79788456080286535588f;
    const float u = k * (x + 0.044715f * x * x * x);
    const float th = tanhf(u);
    const float du = k * (1.f + 3.f * 0.044715f * x * x);
    return 0.5f * (1.f + th) + 0.5f * x * (1.f - th * th) * du;
  }
  if (ACT == TAMD_ACT_QUICK_GELU) {
    const float s = sigmoid_f(1.702f * x);
    return s + x * 1.702f * s * (1.f - s);
  }
  if (ACT == TAMD_ACT_SILU) return dsilu_f(x);
  return 1.f;
}

// LlamaMLP inner product, models/llama/modeling_llama.py:174-176
template <typename T>
__global__ void swiglu_fwd_kernel(const T* __restrict__ gate, const T* __restrict__ up, T* __restrict__ act,
                                  int64_t tokens, int inter, int64_t ld_gu, int64_t ld_act) {
  constexpr int VE = vec16<T>::N;
  const int vpr = inter / VE;
  const int64_t total = tokens * vpr;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = idx / vpr;
    const int c = (int)(idx % vpr) * VE;
    float g[VE], u[VE], o[VE];
    unpack16<T>(ld16(gate + t * ld_gu + c), g);
    unpack16<T>(ld16(up + t * ld_gu + c), u);
#pragma unroll
    for (int i = 0; i < VE; ++i) o[i] = round_through<T>(silu_f(g[i])) * u[i];
    st16(act + t * ld_act + c, pack16<T>(o));
  }
}

template <typename T, bool WRITE_ACT>
__global__ void swiglu_bwd_kernel(const T* gate, const T* up, const T* __restrict__ dact, T* dgate, T* dup,
                                  T* __restrict__ act_out,
                                  int64_t tokens, int inter, int64_t ld_gu, int64_t ld_act) {
  constexpr int VE = vec16<T>::N;
  const int vpr = inter / VE;
  const int64_t total = tokens * vpr;
  // (token, column vector) of a flat index, kept by STEPPING: the grid stride is a constant number of rows and columns
  // (round 5: the loop divided a 64-bit index by vpr in every trip -- ~60 instructions of software division beside 48 bytes of
  // traffic -- and had one vector's three loads in flight per lane; now two vectors per trip, their six loads issued together)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t idx0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t st_t = stride / vpr;
  const int st_c = (int)(stride % vpr);
  int64_t t = idx0 / vpr;
  int cv = (int)(idx0 % vpr);
  auto one = [&](int64_t tt, int cc, const u32x4& gv, const u32x4& uv, const u32x4& dv) {
    float g[VE], u[VE], d[VE], dg[VE], du[VE], a[VE];
    unpack16<T>(gv, g);
    unpack16<T>(uv, u);
    unpack16<T>(dv, d);
#pragma unroll
    for (int i = 0; i < VE; ++i) {
      const float s = round_through<T>(silu_f(g[i]));
      du[i] = d[i] * s;
      dg[i] = round_through<T>(d[i] * u[i]) * dsilu_f(g[i]);
      a[i] = s * u[i];
    }
    // dgate/dup may alias gate/up (in-place gradient): all loads of this vector are done.
    st16(dgate + tt * ld_gu + cc, pack16<T>(dg));
    st16(dup + tt * ld_gu + cc, pack16<T>(du));
    if (WRITE_ACT) st16(act_out + tt * ld_act + cc, pack16<T>(a));
  };
  auto step = [&](int64_t& tt, int& cc) {  // (tt, cc) += grid stride
    tt += st_t;
    cc += st_c;
    if (cc >= vpr) {
      cc -= vpr;
      ++tt;
    }
  };
  for (int64_t idx = idx0; idx < total; idx += 2 * stride) {
    int64_t t2 = t;
    int c2 = cv;
    step(t2, c2);
    const bool two = idx + stride < total;
    const int c = cv * VE, cc2 = c2 * VE;
    const u32x4 g1 = ld16(gate + t * ld_gu + c), u1 = ld16(up + t * ld_gu + c), d1 = ld16(dact + t * ld_act + c);
    u32x4 g2 = g1, u2 = u1, d2 = d1;
    if (two) {
      g2 = ld16(gate + t2 * ld_gu + cc2);
      u2 = ld16(up + t2 * ld_gu + cc2);
      d2 = ld16(dact + t2 * ld_act + cc2);
    }
    one(t, c, g1, u1, d1);
    if (two) one(t2, cc2, g2, u2, d2);
    t = t2;
    cv = c2;
    step(t, cv);
  }
}

template <typename T, int ACT, bool BWD>
__global__ void bias_act_kernel(const T* __restrict__ x, const T* __restrict__ bias, const T* __restrict__ dy,
                                T* __restrict__ out, int64_t rows, int cols) {
  constexpr int VE = vec16<T>::N;
  const int vpr = cols / VE;
  const int64_t total = rows * vpr;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % vpr) * VE;
    float v[VE], o[VE];
    unpack16<T>(ld16(x + idx * VE), v);
    if (bias != nullptr) {
      float b[VE];
      unpack16<T>(ld16(bias + c), b);
#pragma unroll
      for (int i = 0; i < VE; ++i) v[i] = round_through<T>(v[i] + b[i]);
    }
    if (BWD) {
      float d[VE];
      unpack16<T>(ld16(dy + idx * VE), d);
#pragma unroll
      for (int i = 0; i < VE; ++i) o[i] = d[i] * dact_f<ACT>(v[i]);
    } else {
#pragma unroll
      for (int i = 0; i < VE; ++i) o[i] = act_f<ACT>(v[i]);
    }
    st16(out + idx * VE, pack16<T>(o));
  }
}

// dx = dy * act'(x [+ bias]) AND the column sums of dx as stored (the bias gradient of the dense layer whose output x is:
// BertIntermediate's, models/bert/modeling_bert.py:334-337) in one pass: a thread owns ONE 16-byte column vector and walks
// a slab of rows (blockIdx.y), so the sums stay in its registers; partial [gridDim.y, cols] fp32 -> colsum_f32_kernel.
template <typename T, int ACT>
__global__ void bias_act_bwd_colsum_kernel(const T* __restrict__ x, const T* __restrict__ bias, const T* __restrict__ dy,
                                           T* __restrict__ out, float* __restrict__ part, int64_t rows, int cols,
                                           int rows_per_slab) {
  constexpr int VE = vec16<T>::N;
  const int col = (blockIdx.x * blockDim.x + threadIdx.x) * VE;
  if (col >= cols) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
  const int64_t r1 = (r0 + rows_per_slab < rows) ? r0 + rows_per_slab : rows;
  float acc[VE], b[VE];
#pragma unroll
  for (int i = 0; i < VE; ++i) acc[i] = 0.f, b[i] = 0.f;
  if (bias != nullptr) unpack16<T>(ld16(bias + col), b);
  // 4 rows per trip: their 8 loads go out together (one row at a time the slab walk is a chain of dependent round trips:
  // 3.9 TB/s at bert-base's 16384 x 3072 on MI355X against 5.8 for the grid-stride kernel, profiles/r04a)
  constexpr int U = 4;
  for (int64_t r = r0; r < r1; r += U) {
    u32x4 xv[U], dv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t rr = (r + u < r1) ? r + u : r1 - 1;  // (a tail row is re-read, its result dropped below)
      xv[u] = ld16(x + rr * cols + col);
      dv[u] = ld16(dy + rr * cols + col);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r + u < r1) {
        float v[VE], d[VE], o[VE];
        unpack16<T>(xv[u], v);
        unpack16<T>(dv[u], d);
        if (bias != nullptr) {
#pragma unroll
          for (int i = 0; i < VE; ++i) v[i] = round_through<T>(v[i] + b[i]);
        }
#pragma unroll
        for (int i = 0; i < VE; ++i) {
          o[i] = d[i] * dact_f<ACT>(v[i]);
          acc[i] += round_through<T>(o[i]);
        }
        st16(out + (r + u) * cols + col, pack16<T>(o));
      }
    }
  }
#pragma unroll
  for (int i = 0; i < VE; ++i) part[(int64_t)blockIdx.y * cols + col + i] = acc[i];
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int64_t nvec) {
  constexpr int VE = vec16<T>::N;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nvec;
       idx += (int64_t)gridDim.x * blockDim.x) {
    float x[VE], y[VE];
    unpack16<T>(ld16(a + idx * VE), x);
    unpack16<T>(ld16(b + idx * VE), y);
#pragma unroll
    for (int i = 0; i < VE; ++i) x[i] += y[i];
    st16(out + idx * VE, pack16<T>(x));
  }
}

// =============================================================== transpose
// 64x64 tile of 16-bit elements through LDS: 16-byte global reads along the input rows,
// 16-byte global writes along the output rows; LDS rows padded by 2 elements (4 B) so the
// column walk of the second phase is conflict-free for ds_read_u16.
template <typename T>
__global__ void transpose16_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t rows, int64_t cols,
                                   int64_t ld_in, int64_t ld_out) {
  typedef typename elem<T>::raw raw;
  constexpr int TS = 64, PAD = 2;
  __shared__ raw tile[TS][TS + PAD];
  const int64_t r0 = (int64_t)blockIdx.y * TS, c0 = (int64_t)blockIdx.x * TS;
  const int tid = threadIdx.x;  // 256 threads
  // phase 1: each thread loads 2 x 8 elements: row = tid/8 (+32), col chunk = tid%8
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int r = (tid >> 3) + 32 * k, cc = (tid & 7) * 8;
    const int64_t gr = r0 + r, gc = c0 + cc;
    raw v[8];
    if (gr < rows && gc + 8 <= cols) {
      const u32x4 q = ld16(in + gr * ld_in + gc);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[2 * i] = (raw)(q[i] & 0xffffu);
        v[2 * i + 1] = (raw)(q[i] >> 16);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        v[i] = (gr < rows && gc + i < cols) ? reinterpret_cast<const raw*>(in)[gr * ld_in + gc + i] : (raw)0;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) tile[r][cc + i] = v[i];
  }
  __syncthreads();
  // phase 2: output row = input col; each thread writes 2 x 8 elements of an output row
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int oc = (tid >> 3) + 32 * k;  // input column = output row
    const int rr = (tid & 7) * 8;        // input row chunk = output col chunk
    const int64_t go_r = c0 + oc, go_c = r0 + rr;
    if (go_r >= cols) continue;
    raw v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = tile[rr + i][oc];
    if (go_c + 8 <= rows) {
      u32x4 q;
#pragma unroll
      for (int i = 0; i < 4; ++i) q[i] = (unsigned int)v[2 * i] | ((unsigned int)v[2 * i + 1] << 16);
      st16(out + go_r * ld_out + go_c, q);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (go_c + i < rows) reinterpret_cast<raw*>(out)[go_r * ld_out + go_c + i] = v[i];
    }
  }
}

// =============================================================== cross-entropy
// ForCausalLMLoss / fixed_cross_entropy (loss/loss_utils.py:32-71): logits are upcast to fp32
// in registers; one 256-thread block per token row; online max / sum-exp per lane, block combine.
constexpr int kCEThreads = 256;

__device__ __forceinline__ void online_combine(float& m, float& s, float m2, float s2) {
  const float mn = fmaxf(m, m2);
  s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
  m = mn;
}

template <typename T>
__global__ __launch_bounds__(kCEThreads) void cross_entropy_fwd_kernel(const T* __restrict__ logits,
                                                                       const int64_t* __restrict__ labels,
                                                                       float* __restrict__ lse_out,
                                                                       float* __restrict__ row_loss, int64_t vocab,
                                                                       int64_t ld, int64_t ignore_index) {
  constexpr int VE = vec16<T>::N;
  __shared__ float sm[kCEThreads / 64], ss[kCEThreads / 64];
  const int64_t row = blockIdx.x;
  const T* lr = logits + row * ld;
  float m = -INFINITY, s = 0.f;
  const int64_t nvec = vocab / VE;
  for (int64_t i = threadIdx.x; i < nvec; i += kCEThreads) {
    float v[VE];
    unpack16<T>(ld16(lr + i * VE), v);
    float vm = v[0];
#pragma unroll
    for (int k = 1; k < VE; ++k) vm = fmaxf(vm, v[k]);
    const float mn = fmaxf(m, vm);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < VE; ++k) acc += __expf(v[k] - mn);
    s = s * __expf(m - mn) + acc;
    m = mn;
  }
  for (int64_t i = nvec * VE + threadIdx.x; i < vocab; i += kCEThreads) {  // ragged tail
    const float v = elem<T>::to_f32(reinterpret_cast<const typename elem<T>::raw*>(lr)[i]);
    const float mn = fmaxf(m, v);
    s = s * __expf(m - mn) + __expf(v - mn);
    m = mn;
  }
  // wave combine
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float m2 = shfl_xor_f32(m, off), s2 = shfl_xor_f32(s, off);
    if (m2 != -INFINITY || m != -INFINITY) online_combine(m, s, m2, s2);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sm[wave] = m;
    ss[wave] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float M = sm[0], S = ss[0];
    for (int i = 1; i < kCEThreads / 64; ++i)
      if (sm[i] != -INFINITY || M != -INFINITY) online_combine(M, S, sm[i], ss[i]);
    const float lse = M + __logf(S);
    lse_out[row] = lse;
    const int64_t lab = labels[row];
    float loss = 0.f;
    if (lab != ignore_index && lab >= 0 && lab < vocab)
      loss = lse - elem<T>::to_f32(reinterpret_cast<const typename elem<T>::raw*>(lr)[lab]);
    row_loss[row] = loss;
  }
}

template <typename T>
__global__ __launch_bounds__(kCEThreads) void cross_entropy_bwd_kernel(const T* __restrict__ logits,
                                                                       const int64_t* __restrict__ labels,
                                                                       const float* __restrict__ lse,
                                                                       const float* __restrict__ gscale,
                                                                       T* __restrict__ dlogits, int64_t vocab,
                                                                       int64_t ld, int64_t ignore_index) {
  constexpr int VE = vec16<T>::N;
  typedef typename elem<T>::raw raw;
  const int64_t row = blockIdx.x;
  const T* lr = logits + row * ld;
  T* dr = dlogits + row * ld;
  const int64_t lab = labels[row];
  const bool valid = lab != ignore_index && lab >= 0 && lab < vocab;
  const float g = valid ? *gscale : 0.f;
  const float l = lse[row];
  const int64_t nvec = vocab / VE;
  for (int64_t i = threadIdx.x; i < nvec; i += kCEThreads) {
    float v[VE];
    unpack16<T>(ld16(lr + i * VE), v);
#pragma unroll
    for (int k = 0; k < VE; ++k) {
      const float p = __expf(v[k] - l);
      v[k] = (p - ((i * VE + k) == lab ? 1.f : 0.f)) * g;
    }
    st16(dr + i * VE, pack16<T>(v));
  }
  for (int64_t i = nvec * VE + threadIdx.x; i < vocab; i += kCEThreads) {
    const float v = elem<T>::to_f32(reinterpret_cast<const raw*>(lr)[i]);
    reinterpret_cast<raw*>(dr)[i] = elem<T>::from_f32((__expf(v - l) - (i == lab ? 1.f : 0.f)) * g);
  }
  // row padding (include/tamd.h: dlogits is [tokens, ld]): zero, so the buffer is a valid K-padded GEMM operand
  for (int64_t i = vocab + threadIdx.x; i < ld; i += kCEThreads) reinterpret_cast<raw*>(dr)[i] = elem<T>::from_f32(0.f);
}

static unsigned stream_grid(int64_t work_items, int threads) {
  int64_t b = ceil_div(work_items, threads);
  const int64_t cap = 256 * 8;  // 256 CUs x 8 blocks (cdna_hip_programming.md G11), grid-stride the rest
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

template <typename T>
static int bert_embeddings_dispatch(const int64_t* input_ids, const int64_t* token_type_ids,
                                    const int64_t* position_ids, const void* word, const void* type, const void* pos,
                                    const void* ln_w, const void* ln_b, void* out, void* pre_ln, float* mean,
                                    float* rstd, int64_t ntokens, int64_t dim, int64_t vocab, int64_t type_vocab,
                                    int64_t max_pos, float eps, hipStream_t s) {
  constexpr int VE = vec16<T>::N;
  if (dim % VE != 0) return TAMD_E_SHAPE;
  const int64_t chunks = ceil_div(dim, 64 * VE);
  dim3 grid((unsigned)ceil_div(ntokens, 4)), block(256);
#define TAMD_BE(N_)                                                                                               \
  hipLaunchKernelGGL((bert_embeddings_kernel<T, N_>), grid, block, 0, s, input_ids, token_type_ids, position_ids, \
                     (const T*)word, (const T*)type, (const T*)pos, (const T*)ln_w, (const T*)ln_b, (T*)out,      \
                     (T*)pre_ln, mean, rstd, ntokens, (int)dim, vocab, type_vocab, max_pos, eps)
  if (chunks <= 1) {
    TAMD_BE(1);
  } else if (chunks <= 2) {
    TAMD_BE(2);
  } else if (chunks <= 4) {
    TAMD_BE(4);
  } else if (chunks <= 8) {
    TAMD_BE(8);
  } else {
    return TAMD_E_SHAPE;
  }
#undef TAMD_BE
  return launch_status();
}

}  // namespace tamd

using namespace tamd;

extern "C" {

int tamd_rope_inplace(void* x, const void* cos, const void* sin, int64_t tokens, int64_t seq, int64_t row_stride,
                      int64_t nheads, int64_t head_dim, int64_t cos_batch, int conj, int64_t q_heads, float q_scale,
                      int dtype, tamd_stream_t stream) {
  if (!x || !cos || !sin) return TAMD_E_NULL;
  if (q_heads < 0 || q_heads > nheads || (conj && q_heads > 0 && q_scale != 1.f)) return TAMD_E_ARG;
  if (tokens <= 0 || nheads <= 0) return TAMD_OK;
  if (seq <= 0 || tokens % seq != 0) return TAMD_E_SHAPE;
  if (cos_batch != 1 && cos_batch != tokens / seq) return TAMD_E_SHAPE;
  if (!aligned16(x) || !aligned16(cos) || !aligned16(sin)) return TAMD_E_ALIGN;
  TAMD_DISPATCH_DTYPE(dtype, {
    constexpr int VE = vec16<T>::N;
    if (head_dim % (2 * VE) != 0 || row_stride % VE != 0) return TAMD_E_SHAPE;
    const int64_t total = tokens * nheads * (head_dim / 2 / VE);
    hipLaunchKernelGGL((rope_kernel<T>), dim3(stream_grid(total, 256)), dim3(256), 0, TAMD_STREAM(stream), (T*)x,
                       (const T*)cos, (const T*)sin, tokens, seq, row_stride, (int)nheads, (int)head_dim, cos_batch,
                       conj, (int)q_heads, q_scale);
  });
  return launch_status();
}

int tamd_embedding_fwd(const int64_t* ids, const void* table, void* out, int64_t ntokens, int64_t vocab,
                       int64_t dim, int32_t* oob_flag, int dtype, tamd_stream_t stream) {
  if (!ids || !table || !out) return TAMD_E_NULL;
  if (ntokens <= 0) return TAMD_OK;
  if (!aligned16(table) || !aligned16(out)) return TAMD_E_ALIGN;
  TAMD_DISPATCH_DTYPE(dtype, {
    if (dim % vec16<T>::N != 0) return TAMD_E_SHAPE;
    hipLaunchKernelGGL((embedding_fwd_kernel<T>), dim3(stream_grid(ntokens * 64, 256)), dim3(256), 0,
                       TAMD_STREAM(stream), ids, (const T*)table, (T*)out, ntokens, vocab, (int)dim, oob_flag);
  });
  return launch_status();
}

size_t tamd_embedding_bwd_workspace_bytes(int64_t ntokens, int64_t dim) {
  const int64_t nseg = ceil_div(ntokens > 0 ? ntokens : 1, kEmbSeg);
  return (size_t)nseg * 2 * (size_t)dim * sizeof(float);
}

int tamd_embedding_bwd(const int64_t* sorted_ids, const int64_t* perm, const void* dout, void* dtable,
                       void* workspace, size_t workspace_bytes, int64_t ntokens, int64_t vocab, int64_t dim,
                       int64_t padding_idx, int dtype, tamd_stream_t stream) {
  if (!sorted_ids || !perm || !dout || !dtable || !workspace) return TAMD_E_NULL;
  if (ntokens <= 0) return TAMD_OK;
  if (!aligned16(dout) || !aligned16(dtable) || !aligned16(workspace)) return TAMD_E_ALIGN;
  if (workspace_bytes < tamd_embedding_bwd_workspace_bytes(ntokens, dim)) return TAMD_E_ARG;
  const int64_t nseg = ceil_div(ntokens, kEmbSeg);
  TAMD_DISPATCH_DTYPE(dtype, {
    if (dim % vec16<T>::N != 0 || dim % 4 != 0) return TAMD_E_SHAPE;
    hipLaunchKernelGGL((embedding_bwd_seg_kernel<T>), dim3(stream_grid(nseg * 64, 256)), dim3(256), 0,
                       TAMD_STREAM(stream), sorted_ids, perm, (const T*)dout, (T*)dtable, (float*)workspace, ntokens,
                       vocab, (int)dim, padding_idx);
    if (nseg > 1)
      hipLaunchKernelGGL((embedding_bwd_join_kernel<T>), dim3((unsigned)(nseg - 1 < 4096 ? nseg - 1 : 4096)),
                         dim3(kEmbJoinWaves * 64), 0, TAMD_STREAM(stream), sorted_ids, (T*)dtable, (const float*)workspace,
                         ntokens, vocab, (int)dim, padding_idx);
  });
  return launch_status();
}

int tamd_bert_embeddings_fwd(const int64_t* input_ids, const int64_t* token_type_ids, const int64_t* position_ids,
                             const void* word, const void* type, const void* pos, const void* ln_w,
                             const void* ln_b, void* out, void* pre_ln, float* mean, float* rstd, int64_t ntokens,
                             int64_t dim, int64_t vocab, int64_t type_vocab, int64_t max_pos, float eps, int dtype,
                             tamd_stream_t stream) {
  if (!input_ids || !token_type_ids || !position_ids || !word || !type || !pos || !ln_w || !ln_b || !out || !mean ||
      !rstd)
    return TAMD_E_NULL;
  if (ntokens <= 0) return TAMD_OK;
  TAMD_DISPATCH_DTYPE(dtype, {
    return (bert_embeddings_dispatch<T>(input_ids, token_type_ids, position_ids, word, type, pos, ln_w, ln_b, out,
                                        pre_ln, mean, rstd, ntokens, dim, vocab, type_vocab, max_pos, eps,
                                        TAMD_STREAM(stream)));
  });
  return launch_status();
}

int tamd_swiglu_fwd(const void* gate, const void* up, void* act, int64_t tokens, int64_t inter, int64_t ld_gate_up,
                    int64_t ld_act, int dtype, tamd_stream_t stream) {
  if (!gate || !up || !act) return TAMD_E_NULL;
  if (tokens <= 0 || inter <= 0) return TAMD_OK;
  if (!aligned16(gate) || !aligned16(up) || !aligned16(act)) return TAMD_E_ALIGN;
  TAMD_DISPATCH_DTYPE(dtype, {
    constexpr int VE = vec16<T>::N;
    if (inter % VE != 0 || ld_gate_up % VE != 0 || ld_act % VE != 0) return TAMD_E_SHAPE;
    hipLaunchKernelGGL((swiglu_fwd_kernel<T>), dim3(stream_grid(tokens * (inter / VE), 256)), dim3(256), 0,
                       TAMD_STREAM(stream), (const T*)gate, (const T*)up, (T*)act, tokens, (int)inter, ld_gate_up,
                       ld_act);
  });
  return launch_status();
}

int tamd_swiglu_bwd(const void* gate, const void* up, const void* dact, void* dgate, void* dup, void* act_out,
                    int64_t tokens, int64_t inter, int64_t ld_gate_up, int64_t ld_act, int dtype,
                    tamd_stream_t stream) {
  if (!gate || !up || !dact || !dgate || !dup) return TAMD_E_NULL;
  if (tokens <= 0 || inter <= 0) return TAMD_OK;
  if (!aligned16(gate) || !aligned16(up) || !aligned16(dact) || !aligned16(dgate) || !aligned16(dup) ||
      (act_out && !aligned16(act_out)))
    return TAMD_E_ALIGN;
  TAMD_DISPATCH_DTYPE(dtype, {
    constexpr int VE = vec16<T>::N;
    if (inter % VE != 0 || ld_gate_up % VE != 0 || ld_act % VE != 0) return TAMD_E_SHAPE;
    dim3 grid(stream_grid(tokens * (inter / VE), 256)), block(256);
    if (act_out)
      hipLaunchKernelGGL((swiglu_bwd_kernel<T, true>), grid, block, 0, TAMD_STREAM(stream), (const T*)gate,
                         (const T*)up, (const T*)dact, (T*)dgate, (T*)dup, (T*)act_out, tokens, (int)inter,
                         ld_gate_up, ld_act);
    else
      hipLaunchKernelGGL((swiglu_bwd_kernel<T, false>), grid, block, 0, TAMD_STREAM(stream), (const T*)gate,
                         (const T*)up, (const T*)dact, (T*)dgate, (T*)dup, (T*)act_out, tokens, (int)inter,
                         ld_gate_up, ld_act);
  });
  return launch_status();
}

#define TAMD_ACT_SWITCH(BWD_)                                                                                     \
  switch (act) {                                                                                                  \
    case TAMD_ACT_NONE:                                                                                           \
      hipLaunchKernelGGL((bias_act_kernel<T, TAMD_ACT_NONE, BWD_>), grid, block, 0, s, (const T*)x, (const T*)bias, \
                         (const T*)dy, (T*)out, rows, (int)cols);                                                 \
      break;                                                                                                      \
    case TAMD_ACT_GELU_ERF:                                                                                       \
      hipLaunchKernelGGL((bias_act_kernel<T, TAMD_ACT_GELU_ERF, BWD_>), grid, block, 0, s, (const T*)x,           \
                         (const T*)bias, (const T*)dy, (T*)out, rows, (int)cols);                                 \
      break;                                                                                                      \
    case TAMD_ACT_GELU_TANH:                                                                                      \
      hipLaunchKernelGGL((bias_act_kernel<T, TAMD_ACT_GELU_TANH, BWD_>), grid, block, 0, s, (const T*)x,          \
                         (const T*)bias, (const T*)dy, (T*)out, rows, (int)cols);                                 \
      break;                                                                                                      \
    case TAMD_ACT_QUICK_GELU:                                                                                     \
      hipLaunchKernelGGL((bias_act_kernel<T, TAMD_ACT_QUICK_GELU, BWD_>), grid, block, 0, s, (const T*)x,         \
                         (const T*)bias, (const T*)dy, (T*)out, rows, (int)cols);                                 \
      break;                                                                                                      \
    case TAMD_ACT_SILU:                                                                                           \
      hipLaunchKernelGGL((bias_act_kernel<T, TAMD_ACT_SILU, BWD_>), grid, block, 0, s, (const T*)x, (const T*)bias, \
                         (const T*)dy, (T*)out, rows, (int)cols);                                                 \
      break;                                                                                                      \
    default:                                                                                                      \
      return TAMD_E_ARG;                                                                                          \
  }

int tamd_bias_act_fwd(const void* x, const void* bias, void* y, int64_t rows, int64_t cols, int act, int dtype,
                      tamd_stream_t stream) {
  if (!x || !y) return TAMD_E_NULL;
  if (rows <= 0 || cols <= 0) return TAMD_OK;
  if (!aligned16(x) || !aligned16(y) || (bias && !aligned16(bias))) return TAMD_E_ALIGN;
  hipStream_t s = TAMD_STREAM(stream);
  const void* dy = nullptr;
  void* out = y;
  TAMD_DISPATCH_DTYPE(dtype, {
    constexpr int VE = vec16<T>::N;
    if (cols % VE != 0) return TAMD_E_SHAPE;
    dim3 grid(stream_grid(rows * (cols / VE), 256)), block(256);
    TAMD_ACT_SWITCH(false)
  });
  return launch_status();
}

#define TAMD_BC(A_)                                                                                                    \
  hipLaunchKernelGGL((bias_act_bwd_colsum_kernel<T, A_>), grid, block, 0, s, (const T*)x, (const T*)bias, (const T*)dy, \
                     (T*)out, part, rows, (int)cols, rows_per_slab);                                                   \
  break;
int tamd_bias_act_bwd(const void* x, const void* bias, const void* dy, void* dx, void* dbias, void* workspace,
                      size_t workspace_bytes, int64_t rows, int64_t cols, int act, int dtype, tamd_stream_t stream) {
  if (!x || !dy || !dx) return TAMD_E_NULL;
  if (rows <= 0 || cols <= 0) return TAMD_OK;
  if (!aligned16(x) || !aligned16(dy) || !aligned16(dx) || (bias && !aligned16(bias))) return TAMD_E_ALIGN;
  hipStream_t s = TAMD_STREAM(stream);
  void* out = dx;
  if (dbias != nullptr) {  // + the column sums of dx (tamd_colsum's workspace and partial layout)
    if (!workspace) return TAMD_E_NULL;
    if (workspace_bytes < tamd_colsum_workspace_bytes(rows, cols)) return TAMD_E_WORKSPACE;
    int rows_per_slab = (int)ceil_div(rows, kNormMaxPartials);
    if (rows_per_slab < 16) rows_per_slab = 16;
    const int P = (int)ceil_div(rows, rows_per_slab);
    float* part = reinterpret_cast<float*>(workspace);
    TAMD_DISPATCH_DTYPE(dtype, {
      constexpr int VE = vec16<T>::N;
      if (cols % VE != 0) return TAMD_E_SHAPE;
      dim3 grid((unsigned)ceil_div(cols / VE, 128), (unsigned)P), block(128);
      switch (act) {
        case TAMD_ACT_NONE: TAMD_BC(TAMD_ACT_NONE)
        case TAMD_ACT_GELU_ERF: TAMD_BC(TAMD_ACT_GELU_ERF)
        case TAMD_ACT_GELU_TANH: TAMD_BC(TAMD_ACT_GELU_TANH)
        case TAMD_ACT_QUICK_GELU: TAMD_BC(TAMD_ACT_QUICK_GELU)
        case TAMD_ACT_SILU: TAMD_BC(TAMD_ACT_SILU)
        default: return TAMD_E_ARG;
      }
      colsum_partials_reduce<T>(part, dbias, P, (int)cols, s);
    });
    return launch_status();
  }
  TAMD_DISPATCH_DTYPE(dtype, {
    constexpr int VE = vec16<T>::N;
    if (cols % VE != 0) return TAMD_E_SHAPE;
    dim3 grid(stream_grid(rows * (cols / VE), 256)), block(256);
    TAMD_ACT_SWITCH(true)
  });
  return launch_status();
}

int tamd_add(const void* a, const void* b, void* out, int64_t n, int dtype, tamd_stream_t stream) {
  if (!a || !b || !out) return TAMD_E_NULL;
  if (n <= 0) return TAMD_OK;
  if (!aligned16(a) || !aligned16(b) || !aligned16(out)) return TAMD_E_ALIGN;
  TAMD_DISPATCH_DTYPE(dtype, {
    constexpr int VE = vec16<T>::N;
    if (n % VE != 0) return TAMD_E_SHAPE;
    hipLaunchKernelGGL((add_kernel<T>), dim3(stream_grid(n / VE, 256)), dim3(256), 0, TAMD_STREAM(stream),
                       (const T*)a, (const T*)b, (T*)out, n / VE);
  });
  return launch_status();
}

int tamd_transpose(const void* in, void* out, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out, int dtype,
                   tamd_stream_t stream) {
  if (!in || !out) return TAMD_E_NULL;
  if (rows <= 0 || cols <= 0) return TAMD_OK;
  if (!aligned16(in) || !aligned16(out)) return TAMD_E_ALIGN;
  if (ld_in % 8 != 0 || ld_out % 8 != 0) return TAMD_E_SHAPE;
  dim3 grid((unsigned)ceil_div(cols, 64), (unsigned)ceil_div(rows, 64)), block(256);
  TAMD_DISPATCH_HALF(dtype, {
    hipLaunchKernelGGL((transpose16_kernel<T>), grid, block, 0, TAMD_STREAM(stream), (const T*)in, (T*)out, rows,
                       cols, ld_in, ld_out);
  });
  return launch_status();
}

int tamd_cross_entropy_fwd(const void* logits, const int64_t* labels, float* lse, float* row_loss, int64_t tokens,
                           int64_t vocab, int64_t ld, int64_t ignore_index, int dtype, tamd_stream_t stream) {
  if (!logits || !labels || !lse || !row_loss) return TAMD_E_NULL;
  if (tokens <= 0) return TAMD_OK;
  if (!aligned16(logits)) return TAMD_E_ALIGN;
  TAMD_DISPATCH_DTYPE(dtype, {
    if (ld % vec16<T>::N != 0) return TAMD_E_SHAPE;
    hipLaunchKernelGGL((cross_entropy_fwd_kernel<T>), dim3((unsigned)tokens), dim3(kCEThreads), 0,
                       TAMD_STREAM(stream), (const T*)logits, labels, lse, row_loss, vocab, ld, ignore_index);
  });
  return launch_status();
}

int tamd_cross_entropy_bwd(const void* logits, const int64_t* labels, const float* lse, const float* gscale,
                           void* dlogits, int64_t tokens, int64_t vocab, int64_t ld, int64_t ignore_index,
                           int dtype, tamd_stream_t stream) {
  if (!logits || !labels || !lse || !gscale || !dlogits) return TAMD_E_NULL;
  if (tokens <= 0) return TAMD_OK;
  if (!aligned16(logits) || !aligned16(dlogits)) return TAMD_E_ALIGN;
  TAMD_DISPATCH_DTYPE(dtype, {
    if (ld % vec16<T>::N != 0) return TAMD_E_SHAPE;
    hipLaunchKernelGGL((cross_entropy_bwd_kernel<T>), dim3((unsigned)tokens), dim3(kCEThreads), 0,
                       TAMD_STREAM(stream), (const T*)logits, labels, lse, gscale, (T*)dlogits, vocab, ld,
                       ignore_index);
  });
  return launch_status();
}

}  // extern "C"
