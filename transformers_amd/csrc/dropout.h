// dropout.h -- the counter-based dropout mask shared by the attention kernels (probability dropout inside the flash
// kernels) and the norm kernels (hidden-state dropout of the post-LN BERT blocks, models/bert/modeling_bert.py:289-293).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tamd {

// Counter-based dropout mask: 32-bit mix of (seed, element index); identical in forward, backward and on the host
// (tamd_dropout_hash).  Attention: index = ((b*Hq + h)*Sq + q)*Sk + k; hidden states: index = row*cols + col.
__host__ __device__ __forceinline__ unsigned dropout_hash(unsigned seed_lo, unsigned seed_hi, unsigned idx_lo,
                                                          unsigned idx_hi) {
  unsigned x = (idx_lo ^ seed_lo) * 0x9E3779B1u;
  x ^= x >> 15;
  x += (idx_hi * 0x85EBCA77u) ^ seed_hi;
  x *= 0xC2B2AE3Du;
  x ^= x >> 13;
  x *= 0x27D4EB2Fu;
  x ^= x >> 16;
  return x;
}
// keep-scale of element (row_index*Sk + k): 0 or 1/(1-p)
struct DropCtx {
  unsigned thr, seed_lo, seed_hi;
  float scale;
  __device__ __forceinline__ float factor(unsigned long long base, int k) const {
    const unsigned long long idx = base + (unsigned long long)k;
    return dropout_hash(seed_lo, seed_hi, (unsigned)idx, (unsigned)(idx >> 32)) >= thr ? scale : 0.f;
  }
};

// Attention dropout: ONE hash decides a 2 x 2 block of the probability matrix (query pair x key pair), 16 bits per element
// -- the mix above costs 13 integer instructions, and the attention loops are bound by instruction issue (at bert-base the
// per-element hash cost more than the attention itself: 116 / 131 / 154 us per layer against 38 / 131 for forward /
// backward without dropout, profiles/r03c_bert_kernel_stats.csv).  Element (b, h, q, k), bh = b * heads_q + h:
//     block = (bh * ceil(Sq / 2) + (q >> 1)) * ceil(Sk / 2) + (k >> 1)
//     w0 = dropout_hash(seed, block),  w1 = dropout_hash_second(w0);   word = (k & 1) ? w1 : w0
//     field = (q & 1) ? word >> 16 : word & 0xffff;                     kept iff field >= thr16 = floor(p * 65536)
// The forward and dQ kernels own one query row and consecutive key pairs per lane (both words, one half each), the dK/dV
// kernel one key and consecutive query pairs (one word, both halves): every kernel halves its hash count.
__host__ __device__ __forceinline__ unsigned dropout_hash_second(unsigned x) {
  unsigned y = (x ^ 0x85EBCA77u) * 0x9E3779B1u;
  y ^= y >> 15;
  y *= 0xC2B2AE3Du;
  y ^= y >> 16;
  return y;
}
__host__ __device__ __forceinline__ unsigned attn_dropout_field(unsigned seed_lo, unsigned seed_hi, unsigned long long bh,
                                                                unsigned long long seq_q, unsigned long long seq_k,
                                                                unsigned long long q, unsigned long long k) {
  const unsigned long long blk = (bh * ((seq_q + 1) >> 1) + (q >> 1)) * ((seq_k + 1) >> 1) + (k >> 1);
  const unsigned w0 = dropout_hash(seed_lo, seed_hi, (unsigned)blk, (unsigned)(blk >> 32));
  const unsigned w = (k & 1) ? dropout_hash_second(w0) : w0;
  return (q & 1) ? (w >> 16) : (w & 0xffffu);
}
struct AttnDrop {
  unsigned thr16, seed_lo, seed_hi;
  float scale;
  unsigned long long csq, csk;  // ceil(seq_q / 2), ceil(seq_k / 2)
  // block index of (bh, query q, key 0) -- add (k >> 1)
  __device__ __forceinline__ unsigned long long row_base(unsigned long long bh, unsigned long long q) const {
    return (bh * csq + (q >> 1)) * csk;
  }
  __device__ __forceinline__ void words(unsigned long long blk, unsigned& w0, unsigned& w1) const {
    w0 = dropout_hash(seed_lo, seed_hi, (unsigned)blk, (unsigned)(blk >> 32));
    w1 = dropout_hash_second(w0);
  }
  __device__ __forceinline__ float keep(unsigned field) const { return field >= thr16 ? scale : 0.f; }
};

// keep threshold / scale of a dropout probability (0 -> disabled)
struct DropParams {
  unsigned thr;
  float scale;
};
inline DropParams drop_params(double p) {
  DropParams d;
  d.thr = (p > 0.0) ? (unsigned)(p >= 1.0 ? 4294967295.0 : p * 4294967296.0) : 0u;
  d.scale = (p > 0.0 && p < 1.0) ? (float)(1.0 / (1.0 - p)) : 1.f;
  return d;
}

}  // namespace tamd
