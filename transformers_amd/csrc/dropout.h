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
  // non-null: the seed lives in DEVICE memory (one 64-bit word) and replaces seed_lo / seed_hi -- a train step captured in a
  // HIP graph draws a fresh mask on every replay (the word is written by a captured RNG kernel: ops.dropout_seed_tensor)
  const unsigned long long* seed_dev = nullptr;
  __device__ __forceinline__ float factor(unsigned long long base, int k) const {
    const unsigned long long idx = base + (unsigned long long)k;
    return dropout_hash(seed_lo, seed_hi, (unsigned)idx, (unsigned)(idx >> 32)) >= thr ? scale : 0.f;
  }
  // the context the kernel works with: the device seed, if any, read once (wave-uniform address: a scalar load)
  __device__ __forceinline__ DropCtx resolved() const {
    DropCtx d = *this;
    if (seed_dev != nullptr) {
      const unsigned long long s = *seed_dev;
      d.seed_lo = (unsigned)s;
      d.seed_hi = (unsigned)(s >> 32);
    }
    return d;
  }
};

// Attention dropout: ONE hash decides a 2 x 2 block of the probability matrix (query pair x key pair), 16 bits per element.
// Element (b, h, q, k), bh = b * heads_q + h:
//     block = (bh * ceil(Sq / 2) + (q >> 1)) * ceil(Sk / 2) + (k >> 1)
//     (w0, w1) = attn_block_words(mix(seed), block);   word = (k & 1) ? w1 : w0
//     field = (q & 1) ? word >> 16 : word & 0xffff;                     kept iff field >= thr16 = floor(p * 65536)
// The forward and dQ kernels own one query row and consecutive key pairs per lane (both words, one half each), the dK/dV
// kernel one key and consecutive query pairs (one word, both halves).
// Round 4: the attention loops are bound by instruction issue, and 32-bit integer multiplies issue at a quarter of the
// VALU rate on gfx950 -- the round-3 mix (3 + 2 of them per block, dropout_hash + a second word) cost ~34 issue slots per
// block; with dropout 0.1 the bert-base attention kernels ran 93 / 195 us forward / backward against 42 / 129 without
// (profiles/r04a_attn_dropout_ab.jsonl).  attn_block_words is built from v_mul_u32_u24 / v_mad_u32_u24 (full rate: the low
// 24 bits of a 32-bit word by a 24-bit constant; two of them, on bits 0..23 and 8..31, cover the word) and xor-shifts:
// ~17 slots for the 64 bits.  The 63-bit seed is mixed once on the host (splitmix64: seeds that differ in one bit give
// unrelated masks -- the block mix alone would not: its second input enters by addition).  Statistics (keep rate, row /
// column variance, neighbour, head-to-head and seed-to-seed correlation, chi-square of the fields) were checked against the
// round-3 mix: tools/probes/dropout_hash_stats.py.
__host__ __device__ __forceinline__ unsigned long long attn_seed_mix(unsigned long long seed) {
  unsigned long long z = seed;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ unsigned mul24(unsigned a, unsigned k24) { return (a & 0xFFFFFFu) * k24; }  // v_mul_u32_u24
// s0, s1: the two halves of attn_seed_mix(seed)
__host__ __device__ __forceinline__ void attn_block_words(unsigned s0, unsigned s1, unsigned blk_lo, unsigned blk_hi,
                                                          unsigned& w0, unsigned& w1) {
  unsigned x = blk_lo ^ s0;
  x = mul24(x, 0x9E3779u) + mul24(x >> 8, 0x85EBCBu) + (s1 + mul24(blk_hi, 0x632BE5u));
  x ^= x >> 15;
  x = mul24(x, 0xC2B2AFu) + mul24(x >> 8, 0x27D4EBu);
  x ^= x >> 13;
  w0 = x;
  const unsigned t = x ^ 0x165667B1u;
  unsigned z = mul24(t, 0xD3A265u) + mul24(t >> 8, 0x7F4A7Du);
  z ^= z >> 16;
  w1 = z;
}
__host__ __device__ __forceinline__ unsigned attn_dropout_field(unsigned s0, unsigned s1, unsigned long long bh,
                                                                unsigned long long seq_q, unsigned long long seq_k,
                                                                unsigned long long q, unsigned long long k) {
  const unsigned long long blk = (bh * ((seq_q + 1) >> 1) + (q >> 1)) * ((seq_k + 1) >> 1) + (k >> 1);
  unsigned w0, w1;
  attn_block_words(s0, s1, (unsigned)blk, (unsigned)(blk >> 32), w0, w1);
  const unsigned w = (k & 1) ? w1 : w0;
  return (q & 1) ? (w >> 16) : (w & 0xffffu);
}
struct AttnDrop {
  unsigned thr_hi;  // thr16 << 16: a 16-bit field f in the high half of w is kept iff w >= thr_hi, one in the low half iff
                    // (w << 16) >= thr_hi -- no field extraction
  unsigned s0, s1;  // the halves of attn_seed_mix(seed)
  float scale;      // 65536 / (65536 - thr16); the kernels apply it ONCE per output row (O, dV) or inside an fma (dS), not per element
  unsigned long long csq, csk;  // ceil(seq_q / 2), ceil(seq_k / 2)
  // block index of (bh, query q, key 0) -- add (k >> 1)
  __device__ __forceinline__ unsigned long long row_base(unsigned long long bh, unsigned long long q) const {
    return (bh * csq + (q >> 1)) * csk;
  }
  __device__ __forceinline__ void words(unsigned long long blk, unsigned& w0, unsigned& w1) const {
    attn_block_words(s0, s1, (unsigned)blk, (unsigned)(blk >> 32), w0, w1);
  }
  // is the element whose field is the low / high half of w kept?  (lsh = 16 for the low half, 0 for the high half)
  __device__ __forceinline__ bool kept(unsigned w, unsigned lsh) const { return (w << lsh) >= thr_hi; }
  __device__ __forceinline__ bool kept_lo(unsigned w) const { return (w << 16) >= thr_hi; }
  __device__ __forceinline__ bool kept_hi(unsigned w) const { return w >= thr_hi; }
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
