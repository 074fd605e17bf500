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
