// colsum.h -- second stage of every column reduction (LayerNorm / RMSNorm dw and db, bias gradients): fp32 partial rows
// [P, cols] -> out [cols] in the storage dtype.  Shared by norm.hip and elementwise.hip.
#pragma once
#include "common.h"

namespace tamd {

constexpr int kNormMaxPartials = 512;  // partial rows a first stage may produce (workspace planes are sized for it)

// partial [P, cols] fp32  ->  out [cols] T.  One workgroup per 16 columns: 64 row groups x 4 column quads, each thread
// sums P/64 rows with 16-byte loads, the row groups are combined through LDS in a fixed order (deterministic).
// (The first version walked all P rows on one thread per column: 16 workgroups and a 512-deep serial chain, 120-140 us
// per call -- a third of the bert-base step and 9 ms of the Llama-3-8B one, profiles/r02_bert_kernel_stats_before.csv.)
// Up to 3 such reductions of one shape in ONE launch (blockIdx.y = plane): a LayerNorm backward produces dw, db and the
// neighbouring dense layer's bias gradient at once, and three 5 us launches per call were 0.5 ms of a 20 ms bert-base step
// (816 launches in 8 steps, profiles/r04c_bert_kernel_stats.csv).
constexpr int kColsumCols = 16, kColsumGroups = 64;
struct ColsumPlanes {
  const float* part[3];
  void* out[3];
};
template <typename T>
__global__ __launch_bounds__(256) void colsum_f32_kernel(ColsumPlanes pl, int P, int cols) {
  __shared__ float sm[kColsumGroups][kColsumCols + 1];
  const float* __restrict__ part = pl.part[blockIdx.y];
  T* __restrict__ out = reinterpret_cast<T*>(pl.out[blockIdx.y]);
  const int cq = threadIdx.x & 3, rg = threadIdx.x >> 2;
  const int col = blockIdx.x * kColsumCols + cq * 4;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < cols) {  // cols % 4 == 0 (16-byte vectors of the storage type): the quad is valid as a whole
    for (int p = rg; p < P; p += kColsumGroups) {
      const u32x4 v = ld16(part + (int64_t)p * cols + col);
      a0 += u32_as_f32(v[0]);
      a1 += u32_as_f32(v[1]);
      a2 += u32_as_f32(v[2]);
      a3 += u32_as_f32(v[3]);
    }
  }
  sm[rg][cq * 4 + 0] = a0;
  sm[rg][cq * 4 + 1] = a1;
  sm[rg][cq * 4 + 2] = a2;
  sm[rg][cq * 4 + 3] = a3;
  block_sync();
  if (threadIdx.x < kColsumCols) {
    const int c = blockIdx.x * kColsumCols + (int)threadIdx.x;
    if (c < cols) {
      float s = 0.f;
#pragma unroll 8
      for (int g = 0; g < kColsumGroups; ++g) s += sm[g][threadIdx.x];
      reinterpret_cast<typename elem<T>::raw*>(out)[c] = elem<T>::from_f32(s);
    }
  }
}

// planes with a null `out` are skipped
template <typename T>
static inline void colsum_partials_reduce3(const float* p0, void* o0, const float* p1, void* o1, const float* p2, void* o2,
                                           int P, int cols, hipStream_t s) {
  ColsumPlanes pl;
  int n = 0;
  const float* ps[3] = {p0, p1, p2};
  void* os[3] = {o0, o1, o2};
  for (int i = 0; i < 3; ++i) {
    pl.part[i] = nullptr, pl.out[i] = nullptr;
    if (os[i] != nullptr && ps[i] != nullptr) pl.part[n] = ps[i], pl.out[n] = os[i], ++n;
  }
  if (n == 0) return;
  dim3 g2((unsigned)ceil_div(cols, kColsumCols), (unsigned)n), b2(256);
  hipLaunchKernelGGL((colsum_f32_kernel<T>), g2, b2, 0, s, pl, P, cols);
}
template <typename T>
static inline void colsum_partials_reduce(const float* part, void* out, int P, int cols, hipStream_t s) {
  colsum_partials_reduce3<T>(part, out, nullptr, nullptr, nullptr, nullptr, P, cols, s);
}

}  // namespace tamd
