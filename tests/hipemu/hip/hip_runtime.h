// Fake <hip/hip_runtime.h> for the CPU kernel emulator (TEST INFRASTRUCTURE, see ../hipemu.h).
// Maps the HIP language surface the tamd kernels use onto the fiber engine.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>

#include "../hipemu.h"

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__
#endif

using hipemu::dim3;
typedef void* hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipemu::g_fail.exchange(0) ? 719 : hipSuccess; }

#define threadIdx (hipemu::cur_fiber().tid)
#define blockIdx (hipemu::g_blk->bid)
#define blockDim (hipemu::g_blk->bdim)
#define gridDim (hipemu::g_blk->gdim)

#define __builtin_amdgcn_sched_group_barrier(mask, n, id) ((void)0)
inline void __syncthreads() { hipemu::syncthreads(); }

#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) \
  hipemu::launch([=]() { kernel(__VA_ARGS__); }, (grid), (block), (smem))
// (launches are synchronous in the model, so "async" memory operations are plain ones)
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
  memset(p, v, n);
  return hipSuccess;
}

// ---- math
inline float __expf(float x) { return std::exp(x); }
inline float __logf(float x) { return std::log(x); }
inline float __log2f(float x) { return std::log2(x); }
inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
// erff / tanhf / fmaxf / fminf / exp2f / INFINITY come from <cmath> in the global namespace

// ---- atomics (blocks may run on several OS threads)
inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline float atomicAdd(float* p, float v) {
  unsigned int* up = reinterpret_cast<unsigned int*>(p);
  unsigned int old = __atomic_load_n(up, __ATOMIC_RELAXED);
  for (;;) {
    float f;
    memcpy(&f, &old, 4);
    f += v;
    unsigned int nw;
    memcpy(&nw, &f, 4);
    if (__atomic_compare_exchange_n(up, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
      memcpy(&f, &old, 4);
      return f;
    }
  }
}
