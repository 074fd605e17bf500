// optim.hip -- fused AdamW step ("fused optimizer step").
//
// Reference: the optimizer `Trainer` builds by default is torch.optim.AdamW (src/transformers/trainer.py:1783-1799,
// training_args.py `optim="adamw_torch"` / `"adamw_torch_fused"`); its update rule (torch/optim/adam.py,
// `_single_tensor_adam` with decoupled weight decay) is, per element,
//     p   <- p * (1 - lr*wd)
//     m   <- m + (1 - b1) * (g - m)                       (lerp)
//     v   <- b2*v + (1 - b2) * g*g
//     p   <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// Eager torch runs that as ~10 elementwise kernels over four tensors (and rounds to the storage dtype after every
// one); this kernel streams p, g, m, v once (16 B per lane per access), does the arithmetic in fp32 and rounds each
// stored tensor once -- the semantics of torch's own fused implementation (`fused=True`).
// HBM-bound: algorithmic bytes = (4 reads + 3 writes) x element size.
#include <math.h>

#include "common.h"

namespace tamd {

// VE consecutive elements of T <-> fp32 registers with one 16-byte (or, for 4 x 16-bit, one 8-byte) access
template <typename T, int VE>
__device__ __forceinline__ void load_vec(const T* ptr, float* out) {
  if (sizeof(typename elem<T>::raw) == 4 || VE == 8) {
    unpack16<T>(ld16(ptr), out);
  } else {
    const u32x2 q = ld8(ptr);
    typedef typename elem<T>::raw raw;
    out[0] = elem<T>::to_f32((raw)(q[0] & 0xffffu));
    out[1] = elem<T>::to_f32((raw)(q[0] >> 16));
    out[2] = elem<T>::to_f32((raw)(q[1] & 0xffffu));
    out[3] = elem<T>::to_f32((raw)(q[1] >> 16));
  }
}
template <typename T, int VE>
__device__ __forceinline__ void store_vec(T* ptr, const float* in) {
  if (sizeof(typename elem<T>::raw) == 4 || VE == 8) {
    st16(ptr, pack16<T>(in));
  } else {
    st8(ptr, u32x2{pack2<T>(in[0], in[1]), pack2<T>(in[2], in[3])});
  }
}

template <typename T, typename S>
__global__ void adamw_kernel(T* __restrict__ p, const T* __restrict__ g, S* __restrict__ m, S* __restrict__ v,
                             int64_t n, float decay, float b1, float b2, float step_size, float inv_bc2_sqrt,
                             float eps, float grad_scale) {
  // VE elements per thread: one 16-byte access of the wider of the two storage types
  constexpr int VE = vec16<T>::N < vec16<S>::N ? vec16<T>::N : vec16<S>::N;
  for (int64_t idx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VE; idx < n;
       idx += (int64_t)gridDim.x * blockDim.x * VE) {
    float pp[VE], gg[VE], mm[VE], vv[VE];
    load_vec<T, VE>(p + idx, pp);
    load_vec<T, VE>(g + idx, gg);
    load_vec<S, VE>(m + idx, mm);
    load_vec<S, VE>(v + idx, vv);
#pragma unroll
    for (int i = 0; i < VE; ++i) {
      const float gi = gg[i] * grad_scale;
      const float pd = pp[i] * decay;
      const float mn = mm[i] + (1.f - b1) * (gi - mm[i]);
      const float vn = b2 * vv[i] + (1.f - b2) * gi * gi;
      const float denom = sqrtf(vn) * inv_bc2_sqrt + eps;
      pp[i] = pd - step_size * (mn / denom);
      mm[i] = mn;
      vv[i] = vn;
    }
    store_vec<T, VE>(p + idx, pp);
    store_vec<S, VE>(m + idx, mm);
    store_vec<S, VE>(v + idx, vv);
  }
}

__device__ __forceinline__ float load1(const bf16_t* q) { return bf16_bits_to_f32(q->bits); }
__device__ __forceinline__ float load1(const f16_t* q) { return f16_bits_to_f32(q->bits); }
__device__ __forceinline__ float load1(const float* q) { return *q; }
__device__ __forceinline__ void store1(bf16_t* q, float f) { q->bits = f32_to_bf16_bits(f); }
__device__ __forceinline__ void store1(f16_t* q, float f) { q->bits = f32_to_f16_bits(f); }
__device__ __forceinline__ void store1(float* q, float f) { *q = f; }

// elements [start, n) one per thread: the ragged tail (n % VE) of a tensor, or a whole tensor whose storage is not
// 16-byte aligned (scalar parameters such as CLIP's logit_scale, a 2- or 3-label classifier bias)
template <typename T, typename S>
__global__ void adamw_scalar_kernel(T* __restrict__ p, const T* __restrict__ g, S* __restrict__ m, S* __restrict__ v,
                                    int64_t start, int64_t n, float decay, float b1, float b2, float step_size,
                                    float inv_bc2_sqrt, float eps, float grad_scale) {
  for (int64_t idx = start + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const float gi = load1(g + idx) * grad_scale;
    const float pd = load1(p + idx) * decay;
    const float m0 = load1(m + idx), v0 = load1(v + idx);
    const float mn = m0 + (1.f - b1) * (gi - m0);
    const float vn = b2 * v0 + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vn) * inv_bc2_sqrt + eps;
    store1(p + idx, pd - step_size * (mn / denom));
    store1(m + idx, mn);
    store1(v + idx, vn);
  }
}

template <typename T, typename S>
static int adamw_launch(void* p, const void* g, void* m, void* v, int64_t n, float decay, float b1, float b2,
                        float step_size, float inv_bc2_sqrt, float eps, float grad_scale, hipStream_t s) {
  constexpr int VE = vec16<T>::N < vec16<S>::N ? vec16<T>::N : vec16<S>::N;
  const bool vec_ok = aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v);
  const int64_t n_vec = vec_ok ? n - n % VE : 0;  // streamed 16 bytes per lane; the rest one element per thread
  if (n_vec > 0) {
    int64_t blocks = ceil_div(n_vec / VE, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride beyond 16 workgroups per CU
    hipLaunchKernelGGL((adamw_kernel<T, S>), dim3((unsigned)blocks), dim3(256), 0, s, (T*)p, (const T*)g, (S*)m, (S*)v,
                       n_vec, decay, b1, b2, step_size, inv_bc2_sqrt, eps, grad_scale);
  }
  if (n_vec < n) {
    int64_t blocks = ceil_div(n - n_vec, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL((adamw_scalar_kernel<T, S>), dim3((unsigned)blocks), dim3(256), 0, s, (T*)p, (const T*)g, (S*)m,
                       (S*)v, n_vec, n, decay, b1, b2, step_size, inv_bc2_sqrt, eps, grad_scale);
  }
  return launch_status();
}


// ============================================================================================ multi-tensor launches
// Trainer clips the global gradient norm before every optimizer step (training_args.py:856 `max_grad_norm = 1.0`;
// trainer.py:2538-2548 -> accelerate -> torch.nn.utils.clip_grad_norm_): per-tensor norms, a norm of norms, a host-visible
// coefficient and one scaling pass over every gradient -- three more trips over 16 GB for the 8B model -- and then one
// optimizer launch per parameter (291 of them).  Here the whole parameter set of one dtype is ONE table in device memory and
// the step is
//     mt_sumsq_kernel    one workgroup per 64 Ki-element chunk of some gradient -> one fp32 partial per chunk (fixed order:
//                        the result is deterministic, no atomics)
//     mt_norm_finish     one workgroup: norm = sqrt(sum of partials), coef = min(1, max_norm / (norm + 1e-6)) -- the
//                        reference's clamp -- left in device memory (out[0], out[1]); nobody reads it back on the host
//     mt_adamw_kernel    one launch for every tensor of the table; the gradient is scaled by grad_scale * *grad_scale_dev
//                        in registers (the clipped gradient never goes back to HBM)
// (mt_scale_kernel scales the gradients in place, for callers that want torch.nn.utils.clip_grad_norm_'s side effect.)
// Table (int64 words, device memory), n tensors:  [0,n) p | [n,2n) g | [2n,3n) m | [3n,4n) v | [4n,5n) numel |
// [5n,6n] first chunk of tensor i (prefix sum of ceil(numel / kMtChunk)); word 6n = total chunks.
constexpr int kMtChunk = TAMD_MT_CHUNK;
constexpr int kMtThreads = 256;

struct MtSlot {
  int tensor;
  int64_t first, count;  // element range of this workgroup's chunk inside the tensor
};
// which tensor does chunk `c` belong to: the last i with start[i] <= c (wave-uniform: scalar loads)
__device__ __forceinline__ MtSlot mt_slot(const int64_t* __restrict__ table, int n, int64_t c) {
  const int64_t* start = table + 5 * (int64_t)n;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (start[mid] <= c)
      lo = mid;
    else
      hi = mid - 1;
  }
  const int64_t numel = table[4 * (int64_t)n + lo];
  MtSlot s;
  s.tensor = lo;
  s.first = (c - start[lo]) * kMtChunk;
  s.count = numel - s.first < kMtChunk ? numel - s.first : kMtChunk;
  return s;
}

__device__ __forceinline__ float block_sum(float v, float* red) {  // kMtThreads threads; result in every thread
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < kMtThreads / 64; ++w) t += red[w];
  return t;
}

template <typename T>
__global__ __launch_bounds__(kMtThreads) void mt_sumsq_kernel(const int64_t* __restrict__ table, int n,
                                                              float* __restrict__ partials) {
  constexpr int VE = vec16<T>::N;
  __shared__ float red[kMtThreads / 64];
  const int64_t c = blockIdx.x;
  const MtSlot s = mt_slot(table, n, c);
  const T* g = reinterpret_cast<const T*>(table[(int64_t)n + s.tensor]) + s.first;
  float acc = 0.f;
  if (s.count > 0) {
    const int64_t n_vec = ((reinterpret_cast<uintptr_t>(g) & 15u) == 0) ? s.count - s.count % VE : 0;
    for (int64_t i = (int64_t)threadIdx.x * VE; i < n_vec; i += (int64_t)kMtThreads * VE) {
      float gg[VE];
      unpack16<T>(ld16(g + i), gg);
#pragma unroll
      for (int e = 0; e < VE; ++e) acc += gg[e] * gg[e];
    }
    for (int64_t i = n_vec + threadIdx.x; i < s.count; i += kMtThreads) {
      const float x = load1(g + i);
      acc += x * x;
    }
  }
  const float t = block_sum(acc, red);
  if (threadIdx.x == 0) partials[c] = t;
}

__global__ __launch_bounds__(1024) void mt_norm_finish_kernel(const float* __restrict__ partials, int64_t count,
                                                              float* __restrict__ out, float max_norm) {
  __shared__ double red[1024 / 64];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < count; i += 1024) acc += (double)partials[i];
  // wave sum of a double: two 32-bit halves do not add -- go through LDS per wave instead
  __shared__ double lanes[1024];
  lanes[threadIdx.x] = acc;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    double t = 0.0;
    for (int l = 0; l < 64; ++l) t += lanes[wave * 64 + l];
    red[wave] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 1024 / 64; ++w) t += red[w];
    const float norm = (float)sqrt(t);
    out[0] = norm;
    // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1.0; max_norm <= 0: no clipping
    // (a non-finite norm gives a NaN coefficient, as in the reference with error_if_nonfinite=False)
    float coef = 1.f;
    if (max_norm > 0.f) {
      coef = max_norm / (norm + 1e-6f);
      if (coef > 1.f) coef = 1.f;
    }
    out[1] = coef;
  }
}

template <typename T>
__global__ __launch_bounds__(kMtThreads) void mt_scale_kernel(const int64_t* __restrict__ table, int n,
                                                              const float* __restrict__ coef_dev) {
  constexpr int VE = vec16<T>::N;
  const float coef = *coef_dev;
  if (coef == 1.f) return;  // (the reference multiplies by the clamped 1.0: a no-op, bit for bit)
  const MtSlot s = mt_slot(table, n, blockIdx.x);
  T* g = reinterpret_cast<T*>(table[(int64_t)n + s.tensor]) + s.first;
  if (s.count <= 0) return;
  const int64_t n_vec = ((reinterpret_cast<uintptr_t>(g) & 15u) == 0) ? s.count - s.count % VE : 0;
  for (int64_t i = (int64_t)threadIdx.x * VE; i < n_vec; i += (int64_t)kMtThreads * VE) {
    float gg[VE];
    unpack16<T>(ld16(g + i), gg);
#pragma unroll
    for (int e = 0; e < VE; ++e) gg[e] *= coef;
    st16(g + i, pack16<T>(gg));
  }
  for (int64_t i = n_vec + threadIdx.x; i < s.count; i += kMtThreads) store1(g + i, load1(g + i) * coef);
}

template <typename T, typename S>
__global__ __launch_bounds__(kMtThreads) void mt_adamw_kernel(const int64_t* __restrict__ table, int n, float decay, float b1,
                                                              float b2, float step_size, float inv_bc2_sqrt, float eps,
                                                              float grad_scale, const float* __restrict__ grad_scale_dev) {
  constexpr int VE = vec16<T>::N < vec16<S>::N ? vec16<T>::N : vec16<S>::N;
  const MtSlot s = mt_slot(table, n, blockIdx.x);
  if (s.count <= 0) return;
  const int64_t nn = n;
  T* p = reinterpret_cast<T*>(table[s.tensor]) + s.first;
  const T* g = reinterpret_cast<const T*>(table[nn + s.tensor]) + s.first;
  S* m = reinterpret_cast<S*>(table[2 * nn + s.tensor]) + s.first;
  S* v = reinterpret_cast<S*>(table[3 * nn + s.tensor]) + s.first;
  const float gs = grad_scale_dev != nullptr ? grad_scale * *grad_scale_dev : grad_scale;
  const uintptr_t mis = reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                        reinterpret_cast<uintptr_t>(v);
  const int64_t n_vec = (mis & 15u) == 0 ? s.count - s.count % VE : 0;
  for (int64_t idx = (int64_t)threadIdx.x * VE; idx < n_vec; idx += (int64_t)kMtThreads * VE) {
    float pp[VE], gg[VE], mm[VE], vv[VE];
    load_vec<T, VE>(p + idx, pp);
    load_vec<T, VE>(g + idx, gg);
    load_vec<S, VE>(m + idx, mm);
    load_vec<S, VE>(v + idx, vv);
#pragma unroll
    for (int i = 0; i < VE; ++i) {  // (the arithmetic of adamw_kernel, operation for operation)
      const float gi = gg[i] * gs;
      const float pd = pp[i] * decay;
      const float mn = mm[i] + (1.f - b1) * (gi - mm[i]);
      const float vn = b2 * vv[i] + (1.f - b2) * gi * gi;
      const float denom = sqrtf(vn) * inv_bc2_sqrt + eps;
      pp[i] = pd - step_size * (mn / denom);
      mm[i] = mn;
      vv[i] = vn;
    }
    store_vec<T, VE>(p + idx, pp);
    store_vec<S, VE>(m + idx, mm);
    store_vec<S, VE>(v + idx, vv);
  }
  for (int64_t idx = n_vec + threadIdx.x; idx < s.count; idx += kMtThreads) {
    const float gi = load1(g + idx) * gs;
    const float pd = load1(p + idx) * decay;
    const float m0 = load1(m + idx), v0 = load1(v + idx);
    const float mn = m0 + (1.f - b1) * (gi - m0);
    const float vn = b2 * v0 + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vn) * inv_bc2_sqrt + eps;
    store1(p + idx, pd - step_size * (mn / denom));
    store1(m + idx, mn);
    store1(v + idx, vn);
  }
}

}  // namespace tamd

using namespace tamd;

extern "C" int tamd_adamw_step(void* p, const void* g, void* m, void* v, int64_t n, double lr, double beta1,
                               double beta2, double eps, double weight_decay, int64_t step, double grad_scale,
                               int dtype, int state_dtype, tamd_stream_t stream) {
  if (!p || !g || !m || !v) return TAMD_E_NULL;
  if (n <= 0) return TAMD_OK;
  if (step < 1 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return TAMD_E_ARG;
  // bias corrections in double on the host, as torch does (torch/optim/adam.py: python floats)
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  const float decay = (float)(1.0 - lr * weight_decay);
  hipStream_t s = TAMD_STREAM(stream);
#define TAMD_ADAMW(T_, S_)                                                                                        \
  return adamw_launch<T_, S_>(p, g, m, v, n, decay, (float)beta1, (float)beta2, step_size, inv_bc2_sqrt, (float)eps, \
                              (float)grad_scale, s)
  if (dtype == TAMD_BF16 && state_dtype == TAMD_BF16) TAMD_ADAMW(bf16_t, bf16_t);
  if (dtype == TAMD_BF16 && state_dtype == TAMD_F32) TAMD_ADAMW(bf16_t, float);
  if (dtype == TAMD_F16 && state_dtype == TAMD_F16) TAMD_ADAMW(f16_t, f16_t);
  if (dtype == TAMD_F16 && state_dtype == TAMD_F32) TAMD_ADAMW(f16_t, float);
  if (dtype == TAMD_F32 && state_dtype == TAMD_F32) TAMD_ADAMW(float, float);
#undef TAMD_ADAMW
  return TAMD_E_DTYPE;
}

extern "C" int tamd_mt_sumsq(const int64_t* table, int n_tensors, int64_t total_chunks, float* partials, int dtype,
                             tamd_stream_t stream) {
  if (n_tensors <= 0 || total_chunks <= 0) return TAMD_OK;
  if (!table || !partials) return TAMD_E_NULL;
  if (total_chunks > 0x7fffffffLL) return TAMD_E_ARG;
  hipStream_t s = TAMD_STREAM(stream);
  TAMD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((mt_sumsq_kernel<T>), dim3((unsigned)total_chunks), dim3(kMtThreads), 0, s,
                                                table, n_tensors, partials));
  return launch_status();
}

extern "C" int tamd_mt_norm_finish(const float* partials, int64_t count, float* out, double max_norm, tamd_stream_t stream) {
  if (!out || (count > 0 && !partials)) return TAMD_E_NULL;
  if (count < 0) return TAMD_E_ARG;
  hipLaunchKernelGGL(mt_norm_finish_kernel, dim3(1), dim3(1024), 0, TAMD_STREAM(stream), partials, count, out,
                     (float)max_norm);
  return launch_status();
}

extern "C" int tamd_mt_scale(const int64_t* table, int n_tensors, int64_t total_chunks, const float* coef, int dtype,
                             tamd_stream_t stream) {
  if (n_tensors <= 0 || total_chunks <= 0) return TAMD_OK;
  if (!table || !coef) return TAMD_E_NULL;
  if (total_chunks > 0x7fffffffLL) return TAMD_E_ARG;
  hipStream_t s = TAMD_STREAM(stream);
  TAMD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((mt_scale_kernel<T>), dim3((unsigned)total_chunks), dim3(kMtThreads), 0, s,
                                                table, n_tensors, coef));
  return launch_status();
}

extern "C" int tamd_mt_adamw_step(const int64_t* table, int n_tensors, int64_t total_chunks, double lr, double beta1,
                                  double beta2, double eps, double weight_decay, int64_t step, double grad_scale,
                                  const float* grad_scale_dev, int dtype, int state_dtype, tamd_stream_t stream) {
  if (n_tensors <= 0 || total_chunks <= 0) return TAMD_OK;
  if (!table) return TAMD_E_NULL;
  if (total_chunks > 0x7fffffffLL) return TAMD_E_ARG;
  if (step < 1 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return TAMD_E_ARG;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  const float decay = (float)(1.0 - lr * weight_decay);
  hipStream_t s = TAMD_STREAM(stream);
#define TAMD_MT_ADAMW(T_, S_)                                                                                            \
  {                                                                                                                      \
    hipLaunchKernelGGL((mt_adamw_kernel<T_, S_>), dim3((unsigned)total_chunks), dim3(kMtThreads), 0, s, table, n_tensors, \
                       decay, (float)beta1, (float)beta2, step_size, inv_bc2_sqrt, (float)eps, (float)grad_scale,        \
                       grad_scale_dev);                                                                                  \
    return launch_status();                                                                                              \
  }
  if (dtype == TAMD_BF16 && state_dtype == TAMD_BF16) TAMD_MT_ADAMW(bf16_t, bf16_t);
  if (dtype == TAMD_BF16 && state_dtype == TAMD_F32) TAMD_MT_ADAMW(bf16_t, float);
  if (dtype == TAMD_F16 && state_dtype == TAMD_F16) TAMD_MT_ADAMW(f16_t, f16_t);
  if (dtype == TAMD_F16 && state_dtype == TAMD_F32) TAMD_MT_ADAMW(f16_t, float);
  if (dtype == TAMD_F32 && state_dtype == TAMD_F32) TAMD_MT_ADAMW(float, float);
#undef TAMD_MT_ADAMW
  return TAMD_E_DTYPE;
}
