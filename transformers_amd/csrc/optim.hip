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
