// tamd_types.h -- storage types, vector types and pack/unpack helpers shared by
// the gfx950 kernels (pure C++ over clang vector extensions; no hardware builtins).
// Included through <tamd_device.h>.
//
// Hardware model assumed here (MI355X_MICROARCH.md / cdna_hip_programming.md):
//   * wave = 64 lanes, 4 SIMDs per CU, 256 CUs in 8 XCDs
//   * v_mfma_f32_32x32x16_{bf16,f16}: A lane l holds A[l&31][8*(l>>5)..+7],
//     B lane l holds B[8*(l>>5)..+7][l&31], C/D lane l reg r holds
//     C[(r&3)+8*(r>>2)+4*(l>>5)][l&31]
//   * v_mfma_f32_16x16x32_{bf16,f16}: A lane l holds A[l&15][8*(l>>4)..+7],
//     B likewise, C/D lane l reg r holds C[4*(l>>4)+r][l&15]
//   * ds_read_b64_tr_b16: inside each 16-lane group, lane i receives element
//     (i&3) of the 8 bytes addressed by lane 4*j+(i>>2), for j = 0..3
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tamd {

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

constexpr int kWave = 64;

// ---------------------------------------------------------------- scalars
struct bf16_t {  // storage type: 16 raw bits, round-to-nearest-even conversion
  unsigned short bits;
};
struct f16_t {
  unsigned short bits;
};

// NOTE: never write __builtin_bit_cast(T, vec[i]) on an ext_vector element: clang (ROCm 7.2, host and
// device) evaluates it on the address of element 0.  Pass the element BY VALUE through these helpers.
__device__ __forceinline__ float u32_as_f32(unsigned int u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned int f32_as_u32(float f) { return __builtin_bit_cast(unsigned int, f); }

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
  return __builtin_bit_cast(float, (unsigned int)b << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  return __builtin_bit_cast(unsigned short, (__bf16)f);  // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ float f16_bits_to_f32(unsigned short b) {
  return (float)__builtin_bit_cast(_Float16, b);
}
__device__ __forceinline__ unsigned short f32_to_f16_bits(float f) {
  return __builtin_bit_cast(unsigned short, (_Float16)f);
}

// Element-type traits used by the templated kernels.  T is a *storage* tag.
template <typename T>
struct elem;
template <>
struct elem<bf16_t> {
  typedef unsigned short raw;
  typedef bf16x8 mfma8;
  static __device__ __forceinline__ float to_f32(raw b) { return bf16_bits_to_f32(b); }
  static __device__ __forceinline__ raw from_f32(float f) { return f32_to_bf16_bits(f); }
};
template <>
struct elem<f16_t> {
  typedef unsigned short raw;
  typedef f16x8 mfma8;
  static __device__ __forceinline__ float to_f32(raw b) { return f16_bits_to_f32(b); }
  static __device__ __forceinline__ raw from_f32(float f) { return f32_to_f16_bits(f); }
};
template <>
struct elem<float> {
  typedef float raw;
  static __device__ __forceinline__ float to_f32(raw b) { return b; }
  static __device__ __forceinline__ raw from_f32(float f) { return f; }
};

// Round an fp32 value through the storage type (models the reference's
// intermediate `.to(dtype)` roundings so fused kernels stay bit-faithful).
template <typename T>
__device__ __forceinline__ float round_through(float f) {
  return elem<T>::to_f32(elem<T>::from_f32(f));
}
template <>
__device__ __forceinline__ float round_through<float>(float f) {
  return f;
}

// ---------------------------------------------------------------- vector I/O
// 16-byte global accesses: 8 x 16-bit elements or 4 x fp32 per lane.
template <typename T>
struct vec16;  // number of elements of T in 16 bytes + load/store helpers
template <>
struct vec16<bf16_t> {
  static constexpr int N = 8;
};
template <>
struct vec16<f16_t> {
  static constexpr int N = 8;
};
template <>
struct vec16<float> {
  static constexpr int N = 4;
};

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
// streaming store (global_store_dwordx4 ... nt): output nobody reads again soon (activations saved for the backward)
__device__ __forceinline__ void st16_nt(void* p, u32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); }
__device__ __forceinline__ u32x2 ld8(const void* p) { return *reinterpret_cast<const u32x2*>(p); }
__device__ __forceinline__ void st8(void* p, u32x2 v) { *reinterpret_cast<u32x2*>(p) = v; }

// unpack a 16-byte register quad into N floats / pack back
template <typename T>
__device__ __forceinline__ void unpack16(u32x4 v, float* out);
template <>
__device__ __forceinline__ void unpack16<bf16_t>(u32x4 v, float* out) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned int w = v[i];
    out[2 * i] = u32_as_f32(w << 16);
    out[2 * i + 1] = u32_as_f32(w & 0xffff0000u);
  }
}
template <>
__device__ __forceinline__ void unpack16<f16_t>(u32x4 v, float* out) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[2 * i] = f16_bits_to_f32((unsigned short)(v[i] & 0xffffu));
    out[2 * i + 1] = f16_bits_to_f32((unsigned short)(v[i] >> 16));
  }
}
template <>
__device__ __forceinline__ void unpack16<float>(u32x4 v, float* out) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned int w = v[i];
    out[i] = u32_as_f32(w);
  }
}
template <typename T>
__device__ __forceinline__ u32x4 pack16(const float* in);
template <>
__device__ __forceinline__ u32x4 pack16<bf16_t>(const float* in) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    v[i] = (unsigned int)f32_to_bf16_bits(in[2 * i]) | ((unsigned int)f32_to_bf16_bits(in[2 * i + 1]) << 16);
  return v;
}
template <>
__device__ __forceinline__ u32x4 pack16<f16_t>(const float* in) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    v[i] = (unsigned int)f32_to_f16_bits(in[2 * i]) | ((unsigned int)f32_to_f16_bits(in[2 * i + 1]) << 16);
  return v;
}
template <>
__device__ __forceinline__ u32x4 pack16<float>(const float* in) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = f32_as_u32(in[i]);
  return v;
}

__device__ __forceinline__ unsigned int pack2_bf16(float lo, float hi) {
  return (unsigned int)f32_to_bf16_bits(lo) | ((unsigned int)f32_to_bf16_bits(hi) << 16);
}
template <typename T>
__device__ __forceinline__ unsigned int pack2(float lo, float hi) {
  return (unsigned int)elem<T>::from_f32(lo) | ((unsigned int)elem<T>::from_f32(hi) << 16);
}
// two fp32 -> one packed register in ONE v_cvt_pk_{bf16,f16}_f32 (vector conversion; the scalar form above
// costs a convert per element plus shifts/ors)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
template <>
__device__ __forceinline__ unsigned int pack2<bf16_t>(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}
template <>
__device__ __forceinline__ unsigned int pack2<f16_t>(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, f16x2_t));
}

// the two 16-bit elements of a packed register as fp32 (the inverse of pack2: what a value rounded through T reads back as)
template <typename T>
__device__ __forceinline__ void unpack2(unsigned int w, float& lo, float& hi) {
  lo = elem<T>::to_f32((typename elem<T>::raw)(w & 0xffffu));
  hi = elem<T>::to_f32((typename elem<T>::raw)(w >> 16));
}
template <>
__device__ __forceinline__ void unpack2<bf16_t>(unsigned int w, float& lo, float& hi) {
  lo = __builtin_bit_cast(float, w << 16);
  hi = __builtin_bit_cast(float, w & 0xffff0000u);
}

}  // namespace tamd
