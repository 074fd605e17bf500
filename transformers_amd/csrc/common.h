// common.h -- host-side helpers shared by the C-ABI launchers in libtamd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tamd.h"
#include <tamd_device.h>

namespace tamd {

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? TAMD_OK : (int)e;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

inline size_t dtype_bytes(int dtype) { return dtype == TAMD_F32 ? 4 : 2; }

}  // namespace tamd

// Dispatch a templated launcher over the storage dtype.  `CALL` is an
// expression that uses the type name T.
#define TAMD_DISPATCH_DTYPE(dtype, ...)       \
  switch (dtype) {                             \
    case TAMD_BF16: {                          \
      typedef tamd::bf16_t T;                  \
      __VA_ARGS__;                                 \
    } break;                                   \
    case TAMD_F16: {                           \
      typedef tamd::f16_t T;                   \
      __VA_ARGS__;                                 \
    } break;                                   \
    case TAMD_F32: {                           \
      typedef float T;                         \
      __VA_ARGS__;                                 \
    } break;                                   \
    default:                                   \
      return TAMD_E_DTYPE;                     \
  }

#define TAMD_DISPATCH_HALF(dtype, ...)        \
  switch (dtype) {                             \
    case TAMD_BF16: {                          \
      typedef tamd::bf16_t T;                  \
      __VA_ARGS__;                                 \
    } break;                                   \
    case TAMD_F16: {                           \
      typedef tamd::f16_t T;                   \
      __VA_ARGS__;                                 \
    } break;                                   \
    default:                                   \
      return TAMD_E_DTYPE;                     \
  }

#define TAMD_STREAM(s) reinterpret_cast<hipStream_t>(s)
