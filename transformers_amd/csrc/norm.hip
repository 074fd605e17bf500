// norm.hip -- RMSNorm / LayerNorm forward + backward for gfx950 (HBM-bound row kernels).
//
// Reference semantics:
//   LlamaRMSNorm.forward      src/transformers/models/llama/modeling_llama.py:62-67
//   nn.LayerNorm call sites   models/bert/modeling_bert.py:62,106,286,292,344,350;
//                             models/gpt2/modeling_gpt2.py:252-254; models/clip/modeling_clip.py:358-360
//   residual adds fused in    modeling_llama.py:317,323; modeling_bert.py:292,350
//
// Design (MI355X): one row is owned by WPR waves (1, 2 or 4) of a 256-thread block; every lane
// keeps its 16-byte column chunks of the row in registers, so each tensor is read from HBM
// exactly once and written once (algorithmic bytes = 2 * rows * cols * sizeof(T)).  Row
// statistics are fp32: wave-level DPP/shuffle reduction, then a 4-word LDS exchange when a
// row spans several waves.  The backward accumulates dw/db per lane in registers across a
// grid-stride loop over rows, combines the row groups of a block through LDS and writes one
// fp32 partial row per block; a second tiny kernel reduces the partials.
#include "common.h"

namespace tamd {

constexpr int kNormThreads = 256;
constexpr int kNormWaves = kNormThreads / 64;
constexpr int kNormMaxPartials = 512;

template <int WPR>
__device__ __forceinline__ float group_sum(float v, float* red, int wave) {
  v = wave_sum(v);
  if (WPR == 1) return v;
  __syncthreads();
  if (lane_id() == 0) red[wave] = v;
  __syncthreads();
  const int g0 = (wave / WPR) * WPR;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < WPR; ++i) s += red[g0 + i];
  return s;
}

// ---------------------------------------------------------------- forward
template <typename T, int NCH, int WPR, bool LN, bool HAS_RES>
__global__ __launch_bounds__(kNormThreads) void norm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                                const T* __restrict__ w, const T* __restrict__ b,
                                                                T* __restrict__ y, T* __restrict__ hout,
                                                                float* __restrict__ mean_out,
                                                                float* __restrict__ rstd_out, int64_t rows, int cols,
                                                                float eps) {
  constexpr int VE = vec16<T>::N;
  constexpr int RPB = kNormWaves / WPR;  // rows per block
  __shared__ float red[kNormWaves];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * RPB + wave / WPR;
  const int wsub = wave % WPR;
  const bool active = row < rows;
  const T* xr = x + row * cols;
  float v[NCH][VE];
  float s1 = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = ((c * WPR + wsub) * 64 + lane) * VE;
    if (active && col < cols) {
      unpack16<T>(ld16(xr + col), v[c]);
      if (HAS_RES) {
        float r[VE];
        unpack16<T>(ld16(res + row * cols + col), r);
#pragma unroll
        for (int i = 0; i < VE; ++i) v[c][i] = round_through<T>(v[c][i] + r[i]);
        st16(hout + row * cols + col, pack16<T>(v[c]));
      }
#pragma unroll
      for (int i = 0; i < VE; ++i) s1 += LN ? v[c][i] : v[c][i] * v[c][i];
    } else {
#pragma unroll
      for (int i = 0; i < VE; ++i) v[c][i] = 0.f;
    }
  }
  s1 = group_sum<WPR>(s1, red, wave);
  float mu = 0.f, rs;
  if (LN) {
    mu = s1 / (float)cols;
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = ((c * WPR + wsub) * 64 + lane) * VE;
      if (col < cols) {
#pragma unroll
        for (int i = 0; i < VE; ++i) {
          const float d = v[c][i] - mu;
          s2 += d * d;
        }
      }
    }
    s2 = group_sum<WPR>(s2, red, wave);
    rs = rsqrtf(s2 / (float)cols + eps);
  } else {
    rs = rsqrtf(s1 / (float)cols + eps);
  }
  if (active && lane == 0 && wsub == 0) {
    rstd_out[row] = rs;
    if (LN) mean_out[row] = mu;
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = ((c * WPR + wsub) * 64 + lane) * VE;
    if (active && col < cols) {
      float wv[VE], o[VE];
      unpack16<T>(ld16(w + col), wv);
      if (LN) {
        float bv[VE];
        if (b != nullptr) {
          unpack16<T>(ld16(b + col), bv);
        } else {
#pragma unroll
          for (int i = 0; i < VE; ++i) bv[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < VE; ++i) o[i] = (v[c][i] - mu) * rs * wv[i] + bv[i];
      } else {
        // reference: self.weight * hidden_states.to(input_dtype)  (round, then multiply)
#pragma unroll
        for (int i = 0; i < VE; ++i) o[i] = wv[i] * round_through<T>(v[c][i] * rs);
      }
      st16(y + row * cols + col, pack16<T>(o));
    }
  }
}

// ---------------------------------------------------------------- backward
template <typename T, int NCH, int WPR, bool LN, bool HAS_DRES>
__global__ __launch_bounds__(kNormThreads) void norm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ h,
                                                                const T* __restrict__ w,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd,
                                                                const T* __restrict__ dres, T* __restrict__ dx,
                                                                float* __restrict__ dw_part,
                                                                float* __restrict__ db_part, int64_t rows, int cols) {
  constexpr int VE = vec16<T>::N;
  constexpr int RPB = kNormWaves / WPR;
  __shared__ float red[kNormWaves];
  TAMD_DYN_SMEM(smem);  // cols fp32 (x2 for LN): cross-group dw/db combine
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int grp = wave / WPR, wsub = wave % WPR;
  float dwacc[NCH][VE], dbacc[LN ? NCH : 1][VE];
  u32x4 wp[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = ((c * WPR + wsub) * 64 + lane) * VE;
    wp[c] = (col < cols) ? ld16(w + col) : u32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < VE; ++i) {
      dwacc[c][i] = 0.f;
      if (LN) dbacc[c][i] = 0.f;
    }
  }
  const int64_t stride = (int64_t)gridDim.x * RPB;
  const int64_t iters = (rows + stride - 1) / stride;
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t row = it * stride + (int64_t)blockIdx.x * RPB + grp;
    const bool active = row < rows;
    u32x4 hp[NCH], dyp[NCH];
    float mu = 0.f, rs = 0.f;
    if (active) {
      rs = rstd[row];
      if (LN) mu = mean[row];
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = ((c * WPR + wsub) * 64 + lane) * VE;
      const bool ok = active && col < cols;
      hp[c] = ok ? ld16(h + row * cols + col) : u32x4{0, 0, 0, 0};
      dyp[c] = ok ? ld16(dy + row * cols + col) : u32x4{0, 0, 0, 0};
    }
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = ((c * WPR + wsub) * 64 + lane) * VE;
      if (active && col < cols) {
        float hv[VE], dv[VE], wv[VE];
        unpack16<T>(hp[c], hv);
        unpack16<T>(dyp[c], dv);
        unpack16<T>(wp[c], wv);
#pragma unroll
        for (int i = 0; i < VE; ++i) {
          const float xh = LN ? (hv[i] - mu) * rs : hv[i] * rs;
          const float g = dv[i] * wv[i];
          sgx += g * xh;
          if (LN) {
            sg += g;
            dwacc[c][i] += dv[i] * xh;
            dbacc[c][i] += dv[i];
          } else {
            dwacc[c][i] += dv[i] * round_through<T>(xh);
          }
        }
      }
    }
    sgx = group_sum<WPR>(sgx, red, wave) / (float)cols;
    if (LN) sg = group_sum<WPR>(sg, red, wave) / (float)cols;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = ((c * WPR + wsub) * 64 + lane) * VE;
      if (active && col < cols) {
        float hv[VE], dv[VE], wv[VE], o[VE];
        unpack16<T>(hp[c], hv);
        unpack16<T>(dyp[c], dv);
        unpack16<T>(wp[c], wv);
        float rv[VE];
        if (HAS_DRES) unpack16<T>(ld16(dres + row * cols + col), rv);
#pragma unroll
        for (int i = 0; i < VE; ++i) {
          const float xh = LN ? (hv[i] - mu) * rs : hv[i] * rs;
          const float g = dv[i] * wv[i];
          float d = rs * (g - sg - xh * sgx);
          if (HAS_DRES) d += rv[i];
          o[i] = d;
        }
        st16(dx + row * cols + col, pack16<T>(o));
      }
    }
  }
  // combine the RPB row groups of this block through LDS, last group writes the partial row
  float* sdw = reinterpret_cast<float*>(smem);
  float* sdb = sdw + cols;
#pragma unroll 1
  for (int g = 0; g < RPB; ++g) {
    if (grp == g) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = ((c * WPR + wsub) * 64 + lane) * VE;
        if (col < cols) {
#pragma unroll
          for (int i = 0; i < VE; ++i) {
            float a = dwacc[c][i] + (g > 0 ? sdw[col + i] : 0.f);
            float bb = 0.f;
            if (LN) bb = dbacc[c][i] + (g > 0 ? sdb[col + i] : 0.f);
            if (g == RPB - 1) {
              dw_part[(int64_t)blockIdx.x * cols + col + i] = a;
              if (LN) db_part[(int64_t)blockIdx.x * cols + col + i] = bb;
            } else {
              sdw[col + i] = a;
              if (LN) sdb[col + i] = bb;
            }
          }
        }
      }
    }
    if (RPB > 1) __syncthreads();
  }
}

// partial [P, cols] fp32  ->  out [cols] T.  One workgroup per 16 columns: 64 row groups x 4 column quads, each thread
// sums P/64 rows with 16-byte loads, the row groups are combined through LDS in a fixed order (deterministic).
// (The first version walked all P rows on one thread per column: 16 workgroups and a 512-deep serial chain, 120-140 us
// per call -- a third of the bert-base step and 9 ms of the Llama-3-8B one, profiles/r02_bert_kernel_stats_before.csv.)
constexpr int kColsumCols = 16, kColsumGroups = 64;
template <typename T>
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ part, T* __restrict__ out, int P,
                                                         int cols) {
  __shared__ float sm[kColsumGroups][kColsumCols + 1];
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

// x [rows, cols] (ld) T -> partial [P, cols] fp32 ; each thread owns one 16-byte column vector
template <typename T>
__global__ void colsum_stage1_kernel(const T* __restrict__ x, float* __restrict__ part, int64_t rows, int cols,
                                     int64_t ld, int rows_per_slab) {
  constexpr int VE = vec16<T>::N;
  const int col = (blockIdx.x * blockDim.x + threadIdx.x) * VE;
  if (col >= cols) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
  const int64_t r1 = (r0 + rows_per_slab < rows) ? r0 + rows_per_slab : rows;
  float acc[VE];
#pragma unroll
  for (int i = 0; i < VE; ++i) acc[i] = 0.f;
  for (int64_t r = r0; r < r1; ++r) {
    float v[VE];
    unpack16<T>(ld16(x + r * ld + col), v);
#pragma unroll
    for (int i = 0; i < VE; ++i) acc[i] += v[i];
  }
#pragma unroll
  for (int i = 0; i < VE; ++i) part[(int64_t)blockIdx.y * cols + col + i] = acc[i];
}

// ---------------------------------------------------------------- host dispatch
struct NormGeom {
  int nch, wpr;
};
template <typename T>
static bool norm_geometry(int64_t cols, NormGeom* g) {
  constexpr int VE = vec16<T>::N;
  if (cols <= 0 || cols % VE != 0) return false;
  const int64_t chunks = ceil_div(cols, 64 * VE);
  int wpr = chunks >= 16 ? 4 : (chunks >= 5 ? 2 : 1);
  int64_t per = ceil_div(chunks, wpr);
  int nch = per <= 1 ? 1 : (per <= 2 ? 2 : (per <= 4 ? 4 : 8));
  if (per > 8) return false;
  g->nch = nch;
  g->wpr = wpr;
  return true;
}

template <typename T, int NCH, int WPR, bool LN>
static void launch_fwd(const void* x, const void* res, const void* w, const void* b, void* y, void* hout,
                       float* mean, float* rstd, int64_t rows, int cols, float eps, hipStream_t s) {
  constexpr int RPB = kNormWaves / WPR;
  dim3 grid((unsigned)ceil_div(rows, RPB)), block(kNormThreads);
  if (res != nullptr)
    hipLaunchKernelGGL((norm_fwd_kernel<T, NCH, WPR, LN, true>), grid, block, 0, s, (const T*)x, (const T*)res,
                       (const T*)w, (const T*)b, (T*)y, (T*)hout, mean, rstd, rows, cols, eps);
  else
    hipLaunchKernelGGL((norm_fwd_kernel<T, NCH, WPR, LN, false>), grid, block, 0, s, (const T*)x, (const T*)res,
                       (const T*)w, (const T*)b, (T*)y, (T*)hout, mean, rstd, rows, cols, eps);
}

template <typename T, bool LN>
static int norm_fwd_dispatch(const void* x, const void* res, const void* w, const void* b, void* y, void* hout,
                             float* mean, float* rstd, int64_t rows, int64_t cols, float eps, hipStream_t s) {
  NormGeom g;
  if (!norm_geometry<T>(cols, &g)) return TAMD_E_SHAPE;
#define TAMD_NF(N_, W_)                                                                   \
  if (g.nch == N_ && g.wpr == W_) {                                                       \
    launch_fwd<T, N_, W_, LN>(x, res, w, b, y, hout, mean, rstd, rows, (int)cols, eps, s); \
    return launch_status();                                                               \
  }
  TAMD_NF(1, 1) TAMD_NF(2, 1) TAMD_NF(4, 1) TAMD_NF(4, 2) TAMD_NF(8, 2) TAMD_NF(4, 4) TAMD_NF(8, 4)
#undef TAMD_NF
  return TAMD_E_SHAPE;
}

static int bwd_partials(int64_t rows, int rpb) {
  int64_t p = ceil_div(rows, rpb);
  return (int)(p < kNormMaxPartials ? p : kNormMaxPartials);
}

template <typename T, int NCH, int WPR, bool LN>
static void launch_bwd(const void* dy, const void* h, const void* w, const float* mean, const float* rstd,
                       const void* dres, void* dx, float* dwp, float* dbp, int P, int64_t rows, int cols,
                       hipStream_t s) {
  dim3 grid((unsigned)P), block(kNormThreads);
  const size_t smem = (size_t)cols * sizeof(float) * (LN ? 2 : 1);
  if (dres != nullptr)
    hipLaunchKernelGGL((norm_bwd_kernel<T, NCH, WPR, LN, true>), grid, block, smem, s, (const T*)dy, (const T*)h,
                       (const T*)w, mean, rstd, (const T*)dres, (T*)dx, dwp, dbp, rows, cols);
  else
    hipLaunchKernelGGL((norm_bwd_kernel<T, NCH, WPR, LN, false>), grid, block, smem, s, (const T*)dy, (const T*)h,
                       (const T*)w, mean, rstd, (const T*)dres, (T*)dx, dwp, dbp, rows, cols);
}

template <typename T, bool LN>
static int norm_bwd_dispatch(const void* dy, const void* h, const void* w, const float* mean, const float* rstd,
                             const void* dres, void* dx, void* dw, void* db, void* ws, size_t ws_bytes,
                             int64_t rows, int64_t cols, hipStream_t s) {
  NormGeom g;
  if (!norm_geometry<T>(cols, &g)) return TAMD_E_SHAPE;
  const int rpb = kNormWaves / g.wpr;
  const int P = bwd_partials(rows, rpb);
  if (ws_bytes < (size_t)2 * kNormMaxPartials * cols * sizeof(float)) return TAMD_E_WORKSPACE;
  float* dwp = reinterpret_cast<float*>(ws);
  float* dbp = dwp + (size_t)kNormMaxPartials * cols;
  bool launched = false;
#define TAMD_NB(N_, W_)                                                                          \
  if (!launched && g.nch == N_ && g.wpr == W_) {                                                 \
    launch_bwd<T, N_, W_, LN>(dy, h, w, mean, rstd, dres, dx, dwp, dbp, P, rows, (int)cols, s);   \
    launched = true;                                                                             \
  }
  TAMD_NB(1, 1) TAMD_NB(2, 1) TAMD_NB(4, 1) TAMD_NB(4, 2) TAMD_NB(8, 2) TAMD_NB(4, 4) TAMD_NB(8, 4)
#undef TAMD_NB
  if (!launched) return TAMD_E_SHAPE;
  int st = launch_status();
  if (st != TAMD_OK) return st;
  dim3 g2((unsigned)ceil_div(cols, kColsumCols)), b2(256);
  hipLaunchKernelGGL((colsum_f32_kernel<T>), g2, b2, 0, s, dwp, (T*)dw, P, (int)cols);
  if (LN && db != nullptr) hipLaunchKernelGGL((colsum_f32_kernel<T>), g2, b2, 0, s, dbp, (T*)db, P, (int)cols);
  return launch_status();
}

}  // namespace tamd

using namespace tamd;

extern "C" {

size_t tamd_norm_bwd_workspace_bytes(int64_t rows, int64_t cols) {
  (void)rows;
  return (size_t)2 * kNormMaxPartials * (size_t)cols * sizeof(float);
}

int tamd_rmsnorm_fwd(const void* x, const void* residual, const void* w, void* y, void* h_out, float* rstd,
                     int64_t rows, int64_t cols, float eps, int dtype, tamd_stream_t stream) {
  if (!x || !w || !y || !rstd || (residual && !h_out)) return TAMD_E_NULL;
  if (rows <= 0) return rows == 0 ? TAMD_OK : TAMD_E_SHAPE;
  if (!aligned16(x) || !aligned16(w) || !aligned16(y) || (residual && (!aligned16(residual) || !aligned16(h_out))))
    return TAMD_E_ALIGN;
  TAMD_DISPATCH_DTYPE(dtype, return (norm_fwd_dispatch<T, false>(x, residual, w, nullptr, y, h_out, nullptr, rstd,
                                                                  rows, cols, eps, TAMD_STREAM(stream))));
  return TAMD_E_DTYPE;
}

int tamd_rmsnorm_bwd(const void* dy, const void* h, const void* w, const float* rstd, const void* dres, void* dx,
                     void* dw, void* workspace, size_t workspace_bytes, int64_t rows, int64_t cols, int dtype,
                     tamd_stream_t stream) {
  if (!dy || !h || !w || !rstd || !dx || !dw || !workspace) return TAMD_E_NULL;
  if (rows <= 0) return TAMD_E_SHAPE;
  if (!aligned16(dy) || !aligned16(h) || !aligned16(w) || !aligned16(dx) || (dres && !aligned16(dres)))
    return TAMD_E_ALIGN;
  TAMD_DISPATCH_DTYPE(dtype, return (norm_bwd_dispatch<T, false>(dy, h, w, nullptr, rstd, dres, dx, dw, nullptr,
                                                                  workspace, workspace_bytes, rows, cols,
                                                                  TAMD_STREAM(stream))));
  return TAMD_E_DTYPE;
}

int tamd_layernorm_fwd(const void* x, const void* residual, const void* w, const void* b, void* y, void* h_out,
                       float* mean, float* rstd, int64_t rows, int64_t cols, float eps, int dtype,
                       tamd_stream_t stream) {
  if (!x || !w || !y || !mean || !rstd || (residual && !h_out)) return TAMD_E_NULL;
  if (rows <= 0) return rows == 0 ? TAMD_OK : TAMD_E_SHAPE;
  if (!aligned16(x) || !aligned16(w) || !aligned16(y) || (b && !aligned16(b)) ||
      (residual && (!aligned16(residual) || !aligned16(h_out))))
    return TAMD_E_ALIGN;
  TAMD_DISPATCH_DTYPE(dtype, return (norm_fwd_dispatch<T, true>(x, residual, w, b, y, h_out, mean, rstd, rows, cols,
                                                                 eps, TAMD_STREAM(stream))));
  return TAMD_E_DTYPE;
}

int tamd_layernorm_bwd(const void* dy, const void* h, const void* w, const float* mean, const float* rstd,
                       const void* dres, void* dx, void* dw, void* db, void* workspace, size_t workspace_bytes,
                       int64_t rows, int64_t cols, int dtype, tamd_stream_t stream) {
  if (!dy || !h || !w || !mean || !rstd || !dx || !dw || !workspace) return TAMD_E_NULL;
  if (rows <= 0) return TAMD_E_SHAPE;
  if (!aligned16(dy) || !aligned16(h) || !aligned16(w) || !aligned16(dx) || (dres && !aligned16(dres)))
    return TAMD_E_ALIGN;
  TAMD_DISPATCH_DTYPE(dtype, return (norm_bwd_dispatch<T, true>(dy, h, w, mean, rstd, dres, dx, dw, db, workspace,
                                                                 workspace_bytes, rows, cols, TAMD_STREAM(stream))));
  return TAMD_E_DTYPE;
}

size_t tamd_colsum_workspace_bytes(int64_t rows, int64_t cols) {
  (void)rows;
  return (size_t)kNormMaxPartials * (size_t)cols * sizeof(float);
}

int tamd_colsum(const void* x, void* out, void* workspace, size_t workspace_bytes, int64_t rows, int64_t cols,
                int64_t ld, int dtype, tamd_stream_t stream) {
  if (!x || !out || !workspace) return TAMD_E_NULL;
  if (rows <= 0 || cols <= 0) return TAMD_E_SHAPE;
  if (!aligned16(x)) return TAMD_E_ALIGN;
  if (workspace_bytes < tamd_colsum_workspace_bytes(rows, cols)) return TAMD_E_WORKSPACE;
  hipStream_t s = TAMD_STREAM(stream);
  int rows_per_slab = (int)ceil_div(rows, kNormMaxPartials);
  if (rows_per_slab < 16) rows_per_slab = 16;
  const int P = (int)ceil_div(rows, rows_per_slab);
  float* part = reinterpret_cast<float*>(workspace);
  TAMD_DISPATCH_DTYPE(dtype, {
    constexpr int VE = vec16<T>::N;
    if (cols % VE != 0 || ld % VE != 0) return TAMD_E_SHAPE;
    dim3 g1((unsigned)ceil_div(cols / VE, 256), (unsigned)P), b1(256);
    hipLaunchKernelGGL((colsum_stage1_kernel<T>), g1, b1, 0, s, (const T*)x, part, rows, (int)cols, ld,
                       rows_per_slab);
    dim3 g2((unsigned)ceil_div(cols, kColsumCols)), b2(256);
    hipLaunchKernelGGL((colsum_f32_kernel<T>), g2, b2, 0, s, part, (T*)out, P, (int)cols);
  });
  return launch_status();
}

}  // extern "C"
