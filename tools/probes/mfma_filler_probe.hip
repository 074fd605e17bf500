// How many VALU instructions hide beside a v_mfma_f32_32x32x16_bf16 when ONE wave owns the SIMD?  A loop of 16 MFMAs
// (independent accumulators, round robin over 4) with NF filler instructions behind each, fillers of several kinds,
// accumulators in VGPRs or AGPRs, A/B operands in VGPRs or AGPRs.  Prints shader cycles (s_memtime) per MFMA.
// Build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_filler_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// KIND 0: v_fma_f32 (independent chains), 1: v_exp_f32, 2: mix (2 fma, 2 exp, 2 add, 1 cvt_pk per 7), 3: s_nop 0
template <int NF, int KIND, bool ACC_AGPR, bool AB_AGPR>
__global__ __launch_bounds__(256, 1) void probe(const u32x4* in, float* out, unsigned long long* cyc, int iters) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63;
  u32x4 a = in[lane], b = in[64 + lane];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = 1.0f + 0.001f * (lane + i);
  if (AB_AGPR) {
    asm volatile("" : "+a"(a));
    asm volatile("" : "+a"(b));
  }
  if (ACC_AGPR)
    for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(acc[i]));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (ACC_AGPR) {
        if (AB_AGPR)
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m & 3]) : "a"(a), "a"(b));
        else
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m & 3]) : "v"(a), "v"(b));
      } else {
        if (AB_AGPR)
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "a"(a), "a"(b));
        else
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
      }
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        float& x = f[(m * NF + k) & 7];
        if (KIND == 0) {
          asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(f[7 - ((m * NF + k) & 7)]));
        } else if (KIND == 1) {
          asm volatile("v_exp_f32 %0, %0" : "+v"(x));
        } else if (KIND == 2) {
          const int w = k % 7;
          if (w < 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(f[(k + 3) & 7]));
          else if (w < 4) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
          else if (w < 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(f[(k + 5) & 7]));
          else asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(f[(k + 1) & 7]));
        } else {
          asm volatile("s_nop 0");
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (ACC_AGPR)
    for (int i = 0; i < 4; ++i) asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[i]));
  else
    for (int i = 0; i < 4; ++i) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[i]));
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NF, int KIND, bool ACC_AGPR, bool AB_AGPR>
void run(const char* name, const u32x4* in, float* out, unsigned long long* cyc, int nblk) {
  const int iters = 2000;
  hipFuncSetAttribute((const void*)probe<NF, KIND, ACC_AGPR, AB_AGPR>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<NF, KIND, ACC_AGPR, AB_AGPR>), dim3(nblk), dim3(256), 100 * 1024, 0, in, out, cyc, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<NF, KIND, ACC_AGPR, AB_AGPR>), dim3(nblk), dim3(256), 100 * 1024, 0, in, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nblk);
  hipMemcpy(h.data(), cyc, nblk * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double c = 0;
  for (auto v : h) c += (double)v;
  c /= nblk;
  // s_memtime counts at a constant 100 MHz on gfx9; report wall ns per MFMA and TFLOP/s as the primary numbers
  const double mfmas = (double)iters * 16;
  const double ns = ms * 1e6 / mfmas;
  const double tf = (double)nblk * 4 * mfmas * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
  printf("{\"probe\": \"%s\", \"fillers\": %d, \"acc\": \"%s\", \"ab\": \"%s\", \"ns_per_mfma\": %.2f, \"tflops\": %.0f, \"memtime_ticks_per_mfma\": %.3f}\n",
         name, NF, ACC_AGPR ? "agpr" : "vgpr", AB_AGPR ? "agpr" : "vgpr", ns, tf, c / mfmas);
  fflush(stdout);
}

int main() {
  const int nblk = 256;
  u32x4* in;
  float* out;
  unsigned long long* cyc;
  hipMalloc(&in, 128 * sizeof(u32x4));
  hipMalloc(&out, nblk * 256 * sizeof(float));
  hipMalloc(&cyc, nblk * sizeof(unsigned long long));
  std::vector<unsigned> h(512);
  for (int i = 0; i < 512; ++i) h[i] = 0x3f803f80u + (unsigned)(i * 2654435761u >> 20);  // bf16 pairs near 1.0, varied mantissas
  hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice);
#define RUN_ALL(KIND_, NAME_, ACC_, AB_)                                                           \
  run<0, KIND_, ACC_, AB_>(NAME_, in, out, cyc, nblk); run<2, KIND_, ACC_, AB_>(NAME_, in, out, cyc, nblk);   \
  run<4, KIND_, ACC_, AB_>(NAME_, in, out, cyc, nblk); run<5, KIND_, ACC_, AB_>(NAME_, in, out, cyc, nblk);   \
  run<6, KIND_, ACC_, AB_>(NAME_, in, out, cyc, nblk); run<7, KIND_, ACC_, AB_>(NAME_, in, out, cyc, nblk);   \
  run<8, KIND_, ACC_, AB_>(NAME_, in, out, cyc, nblk); run<10, KIND_, ACC_, AB_>(NAME_, in, out, cyc, nblk);
  RUN_ALL(0, "fma", false, true)
  RUN_ALL(0, "fma", true, false)
  RUN_ALL(2, "softmax_mix", false, true)
  RUN_ALL(2, "softmax_mix", true, false)
  RUN_ALL(1, "exp", false, true)
  RUN_ALL(3, "s_nop", false, true)
  return 0;
}
