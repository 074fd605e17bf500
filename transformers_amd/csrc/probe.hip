// probe.hip -- hardware-semantics probe (diagnostic entry point, not on the product path).
//
// Every gfx950 behaviour the kernels rely on is funnelled through tamd_device.h; tests/hipemu
// implements the same wrappers from the documented semantics.  tamd_probe() runs one wrapper on one
// wave with caller-supplied operands so tests/test_gpu_probe.py can compare the hardware against the
// CPU model bit for bit (MFMA fragment layouts, ds_read_b64_tr_b16, v_permlane32_swap, direct-to-LDS
// loads, ballot).  A mismatch means the model -- and therefore every emulator-validated layout -- is wrong.
#include "common.h"

namespace tamd {

template <typename T>
__global__ void probe_kernel(const unsigned int* __restrict__ in, const unsigned int* __restrict__ in2,
                             unsigned int* __restrict__ out, int which) {
  TAMD_DYN_SMEM(smem);
  const int lane = threadIdx.x;
  if (which == 0) {  // mfma 32x32x16
    const u32x4 a = ld16(in + lane * 4), b = ld16(in + 256 + lane * 4);
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = (float)((lane + r) & 7);
    c = mfma32<T>(a, b, c);
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = f32_as_u32(c[r]);
  } else if (which == 1) {  // mfma 16x16x32
    const u32x4 a = ld16(in + lane * 4), b = ld16(in + 256 + lane * 4);
    f32x4 c = {1.f, 2.f, 3.f, 4.f};
    c = mfma16<T>(a, b, c);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = f32_as_u32(c[r]);
  } else if (which == 2) {  // transposing LDS read at caller-chosen per-lane byte offsets
    for (int i = lane; i < 1024; i += 64) lds_write16(smem, (unsigned)i * 16u, ld16(in + i * 4));
    block_sync();
    const u32x2 v = lds_read8_tr16(smem, in2[lane]);
    out[lane * 2] = v[0];
    out[lane * 2 + 1] = v[1];
  } else if (which == 3) {  // permlane32_swap + swap32 + ballot
    unsigned int a = in[lane], b = in[64 + lane];
    permlane32_swap(a, b);
    out[lane] = a;
    out[64 + lane] = b;
    out[128 + lane] = f32_as_u32(swap32_f32(u32_as_f32(in[lane])));
    const unsigned long long m = ballot64((in[lane] & 1u) != 0);
    out[192 + lane] = (unsigned int)(m >> (lane < 32 ? 0 : 32));
    out[256 + lane] = f32_as_u32(wave_sum((float)(in[lane] & 0xffu)));
    out[320 + lane] = f32_as_u32(wave_max((float)(in[lane] & 0xffu)));
  } else if (which == 4) {  // direct-to-LDS load: wave-uniform base + lane*16, per-lane global source
    for (int i = lane; i < 256; i += 64) lds_write16(smem, (unsigned)i * 16u, u32x4{0xdeadbeefu, 0, 0, 0});
    block_sync();
    glds16(reinterpret_cast<const char*>(in) + in2[lane], smem, 1024u);
    wait_vmcnt0();
    block_sync();
    for (int i = lane; i < 256; i += 64) {
      const u32x4 v = lds_read16(smem, (unsigned)i * 16u);
      st16(out + i * 4, v);
    }
  } else if (which == 5) {  // 4-byte direct-to-LDS load: wave-uniform base + lane*4
    for (int i = lane; i < 256; i += 64) lds_write16(smem, (unsigned)i * 16u, u32x4{0xdeadbeefu, 1, 2, 3});
    block_sync();
    glds4(reinterpret_cast<const char*>(in) + (in2[lane] & ~3u), smem, 512u);
    wait_vmcnt0();
    block_sync();
    for (int i = lane; i < 256; i += 64) st16(out + i * 4, lds_read16(smem, (unsigned)i * 16u));
  } else if (which == 6) {  // buffer-addressed direct-to-LDS loads with immediates; reads through the asm wrappers
    for (int i = lane; i < 512; i += 64) lds_write16(smem, (unsigned)i * 16u, u32x4{0xdeadbeefu, 4, 5, 6});
    block_sync();
    const char* base = reinterpret_cast<const char*>(in);
    glds16_buf<0>(base, in2[lane], smem, 1024u);
    glds16_buf<1024>(base, in2[lane], smem, 1024u);
    glds16_buf<3072>(base + 2048, in2[lane], smem, 2048u);
    wait_vmcnt0();
    block_sync();
    const unsigned lds0 = lds_base_u32(smem);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const u32x4 v = lds_read16_abs(lds0 + (unsigned)lane * 16u, j * 1024);
      wait_lgkmcnt0();  // untracked read: the compiler does not wait for it
      st16(out + (j * 64 + lane) * 4, v);
    }
    const u32x2 t = lds_read8_tr16_abs(lds0 + ((in2[lane] >> 1) & 0x7f8u), 1024);
    const u32x2 t2 = lds_read8_tr16_untracked(smem, (in2[lane] >> 1) & 0x7f8u, 2048);
    wait_lgkmcnt0();
    out[2048 + lane * 2] = t[0];
    out[2048 + lane * 2 + 1] = t[1];
    out[2304 + lane * 2] = t2[0];
    out[2304 + lane * 2 + 1] = t2[1];
  } else if (which == 7) {  // range-checked buffer LDS-DMA: lanes past the buffer's size write ZEROS; a scalar offset counts towards the range
    for (int i = lane; i < 512; i += 64) lds_write16(smem, (unsigned)i * 16u, u32x4{0xdeadbeefu, 7, 8, 9});
    block_sync();
    const char* base = reinterpret_cast<const char*>(in);
    glds16_buf_rng(base, 4096u, in2[lane], smem, 1024u);                 // offsets up to 8 KiB: about half out of range
    glds16_buf_rng(base + 1024, 0u, in2[lane], smem, 2048u);              // an empty buffer: every lane zero
    glds4_buf_rng(base, 2048u, in2[lane] >> 2, 4096u, smem, 3072u);       // 4-byte pieces; scalar offset past the size: all zeros
    glds4_buf_rng(base, 6144u, in2[lane] >> 2, 4096u, smem, 3584u);       // ... inside it: offsets below 2048 fetch
    glds4_buf_rng(base, 0x7fffffffu, (unsigned)lane * 4u, 512u, smem, 3328u);
    wait_vmcnt0();
    block_sync();
    for (int i = lane; i < 512; i += 64) st16(out + i * 4, lds_read16(smem, (unsigned)i * 16u));
  }
}

// Bandwidth probe: every wave of every workgroup streams `iters` x 1 KiB pieces of an L2/MALL-resident
// buffer into LDS (MODE 0: LDS-DMA, MODE 1: global_load -> VGPR -> ds_write) with a GEMM-like address pattern:
// `seg` contiguous bytes per row (64 / 128 / 1024), rows `row_stride` bytes apart.
template <int MODE>
__global__ __launch_bounds__(512) void bw_probe_kernel(const char* __restrict__ buf, size_t bytes, int seg,
                                                        size_t row_stride, int iters, unsigned int* sink) {
  TAMD_DYN_SMEM(smem);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lanes_per_row = seg / 16;
  const int rows_per_inst = 64 / lanes_per_row;
  const size_t lane_off = (size_t)(lane / lanes_per_row) * row_stride + (size_t)(lane % lanes_per_row) * 16;
  size_t base = ((size_t)blockIdx.x * 8 + wave) * (size_t)rows_per_inst * row_stride;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    const size_t off = (base + (size_t)it * seg + lane_off) % (bytes - 16);
    const char* src = buf + (off & ~(size_t)15);
    if (MODE == 0) {
      glds16(src, smem, (unsigned)(wave * 8 + (it & 7)) * 1024u);
      if ((it & 7) == 7) wait_vmcnt<4>();
    } else {
      const u32x4 v = ld16(src);
      lds_write16(smem, (unsigned)(wave * 8 + (it & 7)) * 1024u + (unsigned)lane * 16u, v);
    }
  }
  wait_vmcnt<0>();
  block_sync();
  acc = lds_read16(smem, (unsigned)threadIdx.x * 16u)[0];
  if (acc == 0x12345678u) sink[0] = acc;
}


// MFMA power probe: every wave (one per SIMD: 512 registers) runs `iters` rounds of 256 accumulator registers' worth of
// independent MFMAs on operands read once from `in` (random bits -> realistic toggling), rotating through 8 A and 8 B
// operand registers the way a 128x128 wave tile does.  MODE 0: 16 x v_mfma_f32_32x32x16_bf16 per round (4 A x 4 B);
// MODE 1: 64 x v_mfma_f32_16x16x32_bf16 per round (8 A x 8 B): the same flops per round.  Stores {shader ticks,
// real-time ticks} per workgroup: which shape does more work inside the power budget?  tools/mfma_power.py
template <int MODE>
__global__ __launch_bounds__(256, 1) void mfma_power_kernel(const unsigned int* __restrict__ in, int iters,
                                                             unsigned long long* clk, float* sink) {
  const int lane = threadIdx.x & 63;
  u32x4 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = ld16(in + ((size_t)(blockIdx.x * 7 + i) % 61) * 256 + lane * 4);
    b[i] = ld16(in + ((size_t)(blockIdx.x * 5 + i + 8) % 61) * 256 + lane * 4);
  }
  const unsigned long long t0 = device_clock(), r0 = device_realtime();
  float out = 0.f;
  if (MODE == 0) {
    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          acc[i] = mfma32<bf16_t>(a[ks * 4 + (i >> 2)], b[ks * 4 + (i & 3)], acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) out += acc[i][lane & 15];
  } else {
    f32x4 acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 64; ++i) acc[i] = mfma16<bf16_t>(a[i >> 3], b[i & 7], acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 64; ++i) out += acc[i][lane & 3];
  }
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = device_clock() - t0;
    clk[2 * blockIdx.x + 1] = device_realtime() - r0;
  }
  if (out == 12345.678f) sink[0] = out;
}

}  // namespace tamd

using namespace tamd;

extern "C" int tamd_bw_probe(const void* buf, size_t bytes, int seg, size_t row_stride, int iters, int mode,
                             int blocks, void* sink, tamd_stream_t stream) {
  if (!buf || !sink) return TAMD_E_NULL;
  if (seg != 64 && seg != 128 && seg != 256 && seg != 1024) return TAMD_E_ARG;
  if (mode == 0)
    hipLaunchKernelGGL((bw_probe_kernel<0>), dim3((unsigned)blocks), dim3(512), (size_t)65536, TAMD_STREAM(stream),
                       (const char*)buf, bytes, seg, row_stride, iters, (unsigned int*)sink);
  else
    hipLaunchKernelGGL((bw_probe_kernel<1>), dim3((unsigned)blocks), dim3(512), (size_t)65536, TAMD_STREAM(stream),
                       (const char*)buf, bytes, seg, row_stride, iters, (unsigned int*)sink);
  return launch_status();
}


// in: 61 x 256 u32 of operand bits; clk: uint64[2 * blocks]; flops per workgroup = iters * 4 waves * 2 * 32 * 32 * 16 * 32
extern "C" int tamd_mfma_power(const void* in, int iters, int mode, int blocks, void* clk, void* sink, tamd_stream_t stream) {
  if (!in || !clk || !sink) return TAMD_E_NULL;
  if (mode == 0)
    hipLaunchKernelGGL((mfma_power_kernel<0>), dim3((unsigned)blocks), dim3(256), 0, TAMD_STREAM(stream),
                       (const unsigned int*)in, iters, (unsigned long long*)clk, (float*)sink);
  else
    hipLaunchKernelGGL((mfma_power_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, TAMD_STREAM(stream),
                       (const unsigned int*)in, iters, (unsigned long long*)clk, (float*)sink);
  return launch_status();
}

// in: 4096 u32 (16 KiB), in2: 64 u32, out: 4096 u32.  One wave.
extern "C" int tamd_probe(const void* in, const void* in2, void* out, int which, int dtype, tamd_stream_t stream) {
  if (!in || !in2 || !out) return TAMD_E_NULL;
  TAMD_DISPATCH_HALF(dtype, {
    hipLaunchKernelGGL((probe_kernel<T>), dim3(1), dim3(64), (size_t)16384, TAMD_STREAM(stream),
                       (const unsigned int*)in, (const unsigned int*)in2, (unsigned int*)out, which);
  });
  return launch_status();
}
