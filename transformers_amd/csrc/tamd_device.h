// tamd_device.h -- device-side hardware wrappers for the gfx950 (CDNA4 / MI355X) kernels.
//
// Everything the kernels need from the hardware goes through the thin inline
// wrappers in this file (MFMA, LDS transpose reads, direct-to-LDS loads, lane
// exchanges).  The kernels themselves are plain C++ over these wrappers and are
// included as <tamd_device.h>; tests/hipemu/ puts a lane-accurate CPU model of
// the same wrappers first on the include path and runs the *same kernel
// sources* under an emulator in the CPU test-suite (tests/hipemu/README.md).
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
//   * v_permlane32_swap vdst, src: lanes 32-63 of vdst <-> lanes 0-31 of src
// tests/probe_hw.py checks every one of these statements on the GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <tamd_types.h>

namespace tamd {

// ---------------------------------------------------------------- lane exchange
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float shfl_xor_f32(float v, int mask) { return __shfl_xor(v, mask, 64); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f32(v, m);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, shfl_xor_f32(v, m));
  return v;
}
// value of lane (l ^ 32): one v_permlane32_swap instead of a ds_bpermute
__device__ __forceinline__ float swap32_f32(float v) {
  unsigned int u = __builtin_bit_cast(unsigned int, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  // r[0] (vdst): lanes 32-63 now hold the lower half's value; r[1] (src): lanes 0-31 the upper half's
  unsigned int o = (lane_id() < 32) ? r[1] : r[0];
  return __builtin_bit_cast(float, o);
}
// the value of the EVEN (ODD) lane of this lane's pair (l & ~1, l | 1), in both lanes of the pair: one v_mov_b32 with a DPP
// quad_perm of [0,0,2,2] ([1,1,3,3]) -- VALU only, no LDS crossbar, and the DPP combiner folds it into its consumer where it can.
// (attention dropout: the two query rows -- or keys -- of a 2 x 2 hash block sit in neighbouring lanes; one of them hashes.)
__device__ __forceinline__ unsigned int pair_even_u32(unsigned int v) {
  return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0xA0, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned int pair_odd_u32(unsigned int v) {
  return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0xF5, 0xf, 0xf, false);
}
// ... and the value of the OTHER lane of the pair (l ^ 1): quad_perm [1,0,3,2]
__device__ __forceinline__ unsigned int pair_swap_u32(unsigned int v) {
  return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);
}
// Half exchange of two registers: after the call, for lanes 0-31  a = own a, b = upper lanes' a;
// for lanes 32-63 a = lower lanes' b, b = own b.   (v_permlane32_swap vdst=a', src=b')
__device__ __forceinline__ void permlane32_swap(unsigned int& a, unsigned int& b) {
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}

// wave index inside the workgroup as a provably wave-uniform (SGPR) value
__device__ __forceinline__ int wave_id_uniform() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// lane index recomputed from the execution mask (v_mbcnt; all 64 lanes must be active): nothing has to stay in a
// register -- or get spilled -- for it across a long loop (volatile: a plain builtin would be hoisted back above the loop)
__device__ __forceinline__ int lane_id_mbcnt() {
  int x;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
  return x;
}

// 64-bit mask of the lanes whose predicate is true
__device__ __forceinline__ unsigned long long ballot64(bool pred) { return __ballot(pred); }

// The inverse: a WAVE-UNIFORM 64-bit mask (bit l = lane l) as a per-lane choice -- one v_cndmask_b32 with the mask in an SGPR
// pair, no per-lane bit arithmetic.
__device__ __forceinline__ float lane_select(unsigned long long mask, float if_set, float if_clear) {
  return __builtin_amdgcn_inverse_ballot_w64(mask) ? if_set : if_clear;
}
// 64-bit word `idx` of a read-only table at a WAVE-UNIFORM address: a scalar load (s_load_dwordx2 .. x16 when neighbouring
// words are read together) through the constant address space -- the value arrives in SGPRs, ready for lane_select
__device__ __forceinline__ unsigned long long uniform_u64(const unsigned long long* table, long long idx) {
  typedef const __attribute__((address_space(4))) unsigned long long* cptr64;
  return ((cptr64)(table))[idx];
}

// ---------------------------------------------------------------- MFMA
template <typename T>
__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c);
template <>
__device__ __forceinline__ f32x16 mfma32<bf16_t>(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                 0);
}
template <>
__device__ __forceinline__ f32x16 mfma32<f16_t>(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0,
                                                0);
}
template <typename T>
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 mfma16<bf16_t>(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                 0);
}
template <>
__device__ __forceinline__ f32x4 mfma16<f16_t>(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0,
                                                0);
}

// ---------------------------------------------------------------- LDS
// All LDS addressing in the MFMA kernels is by BYTE OFFSET into one dynamic
// array (cdna_hip_programming.md G17: a single 16-byte aligned extern array).
#define TAMD_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]

__device__ __forceinline__ u32x4 lds_read16(const char* smem, unsigned off) {
  return *reinterpret_cast<const u32x4*>(smem + off);
}
__device__ __forceinline__ u32x2 lds_read8(const char* smem, unsigned off) {
  return *reinterpret_cast<const u32x2*>(smem + off);
}
__device__ __forceinline__ void lds_write16(char* smem, unsigned off, u32x4 v) {
  *reinterpret_cast<u32x4*>(smem + off) = v;
}
__device__ __forceinline__ void lds_write8(char* smem, unsigned off, u32x2 v) {
  *reinterpret_cast<u32x2*>(smem + off) = v;
}
__device__ __forceinline__ float lds_read_f32(const char* smem, unsigned off) {
  return *reinterpret_cast<const float*>(smem + off);
}
__device__ __forceinline__ void lds_write_f32(char* smem, unsigned off, float v) {
  *reinterpret_cast<float*>(smem + off) = v;
}
// ds_read_b64_tr_b16: transposing 8-byte read (semantics in the file header).
__device__ __forceinline__ u32x2 lds_read8_tr16(const char* smem, unsigned off) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + off));
  return __builtin_bit_cast(u32x2, v);
}
// The same instruction issued behind the compiler's back.  Why: with LDS-DMA loads in flight, LLVM's waitcnt
// insertion puts `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16 it knows about whenever it cannot
// rule out that the read touches an LDS-DMA destination (seen in gemm_fl_kernel with k-major operands: 43-73 drains inside the loop,
// i.e. no load pipelining at all).  The caller orders reads against landed data itself (counted vmcnt + barrier),
// and must place wait_lgkmcnt0() before the first use of the result: the compiler does not know the register
// is filled asynchronously.
// `imm` (0..65535) must fold to a constant after inlining: it becomes the instruction's offset field.
__device__ __forceinline__ u32x2 lds_read8_tr16_untracked(const char* smem, unsigned off, int imm) {
  typedef __attribute__((address_space(3))) const char lds_char;
  const unsigned addr = (unsigned)reinterpret_cast<__UINTPTR_TYPE__>((lds_char*)smem) + off;
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(imm) : "memory");
  return v;
}
// ds_read_b128 issued the same way (lets a kernel keep ALL its LDS reads out of the compiler's lgkmcnt bookkeeping,
// so that no compiler-placed lgkmcnt(0) drains reads the kernel issued ahead on purpose).
__device__ __forceinline__ u32x4 lds_read16_untracked(const char* smem, unsigned off, int imm) {
  typedef __attribute__((address_space(3))) const char lds_char;
  const unsigned addr = (unsigned)reinterpret_cast<__UINTPTR_TYPE__>((lds_char*)smem) + off;
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(imm) : "memory");
  return v;
}
// LDS byte address of the dynamic shared array (what ds_* instructions take)
__device__ __forceinline__ unsigned lds_base_u32(const char* smem) {
  typedef __attribute__((address_space(3))) const char lds_char;
  return (unsigned)reinterpret_cast<__UINTPTR_TYPE__>((lds_char*)smem);
}
// ... and with the absolute LDS byte address already in a register (no per-read address arithmetic)
__device__ __forceinline__ u32x4 lds_read16_abs(unsigned addr, int imm) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(imm) : "memory");
  return v;
}
__device__ __forceinline__ u32x2 lds_read8_tr16_abs(unsigned addr, int imm) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(imm) : "memory");
  return v;
}
// Direct-to-LDS 16-byte load: the wave writes 64 x 16 B = 1 KiB contiguous at
// `smem + wave_base_off` (must be wave-uniform); each lane supplies its own
// global source address.  Completion is tracked by vmcnt.
// AUX = cache policy bits of the load (0 default, 1 sc0, 2 nt, 16 sc1).
template <int AUX = 0>
__device__ __forceinline__ void glds16(const void* gsrc, char* smem, unsigned wave_base_off) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)(smem + wave_base_off), 16, 0, AUX);
}
// Buffer-addressed LDS-DMA (buffer_load_dwordx4 ... offen lds): wave-uniform `base` (SGPR resource), per-lane 32-bit
// byte offset `voff`, and an instruction immediate IMM (< 4096) that the hardware adds to BOTH addresses:
//     global source = base + voff + IMM          LDS destination = smem + lds_base_off + IMM + lane*16
// so one M0 value serves several pieces (IMM = 0, 1024, 2048, 3072) and no per-lane 64-bit pointer has to be kept
// or advanced: the stage-to-stage step goes into `base`.  (tests/test_gpu_probe.py checks this against silicon.)
template <int IMM>
__device__ __forceinline__ void glds16_buf(const void* base, unsigned voff, char* smem, unsigned lds_base_off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + lds_base_off), 16,
                                           (int)voff, 0, IMM, 0);
}
// The same with the hardware's range check (wave-uniform `bytes` = the buffer's size): a lane whose access does not lie inside
// [0, bytes) fetches nothing and ZEROS land in its LDS slot -- ragged and padding tiles cost no branch and no zero page
// (tests/test_gpu_probe.py probe 7 checks this on silicon -- and that a scalar offset counts towards the range: MI355X
// returned zeros for every lane of a piece whose `soff` alone was past `bytes`).
__device__ __forceinline__ void glds16_buf_rng(const void* base, unsigned bytes, unsigned voff, char* smem, unsigned lds_base_off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + lds_base_off), 16,
                                           (int)voff, 0, 0, 0);
}
// 4-byte pieces (buffer_load_dword ... offen lds): the wave writes 64 x 4 B at smem + lds_base_off; `soff` (wave-uniform) is
// added to the source address AND range-checked with the lane offset: soff + voff + 4 <= bytes
__device__ __forceinline__ void glds4_buf_rng(const void* base, unsigned bytes, unsigned voff, unsigned soff, char* smem,
                                              unsigned lds_base_off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + lds_base_off), 4,
                                           (int)voff, (int)soff, 0, 0);
}
// 16-byte buffer load into registers (buffer_load_dwordx4 ... offen): wave-uniform `base`, per-lane 32-bit byte offset
__device__ __forceinline__ u32x4 buf_load16(const void* base, unsigned voff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, 0x7fffffff, 0x00020000);
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
}
// Range-checked 16-byte buffer accesses: wave-uniform `base` and size in bytes; a lane whose 16 bytes are not inside
// [0, bytes) reads zeros / stores nothing (the hardware's buffer range check; offsets and sizes here are multiples of 16,
// so an access is either wholly inside or wholly outside).  Ragged rows and tails cost no branch -- and hipcc answers a
// branch around a store with s_waitcnt vmcnt(0) at the join: gfx9-family loads and stores share one out-of-order counter.
__device__ __forceinline__ u32x4 buf_load16_rng(const void* base, unsigned bytes, unsigned voff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
}
__device__ __forceinline__ void buf_store16_rng(void* base, unsigned bytes, unsigned voff, u32x4 v) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, (short)0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), r, (int)voff, 0, 0);
}
// ... as a streaming store (nt): output nobody reads again soon (activations saved for the backward)
__device__ __forceinline__ void buf_store16_rng_nt(void* base, unsigned bytes, unsigned voff, u32x4 v) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, (short)0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), r, (int)voff, 0, 2);
}
// 4-byte variant (global_load_lds_dword): the wave writes 64 x 4 B = 256 B contiguous at smem + wave_base_off
__device__ __forceinline__ void glds4(const void* gsrc, char* smem, unsigned wave_base_off) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)(smem + wave_base_off), 4, 0, 0);
}
__device__ __forceinline__ void wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// counted wait: returns when at most N of this wave's vector-memory operations (incl. LDS-DMA) are pending
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkmcnt0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Workgroup barrier WITHOUT the vmcnt(0) drain that __syncthreads() implies while LDS-DMA is in flight
// (cdna_hip_programming.md §5 "Pipelining across barriers"): the caller places its own counted waits.
__device__ __forceinline__ void raw_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// pin the instruction scheduler at a phase boundary (MFMAs are register-only and would otherwise drift)
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void block_sync() { __syncthreads(); }
// A wave executes in lockstep on hardware, so LDS writes of one lane are visible to the other lanes of
// the SAME wave at the next DS instruction (DS ops of a wave retire in order).  This marks such a point
// (compiler scheduling fence only); the CPU model turns it into a real wave rendez-vous.
__device__ __forceinline__ void wave_lockstep_point() { __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ void setprio_hi() { __builtin_amdgcn_s_setprio(1); }
__device__ __forceinline__ void setprio_lo() { __builtin_amdgcn_s_setprio(0); }

// shader clock (s_memtime), for the in-kernel phase traces of the diagnostic GEMM variant
__device__ __forceinline__ unsigned long long device_clock() { return __builtin_amdgcn_s_memtime(); }
// constant-rate (100 MHz) counter: shader-clock ticks / real-time ticks = the clock a kernel actually ran at
__device__ __forceinline__ unsigned long long device_realtime() { return __builtin_amdgcn_s_memrealtime(); }
// the XCD this wave runs on (HW_REG_XCC_ID = 20, bits 3:0), for the diagnostic timelines
__device__ __forceinline__ unsigned device_xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xfu; }


// fast transcendental pieces
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// logistic function of the SiLU / quick-GELU paths (activations.py:92-123): 1 / (1 + exp(-x)) as v_exp_f32 + v_rcp_f32 (1 ulp
// each; every user rounds the product to a 16-bit storage type right after).  ONE definition for the GEMM epilogues, the
// element-wise kernels and the weight-streaming kernels, so that fused and unfused paths agree bit for bit.  (An IEEE
// division here is ten instructions per element -- v_div_scale x2, v_rcp, four fma, v_div_fmas, v_div_fixup -- a third of the
// SwiGLU epilogue of the gate|up GEMM, during which the matrix pipe idles.)
__device__ __forceinline__ float fast_sigmoid(float x) { return fast_rcp(1.f + fast_exp2(x * -1.44269504088896340736f)); }
// erf-GELU (GELUActivation, activations.py:69-89): x * 0.5 * (1 + erf(x / sqrt 2)) and its derivative, from ONE exponential:
//     erfc(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2),  t = 1 / (1 + p z),  z = |x| / sqrt 2     (Abramowitz & Stegun
// 7.1.26, |error| <= 1.5e-7), 1 + erf = erfc(z) for x < 0 (no cancellation in the tail, where fp32 `1 + erff` loses every digit) and
// 2 - erfc(z) otherwise; the same exp(-x^2 / 2) is the Gaussian of the derivative cdf + x pdf.  ~16 instructions where ocml's erff
// is ~100 dependent, divergent ones (the activation kernels of bert-base ran at 3.3 TB/s on it: arithmetic-bound); against the
// fp32 formula with an exact erf 0.2 % of the bf16 outputs move by one ulp (8e-8 norm-relative).  ONE definition for the
// element-wise kernels and the GEMM epilogue, so that fused and unfused paths agree bit for bit.
__device__ __forceinline__ void gelu_erf_parts(float x, float& one_plus_erf, float& gauss) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = fast_rcp(fmaf(0.3275911f, z, 1.f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  gauss = fast_exp2((z * z) * -1.44269504088896340736f);  // exp(-x^2 / 2)
  const float h = poly * gauss;                           // erfc(|x| / sqrt 2)
  one_plus_erf = x < 0.f ? h : 2.f - h;
}
__device__ __forceinline__ float gelu_erf_f(float x) {
  float ope, g;
  gelu_erf_parts(x, ope, g);
  return x * 0.5f * ope;
}
__device__ __forceinline__ float dgelu_erf_f(float x) {
  float ope, g;
  gelu_erf_parts(x, ope, g);
  return 0.5f * ope + x * (0.39894228040143267794f * g);
}
__device__ __forceinline__ float fast_log2(float x) { return __log2f(x); }

}  // namespace tamd
