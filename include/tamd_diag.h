/* tamd_diag.h -- diagnostic entry points of libtamd_diag.so (the kernel sources built with -DTAMD_DIAG plus
 * csrc/probe.hip).  NOT part of the product ABI: libtamd.so does not export these, the package never loads this
 * library; tools/ (ablation timing, phase traces, bandwidth probes) and tests/test_gpu_probe.py (instruction semantics
 * of the CPU execution model against the silicon) do. */
#ifndef TAMD_DIAG_H_
#define TAMD_DIAG_H_
#include "tamd.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic twin of tamd_gemm (bf16, row-major A[M,K], B[N,K], no epilogue): workgroup 0 additionally writes
 * 8 shader-clock stamps per K sub-tile and wave into trace[8 waves][32 sub-tiles][8] (uint64).  Not a product
 * path; tools/gemm_phase_trace.py turns the stamps into a per-phase cycle breakdown. */
int tamd_gemm_trace(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, void* trace,
                    tamd_stream_t stream);

/* Clock probe of the 4-wave GEMM kernels: while `buf` (uint64[2 * workgroups], device memory) is set, every workgroup
 * of a tamd_gemm launch stores {shader-clock ticks, 100 MHz real-time ticks} of its K loop at buf[2 * workgroup].
 * NULL switches it off.  tools/gemm_clock.py */
int tamd_gemm_set_clock_buffer(void* buf);
/* 3 x uint64 per workgroup of the next gemm_fl_kernel launches: {s_memrealtime (100 MHz) at kernel entry, at exit, XCC id} --
 * the grid's timeline: ramp, tail, whether the XCDs finish together (tools/gemm_timeline.py); NULL switches it off */
int tamd_gemm_set_timeline_buffer(void* buf);

/* Ablation / A-B selector for the full-line GEMM kernel (plain epilogue).  Row-major operands, WRONG RESULTS by design:
 * bit mask 1 no LDS-DMA after the prologue, 2 no LDS fragment reads, 4 no vmcnt wait at the hand-off, 8 no barrier
 * (supported: 1, 2, 4, 8, 12, 15; tools/gemm_fl_dbg.py).  CORRECT, bit-identical results, every layout: 32 = the early
 * LDS-DMA piece placement (the product schedule whenever A is row-major), 128 = the late placement (the product
 * schedule of the dW layout).  tools/gemm_piece_ab.py */
int tamd_gemm_set_dbg(int dbg);

/* Phase trace of the attention forward kernel: while `buf` (uint64[32], device memory) is set, workgroup 0 of every
 * tamd_attn_fwd launch stores per-wave shader-clock sums of its tile-loop phases at buf[wave * 8 + phase]
 * (0 tile-load issue, 1 K.Q^T, 2 mask + softmax, 3 P.V, 4 vmcnt wait, 5 barrier).  NULL switches it off. */
int tamd_attn_set_trace(void* buf);

/* Hardware-semantics probe (one wave): which = 0 mfma32, 1 mfma16, 2 ds_read_b64_tr_b16, 3 lane exchanges,
 * 4 direct-to-LDS load.  in: 4096 u32, in2: 64 u32, out: 4096 u32.  Used by tests/test_gpu_probe.py to
 * check the CPU execution model in tests/hipemu against the silicon; not on any product path. */
int tamd_probe(const void* in, const void* in2, void* out, int which, int dtype, tamd_stream_t stream);

/* Diagnostic: global -> LDS streaming rate of `blocks` 512-thread workgroups with a GEMM-tile address pattern
 * (`seg` contiguous bytes per row, rows `row_stride` bytes apart); mode 0 = LDS-DMA, 1 = register staged. */
int tamd_bw_probe(const void* buf, size_t bytes, int seg, size_t row_stride, int iters, int mode, int blocks,
                  void* sink, tamd_stream_t stream);

/* MFMA power probe: `blocks` workgroups of 4 waves run `iters` rounds of 32 x 32x32x16 (mode 0) or 64 x 16x16x32 (mode 1)
 * bf16 MFMAs per wave (1.05 MFLOP per wave and round either way) on operand bits from `in` (61 x 256 uint32) and store
 * {shader ticks, 100 MHz ticks} per workgroup in clk.  tools/mfma_power.py */
int tamd_mfma_power(const void* in, int iters, int mode, int blocks, void* clk, void* sink, tamd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TAMD_DIAG_H_ */
