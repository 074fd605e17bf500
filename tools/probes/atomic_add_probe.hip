// fp32 atomic-add throughput in the access pattern a ONE-PASS attention backward would produce: the dK/dV
// kernel would also emit dQ, i.e. every (query tile x key block) pair adds a [64 x 128] fp32 partial (32 KiB) into the dQ
// accumulator -- 8.6 GB of atomic traffic per Llama-3-8B layer at 128-row query tiles, 17 GB at 64.  What does the memory
// system sustain for that?  Each workgroup (256 threads) adds NT tiles of 32 KiB, one dword per lane per instruction
// (global_atomic_add_f32, no return), consecutive lanes on consecutive dwords.
//   mode 0: private tiles, large footprint (every add goes to a different line of a 1 GiB buffer)      -> DRAM-side rate
//   mode 1: private tile, re-added NT times (32 KiB per workgroup: L2-resident)                         -> L2 atomic rate
//   mode 2: 8 workgroups of ONE XCD (blockIdx & 7 equal) share each tile                                -> contention inside an L2
//   mode 3: 8 workgroups on 8 DIFFERENT XCDs share each tile                                            -> contention across L2s
//   mode 4: plain stores of the same data (what the split backward's dQ kernel does once per tile row)  -> the yardstick
// Build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/probes/atomic_add_probe.hip -o /tmp/aprobe && /tmp/aprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kTileFloats = 64 * 128;  // 32 KiB

__global__ __launch_bounds__(256) void probe(float* buf, size_t tiles_in_buf, int nt, int mode) {
  const int wg = blockIdx.x;
  float v[4];
  for (int i = 0; i < 4; ++i) v[i] = 1.0f + 0.001f * (threadIdx.x + i);
  for (int t = 0; t < nt; ++t) {
    size_t tile;
    if (mode == 0 || mode == 4) tile = ((size_t)wg * nt + t) % tiles_in_buf;
    else if (mode == 1) tile = wg;
    else if (mode == 2) tile = (size_t)(wg & 7) * 4096 + ((wg >> 6) * 64 + t) % 4096;          // wg>>3 & 7: the 8 sharers
    else tile = (size_t)(wg >> 3) * 64 + t;                                                     // wg & 7: the 8 sharers (XCDs)
    float* p = buf + (tile % tiles_in_buf) * kTileFloats;
#pragma unroll
    for (int j = 0; j < kTileFloats / 256 / 4; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float* q = p + (size_t)(j * 4 + i) * 256 + threadIdx.x;
        if (mode == 4)
          __builtin_nontemporal_store(v[i], q);
        else
          __hip_atomic_fetch_add(q, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

int main() {
  const size_t tiles = 32768;  // 1 GiB
  float* buf;
  if (hipMalloc(&buf, tiles * kTileFloats * sizeof(float)) != hipSuccess) return 1;
  hipMemset(buf, 0, tiles * kTileFloats * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[] = {"private tiles, 1 GiB footprint", "private tile, L2-resident", "8 sharers inside one XCD",
                         "8 sharers on 8 XCDs", "plain stores (yardstick)"};
  for (int mode = 0; mode < 5; ++mode) {
    for (int wgs : {256, 1024, 2048}) {
      const int nt = 64;
      hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 0, 0, buf, tiles, 4, mode);  // warm
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 0, 0, buf, tiles, nt, mode);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)wgs * nt * kTileFloats * 4.0;
      printf("{\"mode\": %d, \"what\": \"%s\", \"workgroups\": %d, \"tiles_per_wg\": %d, \"ms\": %.3f, \"GB_per_s_added\": %.1f}\n", mode,
             names[mode], wgs, nt, ms, bytes / ms / 1e6);
      fflush(stdout);
    }
  }
  return 0;
}
