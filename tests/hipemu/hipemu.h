// hipemu.h -- a small lane-accurate CPU execution model for HIP kernels (TEST INFRASTRUCTURE).
//
// Purpose: /root/repo has no GPU at build time, so the *same kernel sources* that hipcc
// compiles for gfx950 are also compiled by host clang against this header set and run on
// the CPU, thread for thread: every HIP thread is a ucontext fiber, a workgroup is a set of
// fibers scheduled round-robin on one OS thread, __syncthreads() and the wave-collective
// operations (shuffles, MFMA, ds_read_b64_tr_b16, permlane32_swap, direct-to-LDS loads) are
// rendez-vous points implemented exactly as tamd_device.h documents the gfx950 behaviour.
// This validates index math, LDS layouts, fragment mappings and barrier structure on CPU;
// tests/probe_hw.py validates the modelled instruction semantics themselves on the GPU.
//
// Not a product path: only tests/ builds or loads anything under tests/hipemu.
#pragma once
#include <ucontext.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

namespace hipemu {

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

constexpr int kWave = 64;
constexpr size_t kStackBytes = 256 * 1024;
constexpr size_t kMaxDynSmem = 160 * 1024;

struct WaveState {
  int arrived = 0;
  uint64_t gen = 0;
  int live = 0;  // fibers of this wave that have not returned
  // collective scratch: per-lane input / output slots
  alignas(16) unsigned char in[kWave][64];
  alignas(16) unsigned char out[kWave][64];
  const void* ptr_in[kWave];
};

struct PendingDma {  // an issued, not yet landed direct-to-LDS load of one lane
  char* dst;
  unsigned char data[16];
  int size = 16;
};

struct Fiber {
  std::vector<PendingDma> pending;
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
  dim3 tid;
  int linear = 0;
};

struct Block {
  dim3 bid, bdim, gdim;
  std::vector<Fiber> fibers;
  std::vector<WaveState> waves;
  ucontext_t sched;
  int cur = -1;
  int live = 0;
  int bar_arrived = 0;
  uint64_t bar_gen = 0;
  uint64_t progress = 0;  // bumped on every arrival / completion (deadlock detection)
  alignas(16) char dyn_smem[kMaxDynSmem];
  size_t dyn_bytes = 0;
  const std::function<void()>* body = nullptr;
};

inline thread_local Block* g_blk = nullptr;

// statistics (bank-conflict model, optional)
struct Stats {
  std::atomic<uint64_t> lds_read16_instr{0}, lds_read16_cycles{0};
  std::atomic<uint64_t> lds_tr_instr{0}, lds_tr_cycles{0};
  std::atomic<uint64_t> lds_write8_instr{0}, lds_write8_cycles{0};
  std::atomic<uint64_t> mfma_instr{0};
};
inline Stats g_stats;
inline std::atomic<int> g_fail{0};

inline Fiber& cur_fiber() { return g_blk->fibers[g_blk->cur]; }
inline WaveState& cur_wave() { return g_blk->waves[cur_fiber().linear / kWave]; }
inline int cur_lane() { return cur_fiber().linear % kWave; }

inline void yield() {
  Block* b = g_blk;
  swapcontext(&b->fibers[b->cur].ctx, &b->sched);
}

inline void fiber_entry() {
  Block* b = g_blk;
  (*b->body)();
  Fiber& f = b->fibers[b->cur];
  f.done = true;
  b->live--;
  b->waves[f.linear / kWave].live--;
  b->progress++;
  swapcontext(&f.ctx, &b->sched);
}

inline void syncthreads() {
  Block* b = g_blk;
  const uint64_t g = b->bar_gen;
  b->bar_arrived++;
  b->progress++;
  if (b->bar_arrived == (int)b->fibers.size()) {  // every thread of the block must arrive
    b->bar_arrived = 0;
    b->bar_gen++;
    return;
  }
  while (b->bar_gen == g) yield();
}

// Wave rendez-vous: every lane deposits, the last arriver runs `compute`, everyone reads its slot.
template <typename Compute>
inline void wave_collective(Compute&& compute) {
  Block* b = g_blk;
  WaveState& w = cur_wave();
  const uint64_t g = w.gen;
  w.arrived++;
  b->progress++;
  const int wave_threads = (int)std::min<size_t>(kWave, b->fibers.size() - (size_t)(cur_fiber().linear / kWave) * kWave);
  if (w.arrived == wave_threads) {
    compute(w, wave_threads);
    w.arrived = 0;
    w.gen++;
    return;
  }
  while (w.gen == g) yield();
}

inline void run_block(Block* b, const std::function<void()>& body, dim3 bid, dim3 bdim, dim3 gdim, size_t smem) {
  g_blk = b;
  b->bid = bid;
  b->bdim = bdim;
  b->gdim = gdim;
  b->dyn_bytes = smem;
  b->body = &body;
  const int n = (int)(bdim.x * bdim.y * bdim.z);
  if ((int)b->fibers.size() != n) {
    for (auto& f : b->fibers) free(f.stack);
    b->fibers.assign(n, Fiber());
    for (auto& f : b->fibers) f.stack = (char*)malloc(kStackBytes);
  }
  b->waves.assign((n + kWave - 1) / kWave, WaveState());
  b->live = n;
  b->bar_arrived = 0;
  for (int i = 0; i < n; ++i) {
    Fiber& f = b->fibers[i];
    f.done = false;
    f.pending.clear();
    f.linear = i;
    f.tid = dim3(i % bdim.x, (i / bdim.x) % bdim.y, i / (bdim.x * bdim.y));
    b->waves[i / kWave].live++;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = kStackBytes;
    f.ctx.uc_link = &b->sched;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
  }
  while (b->live > 0) {
    const uint64_t before = b->progress;
    // HIPEMU_REVERSE=1 walks the fibers backwards: races between wave groups that share a phase show up in
    // one of the two orders (the DMA model poisons at issue and lands at the issuer's wait).
    static const bool reverse = getenv("HIPEMU_REVERSE") != nullptr;
    for (int k = 0; k < n; ++k) {
      const int i = reverse ? n - 1 - k : k;
      if (b->fibers[i].done) continue;
      b->cur = i;
      swapcontext(&b->sched, &b->fibers[i].ctx);
    }
    if (b->live > 0 && b->progress == before) {
      fprintf(stderr,
              "hipemu: DEADLOCK in block (%u,%u,%u): %d threads alive, none progressing "
              "(divergent __syncthreads or wave-collective with exited lanes)\n",
              bid.x, bid.y, bid.z, b->live);
      g_fail.store(1);
      // abandon the block: fibers are simply dropped
      return;
    }
  }
}

// OS threads a launch runs its blocks on (= blocks that make progress at the same time: what a persistent kernel with
// cross-block barriers may rely on)
inline unsigned launch_threads() {
  unsigned nthr = std::thread::hardware_concurrency();
  if (const char* e = getenv("HIPEMU_THREADS")) nthr = (unsigned)atoi(e);
  return nthr < 1 ? 1 : nthr;
}

inline int launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t smem) {
  if (smem > kMaxDynSmem) {
    fprintf(stderr, "hipemu: dynamic LDS request %zu exceeds 160 KiB\n", smem);
    g_fail.store(1);
    return 1;
  }
  const uint64_t nblk = (uint64_t)grid.x * grid.y * grid.z;
  std::atomic<uint64_t> next{0};
  unsigned nthr = launch_threads();
  if (nthr > nblk) nthr = (unsigned)nblk;
  auto worker = [&]() {
    Block* b = new Block();
    for (;;) {
      const uint64_t i = next.fetch_add(1);
      if (i >= nblk) break;
      dim3 bid((unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((uint64_t)grid.x * grid.y)));
      run_block(b, body, bid, block, grid, smem);
    }
    for (auto& f : b->fibers) free(f.stack);
    delete b;
  };
  if (nthr == 1) {
    worker();
  } else {
    std::vector<std::thread> ts;
    for (unsigned t = 0; t < nthr; ++t) ts.emplace_back(worker);
    for (auto& t : ts) t.join();
  }
  return g_fail.load();
}

}  // namespace hipemu
