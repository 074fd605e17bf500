// torch_binding.cpp -- torch.ops.tamd.* : the dispatcher ops of the MI355X path, compiled.
//
// Boundary B3 of INTEGRATION.md: the C-ABI library (include/tamd.h; libtamd.so, or its CPU execution model in the test-suite)
// exposed to PyTorch through `TORCH_LIBRARY(tamd, ...)` -- the reference's own precedent for custom ops is
// src/transformers/integrations/moe.py:245-257 (torch.library op + fake + autograd).  This translation unit is host C++
// only (no device code): every op below checks its operands, allocates its outputs with torch's caching allocator, and
// calls the C-ABI entry points on the calling thread's current HIP stream.  Schemas and implementations live here; the
// fake (Meta) implementations and the autograd formulas are registered from Python (transformers_amd/ops.py,
// layer_ops.py) with torch.library.register_fake / register_autograd.
//
// Round 2 implemented the same ops as Python functions over ctypes (10-20 us of interpreter per launch: what bounded the
// small-model configurations); the composites -- a whole LlamaDecoderLayer / BertLayer forward or backward -- are now one
// dispatcher call that issues all their launches from C++.
//
// The C-ABI is bound at run time (tamd_torch_bind): libtamd.so in the product, libtamd_diag.so for the measurement tools,
// the CPU execution model of the same kernels (tests/hipemu) in the CPU test-suite -- the SAME host logic runs in all three.
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <torch/library.h>

#include <atomic>
#include <mutex>
#include <optional>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/tamd.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<at::Tensor>;

// ------------------------------------------------------------------------------------------------ the bound C ABI
#define TAMD_API_LIST(X)                                                                                              \
  X(tamd_abi_version) X(tamd_error_string) X(tamd_rmsnorm_fwd) X(tamd_norm_bwd_workspace_bytes) X(tamd_rmsnorm_bwd)    \
  X(tamd_layernorm_fwd) X(tamd_layernorm_bwd) X(tamd_layernorm_dropout_fwd) X(tamd_layernorm_dropout_bwd)             \
  X(tamd_rope_inplace) X(tamd_embedding_fwd) X(tamd_embedding_bwd_workspace_bytes) X(tamd_embedding_bwd)              \
  X(tamd_bert_embeddings_fwd) X(tamd_swiglu_fwd) X(tamd_swiglu_bwd) X(tamd_bias_act_fwd) X(tamd_bias_act_bwd)         \
  X(tamd_add) X(tamd_adamw_step) X(tamd_mt_sumsq) X(tamd_mt_norm_finish) X(tamd_mt_scale) X(tamd_mt_adamw_step) X(tamd_colsum_workspace_bytes) X(tamd_colsum) X(tamd_transpose)                      \
  X(tamd_cross_entropy_fwd) X(tamd_cross_entropy_bwd) X(tamd_gemm) X(tamd_gemm_workspace_bytes) X(tamd_gemm_ws)       \
  X(tamd_gemm_swiglu) X(tamd_gemm_swiglu_bwd) X(tamd_gemm_colscale) X(tamd_gemm_seg) X(tamd_gemm_group_workspace_bytes) X(tamd_gemm_group) X(tamd_attn_fwd) X(tamd_attn_bwd)   \
  X(tamd_attn_decode_workspace_bytes) X(tamd_attn_decode)

struct Api {
  void* handle = nullptr;
  bool emulated = false;  // the CPU execution model: CPU tensors are its device memory, there is no stream
  std::string path;
#define X(name) decltype(&::name) name = nullptr;
  TAMD_API_LIST(X)
#undef X
};
std::atomic<const Api*> g_api{nullptr};
std::mutex g_bind_mutex;
std::vector<Api*> g_bound;  // every library ever bound stays loaded (an op in flight may still hold the table)

const Api& api() {
  const Api* a = g_api.load(std::memory_order_acquire);
  TORCH_CHECK(a != nullptr, "tamd: no kernel library is bound (transformers_amd.ops binds libtamd.so on first use; "
                            "build it with `python -m transformers_amd.build`)");
  return *a;
}

void check(int code, const char* what) {
  if (code != 0) TORCH_CHECK(false, "tamd: ", what, " failed: ", api().tamd_error_string(code), " (code ", code, ")");
}

// ------------------------------------------------------------------------------------------------ operand helpers
int code_of(const Tensor& t) {
  switch (t.scalar_type()) {
    case at::kBFloat16: return TAMD_BF16;
    case at::kHalf: return TAMD_F16;
    case at::kFloat: return TAMD_F32;
    default: TORCH_CHECK(false, "tamd: unsupported dtype ", t.scalar_type());
  }
}

// Operand checks (ops.py `_prep` of round 2): every operand on a GPU -- there is no CPU implementation -- and all on ONE
// device; the launch runs with that device current.
struct Launch {
  c10::OptionalDeviceGuard guard;
  tamd_stream_t stream = nullptr;
  explicit Launch(std::initializer_list<const Tensor*> ts) {
    const Api& a = api();
    const Tensor* first = nullptr;
    for (const Tensor* t : ts) {
      if (t == nullptr || !t->defined()) continue;
      if (!a.emulated)
        TORCH_CHECK(t->is_cuda(), "tamd: op received a ", t->device(), " tensor; the HIP kernels need GPU memory "
                                  "(the MI355X path has no CPU/eager fallback)");
      if (first == nullptr)
        first = t;
      else
        TORCH_CHECK(t->device() == first->device(), "tamd: op operands live on different devices: ", first->device(),
                    " and ", t->device());
    }
    if (first != nullptr && first->is_cuda()) {
      guard.reset_device(first->device());
      stream = (tamd_stream_t)c10::hip::getCurrentHIPStream(first->device().index()).stream();
    }
  }
};
const Tensor* p(const OptTensor& t) { return t.has_value() ? &*t : nullptr; }
const void* ptr(const Tensor& t) { return t.defined() ? t.const_data_ptr() : nullptr; }
const void* ptr(const OptTensor& t) { return t.has_value() && t->defined() ? t->const_data_ptr() : nullptr; }
void* mptr(const Tensor& t) { return t.defined() ? t.mutable_data_ptr() : nullptr; }
Tensor contig(const Tensor& t) { return t.is_contiguous() ? t : t.contiguous(); }
Tensor nothing(const Tensor& like) { return at::empty({0}, like.options()); }  // (an op cannot return `Tensor?`)
int64_t rows_of(const Tensor& x) { return x.numel() / x.size(-1); }
Tensor f32_like(const Tensor& x, at::IntArrayRef shape) { return at::empty(shape, x.options().dtype(at::kFloat)); }

// ------------------------------------------------------------------------------------------------ GEMM event log
// bench.py's `roofline` object: HIP-event pairs around every MFMA-GEMM launch while enabled (the launches happen in here,
// out of Python's reach).  Events are recorded on the launch stream; tamd_torch_gemm_log_summary synchronises them.
struct GemmRecord {
  double flops, bytes;
  hipEvent_t start, stop;
  bool fused_bwd;  // the launch carries the SiLU*up backward in its way out (tamd_gemm_swiglu_bwd): elementwise work, formerly a
                   // kernel of its own, whose time the log counts and whose arithmetic it does not
};
std::atomic<bool> g_gemm_log{false};
std::mutex g_gemm_log_mutex;
std::vector<GemmRecord> g_gemm_records;

struct GemmTimerScope {
  bool on;
  GemmRecord rec{};
  hipStream_t s;
  GemmTimerScope(double flops, double bytes, tamd_stream_t stream, bool fused_bwd = false)
      : on(g_gemm_log.load(std::memory_order_relaxed) && !api().emulated), s((hipStream_t)stream) {
    if (!on) return;
    rec.fused_bwd = fused_bwd;
    // a launch being captured into a HIP graph (graph_stack.py, a user's torch.cuda.graph) has no duration of its own:
    // an event recorded there becomes a graph node and hipEventElapsedTime on it fails
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
      on = false;
      return;
    }
    rec.flops = flops;
    rec.bytes = bytes;
    if (hipEventCreate(&rec.start) != hipSuccess || hipEventCreate(&rec.stop) != hipSuccess) {
      on = false;
      return;
    }
    (void)hipEventRecord(rec.start, s);
  }
  ~GemmTimerScope() {
    if (!on) return;
    (void)hipEventRecord(rec.stop, s);
    std::lock_guard<std::mutex> lock(g_gemm_log_mutex);
    g_gemm_records.push_back(rec);
  }
};

// ================================================================================================ kernel launchers
// One function per C-ABI entry point (what ops.py called raw_*): argument checks, output allocation, launch.

// -> (y, h, rstd): h = x + residual, or x itself when there is no residual
std::tuple<Tensor, Tensor, Tensor> k_rmsnorm_fwd(const Tensor& x, const Tensor& w_, double eps, const OptTensor& residual) {
  const int64_t cols = x.size(-1);
  Tensor x2 = contig(x).view({-1, cols});
  Tensor r2 = residual ? contig(*residual).view({-1, cols}) : Tensor();
  Launch L({&x2, &w_, &r2});
  Tensor w = contig(w_);
  Tensor y = at::empty_like(x2);
  Tensor h = r2.defined() ? at::empty_like(x2) : x2;
  Tensor rstd = f32_like(x2, {x2.size(0)});
  check(api().tamd_rmsnorm_fwd(ptr(x2), ptr(r2), ptr(w), mptr(y), r2.defined() ? mptr(h) : nullptr,
                               (float*)mptr(rstd), x2.size(0), cols, (float)eps, code_of(x2), L.stream),
        "tamd_rmsnorm_fwd");
  return {y.view(x.sizes()), h.view(x.sizes()), rstd};
}

std::tuple<Tensor, Tensor> k_rmsnorm_bwd(const Tensor& dy, const Tensor& h, const Tensor& w_, const Tensor& rstd,
                                         const OptTensor& dres) {
  const int64_t cols = h.size(-1);
  Tensor dy2 = contig(dy).view({-1, cols}), h2 = contig(h).view({-1, cols});
  Tensor dr2 = dres ? contig(*dres).view({-1, cols}) : Tensor();
  Launch L({&dy2, &h2, &w_, &dr2});
  Tensor w = contig(w_);
  const int64_t rows = h2.size(0);
  Tensor dx = at::empty_like(h2), dw = at::empty_like(w);
  const size_t nbytes = api().tamd_norm_bwd_workspace_bytes(rows, cols);
  Tensor ws = at::empty({(int64_t)nbytes}, h.options().dtype(at::kByte));
  check(api().tamd_rmsnorm_bwd(ptr(dy2), ptr(h2), ptr(w), (const float*)ptr(rstd), ptr(dr2), mptr(dx), mptr(dw), mptr(ws),
                               nbytes, rows, cols, code_of(h2), L.stream),
        "tamd_rmsnorm_bwd");
  return {dx.view(h.sizes()), dw};
}

// -> (y, h, mean, rstd)
std::tuple<Tensor, Tensor, Tensor, Tensor> k_layernorm_fwd(const Tensor& x, const Tensor& w_, const OptTensor& b_,
                                                           double eps, const OptTensor& residual) {
  const int64_t cols = x.size(-1);
  Tensor x2 = contig(x).view({-1, cols});
  Tensor r2 = residual ? contig(*residual).view({-1, cols}) : Tensor();
  Launch L({&x2, &w_, p(b_), &r2});
  Tensor w = contig(w_);
  Tensor b = b_ ? contig(*b_) : Tensor();
  Tensor y = at::empty_like(x2);
  Tensor h = r2.defined() ? at::empty_like(x2) : x2;
  Tensor mean = f32_like(x2, {x2.size(0)}), rstd = f32_like(x2, {x2.size(0)});
  check(api().tamd_layernorm_fwd(ptr(x2), ptr(r2), ptr(w), ptr(b), mptr(y), r2.defined() ? mptr(h) : nullptr,
                                 (float*)mptr(mean), (float*)mptr(rstd), x2.size(0), cols, (float)eps, code_of(x2),
                                 L.stream),
        "tamd_layernorm_fwd");
  return {y.view(x.sizes()), h.view(x.sizes()), mean, rstd};
}

// -> (dx, dw, db or undefined, column sums of dx or undefined)
std::tuple<Tensor, Tensor, Tensor, Tensor> k_layernorm_bwd(const Tensor& dy, const Tensor& h, const Tensor& w_,
                                                           const Tensor& mean, const Tensor& rstd, const OptTensor& dres,
                                                           bool need_db, bool need_colsum = false) {
  const int64_t cols = h.size(-1);
  Tensor dy2 = contig(dy).view({-1, cols}), h2 = contig(h).view({-1, cols});
  Tensor dr2 = dres ? contig(*dres).view({-1, cols}) : Tensor();
  Launch L({&dy2, &h2, &w_, &dr2});
  Tensor w = contig(w_);
  const int64_t rows = h2.size(0);
  Tensor dx = at::empty_like(h2), dw = at::empty_like(w);
  Tensor db = need_db ? at::empty_like(w) : Tensor();
  Tensor dc = need_colsum ? at::empty_like(w) : Tensor();
  const size_t nbytes = api().tamd_norm_bwd_workspace_bytes(rows, cols);
  Tensor ws = at::empty({(int64_t)nbytes}, h.options().dtype(at::kByte));
  check(api().tamd_layernorm_bwd(ptr(dy2), ptr(h2), ptr(w), (const float*)ptr(mean), (const float*)ptr(rstd), ptr(dr2),
                                 mptr(dx), mptr(dw), mptr(db), mptr(dc), mptr(ws), nbytes, rows, cols, code_of(h2),
                                 L.stream),
        "tamd_layernorm_bwd");
  return {dx.view(h.sizes()), dw, db, dc};
}

// h = dropout(x, p) + residual; y = LayerNorm(h)  ->  (y, h, mean, rstd)
std::tuple<Tensor, Tensor, Tensor, Tensor> k_layernorm_dropout_fwd(const Tensor& x, const Tensor& w_, const OptTensor& b_,
                                                                   double eps, const Tensor& residual, double dropout_p,
                                                                   int64_t seed, const uint64_t* seed_dev = nullptr) {
  const int64_t cols = x.size(-1);
  Tensor x2 = contig(x).view({-1, cols}), r2 = contig(residual).view({-1, cols});
  Launch L({&x2, &w_, p(b_), &r2});
  Tensor w = contig(w_);
  Tensor b = b_ ? contig(*b_) : Tensor();
  Tensor y = at::empty_like(x2), h = at::empty_like(x2);
  Tensor mean = f32_like(x2, {x2.size(0)}), rstd = f32_like(x2, {x2.size(0)});
  check(api().tamd_layernorm_dropout_fwd(ptr(x2), ptr(r2), ptr(w), ptr(b), mptr(y), mptr(h), (float*)mptr(mean),
                                         (float*)mptr(rstd), x2.size(0), cols, (float)eps, (float)dropout_p,
                                         (uint64_t)seed, seed_dev, code_of(x2), L.stream),
        "tamd_layernorm_dropout_fwd");
  return {y.view(x.sizes()), h.view(x.sizes()), mean, rstd};
}

// -> (dx = gradient of the residual input, dxd = gradient of the dropped-out input, dw, db or undefined, column sums of dxd
//     or undefined)
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> k_layernorm_dropout_bwd(const Tensor& dy, const Tensor& h, const Tensor& w_,
                                                                           const Tensor& mean, const Tensor& rstd,
                                                                           double dropout_p, int64_t seed,
                                                                           const OptTensor& dres, bool need_db,
                                                                           bool need_colsum = false,
                                                                           const uint64_t* seed_dev = nullptr) {
  const int64_t cols = h.size(-1);
  Tensor dy2 = contig(dy).view({-1, cols}), h2 = contig(h).view({-1, cols});
  Tensor dr2 = dres ? contig(*dres).view({-1, cols}) : Tensor();
  Launch L({&dy2, &h2, &w_, &dr2});
  Tensor w = contig(w_);
  const int64_t rows = h2.size(0);
  Tensor dx = at::empty_like(h2), dxd = at::empty_like(h2), dw = at::empty_like(w);
  Tensor db = need_db ? at::empty_like(w) : Tensor();
  Tensor dc = need_colsum ? at::empty_like(w) : Tensor();
  const size_t nbytes = api().tamd_norm_bwd_workspace_bytes(rows, cols);
  Tensor ws = at::empty({(int64_t)nbytes}, h.options().dtype(at::kByte));
  check(api().tamd_layernorm_dropout_bwd(ptr(dy2), ptr(h2), ptr(w), (const float*)ptr(mean), (const float*)ptr(rstd),
                                         ptr(dr2), mptr(dx), mptr(dxd), mptr(dw), mptr(db), mptr(dc), mptr(ws), nbytes, rows,
                                         cols, (float)dropout_p, (uint64_t)seed, seed_dev, code_of(h2), L.stream),
        "tamd_layernorm_dropout_bwd");
  return {dx.view(h.sizes()), dxd.view(h.sizes()), dw, db, dc};
}

// in-place rotary on the first `nheads` heads of every row of x2d [tokens, row_stride]; the first `q_heads` of them leave
// multiplied by `q_scale` before their one rounding (include/tamd.h: the pre-scaled queries of the attention kernels)
void k_rope_(const Tensor& x2d, const Tensor& cos_, const Tensor& sin_, int64_t seq, int64_t nheads, int64_t head_dim,
             bool conj, int64_t q_heads = 0, double q_scale = 1.0) {
  Launch L({&x2d, &cos_, &sin_});
  TORCH_CHECK(x2d.dim() == 2 && x2d.stride(1) == 1, "tamd: rope_ takes a row-major [tokens, row] matrix");
  Tensor cos = contig(cos_), sin = contig(sin_);
  if (cos.scalar_type() != x2d.scalar_type()) {
    cos = cos.to(x2d.scalar_type());
    sin = sin.to(x2d.scalar_type());
  }
  const int64_t cos_batch = cos.dim() == 3 ? cos.size(0) : 1;
  check(api().tamd_rope_inplace(mptr(x2d), ptr(cos), ptr(sin), x2d.size(0), seq, x2d.stride(0), nheads, head_dim,
                                cos_batch, (int)conj, q_heads, (float)q_scale, code_of(x2d), L.stream),
        "tamd_rope_inplace");
}

Tensor k_embedding_fwd(const Tensor& ids, const Tensor& table_) {
  Launch L({&ids, &table_});
  Tensor ids_c = contig(ids);
  if (ids_c.scalar_type() != at::kLong) ids_c = ids_c.to(at::kLong);
  Tensor table = contig(table_);
  auto shape = ids.sizes().vec();
  shape.push_back(table.size(1));
  Tensor out = at::empty(shape, table.options());
  check(api().tamd_embedding_fwd((const int64_t*)ptr(ids_c), ptr(table), mptr(out), ids_c.numel(), table.size(0),
                                 table.size(1), nullptr, code_of(table), L.stream),
        "tamd_embedding_fwd");
  return out;
}

// scatter-add of dout rows into a [vocab, dim] table gradient.  The token order comes from a stable sort of the ids
// (at::sort: 0.003 % of the Llama step in profiles/r03a_bench_kernel_stats.csv -- index plumbing; the accumulation, which
// is where the bytes are, is tamd_embedding_bwd's segmented sum).
Tensor k_embedding_bwd(const Tensor& ids, const Tensor& dout, int64_t vocab, int64_t padding_idx) {
  Launch L({&ids, &dout});
  const int64_t dim = dout.size(-1);
  Tensor flat = contig(ids).view({-1}).to(at::kLong);
  auto sorted = at::sort(flat, /*stable=*/true, /*dim=*/-1, /*descending=*/false);
  Tensor sorted_ids = std::get<0>(sorted), perm = std::get<1>(sorted);
  Tensor dtable = at::zeros({vocab, dim}, dout.options());
  Tensor d2 = contig(dout).view({-1, dim});
  const size_t ws_bytes = api().tamd_embedding_bwd_workspace_bytes(flat.numel(), dim);
  Tensor ws = at::empty({(int64_t)ws_bytes}, dout.options().dtype(at::kByte));
  check(api().tamd_embedding_bwd((const int64_t*)ptr(sorted_ids), (const int64_t*)ptr(perm), ptr(d2), mptr(dtable), mptr(ws),
                                 ws_bytes, flat.numel(), vocab, dim, padding_idx, code_of(d2), L.stream),
        "tamd_embedding_bwd");
  return dtable;
}

// -> (out, pre-LayerNorm sum or undefined, mean, rstd)
std::tuple<Tensor, Tensor, Tensor, Tensor> k_bert_embeddings_fwd(const Tensor& input_ids, const Tensor& token_type_ids,
                                                                 const Tensor& position_ids, const Tensor& word_,
                                                                 const Tensor& typ_, const Tensor& pos_, const Tensor& ln_w_,
                                                                 const Tensor& ln_b_, double eps, bool keep_pre_ln) {
  Launch L({&input_ids, &word_, &typ_, &pos_, &ln_w_, &ln_b_});
  const int64_t n = input_ids.numel(), dim = word_.size(1);
  auto shape = input_ids.sizes().vec();
  shape.push_back(dim);
  Tensor out = at::empty(shape, word_.options());
  Tensor pre = keep_pre_ln ? at::empty_like(out) : Tensor();
  Tensor mean = f32_like(word_, {n}), rstd = f32_like(word_, {n});
  Tensor iid = contig(input_ids).to(at::kLong), tid = contig(token_type_ids).to(at::kLong),
         pid = contig(position_ids).to(at::kLong);
  Tensor word = contig(word_), typ = contig(typ_), pos = contig(pos_), ln_w = contig(ln_w_), ln_b = contig(ln_b_);
  check(api().tamd_bert_embeddings_fwd((const int64_t*)ptr(iid), (const int64_t*)ptr(tid), (const int64_t*)ptr(pid),
                                       ptr(word), ptr(typ), ptr(pos), ptr(ln_w), ptr(ln_b), mptr(out), mptr(pre),
                                       (float*)mptr(mean), (float*)mptr(rstd), n, dim, word.size(0), typ.size(0),
                                       pos.size(0), (float)eps, code_of(word), L.stream),
        "tamd_bert_embeddings_fwd");
  return {out, pre, mean, rstd};
}

// gu [T, 2I] = [gate | up]  ->  act [T, I]
Tensor k_swiglu_fwd(const Tensor& gu) {
  Launch L({&gu});
  TORCH_CHECK(gu.dim() == 2 && gu.stride(1) == 1, "tamd: swiglu_fwd takes a row-major [tokens, 2I] matrix");
  const int64_t t = gu.size(0), inter = gu.size(1) / 2;
  Tensor act = at::empty({t, inter}, gu.options());
  check(api().tamd_swiglu_fwd(ptr(gu), (const char*)ptr(gu) + inter * gu.element_size(), mptr(act), t, inter, gu.stride(0),
                              act.stride(0), code_of(gu), L.stream),
        "tamd_swiglu_fwd");
  return act;
}

// -> (d_gate | d_up [T, 2I], act [T, I] or undefined)
std::tuple<Tensor, Tensor> k_swiglu_bwd(const Tensor& gu, const Tensor& dact_, bool want_act) {
  Launch L({&gu, &dact_});
  TORCH_CHECK(gu.dim() == 2 && gu.stride(1) == 1, "tamd: swiglu_bwd takes a row-major [tokens, 2I] matrix");
  const int64_t t = gu.size(0), inter = gu.size(1) / 2;
  Tensor dgu = at::empty_like(gu);
  Tensor dact = contig(dact_);
  Tensor act = want_act ? at::empty_like(dact) : Tensor();
  const int64_t esz = gu.element_size();
  TORCH_CHECK(dgu.stride(0) == gu.stride(0) || gu.is_contiguous(), "tamd: swiglu_bwd needs a dense gate|up matrix");
  check(api().tamd_swiglu_bwd(ptr(gu), (const char*)ptr(gu) + inter * esz, ptr(dact), mptr(dgu),
                              (char*)mptr(dgu) + inter * esz, mptr(act), t, inter, gu.stride(0), dact.stride(0),
                              code_of(gu), L.stream),
        "tamd_swiglu_bwd");
  return {dgu, act};
}

Tensor k_bias_act_fwd(const Tensor& x, const OptTensor& bias, int64_t act) {
  Tensor x2 = contig(x).view({-1, x.size(-1)});
  Launch L({&x2, p(bias)});
  Tensor y = at::empty_like(x2);
  check(api().tamd_bias_act_fwd(ptr(x2), ptr(bias), mptr(y), x2.size(0), x2.size(1), (int)act, code_of(x2), L.stream),
        "tamd_bias_act_fwd");
  return y.view(x.sizes());
}

// -> (dx, column sums of dx [cols] or undefined)
std::tuple<Tensor, Tensor> k_bias_act_bwd(const Tensor& x, const OptTensor& bias, const Tensor& dy, int64_t act,
                                          bool need_colsum = false) {
  Tensor x2 = contig(x).view({-1, x.size(-1)}), dy2 = contig(dy).view({-1, x.size(-1)});
  Launch L({&x2, p(bias), &dy2});
  Tensor dx = at::empty_like(x2);
  Tensor dc, ws;
  size_t nbytes = 0;
  if (need_colsum) {
    dc = at::empty({x2.size(1)}, x2.options());
    nbytes = api().tamd_colsum_workspace_bytes(x2.size(0), x2.size(1));
    ws = at::empty({(int64_t)nbytes}, x2.options().dtype(at::kByte));
  }
  check(api().tamd_bias_act_bwd(ptr(x2), ptr(bias), ptr(dy2), mptr(dx), mptr(dc), mptr(ws), nbytes, x2.size(0), x2.size(1),
                                (int)act, code_of(x2), L.stream),
        "tamd_bias_act_bwd");
  return {dx.view(x.sizes()), dc};
}

Tensor k_add(const Tensor& a_, const Tensor& b_) {
  Tensor a = contig(a_), b = contig(b_);
  Launch L({&a, &b});
  Tensor out = at::empty_like(a);
  check(api().tamd_add(ptr(a), ptr(b), mptr(out), a.numel(), code_of(a), L.stream), "tamd_add");
  return out;
}

void k_adamw_step_(const Tensor& pp, const Tensor& g, const Tensor& m, const Tensor& v, double lr, double beta1,
                   double beta2, double eps, double weight_decay, int64_t step, double grad_scale) {
  Launch L({&pp, &g, &m, &v});
  for (const Tensor* t : {&pp, &g, &m, &v})
    TORCH_CHECK(t->is_contiguous(), "tamd: adamw_step needs contiguous tensors (parameters, gradients and moments)");
  TORCH_CHECK(g.scalar_type() == pp.scalar_type() && m.scalar_type() == v.scalar_type() &&
                  (m.scalar_type() == pp.scalar_type() || m.scalar_type() == at::kFloat),
              "tamd: adamw_step dtypes: p/g ", pp.scalar_type(), "/", g.scalar_type(), ", m/v ", m.scalar_type(), "/",
              v.scalar_type());
  check(api().tamd_adamw_step(mptr(pp), ptr(g), mptr(m), mptr(v), pp.numel(), lr, beta1, beta2, eps, weight_decay, step,
                              grad_scale, code_of(pp), code_of(m), L.stream),
        "tamd_adamw_step");
}

// ---- multi-tensor step (include/tamd.h "multi-tensor step"): `table` is the int64 device table the caller built
// (transformers_amd/optim.py MtTable); the tensors it points at are the caller's to keep alive.
void check_mt_table(const Tensor& table, int64_t n_tensors, int64_t total_chunks) {
  TORCH_CHECK(table.scalar_type() == at::kLong && table.is_contiguous() && table.numel() >= 6 * n_tensors + 1,
              "tamd: multi-tensor table must be a contiguous int64 tensor of 6 * n_tensors + 1 words");
  TORCH_CHECK(n_tensors >= 0 && total_chunks >= 0 && n_tensors < (1 << 30), "tamd: multi-tensor table sizes");
}
void k_mt_sumsq(const Tensor& table, int64_t n_tensors, int64_t total_chunks, const Tensor& partials, int64_t dtype) {
  check_mt_table(table, n_tensors, total_chunks);
  TORCH_CHECK(partials.scalar_type() == at::kFloat && partials.is_contiguous() && partials.numel() >= total_chunks,
              "tamd: mt_sumsq needs `total_chunks` contiguous fp32 partials");
  Launch L({&table, &partials});
  check(api().tamd_mt_sumsq((const int64_t*)ptr(table), (int)n_tensors, total_chunks, (float*)mptr(partials), (int)dtype,
                            L.stream),
        "tamd_mt_sumsq");
}
void k_mt_norm_finish(const Tensor& partials, const Tensor& out, double max_norm) {
  TORCH_CHECK(partials.scalar_type() == at::kFloat && partials.is_contiguous() && out.scalar_type() == at::kFloat &&
                  out.is_contiguous() && out.numel() >= 2,
              "tamd: mt_norm_finish needs contiguous fp32 partials and an fp32 out[2]");
  Launch L({&partials, &out});
  check(api().tamd_mt_norm_finish((const float*)ptr(partials), partials.numel(), (float*)mptr(out), max_norm, L.stream),
        "tamd_mt_norm_finish");
}
void k_mt_scale_(const Tensor& table, int64_t n_tensors, int64_t total_chunks, const Tensor& coef, int64_t dtype) {
  check_mt_table(table, n_tensors, total_chunks);
  TORCH_CHECK(coef.scalar_type() == at::kFloat && coef.numel() >= 1, "tamd: mt_scale_ needs an fp32 coefficient in device memory");
  Launch L({&table, &coef});
  check(api().tamd_mt_scale((const int64_t*)ptr(table), (int)n_tensors, total_chunks, (const float*)ptr(coef), (int)dtype,
                            L.stream),
        "tamd_mt_scale");
}
void k_mt_adamw_step_(const Tensor& table, int64_t n_tensors, int64_t total_chunks, double lr, double beta1, double beta2,
                      double eps, double weight_decay, int64_t step, double grad_scale, const OptTensor& grad_scale_dev,
                      int64_t dtype, int64_t state_dtype) {
  check_mt_table(table, n_tensors, total_chunks);
  if (grad_scale_dev.has_value() && grad_scale_dev->defined())
    TORCH_CHECK(grad_scale_dev->scalar_type() == at::kFloat && grad_scale_dev->numel() >= 1,
                "tamd: mt_adamw_step_ grad_scale_dev must be an fp32 scalar in device memory");
  Launch L({&table, p(grad_scale_dev)});
  check(api().tamd_mt_adamw_step((const int64_t*)ptr(table), (int)n_tensors, total_chunks, lr, beta1, beta2, eps,
                                 weight_decay, step, grad_scale, (const float*)ptr(grad_scale_dev), (int)dtype,
                                 (int)state_dtype, L.stream),
        "tamd_mt_adamw_step");
}

Tensor k_colsum(const Tensor& x2d) {
  Launch L({&x2d});
  TORCH_CHECK(x2d.dim() == 2 && x2d.stride(1) == 1, "tamd: colsum takes a row-major matrix");
  const int64_t rows = x2d.size(0), cols = x2d.size(1);
  Tensor out = at::empty({cols}, x2d.options());
  const size_t nbytes = api().tamd_colsum_workspace_bytes(rows, cols);
  Tensor ws = at::empty({(int64_t)nbytes}, x2d.options().dtype(at::kByte));
  check(api().tamd_colsum(ptr(x2d), mptr(out), mptr(ws), nbytes, rows, cols, x2d.stride(0), code_of(x2d), L.stream),
        "tamd_colsum");
  return out;
}

Tensor k_transpose(const Tensor& x2d) {
  Launch L({&x2d});
  const int64_t rows = x2d.size(0), cols = x2d.size(1);
  Tensor out = at::empty({cols, rows}, x2d.options());
  check(api().tamd_transpose(ptr(x2d), mptr(out), rows, cols, x2d.stride(0), out.stride(0), code_of(x2d), L.stream),
        "tamd_transpose");
  return out;
}

// -> (lse [t], row_loss [t])
std::tuple<Tensor, Tensor> k_cross_entropy_fwd(const Tensor& logits2d, const Tensor& labels, int64_t ignore_index) {
  Launch L({&logits2d, &labels});
  const int64_t t = logits2d.size(0), v = logits2d.size(1);
  Tensor lse = f32_like(logits2d, {t}), row_loss = f32_like(logits2d, {t});
  check(api().tamd_cross_entropy_fwd(ptr(logits2d), (const int64_t*)ptr(labels), (float*)mptr(lse), (float*)mptr(row_loss),
                                     t, v, logits2d.stride(0), ignore_index, code_of(logits2d), L.stream),
        "tamd_cross_entropy_fwd");
  return {lse, row_loss};
}

// -> dlogits [t, v]: a view of a fresh [t, ld] buffer (ld = the row stride of logits2d) whose padding columns v .. ld-1 the
// kernel zeroes; `padded` returns that whole buffer (a K-padded GEMM operand)
Tensor k_cross_entropy_bwd(const Tensor& logits2d, const Tensor& labels, const Tensor& lse, const Tensor& gscale,
                           int64_t ignore_index, bool padded) {
  Launch L({&logits2d, &labels, &lse, &gscale});
  const int64_t t = logits2d.size(0), v = logits2d.size(1), ld = logits2d.stride(0);
  TORCH_CHECK(logits2d.stride(1) == 1 && ld >= v, "tamd: cross_entropy_bwd needs row-major logits");
  Tensor buf = at::empty({t, ld}, logits2d.options());
  check(api().tamd_cross_entropy_bwd(ptr(logits2d), (const int64_t*)ptr(labels), (const float*)ptr(lse),
                                     (const float*)ptr(gscale), mptr(buf), t, v, ld, ignore_index, code_of(logits2d),
                                     L.stream),
        "tamd_cross_entropy_bwd");
  return (padded || ld == v) ? buf : buf.narrow(1, 0, v);
}

// C[M,N] = epi(A . B^T).  a: [M,K] (or [K,M] if a_km); b: [N,K] (or [K,N] if b_kn).  sched: 0 = library default (with the
// split-K policy), else a TAMD_GEMM_SCHED_* hint >> 8 (no split-K).
Tensor k_gemm(const Tensor& a, const Tensor& b, bool a_km, bool b_kn, const OptTensor& bias, const OptTensor& residual_,
              int64_t epilogue, int64_t act, const OptTensor& out_, int64_t sched) {
  Tensor out = out_ ? *out_ : Tensor();
  Launch L({&a, &b, p(bias), p(residual_), &out});
  TORCH_CHECK(a.dim() == 2 && b.dim() == 2 && a.stride(1) == 1 && b.stride(1) == 1, "tamd: gemm operands must be 2-D row-major views");
  const int64_t k_a = a_km ? a.size(0) : a.size(1), m = a_km ? a.size(1) : a.size(0);
  const int64_t k_b = b_kn ? b.size(0) : b.size(1), n = b_kn ? b.size(1) : b.size(0);
  TORCH_CHECK(k_a == k_b, "tamd: gemm K mismatch: ", a.sizes(), " x ", b.sizes(), " (a_km=", a_km, ", b_kn=", b_kn, ")");
  if (!out.defined()) {
    out = at::empty({m, n}, a.options());
  } else {  // a caller's destination (a DDP bucket view, a `.grad` buffer): the kernel stores 16 bytes per lane into it
    TORCH_CHECK(out.dim() == 2 && out.size(0) == m && out.size(1) == n, "tamd: gemm out has shape ", out.sizes(), ", expected [", m,
                ", ", n, "]");
    TORCH_CHECK(out.scalar_type() == a.scalar_type() && out.device() == a.device(), "tamd: gemm out must have the operands' dtype "
                "and device");
    TORCH_CHECK(out.stride(1) == 1 && out.stride(0) % 8 == 0 && (reinterpret_cast<uintptr_t>(out.const_data_ptr()) % 16) == 0,
                "tamd: gemm out must be a row-major view with a 16-byte aligned base and a row stride that is a multiple of 8 "
                "elements");
  }
  const int flags = (a_km ? TAMD_GEMM_A_KM : 0) | (b_kn ? TAMD_GEMM_B_KN : 0) | (int)(sched << 8);
  Tensor residual = residual_ ? *residual_ : Tensor();
  size_t ws_bytes = 0;
  if (residual.defined()) {
    // the kernel reads R as the output's element type through 16-byte accesses: checked here because the rewrite below
    // takes R out of the C call (whose own checks would otherwise catch a stray dtype / stride / alignment)
    TORCH_CHECK(residual.scalar_type() == out.scalar_type(), "tamd: gemm residual dtype ", residual.scalar_type(),
                " != output dtype ", out.scalar_type());
    TORCH_CHECK(residual.dim() == 2 && residual.stride(1) == 1 && residual.stride(0) % 8 == 0 &&
                    (reinterpret_cast<uintptr_t>(residual.const_data_ptr()) % 16) == 0,
                "tamd: gemm residual must be a 2-D row-major view with a 16-byte aligned base and a row stride that is a "
                "multiple of 8 elements");
  }
  if (sched == 0)  // split-K when the tile grid cannot fill the GPU (the reduction applies bias / residual / accumulate)
    ws_bytes = api().tamd_gemm_workspace_bytes(m, n, k_a, flags & 3, (int)epilogue);
  const int64_t ldr = residual.defined() ? residual.stride(0) : 0;
  GemmTimerScope timer(2.0 * (double)m * (double)n * (double)k_a, 2.0 * ((double)m * k_a + (double)n * k_a + (double)m * n),
                       L.stream);
  if (ws_bytes) {  // split-K for tile grids that cannot fill the GPU: needs an fp32 workspace
    Tensor ws = at::empty({(int64_t)ws_bytes}, a.options().dtype(at::kByte));
    check(api().tamd_gemm_ws(ptr(a), ptr(b), mptr(out), ptr(bias), ptr(residual), m, n, k_a, a.stride(0), b.stride(0),
                             out.stride(0), ldr, flags, (int)epilogue, (int)act, code_of(a), mptr(ws), ws_bytes, L.stream),
          "tamd_gemm_ws");
  } else {
    check(api().tamd_gemm(ptr(a), ptr(b), mptr(out), ptr(bias), ptr(residual), m, n, k_a, a.stride(0), b.stride(0),
                          out.stride(0), ldr, flags, (int)epilogue, (int)act, code_of(a), L.stream),
          "tamd_gemm");
  }
  return out;
}
Tensor gemm_plain(const Tensor& a, const Tensor& b, bool a_km = false, bool b_kn = false, const OptTensor& bias = {},
                  const OptTensor& residual = {}, int64_t epilogue = TAMD_EPI_NONE, int64_t act = TAMD_ACT_NONE,
                  const OptTensor& out = {}) {
  return k_gemm(a, b, a_km, b_kn, bias, residual, epilogue, act, out, 0);
}

// x2 . w^T (+ bias) with the first scale_cols columns multiplied by col_scale before the one rounding (tamd_gemm_colscale)
Tensor k_gemm_colscale(const Tensor& x2, const Tensor& w, const OptTensor& bias, int64_t scale_cols, double col_scale) {
  Launch L({&x2, &w, p(bias)});
  TORCH_CHECK(x2.dim() == 2 && w.dim() == 2 && x2.stride(1) == 1 && w.stride(1) == 1 && x2.size(1) == w.size(1),
              "tamd: gemm_colscale takes row-major x [M, K] and w [N, K]");
  const int64_t m = x2.size(0), n = w.size(0), k = x2.size(1);
  Tensor y = at::empty({m, n}, x2.options());
  GemmTimerScope timer(2.0 * (double)m * (double)n * (double)k, 2.0 * ((double)m * k + (double)n * k + (double)m * n), L.stream);
  check(api().tamd_gemm_colscale(ptr(x2), ptr(w), mptr(y), ptr(bias), m, n, k, x2.stride(0), w.stride(0), n, 0, scale_cols,
                                 (float)col_scale, code_of(x2), L.stream),
        "tamd_gemm_colscale");
  return y;
}

// (act(pre), pre = round(x2 . w^T + bias)) for a layer whose backward needs the pre-activation: GEMM with the bias epilogue +
// the activation kernel.  (Both outputs from one GEMM epilogue lost badly on MI355X for erf-GELU at bert-base -- 495 us against
// 84 + 38, profiles/r04a_bert_kernel_stats.csv: one wave per SIMD has nothing to overlap ocml's erff with; the entry point
// went to profiles/r05_removed_variants.patch in round 5.)
std::tuple<Tensor, Tensor> linear_act_pre(const Tensor& x2, const Tensor& w, const Tensor& bias, int64_t act) {
  Tensor pre = gemm_plain(x2, w, false, false, bias, {}, TAMD_EPI_BIAS);
  return {k_bias_act_fwd(pre, {}, act), pre};
}

// Where to cut a weight-gradient product dW[M, N] = dy[K, M]^T . x[K, N] so that its 256 x 256 tile grid packs the 256 CUs:
// a grid of a few dispatch rounds whose last round is mostly empty (Llama-3-8B: q|k|v 384 tiles = 1.5 rounds, down_proj 896 =
// 3.5) used to be split along K as a whole -- every workgroup half the K range, fp32 partial tiles of the WHOLE output through
// HBM and a reduction pass (201 / 470 MB of partials: 1325 / 2820 us where 1.5 / 3.5 rounds of the unsplit kernel are 1070 / 2490).
// Instead: a main part whose tile count is a whole number of rounds (unsplit, no partials) and a remainder of at most half a
// round, which the split-K policy of the library fills (128 tiles -> 2 splits).  Returns axis 0 (cut the M rows at `at`),
// 1 (cut the N columns at `at`) or -1 (one launch: the grid already packs, or no such cut exists).
struct DwCut {
  int axis;
  int64_t at;
};
// `cus`: compute units of the device the product runs on (a dispatch round = one 256 x 256 tile per CU; MI355X: 256)
DwCut dw_balanced_cut(int64_t m, int64_t n, int64_t k, int64_t kCUs = 256) {
  constexpr int64_t kT = 256;
  if (kCUs < 2 || m % kT || n % kT || k % 64) return {-1, 0};
  const int64_t tm = m / kT, tn = n / kT, tiles = tm * tn;
  if (tiles <= kCUs || tiles % kCUs == 0 || tiles > 64 * kCUs) return {-1, 0};
  for (int64_t r = tm - 1; r >= 1; --r)  // the largest main part first
    if ((r * tn) % kCUs == 0 && (tm - r) * tn <= kCUs / 2) return {0, r * kT};
  for (int64_t c = tn - 1; c >= 1; --c)
    if ((c * tm) % kCUs == 0 && (tn - c) * tm <= kCUs / 2) return {1, c * kT};
  return {-1, 0};
}
// A/B switch of the cut (tools/gemm_dw_cut_ab.py through _native.set_dw_balance): not an environment read of the product
std::atomic<bool> g_dw_balance{true};
// compute units of the tensor's device (the CPU execution model of the test-suite stands for an MI355X)
int64_t cus_of(const Tensor& t) {
  if (!t.is_cuda()) return 256;
  hipDeviceProp_t prop;
  static std::mutex mu;
  static std::vector<int> cache;  // by device index
  std::lock_guard<std::mutex> lock(mu);
  const int dev = t.device().index();
  if ((int)cache.size() <= dev) cache.resize(dev + 1, 0);
  if (cache[dev] == 0) cache[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  return cache[dev];
}
// the cut pays for the 16-bit MFMA products it was measured on; fp32 operands keep one launch
DwCut dw_cut_for(const Tensor& dy, int64_t m, int64_t n, int64_t k) {
  if (!g_dw_balance.load(std::memory_order_relaxed) || dy.scalar_type() == at::kFloat) return DwCut{-1, 0};
  return dw_balanced_cut(m, n, k, cus_of(dy));
}

// dW[M, N] = dy[K, M]^T . x[K, N] (both operands k-major) as one launch or as the two of dw_balanced_cut; `out` (optional): the
// destination ([M, N] row-major view, e.g. a DDP bucket view)
Tensor gemm_dw_balanced(const Tensor& dy, const Tensor& x, const OptTensor& out_ = {}) {
  const int64_t k = dy.size(0), m = dy.size(1), n = x.size(1);
  const DwCut cut = dw_cut_for(dy, m, n, k);
  if (cut.axis < 0) return k_gemm(dy, x, true, true, {}, {}, TAMD_EPI_NONE, TAMD_ACT_NONE, out_, 0);
  Tensor out = out_ ? *out_ : at::empty({m, n}, dy.options());
  if (cut.axis == 0) {
    k_gemm(dy.narrow(1, 0, cut.at), x, true, true, {}, {}, TAMD_EPI_NONE, TAMD_ACT_NONE, out.narrow(0, 0, cut.at), 0);
    k_gemm(dy.narrow(1, cut.at, m - cut.at), x, true, true, {}, {}, TAMD_EPI_NONE, TAMD_ACT_NONE, out.narrow(0, cut.at, m - cut.at), 0);
  } else {
    k_gemm(dy, x.narrow(1, 0, cut.at), true, true, {}, {}, TAMD_EPI_NONE, TAMD_ACT_NONE, out.narrow(1, 0, cut.at), 0);
    k_gemm(dy, x.narrow(1, cut.at, n - cut.at), true, true, {}, {}, TAMD_EPI_NONE, TAMD_ACT_NONE, out.narrow(1, cut.at, n - cut.at), 0);
  }
  return out;
}

// dW[M, N] = dy[K, M]^T . x[K, N] with the M rows stored into `segs` (each [rows_i, N] contiguous: tamd_gemm_seg) -- the weight
// gradient of a fused q|k|v / gate|up projection written straight into the members' own gradient buffers
void gemm_dw_segments(const Tensor& dy, const Tensor& x, const std::vector<Tensor>& segs) {
  Launch L({&dy, &x, &segs[0]});
  TORCH_CHECK(dy.dim() == 2 && x.dim() == 2 && dy.stride(1) == 1 && x.stride(1) == 1 && dy.size(0) == x.size(0),
              "tamd: gemm_dw_segments takes dy [K, M] and x [K, N]");
  const int64_t k = dy.size(0), m = dy.size(1), n = x.size(1);
  void* ptrs[3];
  int64_t rows[3];
  int64_t total = 0;
  TORCH_CHECK(segs.size() >= 1 && segs.size() <= 3, "tamd: 1..3 output segments");
  for (size_t i = 0; i < segs.size(); ++i) {
    const Tensor& t = segs[i];
    TORCH_CHECK(t.dim() == 2 && t.size(1) == n && t.is_contiguous() && t.scalar_type() == dy.scalar_type() &&
                    t.device() == dy.device(),
                "tamd: an output segment must be a contiguous [rows, ", n, "] tensor of the operands' dtype and device");
    ptrs[i] = t.mutable_data_ptr();
    rows[i] = t.size(0);
    total += rows[i];
  }
  TORCH_CHECK(total == m, "tamd: the segments hold ", total, " rows, the product has ", m);
  bool tile_aligned = k % 64 == 0;
  for (size_t i = 0; i + 1 < segs.size(); ++i) tile_aligned = tile_aligned && rows[i] % 256 == 0;
  if (!tile_aligned) {  // segments that cut through a 256-row tile (small test models): one product, then the slices
    Tensor whole = gemm_plain(dy, x, true, true);
    int64_t off = 0;
    for (size_t i = 0; i < segs.size(); ++i) {
      const_cast<Tensor&>(segs[i]).copy_(whole.narrow(0, off, rows[i]));
      off += rows[i];
    }
    return;
  }
  const size_t ws_bytes = api().tamd_gemm_workspace_bytes(m, n, k, 3, TAMD_EPI_NONE);
  Tensor ws = ws_bytes ? at::empty({(int64_t)ws_bytes}, dy.options().dtype(at::kByte)) : Tensor();
  GemmTimerScope timer(2.0 * (double)m * (double)n * (double)k, 2.0 * ((double)m * k + (double)n * k + (double)m * n), L.stream);
  check(api().tamd_gemm_seg(ptr(dy), ptr(x), ptrs, rows, (int)segs.size(), n, k, dy.stride(0), x.stride(0), n, TAMD_EPI_NONE,
                            code_of(dy), mptr(ws), ws_bytes, L.stream),
        "tamd_gemm_seg");
}

bool half_type(const Tensor& t) { return t.scalar_type() == at::kBFloat16 || t.scalar_type() == at::kHalf; }

// dW_i[M_i, N_i] = dy_i[K_i, M_i]^T . x_i[K_i, N_i] for up to 4 pairs in ONE launch (tamd_gemm_group): the weight gradients of
// one BERT layer's four dense layers are 9 .. 36 output tiles each; together they fill the GPU with two K splits instead of
// 7 .. 16 (csrc/gemm.hip "grouped launch").  Pairs the grouped kernel does not take (K % 64) fall back to one product each.
std::vector<Tensor> gemm_dw_group(const std::vector<Tensor>& dys, const std::vector<Tensor>& xs) {
  TORCH_CHECK(dys.size() == xs.size() && !dys.empty() && dys.size() <= 4, "tamd: gemm_dw_group takes 1..4 (dy, x) pairs");
  Launch L({&dys[0], &xs[0]});
  std::vector<tamd_gemm_problem> pr(dys.size());
  std::vector<Tensor> outs;
  bool groupable = true;
  double flops = 0, bytes = 0;
  for (size_t i = 0; i < dys.size(); ++i) {
    const Tensor &dy = dys[i], &x = xs[i];
    TORCH_CHECK(dy.dim() == 2 && x.dim() == 2 && dy.stride(1) == 1 && x.stride(1) == 1 && dy.size(0) == x.size(0) &&
                    dy.scalar_type() == dys[0].scalar_type() && x.scalar_type() == dys[0].scalar_type() &&
                    dy.device() == dys[0].device() && x.device() == dys[0].device(),
                "tamd: gemm_dw_group takes dy_i [K, M] and x_i [K, N] of one dtype on one device");
    const int64_t k = dy.size(0), m = dy.size(1), n = x.size(1);
    groupable = groupable && k % 64 == 0 && half_type(dy);
    outs.push_back(at::empty({m, n}, dy.options()));
    pr[i] = tamd_gemm_problem{ptr(dy), ptr(x), mptr(outs.back()), m, n, k, dy.stride(0), x.stride(0), n};
    flops += 2.0 * (double)m * (double)n * (double)k;
    bytes += 2.0 * ((double)m * k + (double)n * k + (double)m * n);
  }
  if (!groupable) {
    for (size_t i = 0; i < dys.size(); ++i) outs[i] = gemm_plain(dys[i], xs[i], true, true);
    return outs;
  }
  const int flags = TAMD_GEMM_A_KM | TAMD_GEMM_B_KN;
  const size_t ws_bytes = api().tamd_gemm_group_workspace_bytes(pr.data(), (int)pr.size(), flags);
  Tensor ws = ws_bytes ? at::empty({(int64_t)ws_bytes}, dys[0].options().dtype(at::kByte)) : Tensor();
  GemmTimerScope timer(flops, bytes, L.stream);
  check(api().tamd_gemm_group(pr.data(), (int)pr.size(), flags, TAMD_EPI_NONE, code_of(dys[0]), mptr(ws), ws_bytes, L.stream),
        "tamd_gemm_group");
  return outs;
}



// shapes the fused gate|up GEMM + SiLU*up epilogue takes (csrc/gemm.hip tamd_gemm_swiglu)
bool gemm_swiglu_supported(const Tensor& x2, const Tensor& wgu) {
  const int64_t two_i = wgu.size(0), k = wgu.size(1);
  return half_type(x2) && wgu.scalar_type() == x2.scalar_type() && k % 64 == 0 && two_i % 16 == 0 && x2.stride(1) == 1 &&
         wgu.stride(1) == 1 && x2.stride(0) % 8 == 0 && wgu.stride(0) % 8 == 0 &&
         two_i * wgu.stride(0) * 2 < ((int64_t)1 << 31) &&
         // small grids: the plain GEMM (128 x 128 tile or split-K by the library's policy) + the swiglu kernel.  "Small" as
         // for a k-major product (flags 3): the forward-layout exception of the policy -- short K goes to the 128 x 128
         // tile instead of split-K -- is about WHICH plain kernel runs, not about whether the grid fills the GPU
         (x2.size(0) <= 16 ||  // (a decode step: the streaming kernel of csrc/gemv.hip behind the same entry point)
          api().tamd_gemm_workspace_bytes(x2.size(0), two_i, k, 3, TAMD_EPI_NONE) == 0);
}

// x2 [T, K], wgu [2I, K] = [gate_proj.weight ; up_proj.weight]  ->  (gu [T, 2I] or undefined, act [T, I])
std::tuple<Tensor, Tensor> k_gemm_swiglu(const Tensor& x2, const Tensor& wgu, bool need_gu) {
  Launch L({&x2, &wgu});
  const int64_t t = x2.size(0), k = x2.size(1), inter = wgu.size(0) / 2;
  Tensor gu = need_gu ? at::empty({t, 2 * inter}, x2.options()) : Tensor();
  Tensor act = at::empty({t, inter}, x2.options());
  GemmTimerScope timer(2.0 * (double)t * (2.0 * inter) * (double)k,
                       2.0 * ((double)t * k + 2.0 * inter * k + (double)t * 2 * inter * (need_gu ? 1.5 : 0.5)), L.stream);
  check(api().tamd_gemm_swiglu(ptr(x2), ptr(wgu), mptr(gu), mptr(act), t, inter, k, x2.stride(0), wgu.stride(0), 2 * inter,
                               inter, code_of(x2), L.stream),
        "tamd_gemm_swiglu");
  return {gu, act};
}

// shapes the down projection's dX GEMM + SiLU*up backward epilogue takes (csrc/gemm.hip tamd_gemm_swiglu_bwd): full 256 x 256
// grids only -- where the library's policy would split K or stream the product (few rows), the plain product + swiglu_bwd kernel
bool gemm_swiglu_bwd_supported(const Tensor& dy, const Tensor& wd, const Tensor& gu) {
  const int64_t k = wd.size(0), inter = wd.size(1), t = dy.size(0);
  return half_type(dy) && wd.scalar_type() == dy.scalar_type() && gu.scalar_type() == dy.scalar_type() && k % 64 == 0 &&
         inter % 8 == 0 && dy.stride(1) == 1 && wd.stride(1) == 1 && dy.stride(0) % 8 == 0 && wd.stride(0) % 8 == 0 &&
         gu.is_contiguous() && gu.size(0) == t && gu.size(1) == 2 * inter && 128 * 2 * inter * 2 < ((int64_t)1 << 31) && t > 16 &&
         api().tamd_gemm_workspace_bytes(t, inter, k, TAMD_GEMM_B_KN, TAMD_EPI_NONE) == 0;
}

// dy [T, K] (the MLP output's gradient), wd [K, I] = down_proj.weight, gu [T, 2I] = the forward's gate | up  ->  d_gu [T, 2I]
Tensor k_gemm_swiglu_bwd(const Tensor& dy, const Tensor& wd, const Tensor& gu) {
  Launch L({&dy, &wd, &gu});
  const int64_t t = dy.size(0), k = dy.size(1), inter = wd.size(1);
  Tensor dgu = at::empty({t, 2 * inter}, dy.options());
  GemmTimerScope timer(2.0 * (double)t * (double)inter * (double)k,
                       2.0 * ((double)t * k + (double)inter * k + (double)t * 2 * inter * 2.0), L.stream, true);
  check(api().tamd_gemm_swiglu_bwd(ptr(dy), ptr(wd), ptr(gu), mptr(dgu), t, inter, k, dy.stride(0), wd.stride(0), 2 * inter,
                                   2 * inter, code_of(dy), L.stream),
        "tamd_gemm_swiglu_bwd");
  return dgu;
}

// Which of the two bit-identical forms of the SiLU*up backward runs -- the dX GEMM's way out (tamd_gemm_swiglu_bwd) or GEMM +
// swiglu_bwd_kernel -- is MEASURED, once per shape, on the operands of the first call.  Round 2 had this way out and retired it:
// standalone and in a one-layer loop it was always ahead, inside the 32-layer model it ran 3.6 or 5.7 ms per launch depending on
// the box and on what else had allocated memory (profiles/r02_regression_note.md: same ISA, same arguments; never explained).
// The round-6 way out (buffer-addressed, four streams instead of five) was ahead on every box it has seen
// (profiles/r06o_swiglu_bwd_ab.jsonl: 3.13 against 3.47 ms, Llama-3-8B step 1241.7 -> 1231.1 ms), but a kernel with a history
// of a slow regime that only shows in the real model is chosen on the real model's tensors, not on faith: two HIP-event timings of
// each form on the stream of the first backward (~25 ms, once; a stream being captured into a HIP graph, or the CPU execution
// model, takes the fused form unmeasured).  TAMD_FUSE_SWIGLU_BWD=0 / 1 in the environment pins the choice.
struct SwigluBwdChoice {
  int64_t t, inter, k;
  int dtype;
  bool fused;
  double fused_ms, two_ms;  // (0: pinned or unmeasured)
};
std::mutex g_swiglu_bwd_mutex;
std::vector<SwigluBwdChoice> g_swiglu_bwd_choices;
bool swiglu_bwd_fused(const Tensor& dy, const Tensor& wd, const Tensor& gu) {
  static const int pinned = [] {
    const char* e = getenv("TAMD_FUSE_SWIGLU_BWD");
    return e == nullptr || *e == 0 ? -1 : (std::string(e) != "0" ? 1 : 0);
  }();
  if (pinned >= 0) return pinned == 1;
  if (api().emulated) return true;
  const int64_t t = dy.size(0), k = dy.size(1), inter = wd.size(1);
  const int dtype = code_of(dy);
  {
    std::lock_guard<std::mutex> lock(g_swiglu_bwd_mutex);
    for (const auto& c : g_swiglu_bwd_choices)
      if (c.t == t && c.inter == inter && c.k == k && c.dtype == dtype) return c.fused;
  }
  Launch L({&dy, &wd, &gu});
  hipStream_t s = (hipStream_t)L.stream;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return true;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess) return true;
  if (hipEventCreate(&e1) != hipSuccess) {
    (void)hipEventDestroy(e0);
    return true;
  }
  const bool log_was = g_gemm_log.exchange(false);  // (the trial launches are not the step's GEMMs)
  auto time_of = [&](auto&& run) {
    double best = 1e30;
    run();  // warm: code objects, workspace
    for (int i = 0; i < 2; ++i) {
      (void)hipEventRecord(e0, s);
      run();
      (void)hipEventRecord(e1, s);
      float ms = 0.f;
      if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1.0;
      best = ms < best ? ms : best;
    }
    return best;
  };
  double fused_ms = -1.0, two_ms = -1.0;
  try {  // (the trials allocate their outputs: out of memory in the middle of a first backward leaves the choice unmeasured)
    fused_ms = time_of([&] { (void)k_gemm_swiglu_bwd(dy, wd, gu); });
    two_ms = time_of([&] { (void)k_swiglu_bwd(gu, gemm_plain(dy, wd, false, true), false); });
  } catch (const c10::Error&) {
    g_gemm_log.store(log_was);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return true;
  }
  g_gemm_log.store(log_was);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  const bool fused = fused_ms < 0 || two_ms < 0 || fused_ms <= two_ms;
  std::lock_guard<std::mutex> lock(g_swiglu_bwd_mutex);
  g_swiglu_bwd_choices.push_back({t, inter, k, dtype, fused, fused_ms < 0 ? 0.0 : fused_ms, two_ms < 0 ? 0.0 : two_ms});
  return fused;
}

// ---- attention
void fill_attn_params(tamd_attn_params* ap, const Tensor& q, const Tensor& k, const Tensor& v, const Tensor& o,
                      const Tensor& lse, const Tensor& key_valid, double scale, bool causal, double dropout_p, int64_t seed,
                      const Tensor& q_start, bool q_prescaled = false, const uint64_t* seed_dev = nullptr) {
  ap->q = ptr(q);
  ap->k = ptr(k);
  ap->v = ptr(v);
  ap->o = mptr(o);
  ap->lse = (float*)mptr(lse);
  ap->key_valid = (const uint8_t*)ptr(key_valid);
  ap->batch = q.size(0);
  ap->seq_q = q.size(1);
  ap->heads_q = q.size(2);
  ap->head_dim = q.size(3);
  ap->seq_k = k.size(1);
  ap->heads_kv = k.size(2);
  for (const Tensor* t : {&q, &k, &v, &o})
    TORCH_CHECK(t->stride(3) == 1, "tamd: attention operands must have a contiguous head_dim");
  ap->q_stride_b = q.stride(0), ap->q_stride_s = q.stride(1), ap->q_stride_h = q.stride(2);
  ap->k_stride_b = k.stride(0), ap->k_stride_s = k.stride(1), ap->k_stride_h = k.stride(2);
  ap->v_stride_b = v.stride(0), ap->v_stride_s = v.stride(1), ap->v_stride_h = v.stride(2);
  ap->o_stride_b = o.stride(0), ap->o_stride_s = o.stride(1), ap->o_stride_h = o.stride(2);
  ap->scale = (float)scale;
  ap->causal = (int)causal;
  ap->dtype = code_of(q);
  ap->dropout_p = (float)dropout_p;
  ap->dropout_seed = (uint64_t)seed;
  ap->q_start = (const int32_t*)ptr(q_start);
  ap->q_prescaled = q_prescaled ? 1 : 0;
  ap->dropout_seed_dev = seed_dev;
}

// word `site` of a device-resident seed tensor (int64, ops.dropout_seed_tensor) or NULL: graph-replay-safe dropout
const uint64_t* seed_word(const OptTensor& seeds, int64_t site) {
  if (!seeds || !seeds->defined() || seeds->numel() == 0) return nullptr;
  TORCH_CHECK(seeds->scalar_type() == at::kLong && seeds->is_contiguous() && seeds->numel() > site,
              "tamd: a device seed tensor must be contiguous int64 with one word per dropout site");
  return reinterpret_cast<const uint64_t*>(seeds->const_data_ptr()) + site;
}

// tamd_attn_decode for q of at most kDecodeMaxRows rows over at least kDecodeMinKeys keys (TAMD_DECODE_KERNEL=0: always the
// training kernel -- the A/B switch of tools/decode_bench.py)
const bool kDecodeKernel = [] { const char* e = getenv("TAMD_DECODE_KERNEL"); return e == nullptr || std::string(e) != "0"; }();
constexpr int64_t kDecodeMaxRows = 16, kDecodeMinKeys = 128;

// [batch, seq_k] key-validity plane (1 = attend): the kernels index it as key_valid[b * seq_k + key]
Tensor checked_key_valid(const OptTensor& key_valid, const Tensor& q, const Tensor& k) {
  if (!key_valid) return Tensor();
  TORCH_CHECK(key_valid->dim() == 2 && key_valid->size(0) == q.size(0) && key_valid->size(1) == k.size(1),
              "tamd: key_valid must be [batch, seq_k] = [", q.size(0), ", ", k.size(1), "], got ", key_valid->sizes());
  // (a bool mask IS one byte per key, 0 / 1: reinterpreted, not converted -- the conversion was a launch per attention call,
  // one per layer and decode step: profiles/r04p_decode_kernel_stats.csv)
  if (key_valid->scalar_type() == at::kBool) return contig(*key_valid).view(at::kByte);
  return contig(key_valid->to(at::kByte));
}
// packed sequences: int32 [2, B, S] = (first token of each query's sequence, last token of each key's sequence)
Tensor checked_q_start(const OptTensor& q_start, const Tensor& q, bool causal) {
  if (!q_start) return Tensor();
  TORCH_CHECK(causal, "tamd: packed sequences (q_start) need causal attention");
  TORCH_CHECK(q_start->scalar_type() == at::kInt && q_start->dim() == 3 && q_start->size(0) == 2 &&
                  q_start->size(1) == q.size(0) && q_start->size(2) == q.size(1),
              "tamd: q_start must be int32 [2, batch, seq], got ", q_start->scalar_type(), " ", q_start->sizes());
  return contig(*q_start);
}

// q [B,Sq,Hq,D], k/v [B,Sk,Hkv,D] (strided views fine) -> o [B,Sq,Hq,D] contiguous, lse [B,Hq,Sq] fp32 or undefined
std::tuple<Tensor, Tensor> k_attn_fwd(const Tensor& q, const Tensor& k, const Tensor& v, double scale, bool causal,
                                      const OptTensor& key_valid_, bool need_lse, double dropout_p, int64_t seed,
                                      const OptTensor& q_start_, bool q_prescaled = false,
                                      const uint64_t* seed_dev = nullptr) {
  Launch L({&q, &k, &v, p(key_valid_)});
  Tensor o = at::empty({q.size(0), q.size(1), q.size(2), q.size(3)}, q.options());
  Tensor lse = need_lse ? f32_like(q, {q.size(0), q.size(2), q.size(1)}) : Tensor();
  Tensor key_valid = checked_key_valid(key_valid_, q, k);
  Tensor q_start = checked_q_start(q_start_, q, causal);
  tamd_attn_params ap;
  fill_attn_params(&ap, q, k, v, o, lse, key_valid, scale, causal, dropout_p, seed, q_start, q_prescaled, seed_dev);
  // decode shapes (a KV cache: a few query rows over a long key range) take the split-KV schedule
  if (kDecodeKernel && q.size(1) <= kDecodeMaxRows && k.size(1) >= kDecodeMinKeys && dropout_p == 0.0 && !q_start.defined()) {
    const size_t nbytes = api().tamd_attn_decode_workspace_bytes(&ap);
    Tensor ws = at::empty({(int64_t)nbytes}, q.options().dtype(at::kByte));
    check(api().tamd_attn_decode(&ap, mptr(ws), nbytes, L.stream), "tamd_attn_decode");
    return {o, lse};
  }
  check(api().tamd_attn_fwd(&ap, L.stream), "tamd_attn_fwd");
  return {o, lse};
}

// gradients written into dq / dk / dv (views with the strides of q / k / v) or freshly allocated.  rope = (cos, sin): q and k
// had been rotated before the attention, dq and dk leave through the transposed rotation.
std::tuple<Tensor, Tensor, Tensor> k_attn_bwd(const Tensor& q, const Tensor& k, const Tensor& v, const Tensor& o,
                                              const Tensor& lse, const Tensor& dout_, double scale, bool causal,
                                              const OptTensor& key_valid_, Tensor dq, Tensor dk, Tensor dv, double dropout_p,
                                              int64_t seed, const OptTensor& q_start_, const Tensor& rope_cos_,
                                              const Tensor& rope_sin_, bool q_prescaled = false,
                                              const uint64_t* seed_dev = nullptr) {
  Launch L({&q, &k, &v, &o, &lse, &dout_, p(key_valid_)});
  Tensor dout = dout_;
  if (dout.strides() != o.strides()) dout = o.is_contiguous() ? dout.contiguous() : dout.clone(at::MemoryFormat::Preserve);
  if (!dq.defined()) dq = at::empty_strided(q.sizes(), q.strides(), q.options());
  if (!dk.defined()) dk = at::empty_strided(k.sizes(), k.strides(), k.options());
  if (!dv.defined()) dv = at::empty_strided(v.sizes(), v.strides(), v.options());
  TORCH_CHECK(dq.strides() == q.strides() && dk.strides() == k.strides() && dv.strides() == v.strides(),
              "tamd: attention gradients must have the strides of q / k / v");
  Tensor key_valid = checked_key_valid(key_valid_, q, k);
  Tensor q_start = checked_q_start(q_start_, q, causal);
  auto dshape = lse.sizes().vec();
  dshape.insert(dshape.begin(), 2);
  Tensor delta = at::empty(dshape, lse.options());  // -delta | -lse*log2(e) (written by the dQ kernel)
  tamd_attn_bwd_params bp;
  fill_attn_params(&bp.fwd, q, k, v, o, lse, key_valid, scale, causal, dropout_p, seed, q_start, q_prescaled, seed_dev);
  bp.dout = ptr(dout);
  bp.dq = mptr(dq);
  bp.dk = mptr(dk);
  bp.dv = mptr(dv);
  bp.delta = (float*)mptr(delta);
  bp.rope_cos = nullptr;
  bp.rope_sin = nullptr;
  bp.rope_cos_batch = 0;
  Tensor cos, sin;
  if (rope_cos_.defined()) {
    cos = rope_cos_.scalar_type() == q.scalar_type() ? contig(rope_cos_) : rope_cos_.to(q.scalar_type()).contiguous();
    sin = rope_sin_.scalar_type() == q.scalar_type() ? contig(rope_sin_) : rope_sin_.to(q.scalar_type()).contiguous();
    bp.rope_cos = ptr(cos);
    bp.rope_sin = ptr(sin);
    bp.rope_cos_batch = cos.dim() == 3 ? cos.size(0) : 1;
  }
  check(api().tamd_attn_bwd(&bp, L.stream), "tamd_attn_bwd");
  return {dq, dk, dv};
}

bool attn_bwd_rope_supported(const Tensor& q, const Tensor& k, const Tensor& cos, int64_t head_dim) {
  return head_dim == 128 && q.size(1) == k.size(1) && cos.size(-1) == 128 && (cos.dim() == 2 || cos.dim() == 3);
}

// ================================================================================================ dispatcher ops
// ---- kernel-level ops (one per C-ABI entry point, no autograd)
std::tuple<Tensor, Tensor, Tensor> op_rmsnorm_fwd(const Tensor& x, const Tensor& w, double eps, const OptTensor& residual) {
  auto [y, h, rstd] = k_rmsnorm_fwd(x, w, eps, residual);
  return {y, residual ? h : nothing(x), rstd};  // (an op output must not alias an input)
}
std::tuple<Tensor, Tensor, Tensor, Tensor> op_layernorm_fwd(const Tensor& x, const Tensor& w, const OptTensor& b, double eps,
                                                            const OptTensor& residual) {
  auto [y, h, mean, rstd] = k_layernorm_fwd(x, w, b, eps, residual);
  return {y, residual ? h : nothing(x), mean, rstd};
}
std::tuple<Tensor, Tensor, Tensor, Tensor> op_layernorm_bwd(const Tensor& dy, const Tensor& h, const Tensor& w,
                                                            const Tensor& mean, const Tensor& rstd, const OptTensor& dres,
                                                            bool need_db, bool need_colsum) {
  auto [dx, dw, db, dc] = k_layernorm_bwd(dy, h, w, mean, rstd, dres, need_db, need_colsum);
  return {dx, dw, db.defined() ? db : nothing(w), dc.defined() ? dc : nothing(w)};
}
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> op_layernorm_dropout_bwd(const Tensor& dy, const Tensor& h, const Tensor& w,
                                                                            const Tensor& mean, const Tensor& rstd,
                                                                            double dropout_p, int64_t seed,
                                                                            const OptTensor& dres, bool need_db,
                                                                            bool need_colsum, const OptTensor& seed_dev) {
  auto [dx, dxd, dw, db, dc] = k_layernorm_dropout_bwd(dy, h, w, mean, rstd, dropout_p, seed, dres, need_db, need_colsum,
                                                       seed_word(seed_dev, 0));
  return {dx, dxd, dw, db.defined() ? db : nothing(w), dc.defined() ? dc : nothing(w)};
}
std::tuple<Tensor, Tensor, Tensor, Tensor> op_layernorm_dropout_fwd(const Tensor& x, const Tensor& w, const OptTensor& b,
                                                                    double eps, const Tensor& residual, double dropout_p,
                                                                    int64_t seed, const OptTensor& seed_dev) {
  return k_layernorm_dropout_fwd(x, w, b, eps, residual, dropout_p, seed, seed_word(seed_dev, 0));
}
std::tuple<Tensor, Tensor> op_bias_act_bwd(const Tensor& x, const OptTensor& bias, const Tensor& dy, int64_t act,
                                           bool need_colsum) {
  auto [dx, dc] = k_bias_act_bwd(x, bias, dy, act, need_colsum);
  return {dx, dc.defined() ? dc : nothing(x)};
}
void op_gemm_dw_segments(const Tensor& dy, const Tensor& x, at::TensorList segs) { gemm_dw_segments(dy, x, segs.vec()); }
std::vector<Tensor> op_gemm_dw_group(at::TensorList dy, at::TensorList x) { return gemm_dw_group(dy.vec(), x.vec()); }
Tensor op_gemm_colscale(const Tensor& x2, const Tensor& w, const OptTensor& bias, int64_t scale_cols, double col_scale) {
  return k_gemm_colscale(x2, w, bias, scale_cols, col_scale);
}
void op_rope_(Tensor& x2d, const Tensor& cos, const Tensor& sin, int64_t seq, int64_t nheads, int64_t head_dim, bool conj) {
  k_rope_(x2d, cos, sin, seq, nheads, head_dim, conj);
}
std::tuple<Tensor, Tensor, Tensor, Tensor> op_bert_embeddings_fwd(const Tensor& input_ids, const Tensor& token_type_ids,
                                                                  const Tensor& position_ids, const Tensor& word,
                                                                  const Tensor& typ, const Tensor& pos, const Tensor& ln_w,
                                                                  const Tensor& ln_b, double eps, bool keep_pre_ln) {
  auto [out, pre, mean, rstd] = k_bert_embeddings_fwd(input_ids, token_type_ids, position_ids, word, typ, pos, ln_w, ln_b, eps,
                                                      keep_pre_ln);
  return {out, pre.defined() ? pre : nothing(out), mean, rstd};
}
std::tuple<Tensor, Tensor> op_swiglu_bwd(const Tensor& gu, const Tensor& dact, bool want_act) {
  auto [dgu, act] = k_swiglu_bwd(gu, dact, want_act);
  return {dgu, act.defined() ? act : nothing(gu)};
}
Tensor op_cross_entropy_bwd(const Tensor& logits2d, const Tensor& labels, const Tensor& lse, const Tensor& gscale,
                            int64_t ignore_index) {
  return k_cross_entropy_bwd(logits2d, labels, lse, gscale, ignore_index, false);
}
void op_adamw_step_(Tensor& pp, const Tensor& g, Tensor& m, Tensor& v, double lr, double beta1, double beta2, double eps,
                    double weight_decay, int64_t step, double grad_scale) {
  k_adamw_step_(pp, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale);
}
Tensor op_gemm(const Tensor& a, const Tensor& b, bool a_km, bool b_kn, const OptTensor& bias, const OptTensor& residual,
               int64_t epilogue, int64_t act, int64_t sched) {
  // a plain weight-gradient product under the default dispatch (lm_head, the module-level linear layers): cut so that its
  // tile grid packs the CUs (gemm_dw_balanced); a schedule hint keeps it one launch
  if (a_km && b_kn && !bias && !residual && epilogue == TAMD_EPI_NONE && sched == 0 && a.dim() == 2 && b.dim() == 2)
    return gemm_dw_balanced(a, b);
  return k_gemm(a, b, a_km, b_kn, bias, residual, epilogue, act, {}, sched);
}
void op_gemm_out(Tensor& out, const Tensor& a, const Tensor& b, bool a_km, bool b_kn, const OptTensor& bias,
                 const OptTensor& residual, int64_t epilogue, int64_t act, int64_t sched) {
  k_gemm(a, b, a_km, b_kn, bias, residual, epilogue, act, out, sched);
}
std::tuple<Tensor, Tensor> op_gemm_swiglu(const Tensor& x2, const Tensor& wgu, bool need_gu) {
  auto [gu, act] = k_gemm_swiglu(x2, wgu, need_gu);
  return {gu.defined() ? gu : nothing(x2), act};
}
// An attention operand as the C ABI takes it (include/tamd.h): head_dim contiguous, rows following each other upwards at most
// 2^24 elements apart, strides multiples of 8 elements.  A caller's view that is something else -- a flipped tensor, or one
// expanded over rows, heads or batch (stride 0 with size > 1: `k.expand(b, s, H, d)` for MQA) -- is copied once; the layer ops
// never pass such views.  The expanded ones matter for the backward: it allocates dK / dV with the operand's strides, and an
// operand that overlaps itself would have every head (or batch entry) write the same gradient memory.
Tensor attn_operand(const Tensor& t) {
  const int64_t d = t.size(3);
  const bool rows_ok = t.size(1) <= 1 || (t.stride(1) >= d && t.stride(1) <= ((int64_t)1 << 24));
  bool ok = t.stride(3) == 1 && rows_ok && t.stride(0) % 8 == 0 && t.stride(1) % 8 == 0 && t.stride(2) % 8 == 0;
  for (int i = 0; i < 3; ++i) ok = ok && (t.size(i) <= 1 || t.stride(i) != 0);
  return ok ? t : t.contiguous();
}
std::tuple<Tensor, Tensor> op_attn_fwd(const Tensor& q_, const Tensor& k_, const Tensor& v_, double scale, bool causal,
                                       const OptTensor& key_valid, bool need_lse, double dropout_p, int64_t seed,
                                       const OptTensor& q_start, const OptTensor& seed_dev) {
  const Tensor q = attn_operand(q_), k = attn_operand(k_), v = attn_operand(v_);
  auto [o, lse] = k_attn_fwd(q, k, v, scale, causal, key_valid, need_lse, dropout_p, seed, q_start, false, seed_word(seed_dev, 0));
  return {o, lse.defined() ? lse : nothing(q)};
}
std::tuple<Tensor, Tensor, Tensor> op_attn_bwd(const Tensor& q_, const Tensor& k_, const Tensor& v_, const Tensor& o_,
                                               const Tensor& lse, const Tensor& dout, double scale, bool causal,
                                               const OptTensor& key_valid, double dropout_p, int64_t seed,
                                               const OptTensor& q_start, const OptTensor& rope_cos,
                                               const OptTensor& rope_sin, const OptTensor& seed_dev) {
  // (the forward op copied such operands too: the same values, and gradients are returned by shape, not by strides)
  const Tensor q = attn_operand(q_), k = attn_operand(k_), v = attn_operand(v_), o = attn_operand(o_);
  return k_attn_bwd(q, k, v, o, lse, dout, scale, causal, key_valid, Tensor(), Tensor(), Tensor(), dropout_p, seed, q_start,
                    rope_cos ? *rope_cos : Tensor(), rope_sin ? *rope_sin : Tensor(), false, seed_word(seed_dev, 0));
}
// the same, gradients written into caller-provided views (one fused d_qkv buffer)
void op_attn_bwd_out(Tensor& dq, Tensor& dk, Tensor& dv, const Tensor& q, const Tensor& k, const Tensor& v, const Tensor& o,
                     const Tensor& lse, const Tensor& dout, double scale, bool causal, const OptTensor& key_valid,
                     double dropout_p, int64_t seed, const OptTensor& q_start, const OptTensor& rope_cos,
                     const OptTensor& rope_sin, const OptTensor& seed_dev) {
  k_attn_bwd(q, k, v, o, lse, dout, scale, causal, key_valid, dq, dk, dv, dropout_p, seed, q_start,
             rope_cos ? *rope_cos : Tensor(), rope_sin ? *rope_sin : Tensor(), false, seed_word(seed_dev, 0));
}

// ---- forward implementations of the differentiable ops (backward formulas: Python, torch.library.register_autograd)
// Convention: forward ops return what the backward needs as extra outputs; a `train` flag tells a forward op whether
// anything will be differentiated (it runs below autograd and cannot see `requires_grad`).
std::tuple<Tensor, Tensor> op_rmsnorm(const Tensor& x, const Tensor& w, double eps) {
  auto [y, h, rstd] = k_rmsnorm_fwd(x, w, eps, {});
  return {y, rstd};
}
std::tuple<Tensor, Tensor, Tensor> op_add_rmsnorm(const Tensor& x, const Tensor& residual, const Tensor& w, double eps) {
  return k_rmsnorm_fwd(x, w, eps, residual);
}
std::tuple<Tensor, Tensor, Tensor> op_layernorm(const Tensor& x, const Tensor& w, const OptTensor& b, double eps) {
  auto [y, h, mean, rstd] = k_layernorm_fwd(x, w, b, eps, {});
  return {y, mean, rstd};
}
std::tuple<Tensor, Tensor, Tensor, Tensor> op_add_layernorm(const Tensor& x, const Tensor& residual, const Tensor& w,
                                                            const OptTensor& b, double eps) {
  return k_layernorm_fwd(x, w, b, eps, residual);
}
std::tuple<Tensor, Tensor, Tensor, Tensor> op_dropout_add_layernorm(const Tensor& x, const Tensor& residual, const Tensor& w,
                                                                    const OptTensor& b, double eps, double dropout_p,
                                                                    int64_t seed, const OptTensor& seed_dev) {
  return k_layernorm_dropout_fwd(x, w, b, eps, residual, dropout_p, seed, seed_word(seed_dev, 0));
}

// y = act(x W^T + b) [+ residual] on the MFMA GEMM -> (y, pre-activation or empty)
std::tuple<Tensor, Tensor> op_linear(const Tensor& x, const Tensor& w, const OptTensor& bias, const OptTensor& residual,
                                     int64_t act, bool train) {
  const int64_t k = x.size(-1);
  Tensor x2 = contig(x).view({-1, k});
  int64_t epi = TAMD_EPI_NONE;
  OptTensor r2;
  TORCH_CHECK(!(act != TAMD_ACT_NONE && (!bias || residual)), "tamd: activation epilogue needs a bias and no residual");
  if (residual) {
    epi = TAMD_EPI_RESIDUAL;
    r2 = contig(*residual).view({-1, w.size(0)});
  } else if (bias && act != TAMD_ACT_NONE) {
    epi = TAMD_EPI_BIAS_ACT;
  } else if (bias) {
    epi = TAMD_EPI_BIAS;
  }
  Tensor pre, y;
  if (epi == TAMD_EPI_BIAS_ACT && train) {  // the pre-activation is kept for the backward
    std::tie(y, pre) = linear_act_pre(x2, w, *bias, act);
  } else {
    pre = nothing(x);
    y = gemm_plain(x2, w, false, false, bias, r2, epi, act);
  }
  auto shape = x.sizes().vec();
  shape.back() = w.size(0);
  return {y.view(shape), pre};
}

// y = x . Wf^T (+ bf), Wf = row-concatenation of the member weights (fused QKV / gate|up, fused_params.py)
Tensor op_fused_linear(const Tensor& x, const Tensor& wf, const OptTensor& bf, at::TensorList /*members*/) {
  Tensor x2 = contig(x).view({-1, x.size(-1)});
  Tensor y = gemm_plain(x2, wf, false, false, bf, {}, bf ? TAMD_EPI_BIAS : TAMD_EPI_NONE);
  auto shape = x.sizes().vec();
  shape.back() = wf.size(0);
  return y.view(shape);
}

// y = x @ W + b with W stored [in, out] (GPT-2 Conv1D, pytorch_utils.py:117-121): the k-major B operand
Tensor op_conv1d(const Tensor& x, const Tensor& w, const OptTensor& b) {
  Tensor x2 = contig(x).view({-1, x.size(-1)});
  Tensor y = gemm_plain(x2, w, false, true, b, {}, b ? TAMD_EPI_BIAS : TAMD_EPI_NONE);
  auto shape = x.sizes().vec();
  shape.back() = w.size(1);
  return y.view(shape);
}

// rotary embedding on the first `nheads` heads of a [B, S, row] projection output, out of place
Tensor op_rope(const Tensor& x, const Tensor& cos, const Tensor& sin, int64_t nheads, int64_t head_dim, bool conj) {
  Tensor y = x.clone(at::MemoryFormat::Contiguous);
  k_rope_(y.view({x.size(0) * x.size(1), x.size(2)}), cos, sin, x.size(1), nheads, head_dim, conj);
  return y;
}

std::tuple<Tensor, Tensor> op_attention(const Tensor& q_, const Tensor& k_, const Tensor& v_, const OptTensor& key_valid,
                                        double scale, bool causal, double dropout_p, int64_t seed, const OptTensor& q_start,
                                        bool train, const OptTensor& seed_dev) {
  const Tensor q = attn_operand(q_), k = attn_operand(k_), v = attn_operand(v_);  // (the backward goes through op_attn_bwd: same copies)
  auto [o, lse] = k_attn_fwd(q, k, v, scale, causal, key_valid, train, dropout_p, seed, q_start, false, seed_word(seed_dev, 0));
  return {o, lse.defined() ? lse : nothing(q)};
}

Tensor op_swiglu(const Tensor& gu) {
  auto shape = gu.sizes().vec();
  Tensor act = k_swiglu_fwd(contig(gu).view({-1, shape.back()}));
  shape.back() /= 2;
  return act.view(shape);
}

Tensor op_embedding(const Tensor& ids, const Tensor& table, int64_t /*padding_idx*/) { return k_embedding_fwd(ids, table); }

std::tuple<Tensor, Tensor, Tensor, Tensor> op_bert_embeddings(const Tensor& input_ids, const Tensor& token_type_ids,
                                                              const Tensor& position_ids, const Tensor& word, const Tensor& typ,
                                                              const Tensor& pos, const Tensor& ln_w, const Tensor& ln_b,
                                                              double eps, int64_t /*padding_idx*/, bool train) {
  return op_bert_embeddings_fwd(input_ids, token_type_ids, position_ids, word, typ, pos, ln_w, ln_b, eps, train);
}

// fixed_cross_entropy on `logits.float()` (loss/loss_utils.py:32-46) without materialising fp32 logits -> (SUM of the
// per-token losses, lse)
std::tuple<Tensor, Tensor> op_cross_entropy_sum(const Tensor& logits2d, const Tensor& labels, int64_t ignore_index) {
  auto [lse, row_loss] = k_cross_entropy_fwd(logits2d, labels, ignore_index);
  return {row_loss.sum(), lse};
}

// lm_head + causal-LM loss chunk by chunk, without the [tokens, vocab] logits: the gradients are
// produced in the forward and only multiplied by the upstream scalar in the backward -> (loss, dh or empty, dw or empty)
std::tuple<Tensor, Tensor, Tensor> op_linear_cross_entropy(const Tensor& h2d, const Tensor& w, const Tensor& labels,
                                                           const Tensor& normaliser, int64_t ignore_index, int64_t chunk,
                                                           bool need_dh, bool need_dw) {
  const int64_t t = h2d.size(0);
  Tensor gs = (1.0 / normaliser.to(at::kFloat)).reshape({1}).contiguous();
  Tensor loss = at::zeros({}, h2d.options().dtype(at::kFloat));
  Tensor dh = need_dh ? at::empty_like(h2d) : nothing(h2d);
  Tensor dw = need_dw ? at::empty_like(w) : nothing(w);
  bool first = true;
  for (int64_t c0 = 0; c0 < t; c0 += chunk) {
    const int64_t c1 = std::min(c0 + chunk, t);
    Tensor hc = h2d.narrow(0, c0, c1 - c0), lc = labels.narrow(0, c0, c1 - c0);
    Tensor logits = gemm_plain(hc, w);
    auto [lse, row_loss] = k_cross_entropy_fwd(logits, lc, ignore_index);
    loss = loss + row_loss.sum();
    if (need_dh || need_dw) {
      Tensor dlog = k_cross_entropy_bwd(logits, lc, lse, gs, ignore_index, false);
      logits = Tensor();
      if (need_dh) gemm_plain(dlog, w, false, true, {}, {}, TAMD_EPI_NONE, TAMD_ACT_NONE, dh.narrow(0, c0, c1 - c0));
      if (need_dw) gemm_plain(dlog, hc, true, true, {}, {}, first ? TAMD_EPI_NONE : TAMD_EPI_ACCUM, TAMD_ACT_NONE, dw);
    }
    first = false;
  }
  return {loss * gs[0], dh, dw};
}

// Vocabulary projection whose width is not a multiple of 8 (+ optional token-level cross-entropy): BERT's MLM head
// (models/bert/modeling_bert.py:483-496, 970-975), through zero-padded weight / bias rows (fused_params.PaddedRows)
//   -> (loss_sum fp32 scalar, logits [M, V] = the row-strided view of an [M, Vp] buffer, lse [M] fp32 or empty)
std::tuple<Tensor, Tensor, Tensor> op_padded_vocab_head(const Tensor& h2d, const Tensor& w_pad, const OptTensor& b_pad,
                                                        const Tensor& w, const OptTensor& /*b*/, const OptTensor& labels,
                                                        int64_t ignore_index, bool /*train*/) {
  const int64_t n = w.size(0);
  Tensor logits_pad = gemm_plain(h2d, w_pad, false, false, b_pad, {}, b_pad ? TAMD_EPI_BIAS : TAMD_EPI_NONE);
  Tensor logits = n != w_pad.size(0) ? logits_pad.narrow(1, 0, n) : logits_pad;
  if (!labels) return {at::zeros({}, h2d.options().dtype(at::kFloat)), logits, f32_like(h2d, {0})};
  auto [lse, row_loss] = k_cross_entropy_fwd(logits, *labels, ignore_index);
  return {row_loss.sum(), logits, lse};
}

// its backward: the dX / dW / db products on the K-padded gradient buffer the loss kernel writes (padding columns zero)
//   -> (dh or empty, dw [V, K] or empty, db [V] or empty)
std::tuple<Tensor, Tensor, Tensor> op_padded_vocab_head_bwd(const OptTensor& g_loss, const OptTensor& g_logits,
                                                            const Tensor& h2d, const Tensor& w_pad, const OptTensor& labels,
                                                            const Tensor& logits, const Tensor& lse, int64_t ignore_index,
                                                            bool need_dh, bool need_dw, bool need_db) {
  const int64_t m = logits.size(0), n = logits.size(1), vp = w_pad.size(0);
  Tensor dlog;
  if (g_loss && labels) {
    Tensor gs = g_loss->detach().to(at::kFloat).reshape({1}).contiguous();
    dlog = k_cross_entropy_bwd(logits, *labels, lse, gs, ignore_index, true);  // [M, Vp], padding zero
  }
  if (g_logits) {  // the scores themselves were differentiated (a custom loss on `logits`): generic path
    if (!dlog.defined()) dlog = at::zeros({m, vp}, logits.options());
    dlog.narrow(1, 0, n).add_(*g_logits);
  }
  Tensor dh = need_dh ? gemm_plain(dlog, w_pad, false, true) : nothing(h2d);                 // dX = dY . W   (K = Vp)
  Tensor dw = need_dw ? gemm_plain(dlog, h2d, true, true).narrow(0, 0, n) : nothing(h2d);    // dW = dY^T . X (rows V.. dropped)
  Tensor db = need_db ? k_colsum(dlog).narrow(0, 0, n) : nothing(h2d);
  return {dh, dw, db};
}

// ================================================================================================ LlamaDecoderLayer
// tamd::llama_layer / tamd::llama_layer_bwd   LlamaDecoderLayer.forward, models/llama/modeling_llama.py:295-324, as ONE op
// (one autograd node).  Knobs (A/B switches for measurements, read once from the environment):
//   TAMD_FUSE_ROPE_BWD=0   the transposed rotary as its own kernel instead of inside the attention backward's way out
//   TAMD_SAVE_SWIGLU_ACT=0 re-materialise silu(gate)*up in the backward instead of keeping it (-30 GB at Llama-3-8B 8 x 4096,
//                          +0.94 GB of HBM writes per layer)
bool env_flag(const char* name, bool dflt) {
  const char* e = getenv(name);
  return e == nullptr ? dflt : (std::string(e) != "0");
}
const bool kFuseRopeBwd = env_flag("TAMD_FUSE_ROPE_BWD", true);
const bool kSaveSwigluAct = env_flag("TAMD_SAVE_SWIGLU_ACT", true);
//   TAMD_FUSE_SWIGLU_BWD   0: the SiLU*up backward as its own kernel behind the down projection's dX GEMM; 1: as that GEMM's way
//                          out; unset: MEASURED on first use per shape (swiglu_bwd_fused below)
// the rotary kernel hands the attention kernels queries that already carry scale*log2(e), applied before its one rounding
// (include/tamd.h q_prescaled; 0: the attention kernels scale and re-round their operand themselves)
const bool kRopePrescale = env_flag("TAMD_ROPE_PRESCALE", true);
constexpr double kLog2e = 1.44269504088896340736;
// BertLayer: the q|k|v GEMM's epilogue delivers the pre-scaled queries (TAMD_BERT_PRESCALE=0: the kernels scale and re-round)
const bool kBertPrescale = env_flag("TAMD_BERT_PRESCALE", true);
// the layer's four weight gradients as ONE grouped launch at the end of its backward (gemm_dw_group) instead of four split-K
// products where they arise: A/B switch
const bool kBertGroupDw = env_flag("TAMD_BERT_GROUP_DW", true);

struct Qkv {
  Tensor q, k, v;
};
Qkv split_qkv(const Tensor& qkv, int64_t b, int64_t s, int64_t hq, int64_t hkv, int64_t d) {
  return {qkv.narrow(1, 0, hq * d).view({b, s, hq, d}), qkv.narrow(1, hq * d, hkv * d).view({b, s, hkv, d}),
          qkv.narrow(1, (hq + hkv) * d, hkv * d).view({b, s, hkv, d})};
}


using LlamaLayerOut = std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>;
// forward : rmsnorm -> QKV GEMM, rotary (kernel, or the GEMM's epilogue) -> attention -> o_proj GEMM(+residual)
//           -> rmsnorm -> gate|up GEMM with the SwiGLU epilogue -> down GEMM(+residual)
LlamaLayerOut op_llama_layer(const Tensor& h_in, const Tensor& cos, const Tensor& sin, const OptTensor& key_valid,
                             const OptTensor& q_start, const Tensor& w_ln1, const Tensor& wqkv, const Tensor& /*wq*/,
                             const Tensor& /*wk*/, const Tensor& /*wv*/, const Tensor& wo, const Tensor& w_ln2,
                             const Tensor& wgu, const Tensor& /*wg*/, const Tensor& /*wu*/, const Tensor& wd, double eps,
                             int64_t hq, int64_t hkv, int64_t d, double scale, bool causal, bool train) {
  const int64_t b = h_in.size(0), s = h_in.size(1), hd = h_in.size(2), t = b * s;
  Tensor x = contig(h_in).view({t, hd});
  auto [xn, h_unused, rstd1] = k_rmsnorm_fwd(x, w_ln1, eps, {});
  Tensor qkv;
  // the rotary kernel works in place on the fresh q|k|v buffer.  (The same embedding as an epilogue of the q|k|v GEMM measured
  // level with GEMM + kernel -- 11.62 vs 11.62 / 11.54 ms per layer forward, profiles/r03a_rope_fwd_ab.txt: a way out has
  // nothing to overlap its arithmetic with -- and went to profiles/r05_removed_variants.patch.)
  const bool q_prescaled = kRopePrescale;  // (the backward reads the same constant)
  qkv = gemm_plain(xn, wqkv);
  k_rope_(qkv, cos, sin, s, hq + hkv, d, false, q_prescaled ? hq : 0, q_prescaled ? scale * kLog2e : 1.0);
  Qkv p3 = split_qkv(qkv, b, s, hq, hkv, d);
  auto [o, lse] = k_attn_fwd(p3.q, p3.k, p3.v, scale, causal, key_valid, train, 0.0, 0, q_start, q_prescaled);
  Tensor h_mid = gemm_plain(o.view({t, hq * d}), wo, false, false, {}, x, TAMD_EPI_RESIDUAL);
  auto [xn2, h_unused2, rstd2] = k_rmsnorm_fwd(h_mid, w_ln2, eps, {});
  Tensor gu, act;
  if (gemm_swiglu_supported(xn2, wgu)) {  // SiLU*up in the gate|up GEMM epilogue; gate|up itself only if saved
    std::tie(gu, act) = k_gemm_swiglu(xn2, wgu, train);
  } else {
    gu = gemm_plain(xn2, wgu);
    act = k_swiglu_fwd(gu);
  }
  Tensor h_out = gemm_plain(act, wd, false, false, {}, h_mid, TAMD_EPI_RESIDUAL).view({b, s, hd});
  if (!train) {
    auto e = [&] { return nothing(h_in); };
    return {h_out, e(), e(), e(), e(), e(), e(), e(), e(), e(), e()};
  }
  return {h_out, xn, qkv, o, lse, h_mid, xn2, gu, rstd1, rstd2, kSaveSwigluAct ? act : nothing(h_in)};
}

// backward: the derivatives of SURVEY.md section 8a in reverse; every weight gradient is a k-major GEMM on the saved
// activations, the SiLU*up product comes from the forward (`act_saved`) or is re-materialised (an empty `act_saved`), and the
// residual-stream gradient is folded into the RMSNorm backward kernels (`dres`).  Nothing saved by the forward is written.
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> op_llama_layer_bwd(
    const Tensor& d_hout, const Tensor& h_in, const Tensor& cos, const Tensor& sin, const OptTensor& key_valid,
    const OptTensor& q_start, const Tensor& w_ln1, const Tensor& wqkv, const Tensor& wo, const Tensor& w_ln2,
    const Tensor& wgu, const Tensor& wd, const Tensor& rstd1, const Tensor& xn, const Tensor& qkv, const Tensor& o,
    const Tensor& lse, const Tensor& h_mid, const Tensor& rstd2, const Tensor& xn2, const Tensor& gu, const Tensor& act_saved,
    int64_t hq, int64_t hkv, int64_t d, double scale, bool causal, const OptTensor& dst_q, const OptTensor& dst_k,
    const OptTensor& dst_v, const OptTensor& dst_o, const OptTensor& dst_g, const OptTensor& dst_u, const OptTensor& dst_d) {
  // dst_*: the weight gradients' destinations (transformers_amd/ddp.py: DDP's bucket views) -- all seven or none.  With them
  // the dW GEMMs store there and the corresponding outputs come back empty.
  const bool to_dst = dst_q && dst_k && dst_v && dst_o && dst_g && dst_u && dst_d;
  // (The four weight gradients as ONE grouped launch -- 3328 tiles = exactly 13 rounds, no split-K -- measured 0.35 % slower than
  // four launches, profiles/r04z_bench_ab.jsonl; removed in round 5, profiles/r05_removed_variants.patch.)
  const int64_t b = h_in.size(0), s = h_in.size(1), hd = h_in.size(2), t = b * s;
  Tensor x = contig(h_in).view({t, hd});
  Tensor dh = contig(d_hout).view({t, hd});
  // ---- MLP
  Tensor d_gu, act;
  if (act_saved.numel() && gemm_swiglu_bwd_supported(dh, wd, gu) && swiglu_bwd_fused(dh, wd, gu)) {
    // the SiLU*up backward as the way out of the dX GEMM: d_act [T, I] is never written (same bits as the two kernels below)
    d_gu = k_gemm_swiglu_bwd(dh, wd, gu);
    act = act_saved;
  } else {
    Tensor d_act = gemm_plain(dh, wd, false, true);  // [T, I]
    if (act_saved.numel()) {
      d_gu = std::get<0>(k_swiglu_bwd(gu, d_act, false));
      act = act_saved;
    } else {
      std::tie(d_gu, act) = k_swiglu_bwd(gu, d_act, true);
    }
  }
  // (the weight gradients whose tile grids end in a mostly empty dispatch round -- down_proj 3.5 rounds, q|k|v 1.5 -- go out as a
  // whole-rounds part + a split-K remainder: gemm_dw_balanced)
  Tensor dwd = to_dst ? (gemm_dw_balanced(dh, act, *dst_d), nothing(h_in)) : gemm_dw_balanced(dh, act);  // [hd, I]
  act = Tensor();
  Tensor d_xn2 = gemm_plain(d_gu, wgu, false, true);  // [T, hd]
  Tensor dwgu;                                        // [2I, hd]
  if (to_dst) {
    gemm_dw_segments(d_gu, xn2, {*dst_g, *dst_u});
    dwgu = nothing(h_in);
  } else {
    dwgu = gemm_plain(d_gu, xn2, true, true);
  }
  d_gu = Tensor();
  auto [d_hmid, dw_ln2] = k_rmsnorm_bwd(d_xn2, h_mid, w_ln2, rstd2, dh);
  d_xn2 = Tensor();
  // ---- attention
  Tensor d_o = gemm_plain(d_hmid, wo, false, true);  // [T, Hq*D]
  Tensor dwo = to_dst ? (gemm_plain(d_hmid, o.view({t, hq * d}), true, true, {}, {}, TAMD_EPI_NONE, TAMD_ACT_NONE, *dst_o),
                         nothing(h_in))
                      : gemm_plain(d_hmid, o.view({t, hq * d}), true, true);
  Tensor d_qkv = at::empty_like(qkv);
  Qkv f = split_qkv(qkv, b, s, hq, hkv, d), g = split_qkv(d_qkv, b, s, hq, hkv, d);
  const bool fused_rope = kFuseRopeBwd && attn_bwd_rope_supported(f.q, f.k, cos, d);  // the transposed rotary inside the kernels
  k_attn_bwd(f.q, f.k, f.v, o, lse, d_o.view({b, s, hq, d}), scale, causal, key_valid, g.q, g.k, g.v, 0.0, 0, q_start,
             fused_rope ? cos : Tensor(), fused_rope ? sin : Tensor(), kRopePrescale);
  d_o = Tensor();
  if (!fused_rope) k_rope_(d_qkv, cos, sin, s, hq + hkv, d, true);
  Tensor d_xn = gemm_plain(d_qkv, wqkv, false, true);
  Tensor dwqkv;  // [(Hq+2Hkv)D, hd]
  if (to_dst) {
    const int64_t nq = hq * d, nkv = 2 * hkv * d;
    const DwCut cut = dw_cut_for(d_qkv, nq + nkv, hd, t);
    if (cut.axis == 0 && cut.at == nq) {  // the cut falls on the q | k|v boundary: q as one product, k|v as a segmented one
      k_gemm(d_qkv.narrow(1, 0, nq), xn, true, true, {}, {}, TAMD_EPI_NONE, TAMD_ACT_NONE, *dst_q, 0);
      gemm_dw_segments(d_qkv.narrow(1, nq, nkv), xn, {*dst_k, *dst_v});
    } else {
      gemm_dw_segments(d_qkv, xn, {*dst_q, *dst_k, *dst_v});
    }
    dwqkv = nothing(h_in);
  } else {
    dwqkv = gemm_dw_balanced(d_qkv, xn);
  }
  auto [d_hin, dw_ln1] = k_rmsnorm_bwd(d_xn, x, w_ln1, rstd1, d_hmid);
  return {d_hin.view({b, s, hd}), dw_ln1, dwqkv, dwo, dw_ln2, dwgu, dwd};
}

// ================================================================================================ BertLayer
// tamd::bert_layer / tamd::bert_layer_bwd   BertLayer.forward (encoder layer), models/bert/modeling_bert.py:164-203, 289-293,
// 334-351, 374-416.  Post-LN blocks with biases:
//     y1 = dropout(attn_out . Wo^T + bo) + x ;   h1 = LayerNorm1(y1)
//     y2 = dropout(act(h1 . Wi^T + bi) . Wo2^T + bo2) + h1 ;   out = LayerNorm2(y2)
// Without hidden dropout the residual adds ride in the dense GEMMs' epilogues; with it the add joins the dropout + LayerNorm
// kernel.  In the backward the two places where a tensor feeds both a projection and a residual (x, h1) get their gradient
// sum from the residual epilogue of the dX GEMM -- as separate ops autograd adds them (49 `at::add` launches per bert-base
// step) and rebuilds d_qkv from three slices (36 fills + copies).
using BertLayerOut =
    std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor>;
BertLayerOut op_bert_layer(const Tensor& h_in, const OptTensor& key_valid, const Tensor& wqkv, const Tensor& bqkv,
                           const Tensor& /*wq*/, const Tensor& /*wk*/, const Tensor& /*wv*/, const Tensor& /*bq*/,
                           const Tensor& /*bk*/, const Tensor& /*bv*/, const Tensor& wo, const Tensor& bo, const Tensor& ln1_w,
                           const Tensor& ln1_b, const Tensor& wi, const Tensor& bi, const Tensor& wo2, const Tensor& bo2,
                           const Tensor& ln2_w, const Tensor& ln2_b, double eps, int64_t heads, int64_t d, double scale,
                           int64_t act, double p_attn, double p_hidden, int64_t seed_attn, int64_t seed1, int64_t seed2,
                           bool train, const OptTensor& seeds_dev) {  // seeds_dev: int64[3] = (attention, hidden 1, hidden 2)
  const int64_t b = h_in.size(0), s = h_in.size(1), hd = h_in.size(2), t = b * s;
  Tensor x = contig(h_in).view({t, hd});
  // the query columns leave the q|k|v GEMM carrying scale*log2(e) before their one rounding (tamd_gemm_colscale): the
  // attention kernels then scale and re-round nothing (q_prescaled) -- the reference's q has one rounding too
  Tensor qkv = kBertPrescale ? k_gemm_colscale(x, wqkv, bqkv, heads * d, scale * kLog2e)
                             : gemm_plain(x, wqkv, false, false, bqkv, {}, TAMD_EPI_BIAS);
  Qkv p3 = split_qkv(qkv, b, s, heads, heads, d);
  auto [o, lse] = k_attn_fwd(p3.q, p3.k, p3.v, scale, false, key_valid, train, p_attn, seed_attn, {}, kBertPrescale,
                             seed_word(seeds_dev, 0));
  auto dense_add_ln = [&](const Tensor& inp, const Tensor& w, const Tensor& bias, const Tensor& res, const Tensor& ln_w,
                          const Tensor& ln_b, int64_t seed, int site) -> std::tuple<Tensor, Tensor, Tensor, Tensor> {
    if (p_hidden > 0.0) {
      Tensor a = gemm_plain(inp, w, false, false, bias, {}, TAMD_EPI_BIAS);
      return k_layernorm_dropout_fwd(a, ln_w, ln_b, eps, res, p_hidden, seed, seed_word(seeds_dev, site));  // (y, pre-norm sum, mean, rstd)
    }
    Tensor y = gemm_plain(inp, w, false, false, bias, res, TAMD_EPI_RESIDUAL);
    auto [out, h_unused, mean, rstd] = k_layernorm_fwd(y, ln_w, ln_b, eps, {});
    return {out, y, mean, rstd};
  };
  auto [h1, y1, mean1, rstd1] = dense_add_ln(o.view({t, hd}), wo, bo, x, ln1_w, ln1_b, seed1, 1);
  Tensor pre, inter;
  if (train) {  // the pre-activation is what the activation's backward needs
    std::tie(inter, pre) = linear_act_pre(h1, wi, bi, act);
  } else {
    pre = nothing(h_in);
    inter = gemm_plain(h1, wi, false, false, bi, {}, TAMD_EPI_BIAS_ACT, act);
  }
  auto [out, y2, mean2, rstd2] = dense_add_ln(inter, wo2, bo2, h1, ln2_w, ln2_b, seed2, 2);
  Tensor out3 = out.view({b, s, hd});
  if (!train) {
    auto e = [&] { return nothing(h_in); };
    return {out3, e(), e(), e(), e(), e(), e(), e(), e(), e(), e(), e(), e()};
  }
  return {out3, qkv, o, lse, y1, mean1, rstd1, h1, pre, inter, y2, mean2, rstd2};
}

BertLayerOut op_bert_layer_bwd(const Tensor& d_out, const Tensor& h_in, const OptTensor& key_valid, const Tensor& wqkv,
                               const Tensor& wo, const Tensor& ln1_w, const Tensor& wi, const Tensor& wo2, const Tensor& ln2_w,
                               const Tensor& qkv, const Tensor& o, const Tensor& lse, const Tensor& y1, const Tensor& mean1,
                               const Tensor& rstd1, const Tensor& h1, const Tensor& pre, const Tensor& inter, const Tensor& y2,
                               const Tensor& mean2, const Tensor& rstd2, int64_t heads, int64_t d, double scale, int64_t act,
                               double p_attn, double p_hidden, int64_t seed_attn, int64_t seed1, int64_t seed2,
                               const OptTensor& seeds_dev) {
  const int64_t b = h_in.size(0), s = h_in.size(1), hd = h_in.size(2), t = b * s;
  Tensor x = contig(h_in).view({t, hd});
  Tensor dy = contig(d_out).view({t, hd});
  // -> (gradient of the residual input, of the dense output, dw, db, column sums of the dense output's gradient = the
  //     dense layer's bias gradient, accumulated by the same kernel)
  auto ln_bwd = [&](const Tensor& g, const Tensor& y, const Tensor& ln_w, const Tensor& mean, const Tensor& rstd,
                    int64_t seed, int site) -> std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> {
    if (p_hidden > 0.0)
      return k_layernorm_dropout_bwd(g, y, ln_w, mean, rstd, p_hidden, seed, {}, true, true, seed_word(seeds_dev, site));
    auto [dx, dw, db, dc] = k_layernorm_bwd(g, y, ln_w, mean, rstd, {}, true, true);
    return {dx, dx, dw, db, dc};
  };
  // ---- BertOutput / BertIntermediate
  auto [d_h1_res, d_b, dw_ln2, db_ln2, dbo2] = ln_bwd(dy, y2, ln2_w, mean2, rstd2, seed2, 2);
  // (the four weight gradients: together, at the end -- gemm_dw_group; TAMD_BERT_GROUP_DW=0: one product each, where they arise)
  Tensor dwo2 = kBertGroupDw ? Tensor() : gemm_plain(d_b, inter, true, true);  // [hd, I]
  Tensor d_inter = gemm_plain(d_b, wo2, false, true);  // [T, I]
  auto [d_pre, dbi] = k_bias_act_bwd(pre, {}, d_inter, act, true);  // (+ its column sums: the intermediate bias gradient)
  d_inter = Tensor();
  Tensor dwi = kBertGroupDw ? Tensor() : gemm_plain(d_pre, h1, true, true);  // [I, hd]
  Tensor d_h1 = gemm_plain(d_pre, wi, false, true, {}, d_h1_res, TAMD_EPI_RESIDUAL);  // + the residual path's gradient
  if (!kBertGroupDw) d_pre = Tensor();
  // ---- BertSelfOutput / BertSelfAttention
  auto [d_x_res, d_a, dw_ln1, db_ln1, dbo] = ln_bwd(d_h1, y1, ln1_w, mean1, rstd1, seed1, 1);
  Tensor dwo = kBertGroupDw ? Tensor() : gemm_plain(d_a, o.view({t, hd}), true, true);
  Tensor d_o = gemm_plain(d_a, wo, false, true);
  Tensor d_qkv = at::empty_like(qkv);
  Qkv f = split_qkv(qkv, b, s, heads, heads, d), g = split_qkv(d_qkv, b, s, heads, heads, d);
  k_attn_bwd(f.q, f.k, f.v, o, lse, d_o.view({b, s, heads, d}), scale, false, key_valid, g.q, g.k, g.v, p_attn, seed_attn, {},
             Tensor(), Tensor(), kBertPrescale, seed_word(seeds_dev, 0));
  d_o = Tensor();
  Tensor dbqkv = k_colsum(d_qkv);
  Tensor d_x = gemm_plain(d_qkv, wqkv, false, true, {}, d_x_res, TAMD_EPI_RESIDUAL);
  Tensor dwqkv;  // [3 hd, hd]
  if (kBertGroupDw) {
    std::vector<Tensor> dws = gemm_dw_group({d_b, d_pre, d_a, d_qkv}, {inter, h1, o.view({t, hd}), x});
    dwo2 = dws[0], dwi = dws[1], dwo = dws[2], dwqkv = dws[3];
  } else {
    dwqkv = gemm_plain(d_qkv, x, true, true);
  }
  return {d_x.view({b, s, hd}), dwqkv, dbqkv, dwo, dbo, dw_ln1, db_ln1, dwi, dbi, dwo2, dbo2, dw_ln2, db_ln2};
}

}  // namespace

// ================================================================================================ registration
TORCH_LIBRARY(tamd, m) {
  // kernel-level ops
  m.def("rmsnorm_fwd(Tensor x, Tensor w, float eps, Tensor? residual=None) -> (Tensor, Tensor, Tensor)");
  m.def("rmsnorm_bwd(Tensor dy, Tensor h, Tensor w, Tensor rstd, Tensor? dres=None) -> (Tensor, Tensor)");
  m.def("layernorm_fwd(Tensor x, Tensor w, Tensor? b, float eps, Tensor? residual=None) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("layernorm_bwd(Tensor dy, Tensor h, Tensor w, Tensor mean, Tensor rstd, Tensor? dres=None, bool need_db=True, "
        "bool need_colsum=False) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("layernorm_dropout_fwd(Tensor x, Tensor w, Tensor? b, float eps, Tensor residual, float dropout_p, int seed, "
        "Tensor? seed_dev=None) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("layernorm_dropout_bwd(Tensor dy, Tensor h, Tensor w, Tensor mean, Tensor rstd, float dropout_p, int seed, "
        "Tensor? dres=None, bool need_db=True, bool need_colsum=False, Tensor? seed_dev=None) -> "
        "(Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("rope_(Tensor(a!) x2d, Tensor cos, Tensor sin, int seq, int nheads, int head_dim, bool conj=False) -> ()");
  m.def("embedding_fwd(Tensor ids, Tensor table) -> Tensor");
  m.def("embedding_bwd(Tensor ids, Tensor dout, int vocab, int padding_idx=-1) -> Tensor");
  m.def("bert_embeddings_fwd(Tensor input_ids, Tensor token_type_ids, Tensor position_ids, Tensor word, Tensor typ, "
        "Tensor pos, Tensor ln_w, Tensor ln_b, float eps, bool keep_pre_ln) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("swiglu_fwd(Tensor gu) -> Tensor");
  m.def("swiglu_bwd(Tensor gu, Tensor dact, bool want_act=False) -> (Tensor, Tensor)");
  m.def("bias_act_fwd(Tensor x, Tensor? bias, int act) -> Tensor");
  m.def("bias_act_bwd(Tensor x, Tensor? bias, Tensor dy, int act, bool need_colsum=False) -> (Tensor, Tensor)");
  m.def("add(Tensor a, Tensor b) -> Tensor");
  m.def("colsum(Tensor x2d) -> Tensor");
  m.def("transpose(Tensor x2d) -> Tensor");
  m.def("cross_entropy_fwd(Tensor logits2d, Tensor labels, int ignore_index=-100) -> (Tensor, Tensor)");
  m.def("cross_entropy_bwd(Tensor logits2d, Tensor labels, Tensor lse, Tensor gscale, int ignore_index=-100) -> Tensor");
  m.def("adamw_step_(Tensor(a!) p, Tensor g, Tensor(b!) m, Tensor(c!) v, float lr, float beta1, float beta2, float eps, "
        "float weight_decay, int step, float grad_scale=1.0) -> ()");
  m.def("mt_sumsq(Tensor table, int n_tensors, int total_chunks, Tensor(a!) partials, int dtype) -> ()");
  m.def("mt_norm_finish(Tensor partials, Tensor(a!) out, float max_norm) -> ()");
  m.def("mt_scale_(Tensor table, int n_tensors, int total_chunks, Tensor coef, int dtype) -> ()");
  m.def("mt_adamw_step_(Tensor table, int n_tensors, int total_chunks, float lr, float beta1, float beta2, float eps, "
        "float weight_decay, int step, float grad_scale, Tensor? grad_scale_dev, int dtype, int state_dtype) -> ()");
  m.def("gemm(Tensor a, Tensor b, bool a_km=False, bool b_kn=False, Tensor? bias=None, Tensor? residual=None, "
        "int epilogue=0, int act=0, int sched=0) -> Tensor");
  m.def("gemm_out(Tensor(a!) out, Tensor a, Tensor b, bool a_km=False, bool b_kn=False, Tensor? bias=None, "
        "Tensor? residual=None, int epilogue=0, int act=0, int sched=0) -> ()");
  m.def("gemm_swiglu(Tensor x2, Tensor wgu, bool need_gu=True) -> (Tensor, Tensor)");
  m.def("gemm_swiglu_bwd(Tensor dy, Tensor wd, Tensor gu) -> Tensor");
  m.def("gemm_dw_segments(Tensor dy, Tensor x, Tensor(a!)[] segs) -> ()");
  m.def("gemm_dw_group(Tensor[] dy, Tensor[] x) -> Tensor[]");
  m.def("gemm_colscale(Tensor x2, Tensor w, Tensor? bias, int scale_cols, float col_scale) -> Tensor");
  m.def("attn_fwd(Tensor q, Tensor k, Tensor v, float scale, bool causal, Tensor? key_valid=None, bool need_lse=True, "
        "float dropout_p=0.0, int seed=0, Tensor? q_start=None, Tensor? seed_dev=None) -> (Tensor, Tensor)");
  m.def("attn_bwd(Tensor q, Tensor k, Tensor v, Tensor o, Tensor lse, Tensor dout, float scale, bool causal, "
        "Tensor? key_valid=None, float dropout_p=0.0, int seed=0, Tensor? q_start=None, Tensor? rope_cos=None, "
        "Tensor? rope_sin=None, Tensor? seed_dev=None) -> (Tensor, Tensor, Tensor)");
  m.def("attn_bwd_out(Tensor(a!) dq, Tensor(b!) dk, Tensor(c!) dv, Tensor q, Tensor k, Tensor v, Tensor o, Tensor lse, "
        "Tensor dout, float scale, bool causal, Tensor? key_valid=None, float dropout_p=0.0, int seed=0, "
        "Tensor? q_start=None, Tensor? rope_cos=None, Tensor? rope_sin=None, Tensor? seed_dev=None) -> ()");
  // differentiable ops (forward implementations here; fake + autograd registered from Python)
  m.def("rmsnorm(Tensor x, Tensor w, float eps) -> (Tensor, Tensor)");
  m.def("add_rmsnorm(Tensor x, Tensor residual, Tensor w, float eps) -> (Tensor, Tensor, Tensor)");
  m.def("layernorm(Tensor x, Tensor w, Tensor? b, float eps) -> (Tensor, Tensor, Tensor)");
  m.def("add_layernorm(Tensor x, Tensor residual, Tensor w, Tensor? b, float eps) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("dropout_add_layernorm(Tensor x, Tensor residual, Tensor w, Tensor? b, float eps, float dropout_p, int seed, "
        "Tensor? seed_dev=None) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("linear(Tensor x, Tensor w, Tensor? bias, Tensor? residual, int act, bool train) -> (Tensor, Tensor)");
  m.def("fused_linear(Tensor x, Tensor wf, Tensor? bf, Tensor[] members) -> Tensor");
  m.def("conv1d(Tensor x, Tensor w, Tensor? b) -> Tensor");
  m.def("rope(Tensor x, Tensor cos, Tensor sin, int nheads, int head_dim, bool conj=False) -> Tensor");
  m.def("attention(Tensor q, Tensor k, Tensor v, Tensor? key_valid, float scale, bool causal, float dropout_p, int seed, "
        "Tensor? q_start, bool train, Tensor? seed_dev=None) -> (Tensor, Tensor)");
  m.def("swiglu(Tensor gu) -> Tensor");
  m.def("bias_act(Tensor x, Tensor? bias, int act) -> Tensor");
  m.def("embedding(Tensor ids, Tensor table, int padding_idx=-1) -> Tensor");
  m.def("bert_embeddings(Tensor input_ids, Tensor token_type_ids, Tensor position_ids, Tensor word, Tensor typ, Tensor pos, "
        "Tensor ln_w, Tensor ln_b, float eps, int padding_idx, bool train) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("cross_entropy_sum(Tensor logits2d, Tensor labels, int ignore_index=-100) -> (Tensor, Tensor)");
  m.def("linear_cross_entropy(Tensor h2d, Tensor w, Tensor labels, Tensor normaliser, int ignore_index, int chunk, "
        "bool need_dh, bool need_dw) -> (Tensor, Tensor, Tensor)");
  m.def("padded_vocab_head(Tensor h2d, Tensor w_pad, Tensor? b_pad, Tensor w, Tensor? b, Tensor? labels, int ignore_index, "
        "bool train) -> (Tensor, Tensor, Tensor)");
  m.def("padded_vocab_head_bwd(Tensor? g_loss, Tensor? g_logits, Tensor h2d, Tensor w_pad, Tensor? labels, Tensor logits, "
        "Tensor lse, int ignore_index, bool need_dh, bool need_dw, bool need_db) -> (Tensor, Tensor, Tensor)");
  m.def("llama_layer(Tensor h_in, Tensor cos, Tensor sin, Tensor? key_valid, Tensor? q_start, Tensor w_ln1, Tensor wqkv, "
        "Tensor wq, Tensor wk, Tensor wv, Tensor wo, Tensor w_ln2, Tensor wgu, Tensor wg, Tensor wu, Tensor wd, float eps, "
        "int hq, int hkv, int d, float scale, bool causal, bool train) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, "
        "Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("llama_layer_bwd(Tensor d_hout, Tensor h_in, Tensor cos, Tensor sin, Tensor? key_valid, Tensor? q_start, "
        "Tensor w_ln1, Tensor wqkv, Tensor wo, Tensor w_ln2, Tensor wgu, Tensor wd, Tensor rstd1, Tensor xn, Tensor qkv, "
        "Tensor o, Tensor lse, Tensor h_mid, Tensor rstd2, Tensor xn2, Tensor gu, Tensor act_saved, int hq, int hkv, int d, "
        "float scale, bool causal, Tensor(a!)? dst_q=None, Tensor(b!)? dst_k=None, Tensor(c!)? dst_v=None, "
        "Tensor(d!)? dst_o=None, Tensor(e!)? dst_g=None, Tensor(f!)? dst_u=None, Tensor(g!)? dst_d=None) -> "
        "(Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("bert_layer(Tensor h_in, Tensor? key_valid, Tensor wqkv, Tensor bqkv, Tensor wq, Tensor wk, Tensor wv, Tensor bq, "
        "Tensor bk, Tensor bv, Tensor wo, Tensor bo, Tensor ln1_w, Tensor ln1_b, Tensor wi, Tensor bi, Tensor wo2, "
        "Tensor bo2, Tensor ln2_w, Tensor ln2_b, float eps, int heads, int d, float scale, int act, float p_attn, "
        "float p_hidden, int seed_attn, int seed1, int seed2, bool train, Tensor? seeds_dev=None) -> (Tensor, Tensor, Tensor, "
        "Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("bert_layer_bwd(Tensor d_out, Tensor h_in, Tensor? key_valid, Tensor wqkv, Tensor wo, Tensor ln1_w, Tensor wi, "
        "Tensor wo2, Tensor ln2_w, Tensor qkv, Tensor o, Tensor lse, Tensor y1, Tensor mean1, Tensor rstd1, Tensor h1, "
        "Tensor pre, Tensor inter, Tensor y2, Tensor mean2, Tensor rstd2, int heads, int d, float scale, int act, "
        "float p_attn, float p_hidden, int seed_attn, int seed1, int seed2, Tensor? seeds_dev=None) -> (Tensor, Tensor, Tensor, "
        "Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
}

// The implementations are registered for the CUDA key (HIP on ROCm) and for CPU: with the product library a CPU tensor is
// refused in `Launch` (there is no CPU implementation); the CPU test-suite binds the CPU execution model of the kernels.
#define TAMD_IMPLS(m)                                                \
  m.impl("rmsnorm_fwd", &op_rmsnorm_fwd);                            \
  m.impl("rmsnorm_bwd", &k_rmsnorm_bwd);                             \
  m.impl("layernorm_fwd", &op_layernorm_fwd);                        \
  m.impl("layernorm_bwd", &op_layernorm_bwd);                        \
  m.impl("layernorm_dropout_fwd", &op_layernorm_dropout_fwd);        \
  m.impl("layernorm_dropout_bwd", &op_layernorm_dropout_bwd);        \
  m.impl("rope_", &op_rope_);                                        \
  m.impl("embedding_fwd", &k_embedding_fwd);                         \
  m.impl("embedding_bwd", &k_embedding_bwd);                         \
  m.impl("bert_embeddings_fwd", &op_bert_embeddings_fwd);            \
  m.impl("swiglu_fwd", &k_swiglu_fwd);                               \
  m.impl("swiglu_bwd", &op_swiglu_bwd);                              \
  m.impl("bias_act_fwd", &k_bias_act_fwd);                           \
  m.impl("bias_act_bwd", &op_bias_act_bwd);                          \
  m.impl("gemm_dw_segments", &op_gemm_dw_segments);                  \
  m.impl("gemm_dw_group", &op_gemm_dw_group);                        \
  m.impl("gemm_colscale", &op_gemm_colscale);                        \
  m.impl("add", &k_add);                                             \
  m.impl("colsum", &k_colsum);                                       \
  m.impl("transpose", &k_transpose);                                 \
  m.impl("cross_entropy_fwd", &k_cross_entropy_fwd);                 \
  m.impl("cross_entropy_bwd", &op_cross_entropy_bwd);                \
  m.impl("adamw_step_", &op_adamw_step_);                            \
  m.impl("mt_sumsq", &k_mt_sumsq);                                   \
  m.impl("mt_norm_finish", &k_mt_norm_finish);                       \
  m.impl("mt_scale_", &k_mt_scale_);                                 \
  m.impl("mt_adamw_step_", &k_mt_adamw_step_);                       \
  m.impl("gemm", &op_gemm);                                          \
  m.impl("gemm_out", &op_gemm_out);                                  \
  m.impl("gemm_swiglu", &op_gemm_swiglu);                            \
  m.impl("gemm_swiglu_bwd", &k_gemm_swiglu_bwd);                     \
  m.impl("attn_fwd", &op_attn_fwd);                                  \
  m.impl("attn_bwd", &op_attn_bwd);                                  \
  m.impl("attn_bwd_out", &op_attn_bwd_out);                          \
  m.impl("rmsnorm", &op_rmsnorm);                                    \
  m.impl("add_rmsnorm", &op_add_rmsnorm);                            \
  m.impl("layernorm", &op_layernorm);                                \
  m.impl("add_layernorm", &op_add_layernorm);                        \
  m.impl("dropout_add_layernorm", &op_dropout_add_layernorm);        \
  m.impl("linear", &op_linear);                                      \
  m.impl("fused_linear", &op_fused_linear);                          \
  m.impl("conv1d", &op_conv1d);                                      \
  m.impl("rope", &op_rope);                                          \
  m.impl("attention", &op_attention);                                \
  m.impl("swiglu", &op_swiglu);                                      \
  m.impl("bias_act", &k_bias_act_fwd);                               \
  m.impl("embedding", &op_embedding);                                \
  m.impl("bert_embeddings", &op_bert_embeddings);                    \
  m.impl("cross_entropy_sum", &op_cross_entropy_sum);                \
  m.impl("linear_cross_entropy", &op_linear_cross_entropy);          \
  m.impl("padded_vocab_head", &op_padded_vocab_head);                \
  m.impl("padded_vocab_head_bwd", &op_padded_vocab_head_bwd);        \
  m.impl("llama_layer", &op_llama_layer);                            \
  m.impl("llama_layer_bwd", &op_llama_layer_bwd);                    \
  m.impl("bert_layer", &op_bert_layer);                              \
  m.impl("bert_layer_bwd", &op_bert_layer_bwd);

TORCH_LIBRARY_IMPL(tamd, CUDA, m) { TAMD_IMPLS(m) }
TORCH_LIBRARY_IMPL(tamd, CPU, m) { TAMD_IMPLS(m) }

// ================================================================================================ C entry points (ctypes)
extern "C" {

// Bind the C-ABI library the ops call into.  `emulated` != 0: the CPU execution model (CPU tensors are its memory).
// Returns 0, or -1 with the reason in tamd_torch_last_error().
static std::string g_last_error;
const char* tamd_torch_last_error(void) { return g_last_error.c_str(); }

int tamd_torch_bind(const char* path, int emulated) {
  std::lock_guard<std::mutex> lock(g_bind_mutex);
  for (Api* a : g_bound)
    if (a->path == path && a->emulated == (emulated != 0)) {
      g_api.store(a, std::memory_order_release);
      return 0;
    }
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (h == nullptr) {
    g_last_error = std::string("dlopen failed: ") + dlerror();
    return -1;
  }
  Api* a = new Api();
  a->handle = h;
  a->emulated = emulated != 0;
  a->path = path;
  std::string missing;
#define X(name)                                              \
  a->name = reinterpret_cast<decltype(a->name)>(dlsym(h, #name)); \
  if (a->name == nullptr) missing += std::string(" ") + #name;
  TAMD_API_LIST(X)
#undef X
  if (!missing.empty()) {
    g_last_error = std::string(path) + " does not export:" + missing;
    delete a;
    return -1;
  }
  if (a->tamd_abi_version() != TAMD_ABI_VERSION) {
    g_last_error = std::string(path) + ": ABI version " + std::to_string(a->tamd_abi_version()) + ", expected " +
                   std::to_string(TAMD_ABI_VERSION);
    delete a;
    return -1;
  }
  g_bound.push_back(a);
  g_api.store(a, std::memory_order_release);
  return 0;
}

// path of the library the ops are currently bound to ("" when none)
const char* tamd_torch_bound_path(void) {
  const Api* a = g_api.load(std::memory_order_acquire);
  return a ? a->path.c_str() : "";
}

// where gemm_dw_balanced cuts a weight-gradient product dW[m, n] over k tokens (host logic, for the tests): returns the axis
// (0 rows, 1 columns, -1 one launch) and the cut position in *at
int tamd_torch_dw_cut(long long m, long long n, long long k, long long* at, int cus) {
  const DwCut c = dw_balanced_cut(m, n, k, cus > 0 ? cus : 256);
  if (at) *at = c.at;
  return c.axis;
}
// The measured choices of swiglu_bwd_fused (bench.py reports them): record i -> out[0..5] = tokens, I, K, fused (1 / 0),
// fused ms, two-kernel ms; returns the number of records
int tamd_torch_swiglu_bwd_choice(int i, double* out) {
  std::lock_guard<std::mutex> lock(g_swiglu_bwd_mutex);
  const int n = (int)g_swiglu_bwd_choices.size();
  if (i >= 0 && i < n && out != nullptr) {
    const SwigluBwdChoice& c = g_swiglu_bwd_choices[(size_t)i];
    out[0] = (double)c.t, out[1] = (double)c.inter, out[2] = (double)c.k, out[3] = c.fused ? 1.0 : 0.0;
    out[4] = c.fused_ms, out[5] = c.two_ms;
  }
  return n;
}
// A/B switch of the cut (on by default); returns the previous setting
int tamd_torch_set_dw_balance(int on) { return g_dw_balance.exchange(on != 0) ? 1 : 0; }

// GEMM event log (bench.py `roofline`): on / off; the summary synchronises the recorded events and clears the log.
//   out[0] launches, out[1] sum of algorithmic FLOPs, out[2] sum of event durations (ms), out[3] sum of algorithmic bytes;
//   out[4..6] launches / FLOPs / ms of the launches among them that carry the SiLU*up backward in their way out
void tamd_torch_gemm_log(int on) {
  if (on) {
    std::lock_guard<std::mutex> lock(g_gemm_log_mutex);
    for (auto& r : g_gemm_records) {
      (void)hipEventDestroy(r.start);
      (void)hipEventDestroy(r.stop);
    }
    g_gemm_records.clear();
  }
  g_gemm_log.store(on != 0);
}
int tamd_torch_gemm_log_summary(double* out) {
  std::lock_guard<std::mutex> lock(g_gemm_log_mutex);
  for (int i = 0; i < 7; ++i) out[i] = 0.0;
  for (auto& r : g_gemm_records) {
    float ms = 0.f;
    if (hipEventSynchronize(r.stop) != hipSuccess || hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess) return -1;
    out[0] += 1.0;
    out[1] += r.flops;
    out[2] += ms;
    out[3] += r.bytes;
    if (r.fused_bwd) {
      out[4] += 1.0;
      out[5] += r.flops;
      out[6] += ms;
    }
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  g_gemm_records.clear();
  return 0;
}
}
