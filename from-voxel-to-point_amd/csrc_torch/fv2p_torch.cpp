// fv2p_torch — compiled torch binding of the hot autograd ops on top of the C ABI (include/fv2p_ops.h).
//
// The reference binds its ops with pybind torch extensions too (pcdet/ops/spconv/src/all.cc:18-62) and keeps the
// autograd Functions in Python (spconv/functional.py:20-175).  At BASELINE's batch size the training step is bound
// by host work, not kernels, and the Python autograd Functions + ctypes crossings are most of it; here
// the two Functions every backbone block runs — sparse conv and BatchNorm1d(+ReLU) — are torch::autograd::Functions:
// their backward runs on the autograd engine's thread without the GIL.  No kernels live here: every call goes
// through libfv2p_ops.so.  Optional: without this module the same ops run through fv2p_native.py (ctypes).
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPCachingAllocator.h>
#include <c10/core/DeviceGuard.h>
#include <hip/hip_runtime_api.h>

#include <cstdlib>

#include <map>
#include <tuple>
#include <vector>
#include <mutex>
#include <string>

#include "../../include/fv2p_ops.h"

namespace {

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

void* cur_stream(const at::Tensor& t) { return static_cast<void*>(c10::hip::getCurrentHIPStream(t.device().index()).stream()); }

void check(int rc, const char* what) {
  TORCH_CHECK(rc >= 0, what, " failed (", rc, "): ", fv2p_last_error());
}

// grow-only scratch per (device, stream): library calls on one stream are ordered by the stream itself
at::Tensor workspace(size_t bytes, const at::Tensor& like, void* stream) {
  static std::mutex mu;
  static std::map<std::pair<int, void*>, at::Tensor> pool;
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_pair(static_cast<int>(like.device().index()), stream);
  auto it = pool.find(key);
  if (it == pool.end() || static_cast<size_t>(it->second.numel()) < bytes) {
    const int64_t cap = static_cast<int64_t>(std::max<size_t>(bytes * 2, size_t(1) << 22));
    pool[key] = at::empty({cap}, like.options().dtype(at::kByte));
    it = pool.find(key);
  }
  return it->second;
}

// Second stream per device for the weight gradient: it depends only on what the backward-data conv depends on, so the
// two run side by side and fill each other's tails (both are tile kernels whose launches end on a few heavy CUs).
// The training stream waits for the side stream before the backward function returns, so nothing downstream (gradient
// accumulation, DDP bucket copies, the optimiser) sees a half-written dW.  FV2P_WGRAD_OVERLAP=0 keeps one stream.
struct SideStream {
  c10::hip::HIPStream stream;
  hipEvent_t fork, join;
};
SideStream& side_stream(int device) {
  static std::mutex mu;
  static std::map<int, SideStream> pool;
  std::lock_guard<std::mutex> lock(mu);
  auto it = pool.find(device);
  if (it == pool.end()) {
    SideStream s{c10::hip::getStreamFromPool(false, static_cast<c10::DeviceIndex>(device)), nullptr, nullptr};
    TORCH_CHECK(hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&s.join, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
    it = pool.emplace(device, s).first;
  }
  return it->second;
}
bool wgrad_overlap() {
  static const bool on = [] { const char* e = std::getenv("FV2P_WGRAD_OVERLAP"); return !(e && e[0] == '0'); }();
  return on;
}

void require_f32_cuda(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), name, ": fv2p ops run on the GPU only (no CPU fallback exists)");
  TORCH_CHECK(t.scalar_type() == at::kFloat, name, ": float32 expected");
}

// Column statistics of a conv that feeds BatchNorm come out of its epilogue (fv2p_sparse_conv_rows_stats).  Two slot
// buffers per (device, stream) alternate: the BatchNorm launch that reads one clears what the previous user left in the
// other, so the next fused conv — later on the same stream — finds it zeroed without a fill launch of its own.
// The backward pass has its own pair: a backward-data conv leaves (sum dz, sum dz * xhat) of the BatchNorm that produced its
// input, the BatchNorm's backward node consumes them a few autograd nodes later (`unread` guards a buffer until then).
struct StatRing {
  at::Tensor buf[2];
  int64_t dirty[2] = {0, 0};   // doubles the last user of each buffer wrote
  bool unread[2] = {false, false};
  int cur = 0;
};
StatRing& stat_ring(const at::Tensor& like, void* stream, int which = 0) {
  static std::mutex mu;
  static std::map<std::tuple<int, void*, int>, StatRing> pool;
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_tuple(static_cast<int>(like.device().index()), stream, which);
  auto it = pool.find(key);
  if (it == pool.end()) {
    StatRing r;
    const int64_t cap = static_cast<int64_t>(fv2p_sparse_conv_stat_slots()) * 2 * 1024;
    for (auto& b : r.buf) b = at::zeros({cap}, like.options().dtype(at::kDouble));
    it = pool.emplace(key, std::move(r)).first;
  }
  return it->second;
}
static int g_bn_epilogue = -1;   // FV2P_BN_EPILOGUE=0 or set_bn_epilogue(false): BatchNorm takes its own sums (tests compare the two)
static bool fuse_bn_stats() {
  if (g_bn_epilogue < 0) { const char* e = std::getenv("FV2P_BN_EPILOGUE"); g_bn_epilogue = !(e && e[0] == '0'); }
  return g_bn_epilogue != 0;
}
void set_bn_epilogue(bool on) { g_bn_epilogue = on ? 1 : 0; }


// ---- BatchNorm1d (+ReLU) on [N, C] ---------------------------------------------------------------------------------------
struct BnReluFn : public torch::autograd::Function<BnReluFn> {
  static at::Tensor forward(AutogradContext* ctx, const at::Tensor& x_, const c10::optional<at::Tensor>& weight, const c10::optional<at::Tensor>& bias,
                            const c10::optional<at::Tensor>& running_mean, const c10::optional<at::Tensor>& running_var,
                            const c10::optional<at::Tensor>& num_batches_tracked, bool training, double momentum, double eps, bool relu,
                            const c10::optional<at::Tensor>& stats, const c10::optional<at::Tensor>& zero_next, int64_t zero_count) {
    require_f32_cuda(x_, "input");
    const at::Tensor x = x_.contiguous();
    const int64_t n = x.size(0), c = x.size(1);
    const bool has_running = running_mean.has_value() && running_mean->defined();
    const bool batch_stats = training || !has_running;
    c10::DeviceGuard guard(x.device());
    void* stream = cur_stream(x);
    at::Tensor y = at::empty_like(x);
    at::Tensor mean, invstd;
    const float* gamma = (weight.has_value() && weight->defined()) ? weight->data_ptr<float>() : nullptr;
    const float* beta = (bias.has_value() && bias->defined()) ? bias->data_ptr<float>() : nullptr;
    if (batch_stats) {
      at::Tensor saved = at::empty({2, c}, x.options());
      mean = saved[0];
      invstd = saved[1];
      const bool track = training && has_running;
      int64_t* nbt = (track && num_batches_tracked.has_value() && num_batches_tracked->defined()) ? num_batches_tracked->data_ptr<int64_t>() : nullptr;
      if (stats.has_value() && stats->defined()) {   // sums taken by the producing conv's epilogue: one launch
        check(fv2p_batchnorm_forward_stats(x.data_ptr<float>(), n, static_cast<int>(c), static_cast<float>(eps), static_cast<float>(momentum), gamma,
                                           beta, relu ? 1 : 0, track ? running_mean->data_ptr<float>() : nullptr,
                                           track ? running_var->data_ptr<float>() : nullptr, nbt, mean.data_ptr<float>(), invstd.data_ptr<float>(),
                                           y.data_ptr<float>(), stats->data_ptr<double>(),
                                           (zero_next.has_value() && zero_next->defined()) ? zero_next->data_ptr<double>() : nullptr, zero_count, stream),
              "fv2p_batchnorm_forward_stats");
      } else {
      at::Tensor ws = workspace(fv2p_batchnorm_ws_bytes(n, static_cast<int>(c)), x, stream);
      check(fv2p_batchnorm_forward(x.data_ptr<float>(), n, static_cast<int>(c), static_cast<float>(eps), static_cast<float>(momentum), gamma, beta,
                                   relu ? 1 : 0, track ? running_mean->data_ptr<float>() : nullptr, track ? running_var->data_ptr<float>() : nullptr,
                                   nbt, mean.data_ptr<float>(), invstd.data_ptr<float>(), y.data_ptr<float>(), ws.data_ptr(),
                                   static_cast<size_t>(ws.numel()), stream),
            "fv2p_batchnorm_forward");
      }
    } else {
      mean = *running_mean;
      invstd = at::rsqrt(*running_var + eps);
      check(fv2p_batchnorm_apply(x.data_ptr<float>(), n, static_cast<int>(c), mean.data_ptr<float>(), invstd.data_ptr<float>(), gamma, beta,
                                 relu ? 1 : 0, y.data_ptr<float>(), stream),
            "fv2p_batchnorm_apply");
    }
    ctx->save_for_backward({x, mean, invstd, weight.has_value() ? *weight : at::Tensor(), bias.has_value() ? *bias : at::Tensor()});
    ctx->saved_data["relu"] = relu;
    ctx->saved_data["batch_stats"] = batch_stats;
    return y;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const at::Tensor &x = saved[0], &mean = saved[1], &invstd = saved[2], &weight = saved[3], &bias = saved[4];
    const at::Tensor dy = grads[0].contiguous();
    const int64_t n = x.size(0), c = x.size(1);
    c10::DeviceGuard guard(x.device());
    void* stream = cur_stream(x);
    at::Tensor dx = at::empty_like(x);
    at::Tensor dpar = at::empty({2, c}, x.options());
    // sums left by the backward-data conv that produced exactly this dy (SparseConvFn::backward): one launch
    auto it = ctx->saved_data.find("stats_buf");
    if (it != ctx->saved_data.end()) {
      const int b = static_cast<int>(it->second.toInt());
      // exactly the tensor that conv wrote, untouched: the engine sums several gradients into a new tensor or in place
      // into the first one (which bumps its version counter)
      const bool mine = ctx->saved_data["stats_dy"].toInt() == reinterpret_cast<int64_t>(dy.data_ptr()) &&
                        ctx->saved_data["stats_ver"].toInt() == static_cast<int64_t>(dy._version());
      ctx->saved_data.erase("stats_buf");
      StatRing& ring = stat_ring(x, stream, 1);
      ring.unread[b] = false;   // read below, or abandoned (dy is a sum of several gradients): the next producer clears it
      if (mine) {
        const int other = 1 - b;
        const int64_t zc = ring.unread[other] ? 0 : ring.dirty[other];
        check(fv2p_batchnorm_backward_stats(x.data_ptr<float>(), dy.data_ptr<float>(), n, static_cast<int>(c), mean.data_ptr<float>(),
                                            invstd.data_ptr<float>(), weight.defined() ? weight.data_ptr<float>() : nullptr,
                                            bias.defined() ? bias.data_ptr<float>() : nullptr, ctx->saved_data["relu"].toBool() ? 1 : 0,
                                            ctx->saved_data["batch_stats"].toBool() ? 1 : 0, dx.data_ptr<float>(), dpar[0].data_ptr<float>(),
                                            dpar[1].data_ptr<float>(), ring.buf[b].data_ptr<double>(), zc ? ring.buf[other].data_ptr<double>() : nullptr,
                                            zc, stream),
              "fv2p_batchnorm_backward_stats");
        if (zc) ring.dirty[other] = 0;
        return {dx, weight.defined() ? dpar[0] : at::Tensor(), bias.defined() ? dpar[1] : at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(),
                at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
      }
    }
    at::Tensor ws = workspace(fv2p_batchnorm_ws_bytes(n, static_cast<int>(c)), x, stream);
    check(fv2p_batchnorm_backward(x.data_ptr<float>(), dy.data_ptr<float>(), n, static_cast<int>(c), mean.data_ptr<float>(), invstd.data_ptr<float>(),
                                  weight.defined() ? weight.data_ptr<float>() : nullptr, bias.defined() ? bias.data_ptr<float>() : nullptr,
                                  ctx->saved_data["relu"].toBool() ? 1 : 0, ctx->saved_data["batch_stats"].toBool() ? 1 : 0, dx.data_ptr<float>(),
                                  dpar[0].data_ptr<float>(), dpar[1].data_ptr<float>(), ws.data_ptr(), static_cast<size_t>(ws.numel()), stream),
          "fv2p_batchnorm_backward");
    return {dx, weight.defined() ? dpar[0] : at::Tensor(), bias.defined() ? dpar[1] : at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(),
            at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
  }
};


// ---- deferred join of the weight-gradient stream ------------------------------------------------------------------------
// Joining the side stream before a conv's backward returns leaves the training stream idle for the tail of every weight
// gradient (measured: 2.14 -> 1.93 ms per step without the joins).  gate_weights() is applied to all conv weights at the
// top of the model's forward pass: one autograd node whose outputs alias the weights and whose backward therefore runs
// when every conv that used them has produced its dW — the end of the backward pass.  A conv whose weight comes from the
// gate launches dW on the side stream, parks references to everything that launch reads (the training stream's
// allocator would otherwise recycle them) and returns; the gate's backward joins the side stream once, drops the
// references and hands the dW tensors on to AccumulateGrad (and the hooks DistributedDataParallel has there).
static std::mutex g_pending_mu;
static std::map<int, std::vector<at::Tensor>> g_pending;   // per device: tensors the not yet joined side-stream work reads or writes

struct WeightGateFn : public torch::autograd::Function<WeightGateFn> {
  static variable_list forward(AutogradContext* ctx, at::TensorList weights) {
    ctx->set_materialize_grads(false);   // a weight no conv used keeps grad None instead of receiving zeros
    return weights.vec();   // returned as-is: autograd turns each into a view of its weight with this node as grad_fn
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    (void)ctx;
    for (const at::Tensor& g : grads) {
      if (!g.defined() || !g.is_cuda()) continue;
      const int dev = g.device().index();
      c10::DeviceGuard guard(g.device());
      std::vector<at::Tensor> parked;
      {
        std::lock_guard<std::mutex> lock(g_pending_mu);
        parked.swap(g_pending[dev]);
      }
      if (!parked.empty()) {
        SideStream& side = side_stream(dev);
        TORCH_CHECK(hipEventRecord(side.join, side.stream.stream()) == hipSuccess, "hipEventRecord failed");
        TORCH_CHECK(hipStreamWaitEvent(c10::hip::getCurrentHIPStream(static_cast<c10::DeviceIndex>(dev)).stream(), side.join, 0) == hipSuccess,
                    "hipStreamWaitEvent failed");
      }   // `parked` is released here, behind the join on the training stream
    }
    return grads;
  }
};
std::vector<at::Tensor> gate_weights(const std::vector<at::Tensor>& weights) { return WeightGateFn::apply(at::TensorList(weights)); }
static bool is_gated(const at::Tensor& w) {
  return w.defined() && w.grad_fn() && dynamic_cast<torch::autograd::CppNode<WeightGateFn>*>(w.grad_fn().get()) != nullptr;
}

// ---- sparse convolution: out[r] = sum_k feat[tab_f[k][r]] . W_k ------------------------------------------------------
struct SparseConvFn : public torch::autograd::Function<SparseConvFn> {
  static at::Tensor forward(AutogradContext* ctx, const at::Tensor& features_, const at::Tensor& weight_, const at::Tensor& tab_f,
                            int64_t flip_f, const at::Tensor& tab_b, int64_t flip_b, int64_t n_out, int64_t centre,
                            const c10::optional<at::Tensor>& pairs, const c10::optional<at::Tensor>& pair_num, int64_t side_src,
                            const c10::optional<at::Tensor>& stats, const c10::optional<at::Tensor>& perm_b) {
    require_f32_cuda(features_, "features");
    require_f32_cuda(weight_, "weight");
    const at::Tensor features = features_.contiguous(), weight = weight_.contiguous();
    const int64_t cin = weight.size(-2), cout = weight.size(-1);
    const int64_t kvol = weight.numel() / (cin * cout);
    TORCH_CHECK(features.dim() == 2 && features.size(1) == cin, "features [N, Cin] expected");
    c10::DeviceGuard guard(features.device());
    void* stream = cur_stream(features);
    at::Tensor out = at::empty({n_out, cout}, features.options());
    if (stats.has_value() && stats->defined())   // BatchNorm follows: its column sums come out of the conv epilogue
      check(fv2p_sparse_conv_rows_stats(features.data_ptr<float>(), features.size(0), static_cast<int>(cin), weight.data_ptr<float>(),
                                        static_cast<int>(kvol), tab_f.data_ptr<int>(), n_out, static_cast<int>(cout), static_cast<int>(flip_f), 0,
                                        nullptr, out.data_ptr<float>(), stats->data_ptr<double>(), stream),
            "fv2p_sparse_conv_rows_stats");
    else
      check(fv2p_sparse_conv_rows(features.data_ptr<float>(), features.size(0), static_cast<int>(cin), weight.data_ptr<float>(),
                                  static_cast<int>(kvol), tab_f.data_ptr<int>(), n_out, static_cast<int>(cout), static_cast<int>(flip_f), 0,
                                  nullptr, out.data_ptr<float>(), stream),
            "fv2p_sparse_conv_rows");
    const bool have_pairs = pairs.has_value() && pairs->defined() && pair_num.has_value() && pair_num->defined();
    ctx->save_for_backward({features, weight, tab_f, tab_b, have_pairs ? *pairs : at::Tensor(), have_pairs ? *pair_num : at::Tensor(),
                            (perm_b.has_value() && perm_b->defined()) ? *perm_b : at::Tensor()});
    ctx->saved_data["side_src"] = side_src;
    ctx->saved_data["flip_f"] = flip_f;
    ctx->saved_data["flip_b"] = flip_b;
    ctx->saved_data["centre"] = centre;
    ctx->saved_data["gated"] = is_gated(weight_);
    // features straight out of a fused BatchNorm(+ReLU): the backward-data conv can take that layer's backward sums
    auto* bn_node = features_.grad_fn() ? dynamic_cast<torch::autograd::CppNode<BnReluFn>*>(features_.grad_fn().get()) : nullptr;
    ctx->saved_data["bn_node"] = reinterpret_cast<int64_t>(bn_node);   // kept alive by this node's edge to it
    return out;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const at::Tensor &features = saved[0], &weight = saved[1], &tab_f = saved[2], &tab_b = saved[3], &pairs = saved[4], &pair_num = saved[5];
    const int* perm_b = (saved[6].defined() && saved[6].numel() == features.size(0)) ? saved[6].data_ptr<int>() : nullptr;
    const at::Tensor g = grads[0].contiguous();
    const int64_t cin = weight.size(-2), cout = weight.size(-1);
    const int64_t kvol = weight.numel() / (cin * cout);
    const int flip_f = static_cast<int>(ctx->saved_data["flip_f"].toInt()), flip_b = static_cast<int>(ctx->saved_data["flip_b"].toInt());
    const int centre = static_cast<int>(ctx->saved_data["centre"].toInt());
    c10::DeviceGuard guard(features.device());
    void* stream = cur_stream(features);
    at::Tensor din, dw;
    const bool both = ctx->needs_input_grad(0) && ctx->needs_input_grad(1);
    const bool overlap = both && wgrad_overlap();
    void* wstream = stream;   // stream of the weight gradient
    SideStream* side = nullptr;
    if (ctx->needs_input_grad(1)) dw = at::empty_like(weight);
    if (overlap) {
      side = &side_stream(features.device().index());
      TORCH_CHECK(hipEventRecord(side->fork, static_cast<hipStream_t>(stream)) == hipSuccess, "hipEventRecord failed");
      TORCH_CHECK(hipStreamWaitEvent(side->stream.stream(), side->fork, 0) == hipSuccess, "hipStreamWaitEvent failed");
      wstream = static_cast<void*>(side->stream.stream());
    }
    if (ctx->needs_input_grad(0)) {
      din = at::empty_like(features);
      // backward data = a conv with W_k^T.  For the shapes of the K-split tile (64 / 128 gradient channels) the transposed slices are
      // materialised once ([K][Cout][Cin], one small copy kernel) and the conv reads them as a plain weight: reading W_k transposed in
      // place gathers 64-byte pieces of 16 rows per load — half of every cache line fetched is thrown away, and the K-split tile is bound
      // by exactly that traffic (measured: subm 128->128 backward data 85 -> 56 us)
      at::Tensor wt;
      int transpose_w = 1;
      if ((cout == 64 || cout == 128) && cin % 64 == 0 && cin <= 128 && kvol > 1) {
        wt = weight.view({kvol, cin, cout}).transpose(1, 2).contiguous();
        transpose_w = 0;
      }
      const float* w_bwd = wt.defined() ? wt.data_ptr<float>() : weight.data_ptr<float>();
      auto* bn_node = reinterpret_cast<torch::autograd::CppNode<BnReluFn>*>(ctx->saved_data["bn_node"].toInt());
      bool fused = false;
      if (bn_node && fuse_bn_stats() && cout <= 128 && cin <= 1024) {
        StatRing& ring = stat_ring(features, stream, 1);
        const int b = ring.cur;
        if (!ring.unread[b]) {
          if (ring.dirty[b]) { ring.buf[b].zero_(); ring.dirty[b] = 0; }   // left by sums nobody read: rare, costs a fill
          AutogradContext& bctx = bn_node->ctx_;
          const auto bsaved = bctx.get_saved_variables();   // x, mean, invstd, weight, bias of the BatchNorm
          const at::Tensor &bx = bsaved[0], &bmean = bsaved[1], &binv = bsaved[2], &bw = bsaved[3], &bb = bsaved[4];
          if (bx.defined() && bx.sizes() == din.sizes() && bx.is_contiguous()) {
            check(fv2p_sparse_conv_rows_bnbwd(g.data_ptr<float>(), g.size(0), static_cast<int>(cout), w_bwd, static_cast<int>(kvol),
                                              tab_b.data_ptr<int>(), features.size(0), static_cast<int>(cin), flip_b, transpose_w, din.data_ptr<float>(),
                                              bx.data_ptr<float>(), bmean.data_ptr<float>(), binv.data_ptr<float>(),
                                              bw.defined() ? bw.data_ptr<float>() : nullptr, bb.defined() ? bb.data_ptr<float>() : nullptr,
                                              bctx.saved_data["relu"].toBool() ? 1 : 0, ring.buf[b].data_ptr<double>(), perm_b, stream),
                  "fv2p_sparse_conv_rows_bnbwd");
            ring.dirty[b] = static_cast<int64_t>(fv2p_sparse_conv_stat_slots()) * 2 * cin;
            ring.unread[b] = true;
            ring.cur = 1 - b;
            bctx.saved_data["stats_buf"] = static_cast<int64_t>(b);
            bctx.saved_data["stats_dy"] = reinterpret_cast<int64_t>(din.data_ptr());
            bctx.saved_data["stats_ver"] = static_cast<int64_t>(din._version());
            fused = true;
          }
        }
      }
      if (!fused)
        check(fv2p_sparse_conv_rows_perm(g.data_ptr<float>(), g.size(0), static_cast<int>(cout), w_bwd, static_cast<int>(kvol),
                                         tab_b.data_ptr<int>(), features.size(0), static_cast<int>(cin), flip_b, transpose_w, nullptr, din.data_ptr<float>(),
                                         perm_b, stream),
              "fv2p_sparse_conv_rows (backward data)");
    }
    if (ctx->needs_input_grad(1)) {
      // scratch of the side stream is allocated under that stream, so the caching allocator recycles it in its order
      c10::optional<c10::hip::HIPStreamGuard> sg;
      if (overlap) sg.emplace(side->stream);
      if (pairs.defined()) {  // compacted pair lists of the rulebook: balanced by pairs, no compaction prologue
        const int64_t plen = pairs.size(2);
        const size_t wsb = fv2p_sparse_conv_wgrad_pairs_ws_bytes(plen, static_cast<int>(cin), static_cast<int>(cout), static_cast<int>(kvol));
        at::Tensor ws = workspace(wsb, features, wstream);
        check(fv2p_sparse_conv_wgrad_pairs(features.data_ptr<float>(), features.size(0), static_cast<int>(cin), g.data_ptr<float>(), g.size(0),
                                           static_cast<int>(cout), pairs.data_ptr<int>(), pair_num.data_ptr<int>(), static_cast<int>(kvol), plen,
                                           static_cast<int>(ctx->saved_data["side_src"].toInt()), dw.data_ptr<float>(), ws.data_ptr(),
                                           static_cast<size_t>(ws.numel()), wstream),
              "fv2p_sparse_conv_wgrad_pairs");
      } else {
        const size_t wsb = fv2p_sparse_conv_wgrad_ws_bytes(g.size(0), static_cast<int>(cin), static_cast<int>(cout), static_cast<int>(kvol));
        at::Tensor ws = workspace(wsb, features, wstream);
        check(fv2p_sparse_conv_wgrad(features.data_ptr<float>(), features.size(0), static_cast<int>(cin), g.data_ptr<float>(), tab_f.data_ptr<int>(),
                                     g.size(0), static_cast<int>(cout), static_cast<int>(kvol), flip_f, centre, dw.data_ptr<float>(), ws.data_ptr(),
                                     static_cast<size_t>(ws.numel()), wstream),
              "fv2p_sparse_conv_wgrad");
      }
    }
    if (overlap && ctx->saved_data["gated"].toBool()) {
      // joined by the weight gate at the end of the backward pass; until then nothing the launch touches may be recycled
      std::lock_guard<std::mutex> lock(g_pending_mu);
      auto& parked = g_pending[features.device().index()];
      for (const at::Tensor& t : {features, g, tab_f, pairs, pair_num, dw})
        if (t.defined()) parked.push_back(t);
    } else if (overlap) {
      TORCH_CHECK(hipEventRecord(side->join, side->stream.stream()) == hipSuccess, "hipEventRecord failed");
      TORCH_CHECK(hipStreamWaitEvent(static_cast<hipStream_t>(stream), side->join, 0) == hipSuccess, "hipStreamWaitEvent failed");
    }
    return {din, dw, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(),
            at::Tensor(), at::Tensor()};
  }
};

at::Tensor sparse_conv(const at::Tensor& features, const at::Tensor& weight, const at::Tensor& tab_f, int64_t flip_f, const at::Tensor& tab_b,
                       int64_t flip_b, int64_t n_out, int64_t centre, const c10::optional<at::Tensor>& pairs,
                       const c10::optional<at::Tensor>& pair_num, int64_t side_src, const c10::optional<at::Tensor>& perm_b) {
  return SparseConvFn::apply(features, weight, tab_f, flip_f, tab_b, flip_b, n_out, centre, pairs, pair_num, side_src, c10::optional<at::Tensor>(),
                             perm_b);
}

at::Tensor batch_norm_relu(const at::Tensor& x, const c10::optional<at::Tensor>& weight, const c10::optional<at::Tensor>& bias,
                           const c10::optional<at::Tensor>& running_mean, const c10::optional<at::Tensor>& running_var,
                           const c10::optional<at::Tensor>& num_batches_tracked, bool training, double momentum, double eps, bool relu) {
  return BnReluFn::apply(x, weight, bias, running_mean, running_var, num_batches_tracked, training, momentum, eps, relu, c10::optional<at::Tensor>(),
                         c10::optional<at::Tensor>(), 0);
}

// conv -> BatchNorm1d (-> ReLU) of one backbone block in one crossing from Python (post_act_block, spconv_backbone.py:8-27)
at::Tensor sparse_conv_bn_relu(const at::Tensor& features, const at::Tensor& weight, const at::Tensor& tab_f, int64_t flip_f, const at::Tensor& tab_b,
                               int64_t flip_b, int64_t n_out, int64_t centre, const c10::optional<at::Tensor>& pairs,
                               const c10::optional<at::Tensor>& pair_num, int64_t side_src, const c10::optional<at::Tensor>& conv_bias,
                               const c10::optional<at::Tensor>& bn_weight, const c10::optional<at::Tensor>& bn_bias,
                               const c10::optional<at::Tensor>& running_mean, const c10::optional<at::Tensor>& running_var,
                               const c10::optional<at::Tensor>& num_batches_tracked, bool training, double momentum, double eps, bool relu,
                               const c10::optional<at::Tensor>& perm_b) {
  if (n_out < 2 && training) return at::Tensor();   // torch raises for one value per channel: let the caller run the module
  const bool has_bias = conv_bias.has_value() && conv_bias->defined();
  const bool batch_stats = training || !(running_mean.has_value() && running_mean->defined());
  const int64_t cout = weight.size(-1);
  if (fuse_bn_stats() && batch_stats && !has_bias && cout <= 1024 && features.is_cuda()) {
    c10::DeviceGuard guard(features.device());
    StatRing& ring = stat_ring(features, cur_stream(features));
    const int cur = ring.cur, other = 1 - cur;
    at::Tensor y = SparseConvFn::apply(features, weight, tab_f, flip_f, tab_b, flip_b, n_out, centre, pairs, pair_num, side_src,
                                       c10::optional<at::Tensor>(ring.buf[cur]), perm_b);
    ring.dirty[cur] = static_cast<int64_t>(fv2p_sparse_conv_stat_slots()) * 2 * cout;
    at::Tensor out = BnReluFn::apply(y, bn_weight, bn_bias, running_mean, running_var, num_batches_tracked, training, momentum, eps, relu,
                                     c10::optional<at::Tensor>(ring.buf[cur]), c10::optional<at::Tensor>(ring.buf[other]), ring.dirty[other]);
    ring.dirty[other] = 0;
    ring.cur = other;
    return out;
  }
  at::Tensor y = SparseConvFn::apply(features, weight, tab_f, flip_f, tab_b, flip_b, n_out, centre, pairs, pair_num, side_src, c10::optional<at::Tensor>(),
                                     perm_b);
  if (has_bias) y = y + *conv_bias;
  return BnReluFn::apply(y, bn_weight, bn_bias, running_mean, running_var, num_batches_tracked, training, momentum, eps, relu,
                         c10::optional<at::Tensor>(), c10::optional<at::Tensor>(), 0);
}


// ---- input-pipeline entry points: one GIL-free call each ----------------------------------------------------------------
// A pipeline thread that prepares the next batch shares the interpreter lock with the training thread.  Issued from
// Python, a batch's voxelisation and rulebook chain is ~120 small calls, i.e. ~120 lock hand-offs per step with the
// training thread; these two functions do the same work (same library calls, same order) inside C++ with the lock
// released, allocating their outputs with at::empty.

// points_to_voxel_batch(mean_vfe=True): one stream per cloud, one host synchronisation for the voxel counts, then
// MeanVFE + collate (voxel_mean_collate) -> (features [sum M, ndim], coords [sum M, 4] int32)
std::vector<at::Tensor> voxelize_batch_mean(const std::vector<at::Tensor>& clouds, std::vector<double> voxel_size, std::vector<double> range_lo,
                                            std::vector<int64_t> grid, int64_t max_points, int64_t max_voxels, bool cloud_streams) {
  // cloud_streams = false keeps every launch on the calling stream: the choice of a caller that is itself a side stream
  // (streams share a few hardware queues, and a cloud stream that lands in the training stream's queue waits behind it)
  TORCH_CHECK(!clouds.empty() && voxel_size.size() == 3 && range_lo.size() == 3 && grid.size() == 3, "voxelize_batch_mean: bad arguments");
  const at::Tensor& first = clouds[0];
  c10::DeviceGuard guard(first.device());
  const int dev = first.device().index();
  const c10::hip::HIPStream main = c10::hip::getCurrentHIPStream(dev);
  const float vs[3] = {static_cast<float>(voxel_size[0]), static_cast<float>(voxel_size[1]), static_cast<float>(voxel_size[2])};
  const float lo[3] = {static_cast<float>(range_lo[0]), static_cast<float>(range_lo[1]), static_cast<float>(range_lo[2])};
  const int gr[3] = {static_cast<int>(grid[0]), static_cast<int>(grid[1]), static_cast<int>(grid[2])};
  const size_t nb = clouds.size();
  static std::mutex mu;
  static std::map<std::pair<int, void*>, std::vector<c10::hip::HIPStream>> pools;   // cloud streams per calling stream
  static std::map<std::pair<int, void*>, std::pair<hipEvent_t, std::vector<hipEvent_t>>> events;
  std::vector<c10::hip::HIPStream>* pool;
  std::pair<hipEvent_t, std::vector<hipEvent_t>>* evs;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_pair(dev, static_cast<void*>(main.stream()));
    pool = &pools[key];
    evs = &events[key];
    while (pool->size() < nb) pool->push_back(c10::hip::getStreamFromPool(false, static_cast<c10::DeviceIndex>(dev)));
    if (!evs->first) TORCH_CHECK(hipEventCreateWithFlags(&evs->first, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
    while (evs->second.size() < nb) {
      hipEvent_t e;
      TORCH_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
      evs->second.push_back(e);
    }
  }
  // contiguous copies (if any) are made on the calling stream BEFORE the fork event and live until the function returns: the
  // per-cloud streams read them behind that event, and the calling stream is joined with them again below
  std::vector<at::Tensor> dense_clouds(nb);
  for (size_t b = 0; b < nb; ++b) dense_clouds[b] = clouds[b].contiguous();
  if (cloud_streams) TORCH_CHECK(hipEventRecord(evs->first, main.stream()) == hipSuccess, "hipEventRecord failed");
  std::vector<at::Tensor> vox(nb), coors(nb), num(nb);
  at::Tensor counts = at::empty({static_cast<int64_t>(nb)}, first.options().dtype(at::kInt));
  for (size_t b = 0; b < nb; ++b) {
    const at::Tensor& pts = dense_clouds[b];
    require_f32_cuda(pts, "points");
    const int64_t n = pts.size(0), ndim = pts.size(1);
    // per-cloud outputs are allocated on the calling stream and handed to the cloud's stream behind the fork event
    vox[b] = at::empty({max_voxels, max_points, ndim}, pts.options());
    coors[b] = at::empty({max_voxels, 3}, pts.options().dtype(at::kInt));
    num[b] = at::empty({max_voxels}, pts.options().dtype(at::kInt));
    hipStream_t st = cloud_streams ? (*pool)[b].stream() : main.stream();
    if (cloud_streams) TORCH_CHECK(hipStreamWaitEvent(st, evs->first, 0) == hipSuccess, "hipStreamWaitEvent failed");
    at::Tensor ws;
    {
      c10::hip::HIPStreamGuard sg(cloud_streams ? (*pool)[b] : main);
      ws = workspace(fv2p_points_to_voxel_ws_bytes(n, static_cast<int>(max_voxels)), pts, static_cast<void*>(st));
    }
    check(fv2p_points_to_voxel(pts.data_ptr<float>(), n, static_cast<int>(ndim), vs, lo, gr, static_cast<int>(max_points),
                               static_cast<int>(max_voxels), vox[b].data_ptr<float>(), coors[b].data_ptr<int>(), num[b].data_ptr<int>(),
                               counts.data_ptr<int>() + b, ws.data_ptr(), static_cast<size_t>(ws.numel()), static_cast<void*>(st)),
          "fv2p_points_to_voxel");
    if (cloud_streams) {
      TORCH_CHECK(hipEventRecord(evs->second[b], st) == hipSuccess, "hipEventRecord failed");
      TORCH_CHECK(hipStreamWaitEvent(main.stream(), evs->second[b], 0) == hipSuccess, "hipStreamWaitEvent failed");
    }
  }
  const at::Tensor counts_host = counts.cpu();   // the one synchronisation of the batch
  const int* ch = counts_host.data_ptr<int>();
  int64_t total = 0;
  std::vector<int64_t> m(nb);
  for (size_t b = 0; b < nb; ++b) { m[b] = std::min<int64_t>(ch[b], max_voxels); total += m[b]; }
  const int64_t ndim = clouds[0].size(1);
  at::Tensor feats = at::empty({total, ndim}, first.options());
  at::Tensor coords = at::empty({total, 4}, first.options().dtype(at::kInt));
  int64_t off = 0;
  for (size_t b = 0; b < nb; ++b) {
    if (m[b] > 0)
      check(fv2p_voxel_mean_collate(vox[b].data_ptr<float>(), coors[b].data_ptr<int>(), num[b].data_ptr<int>(), counts.data_ptr<int>() + b,
                                    static_cast<int>(m[b]), static_cast<int>(max_points), static_cast<int>(ndim), static_cast<int>(b),
                                    feats.data_ptr<float>() + off * ndim, coords.data_ptr<int>() + off * 4, static_cast<void*>(main.stream())),
            "fv2p_voxel_mean_collate");
    off += m[b];
  }
  return {feats, coords};
}

// One rulebook of a chain: geometry as the Python layer computes it (3-D, (z, y, x) order).
struct RulebookSpec {
  int64_t src;   // index of the entry whose output rows are this entry's input rows, -1 = the root coordinates
  std::vector<int64_t> in_shape, out_shape, ksize, stride, padding, dilation;
  bool subm, transpose, symmetric, want_pairs;
};
// -> per entry [outids (or empty for subm), tab_in, tab_out (empty when symmetric), pairs (or empty), pair_num (or empty)]
std::vector<std::vector<at::Tensor>> build_rulebook_chain(const at::Tensor& root, int64_t batch,
                                                          const std::vector<std::tuple<int64_t, std::vector<int64_t>, std::vector<int64_t>, std::vector<int64_t>,
                                                                                       std::vector<int64_t>, std::vector<int64_t>, std::vector<int64_t>, bool, bool, bool, bool>>& specs) {
  TORCH_CHECK(root.is_cuda() && root.scalar_type() == at::kInt && root.dim() == 2 && root.size(1) == 4 && root.is_contiguous(),
              "build_rulebook_chain: root indices must be a contiguous CUDA int32 [N, 4] tensor");
  c10::DeviceGuard guard(root.device());
  void* stream = cur_stream(root);
  std::vector<std::vector<at::Tensor>> out;
  out.reserve(specs.size());
  auto arr3 = [](const std::vector<int64_t>& v, int (&a)[3]) {
    TORCH_CHECK(v.size() == 3, "build_rulebook_chain: 3-D geometry expected");
    for (int i = 0; i < 3; ++i) a[i] = static_cast<int>(v[i]);
  };
  for (const auto& sp : specs) {
    const int64_t src = std::get<0>(sp);
    int in_shape[3], out_shape[3], ksize[3], stride[3], padding[3], dilation[3];
    arr3(std::get<1>(sp), in_shape); arr3(std::get<2>(sp), out_shape); arr3(std::get<3>(sp), ksize);
    arr3(std::get<4>(sp), stride); arr3(std::get<5>(sp), padding); arr3(std::get<6>(sp), dilation);
    const bool subm = std::get<7>(sp), transpose = std::get<8>(sp), symmetric = std::get<9>(sp), want_pairs = std::get<10>(sp);
    TORCH_CHECK(src < static_cast<int64_t>(out.size()), "build_rulebook_chain: source entry comes later in the chain");
    const at::Tensor ind = src < 0 ? root : out[src][0];
    TORCH_CHECK(ind.defined(), "build_rulebook_chain: source entry has no output rows of its own (submanifold)");
    const int64_t n_in = ind.size(0);
    const int kvol = ksize[0] * ksize[1] * ksize[2];
    at::Tensor ws = workspace(fv2p_rulebook_ws_bytes_grid(n_in, static_cast<int>(batch), out_shape, ksize, stride, dilation, subm, transpose), root, stream);
    int64_t n_out = 0;
    check(fv2p_rulebook_begin(ind.data_ptr<int>(), n_in, static_cast<int>(batch), in_shape, out_shape, ksize, stride, padding, dilation, subm,
                              transpose, &n_out, ws.data_ptr(), static_cast<size_t>(ws.numel()), stream),
          "fv2p_rulebook_begin");
    const auto iopt = root.options();
    // tables with room for the conv kernels' tiling plan behind them (fv2p_conv_plan_ints; built on first use, ops.Rulebook._plan)
    auto table = [&](int64_t rows) {
      const int64_t cells = static_cast<int64_t>(kvol) * rows;
      return at::empty({cells + fv2p_conv_plan_ints(rows)}, iopt).narrow(0, 0, cells).view({static_cast<int64_t>(kvol), rows});
    };
    at::Tensor tab_in = table(n_in), tab_out, outids;
    if (subm) {
      if (!symmetric) tab_out = table(n_out);
    } else {
      outids = at::empty({n_out, 4}, iopt);
      tab_out = table(n_out);
    }
    check(fv2p_rulebook_finish(ind.data_ptr<int>(), n_in, static_cast<int>(batch), in_shape, out_shape, ksize, stride, padding, dilation, subm,
                               transpose, n_out, outids.defined() ? outids.data_ptr<int>() : nullptr, tab_in.data_ptr<int>(),
                               tab_out.defined() ? tab_out.data_ptr<int>() : nullptr, nullptr, ws.data_ptr(), static_cast<size_t>(ws.numel()), stream),
          "fv2p_rulebook_finish");
    at::Tensor pairs, pnum;
    if (want_pairs && n_in > 0) {
      pairs = at::empty({kvol, 2, n_in}, iopt);
      pnum = at::empty({kvol}, iopt);
      at::Tensor pws = workspace(fv2p_rulebook_pairs_ws_bytes(n_in, kvol), root, stream);
      check(fv2p_rulebook_pairs(tab_in.data_ptr<int>(), n_in, kvol, 0, pairs.data_ptr<int>(), pnum.data_ptr<int>(), pws.data_ptr(),
                                static_cast<size_t>(pws.numel()), stream),
            "fv2p_rulebook_pairs");
    }
    // strided conv: input rows grouped by parity class, the tile order of its backward-data conv (fv2p_rulebook_class_perm)
    at::Tensor perm;
    if (!subm && !transpose && n_in > 0 && stride[0] * stride[1] * stride[2] > 1 && stride[0] <= 2 && stride[1] <= 2 && stride[2] <= 2 &&
        dilation[0] == 1 && dilation[1] == 1 && dilation[2] == 1) {
      perm = at::empty({n_in}, iopt);
      at::Tensor cws = workspace(fv2p_rulebook_class_perm_ws_bytes(n_in), root, stream);
      check(fv2p_rulebook_class_perm(ind.data_ptr<int>(), n_in, stride, padding, perm.data_ptr<int>(), cws.data_ptr(), static_cast<size_t>(cws.numel()),
                                     stream),
            "fv2p_rulebook_class_perm");
    }
    out.push_back({subm ? at::Tensor() : outids, tab_in, tab_out, pairs, pnum, perm});
    if (subm) out.back()[0] = at::Tensor();
  }
  return out;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "compiled autograd binding of libfv2p_ops (sparse conv, BatchNorm1d+ReLU)";
  m.def("abi_version", []() { return fv2p_abi_version(); });
  m.def("set_bn_epilogue", &set_bn_epilogue, "BatchNorm sums from the conv epilogues (default) or from BatchNorm's own reduce pass");
  m.def("gate_weights", &gate_weights, "aliases of the conv weights whose gradients are joined from the side stream at the end of backward");
  m.def("sparse_conv", &sparse_conv, "fused sparse convolution with autograd (tables from a Rulebook)");
  m.def("batch_norm_relu", &batch_norm_relu, "BatchNorm1d (+ReLU) on [N, C] with autograd");
  m.def("voxelize_batch_mean", &voxelize_batch_mean, py::arg("clouds"), py::arg("voxel_size"), py::arg("range_lo"), py::arg("grid"),
        py::arg("max_points"), py::arg("max_voxels"), py::arg("cloud_streams") = true, py::call_guard<py::gil_scoped_release>(),
        "voxelise a batch of clouds + MeanVFE + collate, without the GIL");
  m.def("build_rulebook_chain", &build_rulebook_chain, py::call_guard<py::gil_scoped_release>(),
        "build a chain of rulebooks (and pair lists) from root coordinates, without the GIL");
  m.def("sparse_conv_bn_relu", &sparse_conv_bn_relu, "sparse conv -> BatchNorm1d (-> ReLU) with autograd, one call per backbone block");
}
