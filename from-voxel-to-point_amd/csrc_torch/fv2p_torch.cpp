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


// two zeroed device words per (device, stream) for the one-launch BatchNorm passes (fv2p_batchnorm_forward_one / _backward_one);
// defined with the round-6 state further down
unsigned* one_counters(const at::Tensor& like, void* stream);
unsigned* wide_counters(const at::Tensor& like, void* stream);
static int g_bn_wide = -1;   // FV2P_BN_WIDE=0 / set_bn_wide(false): reduce on <= 64 workgroups + apply that folds (rounds 2 - 5)
bool bn_wide() {
  if (g_bn_wide < 0) { const char* e = std::getenv("FV2P_BN_WIDE"); g_bn_wide = !(e && e[0] == '0'); }
  return g_bn_wide != 0;
}
void set_bn_wide(bool on) { g_bn_wide = on ? 1 : 0; }
static int g_bn_one = -1;   // FV2P_BN_ONE=0 / set_bn_one(false): the two-launch passes (tests compare the two)
bool bn_one() {
  if (g_bn_one < 0) { const char* e = std::getenv("FV2P_BN_ONE"); g_bn_one = !(e && e[0] == '0'); }
  return g_bn_one != 0;
}
void set_bn_one(bool on) { g_bn_one = on ? 1 : 0; }

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
      } else if (bn_one() && fv2p_batchnorm_one_pays(n, static_cast<int>(c), 0)) {   // reduce, grid barrier, apply: one launch
        at::Tensor ws = workspace(fv2p_batchnorm_one_ws_bytes(static_cast<int>(c)), x, stream);
        check(fv2p_batchnorm_forward_one(x.data_ptr<float>(), n, static_cast<int>(c), static_cast<float>(eps), static_cast<float>(momentum), gamma, beta,
                                         relu ? 1 : 0, track ? running_mean->data_ptr<float>() : nullptr, track ? running_var->data_ptr<float>() : nullptr,
                                         nbt, mean.data_ptr<float>(), invstd.data_ptr<float>(), nullptr, y.data_ptr<float>(), ws.data_ptr(),
                                         static_cast<size_t>(ws.numel()), one_counters(x, stream), stream),
              "fv2p_batchnorm_forward_one");
      } else if (bn_wide()) {   // large tensors: reduce on up to 512 workgroups, finalised by that launch; apply reads mean / invstd
        at::Tensor ws = workspace(fv2p_batchnorm_wide_ws_bytes(static_cast<int>(c)), x, stream);
        check(fv2p_batchnorm_forward_wide(x.data_ptr<float>(), n, static_cast<int>(c), static_cast<float>(eps), static_cast<float>(momentum), gamma, beta,
                                          relu ? 1 : 0, track ? running_mean->data_ptr<float>() : nullptr, track ? running_var->data_ptr<float>() : nullptr,
                                          nbt, mean.data_ptr<float>(), invstd.data_ptr<float>(), nullptr, y.data_ptr<float>(), ws.data_ptr(),
                                          static_cast<size_t>(ws.numel()), wide_counters(x, stream), stream),
              "fv2p_batchnorm_forward_wide");
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
    if (bn_one() && fv2p_batchnorm_one_pays(n, static_cast<int>(c), 1)) {
      at::Tensor ws = workspace(fv2p_batchnorm_one_ws_bytes(static_cast<int>(c)), x, stream);
      check(fv2p_batchnorm_backward_one(x.data_ptr<float>(), dy.data_ptr<float>(), n, static_cast<int>(c), mean.data_ptr<float>(), invstd.data_ptr<float>(),
                                        weight.defined() ? weight.data_ptr<float>() : nullptr, bias.defined() ? bias.data_ptr<float>() : nullptr,
                                        ctx->saved_data["relu"].toBool() ? 1 : 0, ctx->saved_data["batch_stats"].toBool() ? 1 : 0, nullptr, dx.data_ptr<float>(),
                                        nullptr, dpar[0].data_ptr<float>(), dpar[1].data_ptr<float>(), ws.data_ptr(), static_cast<size_t>(ws.numel()),
                                        one_counters(x, stream), stream),
            "fv2p_batchnorm_backward_one");
    } else if (bn_wide()) {
      at::Tensor ws = workspace(fv2p_batchnorm_wide_ws_bytes(static_cast<int>(c)), x, stream);
      check(fv2p_batchnorm_backward_wide(x.data_ptr<float>(), dy.data_ptr<float>(), n, static_cast<int>(c), mean.data_ptr<float>(), invstd.data_ptr<float>(),
                                         weight.defined() ? weight.data_ptr<float>() : nullptr, bias.defined() ? bias.data_ptr<float>() : nullptr,
                                         ctx->saved_data["relu"].toBool() ? 1 : 0, ctx->saved_data["batch_stats"].toBool() ? 1 : 0, nullptr, dx.data_ptr<float>(),
                                         nullptr, dpar[0].data_ptr<float>(), dpar[1].data_ptr<float>(), ws.data_ptr(), static_cast<size_t>(ws.numel()),
                                         wide_counters(x, stream), stream),
            "fv2p_batchnorm_backward_wide");
    } else {
    at::Tensor ws = workspace(fv2p_batchnorm_ws_bytes(n, static_cast<int>(c)), x, stream);
    check(fv2p_batchnorm_backward(x.data_ptr<float>(), dy.data_ptr<float>(), n, static_cast<int>(c), mean.data_ptr<float>(), invstd.data_ptr<float>(),
                                  weight.defined() ? weight.data_ptr<float>() : nullptr, bias.defined() ? bias.data_ptr<float>() : nullptr,
                                  ctx->saved_data["relu"].toBool() ? 1 : 0, ctx->saved_data["batch_stats"].toBool() ? 1 : 0, dx.data_ptr<float>(),
                                  dpar[0].data_ptr<float>(), dpar[1].data_ptr<float>(), ws.data_ptr(), static_cast<size_t>(ws.numel()), stream),
          "fv2p_batchnorm_backward");
    }
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


// ======================================================================================================================
// Round 6: statistics finalised by the conv launch itself, BatchNorm (+ReLU) folded into the consumer conv's gather, the
// residual tail in the normalisation's pass (include/fv2p_ops.h: fv2p_sparse_conv_rows_bnfin and friends).
//
//   conv_fin(features, W, ..., [bias], [BatchNorm that follows], [BatchNorm (+ReLU) the SOURCE rows pass through on the gather])
//       -> (y, saved [2, Cout] = mean / invstd of y's BatchNorm, final when the conv launch ends)
//   bn_apply(y, saved, gamma, beta, relu, [residual]) -> relu?((y - mean) * invstd * gamma + beta [+ residual])      ONE launch, no fold
//
// A post_act_block (conv -> BN -> ReLU, spconv_backbone.py:8-27) is conv_fin + bn_apply: 2 launches forward (as before, but the apply no
// longer folds 64 slots in each of its workgroups) and, where the block's output feeds one conv only, 1 + 1 launches backward.
// A SparseBasicBlock (spconv_backbone.py:32-68) is
//       y1, s1 = conv_fin(x, W1, b1, bn1)            y2, s2 = conv_fin(y1, W2, b2, bn2, source = (s1, bn1, relu))
//       out    = bn_apply(y2, s2, bn2, relu, residual = x)
// 3 launches forward (was 8: two convs, two bias adds, two reduce + two apply passes, add, relu) - relu(bn1(y1)) is never written.
// The conv bias: it feeds a train-mode BatchNorm, which removes every per-column constant - its gradient is identically zero in exact
// arithmetic (torch returns the rounding noise of a column sum of dx).  The fused nodes add it in the conv epilogue and return zeros.
struct FinState {
  at::Tensor counter;   // the completion counters of the finalising launches: forward | backward | the one-launch BatchNorms (zero between launches)
};
FinState& fin_state(const at::Tensor& like, void* stream) {
  static std::mutex mu;
  static std::map<std::pair<int, void*>, FinState> pool;
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_pair(static_cast<int>(like.device().index()), stream);
  auto it = pool.find(key);
  if (it == pool.end()) {
    FinState f;
    f.counter = at::zeros({2 * static_cast<int64_t>(fv2p_sparse_conv_fin_counter_words()) + 4 + fv2p_batchnorm_wide_counter_words(1024)},
                          like.options().dtype(at::kInt));   // conv forward | conv backward | 4 words of the one-launch BatchNorms | the wide BatchNorm reduce
    it = pool.emplace(key, std::move(f)).first;
  }
  return it->second;
}
unsigned* one_counters(const at::Tensor& like, void* stream) {
  return reinterpret_cast<unsigned*>(fin_state(like, stream).counter.data_ptr<int>()) + 2 * fv2p_sparse_conv_fin_counter_words();
}
unsigned* wide_counters(const at::Tensor& like, void* stream) {
  return reinterpret_cast<unsigned*>(fin_state(like, stream).counter.data_ptr<int>()) + 2 * fv2p_sparse_conv_fin_counter_words() + 4;
}
static int g_bn_fold = -1;   // FV2P_BN_FOLD=0 / set_bn_fold(false): the Python layer keeps the round-5 arrangement (tests compare the two)
bool bn_fold() {
  if (g_bn_fold < 0) { const char* e = std::getenv("FV2P_BN_FOLD"); g_bn_fold = !(e && e[0] == '0'); }
  return g_bn_fold != 0;
}
void set_bn_fold(bool on) { g_bn_fold = on ? 1 : 0; }

const float* fptr(const at::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
at::Tensor opt(const c10::optional<at::Tensor>& t) { return (t.has_value() && t->defined()) ? *t : at::Tensor(); }

// ---- bn_apply: y = relu?((x - mean) * invstd * gamma + beta [+ residual]) with finalised mean / invstd ------------------------------
struct BnApplyFn : public torch::autograd::Function<BnApplyFn> {
  static at::Tensor forward(AutogradContext* ctx, const at::Tensor& x_, const at::Tensor& saved_, const c10::optional<at::Tensor>& weight,
                            const c10::optional<at::Tensor>& bias, bool relu, const c10::optional<at::Tensor>& residual_, bool batch_stats) {
    require_f32_cuda(x_, "input");
    const at::Tensor x = x_.contiguous(), saved = saved_.contiguous();
    const at::Tensor residual = opt(residual_).defined() ? opt(residual_).contiguous() : at::Tensor();
    const int64_t n = x.size(0), c = x.size(1);
    TORCH_CHECK(saved.dim() == 2 && saved.size(0) == 2 && saved.size(1) == c, "bn_apply: saved statistics [2, C] expected");
    TORCH_CHECK(!residual.defined() || residual.sizes() == x.sizes(), "bn_apply: residual must have the input's shape");
    c10::DeviceGuard guard(x.device());
    void* stream = cur_stream(x);
    at::Tensor y = at::empty_like(x);
    const at::Tensor w = opt(weight), b = opt(bias);
    check(fv2p_batchnorm_apply_res(x.data_ptr<float>(), n, static_cast<int>(c), saved[0].data_ptr<float>(), saved[1].data_ptr<float>(), fptr(w), fptr(b),
                                   relu ? 1 : 0, fptr(residual), y.data_ptr<float>(), stream),
          "fv2p_batchnorm_apply_res");
    ctx->save_for_backward({x, saved, w, b, residual.defined() ? y : at::Tensor()});
    ctx->saved_data["relu"] = relu;
    ctx->saved_data["batch_stats"] = batch_stats;
    ctx->saved_data["residual"] = residual.defined();
    return y;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto sv = ctx->get_saved_variables();
    const at::Tensor &x = sv[0], &saved = sv[1], &weight = sv[2], &bias = sv[3], &out = sv[4];
    const at::Tensor dy = grads[0].contiguous();
    const int64_t n = x.size(0), c = x.size(1);
    const bool relu = ctx->saved_data["relu"].toBool(), batch_stats = ctx->saved_data["batch_stats"].toBool();
    c10::DeviceGuard guard(x.device());
    void* stream = cur_stream(x);
    at::Tensor dx = at::empty_like(x);
    const float* mean = saved[0].data_ptr<float>();
    const float* invstd = saved[1].data_ptr<float>();
    if (ctx->saved_data["residual"].toBool()) {   // out = relu(bn(x) + identity): mask from out, dz is the identity branch's gradient
      at::Tensor dz = at::empty_like(x), dpar = at::empty({2, c}, x.options());
      at::Tensor ws = workspace(fv2p_batchnorm_ws_bytes(n, static_cast<int>(c)), x, stream);
      if (relu && bn_one() && fv2p_batchnorm_one_pays(n, static_cast<int>(c), 1)) {
        at::Tensor ws1 = workspace(fv2p_batchnorm_one_ws_bytes(static_cast<int>(c)), x, stream);
        check(fv2p_batchnorm_backward_one(x.data_ptr<float>(), dy.data_ptr<float>(), n, static_cast<int>(c), mean, invstd, fptr(weight), fptr(bias), 1,
                                          batch_stats ? 1 : 0, out.data_ptr<float>(), dx.data_ptr<float>(), dz.data_ptr<float>(), dpar[0].data_ptr<float>(),
                                          dpar[1].data_ptr<float>(), ws1.data_ptr(), static_cast<size_t>(ws1.numel()), one_counters(x, stream), stream),
              "fv2p_batchnorm_backward_one");
      } else if (relu && bn_wide()) {
        at::Tensor ws2 = workspace(fv2p_batchnorm_wide_ws_bytes(static_cast<int>(c)), x, stream);
        check(fv2p_batchnorm_backward_wide(x.data_ptr<float>(), dy.data_ptr<float>(), n, static_cast<int>(c), mean, invstd, fptr(weight), fptr(bias), 1,
                                           batch_stats ? 1 : 0, out.data_ptr<float>(), dx.data_ptr<float>(), dz.data_ptr<float>(), dpar[0].data_ptr<float>(),
                                           dpar[1].data_ptr<float>(), ws2.data_ptr(), static_cast<size_t>(ws2.numel()), wide_counters(x, stream), stream),
              "fv2p_batchnorm_backward_wide");
      } else if (relu) {
        check(fv2p_batchnorm_backward_res(x.data_ptr<float>(), out.data_ptr<float>(), dy.data_ptr<float>(), n, static_cast<int>(c), mean, invstd, fptr(weight),
                                          fptr(bias), batch_stats ? 1 : 0, dx.data_ptr<float>(), dz.data_ptr<float>(), dpar[0].data_ptr<float>(),
                                          dpar[1].data_ptr<float>(), ws.data_ptr(), static_cast<size_t>(ws.numel()), stream),
              "fv2p_batchnorm_backward_res");
      } else {   // no ReLU after the sum: the identity branch receives dy itself
        check(fv2p_batchnorm_backward(x.data_ptr<float>(), dy.data_ptr<float>(), n, static_cast<int>(c), mean, invstd, fptr(weight), fptr(bias), 0,
                                      batch_stats ? 1 : 0, dx.data_ptr<float>(), dpar[0].data_ptr<float>(), dpar[1].data_ptr<float>(), ws.data_ptr(),
                                      static_cast<size_t>(ws.numel()), stream),
              "fv2p_batchnorm_backward");
        dz = dy;
      }
      return {dx, at::Tensor(), weight.defined() ? dpar[0] : at::Tensor(), bias.defined() ? dpar[1] : at::Tensor(), at::Tensor(), dz, at::Tensor()};
    }
    // sums finalised by the backward-data conv that produced exactly this dy (ConvFinFn::backward): one launch, nothing folded
    auto it = ctx->saved_data.find("fin_coef");
    if (it != ctx->saved_data.end()) {
      const at::Tensor coef = it->second.toTensor();   // [4, c]: dgamma, dbeta, c1, c2
      const bool mine = ctx->saved_data["fin_dy"].toInt() == reinterpret_cast<int64_t>(dy.data_ptr()) &&
                        ctx->saved_data["fin_ver"].toInt() == static_cast<int64_t>(dy._version());
      ctx->saved_data.erase("fin_coef");
      if (mine) {
        check(fv2p_batchnorm_backward_fin(x.data_ptr<float>(), dy.data_ptr<float>(), n, static_cast<int>(c), mean, invstd, fptr(weight), fptr(bias),
                                          relu ? 1 : 0, coef[2].data_ptr<float>(), dx.data_ptr<float>(), stream),
              "fv2p_batchnorm_backward_fin");
        return {dx, at::Tensor(), weight.defined() ? coef[0] : at::Tensor(), bias.defined() ? coef[1] : at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
      }
    }
    at::Tensor dpar = at::empty({2, c}, x.options());
    if (bn_one() && fv2p_batchnorm_one_pays(n, static_cast<int>(c), 1)) {
      at::Tensor ws1 = workspace(fv2p_batchnorm_one_ws_bytes(static_cast<int>(c)), x, stream);
      check(fv2p_batchnorm_backward_one(x.data_ptr<float>(), dy.data_ptr<float>(), n, static_cast<int>(c), mean, invstd, fptr(weight), fptr(bias), relu ? 1 : 0,
                                        batch_stats ? 1 : 0, nullptr, dx.data_ptr<float>(), nullptr, dpar[0].data_ptr<float>(), dpar[1].data_ptr<float>(),
                                        ws1.data_ptr(), static_cast<size_t>(ws1.numel()), one_counters(x, stream), stream),
            "fv2p_batchnorm_backward_one");
    } else if (bn_wide()) {
      at::Tensor ws2 = workspace(fv2p_batchnorm_wide_ws_bytes(static_cast<int>(c)), x, stream);
      check(fv2p_batchnorm_backward_wide(x.data_ptr<float>(), dy.data_ptr<float>(), n, static_cast<int>(c), mean, invstd, fptr(weight), fptr(bias), relu ? 1 : 0,
                                         batch_stats ? 1 : 0, nullptr, dx.data_ptr<float>(), nullptr, dpar[0].data_ptr<float>(), dpar[1].data_ptr<float>(),
                                         ws2.data_ptr(), static_cast<size_t>(ws2.numel()), wide_counters(x, stream), stream),
            "fv2p_batchnorm_backward_wide");
    } else {
      at::Tensor ws = workspace(fv2p_batchnorm_ws_bytes(n, static_cast<int>(c)), x, stream);
      check(fv2p_batchnorm_backward(x.data_ptr<float>(), dy.data_ptr<float>(), n, static_cast<int>(c), mean, invstd, fptr(weight), fptr(bias), relu ? 1 : 0,
                                    batch_stats ? 1 : 0, dx.data_ptr<float>(), dpar[0].data_ptr<float>(), dpar[1].data_ptr<float>(), ws.data_ptr(),
                                    static_cast<size_t>(ws.numel()), stream),
            "fv2p_batchnorm_backward");
    }
    return {dx, at::Tensor(), weight.defined() ? dpar[0] : at::Tensor(), bias.defined() ? dpar[1] : at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
  }
};

// ---- conv_fin ----------------------------------------------------------------------------------------------------------------------
struct ConvFinFn : public torch::autograd::Function<ConvFinFn> {
  static variable_list forward(AutogradContext* ctx, const at::Tensor& features_, const at::Tensor& weight_, const at::Tensor& tab_f, int64_t flip_f,
                               const at::Tensor& tab_b, int64_t flip_b, int64_t n_out, int64_t centre, const c10::optional<at::Tensor>& pairs,
                               const c10::optional<at::Tensor>& pair_num, int64_t side_src, const c10::optional<at::Tensor>& perm_b,
                               const c10::optional<at::Tensor>& conv_bias,
                               // the BatchNorm that follows (its statistics): none of the three tensors defined + !training = no statistics
                               bool want_stats, const c10::optional<at::Tensor>& running_mean, const c10::optional<at::Tensor>& running_var,
                               const c10::optional<at::Tensor>& num_batches_tracked, bool training, double momentum, double eps,
                               // the BatchNorm (+ReLU) the source rows pass through on the gather
                               const c10::optional<at::Tensor>& pre_saved_, const c10::optional<at::Tensor>& pre_gamma_,
                               const c10::optional<at::Tensor>& pre_beta_, bool pre_relu, bool pre_batch_stats) {
    require_f32_cuda(features_, "features");
    require_f32_cuda(weight_, "weight");
    const at::Tensor features = features_.contiguous(), weight = weight_.contiguous();
    const int64_t cin = weight.size(-2), cout = weight.size(-1);
    const int64_t kvol = weight.numel() / (cin * cout);
    TORCH_CHECK(features.dim() == 2 && features.size(1) == cin, "features [N, Cin] expected");
    c10::DeviceGuard guard(features.device());
    void* stream = cur_stream(features);
    const at::Tensor cb = opt(conv_bias), rm = opt(running_mean), rv = opt(running_var), nbt = opt(num_batches_tracked);
    const at::Tensor pre_saved = opt(pre_saved_).defined() ? opt(pre_saved_).contiguous() : at::Tensor();
    const at::Tensor pre_gamma = opt(pre_gamma_), pre_beta = opt(pre_beta_);
    const bool batch_stats = want_stats && (training || !rm.defined());
    at::Tensor out = at::empty({n_out, cout}, features.options());
    at::Tensor saved;
    FinState& fs = fin_state(features, stream);
    if (batch_stats) saved = at::empty({2, cout}, features.options());
    else if (want_stats) saved = at::stack({rm, at::rsqrt(rv + eps)});
    const bool track = batch_stats && training && rm.defined();
    at::Tensor fin_ws;   // the tiles' rows of column sums + the group slots (written and read inside the launch)
    if (batch_stats) fin_ws = workspace(fv2p_sparse_conv_fin_ws_bytes(n_out, static_cast<int>(cout)), features, stream);
    check(fv2p_sparse_conv_rows_bnfin(features.data_ptr<float>(), features.size(0), static_cast<int>(cin), weight.data_ptr<float>(), static_cast<int>(kvol),
                                      tab_f.data_ptr<int>(), n_out, static_cast<int>(cout), static_cast<int>(flip_f), 0, fptr(cb), out.data_ptr<float>(),
                                      batch_stats ? reinterpret_cast<double*>(fin_ws.data_ptr()) : nullptr, reinterpret_cast<unsigned*>(fs.counter.data_ptr<int>()),
                                      static_cast<float>(eps), static_cast<float>(momentum), track ? rm.data_ptr<float>() : nullptr,
                                      track ? rv.data_ptr<float>() : nullptr, (track && nbt.defined()) ? nbt.data_ptr<int64_t>() : nullptr,
                                      batch_stats ? saved[0].data_ptr<float>() : nullptr, batch_stats ? saved[1].data_ptr<float>() : nullptr,
                                      pre_saved.defined() ? pre_saved[0].data_ptr<float>() : nullptr, pre_saved.defined() ? pre_saved[1].data_ptr<float>() : nullptr,
                                      pre_saved.defined() ? fptr(pre_gamma) : nullptr, pre_saved.defined() ? fptr(pre_beta) : nullptr, pre_relu ? 1 : 0, stream),
          "fv2p_sparse_conv_rows_bnfin");
    const bool have_pairs = pairs.has_value() && pairs->defined() && pair_num.has_value() && pair_num->defined();
    ctx->save_for_backward({features, weight, tab_f, tab_b, have_pairs ? *pairs : at::Tensor(), have_pairs ? *pair_num : at::Tensor(), opt(perm_b),
                            pre_saved, pre_gamma, pre_beta});
    ctx->saved_data["side_src"] = side_src;
    ctx->saved_data["flip_f"] = flip_f;
    ctx->saved_data["flip_b"] = flip_b;
    ctx->saved_data["centre"] = centre;
    ctx->saved_data["gated"] = is_gated(weight_);
    ctx->saved_data["has_bias"] = cb.defined();
    ctx->saved_data["pre_relu"] = pre_relu;
    ctx->saved_data["pre_batch_stats"] = pre_batch_stats;
    // features straight out of a bn_apply (no residual): the backward-data conv can take that layer's backward sums and finalise them
    auto* bn_node = (!pre_saved.defined() && features_.grad_fn()) ? dynamic_cast<torch::autograd::CppNode<BnApplyFn>*>(features_.grad_fn().get()) : nullptr;
    ctx->saved_data["bn_node"] = reinterpret_cast<int64_t>(bn_node);   // kept alive by this node's edge to it
    if (!saved.defined()) saved = at::empty({0}, features.options());
    ctx->mark_non_differentiable({saved});
    return {out, saved};
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto sv = ctx->get_saved_variables();
    const at::Tensor &features = sv[0], &weight = sv[1], &tab_f = sv[2], &tab_b = sv[3], &pairs = sv[4], &pair_num = sv[5];
    const at::Tensor &pre_saved = sv[7], &pre_gamma = sv[8], &pre_beta = sv[9];
    const int* perm_b = (sv[6].defined() && sv[6].numel() == features.size(0)) ? sv[6].data_ptr<int>() : nullptr;
    const at::Tensor g = grads[0].contiguous();
    const int64_t cin = weight.size(-2), cout = weight.size(-1);
    const int64_t kvol = weight.numel() / (cin * cout);
    const int flip_f = static_cast<int>(ctx->saved_data["flip_f"].toInt()), flip_b = static_cast<int>(ctx->saved_data["flip_b"].toInt());
    const int centre = static_cast<int>(ctx->saved_data["centre"].toInt());
    const bool pre = pre_saved.defined(), pre_relu = ctx->saved_data["pre_relu"].toBool(), pre_bs = ctx->saved_data["pre_batch_stats"].toBool();
    c10::DeviceGuard guard(features.device());
    void* stream = cur_stream(features);
    FinState& fs = fin_state(features, stream);
    at::Tensor din, dw, dgamma_src, dbeta_src;
    const bool need_in = ctx->needs_input_grad(0) || pre;   // (needs_input_grad counts tensor arguments that were defined: only 0 and 1 are stable)
    const bool both = need_in && ctx->needs_input_grad(1);
    const bool overlap = both && wgrad_overlap();
    void* wstream = stream;
    SideStream* side = nullptr;
    if (ctx->needs_input_grad(1)) dw = at::empty_like(weight);
    if (overlap) {
      side = &side_stream(features.device().index());
      TORCH_CHECK(hipEventRecord(side->fork, static_cast<hipStream_t>(stream)) == hipSuccess, "hipEventRecord failed");
      TORCH_CHECK(hipStreamWaitEvent(side->stream.stream(), side->fork, 0) == hipSuccess, "hipStreamWaitEvent failed");
      wstream = static_cast<void*>(side->stream.stream());
    }
    if (need_in) {
      din = at::empty_like(features);
      at::Tensor wt;
      int transpose_w = 1;
      if ((cout == 64 || cout == 128) && cin % 64 == 0 && cin <= 128 && kvol > 1) {   // W_k^T materialised for the K-split tile (SparseConvFn::backward)
        wt = weight.view({kvol, cin, cout}).transpose(1, 2).contiguous();
        transpose_w = 0;
      }
      const float* w_bwd = wt.defined() ? wt.data_ptr<float>() : weight.data_ptr<float>();
      unsigned* counter = reinterpret_cast<unsigned*>(fs.counter.data_ptr<int>()) + fv2p_sparse_conv_fin_counter_words();
      at::Tensor fin_ws = workspace(fv2p_sparse_conv_fin_ws_bytes(features.size(0), static_cast<int>(cin)), features, stream);
      if (pre) {
        // d(relu(bn(y_src))) by the backward-data conv, the BatchNorm's backward sums from its epilogue, finalised by its last workgroup;
        // then the BatchNorm's own backward pass in place: din becomes d(y_src)
        at::Tensor coef = at::empty({4, cin}, features.options());
        check(fv2p_sparse_conv_rows_bnbwd_fin(g.data_ptr<float>(), g.size(0), static_cast<int>(cout), w_bwd, static_cast<int>(kvol), tab_b.data_ptr<int>(),
                                              features.size(0), static_cast<int>(cin), flip_b, transpose_w, din.data_ptr<float>(), features.data_ptr<float>(),
                                              pre_saved[0].data_ptr<float>(), pre_saved[1].data_ptr<float>(), fptr(pre_gamma), fptr(pre_beta), pre_relu ? 1 : 0,
                                              reinterpret_cast<double*>(fin_ws.data_ptr()), counter, pre_bs ? 1 : 0, coef[0].data_ptr<float>(), coef[1].data_ptr<float>(),
                                              coef[2].data_ptr<float>(), perm_b, stream),
              "fv2p_sparse_conv_rows_bnbwd_fin");
        check(fv2p_batchnorm_backward_fin(features.data_ptr<float>(), din.data_ptr<float>(), features.size(0), static_cast<int>(cin),
                                          pre_saved[0].data_ptr<float>(), pre_saved[1].data_ptr<float>(), fptr(pre_gamma), fptr(pre_beta), pre_relu ? 1 : 0,
                                          coef[2].data_ptr<float>(), din.data_ptr<float>(), stream),
              "fv2p_batchnorm_backward_fin");
        dgamma_src = coef[0];
        dbeta_src = coef[1];
      } else {
        auto* bn_node = reinterpret_cast<torch::autograd::CppNode<BnApplyFn>*>(ctx->saved_data["bn_node"].toInt());
        bool fused = false;
        if (bn_node && fuse_bn_stats() && cout <= 128 && cin <= 1024) {
          AutogradContext& bctx = bn_node->ctx_;
          if (!bctx.saved_data["residual"].toBool() && bctx.saved_data.find("fin_coef") == bctx.saved_data.end()) {
            const auto bsaved = bctx.get_saved_variables();   // x, saved, weight, bias, out
            const at::Tensor &bx = bsaved[0], &bstat = bsaved[1], &bw = bsaved[2], &bb = bsaved[3];
            if (bx.defined() && bx.sizes() == din.sizes() && bx.is_contiguous()) {
              at::Tensor coef = at::empty({4, cin}, features.options());
              check(fv2p_sparse_conv_rows_bnbwd_fin(g.data_ptr<float>(), g.size(0), static_cast<int>(cout), w_bwd, static_cast<int>(kvol), tab_b.data_ptr<int>(),
                                                    features.size(0), static_cast<int>(cin), flip_b, transpose_w, din.data_ptr<float>(), bx.data_ptr<float>(),
                                                    bstat[0].data_ptr<float>(), bstat[1].data_ptr<float>(), fptr(bw), fptr(bb),
                                                    bctx.saved_data["relu"].toBool() ? 1 : 0, reinterpret_cast<double*>(fin_ws.data_ptr()), counter,
                                                    bctx.saved_data["batch_stats"].toBool() ? 1 : 0, coef[0].data_ptr<float>(), coef[1].data_ptr<float>(),
                                                    coef[2].data_ptr<float>(), perm_b, stream),
                    "fv2p_sparse_conv_rows_bnbwd_fin");
              bctx.saved_data["fin_coef"] = coef;
              bctx.saved_data["fin_dy"] = reinterpret_cast<int64_t>(din.data_ptr());
              bctx.saved_data["fin_ver"] = static_cast<int64_t>(din._version());
              fused = true;
            }
          }
        }
        if (!fused)
          check(fv2p_sparse_conv_rows_perm(g.data_ptr<float>(), g.size(0), static_cast<int>(cout), w_bwd, static_cast<int>(kvol), tab_b.data_ptr<int>(),
                                           features.size(0), static_cast<int>(cin), flip_b, transpose_w, nullptr, din.data_ptr<float>(), perm_b, stream),
                "fv2p_sparse_conv_rows (backward data)");
      }
    }
    if (ctx->needs_input_grad(1)) {
      c10::optional<c10::hip::HIPStreamGuard> sg;
      if (overlap) sg.emplace(side->stream);
      const float* pm = pre ? pre_saved[0].data_ptr<float>() : nullptr;
      const float* pi = pre ? pre_saved[1].data_ptr<float>() : nullptr;
      if (pairs.defined()) {
        const int64_t plen = pairs.size(2);
        const size_t wsb = fv2p_sparse_conv_wgrad_pairs_ws_bytes(plen, static_cast<int>(cin), static_cast<int>(cout), static_cast<int>(kvol));
        at::Tensor ws = workspace(wsb, features, wstream);
        check(fv2p_sparse_conv_wgrad_pairs_pre(features.data_ptr<float>(), features.size(0), static_cast<int>(cin), g.data_ptr<float>(), g.size(0),
                                               static_cast<int>(cout), pairs.data_ptr<int>(), pair_num.data_ptr<int>(), static_cast<int>(kvol), plen,
                                               static_cast<int>(ctx->saved_data["side_src"].toInt()), dw.data_ptr<float>(), pm, pi, pre ? fptr(pre_gamma) : nullptr,
                                               pre ? fptr(pre_beta) : nullptr, pre_relu ? 1 : 0, ws.data_ptr(), static_cast<size_t>(ws.numel()), wstream),
              "fv2p_sparse_conv_wgrad_pairs_pre");
      } else {
        const size_t wsb = fv2p_sparse_conv_wgrad_ws_bytes(g.size(0), static_cast<int>(cin), static_cast<int>(cout), static_cast<int>(kvol));
        at::Tensor ws = workspace(wsb, features, wstream);
        check(fv2p_sparse_conv_wgrad_pre(features.data_ptr<float>(), features.size(0), static_cast<int>(cin), g.data_ptr<float>(), tab_f.data_ptr<int>(), g.size(0),
                                         static_cast<int>(cout), static_cast<int>(kvol), flip_f, centre, dw.data_ptr<float>(), pm, pi, pre ? fptr(pre_gamma) : nullptr,
                                         pre ? fptr(pre_beta) : nullptr, pre_relu ? 1 : 0, ws.data_ptr(), static_cast<size_t>(ws.numel()), wstream),
              "fv2p_sparse_conv_wgrad_pre");
      }
    }
    if (overlap && ctx->saved_data["gated"].toBool()) {
      std::lock_guard<std::mutex> lock(g_pending_mu);
      auto& parked = g_pending[features.device().index()];
      for (const at::Tensor& t : {features, g, tab_f, pairs, pair_num, dw, pre_saved, pre_gamma, pre_beta})
        if (t.defined()) parked.push_back(t);
    } else if (overlap) {
      TORCH_CHECK(hipEventRecord(side->join, side->stream.stream()) == hipSuccess, "hipEventRecord failed");
      TORCH_CHECK(hipStreamWaitEvent(static_cast<hipStream_t>(stream), side->join, 0) == hipSuccess, "hipStreamWaitEvent failed");
    }
    variable_list out(25);   // one per forward argument: 0 features, 1 weight, 12 conv_bias, 21 pre_gamma, 22 pre_beta
    out[0] = din;
    out[1] = dw;
    if (ctx->saved_data["has_bias"].toBool()) out[12] = at::zeros({cout}, features.options());   // see the section comment
    if (pre) { out[21] = dgamma_src; out[22] = dbeta_src; }
    return out;
  }
};

// -> [y, saved]; see the section comment.  Returns an empty list when the arrangement is not the plain one (the caller then runs the modules).
std::vector<at::Tensor> conv_fin(const at::Tensor& features, const at::Tensor& weight, const at::Tensor& tab_f, int64_t flip_f, const at::Tensor& tab_b,
                                 int64_t flip_b, int64_t n_out, int64_t centre, const c10::optional<at::Tensor>& pairs,
                                 const c10::optional<at::Tensor>& pair_num, int64_t side_src, const c10::optional<at::Tensor>& perm_b,
                                 const c10::optional<at::Tensor>& conv_bias, bool want_stats, const c10::optional<at::Tensor>& running_mean,
                                 const c10::optional<at::Tensor>& running_var, const c10::optional<at::Tensor>& num_batches_tracked, bool training,
                                 double momentum, double eps, const c10::optional<at::Tensor>& pre_saved, const c10::optional<at::Tensor>& pre_gamma,
                                 const c10::optional<at::Tensor>& pre_beta, bool pre_relu, bool pre_batch_stats) {
  const bool has_bias = conv_bias.has_value() && conv_bias->defined();
  const bool batch_stats = want_stats && (training || !(running_mean.has_value() && running_mean->defined()));
  if (!features.is_cuda() || !fuse_bn_stats() || weight.size(-1) > 1024) return {};
  if (n_out < 2 && training) return {};      // torch raises for one value per channel: let the caller run the module
  if (has_bias && !batch_stats) return {};   // a bias whose gradient is NOT identically zero (no train-mode BatchNorm behind it): the modules run
  return ConvFinFn::apply(features, weight, tab_f, flip_f, tab_b, flip_b, n_out, centre, pairs, pair_num, side_src, perm_b, conv_bias, want_stats,
                          running_mean, running_var, num_batches_tracked, training, momentum, eps, pre_saved, pre_gamma, pre_beta, pre_relu, pre_batch_stats);
}
at::Tensor bn_apply(const at::Tensor& x, const at::Tensor& saved, const c10::optional<at::Tensor>& weight, const c10::optional<at::Tensor>& bias, bool relu,
                    const c10::optional<at::Tensor>& residual, bool batch_stats) {
  return BnApplyFn::apply(x, saved, weight, bias, relu, residual, batch_stats);
}
bool prenorm_supported(int64_t c_src, int64_t c_dst, int64_t kvol, int64_t n_dst, int64_t flip_k) {
  return fv2p_sparse_conv_prenorm_supported(static_cast<int>(c_src), static_cast<int>(c_dst), static_cast<int>(kvol), n_dst, static_cast<int>(flip_k), 0) == 1;
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
  m.def("conv_fin", &conv_fin, "sparse conv with its BatchNorm statistics finalised by the launch (and optionally the source rows' BatchNorm + ReLU on the gather) -> [y, saved]");
  m.def("bn_apply", &bn_apply, "relu?((x - mean) * invstd * gamma + beta [+ residual]) with finalised statistics, one launch");
  m.def("prenorm_supported", &prenorm_supported, "the conv kernel of this shape can normalise its source rows on the gather");
  m.def("set_bn_fold", &set_bn_fold, "round-6 arrangement of conv / BatchNorm / residual blocks on (default) or off (the round-5 one)");
  m.def("bn_fold", &bn_fold);
  m.def("set_bn_one", &set_bn_one, "BatchNorm passes without conv-epilogue sums as one launch each (default) or as reduce + apply");
  m.def("bn_one", &bn_one);
  m.def("set_bn_wide", &set_bn_wide, "large BatchNorm passes with the wide reduce finalised by its own launch (default) or the <= 64-workgroup reduce + folding apply");
  m.def("bn_wide", &bn_wide);
  m.def("sparse_conv_bn_relu", &sparse_conv_bn_relu, "sparse conv -> BatchNorm1d (-> ReLU) with autograd, one call per backbone block");
}
