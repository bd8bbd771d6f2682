// Host plumbing, not a kernel: the paired MaxSim (mm_maxsim_fwd / mm_maxsim_bwd) and the TK kernel pooling
// (mm_kernel_pool_ex_fwd2 / mm_kernel_pool_ex_bwd2) as C++ torch::autograd::Functions.
//
// Why: the reference trains with batch_size_train 32 x 2 = 64 pairs (config/train/defaults.yaml:114; train.py:347-348 forward,
// :503-524 loss.backward()).  At that size the scoring block's kernels take ~45 us and the step took ~155 us: the rest was
// Python — torch.autograd.Function.apply building the node, and the backward running as PYTHON on the autograd engine's
// device thread (a thread hop that also has to take the GIL).  tools/host_step_profile.py: Function.apply path 28 us,
// run_backward 45 us, of which the backward operator's own Python 20.  The same two C-ABI calls issued from a C++ node need
// neither the interpreter nor the GIL in the backward.
//
// Boundary: this file talks to the scoring library only through include/mm_native.h (dlopen of the path the Python side
// loaded, so that MM_NATIVE_LIB A/B builds are honoured) and to torch only for tensors, streams and the autograd graph.
// matchmaker_amd/colbert.py uses it when it was built (matchmaker_amd/build.py) and the call is in its fast-path shape;
// otherwise the Python autograd.Function (_MaxSimFn) runs — same kernels, same results (tests/test_colbert_dropin_gpu.py
// compares the two bit for bit).  MM_MAXSIM_PY_AUTOGRAD=1 forces the Python node (A/B runs).
#include <torch/extension.h>

#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <dlfcn.h>

#include <algorithm>
#include <string>
#include <vector>

#include "mm_native.h"

namespace {

struct Api {
  void* handle = nullptr;
  decltype(&mm_maxsim_fwd) fwd = nullptr;
  decltype(&mm_maxsim_bwd) bwd = nullptr;
  decltype(&mm_maxsim_workspace_bytes) fwd_ws = nullptr;
  decltype(&mm_maxsim_bwd_workspace_bytes) bwd_ws = nullptr;
  decltype(&mm_kernel_pool_ex_fwd2) kp_fwd = nullptr;
  decltype(&mm_kernel_pool_ex_bwd2) kp_bwd = nullptr;
  decltype(&mm_kernel_pool_workspace_bytes) kp_fwd_ws = nullptr;
  decltype(&mm_kernel_pool_bwd_workspace_bytes2) kp_bwd_ws = nullptr;
  decltype(&mm_tkl_fwd) tkl_fwd = nullptr;
  decltype(&mm_tkl_bwd) tkl_bwd = nullptr;
  decltype(&mm_tkl_workspace_bytes) tkl_fwd_ws = nullptr;
  decltype(&mm_tkl_bwd_workspace_bytes2) tkl_bwd_ws = nullptr;
  decltype(&mm_last_error) last_error = nullptr;
  decltype(&mm_abi_version) abi = nullptr;
} api;

void init(const std::string& lib_path) {
  if (api.handle) return;
  void* h = dlopen(lib_path.c_str(), RTLD_NOW | RTLD_LOCAL);
  TORCH_CHECK(h, "mm_autograd: cannot load ", lib_path, ": ", dlerror());
#define MM_SYM(field, name)                                        \
  api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, #name)); \
  TORCH_CHECK(api.field, "mm_autograd: ", lib_path, " does not export " #name)
  MM_SYM(fwd, mm_maxsim_fwd);
  MM_SYM(bwd, mm_maxsim_bwd);
  MM_SYM(fwd_ws, mm_maxsim_workspace_bytes);
  MM_SYM(bwd_ws, mm_maxsim_bwd_workspace_bytes);
  MM_SYM(kp_fwd, mm_kernel_pool_ex_fwd2);
  MM_SYM(kp_bwd, mm_kernel_pool_ex_bwd2);
  MM_SYM(kp_fwd_ws, mm_kernel_pool_workspace_bytes);
  MM_SYM(kp_bwd_ws, mm_kernel_pool_bwd_workspace_bytes2);
  MM_SYM(tkl_fwd, mm_tkl_fwd);
  MM_SYM(tkl_bwd, mm_tkl_bwd);
  MM_SYM(tkl_fwd_ws, mm_tkl_workspace_bytes);
  MM_SYM(tkl_bwd_ws, mm_tkl_bwd_workspace_bytes2);
  MM_SYM(last_error, mm_last_error);
  MM_SYM(abi, mm_abi_version);
#undef MM_SYM
  TORCH_CHECK(api.abi() == MM_ABI_VERSION, "mm_autograd: ", lib_path, " has ABI version ", api.abi(), ", built against ",
              MM_ABI_VERSION);
  api.handle = h;
}

int dtype_code(const at::Tensor& t) {
  switch (t.scalar_type()) {
    case at::kFloat: return MM_F32;
    case at::kHalf: return MM_F16;
    case at::kBFloat16: return MM_BF16;
    default: TORCH_CHECK(false, "mm_autograd: unsupported dtype ", t.scalar_type());
  }
}

// -> (pointer, kind) of a mask argument in the encodings of include/mm_native.h; `keep` holds a converted copy alive
std::pair<const void*, int> mask_arg(const c10::optional<at::Tensor>& m, int64_t rows, int64_t L, at::Tensor& keep, const char* name,
                                     const at::Device& dev) {
  if (!m.has_value() || !m->defined()) return {nullptr, MM_MASK_NONE};
  const at::Tensor& t = *m;
  // (a mask on ANOTHER device — DataParallel misuse — would hand the kernel a pointer it cannot read: ops._dev_check's rule)
  TORCH_CHECK(t.is_cuda() && t.device() == dev, "mm_autograd: ", name, " must be on the vectors' device (", dev, "), got ", t.device());
  if (t.dim() == 1) {
    TORCH_CHECK(t.size(0) == rows, "mm_autograd: ", name, " has ", t.size(0), " lengths for ", rows, " rows");
    keep = t.scalar_type() == at::kInt && t.is_contiguous() ? t : t.to(at::kInt).contiguous();
    return {keep.data_ptr(), MM_MASK_LEN_I32};
  }
  TORCH_CHECK(t.dim() == 2 && t.size(0) == rows && t.size(1) == L, "mm_autograd: ", name, " must be [", rows, ", ", L, "]");
  int kind;
  keep = t;
  switch (t.scalar_type()) {
    case at::kLong: kind = MM_MASK_I64; break;
    case at::kFloat: kind = MM_MASK_F32; break;
    case at::kByte:
    case at::kBool: kind = MM_MASK_U8; break;
    default: keep = t.ne(0); kind = MM_MASK_U8;
  }
  if (!keep.is_contiguous()) keep = keep.contiguous();
  return {keep.data_ptr(), kind};
}

void check_rc(int rc, const char* what) { TORCH_CHECK(rc == MM_OK, what, " failed (", rc, "): ", api.last_error()); }

class MaxSimPaired : public torch::autograd::Function<MaxSimPaired> {
 public:
  static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& q_in, const at::Tensor& d_in,
                            const c10::optional<at::Tensor>& q_mask, const c10::optional<at::Tensor>& d_mask, int64_t flags) {
    TORCH_CHECK(api.handle, "mm_autograd: init(lib_path) was not called");
    TORCH_CHECK(q_in.is_cuda() && d_in.is_cuda() && q_in.device() == d_in.device(), "mm_autograd: q / d must be on one HIP device");
    TORCH_CHECK(q_in.dim() == 3 && d_in.dim() == 3 && q_in.scalar_type() == d_in.scalar_type(), "mm_autograd: q / d: [rows, tokens, dim], same dtype");
    const at::Tensor q = q_in.contiguous(), d = d_in.contiguous();
    const int64_t B = d.size(0), Q = q.size(1), D = d.size(1), E = d.size(2);
    TORCH_CHECK(q.size(0) == B && q.size(2) == E, "mm_autograd: pair-per-row layout: q ", q.sizes(), " vs d ", d.sizes());
    const int dt = dtype_code(q);
    TORCH_CHECK(E % (dt == MM_F32 ? 4 : 8) == 0, "mm_autograd: rows must be 16-byte multiples (the Python path pads other widths)");
    at::Tensor qk, dk;
    const auto qm = mask_arg(q_mask, B, Q, qk, "q_mask", q.device());
    const auto dm = mask_arg(d_mask, B, D, dk, "d_mask", q.device());
    at::Tensor out = at::empty({B}, q.options().dtype(at::kFloat));
    if (B > 0) {
      const c10::DeviceGuard guard(q.device());
      const size_t wsb = api.fwd_ws(B, 1, (int)Q, (int)D, qm.second, dm.second);
      at::Tensor ws = wsb ? at::empty({(int64_t)wsb}, q.options().dtype(at::kByte)) : at::Tensor();
      void* stream = c10::hip::getCurrentHIPStream(q.device().index()).stream();
      check_rc(api.fwd(q.data_ptr(), d.data_ptr(), qm.first, qm.second, dm.first, dm.second, out.data_ptr<float>(), B, 1, (int)Q,
                       (int)D, (int)E, dt, (int)flags, wsb ? ws.data_ptr() : nullptr, wsb, stream),
               "mm_maxsim_fwd");
    }
    // (nothing of the forward is saved but its inputs: the backward recomputes the similarities on the device)
    ctx->save_for_backward({q, d, qk.defined() ? qk : at::Tensor(), dk.defined() ? dk : at::Tensor()});
    ctx->saved_data["qkind"] = (int64_t)qm.second;
    ctx->saved_data["dkind"] = (int64_t)dm.second;
    return out;
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const at::Tensor &q = saved[0], &d = saved[1], &qk = saved[2], &dk = saved[3];
    const int qkind = (int)ctx->saved_data["qkind"].toInt(), dkind = (int)ctx->saved_data["dkind"].toInt();
    const int64_t B = d.size(0), Q = q.size(1), D = d.size(1), E = d.size(2);
    // gradients in the token vectors' own dtype, as the Python node asks for (grad_dtype = q.dtype): summed in fp32, rounded once
    at::Tensor gq = at::empty_like(q), gd = at::empty_like(d);
    if (B > 0) {
      at::Tensor go = grads[0].reshape({-1});
      TORCH_CHECK(go.device() == q.device(), "mm_autograd: grad_out on ", go.device(), ", the vectors on ", q.device());
      if (go.scalar_type() != at::kFloat) go = go.to(at::kFloat);
      go = go.contiguous();
      TORCH_CHECK(go.numel() == B, "mm_autograd: grad_out has ", go.numel(), " elements for ", B, " pairs");
      const c10::DeviceGuard guard(q.device());
      const size_t wsb = api.bwd_ws(B, (int)Q, (int)D, qkind, dkind);
      at::Tensor ws = wsb ? at::empty({(int64_t)wsb}, q.options().dtype(at::kByte)) : at::Tensor();
      void* stream = c10::hip::getCurrentHIPStream(q.device().index()).stream();
      const int dt = dtype_code(q);
      check_rc(api.bwd(q.data_ptr(), d.data_ptr(), qk.defined() ? qk.data_ptr() : nullptr, qkind, dk.defined() ? dk.data_ptr() : nullptr,
                       dkind, go.data_ptr<float>(), gq.data_ptr(), gd.data_ptr(), dt, B, (int)Q, (int)D, (int)E, dt,
                       wsb ? ws.data_ptr() : nullptr, wsb, stream),
               "mm_maxsim_bwd");
    }
    return {gq, gd, at::Tensor(), at::Tensor(), at::Tensor()};
  }
};

at::Tensor maxsim_paired(const at::Tensor& q, const at::Tensor& d, const c10::optional<at::Tensor>& q_mask,
                         const c10::optional<at::Tensor>& d_mask, int64_t flags) {
  return MaxSimPaired::apply(q, d, q_mask, d_mask, flags);
}

// TK kernel pooling in the pair-per-row layout (ecai20_tk.py:105-124 through mm_kernel_pool_ex_fwd2; train.py:347-348 forward,
// :503-524 backward).  At batch_size_train 32 x 2 the step was 242 us of which 94 were the Python node.
// The forward hands its pooled kernel sums to the backward (one trip of the documents through HBM instead of two).
at::Tensor vec_f32(const at::Tensor& t, int64_t K, const char* name, const at::Device& dev) {
  TORCH_CHECK(t.device() == dev, "mm_autograd: ", name, " on ", t.device(), ", the embeddings on ", dev);
  at::Tensor v = t.reshape({-1});
  if (v.scalar_type() != at::kFloat) v = v.to(at::kFloat);
  v = v.contiguous();
  TORCH_CHECK(v.numel() == K, "mm_autograd: ", name, " has ", v.numel(), " elements for ", K, " kernels");
  return v;
}

class KernelPool : public torch::autograd::Function<KernelPool> {
 public:
  static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& q_in, const at::Tensor& d_in,
                            const c10::optional<at::Tensor>& q_mask, const c10::optional<at::Tensor>& d_mask, const at::Tensor& mu_in,
                            const at::Tensor& sigma_in, const at::Tensor& alpha_in, const at::Tensor& w_in,
                            const c10::optional<at::Tensor>& gate_in, double clamp_min) {
    TORCH_CHECK(api.handle, "mm_autograd: init(lib_path) was not called");
    TORCH_CHECK(q_in.is_cuda() && d_in.is_cuda() && q_in.device() == d_in.device(), "mm_autograd: q / d must be on one HIP device");
    TORCH_CHECK(q_in.dim() == 3 && d_in.dim() == 3 && q_in.scalar_type() == at::kFloat && d_in.scalar_type() == at::kFloat,
                "mm_autograd: kernel pooling takes float32 [rows, tokens, dim] embeddings (tk.yaml use_fp16: False)");
    const at::Tensor q = q_in.contiguous(), d = d_in.contiguous();
    const int64_t B = d.size(0), Q = q.size(1), D = d.size(1), E = d.size(2), K = mu_in.numel();
    TORCH_CHECK(q.size(0) == B && q.size(2) == E, "mm_autograd: pair-per-row layout: q ", q.sizes(), " vs d ", d.sizes());
    TORCH_CHECK(E % 4 == 0, "mm_autograd: rows must be 16-byte multiples (the Python path pads other widths)");
    const at::Device dev = q.device();
    const at::Tensor mu = vec_f32(mu_in, K, "mu", dev), sigma = vec_f32(sigma_in, K, "sigma", dev), alpha = vec_f32(alpha_in, K, "alpha", dev),
                     w = vec_f32(w_in, K, "w", dev);
    at::Tensor gate;
    if (gate_in.has_value() && gate_in->defined()) {
      TORCH_CHECK(gate_in->device() == dev, "mm_autograd: d_gate on ", gate_in->device(), ", the embeddings on ", dev);
      gate = gate_in->reshape({B, -1}).to(at::kFloat).contiguous();
      TORCH_CHECK(gate.size(1) == D, "mm_autograd: d_gate has shape ", gate_in->sizes(), " for ", B, " documents of ", D, " tokens");
    }
    at::Tensor qk, dk;
    const auto qm = mask_arg(q_mask, B, Q, qk, "q_mask", dev);
    const auto dm = mask_arg(d_mask, B, D, dk, "d_mask", dev);
    at::Tensor out = at::empty({B}, q.options());
    at::Tensor pooled = at::empty({B, Q, K}, q.options());
    if (B > 0) {
      const c10::DeviceGuard guard(dev);
      const size_t wsb = api.kp_fwd_ws(B, 1, (int)Q, (int)D, qm.second, dm.second);
      at::Tensor ws = wsb ? at::empty({(int64_t)wsb}, q.options().dtype(at::kByte)) : at::Tensor();
      void* stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
      check_rc(api.kp_fwd(q.data_ptr(), d.data_ptr(), qm.first, qm.second, dm.first, dm.second, gate.defined() ? gate.data_ptr<float>() : nullptr,
                          nullptr, 0, mu.data_ptr<float>(), sigma.data_ptr<float>(), alpha.data_ptr<float>(), w.data_ptr<float>(),
                          (float)clamp_min, out.data_ptr<float>(), nullptr, pooled.data_ptr<float>(), B, 1, (int)Q, (int)D, (int)E, (int)K,
                          MM_F32, wsb ? ws.data_ptr() : nullptr, wsb, stream),
               "mm_kernel_pool_ex_fwd2");
    }
    ctx->save_for_backward({q, d, qk.defined() ? qk : at::Tensor(), dk.defined() ? dk : at::Tensor(), mu, sigma, alpha, w,
                            gate.defined() ? gate : at::Tensor(), pooled});
    ctx->saved_data["qkind"] = (int64_t)qm.second;
    ctx->saved_data["dkind"] = (int64_t)dm.second;
    ctx->saved_data["clamp"] = clamp_min;
    ctx->saved_data["alpha_shape"] = alpha_in.sizes().vec();
    ctx->saved_data["w_shape"] = w_in.sizes().vec();
    ctx->saved_data["gate_shape"] = gate.defined() ? gate_in->sizes().vec() : std::vector<int64_t>{};
    return out;
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const at::Tensor &q = saved[0], &d = saved[1], &qk = saved[2], &dk = saved[3], &mu = saved[4], &sigma = saved[5], &alpha = saved[6],
                     &w = saved[7], &gate = saved[8], &pooled = saved[9];
    const int qkind = (int)ctx->saved_data["qkind"].toInt(), dkind = (int)ctx->saved_data["dkind"].toInt();
    const int64_t B = d.size(0), Q = q.size(1), D = d.size(1), E = d.size(2), K = mu.numel();
    at::Tensor gq = at::empty_like(q), gd = at::empty_like(d);
    at::Tensor gaw = at::zeros({2, B, K}, q.options());   // per-pair rows of grad_alpha, grad_w: one memset, one sum
    at::Tensor ga = gaw.select(0, 0), gw = gaw.select(0, 1);
    at::Tensor gg = gate.defined() ? at::zeros({B, D}, q.options()) : at::Tensor();
    if (B > 0) {
      at::Tensor go = grads[0].reshape({-1});
      TORCH_CHECK(go.device() == q.device(), "mm_autograd: grad_out on ", go.device(), ", the embeddings on ", q.device());
      if (go.scalar_type() != at::kFloat) go = go.to(at::kFloat);
      go = go.contiguous();
      TORCH_CHECK(go.numel() == B, "mm_autograd: grad_out has ", go.numel(), " elements for ", B, " pairs");
      const c10::DeviceGuard guard(q.device());
      const size_t wsb = api.kp_bwd_ws(B, (int)Q, (int)D, (int)E, qkind, dkind);
      at::Tensor ws = wsb ? at::empty({(int64_t)wsb}, q.options().dtype(at::kByte)) : at::Tensor();
      void* stream = c10::hip::getCurrentHIPStream(q.device().index()).stream();
      check_rc(api.kp_bwd(q.data_ptr(), d.data_ptr(), qk.defined() ? qk.data_ptr() : nullptr, qkind, dk.defined() ? dk.data_ptr() : nullptr,
                          dkind, gate.defined() ? gate.data_ptr<float>() : nullptr, mu.data_ptr<float>(), sigma.data_ptr<float>(),
                          alpha.data_ptr<float>(), w.data_ptr<float>(), (float)ctx->saved_data["clamp"].toDouble(), pooled.data_ptr<float>(),
                          go.data_ptr<float>(), gq.data_ptr<float>(), gd.data_ptr<float>(), gg.defined() ? gg.data_ptr<float>() : nullptr,
                          ga.data_ptr<float>(), gw.data_ptr<float>(), B, (int)Q, (int)D, (int)E, (int)K, wsb ? ws.data_ptr() : nullptr, wsb,
                          stream),
               "mm_kernel_pool_ex_bwd2");
    }
    // per-pair parameter rows -> one deterministic sum on the device, in the parameters' own shapes
    const at::Tensor sums = gaw.sum(1);
    at::Tensor gas = sums.select(0, 0).reshape(ctx->saved_data["alpha_shape"].toIntVector());
    at::Tensor gws = sums.select(0, 1).reshape(ctx->saved_data["w_shape"].toIntVector());
    if (gg.defined()) gg = gg.reshape(ctx->saved_data["gate_shape"].toIntVector());
    return {gq, gd, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), gas, gws, gg, at::Tensor()};
  }
};

at::Tensor kernel_pool(const at::Tensor& q, const at::Tensor& d, const c10::optional<at::Tensor>& q_mask,
                       const c10::optional<at::Tensor>& d_mask, const at::Tensor& mu, const at::Tensor& sigma, const at::Tensor& alpha,
                       const at::Tensor& w, const c10::optional<at::Tensor>& gate, double clamp_min) {
  return KernelPool::apply(q, d, q_mask, d_mask, mu, sigma, alpha, w, gate, clamp_min);
}

// TKL scoring (sigir20_tkl.py:180-286 through mm_tkl_fwd / mm_tkl_bwd): the training step's node
// without Python in its backward.  `scoring`: the parameter tensors of the packed vector in pack order; lo / cnt / full: where each
// sits in the packed gradient, how many of its elements the kernels read, its own numel (lo < 0: a parameter the active saturation
// never reads — its gradient stays undefined, as in the reference and in the Python node tkl._TKLScoreFn).
class TklScore : public torch::autograd::Function<TklScore> {
 public:
  static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const at::Tensor& q_in, const at::Tensor& c_in,
                                                const at::Tensor& chunk_mask, const at::Tensor& chunk_slot, const at::Tensor& q_mask,
                                                const at::Tensor& packed_in, int64_t B, int64_t C, int64_t K, int64_t saturation,
                                                at::TensorList scoring, const at::Tensor& layout) {
    // layout: int64 [3, n] on the host = (lo, cnt, full) per scoring tensor (ONE argument: every element of an integer list would
    // count as an input of the node)
    TORCH_CHECK(layout.device().is_cpu() && layout.scalar_type() == at::kLong && layout.dim() == 2 && layout.size(0) == 3 &&
                    layout.size(1) == (int64_t)scoring.size(), "mm_autograd: layout must be a host int64 [3, ", scoring.size(), "] tensor");
    const at::Tensor lay = layout.contiguous();
    const int64_t n_sc = lay.size(1);
    std::vector<int64_t> lo(lay.data_ptr<int64_t>(), lay.data_ptr<int64_t>() + n_sc), cnt(lay.data_ptr<int64_t>() + n_sc, lay.data_ptr<int64_t>() + 2 * n_sc),
        full(lay.data_ptr<int64_t>() + 2 * n_sc, lay.data_ptr<int64_t>() + 3 * n_sc);
    TORCH_CHECK(api.handle, "mm_autograd: init(lib_path) was not called");
    TORCH_CHECK(q_in.is_cuda() && q_in.dim() == 3 && c_in.dim() == 3 && q_in.scalar_type() == at::kFloat && c_in.scalar_type() == at::kFloat,
                "mm_autograd: TKL scoring takes float32 q_ctx [B, Q, E] and chunks [P, 50, E] (tkl.yaml use_fp16: False)");
    const at::Device dev = q_in.device();
    for (const at::Tensor* t : {&c_in, &chunk_mask, &chunk_slot, &q_mask, &packed_in})
      TORCH_CHECK(t->device() == dev, "mm_autograd: every TKL operand must be on ", dev, ", got ", t->device());
    TORCH_CHECK(scoring.size() == lo.size() && lo.size() == cnt.size() && cnt.size() == full.size(), "mm_autograd: scoring layout lists differ in length");
    const at::Tensor q = q_in.contiguous(), c = c_in.contiguous();
    const int64_t Q = q.size(1), E = q.size(2), P = c.size(0);
    TORCH_CHECK(q.size(0) == B && (P == 0 || (c.size(1) == 50 && c.size(2) == E)) && E % 4 == 0, "mm_autograd: bad TKL shapes q_ctx ", q.sizes(),
                " chunks ", c.sizes());
    const at::Tensor cm = chunk_mask.to(at::kFloat).contiguous(), cs = chunk_slot.to(at::kInt).contiguous(),
                     qm = q_mask.to(at::kFloat).contiguous(), packed = packed_in.detach().to(at::kFloat).contiguous();
    TORCH_CHECK(packed.numel() == MM_TKL_NPARAMS(K, E), "mm_autograd: packed parameters have ", packed.numel(), " elements");
    TORCH_CHECK(cm.numel() == P * 50 && cs.numel() == P && qm.numel() == B * Q && C > 0 && P <= B * C,
                "mm_autograd: TKL masks do not fit the operands: chunk_mask ", cm.sizes(), " chunk_slot ", cs.sizes(), " q_mask ", qm.sizes(),
                " for P = ", P, ", B = ", B, ", Q = ", Q, ", C = ", C);
    const int64_t W = (std::max<int64_t>(C * 40, 30) - 30) / 2 + 1;
    at::Tensor out = at::empty({B}, q.options()), win = at::empty({B, W}, q.options());
    if (B > 0) {
      const c10::DeviceGuard guard(dev);
      const size_t wsb = api.tkl_fwd_ws(B, P, (int)C, (int)Q, (int)K);
      at::Tensor ws = wsb ? at::empty({(int64_t)wsb}, q.options().dtype(at::kByte)) : at::Tensor();
      void* stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
      check_rc(api.tkl_fwd(q.data_ptr(), P ? c.data_ptr() : nullptr, P ? cm.data_ptr<float>() : nullptr, P ? cs.data_ptr<int32_t>() : nullptr,
                           qm.data_ptr<float>(), packed.data_ptr<float>(), win.data_ptr<float>(), out.data_ptr<float>(), B, P, (int)C, (int)Q,
                           (int)E, (int)K, (int)saturation, wsb ? ws.data_ptr() : nullptr, wsb, stream),
               "mm_tkl_fwd");
    }
    ctx->save_for_backward({q, c, cm, cs, qm, packed, win});
    ctx->saved_data["C"] = C;
    ctx->saved_data["K"] = K;
    ctx->saved_data["sat"] = saturation;
    ctx->saved_data["lo"] = lo;
    ctx->saved_data["cnt"] = cnt;
    ctx->saved_data["full"] = full;
    std::vector<std::vector<int64_t>> shapes;
    std::vector<int64_t> dtypes;
    for (const at::Tensor& t : scoring) {
      shapes.push_back(t.sizes().vec());
      dtypes.push_back((int64_t)t.scalar_type());
    }
    ctx->saved_data["shapes"] = shapes;
    ctx->saved_data["dtypes"] = dtypes;
    ctx->saved_data["q_dtype"] = (int64_t)q_in.scalar_type();
    ctx->mark_non_differentiable({win});
    return {out, win};
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const at::Tensor &q = saved[0], &c = saved[1], &cm = saved[2], &cs = saved[3], &qm = saved[4], &packed = saved[5], &win = saved[6];
    const int64_t B = q.size(0), Q = q.size(1), E = q.size(2), P = c.size(0);
    const int64_t C = ctx->saved_data["C"].toInt(), K = ctx->saved_data["K"].toInt(), sat = ctx->saved_data["sat"].toInt();
    const auto lo = ctx->saved_data["lo"].toIntVector(), cnt = ctx->saved_data["cnt"].toIntVector(), full = ctx->saved_data["full"].toIntVector();
    const int64_t NP = packed.numel();
    at::Tensor gq = at::empty_like(q), gc = P ? at::empty_like(c) : at::zeros_like(c), gp;
    if (B > 0) {
      at::Tensor go = grads[0].reshape({-1});
      TORCH_CHECK(go.device() == q.device(), "mm_autograd: grad_out on ", go.device(), ", the embeddings on ", q.device());
      if (go.scalar_type() != at::kFloat) go = go.to(at::kFloat);
      go = go.contiguous();
      TORCH_CHECK(go.numel() == B, "mm_autograd: grad_out has ", go.numel(), " elements for ", B, " documents");
      at::Tensor gpd = at::empty({B, NP}, q.options());
      const c10::DeviceGuard guard(q.device());
      const size_t wsb = api.tkl_bwd_ws(B, (int)C, (int)Q, (int)E);
      at::Tensor ws = at::empty({(int64_t)(wsb ? wsb : 16)}, q.options().dtype(at::kByte));
      void* stream = c10::hip::getCurrentHIPStream(q.device().index()).stream();
      check_rc(api.tkl_bwd(q.data_ptr(), P ? c.data_ptr() : nullptr, P ? cm.data_ptr<float>() : nullptr, P ? cs.data_ptr<int32_t>() : nullptr,
                           qm.data_ptr<float>(), packed.data_ptr<float>(), win.data_ptr<float>(), go.data_ptr<float>(), gq.data_ptr<float>(),
                           P ? gc.data_ptr<float>() : nullptr, gpd.data_ptr<float>(), B, P, (int)C, (int)Q, (int)E, (int)K, (int)sat,
                           ws.data_ptr(), wsb, stream),
               "mm_tkl_bwd");
      gp = gpd.sum(0);                      // per-document rows -> one deterministic sum on the device
    } else {
      gp = at::zeros({NP}, q.options());
    }
    torch::autograd::variable_list out = {gq, gc, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(),
                                          at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    const auto shapes = ctx->saved_data["shapes"].toList();
    const auto dtypes = ctx->saved_data["dtypes"].toIntVector();
    for (size_t i = 0; i < lo.size(); ++i) {
      if (lo[i] < 0) {
        out.push_back(at::Tensor());
        continue;
      }
      at::Tensor g = gp.narrow(0, lo[i], cnt[i]);
      if (full[i] != cnt[i]) {              // kernel_mult [4, 1, 1, 1, K]: only row 0 is read (:246)
        at::Tensor z = at::zeros({full[i]}, gp.options());
        z.narrow(0, 0, cnt[i]).copy_(g);
        g = z;
      }
      const auto shape = shapes.get(i).toIntVector();
      out.push_back(g.reshape(shape).to((at::ScalarType)dtypes[i]));
    }
    out.push_back(at::Tensor());            // layout
    return out;
  }
};

std::vector<at::Tensor> tkl_score(const at::Tensor& q_ctx, const at::Tensor& chunks, const at::Tensor& chunk_mask, const at::Tensor& chunk_slot,
                                  const at::Tensor& q_mask, const at::Tensor& packed, int64_t B, int64_t C, int64_t K, int64_t saturation,
                                  std::vector<at::Tensor> scoring, const at::Tensor& layout) {
  return TklScore::apply(q_ctx, chunks, chunk_mask, chunk_slot, q_mask, packed, B, C, K, saturation, at::TensorList(scoring), layout);   // (a TensorList: every element an input of the node)
}

// the torch this file was compiled against: _fast.module() refuses the extension under another one (an artefact that
// dlopens against an ABI-incompatible torch would crash the hot path instead of leaving it to the Python node)
std::string built_torch_version() { return TORCH_VERSION; }

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "C++ autograd node around mm_maxsim_fwd / mm_maxsim_bwd (host plumbing; see the header of mm_autograd.cpp)";
  m.def("init", &init, "dlopen the scoring library (path of libmm_native.so) and resolve the entry points");
  m.def("built_torch_version", &built_torch_version, "TORCH_VERSION of the headers this extension was compiled against");
  m.def("kernel_pool", &kernel_pool, "TK kernel pooling (pair-per-row) with a native autograd node", py::arg("q"), py::arg("d"),
        py::arg("q_mask"), py::arg("d_mask"), py::arg("mu"), py::arg("sigma"), py::arg("alpha"), py::arg("w"), py::arg("gate"),
        py::arg("clamp_min"));
  m.def("tkl_score", &tkl_score, "TKL scoring (score, window scores) with a native autograd node", py::arg("q_ctx"), py::arg("chunks"),
        py::arg("chunk_mask"), py::arg("chunk_slot"), py::arg("q_mask"), py::arg("packed"), py::arg("B"), py::arg("C"), py::arg("K"),
        py::arg("saturation"), py::arg("scoring"), py::arg("layout"));
  m.def("maxsim_paired", &maxsim_paired, "paired MaxSim with a native autograd node", py::arg("q"), py::arg("d"), py::arg("q_mask"),
        py::arg("d_mask"), py::arg("flags"));
}
