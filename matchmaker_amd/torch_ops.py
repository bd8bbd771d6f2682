"""torch.ops.mm_native.* — the native operators as registered PyTorch custom ops.

Importing this module defines

    torch.ops.mm_native.maxsim(q, d, q_mask, d_mask, pairs_per_query=1,
                               sim_round=False, sum_round=False)                    -> [n_pairs]
    torch.ops.mm_native.maxsim_inbatch(q, q_mask, d, d_mask, bug_compatible=False,
                                       sim_round=False, sum_round=False)            -> [Bq, Bd]
    torch.ops.mm_native.kernel_pool(q, d, q_mask, d_mask, mu, sigma, alpha, w,
                                    pairs_per_query=1, d_gate=None, clamp_min=1e-10) -> [n_pairs]
    torch.ops.mm_native.tkl_window_pool(q_ctx, chunks, chunk_mask, chunk_slot, q_mask, params,
                                        B, C, K, saturation)                        -> (score [B], windows [B, W])

for HIP tensors only (dispatch key CUDA; a CPU tensor raises NotImplementedError: there is no CPU kernel), each
with a fake (meta) implementation for tracing, an autograd formula backed by the native backward kernels
(maxsim, kernel_pool) and an autocast rule that mirrors what the reference's eager code does under
`torch.cuda.amp.autocast(enabled=use_fp16)` (colbert.py:60: the bmm runs in fp16 and RETURNS fp16 — the MaxSim ops cast
their vectors to the autocast dtype and set sim_round, so every per-token maximum is rounded as the reference's fp16 `max`
is — and the promoted `sum` returns fp32; the TK family stays fp32 — allennlp's cosine has no fp16 path the configs use, tk.yaml `use_fp16: False`).
The drop-in modules call `matchmaker_amd.ops` directly; these registrations are the boundary for callers
that want dispatcher-level ops (torch.compile graphs, TorchScript-free export, other extensions).
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops

_NS = "mm_native"


# ---------------------------------------------------------------------------------------------- maxsim
@torch.library.custom_op(_NS + "::maxsim", mutates_args=(), device_types="cuda")
def maxsim(q: Tensor, d: Tensor, q_mask: Optional[Tensor], d_mask: Optional[Tensor],
           pairs_per_query: int = 1, sim_round: bool = False, sum_round: bool = False) -> Tensor:
    return ops.maxsim(q, d, q_mask, d_mask, pairs_per_query, sim_round=sim_round, sum_round=sum_round)


@maxsim.register_fake
def _(q, d, q_mask, d_mask, pairs_per_query=1, sim_round=False, sum_round=False):
    return q.new_empty((d.shape[0],), dtype=torch.float32)


def _maxsim_setup(ctx, inputs, output):
    q, d, q_mask, d_mask, ppq, _sim, _sum = inputs
    ctx.save_for_backward(q, d, q_mask, d_mask)
    ctx.ppq = ppq


def _maxsim_backward(ctx, g):
    q, d, q_mask, d_mask = ctx.saved_tensors
    if ctx.ppq != 1:
        raise ops.NativeError("mm_native::maxsim backward needs the pair-per-row layout (pairs_per_query = 1), "
                              "the one train.py feeds")
    gq, gd = ops.maxsim_bwd(q, d, q_mask, d_mask, g, grad_dtype=q.dtype)
    return gq, gd, None, None, None, None, None


maxsim.register_autograd(_maxsim_backward, setup_context=_maxsim_setup)

_FRAG = torch.library.Library(_NS, "FRAGMENT")
_AUTOCAST_KEYS = torch._C.DispatchKeySet(torch._C.DispatchKey.AutocastCPU) | torch._C.DispatchKeySet(torch._C.DispatchKey.AutocastCUDA)


def _ac_cast(t):
    return t.to(torch.get_autocast_dtype("cuda")) if t.dtype == torch.float32 else t


def _maxsim_autocast(_ks, q, d, q_mask, d_mask, pairs_per_query=1, sim_round=False, sum_round=False):
    """What eager does under autocast (colbert.py:60-75): vectors in the autocast dtype, similarities and maxima in it
    (sim_round), the sum promoted to fp32."""
    with torch._C._ExcludeDispatchKeyGuard(_AUTOCAST_KEYS):
        return torch.ops.mm_native.maxsim(_ac_cast(q), _ac_cast(d), q_mask, d_mask, pairs_per_query, True, False)


_FRAG.impl("maxsim", _maxsim_autocast, "AutocastCUDA", with_keyset=True)


@torch.library.custom_op(_NS + "::maxsim_inbatch", mutates_args=(), device_types="cuda")
def maxsim_inbatch(q: Tensor, q_mask: Optional[Tensor], d: Tensor, d_mask: Optional[Tensor],
                   bug_compatible: bool = False, sim_round: bool = False, sum_round: bool = False) -> Tensor:
    return ops.maxsim_inbatch(q, q_mask, d, d_mask, bug_compatible, sim_round=sim_round, sum_round=sum_round)


@maxsim_inbatch.register_fake
def _(q, q_mask, d, d_mask, bug_compatible=False, sim_round=False, sum_round=False):
    return q.new_empty((q.shape[0], d.shape[0]), dtype=torch.float32)


def _maxsim_inbatch_autocast(_ks, q, q_mask, d, d_mask, bug_compatible=False, sim_round=False, sum_round=False):
    with torch._C._ExcludeDispatchKeyGuard(_AUTOCAST_KEYS):
        return torch.ops.mm_native.maxsim_inbatch(_ac_cast(q), q_mask, _ac_cast(d), d_mask, bug_compatible, True, False)


_FRAG.impl("maxsim_inbatch", _maxsim_inbatch_autocast, "AutocastCUDA", with_keyset=True)


# ---------------------------------------------------------------------------------------------- kernel pooling
@torch.library.custom_op(_NS + "::kernel_pool", mutates_args=(), device_types="cuda")
def kernel_pool(q: Tensor, d: Tensor, q_mask: Optional[Tensor], d_mask: Optional[Tensor], mu: Tensor, sigma: Tensor,
                alpha: Tensor, w: Tensor, pairs_per_query: int = 1, d_gate: Optional[Tensor] = None,
                clamp_min: float = 1e-10) -> Tensor:
    return ops.kernel_pool(q, d, q_mask, d_mask, mu, sigma, alpha, w, pairs_per_query=pairs_per_query,
                           d_gate=d_gate, clamp_min=clamp_min)


@kernel_pool.register_fake
def _(q, d, q_mask, d_mask, mu, sigma, alpha, w, pairs_per_query=1, d_gate=None, clamp_min=1e-10):
    return q.new_empty((d.shape[0],), dtype=torch.float32)


def _kp_setup(ctx, inputs, output):
    q, d, q_mask, d_mask, mu, sigma, alpha, w, ppq, d_gate, clamp_min = inputs
    ctx.save_for_backward(q, d, q_mask, d_mask, mu, sigma, alpha, w, d_gate)
    ctx.ppq, ctx.clamp_min = ppq, clamp_min


def _kp_backward(ctx, g):
    q, d, q_mask, d_mask, mu, sigma, alpha, w, d_gate = ctx.saved_tensors
    if ctx.ppq != 1:
        raise ops.NativeError("mm_native::kernel_pool backward needs the pair-per-row layout (pairs_per_query = 1)")
    r = ops.kernel_pool_bwd(q, d, q_mask, d_mask, mu, sigma, alpha, w, g, d_gate=d_gate, clamp_min=ctx.clamp_min)
    gg = r[4].view_as(d_gate) if d_gate is not None else None
    return r[0], r[1], None, None, None, None, r[2].view_as(alpha), r[3].view_as(w), None, gg, None


kernel_pool.register_autograd(_kp_backward, setup_context=_kp_setup)
torch.library.register_autocast(_NS + "::kernel_pool", "cuda", torch.float32)


# ---------------------------------------------------------------------------------------------- TKL
@torch.library.custom_op(_NS + "::tkl_window_pool", mutates_args=(), device_types="cuda")
def tkl_window_pool(q_ctx: Tensor, chunks: Tensor, chunk_mask: Tensor, chunk_slot: Tensor, q_mask: Tensor,
                    params: Tensor, B: int, C: int, K: int, saturation: str) -> Tuple[Tensor, Tensor]:
    score, win = ops.tkl_score(q_ctx, chunks, chunk_mask, chunk_slot, q_mask, params, B, C, K, saturation,
                               return_windows=True)
    return score, win


@tkl_window_pool.register_fake
def _(q_ctx, chunks, chunk_mask, chunk_slot, q_mask, params, B, C, K, saturation):
    W = (max(C * 40, 30) - 30) // 2 + 1
    return q_ctx.new_empty((B,), dtype=torch.float32), q_ctx.new_empty((B, W), dtype=torch.float32)


def _tkl_setup(ctx, inputs, output):
    q_ctx, chunks, chunk_mask, chunk_slot, q_mask, params, B, C, K, saturation = inputs
    ctx.save_for_backward(q_ctx, chunks, chunk_mask, chunk_slot, q_mask, params, output[1])
    ctx.meta = (B, C, K, saturation)


def _tkl_backward(ctx, g, _gwin):
    """mm_tkl_bwd: gradients w.r.t. the contextualised query, the contextualised chunks and the packed parameter vector
    (the window scores are an auxiliary output: no gradient flows through them)."""
    q_ctx, chunks, chunk_mask, chunk_slot, q_mask, params, win = ctx.saved_tensors
    B, C, K, saturation = ctx.meta
    gq, gc, gp = ops.tkl_bwd(q_ctx, chunks, chunk_mask, chunk_slot, q_mask, params, win, g, B, C, K, saturation)
    return gq.to(q_ctx.dtype), gc.to(chunks.dtype), None, None, None, gp.view_as(params).to(params.dtype), None, None, None, None


tkl_window_pool.register_autograd(_tkl_backward, setup_context=_tkl_setup)
torch.library.register_autocast(_NS + "::tkl_window_pool", "cuda", torch.float32)
