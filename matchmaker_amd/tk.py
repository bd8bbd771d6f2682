"""Drop-in TK (ECAI20_TK) for matchmaker: constructor, from_config, forward / forward_representation,
parameter + buffer names (= state_dict keys) as in matchmaker/models/published/ecai20_tk.py; the
cosine match + kernel pooling block (:105-124) runs in libmm_native.so (mm_kernel_pool_fwd).

The contextualiser (positional encoding + nn.TransformerEncoder + mixer, :95-96 / :133-143) stays
PyTorch, as in the reference.  Called from NeuralIR_Encoder.forward (neuralIR_encoder.py:86-87).
"""
import math
import os
from typing import List

import torch
import torch.nn as nn

from . import _fast, ops


def sinusoid_positions(dim: int, length: int, min_timescale: float = 1.0, max_timescale: float = 1.0e4) -> torch.Tensor:
    """Frequency positional features [1, length, dim] (sin half | cos half | zero column if dim is odd);
    same numbers as ecai20_tk.py:145-194."""
    half = dim // 2
    step = math.log(float(max_timescale) / float(min_timescale)) / float(half - 1)
    inv = min_timescale * torch.exp(torch.arange(half, dtype=torch.float32) * -step)
    ang = torch.arange(length, dtype=torch.float32).unsqueeze(1) * inv.unsqueeze(0)
    feats = torch.cat([ang.sin(), ang.cos()], dim=1)
    if dim % 2:
        feats = torch.cat([feats, feats.new_zeros(length, 1)], dim=1)
    return feats.unsqueeze(0)


class _KernelPoolFn(torch.autograd.Function):
    """Native forward (mm_kernel_pool_ex_fwd2) and native backward (mm_kernel_pool_ex_bwd2) of the pooling
    block for the training path (train.py:347-348 / :503-524).  gate / clamp_min: the TK-Sparse and IDCM
    variants (ops.kernel_pool).  The forward hands its pooled kernel sums (ecai20_tk.py:120, [B, Q, K]: 880 bytes per pair at
    Q = 20) to the backward, which then reads the documents once instead of twice."""

    @staticmethod
    def forward(ctx, q, d, q_mask, d_mask, mu, sigma, alpha, w, gate=None, clamp_min=1e-10):
        ctx.clamp_min = clamp_min
        if any(ctx.needs_input_grad) and q.shape[0] == d.shape[0]:
            out, pooled = ops.kernel_pool(q, d, q_mask, d_mask, mu, sigma, alpha, w, d_gate=gate, clamp_min=clamp_min, return_pooled=True)
        else:
            out, pooled = ops.kernel_pool(q, d, q_mask, d_mask, mu, sigma, alpha, w, d_gate=gate, clamp_min=clamp_min), None
        ctx.has_pooled = pooled is not None
        ctx.save_for_backward(q, d, q_mask, d_mask, mu, sigma, alpha, w, gate, pooled)
        return out

    @staticmethod
    def backward(ctx, g):
        q, d, q_mask, d_mask, mu, sigma, alpha, w, gate, pooled = ctx.saved_tensors
        r = ops.kernel_pool_bwd(q, d, q_mask, d_mask, mu, sigma, alpha, w, g, d_gate=gate, clamp_min=ctx.clamp_min,
                                pooled=pooled if ctx.has_pooled else None)
        gq, gd, ga, gw = r[:4]
        gg = r[4].view_as(gate) if gate is not None else None
        return gq, gd, None, None, None, None, ga.view_as(alpha), gw.view_as(w), gg, None


def kernel_pool_train(q, d, q_mask, d_mask, mu, sigma, alpha, w, gate=None, clamp_min=1e-10):
    """The pooling block WITH an autograd node (train.py:347-348 / :503-524): the C++ node of csrc_host/mm_autograd.cpp when the
    host extension was built and the call is in its shape (float32, pair-per-row, 16-byte rows) — at batch_size_train 32 x 2 the
    Python node's apply + backward were 94 of the step's 242 us — else the Python autograd.Function; the same two C-ABI calls
    (mm_kernel_pool_ex_fwd2 / _ex_bwd2) either way.  MM_KP_PY_AUTOGRAD=1 forces the Python node (A/B runs, tests)."""
    fast = _fast.module()
    if (fast is not None and hasattr(fast, "kernel_pool") and os.environ.get("MM_KP_PY_AUTOGRAD", "0") in ("", "0")
            and q.is_cuda and q.dtype == torch.float32 and d.dtype == torch.float32 and q.dim() == 3 and d.dim() == 3
            and q.shape[0] == d.shape[0] and q.shape[-1] % 4 == 0 and q.shape[1] <= 32):
        return fast.kernel_pool(q, d, q_mask, d_mask, mu, sigma, alpha, w, gate, float(clamp_min))
    return _KernelPoolFn.apply(q, d, q_mask, d_mask, mu, sigma, alpha, w, gate, clamp_min)


class ECAI20_TK(nn.Module):
    """TK: Transformer contextualisation + kernel pooling (https://arxiv.org/abs/2002.01854)."""

    @staticmethod
    def from_config(config, word_embeddings_out_dim):      # ecai20_tk.py:22-32
        return ECAI20_TK(word_embeddings_out_dim,
                         kernels_mu=config["tk_kernels_mu"],
                         kernels_sigma=config["tk_kernels_sigma"],
                         att_heads=config["tk_att_heads"],
                         att_layer=config["tk_att_layer"],
                         att_ff_dim=config["tk_att_ff_dim"],
                         max_length=config["max_doc_length"],
                         use_diff_posencoding=config["tk_use_diff_posencoding"],
                         mix_hybrid_context=config["tk_mix_hybrid_context"])

    def __init__(self, _embsize: int, kernels_mu: List[float], kernels_sigma: List[float], att_heads: int,
                 att_layer: int, att_ff_dim: int, max_length: int, use_diff_posencoding: bool,
                 mix_hybrid_context: bool):
        super().__init__()
        if len(kernels_mu) != len(kernels_sigma):
            raise Exception("len(kernels_mu) != len(kernels_sigma)")
        n_kernels = len(kernels_mu)
        self.use_diff_posencoding = use_diff_posencoding
        self.register_buffer("positional_features_q", sinusoid_positions(_embsize, max_length))
        if use_diff_posencoding:
            self.register_buffer("positional_features_d", sinusoid_positions(_embsize, max_length + 500)[:, 500:, :])
        else:
            self.register_buffer("positional_features_d", self.positional_features_q)
        layer = nn.TransformerEncoderLayer(_embsize, att_heads, dim_feedforward=att_ff_dim, dropout=0)
        self.contextualizer = nn.TransformerEncoder(layer, att_layer, norm=None)
        self.mix_hybrid_context = mix_hybrid_context
        self.mixer = nn.Parameter(torch.full([1], 0.5, dtype=torch.float32, requires_grad=True))
        self.register_buffer("mu", torch.tensor(kernels_mu, dtype=torch.float32).view(1, 1, 1, n_kernels))
        self.register_buffer("sigma", torch.tensor(kernels_sigma, dtype=torch.float32).view(1, 1, 1, n_kernels))
        self.kernel_bin_weights = nn.Linear(n_kernels, 1, bias=False)
        torch.nn.init.uniform_(self.kernel_bin_weights.weight, -0.014, 0.014)
        self.kernel_alpha_scaler = nn.Parameter(torch.full([1, 1, n_kernels], 1, dtype=torch.float32, requires_grad=True))

    def forward(self, query_embeddings: torch.Tensor, document_embeddings: torch.Tensor, query_mask: torch.Tensor,
                document_mask: torch.Tensor, output_secondary_output: bool = False):
        """ecai20_tk.py:87-131 — same arguments and return conventions."""
        query_embeddings = self.forward_representation(
            query_embeddings, query_mask, self.positional_features_q[:, :query_embeddings.shape[1], :])
        document_embeddings = self.forward_representation(
            document_embeddings, document_mask, self.positional_features_d[:, :document_embeddings.shape[1], :])

        q = query_embeddings.float()
        d = document_embeddings.float()
        w = self.kernel_bin_weights.weight
        needs_grad = torch.is_grad_enabled() and any(
            t.requires_grad for t in (q, d, w, self.kernel_alpha_scaler))
        if needs_grad:
            score = kernel_pool_train(q, d, query_mask.float(), document_mask.float(), self.mu.view(-1),
                                        self.sigma.view(-1), self.kernel_alpha_scaler.view(-1), w.view(-1))
            per_kernel = None
        else:
            score, per_kernel = ops.kernel_pool(q, d, query_mask, document_mask, self.mu, self.sigma,
                                                self.kernel_alpha_scaler, w, return_per_kernel=True)
        if output_secondary_output:
            if per_kernel is None:
                per_kernel = ops.kernel_pool(q.detach(), d.detach(), query_mask, document_mask, self.mu, self.sigma,
                                             self.kernel_alpha_scaler, w, return_per_kernel=True)[1]
            query_mean_vector = query_embeddings.sum(dim=1) / query_mask.sum(dim=1).unsqueeze(-1)
            qn = q / (q.norm(p=2, dim=-1, keepdim=True) + 1e-13)     # interpretability output only
            dn = d / (d.norm(p=2, dim=-1, keepdim=True) + 1e-13)
            cosine_matrix = torch.bmm(qn, dn.transpose(-1, -2))
            return score, {"score": score, "per_kernel": per_kernel, "query_mean_vector": query_mean_vector,
                           "cosine_matrix": cosine_matrix * document_mask.unsqueeze(1) * query_mask.unsqueeze(-1)}
        return score

    def forward_representation(self, sequence_embeddings: torch.Tensor, sequence_mask: torch.Tensor,
                               positional_features=None) -> torch.Tensor:
        """ecai20_tk.py:133-143."""
        if positional_features is None:
            positional_features = self.positional_features_d[:, :sequence_embeddings.shape[1], :]
        ctx = self.contextualizer((sequence_embeddings + positional_features).transpose(1, 0),
                                  src_key_padding_mask=~sequence_mask.bool()).transpose(1, 0)
        if self.mix_hybrid_context:
            return self.mixer * sequence_embeddings + (1 - self.mixer) * ctx
        return ctx

    def get_param_stats(self):            # ecai20_tk.py:206-207
        return "TK: kernel_bin_weights: " + str(self.kernel_bin_weights.weight.data) + " kernel_alpha_scaler: " + \
            str(self.kernel_alpha_scaler.data) + " mixer: " + str(self.mixer.data)

    def get_param_secondary(self):        # ecai20_tk.py:209-212
        return {"kernel_bin_weights": self.kernel_bin_weights.weight,
                "kernel_alpha_scaler": self.kernel_alpha_scaler,
                "mixer": self.mixer}
