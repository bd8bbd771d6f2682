"""Drop-in KNRM for matchmaker (matchmaker/models/knrm.py): same constructor / from_config / forward /
forward_representation / introspection surface and state_dict keys (`dense.weight` only — mu and sigma
are plain attributes in the reference, :31-32); the cosine match + kernel pooling block (:55-84) runs
in libmm_native.so through the same kernel as TK (mm_kernel_pool_fwd / mm_kernel_pool_bwd).

KNRM differs from TK's block only in constants: no alpha scaler, `log(clamp(pkq, 1e-10)) * 0.01` (:76),
and the cosine matrix is multiplied by the query x document mask before the kernels (:62-65) — which the
second mask multiply (:72) makes irrelevant for the pooled sums (masked positions contribute exactly 0).
So  score = kernel_pool(q, d, q_mask, d_mask, mu, sigma, alpha = 1, w = 0.01 * dense.weight).
Called from NeuralIR_Encoder.forward (neuralIR_encoder.py:86-87); selected by models/all.py:151.
"""
from typing import List

import torch
import torch.nn as nn

from . import ops
from .tk import kernel_pool_train


def kernel_mus(n_kernels: int) -> List[float]:
    """knrm.py:100-114: first kernel = exact match (mu 1), the rest are bin centres over [-1, 1]."""
    mus = [1.0]
    if n_kernels == 1:
        return mus
    bin_size = 2.0 / (n_kernels - 1)
    mus.append(1 - bin_size / 2)
    for i in range(1, n_kernels - 1):
        mus.append(mus[i] - bin_size)
    return mus


def kernel_sigmas(n_kernels: int) -> List[float]:
    """knrm.py:116-130: 1e-4 for the exact-match kernel, half a bin for the others."""
    bin_size = 2.0 / (n_kernels - 1) if n_kernels > 1 else 0.0
    sigmas = [0.0001]
    if n_kernels == 1:
        return sigmas
    return sigmas + [0.5 * bin_size] * (n_kernels - 1)


class KNRM(nn.Module):
    """KNRM (http://www.cs.cmu.edu/~zhuyund/papers/end-end-neural.pdf) with native kernel pooling."""

    @staticmethod
    def from_config(config, word_embeddings_out_dim):          # knrm.py:20-22
        return KNRM(n_kernels=config["knrm_kernels"])

    def __init__(self, n_kernels: int):
        super().__init__()
        # non-persistent buffers: follow .to(device) like the reference's cuda Variables, stay out of state_dict
        self.register_buffer("mu", torch.tensor(kernel_mus(n_kernels), dtype=torch.float32).view(1, 1, 1, n_kernels),
                             persistent=False)
        self.register_buffer("sigma", torch.tensor(kernel_sigmas(n_kernels), dtype=torch.float32).view(1, 1, 1, n_kernels),
                             persistent=False)
        self.register_buffer("_ones", torch.ones(n_kernels, dtype=torch.float32), persistent=False)
        self.dense = nn.Linear(n_kernels, 1, bias=False)                       # :38
        torch.nn.init.uniform_(self.dense.weight, -0.014, 0.014)               # :41

    def forward(self, query_embeddings: torch.Tensor, document_embeddings: torch.Tensor,
                query_pad_oov_mask: torch.Tensor, document_pad_oov_mask: torch.Tensor,
                output_secondary_output: bool = False):
        """knrm.py:44-92 — same arguments and return conventions."""
        q = query_embeddings.float()
        d = document_embeddings.float()
        w = self.dense.weight.view(-1) * 0.01                                  # :76 folded into the bin weights
        needs_grad = torch.is_grad_enabled() and any(t.requires_grad for t in (q, d, self.dense.weight))
        per_kernel = None
        if needs_grad:
            score = kernel_pool_train(q, d, query_pad_oov_mask.float(), document_pad_oov_mask.float(),
                                        self.mu.view(-1), self.sigma.view(-1), self._ones, w)
        else:
            score, per_kernel = ops.kernel_pool(q, d, query_pad_oov_mask, document_pad_oov_mask, self.mu, self.sigma,
                                                self._ones, w, return_per_kernel=True)
        if output_secondary_output:
            if per_kernel is None:
                per_kernel = ops.kernel_pool(q.detach(), d.detach(), query_pad_oov_mask, document_pad_oov_mask, self.mu,
                                             self.sigma, self._ones, w.detach(), return_per_kernel=True)[1]
            per_kernel = per_kernel * 0.01                                     # the reference's per_kernel carries :76
            query_mean_vector = query_embeddings.sum(dim=1) / query_pad_oov_mask.sum(dim=1).unsqueeze(-1)
            qn = q / (q.norm(p=2, dim=-1, keepdim=True) + 1e-13)               # interpretability output only
            dn = d / (d.norm(p=2, dim=-1, keepdim=True) + 1e-13)
            cos = torch.bmm(qn, dn.transpose(-1, -2))
            cos = cos * query_pad_oov_mask.unsqueeze(-1) * document_pad_oov_mask.unsqueeze(1)
            return score, {"score": score, "per_kernel": per_kernel, "query_mean_vector": query_mean_vector,
                           "cosine_matrix_masked": cos}
        return score

    def forward_representation(self, sequence_embeddings: torch.Tensor, sequence_mask: torch.Tensor) -> torch.Tensor:
        return sequence_embeddings * sequence_mask.unsqueeze(-1)               # :94-95

    def get_param_stats(self):                                                 # :97-98
        return "KNRM: linear weight: " + str(self.dense.weight.data)

    def get_param_secondary(self):                                             # knrm.py get_param_secondary
        return {"kernel_weight": self.dense.weight}
