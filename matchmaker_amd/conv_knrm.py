"""Drop-in Conv-KNRM for matchmaker (matchmaker/models/conv_knrm.py): same constructor / from_config /
forward surface and state_dict keys (`convolutions.<n>.1.{weight,bias}`, `dense.weight`).  The n-gram
convolutions (:39-46, :120-126) stay PyTorch as in the reference; the n_grams^2 cosine-match + kernel
pooling blocks (forward_matrix_kernel_pooling, :147-173) run in libmm_native.so through the TK pooling
kernel (inference: ONE launch over all n_grams^2 blocks, mm_kernel_pool_multi_fwd): the concatenation + dense layer of :132-137 is a sum over the (i, t) blocks of
kernel_pool(q_i, d_t, w = 0.01 * dense.weight[block]) — the per-block [B, K] tensors never exist.
Selected by models/all.py:152.
"""
import torch
import torch.nn as nn

from . import ops
from .knrm import kernel_mus
from .tk import kernel_pool_train


def kernel_sigmas(n_kernels: int):
    """conv_knrm.py:190-204: 1e-3 for the exact-match kernel (KNRM uses 1e-4), half a bin for the others."""
    bin_size = 2.0 / (n_kernels - 1) if n_kernels > 1 else 0.0
    sigmas = [0.001]
    if n_kernels == 1:
        return sigmas
    return sigmas + [0.5 * bin_size] * (n_kernels - 1)


class Conv_KNRM(nn.Module):
    """Conv-KNRM (http://www.cs.cmu.edu/~zhuyund/papers/WSDM_2018_Dai.pdf) with native kernel pooling."""

    @staticmethod
    def from_config(config, word_embeddings_out_dim):          # conv_knrm.py:20-25
        return Conv_KNRM(word_embeddings_out_dim=word_embeddings_out_dim, n_grams=config["conv_knrm_ngrams"],
                         n_kernels=config["conv_knrm_kernels"], conv_out_dim=config["conv_knrm_conv_out_dim"])

    def __init__(self, word_embeddings_out_dim: int, n_grams: int, n_kernels: int, conv_out_dim: int):
        super().__init__()
        self.n_kernels = n_kernels
        self.register_buffer("mu", torch.tensor(kernel_mus(n_kernels), dtype=torch.float32).view(1, 1, 1, n_kernels),
                             persistent=False)
        self.register_buffer("sigma", torch.tensor(kernel_sigmas(n_kernels), dtype=torch.float32).view(1, 1, 1, n_kernels),
                             persistent=False)
        self.register_buffer("_ones", torch.ones(n_kernels, dtype=torch.float32), persistent=False)
        self.convolutions = nn.ModuleList([
            nn.Sequential(nn.ConstantPad1d((0, i - 1), 0),
                          nn.Conv1d(kernel_size=i, in_channels=word_embeddings_out_dim, out_channels=conv_out_dim),
                          nn.ReLU())
            for i in range(1, n_grams + 1)])                                   # :39-47
        self.dense = nn.Linear(n_kernels * n_grams * n_grams, 1, bias=False)   # :53
        torch.nn.init.uniform_(self.dense.weight, -0.014, 0.014)               # :56

    def get_param_stats(self):                                                 # :58-59
        return "CONV-KNRM: linear weight: " + str(self.dense.weight.data)

    def get_param_secondary(self):                                             # :60-61
        return {"kernel_weight": self.dense.weight}

    def forward(self, query_embeddings: torch.Tensor, document_embeddings: torch.Tensor,
                query_pad_oov_mask: torch.Tensor, document_pad_oov_mask: torch.Tensor,
                output_secondary_output: bool = False):
        """conv_knrm.py:63-144 — same arguments and return conventions."""
        q_t = query_embeddings.transpose(1, 2)                                 # :112-113
        d_t = document_embeddings.transpose(1, 2)
        q_grams = [conv(q_t).transpose(1, 2).float().contiguous() for conv in self.convolutions]   # :118-126
        d_grams = [conv(d_t).transpose(1, 2).float().contiguous() for conv in self.convolutions]
        K = self.n_kernels
        w = self.dense.weight.view(-1) * 0.01                                  # the * 0.01 of :167
        qm, dm = query_pad_oov_mask, document_pad_oov_mask
        needs_grad = torch.is_grad_enabled() and (self.dense.weight.requires_grad or q_grams[0].requires_grad)
        mu, sigma = self.mu.view(-1), self.sigma.view(-1)
        if not needs_grad and K == 11 and len(q_grams) <= 4:
            # inference: all n_grams^2 match matrices in one launch, summed in the concat's (i, t) order
            score = ops.kernel_pool_multi(q_grams, d_grams, qm, dm, mu, sigma, self._ones, w)
        else:
            score = None
            block = 0
            for qg in q_grams:                                                 # :130-132, same (i, t) order as the concat
                for dg in d_grams:
                    wb = w[block * K:(block + 1) * K]
                    if needs_grad:
                        s = kernel_pool_train(qg, dg, qm.float(), dm.float(), mu, sigma, self._ones, wb)
                    else:
                        s = ops.kernel_pool(qg, dg, qm, dm, mu, sigma, self._ones, wb)
                    score = s if score is None else score + s
                    block += 1
        if output_secondary_output:
            return score, {}
        return score
