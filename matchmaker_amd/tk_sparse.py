"""Drop-in TK-Sparse (CIKM20_TK_Sparse) for matchmaker: constructor, from_config, forward /
forward_representation / reanimate, parameter + buffer names (= state_dict keys) as in
matchmaker/models/published/cikm20_tk_sparse.py; the cosine match + gated kernel pooling block (:106-146)
runs in libmm_native.so (mm_kernel_pool_ex_fwd / mm_kernel_pool_ex_bwd with the stop-word vector as the
document-token gate).

What stays PyTorch, as in the reference: the contextualiser (:162-172) and the two-layer stop-word MLP
(:132-133) whose ReLU output is the gate.  Called from NeuralIR_Encoder.forward (neuralIR_encoder.py:86-87);
its second return value feeds the L1 sparsity loss of train.py.
"""
from typing import List

import torch
import torch.nn as nn

from . import ops
from .tk import kernel_pool_train, sinusoid_positions


class CIKM20_TK_Sparse(nn.Module):
    """TK with a learned per-document-token gate that removes stop words from the score."""

    @staticmethod
    def from_config(config, word_embeddings_out_dim):      # cikm20_tk_sparse.py:18-28
        return CIKM20_TK_Sparse(word_embeddings_out_dim,
                                kernels_mu=config["tk_kernels_mu"],
                                kernels_sigma=config["tk_kernels_sigma"],
                                att_heads=config["tk_att_heads"],
                                att_layer=config["tk_att_layer"],
                                att_proj_dim=config["tk_att_proj_dim"],
                                att_ff_dim=config["tk_att_ff_dim"],
                                max_length=config["max_doc_length"],
                                use_diff_posencoding=config["tk_use_diff_posencoding"])

    def __init__(self, _embsize: int, kernels_mu: List[float], kernels_sigma: List[float], att_heads: int,
                 att_layer: int, att_proj_dim: int, att_ff_dim: int, max_length: int, use_diff_posencoding: bool):
        super().__init__()
        if len(kernels_mu) != len(kernels_sigma):
            raise Exception("len(kernels_mu) != len(kernels_sigma)")
        n_kernels = len(kernels_mu)
        self.mixer_stop = nn.Parameter(torch.full([1], 0.5, dtype=torch.float32, requires_grad=True))
        self.mixer = nn.Parameter(torch.full([1], 0.5, dtype=torch.float32, requires_grad=True))
        self.use_diff_posencoding = use_diff_posencoding
        self.register_buffer("positional_features_q", sinusoid_positions(_embsize, max_length))
        if use_diff_posencoding:
            self.register_buffer("positional_features_d", sinusoid_positions(_embsize, max_length + 500)[:, 500:, :])
        else:
            self.register_buffer("positional_features_d", self.positional_features_q)
        layer = nn.TransformerEncoderLayer(_embsize, att_heads, dim_feedforward=att_ff_dim, dropout=0)
        self.contextualizer = nn.TransformerEncoder(layer, att_layer, norm=None)
        self.register_buffer("mu", torch.tensor(kernels_mu, dtype=torch.float32).view(1, 1, 1, n_kernels))
        self.register_buffer("sigma", torch.tensor(kernels_sigma, dtype=torch.float32).view(1, 1, 1, n_kernels))
        self.kernel_bin_weights = nn.Linear(n_kernels, 1, bias=False)
        torch.nn.init.uniform_(self.kernel_bin_weights.weight, -0.014, 0.014)
        self.kernel_alpha_scaler = nn.Parameter(torch.full([1, 1, n_kernels], 1, dtype=torch.float32, requires_grad=True))
        self.stop_word_reducer = nn.Linear(_embsize, 100, bias=True)
        self.stop_word_reducer2 = nn.Linear(100, 1, bias=True)
        torch.nn.init.constant_(self.stop_word_reducer2.bias, 1)

    def reanimate(self, added_bias):                       # cikm20_tk_sparse.py:91-92
        self.stop_word_reducer2.bias.data += added_bias

    def forward(self, query_embeddings: torch.Tensor, document_embeddings: torch.Tensor, query_mask: torch.Tensor,
                document_mask: torch.Tensor, output_secondary_output: bool = False):
        """cikm20_tk_sparse.py:94-154 — returns (score, document_stop_words), or with the secondary output
        (score, dict, document_stop_words)."""
        query_embeddings, _ = self.forward_representation(
            query_embeddings, query_mask, self.positional_features_q[:, :query_embeddings.shape[1], :])
        document_embeddings_orig = document_embeddings
        document_embeddings, document_context = self.forward_representation(
            document_embeddings, document_mask, self.positional_features_d[:, :document_embeddings.shape[1], :])

        # the gate: ReLU(MLP(mix of the raw and the contextualised document embedding)) on real tokens
        stop_in = self.mixer_stop * document_embeddings_orig + (1 - self.mixer_stop) * document_context
        gate = torch.relu(self.stop_word_reducer2(torch.tanh(self.stop_word_reducer(stop_in)))).squeeze(-1)
        document_stop_words = (gate * document_mask).unsqueeze(1)                      # [B, 1, D]

        q = query_embeddings.float()
        d = document_embeddings.float()
        w = self.kernel_bin_weights.weight
        needs_grad = torch.is_grad_enabled() and any(
            t.requires_grad for t in (q, d, w, self.kernel_alpha_scaler, document_stop_words))
        if needs_grad:
            score = kernel_pool_train(q, d, query_mask.float(), document_mask.float(), self.mu.view(-1),
                                        self.sigma.view(-1), self.kernel_alpha_scaler.view(-1), w.view(-1),
                                        document_stop_words.squeeze(1).float())
            per_kernel = None
        else:
            score, per_kernel = ops.kernel_pool(q, d, query_mask, document_mask, self.mu, self.sigma,
                                                self.kernel_alpha_scaler, w, return_per_kernel=True,
                                                d_gate=document_stop_words.squeeze(1))
        if output_secondary_output:
            if per_kernel is None:
                per_kernel = ops.kernel_pool(q.detach(), d.detach(), query_mask, document_mask, self.mu, self.sigma,
                                             self.kernel_alpha_scaler, w, return_per_kernel=True,
                                             d_gate=document_stop_words.squeeze(1))[1]
            query_mean_vector = query_embeddings.sum(dim=1) / query_mask.sum(dim=1).unsqueeze(-1)
            qn = q / (q.norm(p=2, dim=-1, keepdim=True) + 1e-13)     # interpretability output only
            dn = d / (d.norm(p=2, dim=-1, keepdim=True) + 1e-13)
            cosine = torch.bmm(qn, dn.transpose(-1, -2)) * query_mask.unsqueeze(-1) * document_mask.unsqueeze(1)
            return score, {"score": score, "per_kernel": per_kernel, "query_mean_vector": query_mean_vector,
                           "cosine_matrix_masked": cosine, "document_stop_words": document_stop_words}, document_stop_words
        return score, document_stop_words

    def forward_representation(self, sequence_embeddings: torch.Tensor, sequence_mask: torch.Tensor,
                               positional_features=None):
        """cikm20_tk_sparse.py:156-172: (masked mix of raw and contextualised embeddings, contextualised)."""
        if positional_features is None:
            positional_features = self.positional_features_d[:, :sequence_embeddings.shape[1], :]
        sequence_embeddings = sequence_embeddings * sequence_mask.unsqueeze(-1)
        ctx = self.contextualizer((sequence_embeddings + positional_features).transpose(1, 0),
                                  src_key_padding_mask=~sequence_mask.bool()).transpose(1, 0)
        mixed = (self.mixer * sequence_embeddings + (1 - self.mixer) * ctx) * sequence_mask.unsqueeze(-1)
        return mixed, ctx

    def get_param_stats(self):            # cikm20_tk_sparse.py:248-252
        return "TK: dense w: " + str(self.kernel_bin_weights.weight.data) + "self.kernel_alpha_scaler: " + \
            str(self.kernel_alpha_scaler.data) + "stop_word_reducer" + str(self.stop_word_reducer.weight.data) + \
            str(self.stop_word_reducer.bias.data) + "stop_word_reducer2" + str(self.stop_word_reducer2.weight.data) + \
            str(self.stop_word_reducer2.bias.data) + "mixer: " + str(self.mixer.data) + "self.mixer_stop: " + \
            str(self.mixer_stop.data)

    def get_param_secondary(self):        # cikm20_tk_sparse.py:254-257
        return {"kernel_bin_weights": self.kernel_bin_weights.weight,
                "kernel_alpha_scaler": self.kernel_alpha_scaler,
                "mixer": self.mixer}
