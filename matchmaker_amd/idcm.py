"""IDCM's fast passage-selection module (ESM) on the native pooling operator.

matchmaker/models/published/sigir21_idcm.py:182-186 scores every (query, passage) pair with a cosine match
matrix + 11 RBF kernels + log pooling — the TK pooling block with the floor 1e-4 and a bias on the bin weights.
`sampler_scores` is that block on libmm_native.so (mm_kernel_pool_ex_fwd / mm_kernel_pool_ex_bwd, ragged
passage groups through `pair_query`); it is the whole of IDCM that lies on the hot path.
The rest of the model (passage windowing, BERT passage scorer, sampler losses, top-k combination) is outside
it and stays the reference's own class: `forward_native` below is IDCM.forward with the sampler block swapped for the
operator, and matchmaker_amd.patch.patch_matchmaker() installs it as the forward of a thin subclass of the reference's
class (constructor, from_config, state_dict, introspection: inherited unchanged).
torch.nn.functional.normalize (:169-170, :178-179) need not run in front of it: the native cosine normalises
rows itself (x / (|x| + 1e-13) vs x / max(|x|, 1e-12): the same number for every non-zero row, 0 for a zero
row in both).
"""
from typing import Dict, Optional, Union

import torch
from torch import nn as nn

from . import ops
from .tk import _KernelPoolFn

MU = [1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9]     # sigir21_idcm.py:106
SIGMA = [0.1] * 11                                                    # :107


def sampler_scores(query_ctx: torch.Tensor, document_ctx: torch.Tensor, query_mask: torch.Tensor,
                   document_mask: torch.Tensor, mu: torch.Tensor, sigma: torch.Tensor, alpha: torch.Tensor,
                   binweights: nn.Linear, pair_query: Optional[torch.Tensor] = None) -> torch.Tensor:
    """packed_patch_scores [P, 1] of sigir21_idcm.py:182-186 from the sampler's (un-normalised) token vectors
    document_ctx [P, D, E] and {0,1} masks.  query_ctx is [P, Q, E] (one query copy per passage, as the
    reference builds it, :143-144) or, with pair_query [P], [B, Q, E] + the document index of each passage."""
    q = query_ctx.float()
    d = document_ctx.float()
    w = binweights.weight
    needs_grad = torch.is_grad_enabled() and any(t.requires_grad for t in (q, d, w, alpha))
    if needs_grad:
        if pair_query is not None:                         # the native backward works pair per row
            q, query_mask = q.index_select(0, pair_query), query_mask.index_select(0, pair_query)
        s = _KernelPoolFn.apply(q, d, query_mask.float(), document_mask.float(), mu.reshape(-1), sigma.reshape(-1),
                                alpha.reshape(-1), w.reshape(-1), None, 1e-4)
    else:
        s = ops.kernel_pool(q, d, query_mask, document_mask, mu, sigma, alpha, w, clamp_min=1e-4,
                            pair_query=pair_query)
    return (s + binweights.bias).unsqueeze(-1)


# ---- the sampler's token vectors (sigir21_idcm.py:167-180, without the normalize) -----------------------
def sampler_vectors(self, ids: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """`self`: an IDCM module (the reference's class or any module with its attributes)."""
    emb = self.bert_model.embeddings(ids).detach()
    if self.sample_context == "ck-small":
        return self.sample_cnn3(self.sample_projector(emb).transpose(1, 2)).transpose(1, 2)
    if self.sample_context == "ck":
        return self.sample_cnn3(emb.transpose(1, 2)).transpose(1, 2)
    proj = self.tk_projector(emb)
    return self.tk_contextualizer(proj.transpose(1, 0), src_key_padding_mask=~mask.bool()).transpose(1, 0)

def forward_native(self, query: Dict[str, torch.LongTensor], document: Dict[str, torch.LongTensor], use_fp16: bool = True,
            output_secondary_output: bool = False, bert_part_cached: Union[bool, torch.Tensor] = False):
    """IDCM.forward (sigir21_idcm.py:111-274, same arguments and return conventions) with the passage sampler
    (:167-186) on the native operator.  `self` is the IDCM module: the reference's own class — patch_matchmaker() makes
    this function the forward of a thin subclass of it — or any module with the same attributes."""
    # passage windows over the document (drop [CLS], pad so the windows tile it): :117-141
    document_ids = document["input_ids"][:, 1:]
    n_tok = document_ids.shape[1]
    if n_tok > self.overlap:
        needed_padding = self.extended_chunk_size - ((n_tok % self.chunk_size) - self.overlap)
    else:
        needed_padding = self.extended_chunk_size - self.overlap - n_tok
    document_ids = nn.functional.pad(document_ids, (self.overlap, needed_padding), value=self.padding_idx)
    chunked_ids = document_ids.unfold(1, self.extended_chunk_size, self.chunk_size)
    batch_size, chunk_pieces = chunked_ids.shape[0], chunked_ids.shape[1]
    chunks_flat = chunked_ids.reshape(-1, self.extended_chunk_size)
    packed_indices = (chunks_flat[:, self.overlap:-self.overlap] != self.padding_idx).any(-1)
    orig_packed_indices = packed_indices.clone()
    total_chunks = chunks_flat.shape[0]
    q_ids_all = query["input_ids"].unsqueeze(1).expand(-1, chunk_pieces, -1).reshape(-1, query["input_ids"].shape[1])
    q_mask_all = query["attention_mask"].unsqueeze(1).expand(-1, chunk_pieces, -1).reshape(
        -1, query["attention_mask"].shape[1])

    def pack(indices):
        ids = chunks_flat[indices]
        return q_ids_all[indices], q_mask_all[indices], ids, (ids != self.padding_idx)

    packed_query_ids, packed_query_mask, ids_packed, mask_packed = pack(packed_indices)
    not_cached = isinstance(bert_part_cached, bool) and bert_part_cached is False

    if self.sample_n > -1:
        # the reference contextualises one query copy per passage (:167-178); the copies are identical, so
        # each document's query goes through the sampler once and the passages index it
        query_ctx = sampler_vectors(self, query["input_ids"], query["attention_mask"])
        document_ctx = sampler_vectors(self, ids_packed, mask_packed)
        passage_doc = torch.div(packed_indices.nonzero().squeeze(-1), chunk_pieces, rounding_mode="floor")
        packed_patch_scores = sampler_scores(query_ctx, document_ctx, query["attention_mask"], mask_packed, self.mu,
                                             self.sigma, self.kernel_alpha_scaler, self.sampling_binweights,
                                             pair_query=passage_doc)                                      # :182-186
        sampling_scores_per_doc = packed_patch_scores.new_zeros((total_chunks, 1))
        sampling_scores_per_doc[packed_indices] = packed_patch_scores
        sampling_scores_per_doc = sampling_scores_per_doc.reshape(batch_size, -1)
        sampling_scores_per_doc_orig = sampling_scores_per_doc.clone()
        sampling_scores_per_doc[sampling_scores_per_doc == 0] = -9000
        sampling_sorted = sampling_scores_per_doc.sort(descending=True)
        row_base = torch.arange(0, batch_size * chunk_pieces, chunk_pieces, device=sampling_scores_per_doc.device)
        sampled_indices = (sampling_sorted.indices + row_base.unsqueeze(-1))[:, :self.sample_n]
        sampled_indices_mask = torch.zeros_like(packed_indices).scatter(0, sampled_indices.reshape(-1), 1)
        if not self.training and not_cached:           # evaluation: BERT only reads the sampled passages
            packed_indices = sampled_indices_mask * packed_indices
            packed_query_ids, packed_query_mask, ids_packed, mask_packed = pack(packed_indices)

    # the expensive passage scores: :209-237
    with torch.set_grad_enabled(self.sample_n == -1 and self.training):
        if self.sample_n > -1:
            self.bert_model.eval()
        if bert_part_cached is None or not_cached:
            bert_vecs = self.forward_representation(torch.cat([packed_query_ids, ids_packed], dim=1),
                                                    torch.cat([packed_query_mask, mask_packed], dim=1))
            patch_scores = self._classification_layer(bert_vecs)
            scores_per_doc = patch_scores.new_zeros((total_chunks, 1))
            scores_per_doc[packed_indices] = patch_scores
            scores_per_doc = scores_per_doc.reshape(batch_size, -1)
            scores_per_doc_orig = scores_per_doc.clone()
            scores_per_doc_orig_sorter = scores_per_doc.clone()
        else:
            if bert_part_cached.shape[0] != batch_size or bert_part_cached.shape[1] != chunk_pieces:
                raise Exception("cache sanity check failed! should be:" + str(batch_size) + "," + str(chunk_pieces) +
                                " but is: " + str(bert_part_cached.shape[0]) + "," + str(bert_part_cached.shape[1]))
            scores_per_doc = bert_part_cached
            scores_per_doc_orig = bert_part_cached
            scores_per_doc_orig_sorter = bert_part_cached.clone()
        if self.sample_n > -1:
            scores_per_doc = scores_per_doc * sampled_indices_mask.view(batch_size, -1)
        if scores_per_doc.shape[1] < self.top_k_chunks:
            scores_per_doc = nn.functional.pad(scores_per_doc, (0, self.top_k_chunks - scores_per_doc.shape[1]))
        scores_per_doc[scores_per_doc == 0] = -9000
        scores_per_doc_orig_sorter[scores_per_doc_orig_sorter == 0] = -9000
        score = torch.sort(scores_per_doc, descending=True, dim=-1).values
        score[score <= -8900] = 0
        score = (score[:, :self.top_k_chunks] * self.top_k_scoring).sum(dim=1)

    if self.sample_n == -1:
        if output_secondary_output:
            return score, {"packed_indices": orig_packed_indices.view(batch_size, -1),
                           "bert_scores": scores_per_doc_orig}
        return score
    if output_secondary_output:
        return score, scores_per_doc_orig, {"score": score, "document_ids": document_ids,
                                            "packed_indices": orig_packed_indices.view(batch_size, -1),
                                            "sampling_scores": sampling_scores_per_doc_orig,
                                            "bert_scores": scores_per_doc_orig}, None, None
    orders = [sampling_sorted.indices, scores_per_doc_orig_sorter.sort(descending=True).indices]
    teacher = scores_per_doc_orig.detach()
    if self.sample_train_type == "mseloss":
        loss = torch.nn.MSELoss()(sampling_scores_per_doc_orig, teacher)
    elif self.sample_train_type == "kldivloss":
        loss = torch.nn.KLDivLoss(reduction="batchmean")(torch.softmax(sampling_scores_per_doc_orig, -1),
                                                         torch.softmax(scores_per_doc_orig, -1).detach())
    elif self.sample_train_type == "crossentropy":
        loss = _loss_fn(self)(sampling_scores_per_doc_orig, torch.softmax(scores_per_doc_orig, -1).detach())
    elif self.sample_train_type == "lambdaloss":
        gains_idx = torch.sort(scores_per_doc_orig_sorter, descending=True, dim=-1).indices + row_base.unsqueeze(-1)
        bert_gains = torch.zeros_like(packed_indices).float()
        for i in range(self.sample_n):
            bert_gains.scatter_(0, gains_idx[:, i].reshape(-1), self.sample_n - i)
        bert_gains[~packed_indices] = -9000
        loss = _loss_fn(self)(sampling_scores_per_doc, bert_gains.view(batch_size, -1).detach(),
                               padded_value_indicator=-9000)
    else:
        return None                                      # the reference falls off the end here too
    return score, scores_per_doc_orig, [[loss]], orders


def _loss_fn(self):
    """The sampler losses live in matchmaker.losses (outside this path): imported from there at the first training step
    unless the module carries its own `_sample_loss` callable."""
    if getattr(self, "_sample_loss", None) is not None:
        return self._sample_loss
    if self.sample_train_type == "lambdaloss":         # sigir21_idcm.py:1, :270
        from matchmaker.losses.lambdarank import LambdaLoss
        self._sample_loss = LambdaLoss("ndcgLoss2_scheme")
    else:                                              # :2, :258
        from matchmaker.losses.soft_crossentropy import SoftCrossEntropy
        self._sample_loss = SoftCrossEntropy()
    return self._sample_loss


def native_subclass(reference_cls):
    """A thin subclass of the reference's IDCM class (matchmaker/models/published/sigir21_idcm.py:13) whose forward is
    forward_native: everything else — constructor, from_config (:15-25, which instantiates the module-level name IDCM and
    therefore this subclass once patch_matchmaker() has rebound it), parameters, state_dict keys — is the reference's."""
    return type("IDCM", (reference_cls,), {"forward": forward_native, "__module__": __name__,
                                           "__doc__": "matchmaker's IDCM with the passage sampler on libmm_native.so"})
