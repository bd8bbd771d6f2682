"""TEST FIXTURE — not product code.  An IDCM module that can be built WITHOUT the reference tree (the GPU box has none):
the reference's constructor arguments and attribute names (sigir21_idcm.py:27-108) and a restatement of its forward
(`forward_native`) with the ESM block (:167-186) on the product's operator `matchmaker_amd.idcm.sampler_scores`, so that the
operator can be pinned — in the place the reference calls it — on outputs of the REAL IDCM class (tests/golden/idcm_*.npz,
live runs of /root/reference).  The product holds the operator only; in a matchmaker checkout the route is the three-line
edit of sigir21_idcm.py:182-186 shown in INTEGRATION.md (IDCM's forward itself is out of scope)."""
from typing import Callable, Dict, Optional, Union

import torch
from torch import nn as nn

from matchmaker_amd.idcm import MU as _MU, SIGMA as _SIGMA, sampler_scores, sampler_vectors


def forward_native(self, query: Dict[str, torch.LongTensor], document: Dict[str, torch.LongTensor], use_fp16: bool = True,
            output_secondary_output: bool = False, bert_part_cached: Union[bool, torch.Tensor] = False):
    """TEST FIXTURE.  IDCM.forward (sigir21_idcm.py:111-274, same arguments and return conventions) restated with the passage
    sampler (:167-186) on the native operator `matchmaker_amd.idcm.sampler_scores` — the host logic a parity test needs
    around the operator to compare against outputs of the REAL class.  Not product code (IDCM's forward is out of scope)."""
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


class IDCM(nn.Module):
    """Intra-document cascade: a fast kernel-pooling sampler picks the passages a BERT scorer reads."""

    @staticmethod
    def from_config(config, padding_idx):                  # sigir21_idcm.py:15-25
        return IDCM(bert_model=config["bert_pretrained_model"],
                    trainable=config["bert_trainable"],
                    sample_train_type=config["idcm_sample_train_type"],
                    sample_n=config["idcm_sample_n"],
                    sample_context=config["idcm_sample_context"],
                    top_k_chunks=config["idcm_top_k_chunks"],
                    chunk_size=config["idcm_chunk_size"],
                    overlap=config["idcm_overlap"],
                    padding_idx=padding_idx)

    def __init__(self, bert_model, dropout: float = 0.0, trainable: bool = True, sample_train_type="lambdaloss",
                 sample_n=1, sample_context="ck", top_k_chunks=3, chunk_size=50, overlap=7, padding_idx: int = 0,
                 sample_loss: Optional[Callable] = None) -> None:
        """Arguments as sigir21_idcm.py:27-44.  sample_loss (extra, optional): callable used for the
        "lambdaloss" / "crossentropy" sampler losses, which live in matchmaker.losses (outside this path);
        when omitted they are imported from there at the first training step."""
        super().__init__()
        if isinstance(bert_model, str):
            from transformers import AutoModel
            self.bert_model = AutoModel.from_pretrained(bert_model)
        else:
            self.bert_model = bert_model
        for p in self.bert_model.parameters():
            p.requires_grad = trainable
        self._classification_layer = torch.nn.Linear(self.bert_model.config.hidden_size, 1)
        self.top_k_chunks = top_k_chunks
        self.top_k_scoring = nn.Parameter(torch.full([1, self.top_k_chunks], 1, dtype=torch.float32, requires_grad=True))
        self.padding_idx = padding_idx
        self.chunk_size = chunk_size
        self.overlap = overlap
        self.extended_chunk_size = self.chunk_size + 2 * self.overlap
        self.sample_train_type = sample_train_type
        self.sample_n = sample_n
        self.sample_context = sample_context
        self._sample_loss = sample_loss
        if sample_context == "ck":
            dim = self.bert_model.config.dim
            self.sample_cnn3 = nn.Sequential(nn.ConstantPad1d((0, 2), 0),
                                             nn.Conv1d(kernel_size=3, in_channels=dim, out_channels=dim), nn.ReLU())
        elif sample_context == "ck-small":
            self.sample_projector = nn.Linear(self.bert_model.config.dim, 384)
            self.sample_cnn3 = nn.Sequential(nn.ConstantPad1d((0, 2), 0),
                                             nn.Conv1d(kernel_size=3, in_channels=384, out_channels=128), nn.ReLU())
        elif sample_context == "tk":
            self.tk_projector = nn.Linear(self.bert_model.config.dim, 384)
            layer = nn.TransformerEncoderLayer(384, 8, dim_feedforward=384, dropout=0)
            self.tk_contextualizer = nn.TransformerEncoder(layer, 1, norm=None)
            self.tK_mixer = nn.Parameter(torch.full([1], 0.5, dtype=torch.float32, requires_grad=True))
        self.sampling_binweights = nn.Linear(11, 1, bias=True)
        torch.nn.init.uniform_(self.sampling_binweights.weight, -0.01, 0.01)
        self.kernel_alpha_scaler = nn.Parameter(torch.full([1, 1, 11], 1, dtype=torch.float32, requires_grad=True))
        self.register_buffer("mu", torch.tensor(_MU).view(1, 1, 1, -1))
        self.register_buffer("sigma", torch.tensor(_SIGMA).view(1, 1, 1, -1))

    forward = forward_native

    def forward_representation(self, ids, mask, type_ids=None):
        """sigir21_idcm.py:276-294: the [CLS] / pooled vector of the passage scorer."""
        prefix = self.bert_model.base_model_prefix
        if prefix == "distilbert":
            return self.bert_model(input_ids=ids, attention_mask=mask)[0][:, 0, :]
        if prefix == "longformer":
            _, pooled = self.bert_model(input_ids=ids, attention_mask=mask.long(),
                                        global_attention_mask=((1 - ids) * mask).long())
        elif prefix == "roberta":
            _, pooled = self.bert_model(input_ids=ids, attention_mask=mask)
        else:
            _, pooled = self.bert_model(input_ids=ids, token_type_ids=type_ids, attention_mask=mask)
        return pooled

    def get_param_stats(self):            # sigir21_idcm.py:296-298
        return "IDCM: sampling.conv_binweights: " + str(self.sampling_binweights.weight.data) + \
            str(self.sampling_binweights.bias) + "kernel_alpha_scaler" + str(self.kernel_alpha_scaler) + \
            "top_k_scoring:" + str(self.top_k_scoring.data)

    def get_param_secondary(self):        # sigir21_idcm.py:300-301
        return {}
