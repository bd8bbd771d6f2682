"""Drop-in ColBERT for matchmaker: same constructor / forward / aggregation surface and state_dict
keys as matchmaker/models/colbert.py (reference lines cited per method); the interaction scoring
runs in libmm_native.so (mm_maxsim_fwd / mm_maxsim_inbatch_fwd) instead of bmm + mask + max + sum.

Callers that keep working unchanged: eval.py:108, train.py:347-348 (forward), indexing_heads.py:23,52
(forward_representation), indexing_heads.py:55 (forward_aggregation), dynamic_teacher.py:245-276
(forward_inbatch_aggregation), train.py:241-244 (get_param_stats / get_param_secondary).
"""
from typing import Dict, Optional

import torch
from torch import nn
from transformers import AutoModel, PretrainedConfig, PreTrainedModel

from . import _fast, ops


class ColBERTConfig(PretrainedConfig):
    """Fields of matchmaker/models/colbert.py:10-16.  Declared through __init__ so that the class
    also constructs under transformers >= 5 (the reference's bare annotations do not)."""
    model_type = "ColBERT"

    def __init__(self, bert_model: str = "", compression_dim: int = 768, dropout: float = 0.0,
                 return_vecs: bool = False, trainable: bool = True, **kwargs):
        super().__init__(**kwargs)
        self.bert_model = bert_model
        self.compression_dim = compression_dim
        self.dropout = dropout
        self.return_vecs = return_vecs
        self.trainable = trainable


class _MaxSimFn(torch.autograd.Function):
    """Native forward (mm_maxsim_fwd) and native backward (mm_maxsim_bwd: arg-max routing recomputed
    on the device, no [B,Q,D] tensor) for the training path, train.py:347-348 / :503-524."""

    @staticmethod
    def forward(ctx, q, d, q_mask, d_mask, sim_round=False, sum_round=False):
        ctx.save_for_backward(q, d, q_mask, d_mask)
        return ops.maxsim(q, d, q_mask, d_mask, pairs_per_query=1, sim_round=sim_round, sum_round=sum_round)

    @staticmethod
    def backward(ctx, g):
        # (rounding is piecewise constant: the gradient is that of the unrounded maximum, routed to an arg-max of the
        # fp32 similarities — which is also an arg-max of the rounded ones)
        q, d, q_mask, d_mask = ctx.saved_tensors
        gq, gd = ops.maxsim_bwd(q, d, q_mask, d_mask, g, grad_dtype=q.dtype)     # one launch, gradients in the vectors' dtype
        return gq, gd, None, None, None, None


def _as_reference_would(query_vecs, document_vecs):
    """The token vectors in the dtype the reference's `bmm` / `mm` would see them in, + the rounding flags of its dtype flow
    (ops.reference_rounding): under autocast fp32 vectors are cast to the autocast dtype first (colbert.py:60,68;
    indexing_heads.py:49-56 for the fp32 memmap store)."""
    if torch.is_autocast_enabled("cuda") and query_vecs.dtype == torch.float32:
        ac = torch.get_autocast_dtype("cuda")
        query_vecs, document_vecs = query_vecs.to(ac), document_vecs.to(ac)
    elif query_vecs.dtype != document_vecs.dtype:
        document_vecs = document_vecs.to(query_vecs.dtype)
    return (query_vecs, document_vecs) + ops.reference_rounding(query_vecs)


class ColBERT(PreTrainedModel):
    """ColBERT (https://arxiv.org/abs/2004.12832) with MI355X-native late-interaction scoring."""

    config_class = ColBERTConfig
    base_model_prefix = "bert_model"
    is_teacher_model = False              # overridden by the dynamic teacher (dynamic_teacher.py:174)
    # forward_inbatch_aggregation masks score[i, j] with document i's mask in the reference
    # (colbert.py:158).  True = bit-for-bit the reference's behaviour (incl. its Bq == Bd limit);
    # set False for the mask-by-document-j semantics.
    inbatch_bug_compatible = True

    @staticmethod
    def from_config(config):              # colbert.py:27-35
        cfg = ColBERTConfig(bert_model=config["bert_pretrained_model"],
                            compression_dim=config["colbert_compression_dim"],
                            return_vecs=config.get("in_batch_negatives", False),
                            trainable=config["bert_trainable"])
        return ColBERT(cfg)

    def __init__(self, cfg: ColBERTConfig, bert_model: Optional[nn.Module] = None) -> None:
        super().__init__(cfg)
        self.return_vecs = cfg.return_vecs
        # colbert.py:44 loads by name; an already-built encoder can be injected (offline use/tests)
        self.bert_model = bert_model if bert_model is not None else AutoModel.from_pretrained(cfg.bert_model)
        for p in self.bert_model.parameters():
            p.requires_grad = cfg.trainable
        self._dropout = nn.Dropout(p=cfg.dropout)
        self.compressor = nn.Linear(self.bert_model.config.hidden_size, cfg.compression_dim)

    # ------------------------------------------------------------------ scoring (the hot path)
    @staticmethod
    def _score(query_vecs, document_vecs, query_mask, document_mask):
        """colbert.py:68-75, in the arithmetic the reference's eager ops have in the current autocast state: fp16 similarities
        and maxima, fp32 sum under `use_fp16` (defaults.yaml:21); fp32 throughout for fp32 vectors outside autocast.
        (_as_reference_would() spelled out: eval.py issues thousands of 512-pair calls whose device time is ~4 us, so every
        microsecond of Python on this path is visible.)"""
        ac = torch.is_autocast_enabled("cuda")
        q, d = query_vecs, document_vecs
        if q.dtype is torch.float32:
            if ac:
                adt = torch.get_autocast_dtype("cuda")
                q, d = q.to(adt), d.to(adt)
            sim_round, sum_round = ac, False
        else:
            if d.dtype is not q.dtype:
                d = d.to(q.dtype)
            sim_round, sum_round = True, not ac
        # The host extension (csrc_host/mm_autograd.cpp: the same C-ABI call from C++, with a C++ autograd node when a gradient
        # is needed) takes the call whenever it was built and the rows need no padding: at batch_size_train 32 x 2 the Python
        # node's apply + backward were 2/3 of the training step (train.py:347-348, :503-524: 147 -> 71 us), and eval.py's
        # 512-pair calls are bound by the host too (ops.maxsim's Python: ~10 us around ~4 us of HBM time).  Otherwise the
        # Python paths below — the same kernels, the same bits.
        fast = _fast.module()
        if fast is not None and q.is_cuda and q.dim() == 3 and d.dim() == 3 and q.shape[0] == d.shape[0] and q.shape[-1] % 8 == 0:
            score = fast.maxsim_paired(q, d, query_mask, document_mask, (1 if sim_round else 0) | (2 if sum_round else 0))
        elif torch.is_grad_enabled() and (q.requires_grad or d.requires_grad):
            score = _MaxSimFn.apply(q, d, query_mask, document_mask, sim_round, sum_round)
        else:
            score = ops.maxsim(q, d, query_mask, document_mask, 1, sim_round, sum_round)
        return score.to(q.dtype) if sum_round else score      # (16-bit tensors outside autocast: `sum` returns their dtype)

    @staticmethod
    def score_batches(batches):
        """The scoring block of SEVERAL forward calls at once: `batches` = [(query_vecs, document_vecs, query_mask, document_mask),
        ...] as _score takes them, all of one shape.  One launch for the group (ops.maxsim_batched) where the pair-per-row
        kernel takes the shape — eval.py's 512-pair batches (defaults.yaml:115) are 3.6 us of HBM time behind a ~9 us launch
        chain each — else one _score per batch.  No autograd (evaluation).  Returns the list of score tensors, bit-equal to
        [_score(*b) for b in batches]."""
        batches = list(batches)
        if len(batches) < 2:
            return [ColBERT._score(*b) for b in batches]
        ac = torch.is_autocast_enabled("cuda")
        prepared, flags = [], None
        for q, d, qm, dm in batches:
            if q.dtype is torch.float32:
                if not ac:
                    return [ColBERT._score(*b) for b in batches]          # fp32 scoring: the split-bf16 kernels, call by call
                adt = torch.get_autocast_dtype("cuda")
                q, d = q.to(adt), d.to(adt)
                f = (ac, False)
            else:
                if d.dtype is not q.dtype:
                    d = d.to(q.dtype)
                f = (True, not ac)
            if flags is None:
                flags = f
            same = (f == flags and q.shape[1:] == prepared[0][0].shape[1:] and d.shape[1:] == prepared[0][1].shape[1:]
                    and q.dtype == prepared[0][0].dtype) if prepared else True
            if not same or q.dim() != 3 or q.shape[0] != d.shape[0] or (qm is None) != (dm is None) \
                    or (qm is not None and (qm.dtype != torch.int64 or dm.dtype != torch.int64)):
                return [ColBERT._score(*b) for b in batches]
            prepared.append((q, d, qm, dm))
        try:
            with torch.no_grad():
                scores = ops.maxsim_batched(prepared, sim_round=flags[0], sum_round=flags[1])
        except ops.NativeError:                                           # a shape the pair-per-row kernel does not take
            return [ColBERT._score(*b) for b in batches]
        return [s.to(p[0].dtype) if flags[1] else s for s, p in zip(scores, prepared)]

    def forward(self, query: Dict[str, torch.LongTensor], document: Dict[str, torch.LongTensor],
                use_fp16: bool = True, output_secondary_output: bool = False):
        """colbert.py:54-86 — same arguments and return conventions."""
        with torch.autocast("cuda", dtype=torch.float16, enabled=use_fp16):
            query_vecs = self.forward_representation(query)
            document_vecs = self.forward_representation(document)
            score = self._score(query_vecs, document_vecs, query["attention_mask"], document["attention_mask"])

            if self.is_teacher_model:
                return (score, query_vecs, document_vecs)
            if self.return_vecs:
                score = (score, query_vecs, document_vecs)
            if output_secondary_output:
                return score, {}
            return score

    def forward_representation(self, tokens: Dict[str, torch.LongTensor], sequence_type=None) -> torch.Tensor:
        """colbert.py:88-98 (encoder + compressor stay PyTorch)."""
        vecs = self.bert_model(**tokens)[0]
        vecs = self.compressor(vecs)
        if sequence_type == "doc_encode" or sequence_type == "query_encode":
            vecs = vecs * tokens["attention_mask"].unsqueeze(-1)
        return vecs

    def forward_aggregation(self, query_vecs, document_vecs):
        """colbert.py:100-112 — unmasked MaxSim over pre-encoded vectors (QuerySearcherHead calls it under autocast,
        indexing_heads.py:49-56: fp16 similarities, fp32 sum)."""
        q, d, sim_round, sum_round = _as_reference_would(query_vecs, document_vecs)
        score = ops.maxsim(q, d, None, None, pairs_per_query=1, sim_round=sim_round, sum_round=sum_round)
        return score.to(q.dtype) if sum_round else score

    def forward_inbatch_aggregation(self, query_vecs, query_mask, document_vecs, document_mask):
        """colbert.py:114-162 — all-pairs MaxSim [Bq, Bd]."""
        if self.inbatch_bug_compatible and query_vecs.shape[0] != document_vecs.shape[0]:
            # the reference's mask expansion (:158) raises for Bq != Bd
            raise RuntimeError("forward_inbatch_aggregation (reference-compatible masking) needs the same number "
                               "of queries and documents; set inbatch_bug_compatible = False for the general case")
        # (the dynamic teacher calls this OUTSIDE autocast on the fp16 vectors its forward returned, dynamic_teacher.py:245-246:
        # `mm`, `max` and `sum` are all fp16 ops there, and so is the result)
        q, d, sim_round, sum_round = _as_reference_would(query_vecs, document_vecs)
        score = ops.maxsim_inbatch(q, query_mask, d, document_mask, bug_compatible=self.inbatch_bug_compatible,
                                   sim_round=sim_round, sum_round=sum_round)
        return score.to(q.dtype) if sum_round else score

    def get_param_stats(self):            # colbert.py:164-165
        return "ColBERT: / "

    def get_param_secondary(self):        # colbert.py:166-167
        return {}
