"""Drop-in TKL (TKL_sigir20) for matchmaker: constructor, from_config, forward signature and
parameter names (= state_dict keys) as in matchmaker/models/published/sigir20_tkl.py.

Stays PyTorch, as in the reference: query contextualisation (:139), document chunking into
50-token windows with 5 tokens of overlap on each side, dropping of all-padding chunks and the chunk
Transformer (:142-175).  Everything after that — cosine match, RBF kernels, sliding-window pooling,
saturation, dense layer, top-3 region scoring (:180-286) — runs in libmm_native.so (mm_tkl_fwd).

Training (train.py:347-348, loss.backward() :503-524): the document score is a weighted sum of at most 15
window scores (3 regions x 5 neighbours, :257-286) and the region choice is piecewise constant, so the
exact gradient only involves those windows.  Forward AND backward are native: `_TKLScoreFn` wraps mm_tkl_fwd /
mm_tkl_bwd (the backward kernel recomputes the 15 selected windows of each document — 30 tokens each, out of up
to 2,000 — and differentiates them: gradients w.r.t. the contextualised query, the contextualised chunk rows and
every scoring parameter).  (The same computation in differentiable torch ops, the reference the native backward is tested
against, lives with the tests: tests/tkl_window_reference.py.)
"""
import os
from typing import List

import torch
import torch.nn as nn

from . import _fast, ops
from ._lib import NativeError
from .tk import sinusoid_positions

CHUNK = 40          # sigir20_tkl.py:52
OVERLAP = 5         # :53
EXT_CHUNK = 50      # :54
WINDOW = 30         # :56
TOP_K = 3           # :57


def chunk_documents(document_embeddings: torch.Tensor, document_mask: torch.Tensor):
    """sigir20_tkl.py:142-162.  Pads the document by OVERLAP on the left and up to a whole number of
    chunks (+OVERLAP) on the right, cuts it into C windows of 50 tokens every 40 tokens, and keeps
    the windows whose 40 centre tokens are not all padding.

    Returns packed chunks [P,50,E], their masks [P,50], chunk_slot [P] (flat b*C + c) and C."""
    B, D, E = document_embeddings.shape
    if D > OVERLAP:
        right = EXT_CHUNK - ((D - OVERLAP) % CHUNK)
    else:
        right = EXT_CHUNK - OVERLAP - D
    emb = nn.functional.pad(document_embeddings, (0, 0, OVERLAP, right))
    msk = nn.functional.pad(document_mask, (OVERLAP, right))
    C = (emb.shape[1] - EXT_CHUNK) // CHUNK + 1
    win = torch.arange(C, device=emb.device).unsqueeze(1) * CHUNK + torch.arange(EXT_CHUNK, device=emb.device)
    chunk_mask = msk[:, win].reshape(B * C, EXT_CHUNK)
    keep = chunk_mask[:, OVERLAP:EXT_CHUNK - OVERLAP].sum(-1) != 0
    chunk_slot = keep.nonzero(as_tuple=False).squeeze(1)
    flat = (chunk_slot // C).unsqueeze(1) * emb.shape[1] + (chunk_slot % C).unsqueeze(1) * CHUNK + \
        torch.arange(EXT_CHUNK, device=emb.device)
    chunks = emb.reshape(-1, E)[flat]                       # gather only the kept chunks
    return chunks, chunk_mask[chunk_slot], chunk_slot.to(torch.int32), C


def region_peaks(win: torch.Tensor) -> torch.Tensor:
    """sigir20_tkl.py:254-273 on window scores [B, W]: 0 -> -9900, three arg-max rounds with |r - best| < 15 suppression.
    int64 [B, 3].  The kernels do this search themselves (mm_tkl_fwd_peaks returns its result); this torch form serves the
    autograd path's secondary output and the tests that pin the kernel's indices (same fp32 values, same first-index ties)."""
    score = win
    if score.shape[1] < TOP_K:
        score = nn.functional.pad(score, (0, TOP_K - score.shape[1]))
    work = torch.where(score == 0, torch.full_like(score, -9900.0), score)
    r = torch.arange(work.shape[1], device=work.device)
    peaks = torch.zeros((work.shape[0], TOP_K), dtype=torch.long, device=work.device)
    for c in range(TOP_K):
        best = torch.argmax(work, dim=1)
        peaks[:, c] = best
        work = torch.where((r - best.unsqueeze(-1)).abs() < WINDOW / 2, torch.full_like(work, -10001.0 - c), work)
    return peaks


def region_neighbors(peaks: torch.Tensor, W: int) -> torch.Tensor:
    """:276-278 — the 15 window indices of a document: peaks, -1, +1, -2, +2, clamped to [0, W - 1]."""
    return torch.cat([peaks, peaks - 1, peaks + 1, peaks - 2, peaks + 2], dim=1).clamp_(0, max(W, TOP_K) - 1)


class _TKLScoreFn(torch.autograd.Function):
    """mm_tkl_fwd / mm_tkl_bwd behind autograd.  `scoring` are the parameter tensors in pack order (see
    TKL_sigir20._pack): their gradients come back as slices of the packed gradient vector."""

    @staticmethod
    def forward(ctx, q_ctx, chunks_ctx, chunk_mask, chunk_slot, q_mask, meta, *scoring):
        B, C, K, saturation, packed, sizes = meta
        score, win = ops.tkl_score(q_ctx, chunks_ctx, chunk_mask, chunk_slot, q_mask, packed, B, C, K, saturation,
                                   return_windows=True, check_order=False)      # chunk_documents() packs in ascending order
        ctx.save_for_backward(q_ctx, chunks_ctx, chunk_mask, chunk_slot, q_mask, packed, win)
        ctx.meta = (B, C, K, saturation, sizes, [t.shape for t in scoring], [t.dtype for t in scoring])
        ctx.mark_non_differentiable(win)
        return score, win

    @staticmethod
    def backward(ctx, g, _gwin):
        q_ctx, chunks_ctx, chunk_mask, chunk_slot, q_mask, packed, win = ctx.saved_tensors
        B, C, K, saturation, sizes, shapes, dtypes = ctx.meta
        gq, gc, gp = ops.tkl_bwd(q_ctx, chunks_ctx, chunk_mask, chunk_slot, q_mask, packed, win, g, B, C, K, saturation)
        grads, off = [], 0
        for n, shp, dt in zip(sizes, shapes, dtypes):
            if n is None:                       # a parameter the kernels do not read in this configuration
                grads.append(None)
                continue
            lo, cnt, full = n
            gpar = gp[lo:lo + cnt]
            if full != cnt:                     # kernel_mult [4,1,1,1,K]: only row 0 is used (:246)
                z = torch.zeros(full, dtype=gpar.dtype, device=gpar.device)
                z[:cnt] = gpar
                gpar = z
            grads.append(gpar.reshape(shp).to(dt))
        return (gq.to(q_ctx.dtype), gc.to(chunks_ctx.dtype), None, None, None, None, *grads)


def tkl_score_train(q_ctx, chunks_ctx, chunk_mask, chunk_slot, q_mask, packed, B, C, K, saturation, scoring, sizes, layout=None):
    """(score, window scores) of mm_tkl_fwd with mm_tkl_bwd behind autograd: the C++ node of csrc_host/mm_autograd.cpp (TklScore)
    when the host extension is built, the Python autograd.Function above otherwise (MM_TKL_PY_AUTOGRAD=1 forces it: A/B runs, tests).
    `layout`: the cached host tensor of _layout_tensor(sizes)."""
    fast = None if os.environ.get("MM_TKL_PY_AUTOGRAD", "0") not in ("", "0") else _fast.module()
    if fast is not None and hasattr(fast, "tkl_score") and q_ctx.shape[-1] % 4 == 0:
        if layout is None:
            layout = _layout_tensor(sizes)
        return fast.tkl_score(q_ctx, chunks_ctx, chunk_mask, chunk_slot, q_mask, packed, B, C, K,
                              0 if saturation == "embedding" else 1, list(scoring), layout)
    return _TKLScoreFn.apply(q_ctx, chunks_ctx, chunk_mask, chunk_slot, q_mask, (B, C, K, saturation, packed, sizes), *scoring)


def _layout_tensor(sizes):
    """(offset in the packed gradient, elements the kernels read, numel) per scoring tensor as ONE host int64 [3, n] tensor
    (offset -1: the active saturation never reads the tensor — no gradient)."""
    return torch.tensor([[-1 if n is None else n[0] for n in sizes], [0 if n is None else n[1] for n in sizes],
                         [0 if n is None else n[2] for n in sizes]], dtype=torch.int64)


class TKL_sigir20(nn.Module):
    """TKL: TK for long documents (https://arxiv.org/abs/2005.04908)."""

    @staticmethod
    def from_config(config, word_embeddings_out_dim):      # sigir20_tkl.py:17-29
        if len(config["tk_kernels_mu"]) != 11:
            raise NativeError(f"TKL: the native window kernels are built for the reference's 11 RBF kernels (tkl.yaml), "
                              f"got {len(config['tk_kernels_mu'])}")
        if word_embeddings_out_dim % 4:
            raise NativeError(f"TKL: embedding width {word_embeddings_out_dim} is not a multiple of 4 floats (16-byte rows)")
        return TKL_sigir20(word_embeddings_out_dim,
                           kernels_mu=config["tk_kernels_mu"],
                           kernels_sigma=config["tk_kernels_sigma"],
                           att_heads=config["tk_att_heads"],
                           att_layer=config["tk_att_layer"],
                           att_ff_dim=config["tk_att_ff_dim"],
                           max_length=config["max_doc_length"],
                           use_pos_encoding=config["tk_use_pos_encoding"],
                           use_diff_posencoding=config["tk_use_diff_posencoding"],
                           saturation_type=config["tk_saturation_type"])

    def __init__(self, _embsize: int, kernels_mu: List[float], kernels_sigma: List[float], att_heads: int,
                 att_layer: int, att_ff_dim: int, max_length, use_pos_encoding, use_diff_posencoding, saturation_type):
        super().__init__()
        if len(kernels_mu) != len(kernels_sigma):
            raise Exception("len(kernels_mu) != len(kernels_sigma)")
        n_kernels = len(kernels_mu)
        self.use_pos_encoding = use_pos_encoding
        self.use_diff_posencoding = use_diff_posencoding
        self.re_use_encoding = True
        self.chunk_size = CHUNK
        self.overlap = OVERLAP
        self.extended_chunk_size = EXT_CHUNK
        self.sliding_window_size = WINDOW
        self.top_k_chunks = TOP_K
        self.saturation_type = saturation_type
        self.use_idf_sat = saturation_type == "idf"
        self.use_embedding_sat = saturation_type == "embedding"
        self.use_linear_sat = saturation_type == "linear"
        self.use_log_sat = saturation_type == "log"

        # the reference builds these with torch.cuda.FloatTensor (:68-69); plain tensors move with .to()
        self.mu = nn.Parameter(torch.tensor(kernels_mu, dtype=torch.float32), requires_grad=False)
        self.sigma = nn.Parameter(torch.tensor(kernels_sigma, dtype=torch.float32), requires_grad=False)
        self.positional_features_q = nn.Parameter(sinusoid_positions(_embsize, 30))
        if use_diff_posencoding:
            self.positional_features_d = nn.Parameter(
                sinusoid_positions(_embsize, 2000 + 500 + EXT_CHUNK)[:, 500:, :].clone())
        else:
            self.positional_features_d = self.positional_features_q
        self.mixer = nn.Parameter(torch.full([1], 0.5, dtype=torch.float32, requires_grad=True))
        self.mixer_sat = nn.Parameter(torch.full([1], 0.5, dtype=torch.float32, requires_grad=True))
        layer = nn.TransformerEncoderLayer(_embsize, att_heads, dim_feedforward=att_ff_dim, dropout=0)
        self.contextualizer = nn.TransformerEncoder(layer, att_layer, norm=None)

        def sat_linear():
            lin = nn.Linear(2, 1, bias=True)
            torch.nn.init.constant_(lin.bias, 100)
            torch.nn.init.uniform_(lin.weight, -0.014, 0.014)
            return lin
        self.saturation_linear = sat_linear()
        self.saturation_linear2 = sat_linear()
        self.saturation_linear3 = sat_linear()
        self.sat_normer = nn.LayerNorm(2, elementwise_affine=True)
        self.sat_emb_reduce1 = nn.Linear(_embsize, 1, bias=False)
        self.kernel_mult = nn.Parameter(torch.full([4, 1, 1, 1, n_kernels], 1, dtype=torch.float32, requires_grad=True))
        self.chunk_scoring = nn.Parameter(torch.full([1, TOP_K * 5], 1, dtype=torch.float32, requires_grad=True))
        self.mixer_end = nn.Parameter(torch.full([1], 0.5, dtype=torch.float32, requires_grad=True))
        self.dense = nn.Linear(n_kernels, 1, bias=False)
        torch.nn.init.uniform_(self.dense.weight, -0.014, 0.014)
        self._packed = None
        self._layout, self._layout_key = None, None      # (lo, cnt, full) of the scoring tensors for the C++ node

    # ------------------------------------------------------------------ native parameter vector
    def pack_params(self) -> torch.Tensor:
        """float32 vector in the layout of MM_TKL_NPARAMS (include/mm_native.h); cached until any
        source parameter is modified in place or replaced."""
        src = [self.mu, self.sigma, self.dense.weight, self.kernel_mult,
               self.saturation_linear.weight, self.saturation_linear.bias,
               self.saturation_linear2.weight, self.saturation_linear2.bias,
               self.saturation_linear3.weight, self.saturation_linear3.bias,
               self.sat_normer.weight, self.sat_normer.bias, self.chunk_scoring, self.sat_emb_reduce1.weight]
        key = tuple((t.data_ptr(), t._version, t.device) for t in src)
        if self._packed is None or self._packed[0] != key:
            f = lambda t: t.detach().reshape(-1).float()
            vec = torch.cat([f(self.mu), f(self.sigma), f(self.dense.weight), f(self.kernel_mult[0]),
                             f(self.saturation_linear.weight), f(self.saturation_linear.bias),
                             f(self.saturation_linear2.weight), f(self.saturation_linear2.bias),
                             f(self.saturation_linear3.weight), f(self.saturation_linear3.bias),
                             f(self.sat_normer.weight), f(self.sat_normer.bias), f(self.chunk_scoring),
                             f(self.sat_emb_reduce1.weight)]).contiguous()
            self._packed = (key, vec)
        return self._packed[1]

    def forward(self, query_embeddings: torch.Tensor, document_embeddings: torch.Tensor,
                query_pad_oov_mask: torch.Tensor, document_pad_oov_mask: torch.Tensor,
                output_secondary_output: bool = False):
        """sigir20_tkl.py:128-294 — same arguments; returns score [B] (or (score, dict))."""
        if not (self.use_embedding_sat or self.use_log_sat):
            raise NativeError(f"tk_saturation_type={self.saturation_type!r}: the reference's idf/linear branches read "
                              "`query_idfs`, which forward() does not receive (sigir20_tkl.py:130,214,236)")
        B, Q = query_embeddings.shape[0], query_embeddings.shape[1]
        query_ctx, _ = self.forward_representation(query_embeddings, query_pad_oov_mask,
                                                   self.positional_features_q[:, :Q, :])
        chunks, chunk_mask, chunk_slot, C = chunk_documents(document_embeddings, document_pad_oov_mask)
        if self.re_use_encoding:
            pos = self.positional_features_d[:, :chunks.shape[1], :]
        else:   # not reachable with the reference's constructor (:50) but kept for attribute parity
            raise NativeError("re_use_encoding=False is not supported")
        if chunks.shape[0] > 0:
            chunks_ctx, _ = self.forward_representation(chunks, chunk_mask, pos)
        else:
            chunks_ctx = chunks
        K = self.mu.numel()
        saturation = "embedding" if self.use_embedding_sat else "log"
        if torch.is_grad_enabled() and (query_ctx.requires_grad or chunks_ctx.requires_grad or
                                        any(p.requires_grad for p in self._scoring_parameters())):
            scoring, sizes = self._pack_layout()
            if self._layout_key != tuple(sizes):
                self._layout, self._layout_key = _layout_tensor(sizes), tuple(sizes)
            score, win = tkl_score_train(query_ctx.float(), chunks_ctx.float(), chunk_mask, chunk_slot, query_pad_oov_mask,
                                         self.pack_params(), B, C, K, saturation, scoring, sizes, self._layout)
            peaks = region_peaks(win.detach()) if output_secondary_output else None
        else:
            with torch.no_grad():
                score, win, peaks = ops.tkl_score(query_ctx.float(), chunks_ctx.float(), chunk_mask, chunk_slot, query_pad_oov_mask,
                                                  self.pack_params(), B, C, K, saturation, return_windows=True, check_order=False,
                                                  return_peaks=True)
        if output_secondary_output:
            return score, self._secondary(score, win, peaks, query_ctx, query_pad_oov_mask, document_pad_oov_mask, B, C,
                                          int(chunks.shape[0]))
        return score

    @torch.no_grad()
    def _secondary(self, score, win, peaks, query_ctx, query_mask, document_mask, B, C, P):
        """sigir20_tkl.py:288-292 — the same keys.  top_non_overlapping_idx is the region search's own result (the kernel
        writes it); top_k_non_overlapping is a gather of 15 of the kernel's window scores per document (empty windows are 0
        there, as after :282); sat_influence_from_top_k is the saturation layer's input (:225-228) at those 15 windows:
        LayerNorm2([sat_emb_reduce1(q), lengths]).  `lengths` (:210) counts the window positions whose kernel activations are
        not all zero — the unmasked document positions (an unmasked position always has a kernel within 0.1 of its cosine)."""
        W = win.shape[1]
        neighbors = region_neighbors(peaks, W)                                                  # :276-278
        sec = {"score": score, "orig_score": win, "top_non_overlapping_idx": peaks,
               "orig_doc_len": document_mask.sum(dim=-1), "top_k_non_overlapping": win.gather(1, neighbors),
               "total_chunks": B * C, "packed_chunks": P,
               "query_mean_vector": query_ctx.sum(dim=1) / query_mask.sum(dim=1).unsqueeze(-1)}
        if self.use_embedding_sat:    # (the reference's log branch never defines sat_influencer: NameError at :290)
            D = document_mask.shape[1]
            right = EXT_CHUNK - ((D - OVERLAP) % CHUNK) if D > OVERLAP else EXT_CHUNK - OVERLAP - D
            pos = nn.functional.pad(document_mask.float(), (0, right - OVERLAP))[:, :C * CHUNK]   # centre tokens of all chunks
            pos = nn.functional.pad(pos, (0, max(0, WINDOW - pos.shape[1])))                      # :206-207
            cs = nn.functional.pad(pos.cumsum(1), (1, 0))
            lengths = cs.gather(1, 2 * neighbors + WINDOW) - cs.gather(1, 2 * neighbors)          # [B, 15]  (:209-210)
            emb = self.sat_emb_reduce1(query_ctx.float()).squeeze(-1)                             # [B, Q]   (:224)
            Q = emb.shape[1]
            infl = torch.stack([emb.unsqueeze(1).expand(-1, neighbors.shape[1], -1),
                                lengths.unsqueeze(-1).expand(-1, -1, Q)], dim=-1)                 # [B, 15, Q, 2] (:225-227)
            sec["sat_influence_from_top_k"] = self.sat_normer(infl)                               # :228, :290
        return sec

    # ------------------------------------------------------------------ training
    def _pack_layout(self):
        """The parameter tensors in the order of the packed vector (pack_params) with (offset, used length, numel) per
        tensor; None for tensors the native scoring does not read (mu / sigma are constants)."""
        K, E = self.mu.numel(), self.sat_emb_reduce1.weight.numel()
        o_dense, o_km, o_sat, o_cs, o_emb = 2 * K, 3 * K, 4 * K, 4 * K + 13, 4 * K + 28
        scoring = [self.dense.weight, self.kernel_mult,
                   self.saturation_linear.weight, self.saturation_linear.bias,
                   self.saturation_linear2.weight, self.saturation_linear2.bias,
                   self.saturation_linear3.weight, self.saturation_linear3.bias,
                   self.sat_normer.weight, self.sat_normer.bias, self.chunk_scoring, self.sat_emb_reduce1.weight]
        sizes = [(o_dense, K, K), (o_km, K, self.kernel_mult.numel()),
                 (o_sat + 0, 2, 2), (o_sat + 2, 1, 1), (o_sat + 3, 2, 2), (o_sat + 5, 1, 1), (o_sat + 6, 2, 2), (o_sat + 8, 1, 1),
                 (o_sat + 9, 2, 2), (o_sat + 11, 2, 2), (o_cs, 15, 15), (o_emb, E, E)]
        # parameters the active saturation never reads keep .grad = None as in the reference (optimizers skip them, DDP's
        # unused-parameter bookkeeping matches): kernel_mult under "embedding", the saturation layers under "log"
        live = {id(p) for p in self._scoring_parameters()}
        sizes = [n if id(t) in live else None for t, n in zip(scoring, sizes)]
        return scoring, sizes

    def _scoring_parameters(self):
        ps = [self.dense.weight, self.chunk_scoring]
        if self.use_embedding_sat:
            ps += [self.saturation_linear.weight, self.saturation_linear.bias, self.saturation_linear2.weight,
                   self.saturation_linear2.bias, self.saturation_linear3.weight, self.saturation_linear3.bias,
                   self.sat_normer.weight, self.sat_normer.bias, self.sat_emb_reduce1.weight]
        else:
            ps.append(self.kernel_mult)
        return ps

    def forward_representation(self, sequence_embeddings: torch.Tensor, sequence_mask: torch.Tensor,
                               positional_features=None):
        """sigir20_tkl.py:296-308 — returns (mixed * mask, contextualised)."""
        pos_sequence = sequence_embeddings
        if self.use_pos_encoding:
            if positional_features is None:
                positional_features = self.positional_features_d[:, :sequence_embeddings.shape[1], :]
            pos_sequence = sequence_embeddings + positional_features
        ctx = self.contextualizer(pos_sequence.transpose(1, 0),
                                  src_key_padding_mask=~sequence_mask.bool()).transpose(1, 0)
        mixed = (self.mixer * sequence_embeddings + (1 - self.mixer) * ctx) * sequence_mask.unsqueeze(-1)
        return mixed, ctx

    def get_param_stats(self):            # sigir20_tkl.py:375-382
        return "TK: dense w: " + str(self.dense.weight.data) + \
            " self.chunk_scoring: " + str(self.chunk_scoring.data) + \
            " self.kernel_mult: " + str(self.kernel_mult.data) + \
            " self.saturation_linear: " + str(self.saturation_linear.weight.data) + " bias: " + str(self.saturation_linear.bias.data) + \
            " self.saturation_linear2: " + str(self.saturation_linear2.weight.data) + " bias: " + str(self.saturation_linear2.bias.data) + \
            " self.saturation_linear3: " + str(self.saturation_linear3.weight.data) + " bias: " + str(self.saturation_linear3.bias.data) + \
            "mixer: " + str(self.mixer.data)

    def get_param_secondary(self):        # sigir20_tkl.py:384-393
        return {"dense_weight": self.dense.weight,
                "saturation_linear_weight": self.saturation_linear.weight,
                "saturation_linear_bias": self.saturation_linear.bias,
                "saturation_linear2_weight": self.saturation_linear2.weight,
                "saturation_linear2_bias": self.saturation_linear2.bias,
                "saturation_linear3_weight": self.saturation_linear3.weight,
                "saturation_linear3_bias": self.saturation_linear3.bias,
                "chunk_scoring": self.chunk_scoring,
                "kernel_mult": self.kernel_mult,
                "mixer": self.mixer}
