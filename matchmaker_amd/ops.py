"""Host side of the scoring operators: torch tensors in, torch tensors out, arithmetic in
libmm_native.so (hand-written HIP, gfx950).  torch is used only for device memory and streams.

Every function launches on the *current* torch stream of the tensors' device, allocates only its
output (+ a small mask-packing workspace from torch's caching allocator), never synchronises and
keeps no state, so it is re-entrant from nn.DataParallel's per-GPU threads
(matchmaker/train.py:201).  CPU tensors are rejected: there is no CPU fallback.
"""
import ctypes
import threading
from typing import Optional

import torch

from . import _lib
from ._lib import NativeError

_DT = {torch.float32: _lib.MM_F32, torch.float16: _lib.MM_F16, torch.bfloat16: _lib.MM_BF16}


def _dev_check(*ts):
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise NativeError("matchmaker_amd operators need HIP device tensors (got a CPU tensor); "
                              "there is no CPU fallback")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise NativeError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


def _emb(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype not in _DT:
        raise NativeError(f"{name}: unsupported dtype {t.dtype}")
    if t.dim() != 3:
        raise NativeError(f"{name}: expected [rows, tokens, dim], got {tuple(t.shape)}")
    return t if t.is_contiguous() else t.contiguous()


def _pad_rows(q: torch.Tensor, d: torch.Tensor, mult: int):
    """Token rows must be 16-byte multiples for the native loads.  Zero columns change neither dot products nor
    norms, so other widths (KNRM on 50-d GloVe, colbert_compression_dim 100, ...) are padded up — a copy, taken
    only for such widths.  Returns (q, d, padded E)."""
    E = q.shape[-1]
    Ep = (E + mult - 1) // mult * mult
    if Ep == E:
        return q, d, E
    pad = (0, Ep - E)
    return torch.nn.functional.pad(q, pad), torch.nn.functional.pad(d, pad), Ep


def _mask(m: Optional[torch.Tensor], rows: int, L: int, name: str):
    """-> (tensor kept alive, pointer, kind)"""
    if m is None:
        return None, None, _lib.MASK_NONE
    if m.dim() == 1:
        if m.shape[0] != rows:
            raise NativeError(f"{name}: expected {rows} lengths, got {tuple(m.shape)}")
        m = m.to(torch.int32).contiguous()
        return m, m.data_ptr(), _lib.MASK_LEN_I32
    if tuple(m.shape) != (rows, L):
        raise NativeError(f"{name}: expected [{rows}, {L}], got {tuple(m.shape)}")
    if m.dtype == torch.int64:
        kind = _lib.MASK_I64
    elif m.dtype == torch.float32:
        kind = _lib.MASK_F32
    elif m.dtype in (torch.uint8, torch.bool):
        kind = _lib.MASK_U8
    else:
        m = (m != 0)
        kind = _lib.MASK_U8
    m = m.contiguous()
    return m, m.data_ptr(), kind


# The current stream's raw handle without building a torch.cuda.Stream object (~0.3 us instead of ~2.5: two of those per
# call were a fifth of the host time of a 512-pair call).  torch._C._cuda_getCurrentRawStream is what torch's own
# compiled-code launchers use; the public path stays as the fallback.
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(dev) -> int:
    if _RAW_STREAM is not None and dev.index is not None:
        return _RAW_STREAM(dev.index)
    return torch.cuda.current_stream(dev).cuda_stream


# Mask-packing workspaces of the scoring calls, one per (device, stream): calls on one stream are ordered, so the
# next call may reuse the buffer; eval.py-sized calls (512 pairs) are host-bound and a torch.empty per call is ~2 us
# of their ~15.  Not used under graph capture (a captured graph must not reference a buffer a later call may replace).
_WS = {}
_WS_LOCK = threading.Lock()      # nn.DataParallel drives forward() from one Python thread per GPU (train.py:201)


def _workspace(dev, nbytes: int, stream: Optional[int] = None):
    if nbytes == 0:
        return None
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(nbytes, dtype=torch.uint8, device=dev)
    key = (dev.index, _stream(dev) if stream is None else stream)
    t = _WS.get(key)
    if t is None or t.numel() < nbytes:
        with _WS_LOCK:
            if len(_WS) > 64:
                _WS.clear()
            t = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=dev)
            _WS[key] = t
    return t


def clear_workspaces() -> None:
    """Drops the cached mask-packing workspaces (one per (device, stream) that ever called a scoring operator, sized for the
    largest call seen there, at most 64 of them): they are ordinary torch allocations, so torch.cuda.empty_cache() can
    return their memory afterwards.  Safe at any time — a call in flight holds its own reference."""
    with _WS_LOCK:
        _WS.clear()


class _NoCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NOCTX = _NoCtx()


def _on(dev):
    """Device guard for the native call: torch's context manager costs ~2-3 us per call, which is visible on 512-pair
    calls; nothing to guard when `dev` already is the current device (the single-GPU-per-process case)."""
    return _NOCTX if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)


_WSB = {}


def _ws_bytes(fn, *key):
    """Workspace size of a call shape (a pure function of the integers in `key`): one ctypes round trip per shape, not per call."""
    k = (fn.__name__,) + key
    v = _WSB.get(k)
    if v is None:
        if len(_WSB) > 4096:
            _WSB.clear()
        v = _WSB[k] = fn(*key)
    return v


def _vec(t: torch.Tensor) -> torch.Tensor:
    """A small parameter tensor as float32 contiguous memory.  Only its data pointer and element count are used, so a
    contiguous fp32 tensor of any shape is taken as it is — the models' own buffers / parameters (`mu` [1,1,1,K],
    `kernel_alpha_scaler` [1,1,K], `kernel_bin_weights.weight` [1,K]) cost no dispatch per call."""
    if t.dtype is torch.float32 and t.is_contiguous():
        return t
    return t.detach().reshape(-1).to(torch.float32).contiguous()


def _flags(sim_round: bool, sum_round: bool) -> int:
    return (_lib.SIM_ROUND if sim_round else 0) | (_lib.SUM_ROUND if sum_round else 0)


def reference_rounding(q: torch.Tensor) -> "tuple[bool, bool]":
    """(sim_round, sum_round) that reproduce what the reference's eager ops do with token vectors of q's dtype in the
    CURRENT autocast state: 16-bit vectors give a 16-bit similarity matrix (`bmm` / `mm`, the -1000 fill and `max`,
    colbert.py:68-71); `sum` (:75) is promoted to fp32 under autocast and is a 16-bit op outside it (the dynamic
    teacher's all-pairs call, dynamic_teacher.py:245-246).  fp32 vectors outside autocast: no rounding anywhere."""
    ac = torch.is_autocast_enabled("cuda")
    lowp = q.dtype in (torch.float16, torch.bfloat16)
    return (lowp or ac), (lowp and not ac)


def maxsim(q: torch.Tensor, d: torch.Tensor, q_mask: Optional[torch.Tensor] = None,
           d_mask: Optional[torch.Tensor] = None, pairs_per_query: int = 1, sim_round: bool = False,
           sum_round: bool = False) -> torch.Tensor:
    """ColBERT MaxSim (matchmaker/models/colbert.py:68-75; unmasked: :100-112).

    q [n_queries, Q, E], d [n_pairs, D, E]; pair p scores against query p // pairs_per_query.
    Masks: None | 1-D lengths | [rows, L] bool/uint8/int64/float (nonzero = real token).
    sim_round / sum_round: the reference's dtype flow for fp16 / bf16 vectors (MM_SIM_ROUND / MM_SUM_ROUND in
    include/mm_native.h): per-token maxima rounded to the vectors' dtype before the fp32 sum (what autocast does,
    colbert.py:60-75), and the sum as well (16-bit tensors outside autocast).  Default: fp32 through max and sum.
    Returns float32 [n_pairs] (with sum_round the values are exactly representable in the vectors' dtype)."""
    dev = _dev_check(q, d, q_mask, d_mask)
    q, d = _emb(q, "q"), _emb(d, "d")
    if q.dtype != d.dtype:
        raise NativeError(f"q/d dtype mismatch: {q.dtype} vs {d.dtype}")
    nq, Q, E = q.shape
    B, D, E2 = d.shape
    if E != E2:
        raise NativeError(f"embedding dims differ: {E} vs {E2}")
    if pairs_per_query < 1 or nq != (B + pairs_per_query - 1) // pairs_per_query:
        raise NativeError(f"q has {nq} rows but {B} pairs / {pairs_per_query} per query")
    qm, qp, qk = _mask(q_mask, nq, Q, "q_mask")
    dm, dp, dk = _mask(d_mask, B, D, "d_mask")
    L = _lib.lib()
    out = torch.empty(B, dtype=torch.float32, device=dev)
    if B == 0:
        return out
    q, d, E = _pad_rows(q, d, 4 if q.dtype == torch.float32 else 8)
    with _on(dev):
        wsb = _ws_bytes(L.mm_maxsim_workspace_bytes, B, pairs_per_query, Q, D, qk, dk)
        st = _stream(dev)
        ws = _workspace(dev, wsb, st)
        rc = L.mm_maxsim_fwd(q.data_ptr(), d.data_ptr(), qp, qk, dp, dk, out.data_ptr(), B, pairs_per_query,
                             Q, D, E, _DT[q.dtype], (1 if sim_round else 0) | (2 if sum_round else 0),
                             ws.data_ptr() if ws is not None else None, wsb, st)
    _lib.check(rc, "mm_maxsim_fwd")
    return out


class _MaxsimBatch(ctypes.Structure):      # mm_maxsim_batch_t (include/mm_native.h)
    _fields_ = [("q", ctypes.c_void_p), ("d", ctypes.c_void_p), ("q_mask", ctypes.c_void_p), ("d_mask", ctypes.c_void_p),
                ("out", ctypes.c_void_p), ("n_pairs", ctypes.c_int64)]


MAXSIM_MAX_BATCHES = 16


def maxsim_batched(batches, sim_round: bool = False, sum_round: bool = False):
    """Several pair-per-row batches of ONE shape scored by one launch (mm_maxsim_fwd_batched): `batches` = a sequence of
    (q [B, Q, E], d [B, D, E], q_mask, d_mask) with 16-bit vectors and either int64 tokenizer masks ([B, Q] / [B, D]) or None for
    both, everywhere.  Returns the list of float32 score tensors [B] (views of one allocation), bit-equal to maxsim() on
    each batch.  Raises NativeError(MM_EUNSUPPORTED ...) for shapes the pair-per-row kernel does not take (callers fall back
    to one maxsim() per batch); more than MAXSIM_MAX_BATCHES batches go out in several launches."""
    batches = list(batches)
    if not batches:
        return []
    q0, d0, qm0, dm0 = batches[0]
    dev = _dev_check(*[t for b in batches for t in b])
    Q, E, D = q0.shape[1], q0.shape[2], d0.shape[1]
    has_masks = qm0 is not None
    total = sum(b[1].shape[0] for b in batches)
    out = torch.empty(total, dtype=torch.float32, device=dev)
    L = _lib.lib()
    keep, views, recs, off = [], [], [], 0
    for q, d, qm, dm in batches:
        q, d = _emb(q, "q"), _emb(d, "d")
        if q.dtype != q0.dtype or d.dtype != q0.dtype or q.dtype == torch.float32:
            raise NativeError("maxsim_batched: 16-bit vectors of one dtype in every batch")
        if q.shape[1:] != (Q, E) or d.shape[1:] != (D, E) or q.shape[0] != d.shape[0]:
            raise NativeError(f"maxsim_batched: every batch must be pair-per-row [B, {Q}, {E}] / [B, {D}, {E}], got {tuple(q.shape)} / {tuple(d.shape)}")
        if (qm is not None) != has_masks or (dm is not None) != has_masks:
            raise NativeError("maxsim_batched: masks for every batch or for none")
        B = d.shape[0]
        if has_masks:
            if qm.dtype != torch.int64 or dm.dtype != torch.int64 or tuple(qm.shape) != (B, Q) or tuple(dm.shape) != (B, D):
                raise NativeError("maxsim_batched: masks are the tokenizer's int64 [B, Q] / [B, D] tensors")
            qm, dm = qm.contiguous(), dm.contiguous()
        keep.append((q, d, qm, dm))
        o = out[off:off + B]
        views.append(o)
        recs.append(_MaxsimBatch(q.data_ptr(), d.data_ptr(), qm.data_ptr() if has_masks else None,
                                 dm.data_ptr() if has_masks else None, o.data_ptr(), B))
        off += B
    kind = _lib.MASK_I64 if has_masks else _lib.MASK_NONE
    with _on(dev):
        st = _stream(dev)
        for i in range(0, len(recs), MAXSIM_MAX_BATCHES):
            part = recs[i:i + MAXSIM_MAX_BATCHES]
            arr = (_MaxsimBatch * len(part))(*part)
            rc = L.mm_maxsim_fwd_batched(ctypes.cast(arr, ctypes.c_void_p), len(part), kind, kind, Q, D, E, _DT[q0.dtype],
                                         (1 if sim_round else 0) | (2 if sum_round else 0), st)
            _lib.check(rc, "mm_maxsim_fwd_batched")
    return views


def hbm_stream_probe(t: torch.Tensor, nt: bool = True) -> None:
    """Calibration launch (mm_hbm_stream_probe): streams tensor `t` through the MaxSim kernel's LDS-DMA ring with the
    arithmetic removed.  Returns nothing: it exists to be timed (bench.py extra.hbm_calibration)."""
    dev = _dev_check(t)
    t = t if t.is_contiguous() else t.contiguous()
    nbytes = (t.numel() * t.element_size()) // 8192 * 8192
    with _on(dev):
        rc = _lib.lib().mm_hbm_stream_probe(t.data_ptr(), nbytes, 1 if nt else 0, _stream(dev))
    _lib.check(rc, "mm_hbm_stream_probe")


def _check_ranges(doc_begin: torch.Tensor, doc_end: torch.Tensor, n_rows: int):
    """A stale doc_infos range (another store's, or past the token matrix) would make the LDS-DMA stream read out of
    bounds silently.  A device reduction + one blocking D2H read: callers that validated their ranges when they built
    them (token_store.TokenStore does, on the host copy of doc_infos) pass check_ranges=False and keep the launch
    stream free of synchronisation."""
    if torch.cuda.is_current_stream_capturing():
        raise NativeError("maxsim_ragged: range validation needs a D2H read; validate outside graph capture and pass "
                          "check_ranges=False")
    lo, hi = int(doc_begin.min()), int(torch.maximum(doc_begin, doc_end).max())
    bad = int((doc_begin > doc_end).sum())
    if lo < 0 or hi > n_rows or bad:
        raise NativeError(f"maxsim_ragged: document ranges [{lo}, {hi}) leave the {n_rows}-row token matrix"
                          + (f" ({bad} ranges have begin > end)" if bad else ""))


def maxsim_ragged(q: torch.Tensor, tokens: torch.Tensor, doc_begin: torch.Tensor, doc_end: torch.Tensor,
                  q_mask: Optional[torch.Tensor] = None, pairs_per_query: int = 1,
                  check_ranges: bool = True, sim_round: bool = False, sum_round: bool = False) -> torch.Tensor:
    """Unpadded MaxSim over a resident token store (the ColBERT retrieval aggregate,
    matchmaker/dense_retrieval.py:398-412 + colbert.py:100-112, in ONE launch).

    q [n_queries, Q, E]; tokens [T, E] (the store: token_reps_N.npy rows); document p of the batch is
    tokens[doc_begin[p]:doc_end[p]] (doc_infos ranges); pair p scores against query
    p // pairs_per_query.  Returns float32 [n_pairs]."""
    dev = _dev_check(q, tokens, doc_begin, doc_end, q_mask)
    q = _emb(q, "q")
    if tokens.dim() != 2 or tokens.dtype not in _DT:
        raise NativeError(f"tokens: expected [T, E] float tensor, got {tuple(tokens.shape)} {tokens.dtype}")
    if q.dtype != tokens.dtype:
        raise NativeError(f"q/tokens dtype mismatch: {q.dtype} vs {tokens.dtype} (convert the query to the store's dtype)")
    tokens = tokens if tokens.is_contiguous() else tokens.contiguous()
    nq, Q, E = q.shape
    if tokens.shape[1] != E:
        raise NativeError(f"embedding dims differ: {E} vs {tokens.shape[1]}")
    B = doc_begin.numel()
    if doc_end.numel() != B:
        raise NativeError("doc_begin / doc_end must have one entry per pair")
    if pairs_per_query < 1 or nq != (B + pairs_per_query - 1) // pairs_per_query:
        raise NativeError(f"q has {nq} rows but {B} pairs / {pairs_per_query} per query")
    doc_begin = doc_begin.to(torch.int64).contiguous()
    doc_end = doc_end.to(torch.int64).contiguous()
    qm, qp, qk = _mask(q_mask, nq, Q, "q_mask")
    L = _lib.lib()
    out = torch.empty(B, dtype=torch.float32, device=dev)
    if B == 0:
        return out
    if check_ranges:
        _check_ranges(doc_begin, doc_end, tokens.shape[0])
    with torch.cuda.device(dev):
        wsb = L.mm_maxsim_ragged_workspace_bytes(B, pairs_per_query, Q, qk)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
        rc = L.mm_maxsim_ragged_fwd(q.data_ptr(), tokens.data_ptr(), doc_begin.data_ptr(), doc_end.data_ptr(), qp, qk,
                                    out.data_ptr(), B, pairs_per_query, Q, E, _DT[q.dtype], _flags(sim_round, sum_round),
                                    ws.data_ptr() if ws is not None else None, wsb, _stream(dev))
    _lib.check(rc, "mm_maxsim_ragged_fwd")
    return out


def maxsim_bwd(q: torch.Tensor, d: torch.Tensor, q_mask: Optional[torch.Tensor], d_mask: Optional[torch.Tensor],
               grad_out: torch.Tensor, grad_dtype: Optional[torch.dtype] = None):
    """Backward of the paired MaxSim (pair-per-row layout).  Returns (grad_q [B,Q,E], grad_d [B,D,E]) as float32, or —
    grad_dtype = q.dtype — in the token vectors' own 16-bit type (what autograd hands back to an fp16 / bf16 encoder:
    one launch, summed in fp32, rounded once); see mm_maxsim_bwd in include/mm_native.h."""
    dev = _dev_check(q, d, q_mask, d_mask, grad_out)
    q, d = _emb(q, "q"), _emb(d, "d")
    if q.dtype != d.dtype:
        raise NativeError(f"q/d dtype mismatch: {q.dtype} vs {d.dtype}")
    B, Q, E = q.shape
    B2, D, E2 = d.shape
    if B != B2 or E != E2:
        raise NativeError(f"maxsim_bwd needs the pair-per-row layout: q {tuple(q.shape)} vs d {tuple(d.shape)}")
    go = grad_out.detach().reshape(-1).to(torch.float32).contiguous()
    if go.numel() != B:
        raise NativeError(f"grad_out has {go.numel()} elements for {B} pairs")
    qm, qp, qk = _mask(q_mask, B, Q, "q_mask")
    dm, dp, dk = _mask(d_mask, B, D, "d_mask")
    L = _lib.lib()
    E0 = E
    q, d, E = _pad_rows(q, d, 4 if q.dtype == torch.float32 else 8)
    gdt = torch.float32 if grad_dtype is None else grad_dtype
    if gdt not in (torch.float32, q.dtype):
        raise NativeError(f"maxsim_bwd: gradients are float32 or {q.dtype}, not {gdt}")
    gq = torch.empty((B, Q, E), dtype=gdt, device=dev)
    gd = torch.empty((B, D, E), dtype=gdt, device=dev)
    if B:
        with _on(dev):
            wsb = _ws_bytes(L.mm_maxsim_bwd_workspace_bytes, B, Q, D, qk, dk)
            st = _stream(dev)
            ws = _workspace(dev, wsb, st)
            rc = L.mm_maxsim_bwd(q.data_ptr(), d.data_ptr(), qp, qk, dp, dk, go.data_ptr(), gq.data_ptr(),
                                 gd.data_ptr(), _DT[gdt], B, Q, D, E, _DT[q.dtype], ws.data_ptr() if ws is not None else None,
                                 wsb, st)
        _lib.check(rc, "mm_maxsim_bwd")
    if E != E0:
        gq, gd = gq[..., :E0].contiguous(), gd[..., :E0].contiguous()
    return gq, gd


def maxsim_inbatch(q: torch.Tensor, q_mask: Optional[torch.Tensor], d: torch.Tensor,
                   d_mask: Optional[torch.Tensor], bug_compatible: bool = False, sim_round: bool = False,
                   sum_round: bool = False) -> torch.Tensor:
    """All-pairs MaxSim [Bq, Bd] (matchmaker/models/colbert.py:154-162).  bug_compatible=True masks
    score[i, j] with document i's mask as the reference does (and needs Bq == Bd).  sim_round / sum_round as in
    maxsim() (the dynamic teacher calls this on fp16 vectors outside autocast: both, dynamic_teacher.py:245-246)."""
    dev = _dev_check(q, d, q_mask, d_mask)
    q, d = _emb(q, "q"), _emb(d, "d")
    if q.dtype != d.dtype:
        raise NativeError(f"q/d dtype mismatch: {q.dtype} vs {d.dtype}")
    Bq, Q, E = q.shape
    Bd, D, E2 = d.shape
    if E != E2:
        raise NativeError(f"embedding dims differ: {E} vs {E2}")
    qm, qp, qk = _mask(q_mask, Bq, Q, "q_mask")
    dm, dp, dk = _mask(d_mask, Bd, D, "d_mask")
    L = _lib.lib()
    out = torch.empty((Bq, Bd), dtype=torch.float32, device=dev)
    if Bq == 0 or Bd == 0:
        return out
    q, d, E = _pad_rows(q, d, 4 if q.dtype == torch.float32 else 8)
    with torch.cuda.device(dev):
        wsb = L.mm_maxsim_inbatch_workspace_bytes(Bq, Bd, Q, D, qk, dk)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
        rc = L.mm_maxsim_inbatch_fwd(q.data_ptr(), d.data_ptr(), qp, qk, dp, dk, out.data_ptr(), Bq, Bd, Q, D, E,
                                     _DT[q.dtype], 1 if bug_compatible else 0, _flags(sim_round, sum_round),
                                     ws.data_ptr() if ws is not None else None, wsb, _stream(dev))
    _lib.check(rc, "mm_maxsim_inbatch_fwd")
    return out


def kernel_pool(q: torch.Tensor, d: torch.Tensor, q_mask: Optional[torch.Tensor], d_mask: Optional[torch.Tensor],
                mu: torch.Tensor, sigma: torch.Tensor, alpha: torch.Tensor, w: torch.Tensor,
                pairs_per_query: int = 1, return_per_kernel: bool = False,
                d_gate: Optional[torch.Tensor] = None, clamp_min: float = 1e-10,
                pair_query: Optional[torch.Tensor] = None, return_pooled: bool = False):
    """TK kernel pooling (matchmaker/models/published/ecai20_tk.py:105-124).

    q [n_queries, Q, E], d [n_pairs, D, E] float32 contextualised embeddings; mu/sigma/alpha/w [K].
    d_gate [n_pairs, D] >= 0 (optional): TK-Sparse's stop-word vector (cikm20_tk_sparse.py:133-135);
    clamp_min: floor inside the log (1e-4: IDCM sampler, sigir21_idcm.py:185);
    pair_query [n_pairs] int (optional): row of q each pair scores against (ragged groups; replaces
    pairs_per_query; equal neighbours reuse the query tile).
    Returns float32 [n_pairs] (and per_kernel [n_pairs, K] when asked; and, with return_pooled, the pooled kernel sums
    [n_pairs, Q, K] of :120 as the LAST element — what kernel_pool_bwd(pooled=...) takes to skip its pooling pre-pass; rows of
    padded query tokens are unspecified)."""
    dev = _dev_check(q, d, q_mask, d_mask, mu, sigma, alpha, w, d_gate, pair_query)
    q, d = _emb(q, "q"), _emb(d, "d")
    if q.dtype != torch.float32 or d.dtype != torch.float32:
        raise NativeError("kernel_pool: float32 embeddings only (the reference cosine rejects bf16, "
                          "and tk.yaml sets use_fp16: False)")
    nq, Q, E = q.shape
    B, D, E2 = d.shape
    if E != E2:
        raise NativeError(f"embedding dims differ: {E} vs {E2}")
    if pair_query is not None:
        pq = pair_query.reshape(-1).to(torch.int32).contiguous()
        if pq.numel() != B:
            raise NativeError(f"pair_query has {pq.numel()} entries for {B} pairs")
        if B and not torch.cuda.is_current_stream_capturing():
            lo, hi = int(pq.min()), int(pq.max())
            if lo < 0 or hi >= nq:
                raise NativeError(f"pair_query values [{lo}, {hi}] outside the {nq} query rows")
        pairs_per_query = 1
    elif pairs_per_query < 1 or nq != (B + pairs_per_query - 1) // pairs_per_query:
        raise NativeError(f"q has {nq} rows but {B} pairs / {pairs_per_query} per query")
    else:
        pq = None
    K = mu.numel()
    mu, sigma, alpha, w = _vec(mu), _vec(sigma), _vec(alpha), _vec(w)
    if not (sigma.numel() == alpha.numel() == w.numel() == K):
        raise NativeError("kernel_pool: mu/sigma/alpha/w must all have K elements")
    qm, qp, qk = _mask(q_mask, nq, Q, "q_mask")
    dm, dp, dk = _mask(d_mask, B, D, "d_mask")
    gate = _gate(d_gate, B, D)
    L = _lib.lib()
    out = torch.empty(B, dtype=torch.float32, device=dev)
    pk = torch.empty((B, K), dtype=torch.float32, device=dev) if return_per_kernel else None
    pooled = torch.empty((B, Q, K), dtype=torch.float32, device=dev) if return_pooled else None
    if B:
        q, d, E = _pad_rows(q, d, 4)
        with _on(dev):
            wsb = _ws_bytes(L.mm_kernel_pool_workspace_bytes, max(B, nq), 1 if pq is not None else pairs_per_query, Q, D, qk, dk)
            st = _stream(dev)
            ws = _workspace(dev, wsb, st)
            rc = L.mm_kernel_pool_ex_fwd2(q.data_ptr(), d.data_ptr(), qp, qk, dp, dk,
                                          gate.data_ptr() if gate is not None else None,
                                          pq.data_ptr() if pq is not None else None, nq, mu.data_ptr(),
                                          sigma.data_ptr(), alpha.data_ptr(), w.data_ptr(), float(clamp_min),
                                          out.data_ptr(), pk.data_ptr() if pk is not None else None,
                                          pooled.data_ptr() if pooled is not None else None, B,
                                          pairs_per_query, Q, D, E, K, _lib.MM_F32,
                                          ws.data_ptr() if ws is not None else None, wsb, st)
        _lib.check(rc, "mm_kernel_pool_ex_fwd2")
    res = (out, pk) if return_per_kernel else (out,)
    if return_pooled:
        res = res + (pooled,)
    return res if len(res) > 1 else out


def kernel_pool_multi(q_list, d_list, q_mask: Optional[torch.Tensor], d_mask: Optional[torch.Tensor], mu: torch.Tensor,
                      sigma: torch.Tensor, alpha: torch.Tensor, w: torch.Tensor, clamp_min: float = 1e-10) -> torch.Tensor:
    """Sum over all (i, t) of kernel_pool(q_list[i], d_list[t], bin weights w[i * len(d_list) + t]) in ONE launch
    (+ a deterministic sum): Conv-KNRM's n_grams^2 match matrices and its dense layer (conv_knrm.py:130-137).
    q_list[i] [B, Q, E], d_list[t] [B, D, E] float32 (pair-per-row), w [len(q_list) * len(d_list), K].  Returns [B]."""
    import ctypes
    q_list = [_emb(t, "q") for t in q_list]
    d_list = [_emb(t, "d") for t in d_list]
    dev = _dev_check(*q_list, *d_list, q_mask, d_mask, mu, sigma, alpha, w)
    B, Q, E = q_list[0].shape
    D = d_list[0].shape[1]
    for t in q_list:
        if tuple(t.shape) != (B, Q, E) or t.dtype != torch.float32:
            raise NativeError(f"kernel_pool_multi: query tensors must all be float32 [{B},{Q},{E}]")
    for t in d_list:
        if tuple(t.shape) != (B, D, E) or t.dtype != torch.float32:
            raise NativeError(f"kernel_pool_multi: document tensors must all be float32 [{B},{D},{E}]")
    nq_, nd_ = len(q_list), len(d_list)
    K = mu.numel()
    f = lambda t: t.detach().reshape(-1).to(torch.float32).contiguous()
    mu, sigma, alpha, w = f(mu), f(sigma), f(alpha), f(w)
    if w.numel() != nq_ * nd_ * K or sigma.numel() != K or alpha.numel() != K:
        raise NativeError("kernel_pool_multi: w must hold K weights per (query tensor, document tensor) combination")
    if E % 4:
        pad = (0, 4 - E % 4)
        q_list = [torch.nn.functional.pad(t, pad) for t in q_list]
        d_list = [torch.nn.functional.pad(t, pad) for t in d_list]
        E = q_list[0].shape[-1]
    qm, qp, qk = _mask(q_mask, B, Q, "q_mask")
    dm, dp, dk = _mask(d_mask, B, D, "d_mask")
    L = _lib.lib()
    out = torch.empty(B, dtype=torch.float32, device=dev)
    if B == 0:
        return out
    qa = (ctypes.c_void_p * nq_)(*[t.data_ptr() for t in q_list])
    da = (ctypes.c_void_p * nd_)(*[t.data_ptr() for t in d_list])
    with torch.cuda.device(dev):
        wsb = L.mm_kernel_pool_multi_workspace_bytes(B, 1, nq_, nd_, Q, D, qk, dk)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        rc = L.mm_kernel_pool_multi_fwd(ctypes.cast(qa, ctypes.c_void_p), nq_, ctypes.cast(da, ctypes.c_void_p), nd_, qp, qk, dp, dk,
                                        mu.data_ptr(), sigma.data_ptr(), alpha.data_ptr(), w.data_ptr(), float(clamp_min),
                                        out.data_ptr(), B, 1, Q, D, E, K, _lib.MM_F32, ws.data_ptr(), wsb, _stream(dev))
    _lib.check(rc, "mm_kernel_pool_multi_fwd")
    return out


def _gate(d_gate, B, D):
    if d_gate is None:
        return None
    g = d_gate.detach().reshape(B, -1).to(torch.float32).contiguous()
    if g.shape[1] != D:
        raise NativeError(f"d_gate has shape {tuple(d_gate.shape)} for {B} documents of {D} tokens")
    return g


def kernel_pool_bwd(q: torch.Tensor, d: torch.Tensor, q_mask: Optional[torch.Tensor], d_mask: Optional[torch.Tensor],
                    mu: torch.Tensor, sigma: torch.Tensor, alpha: torch.Tensor, w: torch.Tensor, grad_out: torch.Tensor,
                    d_gate: Optional[torch.Tensor] = None, clamp_min: float = 1e-10, pooled: Optional[torch.Tensor] = None):
    """Backward of kernel_pool in the pair-per-row layout (mm_kernel_pool_ex_bwd2).  Returns float32
    (grad_q [B,Q,E], grad_d [B,D,E], grad_alpha [K], grad_w [K]) and, with d_gate, grad_gate [B,D] as a
    fifth element.  pooled [B,Q,K]: the forward's pooled kernel sums (kernel_pool(..., return_pooled=True) on the same inputs);
    without them the backward pools them itself first (the document crosses HBM twice)."""
    dev = _dev_check(q, d, q_mask, d_mask, mu, sigma, alpha, w, grad_out, d_gate, pooled)
    q, d = _emb(q, "q"), _emb(d, "d")
    if q.dtype != torch.float32 or d.dtype != torch.float32:
        raise NativeError("kernel_pool_bwd: float32 embeddings only")
    B, Q, E = q.shape
    B2, D, E2 = d.shape
    if B != B2 or E != E2:
        raise NativeError(f"kernel_pool_bwd needs the pair-per-row layout: q {tuple(q.shape)} vs d {tuple(d.shape)}")
    K = mu.numel()
    f = lambda t: t.detach().reshape(-1).to(torch.float32).contiguous()
    mu, sigma, alpha, w = f(mu), f(sigma), f(alpha), f(w)
    go = f(grad_out)
    if go.numel() != B:
        raise NativeError(f"grad_out has {go.numel()} elements for {B} pairs")
    qm, qp, qk = _mask(q_mask, B, Q, "q_mask")
    dm, dp, dk = _mask(d_mask, B, D, "d_mask")
    L = _lib.lib()
    E0 = E
    q, d, E = _pad_rows(q, d, 4)
    gq = torch.empty((B, Q, E), dtype=torch.float32, device=dev)
    gd = torch.empty((B, D, E), dtype=torch.float32, device=dev)
    gaw = torch.zeros((2, B, K), dtype=torch.float32, device=dev)    # per-pair rows of grad_alpha, grad_w: one memset, one sum
    ga, gw = gaw[0], gaw[1]
    gate = _gate(d_gate, B, D)
    gg = torch.zeros((B, D), dtype=torch.float32, device=dev) if gate is not None else None
    if pooled is not None:
        if pooled.dtype != torch.float32 or tuple(pooled.shape) != (B, Q, K):
            raise NativeError(f"pooled has shape {tuple(pooled.shape)} / {pooled.dtype}, expected float32 {(B, Q, K)}")
        pooled = pooled.detach().contiguous()
    if B:
        with torch.cuda.device(dev):
            wsb = L.mm_kernel_pool_bwd_workspace_bytes2(B, Q, D, E, qk, dk)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
            rc = L.mm_kernel_pool_ex_bwd2(q.data_ptr(), d.data_ptr(), qp, qk, dp, dk,
                                          gate.data_ptr() if gate is not None else None, mu.data_ptr(),
                                          sigma.data_ptr(), alpha.data_ptr(), w.data_ptr(), float(clamp_min),
                                          pooled.data_ptr() if pooled is not None else None,
                                          go.data_ptr(), gq.data_ptr(), gd.data_ptr(),
                                          gg.data_ptr() if gg is not None else None, ga.data_ptr(), gw.data_ptr(),
                                          B, Q, D, E, K, ws.data_ptr() if ws is not None else None, wsb, _stream(dev))
        _lib.check(rc, "mm_kernel_pool_ex_bwd2")
    if E != E0:
        gq, gd = gq[..., :E0].contiguous(), gd[..., :E0].contiguous()
    gaw = gaw.sum(1)
    if gate is not None:
        return gq, gd, gaw[0], gaw[1], gg
    return gq, gd, gaw[0], gaw[1]


def tkl_score(q_ctx: torch.Tensor, chunks: torch.Tensor, chunk_mask: torch.Tensor, chunk_slot: torch.Tensor,
              q_mask: torch.Tensor, params: torch.Tensor, B: int, C: int, K: int, saturation: str = "embedding",
              return_windows: bool = False, check_order: bool = True, return_peaks: bool = False):
    """TKL windowed kernel pooling + region top-k (sigir20_tkl.py:180-286).  See mm_native.h.

    return_peaks: also return the region search's three arg-max window indices per document, int64 [B, 3] in round order =
    the reference's `top_non_overlapping_idx` (:266-271).  Returns score | (score, win) | (score, win, peaks).

    chunk_slot must be strictly ASCENDING (what boolean-mask packing / torch.nonzero produce, sigir20_tkl.py:159-162): the
    kernels rely on a document's chunks being adjacent and on its last kept chunk coming last.  check_order=True verifies
    that on the device without a host synchronisation (torch._assert_async: a violation raises at the next
    synchronisation point instead of silently zeroing live windows); callers that built chunk_slot with
    tkl.chunk_documents() — the drop-in does — pass False and skip the three small launches."""
    dev = _dev_check(q_ctx, chunks, chunk_mask, chunk_slot, q_mask, params)
    q_ctx, chunks = _emb(q_ctx, "q_ctx"), _emb(chunks, "chunks")
    if q_ctx.dtype != torch.float32 or chunks.dtype != torch.float32:
        raise NativeError("tkl_score: float32 only (tkl.yaml use_fp16: False)")
    Bq, Q, E = q_ctx.shape
    P = chunks.shape[0]
    if Bq != B or chunks.shape[1] != 50 or chunks.shape[2] != E:
        raise NativeError(f"tkl_score: bad shapes q_ctx {tuple(q_ctx.shape)} chunks {tuple(chunks.shape)}")
    sat = {"embedding": _lib.TKL_SAT_EMBEDDING, "log": _lib.TKL_SAT_LOG}.get(saturation)
    if sat is None:
        raise NativeError(f"tkl_score: saturation {saturation!r} is dead code in the reference "
                          "(reads the undefined `query_idfs`, sigir20_tkl.py:214,236)")
    chunk_mask = chunk_mask.to(torch.float32).contiguous()
    chunk_slot = chunk_slot.to(torch.int32).contiguous()
    if check_order and P > 1 and not torch.cuda.is_current_stream_capturing():
        torch._assert_async((chunk_slot[1:] > chunk_slot[:-1]).all(),
                            "tkl_score: chunk_slot must be strictly ascending (include/mm_native.h, mm_tkl_fwd)")
    q_mask = q_mask.to(torch.float32).contiguous()
    params = params.to(torch.float32).contiguous()
    W = (max(C * 40, 30) - 30) // 2 + 1
    L = _lib.lib()
    out = torch.empty(B, dtype=torch.float32, device=dev)
    win = torch.empty((B, W), dtype=torch.float32, device=dev)
    peaks = torch.empty((B, 3), dtype=torch.int32, device=dev) if return_peaks else None
    if B:
        with torch.cuda.device(dev):
            wsb = L.mm_tkl_workspace_bytes(B, P, C, Q, K)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
            rc = L.mm_tkl_fwd_peaks(q_ctx.data_ptr(), chunks.data_ptr(), chunk_mask.data_ptr(), chunk_slot.data_ptr(),
                                    q_mask.data_ptr(), params.data_ptr(), win.data_ptr(), out.data_ptr(),
                                    peaks.data_ptr() if peaks is not None else None, B, P, C, Q, E,
                                    K, sat, ws.data_ptr() if ws is not None else None, wsb, _stream(dev))
        _lib.check(rc, "mm_tkl_fwd_peaks")
    if return_peaks:
        return out, win, peaks.long()
    return (out, win) if return_windows else out


def tkl_bwd(q_ctx: torch.Tensor, chunks: torch.Tensor, chunk_mask: torch.Tensor, chunk_slot: torch.Tensor,
            q_mask: torch.Tensor, params: torch.Tensor, win: torch.Tensor, grad_out: torch.Tensor, B: int, C: int, K: int,
            saturation: str = "embedding"):
    """Backward of tkl_score (mm_tkl_bwd): returns float32 (grad_q_ctx [B,Q,E], grad_chunks [P,50,E],
    grad_params [MM_TKL_NPARAMS] summed over the documents, in the layout of `params`)."""
    dev = _dev_check(q_ctx, chunks, chunk_mask, chunk_slot, q_mask, params, win, grad_out)
    q_ctx, chunks = _emb(q_ctx.detach(), "q_ctx"), _emb(chunks.detach(), "chunks")
    if q_ctx.dtype != torch.float32 or chunks.dtype != torch.float32:
        raise NativeError("tkl_bwd: float32 only")
    Bq, Q, E = q_ctx.shape
    P = chunks.shape[0]
    sat = {"embedding": _lib.TKL_SAT_EMBEDDING, "log": _lib.TKL_SAT_LOG}.get(saturation)
    if sat is None or Bq != B or (P and (chunks.shape[1] != 50 or chunks.shape[2] != E)):
        raise NativeError(f"tkl_bwd: bad arguments (saturation {saturation!r}, q_ctx {tuple(q_ctx.shape)}, chunks {tuple(chunks.shape)})")
    chunk_mask = chunk_mask.to(torch.float32).contiguous()
    chunk_slot = chunk_slot.to(torch.int32).contiguous()
    q_mask = q_mask.to(torch.float32).contiguous()
    params = params.detach().to(torch.float32).contiguous()
    win = win.detach().to(torch.float32).contiguous()
    go = grad_out.detach().reshape(-1).to(torch.float32).contiguous()
    NP = params.numel()
    gq = torch.empty((B, Q, E), dtype=torch.float32, device=dev)
    gc = torch.empty((P, 50, E), dtype=torch.float32, device=dev)
    gp = torch.empty((B, NP), dtype=torch.float32, device=dev)
    if B == 0:
        return gq, gc.zero_(), torch.zeros(NP, dtype=torch.float32, device=dev)
    L = _lib.lib()
    with torch.cuda.device(dev):
        wsb = L.mm_tkl_bwd_workspace_bytes2(B, C, Q, E)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        rc = L.mm_tkl_bwd(q_ctx.data_ptr(), chunks.data_ptr() if P else None, chunk_mask.data_ptr() if P else None,
                          chunk_slot.data_ptr() if P else None, q_mask.data_ptr(), params.data_ptr(), win.data_ptr(), go.data_ptr(),
                          gq.data_ptr(), gc.data_ptr() if P else None, gp.data_ptr(), B, P, C, Q, E, K, sat, ws.data_ptr(), wsb,
                          _stream(dev))
    _lib.check(rc, "mm_tkl_bwd")
    return gq, gc, gp.sum(0)


def dot_topk(queries: torch.Tensor, corpus: torch.Tensor, k: int, max_rounds: int = 6):
    """Exact brute-force inner-product top-k over one shard (faiss IndexFlatIP.search semantics;
    matchmaker/retrieval/faiss_indices.py:22-36, :49-74; score = bert_dot.py:62).

    queries [nq, E], corpus [N, E] float16 / bfloat16 on the same device, E in {128,...,768} (pad
    otherwise).  Returns (scores [nq, k] float32 descending, idx [nq, k] int64 rows of `corpus`,
    -1 / -inf padded when N < k).  The native call thresholds every query from a sample of the
    shard; queries whose threshold let too few / too many candidates through (status != 0, rare)
    are re-run with a moved threshold until every row is exact."""
    dev = _dev_check(queries, corpus)
    if queries.dim() != 2 or corpus.dim() != 2 or queries.shape[1] != corpus.shape[1]:
        raise NativeError(f"dot_topk: expected [nq, E] and [N, E], got {tuple(queries.shape)} {tuple(corpus.shape)}")
    if queries.dtype != corpus.dtype or corpus.dtype not in (torch.float16, torch.bfloat16):
        raise NativeError(f"dot_topk: float16 / bfloat16 vectors of one dtype needed, got {queries.dtype} / {corpus.dtype}")
    queries = queries if queries.is_contiguous() else queries.contiguous()
    corpus = corpus if corpus.is_contiguous() else corpus.contiguous()
    nq, E = queries.shape
    N = corpus.shape[0]
    L = _lib.lib()
    out_s = torch.empty((nq, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=dev)
    if nq == 0:
        return out_s, out_i
    if N == 0:
        return out_s.fill_(float("-inf")), out_i.fill_(-1)

    def run(q, scale, s=None, i=None):
        n = q.shape[0]
        if s is None:
            s = torch.empty((n, k), dtype=torch.float32, device=dev)
            i = torch.empty((n, k), dtype=torch.int64, device=dev)
        st = torch.empty(n, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            wsb = L.mm_dot_topk_workspace_bytes(N, n, k)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            rc = L.mm_dot_topk_fwd(q.data_ptr(), corpus.data_ptr(), N, n, E, _DT[q.dtype], k, scale, s.data_ptr(),
                                   i.data_ptr(), st.data_ptr(), ws.data_ptr(), wsb, _stream(dev))
        _lib.check(rc, "mm_dot_topk_fwd")
        return s, i, st

    _, _, st = run(queries, 1.0, out_s, out_i)        # straight into the caller's tensors (no 84 MB staging copy)
    if int(st.max()) == 0:                            # one 4-byte D2H: the exactness check of the status vector
        return out_s, out_i
    bad = torch.nonzero(st != 0).flatten().tolist()
    # rare path: bisect the threshold scale per query (status 1 = too few survivors -> larger m,
    # status 2 = candidate list overflowed -> smaller m)
    codes = st[bad].tolist()
    lo = {q: (1.0 if c == 1 else None) for q, c in zip(bad, codes)}   # largest scale known to underflow
    hi = {q: (1.0 if c == 2 else None) for q, c in zip(bad, codes)}   # smallest scale known to overflow
    for _ in range(max_rounds):
        groups = {}
        for q in bad:
            if lo[q] is not None and hi[q] is not None:
                sc = (lo[q] * hi[q]) ** 0.5
            else:
                sc = lo[q] * 4.0 if lo[q] is not None else hi[q] * 0.25
            groups.setdefault(round(sc, 6), []).append(q)
        still = []
        for sc, qs in groups.items():
            sel = torch.tensor(qs, dtype=torch.int64, device=dev)
            s2, i2, st2 = run(queries[sel].contiguous(), float(sc))
            st2 = st2.tolist()
            for n, q in enumerate(qs):
                if st2[n] == 0:
                    out_s[q] = s2[n]
                    out_i[q] = i2[n]
                else:
                    if st2[n] == 1:
                        lo[q] = sc
                    else:
                        hi[q] = sc
                    still.append(q)
        bad = still
        if not bad:
            return out_s, out_i
    raise NativeError(f"dot_topk: {len(bad)} queries without an exact top-{k} after {max_rounds} threshold re-runs "
                      "(more than 4k documents tie at the k-th score?)")


def topk_merge(scores: torch.Tensor, ids: torch.Tensor, k: int):
    """Rows of (score, id) candidates [nq, n_in] -> the k best per row (score descending, input order on
    ties); ids < 0 are padding.  The sharded index's final merge (mm_topk_merge)."""
    dev = _dev_check(scores, ids)
    scores = scores.to(torch.float32).contiguous()
    ids = ids.to(torch.int64).contiguous()
    nq, n_in = scores.shape
    out_s = torch.empty((nq, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=dev)
    if nq:
        with torch.cuda.device(dev):
            rc = _lib.lib().mm_topk_merge(scores.data_ptr(), ids.data_ptr(), nq, n_in, k, out_s.data_ptr(),
                                          out_i.data_ptr(), _stream(dev))
        _lib.check(rc, "mm_topk_merge")
    return out_s, out_i
