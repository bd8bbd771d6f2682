"""Drop-in for matchmaker's brute-force faiss index on MI355X: same surface as
`FaissIdIndexer` (matchmaker/retrieval/faiss_indices.py:49-74; base class :13-36) —
`prepare(data_chunks)`, `index(ids, data_chunks)`, `search(query_vec, top_n) -> (scores, ids)` —
with the collection resident in HBM as float16 (what `co.useFloat16` stores on the reference's GPUs)
and the search done by the native Q x C^T + exact top-k kernels (mm_dot_topk_fwd).

Multi-GPU = the reference's `co.shard = True` (:62-66): every rank holds a contiguous shard of the
vectors; `search` runs the local top-k, all-gathers the [nq, k] (score, id) lists over RCCL and
merges them natively (mm_topk_merge).  Used from dense_retrieval.py:308-328 (construction),
:333-336 (prepare / index) and :391 (search).
"""
from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .sharding import shard_range


def _pad_dim(E: int) -> int:
    for e in (128, 256, 384, 512, 768):
        if E <= e:
            return e
    raise ops.NativeError(f"token_dim {E} > 768 is not supported by the native flat index")


class FlatIPIndexer:
    def __init__(self, config, device=None, group=None, topk_fn=None, merge_fn=None, merge_single_rank: bool = False):
        """topk_fn(queries, vectors, k) / merge_fn(scores, ids, k) default to the native operators
        (ops.dot_topk / ops.topk_merge); the CPU test-suite injects oracle stand-ins to exercise the
        sharding logic under gloo.  merge_single_rank: run the two all-gathers + the merge of the sharded search even
        when the process group has ONE rank (rehearsal of the multi-GPU path on a single-GPU box)."""
        self.merge_single_rank = bool(merge_single_rank)
        self._topk = topk_fn if topk_fn is not None else ops.dot_topk
        self._merge = merge_fn if merge_fn is not None else ops.topk_merge
        self.token_dim = config["token_dim"]
        # base_index.py:14 derives fp16 storage from config["token_dtype"] == "float16"; `faiss_use_fp16` (not a reference
        # key) is kept as an explicit override only
        self.use_fp16 = config.get("faiss_use_fp16", config.get("token_dtype", "float16") == "float16")
        if not self.use_fp16:
            # faiss keeps fp32 vectors AND fp32 queries when useFloat16 is off (faiss_indices.py:58-61); this index
            # stores fp16 vectors and rounds the queries to fp16 as well, so near-tie rankings could differ from
            # an fp32 IndexFlatIP.  Refuse instead of silently changing the arithmetic.
            raise ops.NativeError("FlatIPIndexer stores float16 vectors and rounds queries to float16 (faiss useFloat16 "
                                  "semantics): set token_dtype: float16 (base_index.py:14), or keep faiss for an fp32 index")
        self.dtype = torch.float16
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())  # noqa: E501
        self.group = group
        self.vectors: Optional[torch.Tensor] = None           # [n_local, E_pad]
        self.ids: Optional[torch.Tensor] = None               # [n_local] int64 external ids (IndexIDMap)
        self.E_pad = _pad_dim(self.token_dim)

    def _world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group), dist.get_rank(self.group)
        return 1, 0

    def prepare(self, data_chunks: List[np.ndarray]):          # base_index.py: nothing to train for a flat index
        pass

    def index(self, ids: List[np.ndarray], data_chunks: List[np.ndarray]):
        """faiss_indices.py:22-27: one add of all vectors with their ids.  With several ranks each
        keeps its contiguous shard (every rank is given the same full lists, as the reference's single
        process is)."""
        i = np.concatenate(ids).astype(np.int64)
        n = i.shape[0]
        world, rank = self._world()
        lo, hi = shard_range(n, world, rank)
        vec = torch.zeros((hi - lo, self.E_pad), dtype=self.dtype, device=self.device)
        off = 0
        for c in data_chunks:                                  # chunk by chunk: no second host copy
            a, b = max(lo, off), min(hi, off + c.shape[0])
            if a < b:
                vec[a - lo: b - lo, : self.token_dim] = torch.from_numpy(np.ascontiguousarray(c[a - off: b - off])).to(
                    self.device).to(self.dtype)
            off += c.shape[0]
        self.vectors = vec
        self.ids = torch.from_numpy(i[lo:hi]).to(self.device)

    def index_resident(self, ids: torch.Tensor, vectors: torch.Tensor):
        """This rank's shard handed over as device tensors (vectors [n_local, E_pad] float16, ids [n_local] int64) — for
        collections that are produced on the device (bench.py generates each rank's shard of the 8.8 M synthetic passages in
        place instead of materialising 13.6 GB of host arrays per rank)."""
        if vectors.dtype != self.dtype or vectors.dim() != 2 or vectors.shape[1] != self.E_pad or ids.shape[0] != vectors.shape[0]:
            raise ops.NativeError(f"index_resident: need float16 [n, {self.E_pad}] vectors and [n] ids")
        self.vectors, self.ids = vectors.contiguous(), ids.to(torch.int64).contiguous()

    def search(self, query_vec, top_n: int):
        """faiss_indices.py:29-36: (scores [nq, top_n] float32 descending, ids [nq, top_n] int64)."""
        s, ids = self.search_device(query_vec, top_n)
        return s.cpu().numpy(), ids.cpu().numpy()

    def search_device(self, query_vec, top_n: int):
        """search() without the final copy to the host: device tensors (what a caller that keeps working on the GPU wants,
        and what bench.py times)."""
        q = torch.as_tensor(query_vec)
        if q.dim() == 1:
            q = q[None, :]
        if q.is_cuda and q.dtype == self.dtype and q.shape[1] == self.E_pad and q.is_contiguous():
            qd = q
        else:
            qd = torch.zeros((q.shape[0], self.E_pad), dtype=self.dtype, device=self.device)
            qd[:, : self.token_dim] = q.to(self.device).to(self.dtype)
        s, idx = self._topk(qd, self.vectors, top_n)
        ids = torch.where(idx >= 0, self.ids[idx.clamp(min=0)], idx)
        world, _ = self._world()
        if world > 1 or (self.merge_single_rank and dist.is_available() and dist.is_initialized()):
            nq = s.shape[0]
            gs = torch.empty((world * nq, top_n), dtype=s.dtype, device=s.device)        # rank-major concatenation
            gi = torch.empty((world * nq, top_n), dtype=ids.dtype, device=ids.device)
            dist.all_gather_into_tensor(gs, s.contiguous(), group=self.group)             # RCCL over xGMI
            dist.all_gather_into_tensor(gi, ids.contiguous(), group=self.group)
            s, ids = self._merge(gs.view(world, nq, top_n).permute(1, 0, 2).reshape(nq, -1),
                                 gi.view(world, nq, top_n).permute(1, 0, 2).reshape(nq, -1), top_n)
        return s, ids
