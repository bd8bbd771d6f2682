"""HBM-resident mirror of matchmaker's ColBERT retrieval token store + one-launch aggregation.

The reference writes every document's non-zero token vectors back to back into raw memmaps
`token_reps_<n>.npy` of `token_block_size` rows (matchmaker/dense_retrieval.py:205-206, 246-263),
remembers `doc_infos[seq_id] = (file_no, start, end)` (:265) and `storage_filled_to_index` (:249,
:276), saves them to `doc_infos.npz` (:278-279) and reloads them with np.memmap (:292-303).  Its
ColBERT "aggregate" search step then scores ONE candidate per Python iteration:
`storage[file][start:end] -> torch -> forward_aggregation` (:398-412, colbert.py:100-112).

Here the filled parts of all files live in one device tensor [T, E] (8.8 M MSMARCO passages x ~70
tokens x 128 dims x 2 B = 158 GB fits one MI355X's 288 GB) and a whole batch of
(query, candidate list) pairs is scored by ONE mm_maxsim_ragged_fwd launch that reads the candidate
rows in place (no gather, no padding).
"""
import glob
import os
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np
import torch

from . import ops


class TokenStore:
    def __init__(self, tokens: torch.Tensor, seq_ids: Sequence, begin: np.ndarray, end: np.ndarray):
        self._tokens = tokens                         # [T, E] on the scoring device (read-only: the ranges below were validated against it)
        self._tokens_lowp = None                      # fp16 image of an fp32 store, built on the first use_fp16 aggregate()
        self.seq_ids = list(seq_ids)
        self._index = {s: i for i, s in enumerate(self.seq_ids)}
        self._begin = np.asarray(begin, dtype=np.int64)   # global row ranges per document
        self._end = np.asarray(end, dtype=np.int64)
        # validated once, on the host copy of doc_infos: the scoring calls then skip the per-call device check
        if self._begin.size and (self._begin.min() < 0 or self._end.max() > tokens.shape[0] or (self._begin > self._end).any()):
            raise ops.NativeError(f"TokenStore: document ranges leave the {tokens.shape[0]}-row token matrix "
                                  "(doc_infos of another store?)")

    @property
    def tokens(self) -> torch.Tensor:
        """The resident token matrix.  Read-only: aggregate() skips the per-call range check because the document ranges
        were validated against THIS matrix in the constructor — build a new TokenStore for another matrix."""
        return self._tokens

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_reference_parts(cls, storage: List[np.ndarray], doc_infos: Dict, seq_ids: Sequence, device):
        """storage[n] = filled part of token_reps_<n>.npy; doc_infos / seq_ids as saved by the
        reference (dense_retrieval.py:265-266, 278-279)."""
        base = np.concatenate([[0], np.cumsum([s.shape[0] for s in storage])]).astype(np.int64)
        begin = np.empty(len(seq_ids), dtype=np.int64)
        end = np.empty(len(seq_ids), dtype=np.int64)
        for i, sid in enumerate(seq_ids):
            f, a, b = doc_infos[sid]
            begin[i], end[i] = base[f] + a, base[f] + b
        E = storage[0].shape[1]
        T = int(base[-1])
        tokens = torch.empty((T, E), dtype=torch.from_numpy(np.empty(0, dtype=storage[0].dtype)).dtype, device=device)
        for n, part in enumerate(storage):            # file by file: no second host copy of the store
            tokens[int(base[n]): int(base[n + 1])] = torch.from_numpy(np.array(part)).to(device)
        return cls(tokens, seq_ids, begin, end)

    @classmethod
    def load(cls, folder: str, token_dim: int, token_dtype: str, token_block_size: int, device):
        """Reads a folder written by `dense_retrieval.py encode` (same calls as :292-303)."""
        dfs = np.load(os.path.join(folder, "doc_infos.npz"), allow_pickle=True)
        doc_infos = dfs.get("doc_infos")[()]
        seq_ids = dfs.get("seq_ids")[()]
        filled = dfs.get("storage_filled_to_index")[()]
        storage = []
        for f in range(len(glob.glob(os.path.join(folder, "token_reps_*")))):
            mm = np.memmap(os.path.join(folder, "token_reps_" + str(f) + ".npy"), dtype=np.dtype(token_dtype),
                           mode="r", shape=(token_block_size, token_dim))
            storage.append(mm[: int(filled[f])])
        return cls.from_reference_parts(storage, doc_infos, list(seq_ids), device)

    # ------------------------------------------------------------------ lookup + scoring
    def ranges(self, seq_ids: Iterable) -> Tuple[torch.Tensor, torch.Tensor]:
        idx = np.fromiter((self._index[s] for s in seq_ids), dtype=np.int64)
        dev = self.tokens.device
        return torch.from_numpy(self._begin[idx]).to(dev), torch.from_numpy(self._end[idx]).to(dev)

    def aggregate(self, query_vecs: torch.Tensor, candidates: Sequence[Sequence],
                  use_fp16: bool = True) -> List[List[Tuple[object, float]]]:
        """query_vecs [nq, Q, E] (forward_representation output, already multiplied by its mask as in
        `search_type="encode"`); candidates[i] = the seq_ids to re-score for query i (the set the
        reference loops over, :400-402).  use_fp16: the searcher head's autocast switch (indexing_heads.py:49-56,
        `model_config["use_fp16"]` at :407): the stored rows go through `.float()` and autocast's cast back — the same fp16
        values — and `bmm` / `max` return fp16, so every per-token maximum is rounded to fp16 before the fp32 sum;
        False = fp32 similarities of the stored values.  An fp32 store (`token_dtype: float32`) under use_fp16 is scored as
        the reference's autocast scores it: `bmm` casts BOTH operands to fp16 first, so the store's fp16 image (built once, on
        the first such call: +50 % of the store's memory) and the fp16 query are what the kernel reads — MM_SIM_ROUND alone
        would be a no-op on fp32 rows.  Returns, per query, [(seq_id, score)] like `validation_results[query_id]` (:410)."""
        nq = query_vecs.shape[0]
        if len(candidates) != nq:
            raise ValueError(f"{nq} queries but {len(candidates)} candidate lists")
        counts = [len(c) for c in candidates]
        C = max(counts) if counts else 0
        if C == 0:
            return [[] for _ in range(nq)]
        # one launch: pad every list to C pairs with empty ranges (an empty range costs nothing)
        flat, pad = [], []
        for c in candidates:
            flat.extend(c)
            pad.append(C - len(c))
        b, e = self.ranges(flat)
        bb = torch.zeros((nq, C), dtype=torch.int64, device=b.device)
        ee = torch.zeros((nq, C), dtype=torch.int64, device=b.device)
        off = 0
        for i, n in enumerate(counts):
            bb[i, :n], ee[i, :n] = b[off: off + n], e[off: off + n]
            off += n
        tokens = self.tokens
        if use_fp16 and tokens.dtype == torch.float32:
            if self._tokens_lowp is None:
                self._tokens_lowp = tokens.to(torch.float16)
            tokens = self._tokens_lowp
        q = query_vecs.to(tokens.dtype)
        scores = ops.maxsim_ragged(q, tokens, bb.view(-1), ee.view(-1), None, pairs_per_query=C, check_ranges=False,
                                   sim_round=bool(use_fp16)).view(nq, C)
        scores = scores.cpu()
        return [[(candidates[i][j], float(scores[i, j])) for j in range(counts[i])] for i in range(nq)]


def write_reference_store(folder: str, docs: Sequence[np.ndarray], seq_ids: Sequence, token_block_size: int,
                          token_dtype: str = "float16"):
    """Writes `docs` (each [n_tokens, E]) exactly as dense_retrieval.py:205-279 lays them out (raw
    memmap blocks + doc_infos.npz in numpy's npz container, :278 saveCompressed = stored zip of .npy).
    Used by the tests and for building synthetic stores; the reference's own writer is its encode loop."""
    os.makedirs(folder, exist_ok=True)
    E = docs[0].shape[1]
    n, ins = 0, 0
    base = np.memmap(os.path.join(folder, f"token_reps_{n}.npy"), dtype=np.dtype(token_dtype), mode="w+",
                     shape=(token_block_size, E))
    doc_infos, filled = {}, []
    for sid, reps in zip(seq_ids, docs):
        reps = reps[np.abs(reps).sum(-1) > 0, :]                         # :244 zero rows are padding
        k = reps.shape[0]
        if ins + k > token_block_size:                                   # :248-256 start the next file
            filled.append(ins)
            base.flush()
            n, ins = n + 1, 0
            base = np.memmap(os.path.join(folder, f"token_reps_{n}.npy"), dtype=np.dtype(token_dtype), mode="w+",
                             shape=(token_block_size, E))
        base[ins: ins + k] = reps
        doc_infos[sid] = (n, ins, ins + k)
        ins += k
    filled.append(ins)
    base.flush()
    ids = np.empty(len(seq_ids), dtype=object)
    ids[:] = list(seq_ids)
    np.savez(os.path.join(folder, "doc_infos.npz"), doc_infos=np.array(doc_infos, dtype=object), seq_ids=ids,
             storage_filled_to_index=np.array(filled), id_mapping=np.array([], dtype=object))
