"""Multi-GPU re-ranking: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Queries (each with its whole candidate list) are split contiguously over the ranks, so a query's
tile is read by exactly one GPU and there is no data-path collective; the only exchange is ONE
all-gather of the fp32 scores for the final ranking merge.  This replaces
nn.DataParallel's per-forward scatter / parameter broadcast / gather (matchmaker/train.py:194-202).

Ranking rule = the reference's: per query, stable sort by score, descending
(matchmaker/utils/core_metrics.py:502-511).
"""
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced split: the first n_items % world ranks get one extra item."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_scores(local: torch.Tensor, n_total: int, group=None, force: bool = False) -> torch.Tensor:
    """local [n_local, C] fp32 from every rank (shard_range split of n_total rows) -> [n_total, C] on
    every rank.  One all_gather_into_tensor of equally sized (padded) shards.  force: run the padded collective even
    with ONE rank in the group (rehearsal of the multi-GPU path on a single-GPU box)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force and dist.is_initialized()):
        return local
    C = local.shape[1]
    per = (n_total + world - 1) // world
    buf = torch.zeros((per, C), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    out = torch.empty((world * per, C), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    parts = []
    for r in range(world):
        s, e = shard_range(n_total, world, r)
        parts.append(out[r * per: r * per + (e - s)])
    return torch.cat(parts, dim=0)


def rank_candidates(scores: torch.Tensor, k: Optional[int] = None) -> torch.Tensor:
    """[n_queries, C] -> candidate indices by descending score, ties in arrival order (stable)."""
    order = torch.sort(scores, dim=1, descending=True, stable=True).indices
    return order if k is None else order[:, :k]


def rerank_sharded(q: torch.Tensor, d: torch.Tensor, q_mask, d_mask, cands: int,
                   score_fn: Optional[Callable] = None, group=None, n_total_queries: Optional[int] = None,
                   force_collective: bool = False):
    """Each rank passes ITS shard (q [nq_local,Q,E], d [nq_local*cands,D,E], masks alike); returns
    (scores [n_total, cands], ranking [n_total, cands]) on every rank.

    score_fn(q, d, q_mask, d_mask, pairs_per_query) defaults to the native MaxSim operator."""
    if score_fn is None:
        from . import ops
        score_fn = ops.maxsim
    local = score_fn(q, d, q_mask, d_mask, cands).view(q.shape[0], cands)
    if n_total_queries is None:
        n = torch.tensor([q.shape[0]], dtype=torch.int64, device=local.device)
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(n, group=group)
        n_total_queries = int(n.item())
    scores = all_gather_scores(local, n_total_queries, group, force=force_collective)
    return scores, rank_candidates(scores)
