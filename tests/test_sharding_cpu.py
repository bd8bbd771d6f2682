"""CPU (gloo, world_size 2): the N > 1 path — query sharding, the score all-gather and the ranking
merge — with the oracle standing in for the device scorer (tests may use the oracle; the product
default is the native operator)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from matchmaker_amd import sharding


def test_shard_range_is_a_balanced_partition():
    for n, w in [(6980, 8), (7, 2), (3, 8), (0, 4), (64, 1)]:
        edges = [sharding.shard_range(n, w, r) for r in range(w)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
        sizes = [e - s for s, e in edges]
        assert max(sizes) - min(sizes) <= 1
    assert sharding.shard_range(6980, 8, 0) == (0, 873) and sharding.shard_range(6980, 8, 7) == (6108, 6980)


def test_rank_candidates_is_stable_descending():
    s = torch.tensor([[1.0, 3.0, 3.0, -2.0, 3.0], [0.0, 0.0, 0.0, 0.0, 0.0]])
    assert sharding.rank_candidates(s).tolist() == [[1, 2, 4, 0, 3], [0, 1, 2, 3, 4]]
    assert sharding.rank_candidates(s, 2).tolist() == [[1, 2], [0, 1]]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nq, cands, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import np_oracle as O
    Q, D, E = 8, 20, 16
    g = torch.Generator().manual_seed(99)                      # every rank builds the same full problem
    q = torch.randn(nq, Q, E, generator=g)
    d = torch.randn(nq * cands, D, E, generator=g)
    d_len = torch.randint(0, D + 1, (nq * cands,), generator=g)
    q_len = torch.randint(1, Q + 1, (nq,), generator=g)

    def score_fn(q, d, ql, dl, ppq):
        qi = np.arange(d.shape[0]) // ppq
        qm = (np.arange(Q)[None] < ql.numpy()[:, None])[qi]
        dm = np.arange(D)[None] < dl.numpy()[:, None]
        return torch.from_numpy(O.maxsim_paired(q.numpy()[qi], d.numpy(), qm, dm))

    s, e = sharding.shard_range(nq, world, rank)
    scores, ranking = sharding.rerank_sharded(q[s:e], d[s * cands:e * cands], q_len[s:e], d_len[s * cands:e * cands],
                                              cands, score_fn=score_fn)
    full = score_fn(q, d, q_len, d_len, cands).view(nq, cands)
    assert torch.equal(scores, full), "gathered scores differ from the single-process result"
    assert torch.equal(ranking, sharding.rank_candidates(full))
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), ranking.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nq", [5, 6])          # uneven and even shards
def test_two_rank_gloo_rerank_equals_single_process(tmp_path, nq):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), nq, 7, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert a.shape == (nq, 7) and (a == b).all()


# ---- sharded flat inner-product index (retrieval.FlatIPIndexer) under gloo ------------------------------

def _index_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import np_oracle as O
    from matchmaker_amd.retrieval import FlatIPIndexer

    def topk_fn(q, c, k):                                       # oracle stand-in for ops.dot_topk
        s, i = O.dot_topk(q.float().numpy(), c.float().numpy(), k)
        return torch.from_numpy(s), torch.from_numpy(i)

    def merge_fn(s, ids, k):                                    # oracle stand-in for ops.topk_merge
        s = s.clone()
        s[ids < 0] = float("-inf")
        order = torch.sort(s, dim=1, descending=True, stable=True).indices[:, :k]
        return torch.gather(s, 1, order), torch.gather(ids, 1, order)

    rng = np.random.default_rng(17)                             # every rank is handed the same full lists
    E, n = 40, 1001                                             # odd size: uneven shards; E is padded to 128
    chunks = [rng.standard_normal((400, E)).astype(np.float16), rng.standard_normal((601, E)).astype(np.float16)]
    ids = [np.arange(0, 400, dtype=np.int64) * 3 + 5, np.arange(400, 1001, dtype=np.int64) * 3 + 5]
    ix = FlatIPIndexer({"token_dim": E}, device="cpu", topk_fn=topk_fn, merge_fn=merge_fn)
    ix.prepare(chunks)
    ix.index(ids, chunks)
    lo, hi = __import__("matchmaker_amd.sharding", fromlist=["shard_range"]).shard_range(n, world, rank)
    assert ix.vectors.shape == (hi - lo, 128) and ix.ids.tolist() == np.concatenate(ids)[lo:hi].tolist()
    qv = rng.standard_normal((6, E)).astype(np.float32)
    s, i = ix.search(qv, 25)
    allv = np.concatenate(chunks).astype(np.float32)
    ref_s, ref_i = O.dot_topk(qv.astype(np.float16).astype(np.float32), allv, 25)
    np.testing.assert_allclose(s, ref_s, atol=1e-5)
    assert (i == ref_i * 3 + 5).all(), "merged ids differ from the single-index result"
    np.save(os.path.join(out_dir, f"ids{rank}.npy"), i)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharded_index_equals_single_index(tmp_path):
    world = 2
    mp.spawn(_index_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "ids0.npy"), np.load(tmp_path / "ids1.npy")
    assert a.shape == (6, 25) and (a == b).all()
