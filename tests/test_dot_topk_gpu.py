"""GPU parity tests for the brute-force inner-product top-k (mm_dot_topk_fwd / mm_topk_merge) vs the
oracle's restatement of faiss IndexFlatIP.search (retrieval/faiss_indices.py:29-36, bert_dot.py:62).
Scores within 1e-2 (16-bit inputs, fp32 accumulation); the returned SET must be the exact top-k of the
scores the device computed: checked against an fp64 ranking with a tie margin."""
import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from tests import util

pytestmark = pytest.mark.gpu


def _check(q, c, k, s, idx, tol=util.TOL_BF16):
    """q, c: fp32 numpy of the 16-bit values the device saw"""
    nq, N = q.shape[0], c.shape[0]
    full = q.astype(np.float64) @ c.astype(np.float64).T
    ref_s, ref_i = O.dot_topk(q, c, k, dtype=np.float64)
    kk = min(k, N)
    assert s.shape == (nq, k) and idx.shape == (nq, k)
    if kk < k:
        assert (idx[:, kk:] == -1).all() and np.isneginf(s[:, kk:]).all()
    # descending, valid, unique rows
    assert (np.diff(s[:, :kk], axis=1) <= 0).all()
    for r in range(nq):
        assert len(set(idx[r, :kk].tolist())) == kk and idx[r, :kk].min() >= 0 and idx[r, :kk].max() < N
    # reported scores are the true inner products
    got = np.take_along_axis(full, idx[:, :kk], axis=1)
    np.testing.assert_allclose(s[:, :kk], got, atol=tol, rtol=1e-3)
    # exactness of the set: nothing outside the returned set beats the k-th returned score by more
    # than the fp32-accumulation noise, and the k-th reference score is matched
    np.testing.assert_allclose(s[:, :kk], ref_s[:, :kk], atol=tol, rtol=1e-3)
    kth = got[:, kk - 1]
    for r in range(nq):
        mask = np.ones(N, bool)
        mask[idx[r, :kk]] = False
        if mask.any():
            assert full[r, mask].max() <= kth[r] + 1e-3 * (1 + abs(kth[r])), r


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("nq,N,E,k", [(5, 1000, 128, 10), (37, 5000, 768, 100), (130, 40000, 768, 1000),
                                      (300, 70001, 256, 100), (3, 17, 128, 10), (64, 4096, 384, 1000),
                                      (1, 33, 512, 40)])
def test_dot_topk_matches_flat_index_semantics(dtype, nq, N, E, k):
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(nq * 7 + N)
    c = torch.randn(N, E, generator=g).to(dtype)
    q = torch.randn(nq, E, generator=g).to(dtype)
    # a few near-duplicate documents of query 0 so that its head is well separated
    c[:3] = q[0] * torch.tensor([1.5, 1.25, 1.0])[:, None].to(dtype)
    s, idx = ops.dot_topk(q.to(dev), c.to(dev), k)
    _check(q.float().numpy(), c.float().numpy(), k, s.cpu().numpy(), idx.cpu().numpy(),
           tol=util.TOL_BF16 if E <= 256 else 5e-2)
    assert idx[0, 0].item() == 0 and idx[0, 1].item() == 1


@pytest.mark.parametrize("nq,N,k", [(200, 3000, 1000),    # no sampling (N <= 4096): every score passes, partial last block
                                    (150, 5000, 1000),    # sampled threshold near 0: half of all scores pass (on-demand flushes)
                                    (200, 40, 10), (200, 31, 5),   # two blocks / one partial block per range
                                    (260, 70001, 100)])   # two query groups, partial last block, sub-slices
def test_dot_topk_two_tile_dim768_form_edges(nq, N, k):
    """The dim-768 / > 128-query instantiation tests block b - 1 inside block b's K loop (two accumulator sets: csrc/dot_topk.hip
    DEEP): first block of a range (the other set holds zeros — which pass a threshold <= 0), last block tested behind the
    loop, survivor rates far above the staging area's schedule."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(nq + N)
    E = 768
    c = torch.randn(N, E, generator=g).half()
    q = torch.randn(nq, E, generator=g).half()
    s, idx = ops.dot_topk(q.to(dev), c.to(dev), k)
    _check(q.float().numpy(), c.float().numpy(), k, s.cpu().numpy(), idx.cpu().numpy(), tol=5e-2)


def test_dot_topk_skewed_scores_need_threshold_reruns():
    """A collection whose strided sample misses the dense head: the sampled threshold lets too few /
    too many candidates through and the host-side bisection must still deliver the exact top-k."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(5)
    N, E, k, nq = 60000, 128, 1000, 4
    c = (torch.randn(N, E, generator=g) * 0.05).half()
    q = torch.randn(nq, E, generator=g).half()
    # 3000 documents strongly aligned with query 1, placed on odd rows only (the sample stride is 3: mostly missed)
    rows = torch.arange(1, 6001, 2)
    c[rows] = (q[1].float() * (1.0 + 0.001 * torch.arange(rows.numel())[:, None])).half() * 0.1
    s, idx = ops.dot_topk(q.to(dev), c.to(dev), k)
    _check(q.float().numpy(), c.float().numpy(), k, s.cpu().numpy(), idx.cpu().numpy())


def test_topk_merge_and_indexer_surface():
    from matchmaker_amd import ops
    from matchmaker_amd.retrieval import FlatIPIndexer
    dev = util.require_gpu()
    rng = np.random.default_rng(3)
    # merge: 4 "shards" of per-shard top-5 -> top-5
    sc = np.sort(rng.standard_normal((6, 4, 5)).astype(np.float32), axis=-1)[..., ::-1].copy()
    ids = rng.permutation(6 * 20).reshape(6, 4, 5).astype(np.int64)
    sc[0, 3, 2:] = -np.inf
    ids[0, 3, 2:] = -1
    ms, mi = ops.topk_merge(torch.from_numpy(sc.reshape(6, 20)).to(dev), torch.from_numpy(ids.reshape(6, 20)).to(dev), 5)
    flat_s, flat_i = sc.reshape(6, 20), ids.reshape(6, 20)
    order = np.argsort(-flat_s, axis=1, kind="stable")[:, :5]
    np.testing.assert_array_equal(ms.cpu().numpy(), np.take_along_axis(flat_s, order, 1))
    np.testing.assert_array_equal(mi.cpu().numpy(), np.take_along_axis(flat_i, order, 1))
    # the FaissIdIndexer surface (faiss_indices.py:22-36): index(ids, chunks) + search(query, top_n)
    E, N = 96, 3000                                   # E is padded to 128 inside
    chunks = [rng.standard_normal((1000, E)).astype(np.float16) for _ in range(3)]
    ext = [np.arange(i * 1000, (i + 1) * 1000, dtype=np.int64) * 10 + 7 for i in range(3)]   # IndexIDMap ids
    ix = FlatIPIndexer({"token_dim": E}, device=dev)
    ix.prepare(chunks)
    ix.index(ext, chunks)
    qv = rng.standard_normal((9, E)).astype(np.float32)
    s, i = ix.search(qv, 50)
    allv = np.concatenate(chunks).astype(np.float32)
    ref_s, ref_i = O.dot_topk(qv.astype(np.float16).astype(np.float32), allv, 50)
    np.testing.assert_allclose(s, ref_s, atol=util.TOL_BF16, rtol=1e-3)
    agree = np.mean([len(set(a) & set(b)) / 50 for a, b in zip(i.tolist(), (ref_i * 10 + 7).tolist())])
    assert agree > 0.995, agree
    s1, i1 = ix.search(qv[0], 5)                       # a single vector is accepted like faiss_indices.py:31-32
    assert s1.shape == (1, 5) and (i1[0] == i[0, :5]).all()
