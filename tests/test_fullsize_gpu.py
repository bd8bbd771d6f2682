"""GPU: the largest single-GPU workloads BASELINE.json names, checked through size-independent properties.

* configs[3] (MSMARCO-dev re-ranking, 6,980 queries x 1000 candidates over 8 GPUs): ONE rank's shard — 873 queries,
  873,000 pairs, 40 GB of bf16 token embeddings resident — determinism, exact power-of-two linearity, whole
  queries against the oracle.
* configs[4] (BERT_DOT brute force over 8.8 M passages on 8 GPUs): one rank's shard of 1,105,228 x 768 fp16 vectors
  against all 6,980 queries — determinism, planted documents found at their ranks, a sample of queries against an
  exact fp32 ranking (torch matmul + topk on the device as the checker; the numpy oracle covers the small cases).
"""
import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from tests import util

pytestmark = pytest.mark.gpu


def test_maxsim_one_rank_shard_of_config4():
    from matchmaker_amd import ops, sharding, synth
    dev = util.require_gpu()
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 100e9:
        pytest.skip("needs ~85 GB free HBM (40 GB shard + the generator's scratch)")
    s0, s1 = sharding.shard_range(6980, 8, 0)
    nq, C = s1 - s0, 1000
    assert nq == 873
    q, d, q_len, d_len = synth.colbert_batch(nq, C, dtype=torch.bfloat16, device=dev, lengths="msmarco", seed=44)
    out = ops.maxsim(q, d, q_len, d_len, pairs_per_query=C)
    assert out.shape == (nq * C,) and torch.isfinite(out).all()
    assert torch.equal(out, ops.maxsim(q, d, q_len, d_len, pairs_per_query=C)), "non-deterministic"
    half = ops.maxsim((q.float() * 0.5).to(torch.bfloat16), d, q_len, d_len, pairs_per_query=C)
    assert torch.equal(half, out * 0.5)
    # EVERY query of the shard: scores and rank order.  Oracle = the torch port of colbert.py:68-75 on CPU tensors in fp32
    # and fp64 (multi-threaded: 16 candidate lists per call; the numpy restatement took 130 s for the 873 lists)
    from oracle import torch_port as TP
    rows = []
    G = 16
    for i0 in range(0, nq, G):
        i1 = min(nq, i0 + G)
        n = i1 - i0
        dn = d[i0 * C:i1 * C].float().cpu()
        dm = synth.len_to_mask(d_len[i0 * C:i1 * C], 180).cpu()
        qm = synth.len_to_mask(q_len[i0:i1], 32).cpu().repeat_interleave(C, 0)
        qr = q[i0:i1].float().cpu().repeat_interleave(C, 0)
        with torch.no_grad():
            ref = TP.maxsim_forward(qr, dn, qm, dm).numpy()
            ref64 = TP.maxsim_forward(qr.double(), dn.double(), qm, dm).numpy()
        got = out[i0 * C:i1 * C].cpu().numpy()
        np.testing.assert_allclose(got, ref, atol=util.TOL_BF16)
        for j in range(n):
            sl = slice(j * C, (j + 1) * C)
            rows.append(util.rank_parity(got[sl], ref[sl], ref64[sl], (1, 10, 100, 1000), label=f"shard query {i0 + j}"))
    assert util.rank_report("config4_one_rank_shard", rows) >= 0.99
    # the ranking this rank contributes is a permutation of its candidates, best first
    ranking = sharding.rank_candidates(out.view(nq, C))
    assert ranking.shape == (nq, C)
    top = torch.gather(out.view(nq, C), 1, ranking)
    assert (top[:, :-1] >= top[:, 1:]).all()


def test_dot_topk_one_rank_shard_of_config5():
    from matchmaker_amd import ops
    dev = util.require_gpu()
    N, E, nq, k = 1105228, 768, 6980, 1000
    g = torch.Generator(device=dev).manual_seed(55)
    c = torch.empty(N, E, dtype=torch.float16, device=dev)
    for lo in range(0, N, 1 << 17):                        # chunked: no fp32 copy of the shard
        hi = min(N, lo + (1 << 17))
        c[lo:hi] = (torch.randn(hi - lo, E, generator=g, device=dev) * 0.05).half()
    q = (torch.randn(nq, E, generator=g, device=dev) * 0.05).half()
    planted = torch.tensor([5, 777777, N - 1], device=dev)
    c[planted] = (q[123].float() * torch.tensor([3.0, 2.5, 2.0], device=dev)[:, None]).half()
    s, idx = ops.dot_topk(q, c, k)
    assert s.shape == (nq, k) and idx.shape == (nq, k)
    assert idx[123, :3].tolist() == planted.tolist(), "planted near-duplicates must lead query 123's list"
    assert (s[:, :-1] >= s[:, 1:]).all() and (idx >= 0).all() and (idx < N).all()
    s2, idx2 = ops.dot_topk(q, c, k)
    assert torch.equal(s, s2) and torch.equal(idx, idx2), "non-deterministic"
    # 256 sampled queries (every 27th + the planted one) against an fp32 checker: torch.topk over the fp32 products, formed
    # in slabs of 2^18 passages (the full [256, 1.1 M] fp32 score matrix is 1.1 GB and stays on the device)
    sel = torch.unique(torch.cat([torch.arange(0, nq, 27, device=dev)[:255], torch.tensor([123], device=dev)]))
    full = torch.cat([q[sel].float() @ c[lo:lo + (1 << 18)].float().T for lo in range(0, N, 1 << 18)], dim=1)
    ref_s, ref_i = torch.topk(full, k, dim=1)
    np.testing.assert_allclose(s[sel].cpu().numpy(), ref_s.cpu().numpy(), atol=2e-3, rtol=1e-3)
    same = idx[sel] == ref_i                                            # identical rows in identical order
    differing = torch.nonzero(~same.all(dim=1)).flatten().tolist()
    for r in differing:
        got, want = set(idx[sel[r]].tolist()), set(ref_i[r].tolist())
        # the sets agree except for candidates within accumulation noise of the k-th score; inside the list two rows may
        # swap places only when their fp32 scores are that close
        kth = float(ref_s[r, -1])
        for j in got ^ want:
            assert abs(float(full[r, j]) - kth) < 2e-3, (int(sel[r]), j)
        pos = torch.nonzero(~same[r]).flatten()
        assert float((full[r, idx[sel[r], pos]] - ref_s[r, pos]).abs().max()) < 2e-3, int(sel[r])
    print(f"[dot top-k] {sel.numel()} sampled queries: {sel.numel() - len(differing)} lists identical to the fp32 checker's, "
          f"{len(differing)} differ only among near-ties")
    # every returned score is the inner product of its row
    chk = torch.gather(full, 1, idx[sel])
    np.testing.assert_allclose(s[sel].cpu().numpy(), chk.cpu().numpy(), atol=2e-3, rtol=1e-3)


def test_fp32_maxsim_config2_variant_full_size():
    """The fp32 variant of configs[1] (64 queries x 1000 candidates, Q32/D180/E128) on the split-bf16 streaming
    kernel: determinism, permutation equivariance, power-of-two linearity, whole queries against the oracle."""
    from matchmaker_amd import ops, synth
    dev = util.require_gpu()
    nq, C = 64, 1000
    q, d, q_len, d_len = synth.colbert_batch(nq, C, dtype=torch.float32, device=dev, lengths="msmarco", seed=77)
    out = ops.maxsim(q, d, q_len, d_len, pairs_per_query=C)
    assert torch.equal(out, ops.maxsim(q, d, q_len, d_len, pairs_per_query=C))
    perm = torch.stack([torch.randperm(C, device=dev) + i * C for i in range(nq)]).reshape(-1)
    assert torch.equal(ops.maxsim(q, d[perm].contiguous(), q_len, d_len[perm].contiguous(), pairs_per_query=C), out[perm])
    assert torch.equal(ops.maxsim(q * 4.0, d, q_len, d_len, pairs_per_query=C), out * 4.0)
    for i in (0, 31, 63):
        dn = d[i * C:(i + 1) * C].cpu().numpy()
        dm = synth.len_to_mask(d_len[i * C:(i + 1) * C], 180).cpu().numpy()
        qm = np.repeat(synth.len_to_mask(q_len[i:i + 1], 32).cpu().numpy(), C, 0)
        ref = O.maxsim_paired(np.repeat(q[i:i + 1].cpu().numpy(), C, 0), dn, qm, dm, dtype=np.float64)
        np.testing.assert_allclose(out[i * C:(i + 1) * C].cpu().numpy(), ref, atol=util.TOL_FP32, rtol=1e-6)


def test_idcm_sampler_shape_full_size():
    """IDCM's default sampler shape at scale: 2,000 documents x (up to) 41 passages of 64 tokens, 768-wide vectors,
    one query row per document (pair_query), floor 1e-4 — determinism, invariance to a power-of-two rescaling of
    the vectors (cosines are scale free), sampled passages against the oracle."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator(device=dev).manual_seed(88)
    B, Q, D, E = 2000, 30, 64, 768
    n_pass = torch.randint(1, 42, (B,), generator=g, device=dev)
    pq = torch.repeat_interleave(torch.arange(B, device=dev), n_pass)
    P = int(pq.numel())
    q = torch.relu(torch.randn(B, Q, E, generator=g, device=dev))
    d = torch.relu(torch.randn(P, D, E, generator=g, device=dev))
    qm = (torch.arange(Q, device=dev)[None] < torch.randint(2, 13, (B,), generator=g, device=dev)[:, None]).float()
    dm = (torch.arange(D, device=dev)[None] < torch.randint(8, D + 1, (P,), generator=g, device=dev)[:, None]).float()
    prm = [torch.tensor([1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9], device=dev), torch.full((11,), 0.1, device=dev),
           torch.rand(11, generator=g, device=dev) + 0.5, torch.randn(11, generator=g, device=dev)]
    s = ops.kernel_pool(q, d, qm, dm, *prm, clamp_min=1e-4, pair_query=pq)
    assert s.shape == (P,) and torch.isfinite(s).all()
    assert torch.equal(s, ops.kernel_pool(q, d, qm, dm, *prm, clamp_min=1e-4, pair_query=pq))
    s2 = ops.kernel_pool(q * 2.0, d * 0.5, qm, dm, *prm, clamp_min=1e-4, pair_query=pq)
    torch.testing.assert_close(s2, s, rtol=0, atol=1e-4)
    sel = torch.tensor([0, P // 3, P - 1], device=dev)
    ref = O.idcm_sampler_scores(q[pq[sel]].cpu().numpy(), d[sel].cpu().numpy(), qm[pq[sel]].cpu().numpy(), dm[sel].cpu().numpy(),
                                prm[0].cpu().numpy(), prm[1].cpu().numpy(), prm[2].cpu().numpy(), prm[3].cpu().numpy(), 0.0,
                                dtype=np.float64)
    np.testing.assert_allclose(s[sel].cpu().numpy(), ref, atol=util.TOL_FP32, rtol=1e-5)
