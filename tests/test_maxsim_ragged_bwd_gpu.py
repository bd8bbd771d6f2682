"""GPU tests: ragged (CSR) MaxSim over a token store (mm_maxsim_ragged_fwd; dense_retrieval.py:398-412
+ colbert.py:100-112) and the MaxSim backward (mm_maxsim_bwd; autograd through colbert.py:68-75)."""
import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from oracle import torch_port as TP
from tests import util

pytestmark = pytest.mark.gpu


def _store(rng, n_docs, E, dtype, max_len=70, empty=()):
    lens = rng.integers(1, max_len + 1, n_docs)
    for i in empty:
        lens[i] = 0
    end = np.cumsum(lens)
    begin = end - lens
    tok = torch.from_numpy(rng.standard_normal((int(end[-1]) if n_docs else 0, E)).astype(np.float32)).to(dtype)
    return tok, begin.astype(np.int64), end.astype(np.int64)


@pytest.mark.parametrize("dtype,E,Q,tol", [(torch.float16, 768, 32, util.TOL_BF16), (torch.float16, 768, 38, util.TOL_BF16), (torch.bfloat16, 128, 32, util.TOL_BF16),
                                          (torch.float16, 128, 7, util.TOL_BF16), (torch.float32, 128, 32, util.TOL_FP32),
                                          (torch.float32, 24, 40, util.TOL_FP32), (torch.bfloat16, 256, 20, util.TOL_BF16)])
def test_ragged_matches_per_document_aggregation(dtype, E, Q, tol):
    """every candidate scored exactly like the reference's per-candidate forward_aggregation call"""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    rng = np.random.default_rng(E + Q)
    n_docs, nq, C = 300, 3, 41
    tok, begin, end = _store(rng, n_docs, E, dtype, max_len=100)
    q = torch.from_numpy(rng.standard_normal((nq, Q, E)).astype(np.float32) / np.sqrt(E)).to(dtype)
    cand = rng.integers(0, n_docs, (nq, C))          # arbitrary, repeated, unordered candidates
    cand[0, 0] = n_docs - 1                           # the last document of the store (tail clamp)
    b = torch.from_numpy(begin[cand.reshape(-1)]).to(dev)
    e = torch.from_numpy(end[cand.reshape(-1)]).to(dev)
    out = ops.maxsim_ragged(q.to(dev), tok.to(dev), b, e, None, pairs_per_query=C).cpu().numpy().reshape(nq, C)
    qf, tf = q.float(), tok.float()
    for i in range(nq):
        for j in range(C):
            doc = tf[begin[cand[i, j]]: end[cand[i, j]]]
            ref = float(TP.maxsim_aggregation(qf[i:i + 1], doc.unsqueeze(0)))
            assert abs(out[i, j] - ref) <= tol + 1e-4 * abs(ref), (i, j, out[i, j], ref)
    # the padded kernel on the same documents gives the same scores
    D = int((end - begin).max())
    dpad = torch.zeros((nq * C, D, E), dtype=dtype)
    lens = (end - begin)[cand.reshape(-1)]
    for p, c in enumerate(cand.reshape(-1)):
        dpad[p, : lens[p]] = tok[begin[c]: end[c]]
    pad = ops.maxsim(q.to(dev), dpad.to(dev), None, torch.from_numpy(lens.astype(np.int32)).to(dev), pairs_per_query=C)
    # (a padded document adds the -1000 sentinel to the max; it never wins against real similarities here)
    np.testing.assert_allclose(out.reshape(-1), pad.cpu().numpy(), atol=tol, rtol=1e-4)


def test_ragged_empty_ranges_query_masks_and_store_helper(tmp_path):
    from matchmaker_amd import ops
    from matchmaker_amd.token_store import TokenStore, write_reference_store
    dev = util.require_gpu()
    rng = np.random.default_rng(11)
    E, Q = 128, 32
    docs = [rng.standard_normal((int(rng.integers(1, 60)), E)).astype(np.float16) for _ in range(57)]
    ids = [f"d{i}" for i in range(57)]
    write_reference_store(str(tmp_path), docs, ids, token_block_size=500, token_dtype="float16")
    st = TokenStore.load(str(tmp_path), E, "float16", 500, dev)
    q = torch.from_numpy(rng.standard_normal((2, Q, E)).astype(np.float32) / 11.0).half()
    q[1, 20:] = 0                                     # encode-time masking (colbert.py:95-96)
    cands = [[ids[i] for i in (3, 56, 0, 17, 3)], [ids[i] for i in (40, 41)]]
    res = st.aggregate(q.to(dev), cands)
    for i, lst in enumerate(cands):
        assert [sid for sid, _ in res[i]] == lst
        for (sid, sc) in res[i]:
            doc = torch.from_numpy(docs[ids.index(sid)]).float().unsqueeze(0)
            ref = float(TP.maxsim_aggregation(q[i:i + 1].float(), doc))
            assert abs(sc - ref) < util.TOL_BF16, (sid, sc, ref)
    # empty range = fully padded document; query lengths as masks
    b = torch.tensor([0, 5, 5], dtype=torch.int64, device=dev)
    e = torch.tensor([4, 5, 9], dtype=torch.int64, device=dev)
    qlen = torch.tensor([9], dtype=torch.int32, device=dev)
    out = ops.maxsim_ragged(q[:1].to(dev), st.tokens, b, e, qlen, pairs_per_query=3).cpu().numpy()
    assert out[1] == -9000.0
    t = st.tokens.float().cpu()
    for p, (x, y) in enumerate([(0, 4), (5, 9)]):
        ref = (q[0, :9].float() @ t[x:y].T).max(-1).values.sum()
        assert abs(out[[0, 2][p]] - float(ref)) < util.TOL_BF16


def test_backward_of_a_document_too_long_for_the_lds_row_masks():
    """mm_maxsim_bwd keeps, per document row, the bit set of the query tokens whose first arg-max it is in LDS; when
    D x ceil(Q / 32) words do not fit (here 5,200 x 3 x 4 B > 60 KB) it scans the arg-max table instead — same gradients."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(77)
    B, Q, D, E = 2, 70, 5200, 16
    q = torch.randn(B, Q, E, generator=g) / E ** 0.5
    d = torch.randn(B, D, E, generator=g)
    d[0, 4000] = d[0, 17]                              # an exact tie far apart: the first position takes the gradient
    qm = (torch.arange(Q)[None] < torch.tensor([[Q], [51]])).long()
    dm = (torch.arange(D)[None] < torch.tensor([[D], [4999]])).long()
    go = torch.randn(B, generator=g)
    _, ref_gq, ref_gd = TP.maxsim_forward_backward(q, d, qm, dm, go)
    gq, gd = ops.maxsim_bwd(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), go.to(dev))
    np.testing.assert_allclose(gq.cpu().numpy(), ref_gq.numpy(), atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(gd.cpu().numpy(), ref_gd.numpy(), atol=1e-5, rtol=1e-5)
    assert float(gd[0, 4000].abs().max()) == 0.0


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 1e-2), (torch.float16, 1e-2)])
@pytest.mark.parametrize("B,Q,D,E", [(6, 32, 180, 128), (5, 13, 47, 64), (3, 40, 70, 24), (4, 8, 33, 768)])
def test_backward_matches_autograd_of_the_reference_ops(dtype, tol, B, Q, D, E):
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(B * 1000 + D)
    q = (torch.randn(B, Q, E, generator=g) / E ** 0.5).to(dtype)
    d = torch.randn(B, D, E, generator=g).to(dtype)
    q_len = torch.randint(1, Q + 1, (B,), generator=g)
    d_len = torch.randint(1, D + 1, (B,), generator=g)
    d_len[0] = D
    d_len[-1] = 0                                     # fully padded document: no gradient at all
    qm = (torch.arange(Q)[None] < q_len[:, None]).long()
    dm = (torch.arange(D)[None] < d_len[:, None]).long()
    dm[1, 0] = 0                                      # a hole
    go = torch.randn(B, generator=g)
    ref_out, ref_gq, ref_gd = TP.maxsim_forward_backward(q, d, qm, dm, go)
    gq, gd = ops.maxsim_bwd(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), go.to(dev))
    np.testing.assert_allclose(gq.cpu().numpy(), ref_gq.numpy(), atol=tol, rtol=1e-5)
    np.testing.assert_allclose(gd.cpu().numpy(), ref_gd.numpy(), atol=tol, rtol=1e-5)
    assert float(gd[-1].abs().max()) == 0.0 and float(gq[-1].abs().max()) == 0.0
    # through the drop-in's autograd function (what train.py's loss.backward() reaches)
    from matchmaker_amd.colbert import ColBERT
    qd = q.to(dev).requires_grad_(True)
    dd = d.to(dev).requires_grad_(True)
    s = ColBERT._score(qd, dd, qm.to(dev), dm.to(dev))
    if dtype == torch.float32:
        np.testing.assert_allclose(s.detach().cpu().numpy(), ref_out.numpy(), atol=max(tol, 1e-3), rtol=1e-4)
    else:
        # 16-bit vectors outside autocast: the reference's bmm / max / sum are 16-bit ops and so is the score (colbert.py:68-75
        # without :60's autocast) — the drop-in returns that dtype and those values (oracle: the same dtype flow)
        from oracle import np_oracle as O
        lp = np.float16 if dtype == torch.float16 else "bfloat16"
        assert s.dtype == dtype
        flow = O.maxsim_paired(q.float().numpy(), d.float().numpy(), qm.numpy(), dm.numpy(), np.float64, sim_dtype=lp, sum_dtype=lp)
        assert util.ulps16(s.detach().float().cpu().numpy(), flow, lp).max() <= 1.0
    (s * go.to(dev)).sum().backward()
    assert qd.grad.dtype == dtype and dd.grad.dtype == dtype
    np.testing.assert_allclose(qd.grad.float().cpu().numpy(), ref_gq.numpy(), atol=max(tol, 2e-2 if dtype != torch.float32 else tol), rtol=1e-2)
    np.testing.assert_allclose(dd.grad.float().cpu().numpy(), ref_gd.numpy(), atol=max(tol, 2e-2 if dtype != torch.float32 else tol), rtol=1e-2)
