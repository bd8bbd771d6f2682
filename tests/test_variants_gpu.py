"""GPU parity tests of the pooling variants: the gated kernel pooling of TK-Sparse
(mm_kernel_pool_ex_fwd / _ex_bwd with d_gate) and the IDCM passage sampler (clamp_min = 1e-4), through the
drop-in classes, against golden vectors of the real classes and against the oracle.  fp32 tolerance 1e-3
on scores (BASELINE.json)."""
import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from oracle import torch_port as TP
from tests import util

pytestmark = pytest.mark.gpu

MU = [1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9]
SIGMA = [0.1] * 11


def _params(g):
    return {k[len("param."):]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("param.")}


def test_tk_sparse_dropin_matches_reference_golden():
    from matchmaker_amd.tk_sparse import CIKM20_TK_Sparse

    class Bypass(CIKM20_TK_Sparse):      # mirrors oracle/ref_harness.make_tk_sparse(bypass_contextualizer=True)
        def forward_representation(self, emb, mask, positional_features=None):
            return emb * mask.unsqueeze(-1), emb

    dev = util.require_gpu()
    g = util.load("sparse_tk_q20_d200_e300.npz")
    m = Bypass(300, MU, SIGMA, 10, 2, 32, 300, 200, True)
    missing, unexpected = m.load_state_dict(_params(g), strict=False)
    assert not unexpected and all(k.startswith(("contextualizer", "positional")) for k in missing)
    m = m.to(dev).eval()
    t = lambda k: torch.from_numpy(np.ascontiguousarray(g[k])).float().to(dev)
    with torch.no_grad():
        score, sec, stop = m.forward(t("q"), t("d"), t("q_mask"), t("d_mask"), True)
        score2, stop2 = m.forward(t("q"), t("d"), t("q_mask"), t("d_mask"))
    np.testing.assert_allclose(stop.cpu().numpy(), g["document_stop_words"], atol=2e-5)
    np.testing.assert_allclose(sec["per_kernel"].cpu().numpy(), g["per_kernel"], atol=5e-3, rtol=1e-4)
    np.testing.assert_allclose(score.cpu().numpy(), g["score"], atol=util.TOL_FP32, rtol=1e-5)
    assert torch.equal(score, score2) and torch.equal(stop, stop2)
    assert sec["cosine_matrix_masked"].shape == (4, 20, 200)


def test_tk_sparse_full_model_matches_the_real_class_end_to_end_and_trains():
    from matchmaker_amd.tk_sparse import CIKM20_TK_Sparse
    dev = util.require_gpu()
    g = util.load("e2e_sparse_tk_q12_d70_e60.npz")
    m = CIKM20_TK_Sparse(60, MU, SIGMA, 6, 2, 32, 32, 80, True)
    m.load_state_dict(_params(g), strict=True)
    m = m.to(dev).eval()
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    with torch.no_grad():
        score, stop = m.forward(t("q"), t("d"), t("q_mask"), t("d_mask"))
    np.testing.assert_allclose(stop.cpu().numpy(), g["document_stop_words"], atol=1e-4)
    np.testing.assert_allclose(score.cpu().numpy(), g["score"], atol=5e-3, rtol=1e-4)
    # training step: score + L1 sparsity on the gate (train.py), gradients reach the stop-word MLP natively
    m.train()
    score, stop = m.forward(t("q"), t("d"), t("q_mask"), t("d_mask"))
    (score.sum() + 0.1 * stop.sum()).backward()
    for p in (m.stop_word_reducer.weight, m.stop_word_reducer2.weight, m.kernel_bin_weights.weight,
              m.kernel_alpha_scaler, m.mixer, m.mixer_stop):
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0


@pytest.mark.parametrize("B,Q,D,E", [(5, 20, 200, 300), (4, 32, 97, 100), (3, 11, 500, 200), (2, 20, 4100, 100),
                                      (3, 30, 64, 128), (4, 40, 70, 64), (2, 7, 33, 768)])
def test_gated_kernel_pool_vs_oracle(B, Q, D, E):
    """Streaming (E = 100n, D <= 4096) and generic kernels with a gate that is 0 for ~half the tokens."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    gen = torch.Generator().manual_seed(B * 1000 + D + E)
    q = torch.randn(B, Q, E, generator=gen)
    d = torch.randn(B, D, E, generator=gen)
    for b in range(B):
        d[b, (7 * b) % D] = q[b, b % Q]
        d[b, (11 * b + 3) % D] = q[b, (b + 1) % Q] + 0.1 * torch.randn(E, generator=gen)
    q_len = torch.randint(1, Q + 1, (B,), generator=gen)
    d_len = torch.randint(1, D + 1, (B,), generator=gen)
    d_len[0] = D
    qm = (torch.arange(Q)[None] < q_len[:, None]).float()
    dm = (torch.arange(D)[None] < d_len[:, None]).float()
    gate = torch.relu(torch.randn(B, D, generator=gen)) * 1.5
    gate[0, 0] = 1.0
    gate[-1] = 0.0 if B > 2 else gate[-1]                  # a document whose every token is gated off
    alpha = torch.rand(11, generator=gen) + 0.5
    w = torch.randn(11, generator=gen) * 0.3
    mu, sigma = torch.tensor(MU), torch.tensor(SIGMA)
    s, pk = ops.kernel_pool(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), mu.to(dev), sigma.to(dev), alpha.to(dev),
                            w.to(dev), return_per_kernel=True, d_gate=gate.to(dev))
    ref, ref_pk = O.tk_sparse_kernel_pool(q.numpy(), d.numpy(), qm.numpy(), dm.numpy(), (gate * dm).numpy(), MU, SIGMA,
                                          alpha.numpy(), w.numpy(), dtype=np.float64, return_per_kernel=True)
    np.testing.assert_allclose(pk.cpu().numpy(), ref_pk, atol=5e-3, rtol=1e-4)
    np.testing.assert_allclose(s.cpu().numpy(), ref, atol=util.TOL_FP32, rtol=1e-5)
    # gate == 1 everywhere is the ungated operator, bit for bit on the same kernel family
    ones = ops.kernel_pool(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), mu.to(dev), sigma.to(dev), alpha.to(dev),
                           w.to(dev), d_gate=torch.ones(B, D, device=dev))
    plain = ops.kernel_pool(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), mu.to(dev), sigma.to(dev), alpha.to(dev), w.to(dev))
    # (two wavefronts per pair on the ungated call of a 2-pair batch, one on the gated one: fp32 summation order over up to
    # 4,100 positions — 1.1e-4 on a score of 56 at D = 4100, a tenth of the contract's 1e-3)
    np.testing.assert_allclose(ones.cpu().numpy(), plain.cpu().numpy(), atol=3e-4 if D > 1000 else 1e-5, rtol=1e-6)


@pytest.mark.parametrize("B,Q,D,E", [(4, 20, 200, 300), (3, 30, 47, 64), (3, 12, 64, 128), (2, 25, 1500, 100), (3, 20, 90, 384), (2, 8, 40, 448)])
def test_gated_backward_matches_autograd_of_the_reference_ops(B, Q, D, E):
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(B * 100 + D)
    q = torch.randn(B, Q, E, generator=g)
    d = torch.randn(B, D, E, generator=g)
    for b in range(B):
        d[b, b % D] = q[b, b % Q] * 1.3 + 0.05 * torch.randn(E, generator=g)
    q_len = torch.randint(1, Q + 1, (B,), generator=g)
    d_len = torch.randint(1, D + 1, (B,), generator=g)
    d_len[0] = D
    qm = (torch.arange(Q)[None] < q_len[:, None]).float()
    dm = (torch.arange(D)[None] < d_len[:, None]).float()
    gate = (torch.relu(torch.randn(B, D, generator=g)) + 0.0) * dm
    for b in range(B):
        gate[b, b % D] = 0.8                               # the planted match stays open
    alpha = torch.rand(11, generator=g) + 0.5
    w = torch.randn(11, generator=g) * 0.3
    go = torch.randn(B, generator=g)
    mu, sigma = torch.tensor(MU), torch.tensor(SIGMA)
    leaves = [t.detach().double().clone().requires_grad_(True) for t in (q, d, alpha, w, gate)]
    q_, d_, a_, w_, g_ = leaves
    s = TP.tk_sparse_kernel_pool(q_, d_, qm.double(), dm.double(), g_.unsqueeze(1), mu.double().view(1, 1, 1, -1),
                                 sigma.double().view(1, 1, 1, -1), a_.view(1, 1, -1), w_.view(1, -1))
    s.backward(go.double())
    gq, gd, ga, gw, gg = ops.kernel_pool_bwd(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), mu.to(dev), sigma.to(dev),
                                             alpha.to(dev), w.to(dev), go.to(dev), d_gate=gate.to(dev))
    for got, want, name in ((gq, q_.grad, "grad_q"), (gd, d_.grad, "grad_d"), (ga, a_.grad, "grad_alpha"),
                            (gw, w_.grad, "grad_w"), (gg, g_.grad, "grad_gate")):
        want = want.numpy()
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(got.cpu().numpy().astype(np.float64), want, atol=2e-4 * scale, rtol=2e-3, err_msg=name)


@pytest.mark.parametrize("P,Q,D,E", [(9, 30, 64, 768), (7, 12, 64, 128), (5, 30, 64, 384), (6, 20, 200, 300)])
def test_idcm_sampler_scores_vs_oracle_forward_and_backward(P, Q, D, E):
    """sampler_scores (clamp 1e-4, bias) on ReLU-like vectors (zero rows included, as the sampler's CNN makes
    them) vs the oracle; gradients vs autograd through the torch port of sigir21_idcm.py:169-186."""
    from matchmaker_amd import idcm
    dev = util.require_gpu()
    gen = torch.Generator().manual_seed(P * 1000 + E)
    q = torch.relu(torch.randn(P, Q, E, generator=gen))
    d = torch.relu(torch.randn(P, D, E, generator=gen))
    d[0, 5] = 0.0
    d[1, 3] = q[1, 2]
    q_len = torch.randint(1, Q + 1, (P,), generator=gen)
    d_len = torch.randint(1, D + 1, (P,), generator=gen)
    qm = (torch.arange(Q)[None] < q_len[:, None]).long()
    dm = (torch.arange(D)[None] < d_len[:, None])
    lin = torch.nn.Linear(11, 1)
    with torch.no_grad():
        lin.weight.uniform_(-0.5, 0.5, generator=gen)
        lin.bias.fill_(0.3)
    alpha = (torch.rand(1, 1, 11, generator=gen) + 0.5)
    mu, sigma = torch.tensor(MU).view(1, 1, 1, -1), torch.tensor(SIGMA).view(1, 1, 1, -1)
    ref = O.idcm_sampler_scores(q.numpy(), d.numpy(), qm.numpy(), dm.numpy(), MU, SIGMA, alpha.numpy().reshape(-1),
                                lin.weight.detach().numpy().reshape(-1), 0.3, dtype=np.float64)
    lin_d = torch.nn.Linear(11, 1).to(dev)
    lin_d.load_state_dict(lin.state_dict())
    with torch.no_grad():
        s = idcm.sampler_scores(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), mu.to(dev), sigma.to(dev), alpha.to(dev), lin_d)
    assert s.shape == (P, 1)
    np.testing.assert_allclose(s[:, 0].cpu().numpy(), ref, atol=util.TOL_FP32, rtol=1e-5)
    # backward.  An exactly-zero row is left out: its gradient is 1/eps times the upstream one (eps = 1e-12 in
    # torch.nn.functional.normalize, 1e-13 in the native cosine) - meaningless in both, and not equal.
    d[0, 5] = torch.relu(torch.randn(E, generator=gen))
    q64, d64 = q.double().requires_grad_(True), d.double().requires_grad_(True)
    a64 = alpha.double().requires_grad_(True)
    w64, b64 = lin.weight.detach().double().requires_grad_(True), lin.bias.detach().double().requires_grad_(True)
    TP.idcm_sampler_scores(q64, d64, qm.double(), dm.double(), mu.double(), sigma.double(), a64, w64, b64).sum().backward()
    qd, dd = q.to(dev).requires_grad_(True), d.to(dev).requires_grad_(True)
    ad = alpha.to(dev).requires_grad_(True)
    idcm.sampler_scores(qd, dd, qm.to(dev), dm.to(dev), mu.to(dev), sigma.to(dev), ad, lin_d).sum().backward()
    for got, want, name in ((qd.grad, q64.grad, "grad_q"), (dd.grad, d64.grad, "grad_d"), (ad.grad, a64.grad, "grad_alpha"),
                            (lin_d.weight.grad, w64.grad, "grad_w"), (lin_d.bias.grad, b64.grad, "grad_bias")):
        want = want.numpy()
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(got.cpu().numpy().astype(np.float64), want, atol=2e-4 * scale, rtol=2e-3, err_msg=name)


@pytest.mark.parametrize("E", [128, 384, 300, 60, 768])
def test_pair_query_indexing_equals_one_query_copy_per_pair(E):
    """Ragged groups (IDCM: a different number of passages per document): q [n_queries] + pair_query must
    equal the pair-per-row layout the reference materialises (sigir21_idcm.py:143-144), on every kernel."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    gen = torch.Generator().manual_seed(E)
    nq, Q, D = 7, 30, 64
    groups = [5, 0, 1, 9, 3, 0, 4]
    pq = torch.repeat_interleave(torch.arange(nq), torch.tensor(groups))
    P = int(pq.numel())
    q = torch.randn(nq, Q, E, generator=gen)
    d = torch.randn(P, D, E, generator=gen)
    d[2, 3] = q[pq[2], 4]
    qm = (torch.arange(Q)[None] < torch.randint(1, Q + 1, (nq,), generator=gen)[:, None]).float()
    dm = (torch.arange(D)[None] < torch.randint(1, D + 1, (P,), generator=gen)[:, None]).float()
    prm = [torch.tensor(MU), torch.tensor(SIGMA), torch.rand(11, generator=gen) + 0.5, torch.randn(11, generator=gen)]
    prm = [t.to(dev) for t in prm]
    a = ops.kernel_pool(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), *prm, clamp_min=1e-4, pair_query=pq.to(dev))
    b = ops.kernel_pool(q[pq].to(dev), d.to(dev), qm[pq].to(dev), dm.to(dev), *prm, clamp_min=1e-4)
    assert torch.equal(a, b)
    ref = O.idcm_sampler_scores(q[pq].numpy(), d.numpy(), qm[pq].numpy(), dm.numpy(), MU, SIGMA, prm[2].cpu().numpy(),
                                prm[3].cpu().numpy(), 0.0, dtype=np.float64)
    np.testing.assert_allclose(a.cpu().numpy(), ref, atol=util.TOL_FP32, rtol=1e-5)
    with pytest.raises(ops.NativeError):
        ops.kernel_pool(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), *prm, pair_query=(pq + 1).to(dev))


def _tiny_distilbert():
    from transformers import DistilBertConfig, DistilBertModel
    cfg = DistilBertConfig(vocab_size=200, dim=64, n_heads=4, hidden_dim=128, n_layers=2,
                           max_position_embeddings=128, dropout=0.0, attention_dropout=0.0)
    return DistilBertModel(cfg).eval()


@pytest.mark.parametrize("fname,ctx_kind", [("idcm_ck.npz", "ck"), ("idcm_ck_small.npz", "ck-small")])
def test_idcm_dropin_matches_the_real_class_end_to_end(fname, ctx_kind):
    """idcm_*.npz: outputs of the REAL IDCM.forward (sigir21_idcm.py:111-274) around a tiny random DistilBERT.
    The drop-in loads its state_dict strictly; sampler pooling native, everything else PyTorch on the GPU."""
    from tests import idcm_host_fixture as idcm
    dev = util.require_gpu()
    g = util.load(fname)
    m = idcm.IDCM(_tiny_distilbert(), sample_n=2, sample_context=ctx_kind, top_k_chunks=2, chunk_size=50, overlap=7,
                  padding_idx=0, sample_train_type="mseloss")
    m.load_state_dict(_params(g), strict=True)
    m = m.to(dev).eval()
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    query = {"input_ids": t("q_ids"), "attention_mask": t("q_mask")}
    doc = {"input_ids": t("d_ids"), "attention_mask": t("d_mask")}
    with torch.no_grad():
        score, bert_scores, sec, _, _ = m.forward(query, doc, use_fp16=False, output_secondary_output=True)
    assert (sec["packed_indices"].cpu().numpy() == g["packed_indices"]).all()
    np.testing.assert_allclose(sec["sampling_scores"].cpu().numpy(), g["sampling_scores"], atol=util.TOL_FP32, rtol=1e-5)
    np.testing.assert_allclose(bert_scores.cpu().numpy(), g["bert_scores"], atol=1e-4)     # same passages selected
    np.testing.assert_allclose(score.cpu().numpy(), g["score"], atol=1e-4)
    # a training step of the sampler against the BERT passage scores (mseloss): gradients through the native op
    m.train()
    _, _, loss, orders = m.forward(query, doc, use_fp16=False)
    loss[0][0].backward()
    assert m.sampling_binweights.weight.grad.abs().sum() > 0 and torch.isfinite(m.kernel_alpha_scaler.grad).all()
    assert any(p.grad is not None and p.grad.abs().sum() > 0 for p in m.sample_cnn3.parameters())


@pytest.mark.parametrize("K,E", [(1, 300), (5, 64), (21, 300), (32, 128), (12, 100)])
def test_kernel_counts_other_than_eleven(K, E):
    """tk_kernels_mu / knrm_kernels are configuration: any K <= 32 runs (generic kernel with run-time K), forward
    and backward; KNRM's own kernel table (knrm.py:100-130) at K = 21 goes through the drop-in."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    gen = torch.Generator().manual_seed(K * 10 + E)
    B, Q, D = 4, 14, 70
    q = torch.randn(B, Q, E, generator=gen)
    d = torch.randn(B, D, E, generator=gen)
    d[0, 3] = q[0, 1]
    qm = (torch.arange(Q)[None] < torch.randint(1, Q + 1, (B,), generator=gen)[:, None]).float()
    dm = (torch.arange(D)[None] < torch.randint(1, D + 1, (B,), generator=gen)[:, None]).float()
    mu = torch.linspace(1.0, -0.9, K) if K > 1 else torch.tensor([0.3])
    sigma = torch.full((K,), 0.1)
    alpha, w = torch.rand(K, generator=gen) + 0.5, torch.randn(K, generator=gen)
    s, pk = ops.kernel_pool(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), mu.to(dev), sigma.to(dev), alpha.to(dev), w.to(dev),
                            return_per_kernel=True)
    ref, ref_pk = O.tk_kernel_pool(q.numpy(), d.numpy(), qm.numpy(), dm.numpy(), mu.numpy(), sigma.numpy(), alpha.numpy(),
                                   w.numpy(), dtype=np.float64, return_per_kernel=True)
    assert pk.shape == (B, K)
    np.testing.assert_allclose(pk.cpu().numpy(), ref_pk, atol=5e-3, rtol=1e-4)
    np.testing.assert_allclose(s.cpu().numpy(), ref, atol=util.TOL_FP32 * max(1.0, K / 11), rtol=1e-5)
    leaves = [t.double().clone().requires_grad_(True) for t in (q, d, alpha, w)]
    go = torch.randn(B, generator=gen)
    TP.tk_kernel_pool(leaves[0], leaves[1], qm.double(), dm.double(), mu.double().view(1, 1, 1, -1),
                      sigma.double().view(1, 1, 1, -1), leaves[2].view(1, 1, -1), leaves[3].view(1, -1)).backward(go.double())
    got = ops.kernel_pool_bwd(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), mu.to(dev), sigma.to(dev), alpha.to(dev),
                              w.to(dev), go.to(dev))
    for g_, leaf, name in zip(got, leaves, ("grad_q", "grad_d", "grad_alpha", "grad_w")):
        want = leaf.grad.numpy()
        np.testing.assert_allclose(g_.cpu().numpy().astype(np.float64), want, atol=2e-4 * max(1.0, np.abs(want).max()),
                                   rtol=2e-3, err_msg=name)
    with pytest.raises(ops.NativeError):
        z = torch.zeros(33, device=dev)
        ops.kernel_pool(q.to(dev), d.to(dev), None, None, z, z + 0.1, z + 1, z)


def test_knrm_with_21_kernels():
    from matchmaker_amd.knrm import KNRM
    dev = util.require_gpu()
    torch.manual_seed(21)
    m = KNRM(21).to(dev).eval()
    B, Q, D, E = 3, 10, 40, 300
    q, d = torch.randn(B, Q, E), torch.randn(B, D, E)
    d[0, 2] = q[0, 0]
    qm = (torch.arange(Q)[None] < torch.tensor([10, 4, 1])[:, None]).float()
    dm = (torch.arange(D)[None] < torch.tensor([40, 7, 22])[:, None]).float()
    qi, di = q * qm[..., None], d * dm[..., None]
    with torch.no_grad():
        s = m.forward(qi.to(dev), di.to(dev), qm.to(dev), dm.to(dev))
    ref = O.knrm_kernel_pool(qi.numpy(), di.numpy(), qm.numpy(), dm.numpy(), np.asarray(m.mu.cpu()).reshape(-1),
                             np.asarray(m.sigma.cpu()).reshape(-1), m.dense.weight.detach().cpu().numpy().reshape(-1),
                             dtype=np.float64)
    np.testing.assert_allclose(s.cpu().numpy(), ref, atol=util.TOL_FP32, rtol=1e-5)


@pytest.mark.parametrize("E,D", [(128, 64), (64, 96), (128, 33)])
def test_short_document_form_with_two_wavefronts_per_simd_matches_the_oracle(E, D):
    """kernel_pool_split128_kernel<.., OCC = 2> (csrc/kernel_pool128.hip): chosen for documents of <= 96 tokens at E <= 128
    once the call holds >= 8,192 pairs (IDCM's ck-small sampler, sigir21_idcm.py:182-186, on 64-token passages).  9,000
    pairs in the three query layouts — shared query tile (pairs_per_query), one tile per pair, ragged groups through
    pair_query — with ragged lengths, an empty document and hole masks, every pair against the fp64 oracle; the three
    layouts must also agree bit for bit (same arithmetic, different addressing)."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    gen = torch.Generator().manual_seed(E * 100 + D)
    nq, C, Q = 9, 1000, 30
    B = nq * C
    q = torch.relu(torch.randn(nq, Q, E, generator=gen))
    d = torch.relu(torch.randn(B, D, E, generator=gen))
    for b in range(0, B, 7):
        d[b, b % D] = q[b // C, b % Q] * 1.3                      # planted exact matches
    q_len = torch.randint(1, Q + 1, (nq,), generator=gen)
    d_len = torch.randint(1, D + 1, (B,), generator=gen)
    d_len[5] = 0
    d_len[B - 1] = D
    qm = (torch.arange(Q)[None] < q_len[:, None])
    dm = (torch.arange(D)[None] < d_len[:, None])
    dm[11, 2] = False                                             # a hole (not a prefix mask)
    alpha = torch.rand(11, generator=gen) + 0.5
    w = torch.rand(11, generator=gen) - 0.5
    mu, sg = torch.tensor(MU), torch.tensor(SIGMA)
    t = lambda x: x.to(dev)
    with torch.no_grad():
        shared = ops.kernel_pool(t(q), t(d), t(qm), t(dm), t(mu), t(sg), t(alpha), t(w), pairs_per_query=C, clamp_min=1e-4)
        qrep, qmrep = q.repeat_interleave(C, 0), qm.repeat_interleave(C, 0)
        paired = ops.kernel_pool(t(qrep), t(d), t(qmrep), t(dm), t(mu), t(sg), t(alpha), t(w), pairs_per_query=1, clamp_min=1e-4)
        pq = (torch.arange(B) // C).to(torch.int32)
        ragged = ops.kernel_pool(t(q), t(d), t(qm), t(dm), t(mu), t(sg), t(alpha), t(w), clamp_min=1e-4, pair_query=t(pq))
    assert torch.equal(shared, paired) and torch.equal(shared, ragged)
    ref = O.idcm_sampler_scores(qrep.numpy(), d.numpy(), qmrep.numpy(), dm.numpy(), MU, SIGMA, alpha.numpy(), w.numpy(), 0.0,
                                dtype=np.float64)
    np.testing.assert_allclose(shared.cpu().numpy(), ref, atol=util.TOL_FP32, rtol=1e-5)
