"""GPU: the ColBERT drop-in (tiny random BERT encoder in PyTorch + native MaxSim) scores exactly what
the oracle computes from the module's own token vectors; fp16 autocast path, teacher / return_vecs
conventions, aggregation entry points (colbert.py:54-162), training backward."""
import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from tests import util

pytestmark = pytest.mark.gpu


def _model(dev, dim=128):
    from transformers import BertConfig, BertModel
    from matchmaker_amd.colbert import ColBERT, ColBERTConfig
    torch.manual_seed(0)
    enc = BertModel(BertConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                               vocab_size=500, max_position_embeddings=256))
    return ColBERT(ColBERTConfig(bert_model="(injected)", compression_dim=dim), bert_model=enc).to(dev).eval()


def _batch(dev, B=9, Q=32, D=180):
    g = torch.Generator().manual_seed(1)
    ql = torch.randint(3, Q + 1, (B,), generator=g)
    dl = torch.randint(8, D + 1, (B,), generator=g)
    mk = lambda L, n: {"input_ids": torch.randint(1, 500, (B, n), generator=g).to(dev),
                       "attention_mask": (torch.arange(n)[None] < L[:, None]).long().to(dev)}
    return mk(ql, Q), mk(dl, D)


@pytest.mark.parametrize("use_fp16", [False, True])
def test_forward_equals_oracle_on_module_vectors(use_fp16):
    dev = util.require_gpu()
    m = _model(dev)
    query, doc = _batch(dev)
    with torch.no_grad():
        score = m.forward(query, doc, use_fp16=use_fp16)
        m.is_teacher_model = True
        s2, qv, dv = m.forward(query, doc, use_fp16=use_fp16)
        m.is_teacher_model = False
    assert score.dtype == torch.float32 and torch.equal(score, s2)
    # use_fp16: the reference's autocast arithmetic — fp16 similarities and maxima, fp32 sum (colbert.py:60-75)
    ref = O.maxsim_paired(qv.float().cpu().numpy(), dv.float().cpu().numpy(), query["attention_mask"].cpu().numpy(),
                          doc["attention_mask"].cpu().numpy(), sim_dtype=np.float16 if use_fp16 else None)
    np.testing.assert_allclose(score.cpu().numpy(), ref, atol=util.TOL_BF16 if use_fp16 else util.TOL_FP32)
    if use_fp16:
        assert qv.dtype == torch.float16
    with torch.no_grad():
        s, sec = m.forward(query, doc, use_fp16=use_fp16, output_secondary_output=True)
        assert sec == {} and torch.equal(s, score)
        m.return_vecs = True
        out = m.forward(query, doc, use_fp16=use_fp16)
        assert isinstance(out, tuple) and len(out) == 3
        m.return_vecs = False


def test_aggregation_entry_points():
    dev = util.require_gpu()
    m = _model(dev)
    query, doc = _batch(dev, B=6)
    with torch.no_grad():
        qv = m.forward_representation(query, "query_encode")      # zeroes padded rows (colbert.py:95-96)
        dv = m.forward_representation(doc, "doc_encode")
        agg = m.forward_aggregation(qv, dv)
        inb = m.forward_inbatch_aggregation(qv, query["attention_mask"], dv, doc["attention_mask"])
        m.inbatch_bug_compatible = False
        inb_ok = m.forward_inbatch_aggregation(qv, query["attention_mask"], dv, doc["attention_mask"])
        rect = m.forward_inbatch_aggregation(qv[:2], query["attention_mask"][:2], dv, doc["attention_mask"])
        m.inbatch_bug_compatible = True
        with pytest.raises(RuntimeError):
            m.forward_inbatch_aggregation(qv[:2], query["attention_mask"][:2], dv, doc["attention_mask"])
    qn, dn = qv.cpu().numpy(), dv.cpu().numpy()
    qm, dm = query["attention_mask"].cpu().numpy(), doc["attention_mask"].cpu().numpy()
    np.testing.assert_allclose(agg.cpu().numpy(), O.maxsim_unmasked(qn, dn), atol=util.TOL_FP32)
    np.testing.assert_allclose(inb.cpu().numpy(), O.maxsim_inbatch(qn, qm, dn, dm, True), atol=util.TOL_FP32)
    np.testing.assert_allclose(inb_ok.cpu().numpy(), O.maxsim_inbatch(qn, qm, dn, dm, False), atol=util.TOL_FP32)
    assert rect.shape == (2, 6)


def test_training_backward_flows_through_native_forward():
    dev = util.require_gpu()
    m = _model(dev)          # eval(): no dropout noise between the two passes; parameters still require grad
    query, doc = _batch(dev, B=4)
    score = m.forward(query, doc, use_fp16=False)
    score.sum().backward()
    g = m.compressor.weight.grad.clone()
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
    # same gradient as the eager formula of colbert.py:68-75
    m.zero_grad()
    qv, dv = m.forward_representation(query), m.forward_representation(doc)
    s = torch.bmm(qv, dv.transpose(2, 1))
    s = s.masked_fill(~doc["attention_mask"].bool().unsqueeze(1), -1000)
    ref = s.max(-1).values.masked_fill(~query["attention_mask"].bool(), 0).sum(-1)
    ref.sum().backward()
    torch.testing.assert_close(m.compressor.weight.grad, g, rtol=1e-3, atol=1e-4)


def test_full_forward_matches_the_real_colbert_class_end_to_end():
    """tests/golden/e2e_colbert_tinybert.npz holds the REAL ColBERT class's outputs (colbert.py:54-98: tiny
    random BERT + compressor + scoring) on token-id batches, with its state_dict.  The drop-in loads that
    state_dict unchanged (strict) and must reproduce forward() — the call eval.py:108 / train.py:347 makes —
    and the encode + forward_aggregation route of dense_retrieval.py."""
    from transformers import BertConfig, BertModel
    from matchmaker_amd.colbert import ColBERT, ColBERTConfig
    dev = util.require_gpu()
    g = util.load("e2e_colbert_tinybert.npz")
    enc = BertModel(BertConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                               vocab_size=500, max_position_embeddings=256, hidden_dropout_prob=0.0,
                               attention_probs_dropout_prob=0.0))
    m = ColBERT(ColBERTConfig(bert_model="(injected)", compression_dim=128), bert_model=enc)
    m.load_state_dict({k[len("param."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param.")}, strict=True)
    m = m.to(dev).eval()
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    query = {"input_ids": t("q_ids"), "attention_mask": t("q_mask")}
    doc = {"input_ids": t("d_ids"), "attention_mask": t("d_mask")}
    with torch.no_grad():
        score = m.forward(query, doc, use_fp16=False)
        agg = m.forward_aggregation(m.forward_representation(query, "query_encode"),
                                    m.forward_representation(doc, "doc_encode"))
        score16 = m.forward(query, doc, use_fp16=True)
    # the encoder runs on a different device / BLAS than the golden run: scores are sums of ~20 maxima of size ~30
    np.testing.assert_allclose(score.cpu().numpy(), g["forward"], atol=5e-3, rtol=2e-5)
    np.testing.assert_allclose(agg.cpu().numpy(), g["forward_aggregation"], atol=5e-3, rtol=2e-5)
    np.testing.assert_allclose(score16.cpu().numpy(), g["forward"], atol=0.5, rtol=2e-3)     # fp16 autocast encoder
    assert np.array_equal(np.argsort(-score.cpu().numpy(), kind="stable"), np.argsort(-g["forward"], kind="stable"))


@pytest.mark.parametrize("dt,Q,D,E,masks", [(torch.bfloat16, 32, 180, 128, True), (torch.float16, 32, 180, 128, True),
                                             (torch.float16, 30, 200, 768, True), (torch.bfloat16, 32, 180, 128, False)])
def test_batched_scoring_entry_is_bit_equal_to_per_batch_calls(dt, Q, D, E, masks):
    """mm_maxsim_fwd_batched (ops.maxsim_batched / ColBERT.score_batches): several eval.py-sized pair-per-row batches in one
    launch — same kernel body as the per-batch call, so every score must be the same bits; batches of different sizes, ragged
    int64 tokenizer masks with holes, more batches than one launch takes, and a shape it refuses (odd D) falling back."""
    from matchmaker_amd import ops
    from matchmaker_amd.colbert import ColBERT
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(Q * 1000 + E)
    batches = []
    for B in (512, 512, 37, 512, 1, 300) + (512,) * 13:          # 19 batches: two launches
        q = (torch.randn(B, Q, E, generator=g) / E ** 0.5).to(dt).to(dev)
        d = (torch.randn(B, D, E, generator=g) / E ** 0.5).to(dt).to(dev)
        if masks:
            qm = (torch.arange(Q)[None] < torch.randint(1, Q + 1, (B, 1), generator=g)).long()
            dm = (torch.arange(D)[None] < torch.randint(1, D + 1, (B, 1), generator=g)).long()
            dm[0, min(3, D - 1)] = 0
            batches.append((q, d, qm.to(dev), dm.to(dev)))
        else:
            batches.append((q, d, None, None))
    for sim_round, sum_round in ((True, False), (True, True), (False, False)):
        got = ops.maxsim_batched(batches, sim_round=sim_round, sum_round=sum_round)
        for b, s in zip(batches, got):
            assert torch.equal(s, ops.maxsim(b[0], b[1], b[2], b[3], 1, sim_round, sum_round))
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        a = ColBERT.score_batches(batches)
        w = [ColBERT._score(*b) for b in batches]
    assert all(torch.equal(x, y) and x.dtype == y.dtype for x, y in zip(a, w))
    # a shape the pair-per-row kernel refuses (odd D with int64 masks): score_batches falls back to per-batch calls
    if masks:
        odd = [(b[0], b[1][:, :D - 1].contiguous(), b[2], b[3][:, :D - 1].contiguous()) for b in batches[:3]]
        with pytest.raises(ops.NativeError):
            ops.maxsim_batched(odd)
        with torch.no_grad():
            a = ColBERT.score_batches(odd)
            w = [ColBERT._score(*b) for b in odd]
        assert all(torch.equal(x, y) for x, y in zip(a, w))


def test_grouped_evaluation_equals_the_batch_by_batch_loop():
    """rerank.evaluate_batches(score_group=4): encoder per batch, scoring block of four batches per launch, one .cpu() per group —
    the same unrolled results as eval.py's batch-by-batch loop (scores and arrival order), across a shape change and a short
    last batch."""
    from matchmaker_amd import rerank
    dev = util.require_gpu()
    m = _model(dev)
    g = torch.Generator().manual_seed(5)

    def batch(B, Q, D, tag):
        ql, dl = torch.randint(3, Q + 1, (B,), generator=g), torch.randint(8, D + 1, (B,), generator=g)
        mk = lambda L, n: {"input_ids": torch.randint(1, 500, (B, n), generator=g),
                           "attention_mask": (torch.arange(n)[None] < L[:, None]).long()}
        return {"query_tokens": mk(ql, Q), "doc_tokens": mk(dl, D), "query_id": [f"q{tag}_{i % 3}" for i in range(B)],
                "doc_id": [f"d{tag}_{i}" for i in range(B)]}
    batches = [batch(16, 32, 180, i) for i in range(6)] + [batch(16, 30, 170, 6), batch(16, 32, 180, 7), batch(5, 32, 180, 8)]
    eager = rerank.evaluate_batches(m, batches, use_fp16=True)
    grouped = rerank.evaluate_batches(m, batches, use_fp16=True, score_group=4)
    assert eager.keys() == grouped.keys()
    for k in eager:
        assert [d for d, _ in eager[k]] == [d for d, _ in grouped[k]]
        np.testing.assert_array_equal([s for _, s in eager[k]], [s for _, s in grouped[k]])


def test_graphed_evaluation_equals_eager_and_its_cache_is_bounded():
    """rerank.evaluate_batches(graph=True): one HIP-graph capture per batch shape, replayed — same scores as the eager loop,
    incl. a short last batch (another shape) and the secondary-output convention; the capture cache is an LRU (eval.py pads
    every batch to its longest sequence: real runs see many shapes)."""
    from matchmaker_amd import rerank
    dev = util.require_gpu()
    m = _model(dev)
    g = torch.Generator().manual_seed(4)

    def batch(B, Q, D, tag):
        ql, dl = torch.randint(3, Q + 1, (B,), generator=g), torch.randint(8, D + 1, (B,), generator=g)
        mk = lambda L, n: {"input_ids": torch.randint(1, 500, (B, n), generator=g),
                           "attention_mask": (torch.arange(n)[None] < L[:, None]).long()}
        return {"query_tokens": mk(ql, Q), "doc_tokens": mk(dl, D), "query_id": [f"q{tag}_{i % 3}" for i in range(B)],
                "doc_id": [f"d{tag}_{i}" for i in range(B)]}
    batches = [batch(16, 32, 180, 0), batch(16, 32, 180, 1), batch(16, 30, 170, 2), batch(16, 32, 180, 3), batch(5, 32, 180, 4)]
    for sec in (False, True):
        eager = rerank.evaluate_batches(m, batches, use_fp16=True, output_secondary_output=sec)
        graphed = rerank.evaluate_batches(m, batches, use_fp16=True, output_secondary_output=sec, graph=True)
        assert eager.keys() == graphed.keys()
        for k in eager:
            assert [d for d, _ in eager[k]] == [d for d, _ in graphed[k]]
            np.testing.assert_array_equal([s for _, s in eager[k]], [s for _, s in graphed[k]])
    # LRU bound: three shapes through a cache of two -> two captures alive, every score still right
    gf = rerank._GraphedForward(m, True, False, dev, max_graphs=2)
    with torch.no_grad():
        outs = [gf(b).clone() for b in batches]
        again = gf(batches[0]).clone()
    assert len(gf.entries) == 2 and gf.eager_calls == 0
    assert torch.equal(again, outs[0]) and outs[4].shape == (5,)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        want = m.forward({k: v.to(dev) for k, v in batches[2]["query_tokens"].items()},
                         {k: v.to(dev) for k, v in batches[2]["doc_tokens"].items()})
    assert torch.equal(outs[2], want)


@pytest.mark.parametrize("dtype,mask_kind", [(torch.float16, "int64"), (torch.bfloat16, "float"), (torch.float32, "bool"),
                                              (torch.float16, "lengths"), (torch.float16, "none")])
def test_cpp_autograd_node_equals_the_python_node_bit_for_bit(dtype, mask_kind):
    """matchmaker_amd/csrc_host/mm_autograd.cpp (the training step's node in C++: no Python in the backward) against
    colbert._MaxSimFn (the Python autograd.Function): the same two C-ABI calls, so scores and both gradients must be
    bit-equal — for every mask encoding the drop-in can be handed, through `ColBERT._score` as train.py drives it."""
    from matchmaker_amd import _fast
    from matchmaker_amd.colbert import ColBERT, _MaxSimFn
    dev = util.require_gpu()
    fast = _fast.module()
    if fast is None:      # (an OPTIONAL extension: the scoring path does not depend on it)
        pytest.skip("host extension not built / not usable under this torch (python -m matchmaker_amd.build)")
    g = torch.Generator().manual_seed(11)
    B, Q, D, E = 37, 32, 180, 128
    q0 = torch.nn.functional.normalize(torch.randn(B, Q, E, generator=g), dim=-1).to(dtype).to(dev)
    d0 = torch.nn.functional.normalize(torch.randn(B, D, E, generator=g), dim=-1).to(dtype).to(dev)
    ql, dl = torch.randint(1, Q + 1, (B,), generator=g), torch.randint(1, D + 1, (B,), generator=g)
    mk = {"int64": lambda L, n: (torch.arange(n)[None] < L[:, None]).long(), "float": lambda L, n: (torch.arange(n)[None] < L[:, None]).float(),
          "bool": lambda L, n: (torch.arange(n)[None] < L[:, None]), "lengths": lambda L, n: L.to(torch.int32), "none": lambda L, n: None}[mask_kind]
    qm, dm = mk(ql, Q), mk(dl, D)
    qm, dm = (None if qm is None else qm.to(dev)), (None if dm is None else dm.to(dev))
    go = torch.randn(B, generator=g).to(dev)
    sim_round = dtype != torch.float32
    res = []
    for use_cpp in (True, False):
        q, d = q0.clone().requires_grad_(True), d0.clone().requires_grad_(True)
        s = fast.maxsim_paired(q, d, qm, dm, 1 if sim_round else 0) if use_cpp else _MaxSimFn.apply(q, d, qm, dm, sim_round, False)
        assert s.requires_grad and s.dtype == torch.float32
        s.backward(go)
        res.append((s.detach(), q.grad, d.grad))
    for a, b, name in zip(res[0], res[1], ("scores", "grad_q", "grad_d")):
        assert a.dtype == b.dtype and torch.equal(a, b), name
    assert res[0][1].dtype == dtype and float(res[0][2].float().abs().sum()) > 0
    # ColBERT._score picks the C++ node by itself (grad enabled, 16-byte rows) and matches it
    if mask_kind in ("int64", "float", "bool"):
        q, d = q0.clone().requires_grad_(True), d0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=dtype == torch.float16):
            sc = ColBERT._score(q, d, qm, dm)
        names = {sc.grad_fn.name()} | {f[0].name() for f in sc.grad_fn.next_functions if f[0] is not None}
        assert any("MaxSimPaired" in n for n in names), names      # (16-bit vectors outside autocast: a cast node sits on top)
        sc.float().backward(go)
        if dtype == torch.float16:
            assert torch.equal(sc.detach(), res[0][0]) and torch.equal(q.grad, res[0][1]) and torch.equal(d.grad, res[0][2])
