"""GPU: "identical top-k rank order versus the reference" (BASELINE.json north_star) made decisive.

Every query of BASELINE configs[1] (64 queries x 1000 candidates, Q32/D180/E128) is ranked — by the shared-Q bf16
kernel, by the fp32 (three-term split-bf16) kernel, and through `ColBERT.forward` exactly as eval.py drives it (pair-per-row
batches, HF int64 masks, fp16 autocast and fp32) — and compared with the reference's ranking rule
(utils/core_metrics.py:502-511: stable descending sort) under the tie policy of tests/util.rank_parity:
the noise bound is measured on the arithmetic (2 x the larger of |fp32 oracle - fp64 oracle| and |device - fp64 oracle|,
a few 1e-6; the device error may be at most 4 x the fp32 oracle's), every candidate pair further apart than the
bound must be ordered as the fp64 scores order it, and the top-1 / 10 / 100 / 1000 cuts are compared as sets.
>= 99 % of all rank positions must be decided and >= 99.5 % must equal the plain stable sort of the fp32 oracle;
the per-query undecided counts are printed and written to gpurun_out/rank_parity_*.json."""
import numpy as np
import pytest
import torch
from torch import nn

from oracle import np_oracle as O
from tests import util

pytestmark = pytest.mark.gpu
KS = (1, 10, 100, 1000)


def _oracle_query(qn, dn, qm, dm, split=False):
    """One query's candidate list -> (fp32 oracle, fp64 oracle, noise or None)."""
    C = dn.shape[0]
    qr, qmr = np.repeat(qn[None], C, 0), np.repeat(qm[None], C, 0)
    r32 = O.maxsim_paired(qr, dn, qmr, dm)
    r64 = O.maxsim_paired(qr, dn, qmr, dm, dtype=np.float64)
    noise = None
    if split:
        emu = O.maxsim_paired_split_bf16(qr, dn, qmr, dm)
        noise = 8.0 * max(float(np.abs(r32 - r64).max()), float(np.abs(emu.astype(np.float64) - r64).max()))
    return r32, r64, noise


def _check_all_queries(name, out, q, d, q_len, d_len, C, split=False, fp32=False, min_decided=0.99):
    nq = q.shape[0]
    Q, D = q.shape[1], d.shape[1]
    rows = []
    qn = q.float().cpu().numpy()
    for i in range(nq):
        dn = d[i * C:(i + 1) * C].float().cpu().numpy()
        dm = util_mask(d_len[i * C:(i + 1) * C], D)
        qm = util_mask(q_len[i:i + 1], Q)[0]
        r32, r64, noise = _oracle_query(qn[i], dn, qm, dm, split)
        got = out[i * C:(i + 1) * C].cpu().numpy()
        np.testing.assert_allclose(got, r32, atol=util.TOL_FP32 if (split or fp32) else util.TOL_BF16)
        rows.append(util.rank_parity(got, r32, r64, KS, noise=noise, label=f"{name} query {i}"))
    frac = util.rank_report(name, rows)
    assert frac >= min_decided, f"{name}: only {frac:.4f} of the rank positions are decided"
    same = sum(r["identical_positions_vs_fp32_sort"] for r in rows) / sum(r["n"] for r in rows)
    assert same >= 0.995, f"{name}: only {same:.4f} of the positions equal the stable sort of the fp32 oracle"
    return rows


def util_mask(lens, L):
    from matchmaker_amd import synth
    return synth.len_to_mask(lens, L).cpu().numpy()


def test_rank_order_all_queries_bf16_shared_query():
    from matchmaker_amd import ops, synth
    dev = util.require_gpu()
    nq, C = 64, 1000
    q, d, q_len, d_len = synth.colbert_batch(nq, C, dtype=torch.bfloat16, device=dev, lengths="msmarco")
    out = ops.maxsim(q, d, q_len, d_len, pairs_per_query=C)
    _check_all_queries("bf16_shared_q", out, q, d, q_len, d_len, C)


def test_rank_order_all_queries_fp32_split_bf16():
    from matchmaker_amd import ops, synth
    dev = util.require_gpu()
    nq, C = 64, 1000
    q, d, q_len, d_len = synth.colbert_batch(nq, C, dtype=torch.float32, device=dev, lengths="msmarco", seed=77)
    out = ops.maxsim(q, d, q_len, d_len, pairs_per_query=C)
    # three-term split (x = hi + lo + c): fp32-class scores, so the SAME noise bound as a true fp32 evaluation
    _check_all_queries("fp32_split3_bf16", out, q, d, q_len, d_len, C, fp32=True)


class _TableEncoder(nn.Module):
    """Stands in for the BERT encoder: token id -> a pre-made vector (SURVEY.md §8c drives the real
    ColBERT.forward the same way).  `bert_model(**tokens)[0]` is all colbert.py:90 uses."""

    class _Cfg:
        def __init__(self, h):
            self.hidden_size = h

    def __init__(self, table):
        super().__init__()
        self.register_buffer("table", table, persistent=False)
        self.config = self._Cfg(table.shape[1])

    def forward(self, input_ids=None, attention_mask=None, **kw):
        return (self.table[input_ids],)


def _eval_batches(q_ids, d_ids, q_len, d_len, nq, C, Q, D, batch):
    """eval.py's input: pair-per-row batches (query replicated per candidate), int64 HF masks, ids."""
    B = nq * C
    ar_q, ar_d = torch.arange(Q), torch.arange(D)
    for s in range(0, B, batch):
        p = torch.arange(s, min(B, s + batch))
        qi = p // C
        yield {"query_tokens": {"input_ids": q_ids[qi], "attention_mask": (ar_q[None] < q_len[qi][:, None]).long()},
               "doc_tokens": {"input_ids": d_ids[p], "attention_mask": (ar_d[None] < d_len[p][:, None]).long()},
               "query_id": [f"q{int(i)}" for i in qi], "doc_id": [f"d{int(i)}" for i in p]}


@pytest.mark.parametrize("use_fp16", [True, False])
def test_rank_order_through_colbert_forward_as_eval_drives_it(use_fp16):
    """eval.py:82-108,161-203 around the drop-in: 16 queries x 1000 candidates in batches of 2,500 pairs (so batches
    straddle query boundaries), scores unrolled per query id and ranked with the reference's stable sort."""
    from matchmaker_amd import synth, rerank
    from matchmaker_amd.colbert import ColBERT, ColBERTConfig
    dev = util.require_gpu()
    nq, C, Q, D, E = 16, 1000, 32, 180, 128
    dt = torch.float16 if use_fp16 else torch.float32
    q, d, q_len, d_len = synth.colbert_batch(nq, C, Q, D, E, dt, "cpu", seed=99, lengths="msmarco")
    table = torch.cat([q.reshape(-1, E), d.reshape(-1, E)]).float()        # autocast rounds back to the same fp16 values
    q_ids = torch.arange(nq * Q).view(nq, Q)
    d_ids = nq * Q + torch.arange(nq * C * D).view(nq * C, D)
    m = ColBERT(ColBERTConfig(bert_model="(injected)", compression_dim=E), bert_model=_TableEncoder(table))
    m.compressor = nn.Identity()
    m = m.to(dev).eval()
    res = rerank.evaluate_batches(m, _eval_batches(q_ids, d_ids, q_len, d_len, nq, C, Q, D, 2500), use_fp16=use_fp16)
    assert len(res) == nq and all(len(v) == C for v in res.values())
    ranked = rerank.unrolled_to_ranked_result(res)
    rows = []
    for i in range(nq):
        got = np.array([s for _, s in res[f"q{i}"]], dtype=np.float32)
        assert [doc for doc, _ in res[f"q{i}"]] == [f"d{i * C + j}" for j in range(C)]
        dn = d[i * C:(i + 1) * C].float().numpy()
        r32, r64, noise = _oracle_query(q[i].float().numpy(), dn, util_mask(q_len[i:i + 1], Q)[0],
                                        util_mask(d_len[i * C:(i + 1) * C], D))
        rows.append(util.rank_parity(got, r32, r64, KS, noise=noise, label=f"forward fp16={use_fp16} query {i}"))
        # the list eval.py would hand to the metrics = stable descending sort of those scores
        assert ranked[f"q{i}"] == [f"d{i * C + j}" for j in np.argsort(-got, kind="stable")]
    frac = util.rank_report(f"colbert_forward_{'fp16' if use_fp16 else 'fp32'}", rows)
    assert frac >= 0.99
