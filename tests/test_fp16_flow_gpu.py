"""GPU: the reference's DEFAULT precision mode — fp16 autocast (config/train/defaults.yaml:21 `use_fp16: True`,
colbert.py:60) — reproduced on the device and checked against the reference's own statements executed by torch ON THE GPU.

Under autocast `bmm` returns an fp16 similarity matrix (fp32 accumulation, one rounding per element), the -1000 fill and
`max` are fp16 ops and `sum` is promoted to fp32 (colbert.py:68-75).  The native kernels keep fp32 MFMA accumulators and,
with MM_SIM_ROUND, round every per-token maximum to fp16 before the fp32 sum — by monotonicity of rounding the same
function.  The checker here is oracle/torch_port.maxsim_forward (the reference's statements, verbatim) under
`torch.autocast("cuda", torch.float16)`: what matchmaker itself executes on this GPU.  Two fp32-accumulating evaluations
of one dot product differ by ~1e-7 relative (accumulation order: the vendor GEMM's vs the MFMA loop's), so a similarity
within that distance of an fp16 rounding boundary can round either way: a small fraction of scores sits ONE fp16 ulp of
ONE token apart, in either implementation relative to the exact arithmetic (np_oracle with the product in fp64) — that
fraction is measured for both and must be of the same size.

The all-16-bit mode (fp16 tensors outside autocast: `sum` is an fp16 op too — the dynamic teacher's all-pairs call,
dynamic_teacher.py:245-246; MM_SIM_ROUND | MM_SUM_ROUND) is pinned on outputs of the REAL class (tests/golden/flow16_*.npz).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from oracle import torch_port as TP
from tests import util

pytestmark = pytest.mark.gpu


def _eager_autocast(q, d, qm, dm):
    """The reference's scoring statements under the reference's autocast, on the GPU."""
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        out = TP.maxsim_forward(q, d, qm, dm)
    assert out.dtype == torch.float32            # `sum` is on autocast's fp32 list: promoted (colbert.py:75)
    return out


def _flow_stats(name, got, eager, exact, C, token_ulp):
    """got / eager / exact: [nq * C] scores of the device, of torch eager under autocast, of the order-free oracle."""
    got, eager, exact = (np.asarray(x, dtype=np.float64) for x in (got, eager, exact))
    nq = got.shape[0] // C
    same_pos = 0
    eager_vs_exact_pos = 0
    for i in range(nq):
        s = slice(i * C, (i + 1) * C)
        og, oe, ox = (np.argsort(-x[s], kind="stable") for x in (got, eager, exact))
        same_pos += int((og == oe).sum())
        eager_vs_exact_pos += int((oe == ox).sum())
    rep = {"test": name, "queries": nq, "positions": int(got.shape[0]),
           "scores_bit_equal_to_eager_autocast": float((got == eager).mean()),
           "max_abs_diff_vs_eager_in_token_ulps": float(np.abs(got - eager).max() / token_ulp),
           "device_bit_equal_to_exact_flow": float((got == exact).mean()),
           "eager_bit_equal_to_exact_flow": float((eager == exact).mean()),
           "max_abs_diff_device_vs_exact_in_token_ulps": float(np.abs(got - exact).max() / token_ulp),
           "max_abs_diff_eager_vs_exact_in_token_ulps": float(np.abs(eager - exact).max() / token_ulp),
           "identical_rank_positions_vs_eager_autocast": same_pos / got.shape[0],
           "identical_rank_positions_eager_vs_exact_flow": eager_vs_exact_pos / got.shape[0],
           "token_ulp": token_ulp}
    print("[fp16 flow] " + json.dumps(rep))
    out = os.path.join(os.path.dirname(util.GOLDEN), "..", "gpurun_out")
    if os.environ.get("GRAFT_REPO_ROOT"):
        os.makedirs(out, exist_ok=True)
    if os.path.isdir(out):
        with open(os.path.join(out, f"rank_parity_{name}.json"), "w") as f:
            json.dump(rep, f)
    return rep


def _exact_flow(q, d, qm, dm, C, lowp=np.float16):
    """np_oracle, product in fp64 (order-free rounding decisions), one candidate list at a time."""
    nq = q.shape[0] // C
    out = []
    for i in range(nq):
        s = slice(i * C, (i + 1) * C)
        out.append(O.maxsim_paired(q[s], d[s], qm[s], dm[s], np.float64, sim_dtype=lowp))
    return np.concatenate(out)


def test_config2_lists_match_eager_autocast_scores_and_ranks():
    """16 queries x 1000 candidates, Q32 / D180 / E128, MSMARCO-shaped lengths: through ColBERT._score under autocast (the
    drop-in's path from fp32 encoder outputs), the shared-query kernel with sim_round, and torch.ops.mm_native.maxsim under
    autocast."""
    from matchmaker_amd import ops, synth, torch_ops  # noqa: F401
    from matchmaker_amd.colbert import ColBERT
    dev = util.require_gpu()
    nq, C, Q, D, E = 16, 1000, 32, 180, 128
    q, d, q_len, d_len = synth.colbert_batch(nq, C, Q, D, E, torch.float16, dev, seed=99, lengths="msmarco")
    qm = synth.len_to_mask(q_len, Q).to(dev).long()
    dm = synth.len_to_mask(d_len, D).to(dev).long()
    qr = q.repeat_interleave(C, 0)
    qmr = qm.repeat_interleave(C, 0)
    eager = _eager_autocast(qr, d, qmr, dm).cpu().numpy()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        via_dropin = ColBERT._score(qr.float(), d.float(), qmr, dm)          # fp32 vectors, as a compressor outside
        via_op = torch.ops.mm_native.maxsim(qr.float(), d.float(), qmr, dm, 1)   # autocast's lists would hand them over
    assert via_dropin.dtype == torch.float32 and via_op.dtype == torch.float32
    shared = ops.maxsim(q, d, q_len, d_len, pairs_per_query=C, sim_round=True)
    shared_masks = ops.maxsim(q, d, qm, dm, pairs_per_query=C, sim_round=True)
    # one function, four routes (pair kernel with in-kernel int64 masks, shared-query kernel with lengths / packed masks)
    for other in (via_op, shared, shared_masks):
        assert torch.equal(via_dropin, other)
    # without the flag the fp32 contract is unchanged — and is a different function
    plain = ops.maxsim(q, d, q_len, d_len, pairs_per_query=C)
    assert (plain != shared).float().mean() > 0.5
    exact = _exact_flow(qr.float().cpu().numpy(), d.float().cpu().numpy(), qmr.cpu().numpy(), dm.cpu().numpy(), C)
    rep = _flow_stats("colbert_forward_fp16_autocast", via_dropin.cpu().numpy(), eager, exact, C, 2.0 ** -11)
    # unit-norm vectors: every token maximum is < 1, one fp16 ulp there is 2^-11; a pair may hold a few boundary cases
    # measured on MI355X / ROCm 7.2: ALL 16,000 scores bit-equal to torch's autocast run and every rank position identical
    # (hipBLASLt's fp16 GEMM and the kernel's MFMA loop accumulate K in the same order); both differ from the exact
    # arithmetic in the same 0.16 % of the scores (profiles/r04_rank_parity/).  The bounds leave room for another GEMM kernel.
    assert rep["max_abs_diff_vs_eager_in_token_ulps"] <= 2.0
    assert rep["max_abs_diff_device_vs_exact_in_token_ulps"] <= 2.0
    assert rep["scores_bit_equal_to_eager_autocast"] >= 0.99
    # the device is as close to the exact arithmetic as torch's own GEMM is
    assert rep["device_bit_equal_to_exact_flow"] >= rep["eager_bit_equal_to_exact_flow"] - 0.005
    assert rep["identical_rank_positions_vs_eager_autocast"] >= 0.99


def test_published_checkpoint_shapes_match_eager_autocast():
    """Q38 (30 + 8 [MASK]) / D200 / E768, one eval.py batch of 512 pairs, unnormalised vectors (no L2 norm in the reference,
    colbert.py:62-63): the two-tile streaming kernel with in-kernel masks."""
    from matchmaker_amd.colbert import ColBERT
    dev = util.require_gpu()
    B, Q, D, E = 512, 38, 200, 768
    g = torch.Generator().manual_seed(4242)
    q = (torch.randn(B, Q, E, generator=g) * 0.08).half().to(dev)
    d = (torch.randn(B, D, E, generator=g) * 0.08).half().to(dev)
    q_len = torch.randint(3, 31, (B,), generator=g)
    d_len = torch.randint(8, D + 1, (B,), generator=g)
    qm = (torch.arange(Q)[None] < q_len[:, None]).long()
    qm[:, 30:] = 1                                                   # the [MASK] augmentation counts (loader :106-112)
    dm = (torch.arange(D)[None] < d_len[:, None]).long()
    qm, dm = qm.to(dev), dm.to(dev)
    eager = _eager_autocast(q, d, qm, dm).cpu().numpy()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        got = ColBERT._score(q, d, qm, dm).cpu().numpy()
    exact = O.maxsim_paired(q.float().cpu().numpy(), d.float().cpu().numpy(), qm.cpu().numpy(), dm.cpu().numpy(), np.float64,
                            sim_dtype=np.float16)
    tok = float(np.abs(exact).max() / Q)                             # scale of a token maximum
    ulp = 2.0 ** (np.floor(np.log2(max(tok, 1e-3))) - 10 + 1)        # one fp16 ulp at (the upper end of) that scale
    rep = _flow_stats("colbert_forward_fp16_autocast_q38_d200_e768", got, eager, exact, B, ulp)
    assert rep["max_abs_diff_vs_eager_in_token_ulps"] <= 2.0
    assert rep["scores_bit_equal_to_eager_autocast"] >= 0.99          # measured: 512 of 512
    assert rep["device_bit_equal_to_exact_flow"] >= rep["eager_bit_equal_to_exact_flow"] - 0.01


@pytest.mark.parametrize("fname", util.golden_files("flow16_"))
def test_all_16bit_flow_matches_the_real_class(fname):
    """fp16 / bf16 tensors OUTSIDE autocast through the drop-in's methods vs the REAL class's outputs on the same tensors
    (tests/golden/gen_golden.py gen_colbert_16bit_flow): sum rounded too, result in the vectors' dtype."""
    from matchmaker_amd.colbert import ColBERT
    dev = util.require_gpu()
    g = util.load_flow16(fname)
    dt = torch.bfloat16 if g["lowp"] == "bfloat16" else torch.float16
    q = torch.from_numpy(g["q"]).to(dev).to(dt)
    d = torch.from_numpy(g["d"]).to(dev).to(dt)
    qm = torch.from_numpy(g["q_mask"].astype(np.int64)).to(dev)
    dm = torch.from_numpy(g["d_mask"].astype(np.int64)).to(dev)
    m = ColBERT.__new__(ColBERT)                                   # the methods below use no module state
    torch.nn.Module.__init__(m)
    with torch.no_grad():
        outs = {"forward": ColBERT._score(q, d, qm, dm), "forward_aggregation": m.forward_aggregation(q, d),
                "forward_inbatch_aggregation": m.forward_inbatch_aggregation(q, qm, d, dm)}
    for key, got in outs.items():
        assert got.dtype == dt, key                                 # the reference's `sum` returns the tensors' dtype
        u = util.ulps16(got.float().cpu().numpy(), g[key], g["lowp"])
        assert u.max() <= 1.0, (fname, key, float(u.max()))
        assert (u == 0).mean() >= 0.9, (fname, key, float((u == 0).mean()))


@pytest.mark.parametrize("Bq,Bd", [(32, 32), (96, 80)])
def test_teacher_all_pairs_fp16_outside_autocast_matches_eager(Bq, Bd):
    """dynamic_teacher.py:245-246: forward_inbatch_aggregation on the fp16 vectors the teacher's forward returned, outside
    autocast — `mm`, `max`, `sum` all fp16.  Checked against the torch port run on the GPU in fp16 (32 x 32: the tiled
    kernel; 96 x 80: the workgroup-shared ring)."""
    from matchmaker_amd import ops, synth
    dev = util.require_gpu()
    Q, D, E = 32, 180, 128
    q, _, q_len, _ = synth.colbert_batch(Bq, 1, Q, D, E, torch.float16, dev, seed=5, lengths="msmarco")
    _, d, _, d_len = synth.colbert_batch(Bd, 1, Q, D, E, torch.float16, dev, seed=6, lengths="msmarco")
    qm = synth.len_to_mask(q_len, Q).to(dev).long()
    dm = synth.len_to_mask(d_len, D).to(dev).long()
    with torch.no_grad():
        eager = TP.maxsim_inbatch(q, qm, d, dm, bug_compatible=False)
    assert eager.dtype == torch.float16
    got = ops.maxsim_inbatch(q, qm, d, dm, bug_compatible=False, sim_round=True, sum_round=True)
    u = util.ulps16(got.cpu().numpy(), eager.float().cpu().numpy(), np.float16)
    assert u.max() <= 1.0 and (u == 0).mean() >= 0.95, (float(u.max()), float((u == 0).mean()))
    # autocast mode of the same call: fp32 sums of fp16 maxima
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        eager_ac = TP.maxsim_inbatch(q, qm, d, dm, bug_compatible=False)
    got_ac = ops.maxsim_inbatch(q, qm, d, dm, bug_compatible=False, sim_round=True)
    assert eager_ac.dtype == torch.float32
    diff = (got_ac - eager_ac).abs().cpu().numpy()
    assert diff.max() <= 3 * 2.0 ** -11 and (diff == 0).mean() >= 0.93, (float(diff.max()), float((diff == 0).mean()))


def test_ragged_store_aggregate_rounds_like_the_padded_call():
    """TokenStore.aggregate's kernel (mm_maxsim_ragged_fwd) with sim_round == the paired kernel on the padded copy
    (indexing_heads.py:49-56 runs forward_aggregation under autocast)."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(31)
    n, Q, E, Dmax = 300, 32, 128, 90
    lens = torch.randint(1, Dmax + 1, (n,), generator=g)
    end = torch.cumsum(lens, 0)
    begin = end - lens
    tokens = torch.nn.functional.normalize(torch.randn(int(end[-1]), E, generator=g), dim=-1).half().to(dev)
    q = torch.nn.functional.normalize(torch.randn(1, Q, E, generator=g), dim=-1).half().to(dev)
    rag = ops.maxsim_ragged(q, tokens, begin.to(dev), end.to(dev), None, pairs_per_query=n, sim_round=True)
    pad = torch.zeros(n, Dmax, E, dtype=torch.float16, device=dev)
    for i in range(n):
        pad[i, :int(lens[i])] = tokens[int(begin[i]):int(end[i])]
    # (padded rows are zero vectors: similarity 0, below every real maximum only if one is positive — mask them instead)
    ref = ops.maxsim(q, pad, None, lens.to(dev), pairs_per_query=n, sim_round=True)
    # a document with padding gets the -1000 fill in the padded call, never the maximum: identical scores
    assert torch.equal(rag, ref), float((rag - ref).abs().max())
    exact = O.maxsim_paired(np.repeat(q.float().cpu().numpy(), n, 0), pad.float().cpu().numpy(), np.ones((n, Q)),
                            (np.arange(Dmax)[None] < lens.numpy()[:, None]), np.float64, sim_dtype=np.float16)
    assert np.abs(rag.cpu().numpy() - exact).max() <= 3 * 2.0 ** -11


def test_flags_are_noops_for_fp32_and_unknown_flags_are_refused():
    from matchmaker_amd import _lib, ops, synth
    dev = util.require_gpu()
    q, d, q_len, d_len = synth.colbert_batch(2, 50, 32, 180, 128, torch.float32, dev, seed=3, lengths="msmarco")
    a = ops.maxsim(q, d, q_len, d_len, pairs_per_query=50)
    b = ops.maxsim(q, d, q_len, d_len, pairs_per_query=50, sim_round=True, sum_round=True)
    assert torch.equal(a, b)
    out = torch.empty(100, dtype=torch.float32, device=dev)
    rc = _lib.lib().mm_maxsim_fwd(q.data_ptr(), d.data_ptr(), None, _lib.MASK_NONE, None, _lib.MASK_NONE, out.data_ptr(), 100, 50,
                                  32, 180, 128, _lib.MM_F32, 8, None, 0, None)
    assert rc == -1 and b"flags" in _lib.lib().mm_last_error()


def test_an_fp32_token_store_under_use_fp16_is_scored_as_autocast_scores_it():
    """TokenStore.aggregate(use_fp16=True) on a `token_dtype: float32` store: the reference's searcher head runs
    forward_aggregation under autocast (indexing_heads.py:49-56), whose bmm casts BOTH operands to fp16 — fp16 values, fp16
    maxima, fp32 sum.  (Round 4 only set MM_SIM_ROUND, which is a no-op on fp32 rows: fp32 scores, silently.)  Checked against
    torch's own statements under autocast on the padded copy, and against the fp16 store of the same values."""
    from matchmaker_amd.token_store import TokenStore
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(77)
    n, Q, E, Dmax = 120, 32, 128, 70
    lens = torch.randint(1, Dmax + 1, (n,), generator=g)
    end = torch.cumsum(lens, 0).numpy()
    begin = end - lens.numpy()
    tok32 = torch.nn.functional.normalize(torch.randn(int(end[-1]), E, generator=g), dim=-1)
    q32 = torch.nn.functional.normalize(torch.randn(2, Q, E, generator=g), dim=-1)
    q32[1, 20:] = 0                                                      # an encoded query with padded (zeroed) positions
    ids = list(range(n))
    st32 = TokenStore(tok32.to(dev), ids, begin, end)
    st16 = TokenStore(tok32.half().to(dev), ids, begin, end)
    cands = [ids[:90], ids[30:]]
    got = st32.aggregate(q32.to(dev), cands, use_fp16=True)
    same = st16.aggregate(q32.to(dev), cands, use_fp16=True)
    assert got == same                                                   # the fp16 image of the store IS what gets scored
    plain = st32.aggregate(q32.to(dev), cands, use_fp16=False)
    assert any(abs(a[1] - b[1]) > 0 for ra, rb in zip(got, plain) for a, b in zip(ra, rb))      # and it is not the fp32 scoring
    # torch's own statements under autocast (colbert.py:100-112), candidate by candidate as dense_retrieval.py:400-409 loops
    for qi, cl in enumerate(cands):
        for j in (0, len(cl) // 2, len(cl) - 1):
            d = tok32[begin[cl[j]]:end[cl[j]]].unsqueeze(0).to(dev)
            with torch.autocast("cuda", dtype=torch.float16):
                sc = torch.bmm(q32[qi:qi + 1].to(dev), d.transpose(2, 1)).max(-1).values.sum(-1)
            assert got[qi][j][0] == cl[j]
            assert abs(got[qi][j][1] - float(sc)) <= 2 * 2.0 ** -11 * Q, (got[qi][j], float(sc))
