"""GPU parity tests for the MaxSim path: HIP kernels (through the C ABI / ctypes) vs the oracle and
the committed golden vectors of the real reference.  Tolerances from BASELINE.json: 1e-3 (fp32),
1e-2 (bf16 / fp16), identical rank order (tie policy: SURVEY.md §7)."""
import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from tests import util

pytestmark = pytest.mark.gpu

DTYPES = [(torch.bfloat16, util.TOL_BF16), (torch.float16, util.TOL_BF16), (torch.float32, util.TOL_FP32)]


def _to(x, dtype, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dtype).to(dev)


def _rounded(x, dtype):
    """the values the kernel actually sees, as fp32 numpy (oracle input)"""
    return torch.from_numpy(np.ascontiguousarray(x)).to(dtype).float().numpy()


@pytest.mark.parametrize("fname", util.golden_files("colbert_"))
@pytest.mark.parametrize("dtype,tol", DTYPES)
def test_forward_matches_reference_golden(fname, dtype, tol):
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = util.load(fname)
    q, d = _to(g["q"], dtype, dev), _to(g["d"], dtype, dev)
    ref = g["forward"]
    if dtype == torch.float16:   # inputs are bf16-exact; fp16 may round tiny values: re-derive with the oracle
        ref = O.maxsim_paired(_rounded(g["q"], dtype), _rounded(g["d"], dtype), g["q_mask"], g["d_mask"])
    for mk in (torch.int64, torch.uint8, torch.bool, torch.float32, torch.int32):
        qm = torch.from_numpy(g["q_mask"]).to(mk).to(dev)
        dm = torch.from_numpy(g["d_mask"]).to(mk).to(dev)
        out = ops.maxsim(q, d, qm, dm, pairs_per_query=1).cpu().numpy()
        np.testing.assert_allclose(out, ref, atol=tol, rtol=1e-4, err_msg=f"mask dtype {mk}")
    # unmasked aggregation (colbert.py:100-112)
    ref_agg = g["forward_aggregation"]
    if dtype == torch.float16:
        ref_agg = O.maxsim_unmasked(_rounded(g["q"], dtype), _rounded(g["d"], dtype))
    out = ops.maxsim(q, d, None, None).cpu().numpy()
    np.testing.assert_allclose(out, ref_agg, atol=tol, rtol=1e-4)


@pytest.mark.parametrize("fname", util.golden_files("colbert_"))
@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, util.TOL_BF16), (torch.float32, util.TOL_FP32)])
def test_inbatch_matches_reference_golden(fname, dtype, tol):
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = util.load(fname)
    q, d = _to(g["q"], dtype, dev), _to(g["d"], dtype, dev)
    qm = torch.from_numpy(g["q_mask"]).to(torch.int64).to(dev)
    dm = torch.from_numpy(g["d_mask"]).to(torch.int64).to(dev)
    bug = ops.maxsim_inbatch(q, qm, d, dm, bug_compatible=True).cpu().numpy()
    np.testing.assert_allclose(bug, g["forward_inbatch_aggregation"], atol=tol, rtol=1e-4)
    ok = ops.maxsim_inbatch(q, qm, d, dm, bug_compatible=False).cpu().numpy()
    ref_ok = O.maxsim_inbatch(g["q"], g["q_mask"], g["d"], g["d_mask"], bug_compatible=False)
    np.testing.assert_allclose(ok, ref_ok, atol=tol, rtol=1e-4)
    # Bq != Bd: the reference raises; bug-compatible mode refuses, correct mode works
    with pytest.raises(Exception):
        ops.maxsim_inbatch(q[:1], qm[:1], d[:3], dm[:3], bug_compatible=True)
    ok13 = ops.maxsim_inbatch(q[:1], qm[:1], d[:3], dm[:3], bug_compatible=False).cpu().numpy()
    np.testing.assert_allclose(ok13, ref_ok[:1, :3], atol=tol, rtol=1e-4)


@pytest.mark.parametrize("dtype,tol", DTYPES)
@pytest.mark.parametrize("Q,D,E,ppq,nq", [(32, 180, 128, 37, 5), (17, 64, 128, 8, 3), (32, 33, 128, 1, 40),
                                           (20, 200, 64, 10, 2), (40, 96, 256, 4, 3), (1, 1, 16, 3, 2),
                                           (32, 180, 768, 6, 3), (30, 200, 768, 1, 5), (20, 70, 256, 3, 3),
                                           (32, 65, 384, 4, 2), (7, 31, 512, 2, 3)])
def test_shared_query_layout_random(dtype, tol, Q, D, E, ppq, nq):
    """1 query x C candidates layout (pairs_per_query > 1), ragged lengths incl. empty docs,
    last query with fewer candidates."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(Q * 1000 + D)
    B = nq * ppq - (ppq // 2 if nq > 1 else 0)
    q = (torch.randn(nq, Q, E, generator=g) * 0.25).to(dtype)
    d = (torch.randn(B, D, E, generator=g) * 0.25).to(dtype)
    q_len = torch.randint(1, Q + 1, (nq,), generator=g).to(torch.int32)
    d_len = torch.randint(0, D + 1, (B,), generator=g).to(torch.int32)
    d_len[0] = D
    d_len[-1] = 0
    out = ops.maxsim(q.to(dev), d.to(dev), q_len.to(dev), d_len.to(dev), pairs_per_query=ppq).cpu().numpy()
    qi = np.arange(B) // ppq
    qm = (np.arange(Q)[None, :] < q_len.numpy()[:, None])
    dm = (np.arange(D)[None, :] < d_len.numpy()[:, None])
    ref = O.maxsim_paired(q.float().numpy()[qi], d.float().numpy(), qm[qi], dm)
    np.testing.assert_allclose(out, ref, atol=tol, rtol=1e-4)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, util.TOL_BF16), (torch.float32, util.TOL_FP32)])
def test_padding_content_is_ignored_and_holes_are_sentinel(dtype, tol):
    """Garbage (inf/nan) in padded document rows must not leak; a hole inside the document counts
    as -1000 exactly like the reference's masked assignment (colbert.py:69).  bf16: the roofline kernel;
    fp32: the split-bf16 streaming kernel (kernel_pool128.hip, MaxSim epilogue)."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(5)
    B, Q, D, E = 6, 32, 180, 128
    q = torch.nn.functional.normalize(torch.randn(1, Q, E, generator=g), dim=-1).to(dtype)
    d = torch.nn.functional.normalize(torch.randn(B, D, E, generator=g), dim=-1).to(dtype)
    dm = torch.ones(B, D, dtype=torch.int64)
    dm[0, 100:] = 0
    dm[1, 31:] = 0
    dm[2, 32:] = 0
    dm[3, ::2] = 0          # every other token masked
    dm[4, :] = 0
    d2 = d.clone()
    d2[0, 100:] = float("inf")
    d2[1, 31:] = float("nan")
    d2[2, 32:] = -float("inf")
    out = ops.maxsim(q.to(dev), d2.to(dev), None, dm.to(dev), pairs_per_query=B).cpu().numpy()
    ref = O.maxsim_paired(np.repeat(q.float().numpy(), B, 0), d.float().numpy(), np.ones((B, Q)), dm.numpy())
    assert np.isfinite(out).all()
    np.testing.assert_allclose(out, ref, atol=tol)
    assert out[4] == -1000.0 * Q


def test_config2_full_size_properties_and_rank_order():
    """BASELINE.json config 2 at full size (64 queries x 1000 candidates, Q32/D180/E128 bf16):
    determinism, permutation equivariance over candidates, exact power-of-two linearity, a sampled
    comparison with the oracle, and identical top-k rank order under the documented tie policy."""
    from matchmaker_amd import ops, synth
    dev = util.require_gpu()
    nq, C = 64, 1000
    q, d, q_len, d_len = synth.colbert_batch(nq, C, dtype=torch.bfloat16, device=dev, lengths="msmarco")
    out = ops.maxsim(q, d, q_len, d_len, pairs_per_query=C)
    out2 = ops.maxsim(q, d, q_len, d_len, pairs_per_query=C)
    assert torch.equal(out, out2), "non-deterministic"
    # permutation of candidates inside each query permutes the scores
    perm = torch.stack([torch.randperm(C, device=dev) + i * C for i in range(nq)]).reshape(-1)
    outp = ops.maxsim(q, d[perm].contiguous(), q_len, d_len[perm].contiguous(), pairs_per_query=C)
    assert torch.equal(outp, out[perm])
    # linearity: 2*q -> 2*score exactly where no -1000 sentinel takes part (d_len >= 8 > 0, but padded docs
    # contribute the sentinel only if it wins the max, which it cannot against cosines in [-1, 1])
    outs = ops.maxsim((q.float() * 2).to(torch.bfloat16), d, q_len, d_len, pairs_per_query=C)
    assert torch.equal(outs, out * 2)
    # sampled oracle comparison + per-query rank order on 4 whole queries
    sel = [0, 17, 40, 63]
    qn = q.float().cpu().numpy()
    for i in sel:
        dn = d[i * C:(i + 1) * C].float().cpu().numpy()
        dm = synth.len_to_mask(d_len[i * C:(i + 1) * C], 180).cpu().numpy()
        qm = np.repeat(synth.len_to_mask(q_len[i:i + 1], 32).cpu().numpy(), C, 0)
        ref32 = O.maxsim_paired(np.repeat(qn[i:i + 1], C, 0), dn, qm, dm)
        ref64 = O.maxsim_paired(np.repeat(qn[i:i + 1], C, 0), dn, qm, dm, dtype=np.float64)
        got = out[i * C:(i + 1) * C].cpu().numpy()
        np.testing.assert_allclose(got, ref32, atol=util.TOL_BF16)
        # rank order under the measured-noise tie policy (tests/util.rank_parity; every query: test_zz_rank_order_gpu.py)
        util.rank_parity(got, ref32, ref64, (1, 10, 100, 1000), label=f"query {i}")


def test_errors_are_loud():
    from matchmaker_amd import ops, NativeError
    dev = util.require_gpu()
    q = torch.zeros(1, 4, 16, dtype=torch.bfloat16, device=dev)
    d = torch.zeros(2, 5, 16, dtype=torch.bfloat16, device=dev)
    with pytest.raises(NativeError):
        ops.maxsim(q, d.half(), pairs_per_query=2)                   # dtype mismatch
    # the C ABI itself refuses rows that are not 16-byte multiples (ops.maxsim pads them before the call)
    from matchmaker_amd import _lib
    out = torch.empty(2, dtype=torch.float32, device=dev)
    rc = _lib.lib().mm_maxsim_fwd(q.data_ptr(), d.data_ptr(), None, _lib.MASK_NONE, None, _lib.MASK_NONE, out.data_ptr(), 2, 2,
                                  4, 5, 12, _lib.MM_BF16, 0, None, 0, torch.cuda.current_stream(dev).cuda_stream)
    assert rc != 0 and "16-byte" in _lib.lib().mm_last_error().decode()
    with pytest.raises(NativeError):
        ops.maxsim(q.cpu(), d.cpu(), pairs_per_query=2)              # no CPU fallback
    with pytest.raises(NativeError):
        ops.maxsim(torch.zeros(3, 4, 16, device=dev), torch.zeros(2, 5, 16, device=dev), pairs_per_query=2)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, util.TOL_BF16), (torch.float16, util.TOL_BF16)])
@pytest.mark.parametrize("Q,D,E,ppq", [(38, 180, 128, 7), (64, 200, 128, 1), (33, 47, 256, 3), (40, 100, 512, 2),
                                       (38, 90, 768, 2), (40, 200, 768, 1), (64, 64, 768, 3), (65, 40, 128, 2)])
def test_queries_longer_than_one_tile(dtype, tol, Q, D, E, ppq):
    """Q = 30 + 8 [MASK] tokens (ColBERT query augmentation, independent_reranking_loader.py:106-112; the published
    checkpoint's config has query_augment_mask_number: 8 at colbert_compression_dim: 768) needs two query tiles: the
    streaming kernel with NQT = 2 for every streamed width up to Q = 64; the generic kernel beyond."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(Q * 17 + E)
    nq = 3
    B = nq * ppq
    q = (torch.randn(nq, Q, E, generator=g) / E ** 0.5).to(dtype)
    d = (torch.randn(B, D, E, generator=g) / E ** 0.5).to(dtype)
    q_len = torch.randint(1, Q + 1, (nq,), generator=g)
    q_len[0] = Q
    d_len = torch.randint(0, D + 1, (B,), generator=g)
    d_len[0] = D
    qm = (torch.arange(Q)[None] < q_len[:, None]).long()
    qm[1, 0] = 0                                      # hole in the first query word
    if Q > 33:
        qm[0, 33] = 0                                 # hole in the second query word
    dm = (torch.arange(D)[None] < d_len[:, None]).long()
    out = ops.maxsim(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), pairs_per_query=ppq).cpu().numpy()
    qi = np.arange(B) // ppq
    ref = O.maxsim_paired(q.float().numpy()[qi], d.float().numpy(), qm.numpy()[qi], dm.numpy())
    np.testing.assert_allclose(out, ref, atol=tol, rtol=1e-4)
    out2 = ops.maxsim(q.to(dev), d.to(dev), None, None, pairs_per_query=ppq).cpu().numpy()
    np.testing.assert_allclose(out2, O.maxsim_unmasked(q.float().numpy()[qi], d.float().numpy()), atol=tol, rtol=1e-4)


def test_widths_that_are_not_16_byte_rows_are_zero_padded():
    """colbert_compression_dim = 100 (fp16 rows of 200 B) and 50-d fp32 vectors: the host pads the width with zero
    columns (no effect on dot products), instead of failing where the reference runs; all-pairs with an empty side
    returns an empty matrix."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(100)
    for dtype, E, tol in ((torch.float16, 100, util.TOL_BF16), (torch.float32, 50, util.TOL_FP32), (torch.bfloat16, 36, util.TOL_BF16)):
        q = (torch.randn(2, 9, E, generator=g) / E ** 0.5).to(dtype)
        d = (torch.randn(6, 41, E, generator=g) / E ** 0.5).to(dtype)
        dl = torch.tensor([41, 3, 0, 17, 40, 1])
        out = ops.maxsim(q.to(dev), d.to(dev), None, dl.to(dev), pairs_per_query=3).cpu().numpy()
        qi = np.arange(6) // 3
        ref = O.maxsim_paired(q.float().numpy()[qi], d.float().numpy(), np.ones((6, 9)), (np.arange(41)[None] < dl.numpy()[:, None]))
        np.testing.assert_allclose(out, ref, atol=tol)
        inb = ops.maxsim_inbatch(q.to(dev), None, d.to(dev), None).cpu().numpy()
        np.testing.assert_allclose(inb, O.maxsim_inbatch(q.float().numpy(), np.ones((2, 9)), d.float().numpy(), np.ones((6, 41))), atol=tol)
        gq, gd = ops.maxsim_bwd(q[qi].to(dev), d.to(dev), None, dl.to(dev), torch.ones(6, device=dev))
        assert gq.shape == (6, 9, E) and gd.shape == (6, 41, E)
    e = ops.maxsim_inbatch(torch.zeros(0, 4, 16, device=dev), None, torch.zeros(3, 5, 16, device=dev), None)
    assert e.shape == (0, 3)
    with pytest.raises(ops.NativeError):          # a document range past the token matrix is refused, not read
        ops.maxsim_ragged(torch.zeros(1, 4, 128, dtype=torch.bfloat16, device=dev), torch.zeros(10, 128, dtype=torch.bfloat16, device=dev),
                          torch.tensor([0, 8], device=dev), torch.tensor([5, 12], device=dev), None, pairs_per_query=2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Q,D,E", [(32, 180, 128), (30, 200, 768), (8, 64, 256), (20, 256, 384), (32, 34, 512), (2, 2, 128),
                                   (31, 181, 128), (32, 300, 128)])
def test_pair_per_row_layout_with_tokenizer_masks(dtype, Q, D, E):
    """The reference's own batch layout (eval.py:108 -> colbert.py:68-75): one query tile PER PAIR, int64 HF masks — the
    pair kernel (query tile through the LDS ring, masks fetched and converted inside the kernel; odd Q / D or D > 256
    take the packed-mask variant).  Holes, all-padding documents, an all-padding query, many pairs per wavefront and
    fewer pairs than wavefronts, lengths / no masks through the same kernel."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(Q * 1000 + D + E)
    for B in (3, 2500):
        q = (torch.randn(B, Q, E, generator=g) / E ** 0.5).to(dtype)
        d = (torch.randn(B, D, E, generator=g) / E ** 0.5).to(dtype)
        ql = torch.randint(1, Q + 1, (B,), generator=g)
        dl = torch.randint(0, D + 1, (B,), generator=g)
        dl[0] = D
        dl[1] = 0                                           # an empty document: -1000 per real query token
        qm = (torch.arange(Q)[None] < ql[:, None]).long()
        dm = (torch.arange(D)[None] < dl[:, None]).long()
        qm[2] = 0                                           # an all-padding query scores 0
        if D > 4:
            dm[0, 1] = 0                                    # holes
            dm[0, D - 2] = 0
        if Q > 2:
            qm[0, 1] = 0
        dm[dm != 0] = torch.randint(1, 1 << 40, (int((dm != 0).sum()),), generator=g)     # any non-zero word is a real token
        out = ops.maxsim(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), pairs_per_query=1).cpu().numpy()
        ref = O.maxsim_paired(q.float().numpy(), d.float().numpy(), qm.numpy(), dm.numpy())
        np.testing.assert_allclose(out, ref, atol=util.TOL_BF16, rtol=1e-4)
        assert out[2] == 0.0 and out[1] == -1000.0 * int((qm[1] != 0).sum())
        # the same pairs with lengths and with no masks at all
        out_len = ops.maxsim(q.to(dev), d.to(dev), ql.to(dev), dl.to(dev), pairs_per_query=1).cpu().numpy()
        ref_len = O.maxsim_paired(q.float().numpy(), d.float().numpy(), (torch.arange(Q)[None] < ql[:, None]).numpy(),
                                  (torch.arange(D)[None] < dl[:, None]).numpy())
        np.testing.assert_allclose(out_len, ref_len, atol=util.TOL_BF16, rtol=1e-4)
        out_none = ops.maxsim(q.to(dev), d.to(dev), None, None, pairs_per_query=1).cpu().numpy()
        np.testing.assert_allclose(out_none, O.maxsim_unmasked(q.float().numpy(), d.float().numpy()), atol=util.TOL_BF16, rtol=1e-4)
        # identical to the shared-query kernel on the same numbers (pairs of one query stored once)
        if B == 3:
            shared = ops.maxsim(q[:1].to(dev), d.to(dev), qm[:1].to(dev), dm.to(dev), pairs_per_query=B)
            rep = ops.maxsim(q[:1].expand(B, -1, -1).contiguous().to(dev), d.to(dev), qm[:1].expand(B, -1).contiguous().to(dev),
                             dm.to(dev), pairs_per_query=1)
            assert torch.equal(shared, rep)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Q,D,E", [(38, 200, 768), (38, 180, 128), (64, 256, 256), (33, 47, 384), (40, 2, 512), (38, 193, 768)])
def test_long_queries_in_eval_sized_calls_read_the_tokenizer_masks_in_the_kernel(dtype, Q, D, E):
    """eval.py's call at the published checkpoint's shapes (batch_size_eval 512, Q = 30 + 8 [MASK], int64 HF masks): every
    wavefront of the two-tile streaming kernel scores one pair and reads the int64 masks itself (no packing launch); with
    at most 512 pairs TWO wavefronts share a pair (alternate blocks, maxima combined in LDS).
    The same pairs with bool masks take the packed-mask path of the same kernel: the scores must be bit-identical; a call
    too large for one pair per wavefront (int64 masks packed in their own launch) must be, too."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(Q * 131 + D + E)
    # 512 / 5: two wavefronts per pair; 1500: one pair per wavefront; 2600: more pairs than wavefront slots (packed masks)
    for B in (512, 5, 1500, 2600):
        q = (torch.randn(B, Q, E, generator=g) / E ** 0.5).to(dtype)
        d = (torch.randn(B, D, E, generator=g) / E ** 0.5).to(dtype)
        ql = torch.randint(1, Q + 1, (B,), generator=g)
        dl = torch.randint(0, D + 1, (B,), generator=g)
        dl[0] = D
        dl[1] = 0                                           # an empty document
        qm = (torch.arange(Q)[None] < ql[:, None]).long()
        dm = (torch.arange(D)[None] < dl[:, None]).long()
        qm[2] = 0                                           # an all-padding query
        qm[0, 1] = 0
        qm[3, Q - 1] = 1                                    # a real token after padding, in the second query word
        if D > 4:
            dm[0, 1] = 0                                    # holes
            dm[0, D - 2] = 0
            dm[4, D - 1] = 1                                # a real token after padding: the length is last + 1
        nz = dm != 0
        dm[nz] = torch.randint(1, 1 << 40, (int(nz.sum()),), generator=g) * (1 - 2 * torch.randint(0, 2, (int(nz.sum()),), generator=g))
        qd, dd = q.to(dev), d.to(dev)
        out = ops.maxsim(qd, dd, qm.to(dev), dm.to(dev), pairs_per_query=1)
        packed = ops.maxsim(qd, dd, (qm != 0).to(dev), (dm != 0).to(dev), pairs_per_query=1)
        assert torch.equal(out, packed)
        m = min(B, 600)
        ref = O.maxsim_paired(q[:m].float().numpy(), d[:m].float().numpy(), qm[:m].numpy(), dm[:m].numpy())
        np.testing.assert_allclose(out[:m].cpu().numpy(), ref, atol=util.TOL_BF16, rtol=1e-4)
        o = out.cpu().numpy()
        assert o[2] == 0.0 and o[1] == -1000.0 * int((qm[1] != 0).sum())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Bq,Bd,Q,D,E", [(70, 300, 32, 180, 128), (9, 1100, 38, 200, 128), (33, 33, 30, 64, 768), (5, 2000, 20, 47, 256),
                                          (130, 70, 32, 33, 128), (1030, 37, 17, 64, 128), (3, 5, 9, 41, 256), (32, 32, 32, 180, 128), (7, 20, 38, 200, 768),
                                          (2, 1, 1, 1, 128), (6, 9, 32, 32, 128), (4, 3, 5, 97, 256)])
def test_all_pairs_on_the_streaming_kernel(dtype, Bq, Bd, Q, D, E):
    """forward_inbatch_aggregation (colbert.py:154-162) at teacher-batch sizes: the streaming kernel in all-pairs mode
    (query tile resident, documents of one query consecutive) — holes, empty documents, padded queries, Bq != Bd,
    Q = 38 (two query tiles), both masking conventions (by document j; by row i as the reference's :158 does).
    E <= 256 with Q <= 32 and the documents' own masks runs TILED over queries (four / two queries per wavefront, one
    document slice per XCD): query counts that are no multiple of the tile, more query groups than group lanes (1030),
    fewer documents than slices (5), the dynamic teacher's own 32 x 32."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(Bq * 7 + Bd)
    q = (torch.randn(Bq, Q, E, generator=g) / E ** 0.5).to(dtype)
    d = (torch.randn(Bd, D, E, generator=g) / E ** 0.5).to(dtype)
    ql = torch.randint(1, Q + 1, (Bq,), generator=g)
    dl = torch.randint(0, D + 1, (Bd,), generator=g)
    dl[0] = D
    if Bd > 1:
        dl[1] = 0
    qm = (torch.arange(Q)[None] < ql[:, None]).long()
    dm = (torch.arange(D)[None] < dl[:, None]).long()
    if D > 1:
        dm[0, 1] = 0
    qm[0, 0] = 0
    out = ops.maxsim_inbatch(q.to(dev), qm.to(dev), d.to(dev), dm.to(dev), bug_compatible=False).cpu().numpy()
    ref = O.maxsim_inbatch(q.float().numpy(), qm.numpy(), d.float().numpy(), dm.numpy(), bug_compatible=False)
    np.testing.assert_allclose(out, ref, atol=util.TOL_BF16, rtol=1e-4)
    if Bq == Bd:
        bug = ops.maxsim_inbatch(q.to(dev), qm.to(dev), d.to(dev), dm.to(dev), bug_compatible=True).cpu().numpy()
        np.testing.assert_allclose(bug, O.maxsim_inbatch(q.float().numpy(), qm.numpy(), d.float().numpy(), dm.numpy(), bug_compatible=True),
                                   atol=util.TOL_BF16, rtol=1e-4)
    # every row of the matrix equals the paired operator on that query against all documents
    for ri in {min(3, Bq - 1), Bq - 1, Bq // 2}:
        row = ops.maxsim(q[ri:ri + 1].to(dev), d.to(dev), qm[ri:ri + 1].to(dev), dm.to(dev), pairs_per_query=Bd)
        assert torch.equal(row.cpu(), torch.from_numpy(out[ri]))


@pytest.mark.parametrize("dtype,Q,D,Bq,Bd,E", [(torch.bfloat16, 32, 180, 80, 200, 128), (torch.float16, 20, 50, 130, 77, 128),
                                                (torch.bfloat16, 32, 33, 64, 64, 128), (torch.bfloat16, 30, 70, 70, 90, 256)])
def test_all_pairs_on_the_workgroup_shared_ring(dtype, Q, D, Bq, Bd, E):
    """Teacher batches of >= 64 x 64 run on maxsim_allpairs_wg_kernel (four wavefronts share one document ring, a
    quarter of every slab each): every (query, document) score vs the oracle, vs the shared-query kernel bit for bit,
    ragged / empty documents, hole masks, query counts that are not a multiple of 16."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(Bq * 1000 + Bd)
    q = torch.nn.functional.normalize(torch.randn(Bq, Q, E, generator=g), dim=-1).to(dtype)
    d = torch.nn.functional.normalize(torch.randn(Bd, D, E, generator=g), dim=-1).to(dtype)
    q_len = torch.randint(1, Q + 1, (Bq,), generator=g)
    d_len = torch.randint(0, D + 1, (Bd,), generator=g)
    d_len[:3] = torch.tensor([0, D, 1])
    qm = (torch.arange(Q)[None] < q_len[:, None]).long()
    dm = (torch.arange(D)[None] < d_len[:, None]).long()
    dm[5, 0] = 0
    qm[2, 0] = 0
    out = ops.maxsim_inbatch(q.to(dev), qm.to(dev), d.to(dev), dm.to(dev), bug_compatible=False)
    assert out.shape == (Bq, Bd)
    ref = O.maxsim_inbatch(q.float().numpy(), qm.numpy(), d.float().numpy(), dm.numpy(), bug_compatible=False)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=util.TOL_BF16)
    # the same pairs through the shared-query kernel (one query x Bd candidates): the same bits
    for i in (0, 2, Bq - 1):
        row = ops.maxsim(q[i:i + 1].to(dev), d.to(dev), qm[i:i + 1].to(dev), dm.to(dev), pairs_per_query=Bd)
        assert torch.equal(row, out[i])
    assert torch.equal(out, ops.maxsim_inbatch(q.to(dev), qm.to(dev), d.to(dev), dm.to(dev), bug_compatible=False))
