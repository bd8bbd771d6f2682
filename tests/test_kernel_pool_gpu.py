"""GPU parity tests for TK kernel pooling (mm_kernel_pool_fwd) vs the oracle and the golden vectors
of the real ECAI20_TK.forward (ecai20_tk.py:105-124).  fp32 tolerance 1e-3 (BASELINE.json)."""
import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from tests import util

pytestmark = pytest.mark.gpu

MU = [1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9]
SIGMA = [0.1] * 11


def _t(x, dev, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dtype).to(dev)


def test_tk_matches_reference_golden():
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = util.load("tk_q20_d200_e300.npz")
    B = g["d"].shape[0]
    args = [_t(g[k], dev) for k in ("mu", "sigma", "alpha", "w")]
    # shared-query layout (1 query x B candidates) -> streaming kernel (E = 300)
    out, pk = ops.kernel_pool(_t(g["q"], dev), _t(g["d"], dev), _t(g["q_mask"][:1], dev), _t(g["d_mask"], dev),
                              *args, pairs_per_query=B, return_per_kernel=True)
    np.testing.assert_allclose(pk.cpu().numpy(), g["per_kernel"], atol=2e-3, rtol=1e-4)
    np.testing.assert_allclose(out.cpu().numpy(), g["score"], atol=util.TOL_FP32)
    # reference layout: query replicated per pair, float masks [B, Q]
    q_rep = _t(np.repeat(g["q"], B, 0), dev)
    out2 = ops.kernel_pool(q_rep, _t(g["d"], dev), _t(g["q_mask"], dev), _t(g["d_mask"], dev), *args)
    np.testing.assert_allclose(out2.cpu().numpy(), g["score"], atol=util.TOL_FP32)


@pytest.mark.parametrize("Q,D,E,ppq,nq", [(20, 200, 300, 10, 3), (30, 180, 300, 1, 12), (32, 64, 100, 7, 4),
                                           (11, 33, 200, 5, 2), (20, 50, 64, 4, 3), (40, 70, 128, 3, 2),
                                           (5, 1, 4, 2, 2)])
def test_kernel_pool_random(Q, D, E, ppq, nq):
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(Q * 131 + D)
    B = nq * ppq - (ppq // 2 if nq > 1 else 0)
    q = torch.randn(nq, Q, E, generator=g)
    d = torch.randn(B, D, E, generator=g)
    # plant near matches so the high-mu kernels see mass
    for b in range(B):
        d[b, b % D] = q[b // ppq, b % Q] * (1.0 + 0.02 * b)
    d[0, 0] = 0.0
    q_len = torch.randint(1, Q + 1, (nq,), generator=g)
    d_len = torch.randint(0, D + 1, (B,), generator=g)
    d_len[0] = D
    d_len[-1] = 0
    qm = (torch.arange(Q)[None] < q_len[:, None]).float()
    dm = (torch.arange(D)[None] < d_len[:, None]).float()
    dm[0, D // 2] = 0.0           # a hole (non-prefix mask)
    alpha = torch.rand(11, generator=g) + 0.5
    w = (torch.rand(11, generator=g) - 0.5) * 0.03
    mu, sigma = torch.tensor(MU), torch.tensor(SIGMA)
    out, pk = ops.kernel_pool(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), mu.to(dev), sigma.to(dev),
                              alpha.to(dev), w.to(dev), pairs_per_query=ppq, return_per_kernel=True)
    qi = np.arange(B) // ppq
    ref, ref_pk = O.tk_kernel_pool(q.numpy()[qi], d.numpy(), qm.numpy()[qi], dm.numpy(), MU, SIGMA, alpha.numpy(),
                                   w.numpy(), dtype=np.float64, return_per_kernel=True)
    np.testing.assert_allclose(pk.cpu().numpy(), ref_pk, atol=5e-3, rtol=1e-4)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=util.TOL_FP32)
    # lengths instead of dense masks give the same result when the masks are prefixes
    dm2 = (torch.arange(D)[None] < d_len[:, None]).float()
    a = ops.kernel_pool(q.to(dev), d.to(dev), q_len.to(dev), d_len.to(dev), mu.to(dev), sigma.to(dev),
                        alpha.to(dev), w.to(dev), pairs_per_query=ppq)
    b = ops.kernel_pool(q.to(dev), d.to(dev), qm.to(dev), dm2.to(dev), mu.to(dev), sigma.to(dev),
                        alpha.to(dev), w.to(dev), pairs_per_query=ppq)
    assert torch.equal(a, b)


def test_kernel_pool_config1_scale_properties():
    """Config-1 shapes at batch scale (1000 candidates): determinism, permutation equivariance, and
    invariance to positive rescaling of any token vector (cosine), and all 2,000 pairs against the fp64 oracle."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator(device=dev).manual_seed(9)
    nq, C, Q, D, E = 2, 1000, 20, 200, 300
    q = torch.randn(nq, Q, E, generator=g, device=dev)
    d = torch.randn(nq * C, D, E, generator=g, device=dev)
    d_len = torch.randint(10, D + 1, (nq * C,), generator=g, device=dev).to(torch.int32)
    q_len = torch.tensor([20, 7], dtype=torch.int32, device=dev)
    p = [torch.tensor(MU, device=dev), torch.tensor(SIGMA, device=dev), torch.ones(11, device=dev),
         torch.linspace(-0.014, 0.014, 11, device=dev)]
    out = ops.kernel_pool(q, d, q_len, d_len, *p, pairs_per_query=C)
    assert torch.equal(out, ops.kernel_pool(q, d, q_len, d_len, *p, pairs_per_query=C))
    perm = torch.cat([torch.randperm(C, device=dev) + i * C for i in range(nq)])
    outp = ops.kernel_pool(q, d[perm].contiguous(), q_len, d_len[perm].contiguous(), *p, pairs_per_query=C)
    assert torch.equal(outp, out[perm])
    outs = ops.kernel_pool(q * 4.0, d * 0.5, q_len, d_len, *p, pairs_per_query=C)   # powers of two: exact
    assert torch.equal(outs, out)
    # EVERY pair against the fp64 evaluation of ecai20_tk.py:105-124 (torch port on CPU tensors, one candidate list per call)
    from oracle import torch_port as TP
    f = lambda t: t.detach().cpu().double()
    for i in range(nq):
        sl = slice(i * C, (i + 1) * C)
        qm = (torch.arange(Q)[None] < int(q_len[i])).double().expand(C, -1).contiguous()
        dm = (torch.arange(D)[None] < d_len[sl].cpu()[:, None]).double()
        with torch.no_grad():
            ref = TP.tk_kernel_pool(f(q[i:i + 1]).expand(C, -1, -1).contiguous(), f(d[sl]), qm, dm, f(p[0]).view(1, 1, 1, -1),
                                    f(p[1]).view(1, 1, 1, -1), f(p[2]).view(1, 1, -1), f(p[3]).view(1, -1)).numpy()
        np.testing.assert_allclose(out[sl].cpu().numpy(), ref, atol=util.TOL_FP32)


@pytest.mark.parametrize("B,Q,D,E", [(4, 20, 200, 300), (3, 30, 47, 64), (2, 5, 3, 8), (3, 33, 70, 128),
                                     (2, 30, 2000, 64), (2, 20, 1000, 50),     # long documents: the [Q, D] tiles are swept in pieces
                                     (3, 24, 100, 384), (2, 32, 60, 512), (2, 8, 40, 448)])      # wide rows: eight 16-byte chunks per thread and block, two column tiles per wavefront
def test_kernel_pool_backward_matches_autograd_of_the_reference_ops(B, Q, D, E):
    """mm_kernel_pool_bwd vs autograd through the torch port of ecai20_tk.py:105-124 (CPU, float64 and
    float32), incl. the two trainable pooling parameters, and through the drop-in's autograd function."""
    from matchmaker_amd import ops
    from oracle import torch_port as TP
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(B * 100 + D)
    q = torch.randn(B, Q, E, generator=g)
    d = torch.randn(B, D, E, generator=g)
    for b in range(B):                                    # near matches so that the high-mu kernels carry gradient
        d[b, b % D] = q[b, b % Q] * 1.3 + 0.05 * torch.randn(E, generator=g)
    q_len = torch.randint(1, Q + 1, (B,), generator=g)
    d_len = torch.randint(1, D + 1, (B,), generator=g)
    d_len[0] = D
    qm = (torch.arange(Q)[None] < q_len[:, None]).float()
    dm = (torch.arange(D)[None] < d_len[:, None]).float()
    if D > 2:
        dm[0, 1] = 0.0                                    # a hole
    alpha = torch.rand(11, generator=g) + 0.5
    w = torch.randn(11, generator=g) * 0.3
    go = torch.randn(B, generator=g)
    mu, sigma = torch.tensor(MU), torch.tensor(SIGMA)

    def ref(dtype):
        leaves = [t.detach().to(dtype).clone().requires_grad_(True) for t in (q, d, alpha, w)]
        q_, d_, a_, w_ = leaves
        s = TP.tk_kernel_pool(q_, d_, qm.to(dtype), dm.to(dtype), mu.to(dtype).view(1, 1, 1, -1),
                              sigma.to(dtype).view(1, 1, 1, -1), a_.view(1, 1, -1), w_.view(1, -1))
        s.backward(go.to(dtype))
        return [t.grad for t in leaves]

    r64 = ref(torch.float64)
    gq, gd, ga, gw = ops.kernel_pool_bwd(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), mu.to(dev), sigma.to(dev),
                                         alpha.to(dev), w.to(dev), go.to(dev))
    for got, want, name in ((gq, r64[0], "grad_q"), (gd, r64[1], "grad_d"), (ga, r64[2], "grad_alpha"), (gw, r64[3], "grad_w")):
        want = want.numpy()
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(got.cpu().numpy().astype(np.float64), want, atol=2e-4 * scale, rtol=2e-3, err_msg=name)
    # through the drop-in's autograd function (what train.py's loss.backward() reaches)
    from matchmaker_amd.tk import _KernelPoolFn
    leaves = [t.to(dev).requires_grad_(True) for t in (q, d, alpha, w)]
    s = _KernelPoolFn.apply(leaves[0], leaves[1], qm.to(dev), dm.to(dev), mu.to(dev), sigma.to(dev), leaves[2], leaves[3])
    (s * go.to(dev)).sum().backward()
    # the function hands the forward's pooled kernel sums to the backward (mm_kernel_pool_ex_fwd2 -> _ex_bwd2): bit-equal to the
    # same two calls made by hand, and within rounding of the backward that pools the sums itself (a different summation order)
    _, pooled = ops.kernel_pool(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), mu.to(dev), sigma.to(dev), alpha.to(dev), w.to(dev),
                                return_pooled=True)
    gq2, gd2, ga2, gw2 = ops.kernel_pool_bwd(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), mu.to(dev), sigma.to(dev),
                                             alpha.to(dev), w.to(dev), go.to(dev), pooled=pooled)
    assert torch.equal(leaves[0].grad, gq2) and torch.equal(leaves[1].grad, gd2)
    np.testing.assert_allclose(leaves[3].grad.cpu().numpy(), gw2.cpu().numpy(), atol=1e-6)
    # (the two paths may sit on different forward kernels — E = 8 pools on the generic exact-f32 kernel, the pre-pass always on
    # split-bf16 cosines — and the sigma = 1e-3 exact-match kernel turns a 2^-18 cosine difference into 4e-3 of its activation:
    # both are held to the fp64 gradients below, and to each other at twice that bound)
    for got, want in ((gq2, gq), (gd2, gd), (ga2, ga), (gw2, gw)):
        scale = max(1.0, float(want.abs().max()))
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), atol=4e-4 * scale, rtol=4e-3)
    for got, want, name in ((gq2, r64[0], "grad_q"), (gd2, r64[1], "grad_d"), (ga2, r64[2], "grad_alpha"), (gw2, r64[3], "grad_w")):
        want = want.numpy()
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(got.cpu().numpy().astype(np.float64), want, atol=2e-4 * scale, rtol=2e-3, err_msg=name + " (pooled path)")


@pytest.mark.parametrize("Q,D,E,gated", [(20, 200, 300, False), (30, 180, 128, True), (9, 40, 64, False), (20, 33, 300, True)])
def test_small_batch_backward_shares_a_pair_between_workgroups_and_equals_the_large_batch_launch(Q, D, E, gated):
    """kernel_pool_bwd_split.hip: up to 128 pairs (the reference trains 64, defaults.yaml:114) the document blocks of a pair are
    shared by up to four workgroups and kp_bwd_combine_kernel finishes grad_q from their partial sums; from 129 pairs on one
    workgroup walks the whole pair.  Same pairs, both launches: grad_d and the parameter gradients bit-equal (each block is one
    workgroup's work either way), grad_q within summation-order rounding.  A caller with the older, smaller workspace
    (mm_kernel_pool_bwd_workspace_bytes) gets the one-workgroup launch: bit-equal everywhere."""
    from matchmaker_amd import _lib, ops
    dev = torch.device("cuda", 0)
    n, big = 50, 160
    g = torch.Generator().manual_seed(Q * 1000 + D)
    q, d = torch.randn(big, Q, E, generator=g).to(dev), torch.randn(big, D, E, generator=g).to(dev)
    ql = torch.randint(1, Q + 1, (big,), generator=g).to(torch.int32).to(dev)
    dl = torch.randint(0, D + 1, (big,), generator=g).to(torch.int32).to(dev)
    dl[0], dl[1], dl[2] = D, 0, min(D, 31)
    mu = torch.tensor([1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9], device=dev)
    sigma = torch.tensor([1e-3] + [0.1] * 10, device=dev)
    alpha, w = torch.rand(11, generator=g).to(dev) + 0.5, (torch.randn(11, generator=g) * 0.1).to(dev)
    gate = torch.relu(torch.randn(big, D, generator=g)).to(dev) if gated else None
    go = torch.randn(big, generator=g).to(dev)
    _, pooled = ops.kernel_pool(q, d, ql, dl, mu, sigma, alpha, w, d_gate=gate, return_pooled=True)
    ref = ops.kernel_pool_bwd(q, d, ql, dl, mu, sigma, alpha, w, go, d_gate=gate, pooled=pooled)
    sl = lambda t: None if t is None else t[:n].contiguous()
    got = ops.kernel_pool_bwd(sl(q), sl(d), sl(ql), sl(dl), mu, sigma, alpha, w, sl(go), d_gate=sl(gate), pooled=sl(pooled))
    assert torch.equal(got[1], ref[1][:n]), "grad_d"
    scale = float(ref[0][:n].abs().max())
    assert float((got[0] - ref[0][:n]).abs().max()) <= 2e-6 * max(scale, 1.0)
    assert not torch.isnan(got[0]).any()
    if (D + 31) // 32 > 1:
        assert not torch.equal(got[0], ref[0][:n]) or Q * E < 64, "the 50-pair launch was expected on the shared-pair path"
    # the smaller workspace of ABI <= 4 callers: one workgroup per pair, bit-equal to the large launch
    L = _lib.lib()
    qs, ds, gos, ps = sl(q), sl(d), sl(go), sl(pooled)
    _, qp, qk = ops._mask(sl(ql), n, Q, "q_mask")
    _, dp, dk = ops._mask(sl(dl), n, D, "d_mask")
    gs = sl(gate)
    gq, gd = torch.empty_like(qs), torch.empty_like(ds)
    ga, gw = torch.zeros(n, 11, device=dev), torch.zeros(n, 11, device=dev)
    gg = torch.zeros(n, D, device=dev) if gated else None
    wsb = L.mm_kernel_pool_bwd_workspace_bytes(n, Q, D, qk, dk)
    assert wsb < L.mm_kernel_pool_bwd_workspace_bytes2(n, Q, D, E, qk, dk)
    assert L.mm_kernel_pool_bwd_workspace_bytes2(big, Q, D, E, qk, dk) - L.mm_kernel_pool_bwd_workspace_bytes(big, Q, D, qk, dk) < 512
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    rc = L.mm_kernel_pool_ex_bwd2(qs.data_ptr(), ds.data_ptr(), qp, qk, dp, dk, gs.data_ptr() if gated else None, mu.data_ptr(),
                                  sigma.data_ptr(), alpha.data_ptr(), w.data_ptr(), 1e-10, ps.data_ptr(), gos.data_ptr(),
                                  gq.data_ptr(), gd.data_ptr(), gg.data_ptr() if gated else None, ga.data_ptr(), gw.data_ptr(),
                                  n, Q, D, E, 11, ws.data_ptr(), wsb, ops._stream(dev))
    assert rc == 0, L.mm_last_error()
    torch.cuda.synchronize()
    assert torch.equal(gq, ref[0][:n]) and torch.equal(gd, ref[1][:n])
    # (per-pair rows equal either way; the operator sums them in one [2, n, K] reduction)
    np.testing.assert_allclose(got[2].cpu().numpy(), ga.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg="grad_alpha")
    np.testing.assert_allclose(got[3].cpu().numpy(), gw.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg="grad_w")
    if gated:
        assert torch.equal(got[4], gg), "grad_gate"


@pytest.mark.parametrize("gated", [False, True])
def test_cpp_autograd_node_of_the_pooling_block_equals_the_python_node(monkeypatch, gated):
    """tk.kernel_pool_train: the C++ torch::autograd::Function (csrc_host/mm_autograd.cpp KernelPool) and the Python
    autograd.Function issue the same two C-ABI calls: scores and every gradient bit-equal (TK shape, float masks with a hole,
    TK-Sparse's gate, IDCM's floor)."""
    from matchmaker_amd import _fast
    from matchmaker_amd.tk import kernel_pool_train
    dev = util.require_gpu()
    if _fast.module() is None or not hasattr(_fast.module(), "kernel_pool"):
        pytest.skip("host extension not built (python -m matchmaker_amd.build)")
    g = torch.Generator().manual_seed(77)
    B, Q, D, E = 6, 20, 200, 300
    q = torch.randn(B, Q, E, generator=g)
    d = torch.randn(B, D, E, generator=g)
    qm = (torch.arange(Q)[None] < torch.randint(1, Q + 1, (B, 1), generator=g)).float()
    dm = (torch.arange(D)[None] < torch.randint(1, D + 1, (B, 1), generator=g)).float()
    dm[0, 1] = 0.0
    gate = torch.relu(torch.randn(B, D, generator=g)) if gated else None
    alpha = torch.rand(11, generator=g) + 0.5
    w = torch.randn(11, generator=g) * 0.3
    go = torch.randn(B, generator=g)
    mu, sigma = torch.tensor(MU), torch.tensor(SIGMA)

    def run(py):
        monkeypatch.setenv("MM_KP_PY_AUTOGRAD", "1" if py else "0")
        leaves = [t.to(dev).requires_grad_(True) for t in ((q, d, alpha, w) + ((gate,) if gated else ()))]
        s = kernel_pool_train(leaves[0], leaves[1], qm.to(dev), dm.to(dev), mu.to(dev), sigma.to(dev), leaves[2].view(1, 1, -1),
                              leaves[3].view(1, -1), leaves[4] if gated else None, 1e-4 if gated else 1e-10)
        (s * go.to(dev)).sum().backward()
        return [s.detach()] + [t.grad for t in leaves]

    a, b = run(False), run(True)
    for x, y in zip(a, b):
        assert x.shape == y.shape and torch.equal(x, y)


def test_knrm_dropin_matches_reference_golden_and_trains():
    """matchmaker_amd.knrm.KNRM (native pooling) vs the real KNRM.forward outputs (knrm.py:44-92), and its
    backward vs autograd through the oracle's torch port of the same ops."""
    from matchmaker_amd.knrm import KNRM
    from oracle import torch_port as TP
    dev = util.require_gpu()
    g = util.load("knrm_q14_d60_e300.npz")
    m = KNRM(11).to(dev)
    with torch.no_grad():
        m.dense.weight.copy_(_t(g["w"], dev).view(1, -1))
    q, d, qm, dm = (_t(g[k], dev) for k in ("q", "d", "q_mask", "d_mask"))
    with torch.no_grad():
        score, sec = m(q, d, qm, dm, output_secondary_output=True)
    np.testing.assert_allclose(score.cpu().numpy(), g["score"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(sec["per_kernel"].cpu().numpy(), g["per_kernel"], atol=2e-4, rtol=1e-4)
    # training: gradients w.r.t. the embeddings and dense.weight
    qg = q.clone().requires_grad_(True)
    dg = d.clone().requires_grad_(True)
    go = torch.linspace(-1, 1, q.shape[0], device=dev)
    (m(qg, dg, qm, dm) * go).sum().backward()
    qc = q.cpu().double().requires_grad_(True)
    dc = d.cpu().double().requires_grad_(True)
    wc = torch.from_numpy(g["w"]).double().requires_grad_(True)
    s = TP.tk_kernel_pool(qc, dc, qm.cpu().double(), dm.cpu().double(), torch.from_numpy(g["mu"]).double().view(1, 1, 1, -1),
                          torch.from_numpy(g["sigma"]).double().view(1, 1, 1, -1), torch.ones(1, 1, 11, dtype=torch.float64),
                          (wc * 0.01).view(1, -1))
    (s * go.cpu().double()).sum().backward()
    for got, want, name in ((qg.grad, qc.grad, "q"), (dg.grad, dc.grad, "d"), (m.dense.weight.grad.view(-1), wc.grad, "w")):
        want = want.numpy()
        scale = max(1e-3, float(np.abs(want).max()))
        np.testing.assert_allclose(got.cpu().numpy().astype(np.float64), want, atol=5e-4 * scale, rtol=5e-3, err_msg=name)


def test_conv_knrm_dropin_matches_reference_golden():
    """matchmaker_amd.conv_knrm.Conv_KNRM (torch convolutions + 9 native poolings, E = 128 -> generic fp32
    kernel) vs the real Conv_KNRM.forward outputs; gradients reach the convolutions and the dense layer."""
    from matchmaker_amd.conv_knrm import Conv_KNRM
    dev = util.require_gpu()
    g = util.load("conv_knrm_q12_d50_e64.npz")
    m = Conv_KNRM(64, 3, 11, 128)
    m.load_state_dict({k[len("param."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param.")})
    m = m.to(dev).eval()
    q, d, qm, dm = (_t(g[k], dev) for k in ("q", "d", "q_mask", "d_mask"))
    with torch.no_grad():
        s = m(q, d, qm, dm)
    np.testing.assert_allclose(s.cpu().numpy(), g["score"], atol=5e-5, rtol=1e-3)
    m.train()
    out = m(q, d, qm, dm)
    out.sum().backward()
    assert m.dense.weight.grad is not None and float(m.dense.weight.grad.abs().sum()) > 0
    assert all(c[1].weight.grad is not None and torch.isfinite(c[1].weight.grad).all() for c in m.convolutions)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["score"], atol=5e-5, rtol=1e-3)


@pytest.mark.parametrize("qlen", [1, 2, 3, 4, 5, 8, 9, 16, 17, 20])
@pytest.mark.parametrize("E,gated", [(300, False), (100, True)])
def test_short_queries_take_the_redistributed_epilogue(qlen, E, gated):
    """Effective query lengths on both sides of every lanes-per-token switch (2 / 4 / 8 / 16 / 32), a hole inside
    the query, prefix-masked and hole-masked documents, with and without the TK-Sparse gate."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    gen = torch.Generator().manual_seed(100 * qlen + E)
    nq, C, Q, D = 3, 5, 20, 130
    B = nq * C
    q = torch.randn(nq, Q, E, generator=gen)
    d = torch.randn(B, D, E, generator=gen)
    d[0, 3] = q[0, 0]
    qm = (torch.arange(Q)[None] < torch.tensor([qlen, max(1, qlen - 1), Q])[:, None]).float()
    if qlen > 2:
        qm[0, 1] = 0.0                                     # a hole: effective length stays qlen
    dm = (torch.arange(D)[None] < torch.randint(1, D + 1, (B,), generator=gen)[:, None]).float()
    dm[1, ::3] = 0.0
    gate = torch.relu(torch.randn(B, D, generator=gen)) if gated else None
    alpha, w = torch.rand(11, generator=gen) + 0.5, torch.randn(11, generator=gen) * 0.3
    t = lambda x: None if x is None else x.to(dev)
    s, pk = ops.kernel_pool(t(q), t(d), t(qm), t(dm), t(torch.tensor(MU)), t(torch.tensor(SIGMA)), t(alpha), t(w),
                            pairs_per_query=C, return_per_kernel=True, d_gate=t(gate))
    qi = np.arange(B) // C
    eff = dm.numpy() * (gate.numpy() if gated else 1.0)
    ref, ref_pk = O.tk_kernel_pool(q.numpy()[qi], d.numpy(), qm.numpy()[qi], eff, MU, SIGMA, alpha.numpy(), w.numpy(),
                                   dtype=np.float64, return_per_kernel=True)
    np.testing.assert_allclose(pk.cpu().numpy(), ref_pk, atol=5e-3, rtol=1e-4)
    np.testing.assert_allclose(s.cpu().numpy(), ref, atol=util.TOL_FP32, rtol=1e-5)


@pytest.mark.parametrize("E,D", [(128, 180), (64, 50), (300, 70), (36, 33)])
def test_multi_block_pooling_equals_the_sum_of_single_launches(E, D):
    """mm_kernel_pool_multi_fwd (Conv-KNRM: n_grams^2 match matrices in one launch, conv_knrm.py:130-137) vs the sum of
    per-combination launches and vs the oracle — on the 64n, 100n and generic kernel families."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(E + D)
    B, Q, n = 37, 12, 3
    qs = [torch.relu(torch.randn(B, Q, E, generator=g)) for _ in range(n)]
    ds = [torch.relu(torch.randn(B, D, E, generator=g)) for _ in range(n)]
    qm = (torch.arange(Q)[None] < torch.randint(1, Q + 1, (B,), generator=g)[:, None]).float()
    dm = (torch.arange(D)[None] < torch.randint(0, D + 1, (B,), generator=g)[:, None]).float()
    mu, sigma = torch.tensor(MU), torch.tensor(SIGMA)
    ones = torch.ones(11)
    w = torch.randn(n * n, 11, generator=g) * 0.01
    t = lambda x: x.to(dev)
    got = ops.kernel_pool_multi([t(x) for x in qs], [t(x) for x in ds], t(qm), t(dm), t(mu), t(sigma), t(ones), t(w))
    want = torch.zeros(B, device=dev)
    ref = np.zeros(B)
    for i in range(n):
        for j in range(n):
            want = want + ops.kernel_pool(t(qs[i]), t(ds[j]), t(qm), t(dm), t(mu), t(sigma), t(ones), t(w[i * n + j]))
            ref += O.tk_kernel_pool(qs[i].numpy(), ds[j].numpy(), qm.numpy(), dm.numpy(), MU, SIGMA, ones.numpy(),
                                    w[i * n + j].numpy(), dtype=np.float64)
    # same kernels, same (i, t) summation order; not bit-equal in general: a single launch of few pairs splits every pair's
    # blocks over two wavefronts (eval.py-sized calls), the nine-combination launch does not, and the order in which a
    # pair's block sums are added differs
    torch.testing.assert_close(got, want, rtol=2e-6, atol=2e-5)
    np.testing.assert_allclose(got.cpu().numpy(), ref, atol=util.TOL_FP32, rtol=1e-5)


@pytest.mark.parametrize("Q", [20, 32, 9])
def test_shared_query_candidate_lists_every_pair(Q):
    """"1 query x C candidates" lists at E = 300 (the shared-query layout): every pair vs the fp64 oracle, per-kernel
    outputs, dense masks with holes, empty documents, queries of every epilogue class (MFMA layout, 11 / 8 / 4 / 2 / 1
    rows per lane), a list length that is not a multiple of anything, determinism, permutation equivariance, and the
    same pairs in the pair-per-row layout."""
    from matchmaker_amd import ops
    from oracle import torch_port as TP
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(300 + Q)
    nq, C, D, E = 6, 333, 200, 300
    q = torch.randn(nq, Q, E, generator=g)
    d = torch.randn(nq * C, D, E, generator=g)
    q_len = torch.tensor([Q, min(Q, 17), min(Q, 12), min(Q, 7), 3, 1])
    d_len = torch.randint(0, D + 1, (nq * C,), generator=g)
    d_len[:5] = torch.tensor([0, 1, 31, 32, 200])
    for p_ in range(0, nq * C, 3):                                     # planted exact / near matches
        i = p_ // C
        if d_len[p_] > 0:
            d[p_, int(torch.randint(0, int(d_len[p_]), (1,), generator=g))] = q[i, int(torch.randint(0, int(q_len[i]), (1,), generator=g))] * 2.0
    qm = (torch.arange(Q)[None] < q_len[:, None]).float()
    dm = (torch.arange(D)[None] < d_len[:, None]).float()
    dm[7, 3] = 0.0
    dm[400, 0] = 0.0                                                   # holes: per-position validity bits
    qm[1, 0] = 0.0
    alpha = torch.rand(11, generator=g) + 0.5
    w = torch.randn(11, generator=g) * 0.2
    mu, sg = torch.tensor(MU), torch.tensor(SIGMA)
    t = lambda x: x.to(dev)
    out, pk = ops.kernel_pool(t(q), t(d), t(qm), t(dm), t(mu), t(sg), t(alpha), t(w), pairs_per_query=C, return_per_kernel=True)
    out2 = ops.kernel_pool(t(q), t(d), t(qm), t(dm), t(mu), t(sg), t(alpha), t(w), pairs_per_query=C)
    assert torch.equal(out, out2)
    f = lambda x: x.double()
    refs, refpk = [], []
    for i in range(nq):
        sl = slice(i * C, (i + 1) * C)
        qi = f(q[i:i + 1]).expand(C, -1, -1).contiguous()
        cos = TP.cosine_matrix(qi, f(d[sl])).unsqueeze(-1)
        act = torch.exp(-torch.pow(cos - f(mu).view(1, 1, 1, -1), 2) / (2 * torch.pow(f(sg).view(1, 1, 1, -1), 2))) * f(dm[sl]).unsqueeze(1).unsqueeze(-1)
        lg = torch.log(torch.clamp(act.sum(2) * f(alpha).view(1, 1, -1), min=1e-10)) * f(qm[i]).view(1, -1, 1)
        refpk.append(lg.sum(1))
        refs.append(lg.sum(1) @ f(w))
    ref, ref_pk = torch.cat(refs).numpy(), torch.cat(refpk).numpy()
    np.testing.assert_allclose(pk.cpu().numpy(), ref_pk, atol=5e-3, rtol=1e-4)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=util.TOL_FP32)
    # permutation inside every list: bit-equal (nothing may depend on which wavefront scores a pair)
    perm = torch.cat([torch.randperm(C, generator=g) + i * C for i in range(nq)])
    outp = ops.kernel_pool(t(q), t(d[perm].contiguous()), t(qm), t(dm[perm].contiguous()), t(mu), t(sg), t(alpha), t(w), pairs_per_query=C)
    assert torch.equal(outp.cpu(), out.cpu()[perm])
    # the same pairs in the pair-per-row layout (one-wavefront-per-SIMD kernel): same arithmetic class
    rep = ops.kernel_pool(t(q.repeat_interleave(C, 0).contiguous()), t(d), t(qm.repeat_interleave(C, 0).contiguous()), t(dm), t(mu), t(sg),
                          t(alpha), t(w), pairs_per_query=1)
    np.testing.assert_allclose(rep.cpu().numpy(), out.cpu().numpy(), rtol=2e-6, atol=2e-5)


def test_scores_of_a_pair_do_not_depend_on_the_batch_it_arrives_in_beyond_fp32_rounding():
    """Calls of <= 512 pairs split every pair's document blocks over TWO wavefronts (eval.py-sized batches) and add the
    partial document sums afterwards; larger calls sum them in one wavefront.  The same (query, document) pair can
    therefore differ in the last bits between a final partial evaluation batch and a full one — fp32 summation order,
    nothing else: bounded here (relative 4e-6 on the score) and rank-neutral wherever two scores differ by more than it."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(77)
    B, Q, D, E = 1536, 20, 200, 300
    q = torch.randn(B, Q, E, generator=g).to(dev)
    d = torch.randn(B, D, E, generator=g).to(dev)
    qm = (torch.arange(Q)[None] < torch.randint(3, Q + 1, (B,), generator=g)[:, None]).float().to(dev)
    dm = (torch.arange(D)[None] < torch.randint(40, D + 1, (B,), generator=g)[:, None]).float().to(dev)
    prm = [torch.tensor(MU, device=dev), torch.full((11,), 0.1, device=dev), torch.ones(11, device=dev),
           torch.linspace(-0.5, 0.5, 11, device=dev)]
    big = ops.kernel_pool(q, d, qm, dm, *prm)[:512]                       # one wavefront per pair
    small = ops.kernel_pool(q[:512], d[:512], qm[:512], dm[:512], *prm)   # two wavefronts per pair
    scale = float(big.abs().max())
    diff = float((big - small).abs().max())
    assert diff <= 4e-6 * scale, (diff, scale)
    a, b = big.cpu().numpy(), small.cpu().numpy()
    order_a, order_b = np.argsort(-a, kind="stable"), np.argsort(-b, kind="stable")
    gaps = np.abs(np.diff(a[order_a]))
    decided = np.concatenate([[True], gaps > 2 * diff]) & np.concatenate([gaps > 2 * diff, [True]])
    assert (order_a[decided] == order_b[decided]).all() and decided.mean() > 0.98


@pytest.mark.parametrize("E,ppq,nq,D", [(300, 37, 70, 200), (100, 1000, 3, 64), (200, 5, 900, 40)])
def test_candidate_lists_with_a_shared_query_equal_the_pair_per_row_call(E, ppq, nq, D):
    """The "1 query x C candidates" layout (query tile read once per list, int32 lengths) against the pair-per-row call of
    the same pairs (each pair its own query copy, float query masks) and against the oracle.  Lengths are chosen to hit every
    seam: empty documents (also several in a row, at list ends and wavefront ends), 1 / 31 / 32 / 33 rows, full length, many
    8-row documents, lists that straddle wavefront ranges.  (Written for round 4's row-sequence kernel, which passed it and
    was retired as slower: profiles/r04_experiments/tk_row_sequence_ab.txt.)"""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(E + ppq)
    Q = 20
    B = nq * ppq
    q = torch.randn(nq, Q, E, generator=g)
    d = torch.randn(B, D, E, generator=g)
    special = torch.tensor([0, 0, 0, 1, 31, 32, 33, D, D, 8, 8, 8, 8, 8, 8, 2, 0, D - 1, 7, 0][: max(4, min(20, D))]).clamp(max=D)
    d_len = torch.randint(0, D + 1, (B,), generator=g)
    pos = torch.randint(0, B, (min(B, 400),), generator=g)
    d_len[pos] = special[torch.arange(pos.numel()) % special.numel()]
    d_len[:3] = 0
    d_len[-2:] = 0
    d_len[ppq - 1] = 0                                                 # the last document of the first list
    q_len = torch.randint(1, Q + 1, (nq,), generator=g)
    q_len[0] = Q
    prm = [torch.tensor(MU), torch.tensor(SIGMA), torch.rand(11, generator=g) + 0.5, torch.randn(11, generator=g) * 0.3]
    dp = [t.to(dev) for t in prm]
    out, pk = ops.kernel_pool(q.to(dev), d.to(dev), q_len.to(dev), d_len.to(dev), *dp, pairs_per_query=ppq, return_per_kernel=True)
    qm = (torch.arange(Q)[None] < q_len[:, None]).float().repeat_interleave(ppq, 0)
    out_pp, pk_pp = ops.kernel_pool(q.repeat_interleave(ppq, 0).contiguous().to(dev), d.to(dev), qm.to(dev), d_len.to(dev), *dp,
                                    pairs_per_query=1, return_per_kernel=True)
    np.testing.assert_allclose(out.cpu().numpy(), out_pp.cpu().numpy(), atol=3e-5, rtol=2e-6)
    np.testing.assert_allclose(pk.cpu().numpy(), pk_pp.cpu().numpy(), atol=2e-3, rtol=2e-5)
    assert torch.equal(out, ops.kernel_pool(q.to(dev), d.to(dev), q_len.to(dev), d_len.to(dev), *dp, pairs_per_query=ppq))
    sel = torch.cat([torch.arange(0, min(B, 3 * ppq)), torch.arange(max(0, B - 50), B)]).unique()
    dm = (torch.arange(D)[None] < d_len[sel, None]).float()
    ref = O.tk_kernel_pool(q[sel // ppq].numpy(), d[sel].numpy(), qm[sel].numpy(), dm.numpy(), MU, SIGMA, prm[2].numpy(), prm[3].numpy(),
                           dtype=np.float64)
    np.testing.assert_allclose(out[sel.to(dev)].cpu().numpy(), ref, atol=util.TOL_FP32, rtol=1e-5)


def _multi_case(dev, seed=5):
    g = torch.Generator().manual_seed(seed)
    B, Q, D, E, n = 9000, 30, 70, 128, 3
    qs = [torch.relu(torch.randn(B, Q, E, generator=g)) for _ in range(n)]
    ds = [torch.relu(torch.randn(B, D, E, generator=g)) for _ in range(n)]
    qm = (torch.arange(Q)[None] < torch.randint(1, Q + 1, (B,), generator=g)[:, None]).float()
    dm = (torch.arange(D)[None] < torch.randint(0, D + 1, (B,), generator=g)[:, None]).float()
    if seed != 5:                 # holes: the masks travel as bit words instead of lengths
        dm[torch.arange(0, B, 7), torch.randint(0, D, (len(range(0, B, 7)),), generator=g)] = 0.0
        qm[torch.arange(0, B, 11), 0] = 0.0
    w = torch.randn(n * n, 11, generator=g) * 0.01
    return qs, ds, qm, dm, w


def _multi_run(path=None, seed=5):
    from matchmaker_amd import ops
    dev = util.require_gpu()
    qs, ds, qm, dm, w = _multi_case(dev, seed)
    t = lambda x: x.to(dev)
    got = ops.kernel_pool_multi([t(x) for x in qs], [t(x) for x in ds], t(qm), t(dm), t(torch.tensor(MU)), t(torch.tensor(SIGMA)),
                                t(torch.ones(11)), t(w))
    if path:
        np.save(path, got.cpu().numpy())
    return got


@pytest.mark.parametrize("seed", [5, 6])
def test_multi_launch_in_flat_xcd_order_with_two_wavefronts_per_simd(tmp_path, seed):
    """Conv-KNRM's 3 x 3 match matrices at a size where kp128_launch picks the block-once form (>= 2,048 pairs, E = 128,
    three query tensors: one wavefront per document tensor looping over the query tensors, kernel_pool_multi128_kernel).
    9,000 pairs vs the fp64 oracle.  The per-combination forms, each in a child process, are BIT-EQUAL to one another — the
    flat XCD-grouped order with two wavefronts per SIMD (round 5's default: MM_KP_MULTI_LOOP=0), the 2-D grid with one
    wavefront per SIMD of rounds 1-4 (+ MM_KP_MULTI_2D=1 MM_KP128_OCC=1), the wavefront-per-query-tensor workgroups with a rate
    barrier per block (+ MM_KP_MULTI_WG=1): in which order workgroups run and how many share a SIMD must not reach the
    arithmetic.  The block-once kernel is a different instruction stream (the compiler contracts its multiply-adds
    differently): its scores agree with theirs to fp32 rounding (5e-6 on scores of order 1), not bit for bit — measured with
    the direct RBF form as well as with the recurrence (tools/scratch/cmp_forms.py) — and are the same bits whether chosen by
    size or forced (MM_KP_MULTI_LOOP=1).  seed 5: prefix masks (lengths), seed 6: masks with holes (bit words)."""
    import os, subprocess, sys
    dev = util.require_gpu()
    got = _multi_run(seed=seed)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    forms = (("loop_form", {"MM_KP_MULTI_LOOP": "1"}),
             ("flat_form", {"MM_KP_MULTI_LOOP": "0"}),
             ("old_form", {"MM_KP_MULTI_LOOP": "0", "MM_KP_MULTI_2D": "1", "MM_KP128_OCC": "1"}),
             ("wg_form", {"MM_KP_MULTI_LOOP": "0", "MM_KP_MULTI_WG": "1"}))
    res = {}
    for name, env in forms:
        path = str(tmp_path / f"{name}.npy")
        r = subprocess.run([sys.executable, "-c", f"from tests.test_kernel_pool_gpu import _multi_run; _multi_run({path!r}, {seed})"], cwd=root,
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[name] = np.load(path)
    assert got.cpu().numpy().tobytes() == res["loop_form"].tobytes(), "the default at this size is the block-once form"
    assert res["flat_form"].tobytes() == res["old_form"].tobytes(), "old_form"
    assert res["flat_form"].tobytes() == res["wg_form"].tobytes(), "wg_form"
    np.testing.assert_allclose(res["loop_form"], res["flat_form"], atol=2e-5, rtol=2e-5)
    qs, ds, qm, dm, w = _multi_case(dev, seed)
    sample = np.arange(0, 9000, 9)                                   # every ninth pair against the oracle (1,000 pairs x 9 combinations)
    ref = np.zeros(sample.size)
    for i in range(3):
        for j in range(3):
            ref += O.tk_kernel_pool(qs[i][sample].numpy(), ds[j][sample].numpy(), qm[sample].numpy(), dm[sample].numpy(), MU, SIGMA,
                                    np.ones(11, np.float32), w[i * 3 + j].numpy(), dtype=np.float64)
    np.testing.assert_allclose(got.cpu().numpy()[sample], ref, atol=util.TOL_FP32, rtol=1e-5)


# ---- the ten equally spaced kernels by middle-out recurrence (kp_device.h rbf_geo_one) ---------------------------------------
# The recurrence is chosen per launch from the parameter VALUES: a sigma one part in 10^6 off leaves the equal-width test
# and makes the same launch run the direct form (twelve exp2 per cosine) — the two must tell the same story.

@pytest.mark.parametrize("Q,D,E,ppq,nq,gated", [(20, 200, 300, 8, 3, False),    # redistributed rows (3 lanes x 11) on the E = 100n kernel
                                                  (30, 180, 300, 1, 10, False),   # MFMA-layout epilogue
                                                  (20, 200, 300, 8, 3, True),     # TK-Sparse gate in the exponent
                                                  (30, 70, 128, 5, 4, False),     # 64n-wide kernel (Conv-KNRM's tensors, IDCM's ck-small)
                                                  (8, 64, 128, 16, 6, False),     # short query, short documents: two wavefronts per SIMD
                                                  (32, 100, 384, 3, 4, True)])
def test_rbf_recurrence_agrees_with_the_direct_form(Q, D, E, ppq, nq, gated):
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(Q * 977 + D + E)
    B = nq * ppq
    q = torch.randn(nq, Q, E, generator=g)
    d = torch.randn(B, D, E, generator=g) * 0.5
    for b in range(B):            # cosines over the whole range, both ends included: d = +-q (c = +-1), mixtures in between
        qq = q[b // ppq]
        d[b, 0] = qq[b % Q] * 3.0
        d[b, 1 % D] = -qq[(b + 1) % Q] * 0.25
        for j in range(2, min(D, 24)):
            lam = (j - 2) / 21.0 * 2.0 - 1.0
            n = torch.randn(E, generator=g)
            d[b, j] = lam * qq[(b + j) % Q] / qq[(b + j) % Q].norm() + (1 - abs(lam)) * n / n.norm()
    q_len = torch.randint(max(1, Q // 2), Q + 1, (nq,), generator=g)
    d_len = torch.randint(D // 3, D + 1, (B,), generator=g)
    d_len[0] = D
    d_len[-1] = 0                 # an empty document: every pooled sum exactly 0 in both forms
    qm = (torch.arange(Q)[None] < q_len[:, None]).float()
    dm = (torch.arange(D)[None] < d_len[:, None]).float()
    dm[0, D // 2] = 0.0
    dm[1, 0] = 0.0                # the exact match of pair 1 is masked
    gate = (torch.rand(B, D, generator=g) * 1.5).clamp(min=0.0) if gated else None
    if gated:
        gate[:, 3] = 0.0
    alpha = torch.rand(11, generator=g) + 0.5
    w = (torch.rand(11, generator=g) - 0.5) * 0.03
    mu = torch.tensor(MU)
    sigma = torch.tensor(SIGMA)
    sigma_off = sigma.clone()
    sigma_off[3] = 0.1 * (1.0 + 1.0e-6)                        # != its neighbours: direct form
    assert sigma_off[3] != sigma_off[2]

    def run(sg):
        return ops.kernel_pool(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), mu.to(dev), sg.to(dev), alpha.to(dev), w.to(dev),
                               pairs_per_query=ppq, return_per_kernel=True, d_gate=None if gate is None else gate.to(dev),
                               return_pooled=True)
    s_geo, pk_geo, pooled_geo = run(sigma)
    s_dir, pk_dir, pooled_dir = run(sigma_off)
    # pooled kernel sums of real query tokens (the backward reads exactly these)
    real = (qm[torch.arange(B) // ppq] > 0).to(dev)
    a, b_ = pooled_geo[real].double(), pooled_dir[real].double()
    rel = ((a - b_).abs() / (b_.abs() + 1e-6)).max().item()
    assert rel < 1e-4, f"pooled kernel sums differ by {rel:.2e} relative between the recurrence and the direct form"
    np.testing.assert_allclose(pk_geo.cpu().numpy(), pk_dir.cpu().numpy(), atol=2e-4, rtol=2e-5)
    np.testing.assert_allclose(s_geo.cpu().numpy(), s_dir.cpu().numpy(), atol=2e-5, rtol=1e-5)
    # the empty document and masked rows add exactly nothing: log(clamp_min) per real token in every kernel, both forms
    assert torch.equal(pk_geo[-1], pk_dir[-1])
    assert float(pooled_geo[-1][real[-1]].abs().max()) == 0.0
    # and both against the fp64 oracle
    qi = np.arange(B) // ppq
    if gate is None:
        ref = O.tk_kernel_pool(q.numpy()[qi], d.numpy(), qm.numpy()[qi], dm.numpy(), MU, SIGMA, alpha.numpy(), w.numpy(),
                               dtype=np.float64)
        np.testing.assert_allclose(s_geo.cpu().numpy(), ref, atol=util.TOL_FP32)


def test_rbf_recurrence_only_for_the_kernel_sets_it_was_derived_for():
    """Unequal spacing, unequal widths, ascending order, a width too small for the middle kernels to stay representable over
    [-1, 1]: each leaves the launch on the direct form, which the fp64 oracle pins (a recurrence over any of these would be
    wrong by orders of magnitude, not by rounding)."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    g = torch.Generator().manual_seed(77)
    Q, D, E, B = 30, 96, 128, 12
    q = torch.randn(B, Q, E, generator=g)
    d = torch.randn(B, D, E, generator=g)
    for b in range(B):
        d[b, b] = q[b, b] * 2.0
        d[b, b + 1] = -q[b, b + 1]
    qm = torch.ones(B, Q)
    dm = torch.ones(B, D)
    alpha = torch.ones(11)
    w = (torch.rand(11, generator=g) - 0.5) * 0.03
    sets = {
        "unequal spacing": (MU[:4] + [0.35] + MU[5:], SIGMA),
        "unequal widths": (MU, SIGMA[:6] + [0.15] + SIGMA[7:]),
        "ascending": ([1.0] + MU[:0:-1], SIGMA),
        "narrow": (MU, [0.001] + [0.04] * 10),
        "shifted": ([1.0] + [m + 0.5 for m in MU[1:]], SIGMA),
    }
    for name, (mu, sg) in sets.items():
        out = ops.kernel_pool(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), torch.tensor(mu).to(dev), torch.tensor(sg).to(dev),
                              alpha.to(dev), w.to(dev))
        ref = O.tk_kernel_pool(q.numpy(), d.numpy(), qm.numpy(), dm.numpy(), mu, sg, alpha.numpy(), w.numpy(), dtype=np.float64)
        np.testing.assert_allclose(out.cpu().numpy(), ref, atol=util.TOL_FP32, err_msg=name)
