"""GPU: torch.ops.mm_native.* — same numbers as matchmaker_amd.ops, autograd through the native backward kernels,
and the autocast rules (fp16 MaxSim as under the reference's torch.cuda.amp.autocast, colbert.py:60; fp32 pooling)."""
import numpy as np
import pytest
import torch

import matchmaker_amd.torch_ops  # noqa: F401
from matchmaker_amd import ops
from oracle import np_oracle as O
from tests import util

pytestmark = pytest.mark.gpu

MU = [1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9]


def test_maxsim_op_autograd_and_autocast():
    dev = util.require_gpu()
    gen = torch.Generator().manual_seed(1)
    B, Q, D, E = 6, 32, 180, 128
    q = torch.randn(B, Q, E, generator=gen).to(dev)
    d = torch.randn(B, D, E, generator=gen).to(dev)
    qm = (torch.arange(Q)[None] < torch.randint(1, Q + 1, (B,), generator=gen)[:, None]).long().to(dev)
    dm = (torch.arange(D)[None] < torch.randint(1, D + 1, (B,), generator=gen)[:, None]).long().to(dev)
    s = torch.ops.mm_native.maxsim(q, d, qm, dm, 1)
    assert torch.equal(s, ops.maxsim(q, d, qm, dm))
    ref = O.maxsim_paired(q.cpu().numpy(), d.cpu().numpy(), qm.cpu().numpy(), dm.cpu().numpy(), dtype=np.float64)
    np.testing.assert_allclose(s.cpu().numpy(), ref, atol=util.TOL_FP32, rtol=1e-5)
    # autocast: fp32 inputs run through the fp16 kernel AND every per-token maximum is rounded to fp16 — what the reference's
    # autocast does to its bmm / max (colbert.py:60-75) — fp32 scores; bit-equal to torch's own eager ops under autocast
    with torch.autocast("cuda", dtype=torch.float16):
        sa = torch.ops.mm_native.maxsim(q, d, qm, dm, 1)
        from oracle import torch_port as TP
        eager = TP.maxsim_forward(q, d, qm, dm)
    assert sa.dtype == torch.float32 and torch.equal(sa, ops.maxsim(q.half(), d.half(), qm, dm, sim_round=True))
    assert eager.dtype == torch.float32 and float((sa - eager).abs().max()) <= 2.0 ** -6      # (one fp16 ulp of a token of size < 32)
    np.testing.assert_allclose(sa.cpu().numpy(), ref, atol=0.3, rtol=util.TOL_BF16)
    # autograd through mm_maxsim_bwd
    ql, dl = q.clone().requires_grad_(True), d.clone().requires_grad_(True)
    go = torch.randn(B, generator=gen).to(dev)
    (torch.ops.mm_native.maxsim(ql, dl, qm, dm, 1) * go).sum().backward()
    gq, gd = ops.maxsim_bwd(q, d, qm, dm, go)
    assert torch.equal(ql.grad, gq) and torch.equal(dl.grad, gd)
    ib = torch.ops.mm_native.maxsim_inbatch(q, qm, d, dm, False)
    assert torch.equal(ib, ops.maxsim_inbatch(q, qm, d, dm))
    np.testing.assert_allclose(torch.diagonal(ib).cpu().numpy(), ref, atol=util.TOL_FP32, rtol=1e-5)


def test_kernel_pool_op_autograd_and_autocast():
    dev = util.require_gpu()
    gen = torch.Generator().manual_seed(2)
    B, Q, D, E = 5, 20, 200, 300
    q = torch.randn(B, Q, E, generator=gen).to(dev)
    d = torch.randn(B, D, E, generator=gen).to(dev)
    qm = (torch.arange(Q)[None] < torch.randint(1, Q + 1, (B,), generator=gen)[:, None]).float().to(dev)
    dm = (torch.arange(D)[None] < torch.randint(1, D + 1, (B,), generator=gen)[:, None]).float().to(dev)
    mu, sigma = torch.tensor(MU).to(dev), torch.full((11,), 0.1).to(dev)
    alpha, w = (torch.rand(11, generator=gen) + 0.5).to(dev), torch.randn(11, generator=gen).to(dev)
    gate = torch.relu(torch.randn(B, D, generator=gen)).to(dev)
    s = torch.ops.mm_native.kernel_pool(q, d, qm, dm, mu, sigma, alpha, w, 1, gate, 1e-10)
    assert torch.equal(s, ops.kernel_pool(q, d, qm, dm, mu, sigma, alpha, w, d_gate=gate))
    with torch.autocast("cuda", dtype=torch.float16):                # the pooling family stays fp32 under autocast
        sa = torch.ops.mm_native.kernel_pool(q.half(), d.half(), qm, dm, mu, sigma, alpha, w, 1, gate, 1e-10)
    assert torch.equal(sa, ops.kernel_pool(q.half().float(), d.half().float(), qm, dm, mu, sigma, alpha, w, d_gate=gate))
    leaves = [t.clone().requires_grad_(True) for t in (q, d, alpha, w, gate)]
    go = torch.randn(B, generator=gen).to(dev)
    (torch.ops.mm_native.kernel_pool(leaves[0], leaves[1], qm, dm, mu, sigma, leaves[2], leaves[3], 1, leaves[4], 1e-10) * go).sum().backward()
    want = ops.kernel_pool_bwd(q, d, qm, dm, mu, sigma, alpha, w, go, d_gate=gate)
    for leaf, g in zip(leaves, want):
        assert torch.equal(leaf.grad, g.view_as(leaf))


def test_tkl_window_pool_op_equals_the_dropin_path():
    from matchmaker_amd.tkl import chunk_documents
    from tests.test_tkl_gpu import make_model
    dev = util.require_gpu()
    torch.manual_seed(3)
    m = make_model(64, "embedding", dev)
    B, Q, D = 3, 12, 500
    q, d = torch.randn(B, Q, 64, device=dev), torch.randn(B, D, 64, device=dev)
    qm = torch.ones(B, Q, device=dev)
    dm = (torch.arange(D, device=dev)[None] < torch.tensor([500, 77, 300], device=dev)[:, None]).float()
    with torch.no_grad():
        want, sec = m.forward(q, d, qm, dm, output_secondary_output=True)
        q_ctx, _ = m.forward_representation(q, qm)
        chunks, cmask, slot, C = chunk_documents(d, dm)
        chunks_ctx, _ = m.forward_representation(chunks, cmask)
        score, win = torch.ops.mm_native.tkl_window_pool(q_ctx, chunks_ctx, cmask, slot, qm, m.pack_params(), B, C, 11, "embedding")
    assert torch.equal(score, want) and torch.equal(win, sec["orig_score"])
    # autograd through the registered op = mm_tkl_bwd: the same gradients as the drop-in's training path
    q_l = q_ctx.clone().requires_grad_(True)
    c_l = chunks_ctx.clone().requires_grad_(True)
    p_l = m.pack_params().clone().requires_grad_(True)
    s2, _ = torch.ops.mm_native.tkl_window_pool(q_l, c_l, cmask, slot, qm, p_l, B, C, 11, "embedding")
    s2.sum().backward()
    m.train()
    q2, d2 = q.clone().requires_grad_(True), d.clone().requires_grad_(True)
    m.forward(q2, d2, qm, dm).sum().backward()
    assert torch.isfinite(q_l.grad).all() and float(c_l.grad.abs().sum()) > 0
    torch.testing.assert_close(p_l.grad[22:33], m.dense.weight.grad.view(-1), rtol=1e-5, atol=1e-6)     # dense block of the packed vector
    torch.testing.assert_close(q_l.grad * qm.unsqueeze(-1), q2.grad, rtol=1e-4, atol=1e-5)             # bypass model: q_ctx = q * mask


def test_operators_are_reentrant_across_threads_and_streams():
    """nn.DataParallel drives forward() from one Python thread per replica (train.py:201): the operators must hold
    no shared mutable state and honour the calling thread's current stream."""
    import threading
    dev = util.require_gpu()
    gen = torch.Generator().manual_seed(4)
    q = torch.randn(8, 32, 128, generator=gen).to(dev).to(torch.bfloat16)
    d = torch.randn(8 * 50, 180, 128, generator=gen).to(dev).to(torch.bfloat16)
    d_len = torch.randint(1, 181, (400,), generator=gen).to(torch.int32).to(dev)
    qf, df = torch.randn(6, 20, 300, generator=gen).to(dev), torch.randn(6, 200, 300, generator=gen).to(dev)
    prm = [torch.tensor(MU).to(dev), torch.full((11,), 0.1).to(dev), torch.ones(11).to(dev), torch.randn(11, generator=gen).to(dev)]
    want_a = ops.maxsim(q, d, None, d_len, pairs_per_query=50)
    want_b = ops.kernel_pool(qf, df, None, None, *prm)
    torch.cuda.synchronize()
    errors = []

    def worker(i):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                for _ in range(25):
                    a = ops.maxsim(q, d, None, d_len, pairs_per_query=50)
                    b = ops.kernel_pool(qf, df, None, None, *prm)
                st.synchronize()
            if not (torch.equal(a, want_a) and torch.equal(b, want_b)):
                errors.append("thread %d: results differ" % i)
        except Exception as e:           # noqa: BLE001
            errors.append("thread %d: %r" % (i, e))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors


def test_operators_are_graph_capturable():
    """Fixed-shape re-ranking loops are launch-bound at small batches (TKL's scoring alone is seven launches): the
    operators neither synchronise nor touch the host, so a scoring step can be captured in a HIP graph
    (torch.cuda.graph) and replayed on new data in the same buffers.  (TKL's chunk PACKING is data dependent —
    `nonzero`, as in the reference — so the graph starts at the packed chunks.)"""
    from matchmaker_amd.tkl import chunk_documents
    from tests.test_tkl_gpu import make_model
    dev = util.require_gpu()
    gen = torch.Generator().manual_seed(9)
    q = torch.randn(4, 32, 128, generator=gen).to(dev).to(torch.bfloat16)
    d = torch.randn(4 * 25, 180, 128, generator=gen).to(dev).to(torch.bfloat16)
    d_len = torch.randint(1, 181, (100,), generator=gen).to(torch.int32).to(dev)
    qf, df = torch.randn(6, 20, 300, generator=gen).to(dev), torch.randn(6, 200, 300, generator=gen).to(dev)
    dm = (torch.arange(200)[None] < torch.randint(1, 201, (6,), generator=gen)[:, None]).float().to(dev)
    prm = [torch.tensor(MU).to(dev), torch.full((11,), 0.1).to(dev), torch.ones(11).to(dev), torch.randn(11, generator=gen).to(dev)]
    torch.manual_seed(5)
    tkl = make_model(64, "embedding", dev)
    ql, dl = torch.randn(3, 12, 64, device=dev), torch.randn(3, 500, 64, device=dev)
    qml = torch.ones(3, 12, device=dev)
    dml = (torch.arange(500, device=dev)[None] < torch.tensor([500, 77, 300], device=dev)[:, None]).float()
    chunks, cmask, slot, C = chunk_documents(dl, dml)
    params = tkl.pack_params()

    def step():
        with torch.no_grad():
            return (ops.maxsim(q, d, None, d_len, pairs_per_query=25), ops.kernel_pool(qf, df, None, dm, *prm),
                    ops.tkl_score(ql, chunks, cmask, slot, qml, params, 3, C, 11, "embedding"))

    step()                                               # warm-up outside the capture (lazy library / attribute state)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = step()
    # new contents in the captured input buffers
    d.copy_(torch.randn(100, 180, 128, generator=gen).to(dev).to(torch.bfloat16))
    df.copy_(torch.randn(6, 200, 300, generator=gen).to(dev))
    chunks.copy_(torch.randn(chunks.shape, generator=gen).to(dev))
    graph.replay()
    torch.cuda.synchronize()
    want = step()
    for got, ref in zip(outs, want):
        assert torch.equal(got, ref)
