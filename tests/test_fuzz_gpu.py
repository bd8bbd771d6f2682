"""GPU: seeded random-shape sweeps of every native operator against the oracle — shapes the targeted tests
do not enumerate (odd token counts, every streaming width, ragged tails that end inside a block, runs of
chunks of every length).  Each case is small; the whole file runs in a few seconds."""
import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from tests import util

pytestmark = pytest.mark.gpu

MU = [1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9]
SIGMA = [0.1] * 11


@pytest.mark.parametrize("seed", range(12))
def test_maxsim_random_shapes(seed):
    from matchmaker_amd import ops
    dev = util.require_gpu()
    rng = np.random.default_rng(1000 + seed)
    E = int(rng.choice([8, 24, 64, 128, 256, 384, 512, 768]))
    Q = int(rng.integers(1, 65))
    D = int(rng.integers(1, 260))
    ppq = int(rng.integers(1, 9))
    nq = int(rng.integers(1, 5))
    B = nq * ppq - int(rng.integers(0, ppq)) if nq > 1 else nq * ppq
    dtype = [torch.bfloat16, torch.float16, torch.float32][seed % 3]
    tol = util.TOL_FP32 if dtype == torch.float32 else util.TOL_BF16
    g = torch.Generator().manual_seed(seed)
    q = (torch.randn(nq, Q, E, generator=g) / E ** 0.5).to(dtype)
    d = (torch.randn(B, D, E, generator=g) / E ** 0.5).to(dtype)
    qm = (torch.rand(nq, Q, generator=g) > 0.2).long()
    dm = (torch.rand(B, D, generator=g) > 0.3).long()          # arbitrary (non-prefix) masks
    dm[0] = 1
    if B > 1:
        dm[-1] = 0
    out = ops.maxsim(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), pairs_per_query=ppq).cpu().numpy()
    qi = np.arange(B) // ppq
    ref = O.maxsim_paired(q.float().numpy()[qi], d.float().numpy(), qm.numpy()[qi], dm.numpy())
    np.testing.assert_allclose(out, ref, atol=tol, rtol=1e-4, err_msg=f"E={E} Q={Q} D={D} ppq={ppq} B={B} {dtype}")


@pytest.mark.parametrize("seed", range(8))
def test_ragged_maxsim_random_stores(seed):
    from matchmaker_amd import ops
    dev = util.require_gpu()
    rng = np.random.default_rng(2000 + seed)
    E = int(rng.choice([128, 256, 384, 512, 768, 40]))
    Q = int(rng.integers(1, 65 if E <= 512 else 33))
    dtype = [torch.float16, torch.bfloat16][seed % 2] if E != 40 else torch.float32
    n_docs = int(rng.integers(1, 60))
    lens = rng.integers(0, 150, n_docs)
    end = np.cumsum(lens)
    begin = end - lens
    T = max(int(end[-1]), 1)
    tok = torch.from_numpy(rng.standard_normal((T, E)).astype(np.float32) / np.sqrt(E)).to(dtype)
    nq, C = 2, int(rng.integers(1, 12))
    q = torch.from_numpy(rng.standard_normal((nq, Q, E)).astype(np.float32) / np.sqrt(E)).to(dtype)
    cand = rng.integers(0, n_docs, (nq, C))
    b = torch.from_numpy(begin[cand.reshape(-1)].astype(np.int64)).to(dev)
    e = torch.from_numpy(end[cand.reshape(-1)].astype(np.int64)).to(dev)
    out = ops.maxsim_ragged(q.to(dev), tok.to(dev), b, e, None, pairs_per_query=C).cpu().numpy().reshape(nq, C)
    tol = util.TOL_FP32 if dtype == torch.float32 else util.TOL_BF16
    for i in range(nq):
        for j in range(C):
            doc = tok[begin[cand[i, j]]: end[cand[i, j]]].float().numpy()
            ref = -1000.0 * Q if doc.shape[0] == 0 else float((q[i].float().numpy() @ doc.T).max(-1).sum())
            assert abs(out[i, j] - ref) <= tol + 1e-4 * abs(ref), (E, Q, i, j, out[i, j], ref)


@pytest.mark.parametrize("seed", range(8))
def test_kernel_pool_random_shapes(seed):
    from matchmaker_amd import ops
    dev = util.require_gpu()
    rng = np.random.default_rng(3000 + seed)
    E = int(rng.choice([100, 200, 300, 300, 64, 12]))
    Q = int(rng.integers(1, 41))
    D = int(rng.integers(1, 230))
    ppq = int(rng.integers(1, 6))
    nq = int(rng.integers(1, 4))
    B = nq * ppq
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(nq, Q, E, generator=g)
    d = torch.randn(B, D, E, generator=g)
    for bb in range(B):
        d[bb, bb % D] = q[bb // ppq, bb % Q] * 0.7
    qm = (torch.rand(nq, Q, generator=g) > 0.2).float()
    dm = (torch.rand(B, D, generator=g) > 0.3).float()
    alpha = torch.rand(11, generator=g) + 0.5
    w = torch.randn(11, generator=g) * 0.05
    out = ops.kernel_pool(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), torch.tensor(MU).to(dev), torch.tensor(SIGMA).to(dev),
                          alpha.to(dev), w.to(dev), pairs_per_query=ppq).cpu().numpy()
    qi = np.arange(B) // ppq
    ref = O.tk_kernel_pool(q.numpy()[qi], d.numpy(), qm.numpy()[qi], dm.numpy(), MU, SIGMA, alpha.numpy(), w.numpy(),
                           dtype=np.float64)
    np.testing.assert_allclose(out, ref, atol=util.TOL_FP32, rtol=1e-4, err_msg=f"E={E} Q={Q} D={D} ppq={ppq}")


@pytest.mark.parametrize("seed", range(6))
def test_tkl_random_lengths(seed):
    """document lengths that leave 1, 2, 3 or 4 kept chunks in the last run, holes, empty documents"""
    from tests.test_tkl_gpu import make_model
    dev = util.require_gpu()
    rng = np.random.default_rng(4000 + seed)
    E = [300, 100, 200, 300, 64, 300][seed]
    B, Q = 5, int(rng.integers(1, 21))
    D = int(rng.choice([37, 160, 333, 700, 1201, 2048]))
    m = make_model(E, "embedding" if seed % 2 == 0 else "log", dev)
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, Q, E, generator=g)
    d = torch.randn(B, D, E, generator=g)
    q_len = torch.randint(1, Q + 1, (B,), generator=g)
    d_len = torch.tensor([D, max(1, D // 3), min(D, 41), min(D, 121), 1])
    qm = (torch.arange(Q)[None] < q_len[:, None]).float()
    dm = (torch.arange(D)[None] < d_len[:, None]).float()
    if D > 60:
        dm[0, 50:55] = 0                                       # a hole inside a chunk
    with torch.no_grad():
        out = m.forward(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev)).cpu().numpy()
    params = O.tkl_params_from_state({k: v.cpu() for k, v in m.state_dict().items()})
    ref = O.tkl_forward_bypass(q.numpy(), d.numpy(), qm.numpy(), dm.numpy(), params, "embedding" if seed % 2 == 0 else "log",
                               dtype=np.float64)
    np.testing.assert_allclose(out, ref, atol=2e-3, rtol=2e-4, err_msg=f"E={E} Q={Q} D={D}")



@pytest.mark.parametrize("seed", range(6))
def test_dot_topk_random_shapes(seed):
    from matchmaker_amd import ops
    dev = util.require_gpu()
    rng = np.random.default_rng(5000 + seed)
    E = int(rng.choice([128, 256, 384, 512, 768]))
    N = int(rng.choice([1, 31, 4097, 9000, 33333]))
    nq = int(rng.integers(1, 300))
    k = int(rng.choice([1, 10, 100, 1000]))
    dtype = [torch.float16, torch.bfloat16][seed % 2]
    g = torch.Generator().manual_seed(seed)
    c = torch.randn(N, E, generator=g).to(dtype)
    q = torch.randn(nq, E, generator=g).to(dtype)
    s, idx = ops.dot_topk(q.to(dev), c.to(dev), k)
    s, idx = s.cpu().numpy(), idx.cpu().numpy()
    full = q.double().numpy() @ c.double().numpy().T
    kk = min(k, N)
    ref = -np.sort(-full, axis=1)[:, :kk]
    np.testing.assert_allclose(s[:, :kk], ref, atol=5e-2, rtol=2e-3, err_msg=f"E={E} N={N} nq={nq} k={k}")
    got = np.take_along_axis(full, idx[:, :kk], axis=1)
    np.testing.assert_allclose(s[:, :kk], got, atol=5e-2, rtol=2e-3)
    if kk < k:
        assert (idx[:, kk:] == -1).all()


@pytest.mark.parametrize("seed", range(10))
def test_fp32_maxsim_streaming_widths(seed):
    """fp32 MaxSim at the widths the split-bf16 streaming kernel serves (E = 64n <= 384, Q <= 32) and just outside
    them: arbitrary masks, shared-query layout with a short last group."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    rng = np.random.default_rng(5000 + seed)
    E = int(rng.choice([64, 128, 192, 256, 384, 320, 512]))
    Q = int(rng.integers(1, 36))
    D = int(rng.integers(1, 300))
    ppq = int(rng.integers(1, 9))
    nq = int(rng.integers(1, 6))
    B = nq * ppq - (int(rng.integers(0, ppq)) if nq > 1 else 0)
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(nq, Q, E, generator=g) / E ** 0.5
    d = torch.randn(B, D, E, generator=g) / E ** 0.5
    qm = (torch.rand(nq, Q, generator=g) > 0.2).long()
    dm = (torch.rand(B, D, generator=g) > 0.3).long()
    if seed % 2:
        dm = (torch.arange(D)[None] < torch.randint(0, D + 1, (B,), generator=g)[:, None]).long()   # prefix masks
    dm[0] = 1
    out = ops.maxsim(q.to(dev), d.to(dev), qm.to(dev), dm.to(dev), pairs_per_query=ppq).cpu().numpy()
    qi = np.arange(B) // ppq
    ref = O.maxsim_paired(q.numpy()[qi], d.numpy(), qm.numpy()[qi], dm.numpy(), dtype=np.float64)
    np.testing.assert_allclose(out, ref, atol=util.TOL_FP32, rtol=1e-5, err_msg=f"E={E} Q={Q} D={D} ppq={ppq} B={B}")


@pytest.mark.parametrize("seed", range(12))
def test_pooling_variants_random(seed):
    """Random combinations of the pooling variants — gate, floor, ragged query groups, kernel count — over the
    widths of all three kernel families (100n, 64n incl. the two-wave 512/768, generic)."""
    from matchmaker_amd import ops
    dev = util.require_gpu()
    rng = np.random.default_rng(6000 + seed)
    E = int(rng.choice([100, 300, 64, 128, 384, 512, 768, 36]))
    K = 11 if seed % 3 else int(rng.integers(1, 33))
    Q = int(rng.integers(1, 33))
    D = int(rng.integers(1, 150))
    nq = int(rng.integers(1, 6))
    groups = rng.integers(0, 5, nq)
    groups[0] = max(groups[0], 1)
    pq = np.repeat(np.arange(nq), groups)
    P = int(pq.size)
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(nq, Q, E, generator=g)
    d = torch.randn(P, D, E, generator=g)
    d[0, 0] = q[pq[0], 0]
    qm = (torch.arange(Q)[None] < torch.randint(1, Q + 1, (nq,), generator=g)[:, None]).float()
    dm = (torch.rand(P, D, generator=g) > 0.25).float()
    use_gate = bool(seed % 2)
    gate = torch.relu(torch.randn(P, D, generator=g)) if use_gate else None
    clamp = [1e-10, 1e-4, 1e-7][seed % 3]
    mu = torch.linspace(1.0, -0.9, K) if K > 1 else torch.tensor([0.5])
    sigma = torch.full((K,), 0.1)
    alpha, w = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g)
    t = lambda x: None if x is None else x.to(dev)
    out = ops.kernel_pool(t(q), t(d), t(qm), t(dm), t(mu), t(sigma), t(alpha), t(w), d_gate=t(gate), clamp_min=clamp,
                          pair_query=torch.from_numpy(pq).to(dev)).cpu().numpy()
    # oracle: TK pooling with the gate folded into the document mask and the floor as given
    eff = dm.numpy() * (gate.numpy() if use_gate else 1.0)
    cos = O.cosine_matrix(q.numpy()[pq], d.numpy(), np.float64)
    act = np.exp(-(cos[..., None] - mu.numpy().reshape(1, 1, 1, -1)) ** 2 / (2 * 0.1 ** 2)) * eff[:, None, :, None]
    lg = np.log(np.maximum(act.sum(2) * alpha.numpy().reshape(1, 1, -1), clamp)) * qm.numpy()[pq][..., None]
    ref = lg.sum(1) @ w.numpy().astype(np.float64)
    np.testing.assert_allclose(out, ref, atol=util.TOL_FP32 * max(1.0, K / 11), rtol=1e-5,
                               err_msg=f"E={E} K={K} Q={Q} D={D} gate={use_gate} clamp={clamp}")
