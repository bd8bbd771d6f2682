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
