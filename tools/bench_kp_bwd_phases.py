"""Phase clocks of the tiled TK backward kernel (kernel_pool_bwd_tiled_kernel built with -DMM_KP_BWD_PHASE_TIMES=1):

    tools/build_variant.sh phases kernel_pool_bwd -DMM_KP_BWD_PHASE_TIMES=1
    MM_NATIVE_LIB=variants/libmm_native_phases.so python tools/bench_kp_bwd_phases.py [pairs]

prints thread 0 / pair 0's s_memtime ticks per phase (100 MHz constant clock on gfx950) and the share of each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matchmaker_amd import ops  # noqa: E402

NAMES = ["setup", "s1 commit+barrier", "s1 norms", "s1 fetch+cosines", "s1 reduce", "s1 pool", "A_ik", "s2 commit+fetch+G", "s2 td/sq",
         "s2 grad_d", "s2 grad_q+barrier", "final grad_q store"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    dev = "cuda:0"
    Q, D, E, K = 20, 200, 300, 11
    g = torch.Generator(device=dev).manual_seed(1)
    q = torch.randn(n, Q, E, device=dev, generator=g)
    d = torch.randn(n, D, E, device=dev, generator=g)
    ql = torch.full((n,), Q, dtype=torch.int32, device=dev)
    dl = torch.full((n,), D, dtype=torch.int32, device=dev)
    mu = torch.tensor([1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9], device=dev)
    sigma = torch.tensor([0.001] + [0.1] * 10, device=dev)
    alpha = torch.ones(K, device=dev)
    w = torch.full((K,), 0.1, device=dev)
    go = torch.ones(n, device=dev)
    from matchmaker_amd import _lib
    L = _lib.lib()
    _, qp, qk = ops._mask(ql, n, Q, "q_mask")
    _, dp, dk = ops._mask(dl, n, D, "d_mask")
    gq, gd = torch.empty_like(q), torch.empty_like(d)
    ga = torch.zeros(n, K, device=dev)
    gw = torch.zeros(n, K, device=dev)
    wsb = L.mm_kernel_pool_bwd_workspace_bytes(n, Q, D, qk, dk)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(4):
        if it == 3:
            ev[0].record()
        rc = L.mm_kernel_pool_ex_bwd(q.data_ptr(), d.data_ptr(), qp, qk, dp, dk, None, mu.data_ptr(), sigma.data_ptr(),
                                     alpha.data_ptr(), w.data_ptr(), 1e-10, go.data_ptr(), gq.data_ptr(), gd.data_ptr(), None,
                                     ga.data_ptr(), gw.data_ptr(), n, Q, D, E, K, ws.data_ptr(), wsb, ops._stream(torch.device("cuda", 0)))
        assert rc == 0, rc
    ev[1].record()
    torch.cuda.synchronize()
    print(f"{n} pairs: {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us per launch")
    t = torch.cat([gw[0], ga[0][:1]]).tolist()
    tot = sum(t)
    for name, v in zip(NAMES, t):
        print(f"{name:24s} {v:10.0f} ticks  {100 * v / tot:5.1f} %")
    print(f"{'total':24s} {tot:10.0f} ticks = {tot / 100:.1f} us")


if __name__ == "__main__":
    main()
