#!/usr/bin/env python
"""Experiment (round 5): do the windows of one sub-batch run under stage 1 of the next when the sub-batches' mm_tkl_fwd calls
alternate between two streams?  1,024 config-3 documents as ONE call vs 4 x 256 / 8 x 128 on two streams.
    [MM_NATIVE_LIB=variants/libmm_native_wpc2.so] python tools/exp_tkl_two_streams.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from matchmaker_amd import ops  # noqa: E402
from matchmaker_amd.tkl import TKL_sigir20, chunk_documents  # noqa: E402

MU = [1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9]


def main():
    dev = torch.device("cuda", 0)
    B, Q, D, E = 1024, 20, 2048, 300
    g = torch.Generator(device=dev).manual_seed(5)
    torch.manual_seed(1)
    m = TKL_sigir20(E, MU, [0.1] * 11, 10, 1, 32, 2000, True, True, "embedding").to(dev)
    params = m.pack_params()
    q_len = torch.randint(3, Q + 1, (B,), generator=g, device=dev)
    d_len = torch.randint(50, D + 1, (B,), generator=g, device=dev)
    qm = (torch.arange(Q, device=dev)[None] < q_len[:, None]).float()
    dm = (torch.arange(D, device=dev)[None] < d_len[:, None]).float()
    q_ctx = torch.randn(B, Q, E, generator=g, device=dev) * qm.unsqueeze(-1)

    def parts(n):
        out = []
        for s in range(n):
            lo, hi = B * s // n, B * (s + 1) // n
            d = torch.randn(hi - lo, D, E, generator=torch.Generator(device=dev).manual_seed(100 + lo), device=dev) * dm[lo:hi].unsqueeze(-1)
            ch, cm, sl, C = chunk_documents(d, dm[lo:hi])
            out.append((q_ctx[lo:hi].contiguous(), ch, cm, sl, qm[lo:hi].contiguous(), hi - lo, C))
        return out

    def timed(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t) / n

    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    for n in (1, 2, 4, 8):
        ps = parts(n)

        def one_stream():
            return [ops.tkl_score(p[0], p[1], p[2], p[3], p[4], params, p[5], p[6], 11, "embedding", check_order=False) for p in ps]

        def two_streams():
            main_s = torch.cuda.current_stream(dev)
            s1.wait_stream(main_s); s2.wait_stream(main_s)
            outs = []
            for i, p in enumerate(ps):
                with torch.cuda.stream(s1 if i % 2 == 0 else s2):
                    outs.append(ops.tkl_score(p[0], p[1], p[2], p[3], p[4], params, p[5], p[6], 11, "embedding", check_order=False))
            main_s.wait_stream(s1); main_s.wait_stream(s2)
            return outs
        a, b = one_stream(), two_streams()
        torch.cuda.synchronize()
        assert all(torch.equal(x, y) for x, y in zip(a, b))
        print(f"{n} sub-batches of {B // n}: one stream {timed(one_stream):.4f} ms, two streams {timed(two_streams):.4f} ms", flush=True)


if __name__ == "__main__":
    main()
