#!/usr/bin/env python
"""Where the host time of an eval.py-sized call goes (bench.py extra.eval_batch reports ~14-17 us per call issued):
cProfile over back-to-back 512-pair calls of the three shapes.   python tools/host_path_profile.py [n_calls]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from matchmaker_amd import ops  # noqa: E402
from matchmaker_amd.colbert import ColBERT  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    dev = torch.device("cuda", 0)
    B = 512
    g = torch.Generator(device=dev).manual_seed(1)

    def cb(Q, D, E, dt):
        return ((torch.randn(B, Q, E, generator=g, device=dev) / E ** 0.5).to(dt), (torch.randn(B, D, E, generator=g, device=dev) / E ** 0.5).to(dt),
                torch.ones(B, Q, dtype=torch.long, device=dev), torch.ones(B, D, dtype=torch.long, device=dev))
    mu = torch.tensor([1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9], device=dev)
    prm = [mu, torch.full((11,), 0.1, device=dev), torch.ones(11, device=dev), torch.linspace(-0.014, 0.014, 11, device=dev)]
    tk = (torch.randn(B, 20, 300, generator=g, device=dev), torch.randn(B, 200, 300, generator=g, device=dev),
          torch.ones(B, 20, device=dev), torch.ones(B, 200, device=dev))
    cases = {"colbert_dim128_bf16": (lambda b=cb(32, 180, 128, torch.bfloat16): ColBERT._score(*b)),
             "colbert_published_dim768_fp16": (lambda b=cb(38, 200, 768, torch.float16): ColBERT._score(*b)),
             "tk_dim300_fp32": (lambda: ops.kernel_pool(tk[0], tk[1], tk[2], tk[3], *prm, pairs_per_query=1))}
    for name, fn in cases.items():
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"== {name}: {1e6 * (t1 - t0) / n:.2f} us per call issued (no profiler)")
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(n):
            fn()
        pr.disable()
        torch.cuda.synchronize()
        st = pstats.Stats(pr)
        st.sort_stats("tottime")
        rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:14]
        for (f, ln, fnname), (cc, nc, tt, ct, _) in rows:
            print(f"   {1e6 * tt / n:7.2f} us own  {1e6 * ct / n:7.2f} us cum  {nc / n:5.1f} calls  {os.path.basename(f)}:{ln} {fnname}")


if __name__ == "__main__":
    main()
