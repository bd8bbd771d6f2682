#!/usr/bin/env python
"""Where the host time of a 64-pair ColBERT training step goes (bench.py extra.train_step: ~155 us per step around ~45 us of
kernels): wall time per step, then cProfile over the same loop.   python tools/host_step_profile.py [n_steps]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from matchmaker_amd import ops, synth  # noqa: E402
from matchmaker_amd.colbert import ColBERT, _MaxSimFn  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    dev = torch.device("cuda", 0)
    B, Q, D, E = 64, 32, 180, 128
    g = torch.Generator(device=dev).manual_seed(64)
    q = torch.nn.functional.normalize(torch.randn(B, Q, E, generator=g, device=dev), dim=-1).half().requires_grad_(True)
    d = torch.nn.functional.normalize(torch.randn(B, D, E, generator=g, device=dev), dim=-1).half().requires_grad_(True)
    qm = synth.len_to_mask(torch.randint(4, Q + 1, (B,), generator=g, device=dev), Q)
    dm = synth.len_to_mask(synth.msmarco_doc_lengths(B, D, g, dev), D)
    go = torch.randn(B, generator=g, device=dev)

    def step():
        q.grad = d.grad = None
        with torch.autocast("cuda", dtype=torch.float16):
            ColBERT._score(q, d, qm, dm).backward(go)

    def fwd_only():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            ColBERT._score(q, d, qm, dm)

    def apply_only():
        with torch.autocast("cuda", dtype=torch.float16):
            ColBERT._score(q, d, qm, dm)

    def bwd_op_only():
        ops.maxsim_bwd(q, d, qm, dm, go, grad_dtype=q.dtype)

    def trivial_autograd():          # the framework alone: one elementwise node through the engine
        q.grad = None
        (q * 2.0).backward(q)

    for name, fn in (("step", step), ("forward, no grad", fwd_only), ("forward with graph (Function.apply)", apply_only),
                     ("backward operator alone", bwd_op_only), ("torch: (q * 2).backward(q)", trivial_autograd)):
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"== {name}: {1e6 * (t1 - t0) / n:.1f} us per call issued, {1e6 * (t2 - t0) / n:.1f} us completed")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:22]
    for (f, ln, fnname), (cc, nc, tt, ct, _) in rows:
        print(f"   {1e6 * tt / n:7.2f} us own  {1e6 * ct / n:7.2f} us cum  {nc / n:5.1f} calls  {os.path.basename(f)}:{ln} {fnname}")


if __name__ == "__main__":
    main()
