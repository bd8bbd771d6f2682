#!/usr/bin/env python
"""Host-side cost of one operator call (ctypes + workspace + launch) at eval.py-sized batches: wall time per call with the
GPU kept busy (async launches), and the device time of the same call."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matchmaker_amd import ops
dev = torch.device("cuda:0")


def measure(name, fn, n=300):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print(f"{name}: host {t_host * 1e6:.1f} us / call issued, {t_all * 1e6:.1f} us / call completed")


for B, Q, D, E, dt in ((256, 32, 180, 128, torch.bfloat16), (256, 38, 200, 768, torch.float16), (1000, 32, 180, 128, torch.bfloat16)):
    q = torch.randn(B, Q, E, device=dev).to(dt); d = torch.randn(B, D, E, device=dev).to(dt)
    qm = torch.ones(B, Q, dtype=torch.long, device=dev); dm = torch.ones(B, D, dtype=torch.long, device=dev)
    measure(f"maxsim pair-per-row B={B} Q={Q} D={D} E={E}", lambda: ops.maxsim(q, d, qm, dm, 1))
B, Q, D, E = 256, 20, 200, 300
q = torch.randn(B, Q, E, device=dev); d = torch.randn(B, D, E, device=dev)
qm = torch.ones(B, Q, device=dev); dm = torch.ones(B, D, device=dev)
mu = torch.linspace(1.0, -0.9, 11, device=dev); sg = torch.full((11,), 0.1, device=dev); al = torch.ones(11, device=dev); w = torch.ones(11, device=dev)
measure(f"kernel_pool pair-per-row B={B} Q={Q} D={D} E={E}", lambda: ops.kernel_pool(q, d, qm, dm, mu, sg, al, w))
