#!/usr/bin/env python
"""All-pairs (in-batch) MaxSim at dynamic-teacher batch sizes (colbert.py:154-162, dynamic_teacher.py:245-276)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matchmaker_amd import ops
dev = torch.device("cuda:0")
SIZES = ((32, 32), (64, 64), (128, 128), (256, 256), (1024, 1024), (512, 4096))
if len(sys.argv) == 3:            # one size only (profiling): python tools/bench_inbatch.py 1024 1024
    SIZES = ((int(sys.argv[1]), int(sys.argv[2])),)
for Bq, Bd in SIZES:
    q = torch.randn(Bq, 32, 128, device=dev).bfloat16(); d = torch.randn(Bd, 180, 128, device=dev).bfloat16()
    qm = torch.ones(Bq, 32, dtype=torch.long, device=dev); dm = torch.ones(Bd, 180, dtype=torch.long, device=dev)
    for _ in range(3): ops.maxsim_inbatch(q, qm, d, dm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): ops.maxsim_inbatch(q, qm, d, dm)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 50
    print(Bq, Bd, f"{t*1e6:.1f} us per call, {Bq*Bd/t/1e6:.1f} M pairs/s")
