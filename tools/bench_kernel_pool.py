#!/usr/bin/env python
"""Micro-bench of mm_kernel_pool_fwd at BASELINE.json config-1 shapes scaled to a full GPU
(Q=20 / D=200 / E=300 fp32, 1000 candidates per query).  Prints pairs/s and algorithmic GB/s."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matchmaker_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--queries", type=int, default=16)
ap.add_argument("--cands", type=int, default=1000)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--full", action="store_true", help="all documents full length")
ap.add_argument("--gate", action="store_true", help="TK-Sparse: a ReLU-like per-token gate (d_gate)")
ap.add_argument("--shape", default="20,200,300", help="Q,D,E (IDCM sampler: 30,64,768 / 30,64,128 with --clamp 1e-4)")
ap.add_argument("--clamp", type=float, default=1e-10)
ap.add_argument("--qlen", default="full", help="'full' (every query token real) | 'config1' (U{3..Q} per query, SURVEY.md 8d) | an integer")
a = ap.parse_args()
dev = torch.device("cuda:0")
Q, D, E = (int(x) for x in a.shape.split(","))
g = torch.Generator(device=dev).manual_seed(1)
B = a.queries * a.cands
q = torch.randn(a.queries, Q, E, generator=g, device=dev)
d = torch.randn(B, D, E, generator=g, device=dev)
d_len = torch.full((B,), D, dtype=torch.int32, device=dev) if a.full else torch.randint(min(10, D), D + 1, (B,), generator=g, device=dev).to(torch.int32)
if a.qlen == "full":
    q_len = torch.full((a.queries,), Q, dtype=torch.int32, device=dev)
elif a.qlen == "config1":
    q_len = torch.randint(3, Q + 1, (a.queries,), generator=g, device=dev).to(torch.int32)
else:
    q_len = torch.full((a.queries,), int(a.qlen), dtype=torch.int32, device=dev)
p = [torch.tensor([1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9], device=dev), torch.full((11,), 0.1, device=dev),
     torch.ones(11, device=dev), torch.linspace(-0.014, 0.014, 11, device=dev)]
kw = dict(pairs_per_query=a.cands, clamp_min=a.clamp,
          d_gate=torch.relu(torch.randn(B, D, generator=g, device=dev)) if a.gate else None)
import bench
ms = bench.gpu_time_ms(lambda: ops.kernel_pool(q, d, q_len, d_len, *p, **kw), a.steps, warmup=3)
useful = int(((d_len + 31) // 32 * 32).clamp(max=D).sum().item()) * E * 4
padded = B * D * E * 4
print(json.dumps({"pairs_per_s": B / (ms * 1e-3), "ms": ms, "GBps_padded_bytes": padded / ms / 1e6,
                  "GBps_bytes_read": useful / ms / 1e6, "B": B, "full": a.full, "gate": a.gate, "shape": [Q, D, E], "clamp": a.clamp, "qlen": a.qlen}))
