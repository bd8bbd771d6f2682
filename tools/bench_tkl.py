#!/usr/bin/env python
"""Micro-bench of mm_tkl_fwd at BASELINE.json config 3 (D=2048 sliding windows, E=300, Q=20, fp32):
pre-contextualised packed chunks resident in HBM -> scores.  Prints docs/s and algorithmic GB/s
(bytes = P*50*E*4 chunk stream + query + masks)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matchmaker_amd import ops
from matchmaker_amd.tkl import TKL_sigir20, chunk_documents

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=256)
ap.add_argument("--D", type=int, default=2048)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--Q", type=int, default=20, help="padded query length (effective lengths are U{3..Q})")
ap.add_argument("--full", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
B, Q, D, E = a.docs, a.Q, a.D, 300
g = torch.Generator(device=dev).manual_seed(3003)
MU = [1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9]
m = TKL_sigir20(E, MU, [0.1] * 11, 10, 1, 32, 2000, True, True, "embedding").to(dev).eval()
q = torch.randn(B, Q, E, generator=g, device=dev)
d = torch.randn(B, D, E, generator=g, device=dev)
d_len = torch.full((B,), D, device=dev) if a.full else torch.randint(50, D + 1, (B,), generator=g, device=dev)
q_len = torch.randint(3, Q + 1, (B,), generator=g, device=dev)
qm = (torch.arange(Q, device=dev)[None] < q_len[:, None]).float()
dm = (torch.arange(D, device=dev)[None] < d_len[:, None]).float()
q_ctx = q * qm.unsqueeze(-1)
chunks, cmask, slot, C = chunk_documents(d * dm.unsqueeze(-1), dm)      # stands in for the contextualised chunks
del d
params = m.pack_params()
P = chunks.shape[0]
fn = lambda: ops.tkl_score(q_ctx, chunks, cmask, slot, qm, params, B, C, 11, "embedding", check_order=False)
import bench
ms = bench.gpu_time_ms(fn, a.steps, warmup=3)      # steady state: warmed up for ~40 ms of device time first
byt = P * 50 * E * 4 + B * Q * E * 4 + P * 50 * 4 + 4 * B
print(json.dumps({"docs_per_s": B / (ms * 1e-3), "ms": ms, "GBps_algorithmic": byt / ms / 1e6, "B": B, "P": P, "C": C,
                  "bytes": byt}))
