"""Scratch: the bench's TKL leg with another query-length range (tools/scratch/tkl_qlen.py <max q len>) — stage-1 time without two-N-tile documents."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from matchmaker_amd import ops
from matchmaker_amd.tkl import TKL_sigir20, chunk_documents
MU = [1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9]
qmax = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
B, Qt, Dt, Et = 256, 20, 2048, 300
g = torch.Generator(device=dev).manual_seed(3003)
m = TKL_sigir20(Et, MU, [0.1] * 11, 10, 2, 300, 2000, True, True, "embedding").to(dev).eval()
q = torch.randn(B, Qt, Et, generator=g, device=dev)
d = torch.randn(B, Dt, Et, generator=g, device=dev)
d_len = torch.randint(50, Dt + 1, (B,), generator=g, device=dev)
q_len = torch.randint(3, qmax + 1, (B,), generator=g, device=dev)
qm = (torch.arange(Qt, device=dev)[None] < q_len[:, None]).float()
dm = (torch.arange(Dt, device=dev)[None] < d_len[:, None]).float()
q_ctx = q * qm.unsqueeze(-1)
chunks, cmask, slot, C = chunk_documents(d * dm.unsqueeze(-1), dm)
params = m.pack_params()
fn = lambda: ops.tkl_score(q_ctx, chunks, cmask, slot, qm, params, B, C, 11, "embedding", check_order=False)
for _ in range(20): fn()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
for a, b in ev:
    a.record(); fn(); b.record()
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in ev)
print(f"qmax {qmax}: median {ts[100] * 1e3:.1f} us per call, chunks {chunks.shape[0]}")
