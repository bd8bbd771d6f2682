"""Scratch: mm_tkl_bwd at 64 documents of config 3's lengths, repeated (for rocprofv3 --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from matchmaker_amd import ops
from matchmaker_amd.tkl import TKL_sigir20, chunk_documents
MU = [1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9]
dev = torch.device("cuda", 0)
B, Qt, Dt, Et = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 20, 2048, 300
g = torch.Generator(device=dev).manual_seed(640 + B)
m = TKL_sigir20(Et, MU, [0.1] * 11, 10, 2, 300, 2000, True, True, "embedding").to(dev).eval()
d_len = torch.randint(50, Dt + 1, (B,), generator=g, device=dev)
q_len = torch.randint(3, Qt + 1, (B,), generator=g, device=dev)
q = torch.randn(B, Qt, Et, generator=g, device=dev)
d = torch.randn(B, Dt, Et, generator=g, device=dev)
qm = (torch.arange(Qt, device=dev)[None] < q_len[:, None]).float()
dm = (torch.arange(Dt, device=dev)[None] < d_len[:, None]).float()
chunks, cmask, slot, C = chunk_documents(d * dm.unsqueeze(-1), dm)
params = m.pack_params()
s, win = ops.tkl_score(q * qm.unsqueeze(-1), chunks, cmask, slot, qm, params, B, C, 11, "embedding", return_windows=True)
go = torch.randn(B, generator=g, device=dev)
for _ in range(60):
    ops.tkl_bwd(q * qm.unsqueeze(-1), chunks, cmask, slot, qm, params, win, go, B, C, 11, "embedding")
torch.cuda.synchronize()
print("chunks", chunks.shape[0])
