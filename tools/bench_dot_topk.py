#!/usr/bin/env python
"""Micro-bench of mm_dot_topk_fwd at BASELINE.json config-5 shapes for ONE GPU's shard
(8,841,823 / 8 = 1.1 M passages x dim 768 fp16, top-1000).  Prints queries/s and achieved TFLOP/s
(2 * nq * N * E / time, the full pass only does useful MFMA work)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matchmaker_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=1105228)
ap.add_argument("--queries", type=int, default=6980)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--k", type=int, default=1000)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--raw", action="store_true", help="time the bare mm_dot_topk_fwd call and ignore the status vector (by-removal builds whose results are wrong)")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5005)
c = torch.empty((a.docs, a.dim), dtype=torch.float16, device=dev)
for s in range(0, a.docs, 1 << 18):
    n = min(1 << 18, a.docs - s)
    c[s:s + n] = torch.randn(n, a.dim, generator=g, device=dev).half()
q = torch.randn(a.queries, a.dim, generator=g, device=dev).half()
if a.raw:
    from matchmaker_amd import _lib
    L = _lib.lib()
    s_ = torch.empty((a.queries, a.k), dtype=torch.float32, device=dev)
    i_ = torch.empty((a.queries, a.k), dtype=torch.int64, device=dev)
    st = torch.empty(a.queries, dtype=torch.int32, device=dev)
    wsb = L.mm_dot_topk_workspace_bytes(a.docs, a.queries, a.k)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)

    def call():
        return L.mm_dot_topk_fwd(q.data_ptr(), c.data_ptr(), a.docs, a.queries, a.dim, ops._DT[q.dtype], a.k, 1.0, s_.data_ptr(),
                                 i_.data_ptr(), st.data_ptr(), ws.data_ptr(), wsb, ops._stream(dev))
    call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        call()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"raw_ms": t * 1e3, "bad_status": int((st != 0).sum())}))
    sys.exit(0)
ops.dot_topk(q, c, a.k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    s, i = ops.dot_topk(q, c, a.k)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / a.steps
flop = 2.0 * a.queries * a.docs * a.dim
print(json.dumps({"queries_per_s": a.queries / t, "ms": t * 1e3, "TFLOPs": flop / t / 1e12, "docs": a.docs,
                  "queries": a.queries, "dim": a.dim, "k": a.k,
                  "frac_of_2500TF": flop / t / 2.5e15}))
