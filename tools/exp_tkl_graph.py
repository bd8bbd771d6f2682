#!/usr/bin/env python
"""Experiment (round 6): what do the three kernel boundaries of one mm_tkl_fwd call cost?  The bench's TKL workload (256
config-3 documents) timed three ways on one box: per-call HIP events around the eager call (what extra.tkl.ms is), the same
call replayed from a hipGraph (torch.cuda.CUDAGraph around ops.tkl_score), and N calls back to back divided by N (both forms).
    python tools/exp_tkl_graph.py  ->  four lines, ms per call"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from matchmaker_amd import ops  # noqa: E402
from matchmaker_amd.tkl import TKL_sigir20, chunk_documents  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    B, Q, D, E = 256, 20, 2048, 300
    g = torch.Generator(device=dev).manual_seed(3003)
    m = TKL_sigir20(E, bench.MU, [0.1] * 11, 10, 2, 300, 2000, True, True, "embedding").to(dev).eval()
    q = torch.randn(B, Q, E, generator=g, device=dev)
    d = torch.randn(B, D, E, generator=g, device=dev)
    d_len = torch.randint(50, D + 1, (B,), generator=g, device=dev)
    q_len = torch.randint(3, Q + 1, (B,), generator=g, device=dev)
    qm = (torch.arange(Q, device=dev)[None] < q_len[:, None]).float()
    dm = (torch.arange(D, device=dev)[None] < d_len[:, None]).float()
    q_ctx = q * qm.unsqueeze(-1)
    chunks, cmask, slot, C = chunk_documents(d * dm.unsqueeze(-1), dm)
    params = m.pack_params()
    fn = lambda: ops.tkl_score(q_ctx, chunks, cmask, slot, qm, params, B, C, 11, "embedding", check_order=False)
    ref = fn()
    if isinstance(ref, tuple):
        ref = ref[0]
    ref = ref.clone()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = fn()
    if isinstance(out, tuple):
        out = out[0]
    graph.replay()
    torch.cuda.synchronize()
    print("graph replay bit-equal to the eager call:", bool(torch.equal(out, ref)))

    def back_to_back(f, n=400):
        for _ in range(100):
            f()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            f()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    for rnd in range(2):                                      # round-robin: the box drifts
        print(f"round {rnd}: eager per-call events {bench.gpu_time_ms(fn, 50):.4f} ms | graph per-call events "
              f"{bench.gpu_time_ms(graph.replay, 50):.4f} ms | eager back to back {back_to_back(fn):.4f} ms | "
              f"graph back to back {back_to_back(graph.replay):.4f} ms")


if __name__ == "__main__":
    main()
