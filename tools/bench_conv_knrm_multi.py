#!/usr/bin/env python
"""Conv-KNRM's 3 x 3 multi launch alone (bench.py extra.variants' shape: 64 x 1000 pairs, Q30 / D200 / E128), for counter passes:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o c -- python tools/bench_conv_knrm_multi.py [steps]
    MM_KP_MULTI_2D=1 MM_KP128_OCC=1 ... (rounds 1-4's 2-D grid)   |   MM_KP_MULTI_WG=1 ... (wavefront per query tensor)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from matchmaker_amd import ops  # noqa: E402

MU = [1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(9090)
    B, Q, D, E = 64000, 30, 200, 128
    q = [torch.randn(B, Q, E, generator=g, device=dev) for _ in range(3)]
    d = [torch.randn(B, D, E, generator=g, device=dev) for _ in range(3)]
    q_len = torch.randint(3, Q + 1, (B,), generator=g, device=dev).to(torch.int32)
    d_len = torch.randint(50, D + 1, (B,), generator=g, device=dev).to(torch.int32)
    prm = [torch.tensor(MU, device=dev), torch.full((11,), 0.1, device=dev), torch.ones(11, device=dev)]
    w9 = torch.linspace(-0.014, 0.014, 99, device=dev)
    for _ in range(2):
        ops.kernel_pool_multi(q, d, q_len, d_len, prm[0], prm[1], prm[2], w9)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        ops.kernel_pool_multi(q, d, q_len, d_len, prm[0], prm[1], prm[2], w9)
    b.record()
    torch.cuda.synchronize()
    rows = int(((d_len + 31) // 32 * 32).clamp(max=D).sum())
    print(f"conv_knrm 3x3: {a.elapsed_time(b) / steps:.3f} ms per launch; document bytes below the lengths, read ONCE: {3 * rows * E * 4 / 1e9:.2f} GB")


if __name__ == "__main__":
    main()
