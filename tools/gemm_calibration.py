#!/usr/bin/env python
"""What the vendor GEMM (torch.mm -> hipBLASLt / rocBLAS) reaches on this box at the shapes of the MFMA-bound legs — the
power-feasible matrix rate the dot top-k filter (0.94 PFLOP/s incl. its threshold epilogue) and the all-pairs MaxSim kernel
(1.1 PFLOP/s incl. the running maxima) can be compared with; the 2.5 PFLOP/s bf16 peak assumes 2.4 GHz at 100 % MFMA issue.
    python tools/gemm_calibration.py"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from bench import gpu_time_ms  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    out = {}
    for name, (M, N, K, dt) in {
        "dot_topk_shape_fp16 (6980 x 768) x (768 x 262144)": (6980, 262144, 768, torch.float16),
        "dot_topk_shape_bf16": (6980, 262144, 768, torch.bfloat16),
        "all_pairs_shape_bf16 (32768 x 128) x (128 x 184320)": (32768, 184320, 128, torch.bfloat16),
        "square_8192_bf16": (8192, 8192, 8192, torch.bfloat16),
        "square_8192_fp16": (8192, 8192, 8192, torch.float16),
    }.items():
        a = torch.randn(M, K, device=dev, dtype=dt)
        b = torch.randn(N, K, device=dev, dtype=dt)
        c = torch.empty(M, N, device=dev, dtype=dt)
        ms = gpu_time_ms(lambda: torch.mm(a, b.t(), out=c), 10, warm_ms=200.0, timed_ms=200.0)
        out[name] = {"ms": ms, "tflops": 2.0 * M * N * K / (ms * 1e-3) / 1e12, "output_gb": M * N * c.element_size() / 1e9}
        del a, b, c
        torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
