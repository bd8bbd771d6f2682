"""Shared helpers for the test-suite (golden fixture loading, tolerances)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# BASELINE.json north_star: "scores within 1e-3 fp32 (1e-2 bf16)"
TOL_FP32 = 1e-3
TOL_BF16 = 1e-2


def bf16_bits_to_f32(bits: np.ndarray) -> np.ndarray:
    return (bits.astype(np.uint32) << 16).view(np.float32)


def load(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    out = {k: z[k] for k in z.files}
    for k in list(out):
        if k.endswith("_bf16"):
            out[k[:-5]] = bf16_bits_to_f32(out[k])
        elif k.endswith("_fp16"):
            out[k[:-5]] = out[k].astype(np.float32)
    return out


def golden_files(prefix):
    return sorted(f for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def tkl_params(g):
    from oracle import np_oracle as O
    return O.tkl_params_from_state({k[len("param."):]: v for k, v in g.items() if k.startswith("param.")})


def require_gpu():
    import torch
    assert torch.cuda.is_available(), "this test is marked gpu and needs a real MI355X"
    return torch.device("cuda:0")
