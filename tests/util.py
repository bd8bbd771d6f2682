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


def load_flow16(name):
    """tests/golden/flow16_*.npz: outputs of the REAL reference class on fp16 / bf16 CPU tensors (all-16-bit dtype flow).
    Returns the dict with q / d as float32 arrays holding the 16-bit values, and `lowp` = np.float16 | "bfloat16"."""
    g = load(name)
    bf = str(g["dtype"]) == "bf16"
    for k in ("q", "d"):
        bits = g[k + "_bits"]
        g[k] = bf16_bits_to_f32(bits) if bf else bits.view(np.float16).astype(np.float32)
    g["lowp"] = "bfloat16" if bf else np.float16
    return g


def ulps16(got, ref, lowp):
    """|got - ref| in units of ref's 16-bit ulp (fp16: 11 significant bits, bf16: 8)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    bits = 8 if lowp == "bfloat16" else 11
    mag = np.maximum(np.abs(ref), np.abs(got))
    e = np.floor(np.log2(np.maximum(mag, 2.0 ** (-126 if lowp == "bfloat16" else -14))))
    return np.abs(got - ref) / 2.0 ** (e - (bits - 1))


def golden_files(prefix):
    return sorted(f for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def tkl_params(g):
    from oracle import np_oracle as O
    return O.tkl_params_from_state({k[len("param."):]: v for k, v in g.items() if k.startswith("param.")})


def require_gpu():
    import torch
    assert torch.cuda.is_available(), "this test is marked gpu and needs a real MI355X"
    return torch.device("cuda:0")


def rank_parity(got, ref32, ref64, ks=(1, 10, 100, 1000), noise=None, label=""):
    """north_star: "identical top-k rank order versus the reference".  The reference ranks a query's candidates
    with a stable descending sort of its fp32 scores (utils/core_metrics.py:502-511).  Two correct fp32
    evaluations of the same sums (the reference's bmm on the CPU, the MFMA kernel here) differ by accumulation
    order, so the order of two candidates is DEFINED only when their exact (fp64) scores are further apart than
    that rounding noise.  The noise is measured on the arithmetic itself, not assumed.  With
        b = max |fp32 oracle - fp64 oracle|   (the rounding error of the reference's own fp32 evaluation)
        a = max |device      - fp64 oracle|   (asserted <= 4 b: the device is an fp32-class evaluation)
    two candidates whose fp64 scores differ by more than  noise = 2 max(a, b)  are ordered the same way by both
    evaluations (each score moves by at most max(a, b)); closer pairs are ties whose order the reference's own
    arithmetic does not define.  a, b are ~1e-6 at config 2, i.e. noise is a few 1e-6.  Checked per query:
      * every pair further apart than the noise is ordered as in the fp64 ranking (suffix-max test over the
        device order: O(n), covers all pairs);
      * positions whose two fp64 neighbour gaps both exceed the noise ("decided") hold the same candidate;
      * top-k sets for every k in `ks`: identical when the k-th / (k+1)-th gap is decided, otherwise they may
        only differ by candidates within `noise` of the k-th score;
      * assumption-free statistic: positions at which the device ranking equals the stable descending sort of
        the fp32 oracle (what the reference's metrics code would see).
    Returns counters for the caller's >= 99 % "decided" assertion and report."""
    got = np.asarray(got, dtype=np.float64)
    ref32 = np.asarray(ref32, dtype=np.float64)
    ref64 = np.asarray(ref64, dtype=np.float64)
    n = got.shape[0]
    err = float(np.abs(got - ref64).max())
    err_ref = float(np.abs(ref32 - ref64).max())
    floor = 2 * np.finfo(np.float32).eps * float(np.abs(ref64).max())      # two ulps of the largest fp32 score
    if noise is None:
        assert err <= 4 * err_ref + floor, (f"{label}: device scores {err:.3e} away from the fp64 scores; the fp32 "
                                            f"oracle is {err_ref:.3e} away")
        noise = 2.0 * max(err, err_ref)
    else:
        assert err <= noise / 2, f"{label}: device scores {err:.3e} away from the fp64 scores, bound {noise / 2:.3e}"
    noise = float(max(noise, floor))
    order_ref = np.argsort(-ref64, kind="stable")
    order_got = np.argsort(-got, kind="stable")
    # (1) every pair further apart than the noise is ordered as in the reference
    v = ref64[order_got]
    suffix_max = np.maximum.accumulate(v[::-1])[::-1]
    worst = float((suffix_max[1:] - v[:-1]).max()) if n > 1 else 0.0
    assert worst <= noise, f"{label}: a candidate pair {worst:.3e} apart in fp64 is ranked in the opposite order"
    # (2) decided positions hold the same candidate
    gaps = np.abs(np.diff(ref64[order_ref]))
    big = gaps > noise
    decided = np.concatenate([[True], big]) & np.concatenate([big, [True]])
    assert (order_ref[decided] == order_got[decided]).all(), f"{label}: a decided rank position differs"
    # (3) top-k sets
    k_exact, k_tied = [], []
    for k in ks:
        if k > n:
            continue
        a, b = set(order_ref[:k].tolist()), set(order_got[:k].tolist())
        if k == n or gaps[k - 1] > noise:
            assert a == b, f"{label}: top-{k} set differs although the cut is decided"
            k_exact.append(k)
        else:
            kth = ref64[order_ref[k - 1]]
            for j in a ^ b:
                assert abs(ref64[j] - kth) <= noise, f"{label}: top-{k} sets differ by candidate {j}, not a tie"
            k_tied.append(k)
    same_as_fp32_sort = int((np.argsort(-ref32, kind="stable") == order_got).sum())
    return {"n": n, "decided": int(decided.sum()), "undecided": int(n - decided.sum()), "noise": noise, "err": err,
            "k_exact": k_exact, "k_tied": k_tied, "identical_positions_vs_fp32_sort": same_as_fp32_sort}


def rank_report(name, rows):
    """Print (pytest -s / captured log) and, on the GPU box, drop under gpurun_out/ a per-query table of
    decided / undecided rank positions; returns the decided fraction."""
    import json
    tot = sum(r["n"] for r in rows)
    dec = sum(r["decided"] for r in rows)
    ident = sum(r["identical_positions_vs_fp32_sort"] for r in rows)
    summary = {"test": name, "queries": len(rows), "positions": tot, "decided": dec, "undecided": tot - dec,
               "decided_frac": dec / max(tot, 1), "identical_positions_vs_stable_sort_of_fp32_reference": ident,
               "max_noise": max(r["noise"] for r in rows), "max_err_vs_fp64": max(r["err"] for r in rows),
               "topk_cuts_decided": sum(len(r["k_exact"]) for r in rows),
               "topk_cuts_tied": sum(len(r["k_tied"]) for r in rows),
               "per_query_undecided": [r["undecided"] for r in rows]}
    print("[rank parity] " + json.dumps(summary))
    out = os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out")
    if os.environ.get("GRAFT_REPO_ROOT"):
        os.makedirs(out, exist_ok=True)
    if os.path.isdir(out):
        with open(os.path.join(out, f"rank_parity_{name}.json"), "w") as f:
            json.dump(summary, f)
    return summary["decided_frac"]
