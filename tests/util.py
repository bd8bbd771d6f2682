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


# ---- TKL region arg-max: tie policy -----------------------------------------------------------------------
# sigir20_tkl.py:262-286 picks three windows by arg-max with +-15 suppression and sums 15 window scores around them: the
# document score is DISCONTINUOUS in the window scores.  When two candidate peaks lie closer together than the rounding
# noise of the window arithmetic, two correct evaluations (the reference's own fp32 and fp64 runs among them) pick different
# regions and their document scores differ by far more than the window noise.  So "within 1e-3 of the reference" is only
# defined for documents whose region choice is decided; the others are *region-tied* and get their own, equally strict checks.

def tkl_region_search(win_row):
    """:254-273 on ONE document's window scores (numpy, any float dtype): 0 -> -9900, three arg-max rounds (first maximal
    index, like torch.argmax), |r - best| < 15 suppression to -10001 - c.  Returns the three peaks in round order."""
    w = np.asarray(win_row).copy()
    if w.shape[0] < 3:
        w = np.concatenate([w, np.zeros(3 - w.shape[0], w.dtype)])
    w[w == 0] = -9900
    r = np.arange(w.shape[0])
    peaks = []
    for c in range(3):
        b = int(np.argmax(w))
        peaks.append(b)
        w[np.abs(r - b) < 15] = -10001 - c
    return peaks


def tkl_score_at(win_row, peaks, chunk_scoring):
    """:276-286 — the document score the window scores `win_row` give for the regions `peaks` (fp64)."""
    w = np.asarray(win_row, dtype=np.float64)
    if w.shape[0] < 3:
        w = np.concatenate([w, np.zeros(3 - w.shape[0])])
    p = np.asarray(peaks, dtype=np.int64)
    idx = np.clip(np.concatenate([p, p - 1, p + 1, p - 2, p + 2]), 0, w.shape[0] - 1)
    return float((w[idx] * np.asarray(chunk_scoring, dtype=np.float64).reshape(-1)).sum())


def tkl_region_classify(w64_row, dev_peaks, gap_tol):
    """Replays the three arg-max rounds on the fp64 oracle's window scores FOLLOWING the device's choices.  Returns
    ("same" | "tied" | "wrong", gap): "same" = the device's peaks are the fp64 oracle's; otherwise gap = the largest amount by
    which, in some round, the device's choice lies below the best window still available in fp64 — "tied" when gap <=
    gap_tol (the choice is within the window arithmetic's noise of the best one), "wrong" when not (or when the device chose
    a suppressed window)."""
    ref = tkl_region_search(w64_row)
    dev = [int(x) for x in dev_peaks]
    if dev == ref:
        return "same", 0.0
    w = np.asarray(w64_row, dtype=np.float64).copy()
    if w.shape[0] < 3:
        w = np.concatenate([w, np.zeros(3 - w.shape[0])])
    w[w == 0] = -9900
    r = np.arange(w.shape[0])
    gap = 0.0
    for c in range(3):
        gap = max(gap, float(w.max() - w[dev[c]]))
        w[np.abs(r - dev[c]) < 15] = -10001 - c
    return ("tied" if gap <= gap_tol else "wrong"), gap


def tkl_check_documents(score, win, peaks, s64, w64, chunk_scoring, label="tkl", verbose=True):
    """The tie policy applied to one batch (see run_tkl_rank in tests/test_zz_rank_order_gpu.py and DESIGN.md §4).  score [B],
    win [B, W], peaks [B, 3] from the device; s64 [B], w64 [B, >= W] from the fp64 oracle.  Asserts: every window within 1e-3,
    empty windows exactly 0 on both sides, the kernel's peaks = the search on its own windows, its score = the 15-term sum of
    its own windows, `same` documents within 1e-3 of the oracle, no `wrong` region, `tied` documents within 1e-3 of the fp64
    windows evaluated at the device's regions.  Returns (tied mask [B], device window error, largest tied gap)."""
    score, win, peaks = np.asarray(score), np.asarray(win), np.asarray(peaks)
    B, W = win.shape
    w64 = np.asarray(w64)[:, :W]
    np.testing.assert_allclose(win, w64, atol=TOL_FP32, rtol=1e-5, err_msg=f"{label}: window scores")
    assert ((win == 0) == (w64 == 0)).all(), f"{label}: empty windows must be exactly 0 on both sides (:248, :257)"
    aw = float(np.abs(win - w64).max()) if win.size else 0.0
    gap_tol = 4.0 * max(aw, 1e-7)
    tied = np.zeros(B, dtype=bool)
    worst = 0.0
    for b in range(B):
        own = tkl_region_search(win[b])
        assert own == [int(x) for x in peaks[b]], f"{label} doc {b}: kernel peaks {peaks[b]} != search on its own windows {own}"
        at_own = tkl_score_at(win[b], own, chunk_scoring)
        assert abs(score[b] - at_own) <= 2e-5 + 2e-6 * abs(at_own), f"{label} doc {b}: score {score[b]} is not the 15-term sum of its windows {at_own}"
        cls, gap = tkl_region_classify(w64[b], peaks[b], gap_tol)
        assert cls != "wrong", (f"{label} doc {b}: the device chose a region {gap:.3e} below the best available window "
                                f"(window noise {aw:.3e}, tie bound {gap_tol:.3e})")
        if cls == "same":
            assert abs(score[b] - s64[b]) <= TOL_FP32 + 1e-5 * abs(s64[b]), \
                f"{label} doc {b}: same regions as the oracle, score {score[b]} vs {s64[b]}"
        else:
            tied[b] = True
            worst = max(worst, gap)
            at_dev = tkl_score_at(w64[b], peaks[b], chunk_scoring)
            assert abs(score[b] - at_dev) <= TOL_FP32 + 1e-5 * abs(at_dev), \
                f"{label} doc {b}: region-tied, score {score[b]} vs fp64 windows at the device's regions {at_dev}"
            if verbose:
                print(f"[tkl tie] {label} doc {b}: peaks {peaks[b].tolist()} vs oracle {tkl_region_search(w64[b])}, "
                      f"gap {gap:.3e} <= {gap_tol:.3e}; score {score[b]:.6f} oracle {s64[b]:.6f}")
    return tied, aw, worst
