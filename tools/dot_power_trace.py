#!/usr/bin/env python
"""Socket power and shader clock sampled DURING back-to-back launches (prove or refute the power-cap reading
of the dot top-k filter launch): for each workload a ~6 s loop on the GPU while a sampler thread polls rocm-smi, then
TFLOP/s, median power, median shader clock and energy per FLOP.

    python tools/dot_power_trace.py            (on the GPU box; prints one JSON line per workload + a table)

Workloads: the whole mm_dot_topk_fwd call of one rank's shard of BASELINE.json configs[4] (6,980 queries x 1,105,228 passages x
768, fp16, top-1000: 97 % of it is the filter launch dot_stream_kernel), the vendor GEMM (torch.mm -> hipBLASLt) of the SAME
product, and the vendor's 8192^3 fp16 GEMM (what the board sustains when every LDS byte feeds more MFMAs)."""
import json
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from matchmaker_amd import ops  # noqa: E402


def sample():
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=30)
    out = {}
    for line in r.stdout.splitlines():
        if not line.startswith("GPU[0]"):
            continue
        m = re.search(r"(sclk|mclk) clock level: \d+: \((\d+)Mhz\)", line)
        if m:
            out[m.group(1)] = int(m.group(2))
        m = re.search(r"Power \(W\): ([0-9.]+)", line)
        if m:
            out["power_w"] = float(m.group(1))
    return out


def run(name, fn, flop_per_call, seconds=6.0):
    fn(); torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def poll():
        time.sleep(1.0)                       # past the ramp
        while not stop.is_set():
            try:
                s = sample()
                if s:
                    samples.append(s)
            except Exception:
                pass
    th = threading.Thread(target=poll)
    th.start()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        n += 4
    t = time.perf_counter() - t0
    stop.set(); th.join()
    med = lambda k: sorted(s[k] for s in samples if k in s)[len(samples) // 2] if samples else None
    tf = flop_per_call * n / t / 1e12
    rec = {"workload": name, "calls": n, "ms_per_call": 1e3 * t / n, "TFLOPs": tf, "frac_of_2500TF": tf / 2500.0, "samples": len(samples),
           "median_power_w": med("power_w"), "median_sclk_mhz": med("sclk"), "median_mclk_mhz": med("mclk"),
           "pJ_per_FLOP": (med("power_w") / (tf * 1e12) * 1e12) if samples and med("power_w") and tf > 0 else None,
           "power_min_max_w": [min(s["power_w"] for s in samples if "power_w" in s), max(s["power_w"] for s in samples if "power_w" in s)] if samples else None}
    print(json.dumps(rec), flush=True)
    return rec


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5005)
    N, E, nq = 1105228, 768, 6980
    c = torch.empty((N, E), dtype=torch.float16, device=dev)
    for s in range(0, N, 1 << 18):
        n = min(1 << 18, N - s)
        c[s:s + n] = torch.randn(n, E, generator=g, device=dev).half()
    q = torch.randn(nq, E, generator=g, device=dev).half()
    recs = [run("idle (no launches)", lambda: time.sleep(0.05), 0.0, seconds=3.0)]
    recs.append(run("mm_dot_topk_fwd (filter launch + selection), one rank's shard of configs[4]", lambda: ops.dot_topk(q, c, 1000), 2.0 * nq * N * E))
    qt = q.t().contiguous()
    out = torch.empty((N, nq), dtype=torch.float16, device=dev) if N * nq * 2 < 40e9 else None
    recs.append(run("vendor GEMM of the same product (torch.mm corpus x queries^T, fp16)", lambda: torch.mm(c, qt, out=out), 2.0 * nq * N * E))
    del out
    a = torch.randn(8192, 8192, generator=g, device=dev).half()
    b = torch.randn(8192, 8192, generator=g, device=dev).half()
    o = torch.empty(8192, 8192, dtype=torch.float16, device=dev)
    recs.append(run("vendor GEMM 8192^3 fp16", lambda: torch.mm(a, b, out=o), 2.0 * 8192 ** 3))
    print(f"{'workload':78s} {'TFLOP/s':>8s} {'W':>7s} {'sclk':>6s} {'pJ/FLOP':>8s}")
    for r in recs:
        pj = f"{r['pJ_per_FLOP']:.3f}" if r["pJ_per_FLOP"] and r["TFLOPs"] > 1 else "-"
        print(f"{r['workload'][:78]:78s} {r['TFLOPs']:8.0f} {r['median_power_w'] or 0:7.0f} {r['median_sclk_mhz'] or 0:6d} {pj:>8s}")


if __name__ == "__main__":
    main()
