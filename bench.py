#!/usr/bin/env python
"""bench.py — query-doc pairs scored / second, ColBERT MaxSim (Q32 / D180 / dim128, bf16,
1000 candidates per query; BASELINE.json configs[1]), on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 starts itself: when no torch.distributed environment is present the script re-executes
under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` (one process
per GPU over RCCL), the way the reference starts all GPUs from one `python train.py`
(matchmaker/train.py:194-202); launched under torchrun by someone else it just joins the group.

A "step" = one pass of the hot path (mm_maxsim_fwd) over one resident batch of `--queries` x 1000 synthetic
(query, candidate) pairs per GPU.  Inputs are generated on the device before the timed region (resident in
HBM).  Every rank scores its own shard of queries (weak scaling, no data-path collective); with N > 1 one
RCCL all-gather of the fp32 scores per step is part of the timed region (the ranking merge of SURVEY.md §8e).

Prints ONE JSON line (rank 0).  `roofline`: algorithmic bytes per launch / average kernel
time measured with HIP events on the launch stream; `cpu_baseline`: the torch CPU port of the reference's ops
(oracle/torch_port.py) timed on this host's cores on a bounded sample of the same workload.  At N = 1 the line
also carries `extra`: the reference's own batch layout through the same operator (`dropin_forward`: query
replicated per pair, HF int64 masks — what eval.py:108 hands to ColBERT.forward), and the other BASELINE.json
configs on this GPU (`tk` configs[0] shapes at scale, `tkl` configs[2], `dot_topk` one rank's shard of
configs[4]), each with its own roofline fraction, CPU leg and the rocprof summary it can be checked against.

`--only <leg>` runs ONE leg (headline | dropin_forward | published_checkpoint | all_pairs | tk | tkl | dot_topk |
maxsim_fp32 | eval_batch) and prints it alone: the command tools/profile_round.sh puts under `rocprofv3 --kernel-trace`, so
that every fraction in the line can be re-derived from a profile of the very code that produced it.

`--dry` (with `--backend gloo --device cpu`) exercises ONLY the launch + collective plumbing on a machine
without GPUs (tests/test_bench_launch_cpu.py): scores are fabricated, `value` is null — never a measurement.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

Q, D, E, CANDS = 32, 180, 128, 1000
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_PEAK_16BIT = 2.5e15     # dense bf16 / fp16 MFMA peak, FLOP/s (MI355X_MICROARCH.md; no sparsity)
MU = [1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9]
LEAN = False                 # --lean: a leg runs its main measurement only (profiling runs: one workload per kernel name)


def algorithmic_bytes(n_queries: int, cands: int) -> int:
    """B*D*E*s + Nq*Q*E*s + 4*(B + Nq) + 4*B  (shared-Q layout, int32 lengths)."""
    B = n_queries * cands
    return B * D * E * 2 + n_queries * Q * E * 2 + 4 * (B + n_queries) + 4 * B


def dropin_bytes(B: int) -> int:
    """The reference's pair-per-row layout (54,280 B/pair) + its int64 HF masks as they are read."""
    return B * ((D + Q) * E * 2 + 8 * (D + Q) + 4)


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args) -> int:
    """One command starts all ranks (train.py:194-202 starts its GPUs from one process too)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL / cross-process device memory)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def timed_cpu(fn, budget_s, unit_per_call):
    """2 warm-ups, then whole calls until ~budget_s seconds -> (units/s, calls, seconds)."""
    fn(); fn()
    t_total, n = 0.0, 0
    while t_total < budget_s or n < 5:
        t0 = time.perf_counter()
        fn()
        t_total += time.perf_counter() - t0
        n += 1
        if n >= 5 and t_total >= budget_s:
            break
    return unit_per_call * n / t_total, n, t_total


def both_thread_settings(make_fn, budget_s, unit_per_call):
    """(all-threads rate, single-thread rate, threads): BASELINE.md §2 asks for torch.set_num_threads(1) — what
    the reference's scripts run with (train.py:12 exports OMP_NUM_THREADS=1) — and the host's best case."""
    import torch
    threads = torch.get_num_threads()
    r_all = timed_cpu(make_fn(), budget_s, unit_per_call)
    torch.set_num_threads(1)
    try:
        r_one = timed_cpu(make_fn(), budget_s / 2, unit_per_call)
    finally:
        torch.set_num_threads(threads)
    return r_all, r_one, threads


def cpu_baseline(q_cpu, d_cpu, q_len, d_len, cands, budget_s=12.0):
    """The reference's CPU path for this block = the torch ops of colbert.py:68-75, repeated op by op
    in oracle/torch_port.py (pinned on the real reference's golden outputs); /root/reference itself
    does not exist on the GPU box, hence kind "port".  Reference layout: fp32, query replicated per
    pair, int64 HF masks, one candidate list (1000 pairs) per forward call.  Bounded sample: whole
    queries until ~budget_s seconds; a short single-thread leg is reported too because the
    reference's scripts export OMP_NUM_THREADS=1 (train.py:12)."""
    import torch
    from oracle import torch_port as TP
    from matchmaker_amd import synth
    qn = q_cpu.float()
    nq = qn.shape[0]

    def run(i):
        dn = d_cpu[i * cands:(i + 1) * cands].float()
        dm = synth.len_to_mask(d_len[i * cands:(i + 1) * cands], D)
        qr = qn[i:i + 1].expand(cands, -1, -1).contiguous()
        qm = synth.len_to_mask(q_len[i:i + 1], Q).expand(cands, -1).contiguous()
        t0 = time.perf_counter()
        with torch.no_grad():
            TP.maxsim_forward(qr, dn, qm, dm)
        return time.perf_counter() - t0

    def leg(budget):
        run(0)
        t_total, pairs, i = 0.0, 0, 0
        while t_total < budget:
            t_total += run(i % nq)
            pairs += cands
            i += 1
        return pairs / t_total, i, pairs, t_total

    threads = torch.get_num_threads()
    rate, n, pairs, t_total = leg(budget_s)
    torch.set_num_threads(1)
    rate1, n1, pairs1, t1 = leg(min(4.0, budget_s / 3))
    torch.set_num_threads(threads)
    return {"value": rate, "unit": "pairs/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
            "single_thread_value": rate1,
            "sample": f"{n} forward calls over whole queries ({nq} distinct) x {cands} candidates = {pairs} pairs of the "
                      f"bench workload in {t_total:.1f} s with {threads} torch threads (host has "
                      f"{os.cpu_count()} logical CPUs); fp32 torch CPU port of colbert.py:68-75 "
                      f"(oracle/torch_port.py: bmm, masked assign, max, sum); single-thread leg "
                      f"(OMP_NUM_THREADS=1 as in train.py:12): {pairs1} pairs in {t1:.1f} s"}


def profile_summary(pattern, kernel_substr, key):
    """A figure from the committed rocprofv3 summaries (profiles/*.json, tools/summarize_rocprof.py):
    returns (value, file) of the newest matching profile, or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern))):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        for k, c in j.get("pmc", {}).items():
            if kernel_substr in k and key in c:
                v = c[key]
                best = (v["total"] if isinstance(v, dict) and "total" in v else v, os.path.basename(f), j.get("workload", {}))
    return best


def gpu_time_ms(fn, steps, warmup=2, warm_ms=40.0, timed_ms=25.0, max_calls=400):
    """Median of per-call HIP-event times on the current stream (the stream the operators launch on), in steady state:
    after `warmup` calls the call is repeated until ~warm_ms of device time have passed, and at least `steps` calls — as
    many as ~timed_ms of device time hold — are timed.  A leg whose call takes 0.15-1.5 ms is otherwise measured while the
    GPU still climbs out of its idle clocks: the first ten launches of the 1.4 ms all-pairs kernel ran 12 % slower than
    the next ten in the same process (1.54 vs 1.37 ms); an evaluation loop keeps the device busy."""
    import torch
    for _ in range(warmup):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = max(a.elapsed_time(b), 1e-3)
    for _ in range(min(max_calls, int(warm_ms / t))):
        fn()
    n = min(max_calls, max(steps, int(timed_ms / t)))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]


def vendor_gemm_tflops(shapes):
    """torch.mm (hipBLASLt / rocBLAS) at the given {name: (M, N, K, dtype)} shapes, 16-bit output written: a calibration of
    what matrix rate this board sustains, not part of any product path."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    res = {}
    for name, (M, N, K, dt) in shapes.items():
        a = torch.randn(M, K, device=dev, dtype=dt)
        b = torch.randn(N, K, device=dev, dtype=dt)
        c = torch.empty(M, N, device=dev, dtype=dt)
        ms = gpu_time_ms(lambda: torch.mm(a, b.t(), out=c), 5, warm_ms=100.0, timed_ms=60.0)
        res[name] = {"shape_mnk": [M, N, K], "ms": ms, "tflops": 2.0 * M * N * K / (ms * 1e-3) / 1e12}
        del a, b, c
    torch.cuda.empty_cache()
    return res


# ------------------------------------------------------------------------------------------ extras (N = 1)
def extra_dropin_forward(q, d, q_len, d_len, steps):
    """eval.py:108 -> ColBERT.forward -> colbert.py:68-75 exactly as the reference batches it: the query
    replicated per pair, HF int64 attention masks, pairs_per_query = 1 — on the SAME pairs as the headline."""
    import torch
    from matchmaker_amd import ops, synth
    from matchmaker_amd.colbert import ColBERT
    nq, B = q.shape[0], d.shape[0]
    qp = q.repeat_interleave(CANDS, 0).contiguous()
    qm = synth.len_to_mask(q_len, Q, torch.int64).repeat_interleave(CANDS, 0).contiguous()
    dm = synth.len_to_mask(d_len, D, torch.int64)
    ref = ops.maxsim(q, d, q_len, d_len, pairs_per_query=CANDS)
    got = ColBERT._score(qp, d, qm, dm)                   # the drop-in's scoring entry (no_grad: native forward)
    same = bool(torch.equal(ref, got))
    ms = gpu_time_ms(lambda: ColBERT._score(qp, d, qm, dm), steps)
    by = dropin_bytes(B)
    gbs = by / (ms * 1e-3) / 1e9
    return {"workload": f"the headline's {nq} x {CANDS} pairs in the reference's batch layout: Q replicated per pair "
                        f"[{B},{Q},{E}] bf16, int64 HF masks [{B},{Q}] / [{B},{D}], pairs_per_query = 1 (ColBERT._score)",
            "dtype": "bf16", "ms": ms, "pairs_per_s": B / (ms * 1e-3), "algorithmic_bytes": by,
            "bytes_per_pair": by // B, "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                    "frac": gbs / HBM_PEAK_GBS},
            "bit_identical_to_shared_q_scores": same, "profile": "profiles/r03_dropin_forward_trace.json"}


def extra_sustained(score_shard, B, seconds=4.0):
    """The same launch back to back for a few seconds of wall clock: long enough for an external sampler (rocm-smi at a
    multi-second cadence) to see the GPU busy, and a cross-check of the 20-step figure on the host clock."""
    import torch
    score_shard()
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while True:
        for _ in range(100):
            score_shard()
        torch.cuda.synchronize()
        n += 100
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    return {"workload": "the headline launch repeated back to back", "steps": n, "seconds": dt, "ms_per_step": 1e3 * dt / n,
            "pairs_per_s": n * B / dt}


def tk_exact_f32_subprocess():
    """TK pooling on the exact-f32 MFMA kernel (MM_KP_F32MFMA=1, read once per process -> a child process), timed
    beside the split-bf16 default: VERDICT r01 asked for the A/B in the driver-timed line."""
    env = dict(os.environ, MM_KP_F32MFMA="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_kernel_pool.py"), "--full", "--queries", "64", "--steps", "5"],
                       capture_output=True, text=True, env=env, timeout=300)
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            j = json.loads(line)
            return {"kernel": "kernel_pool_stream_kernel (v_mfma_f32_32x32x2_f32, exact fp32 operands)", "ms": j["ms"],
                    "pairs_per_s": j["pairs_per_s"], "GBps": j["GBps_padded_bytes"]}
    return {"error": (r.stderr or r.stdout)[-300:]}


def extra_tk(steps, cpu_budget):
    """BASELINE.json configs[0] shapes (TK kernel pooling, Q=20 / D=200 / dim=300, fp32) at GPU scale:
    64 queries x 1000 candidates, every position real (the padded figure of DESIGN.md §3.3)."""
    import torch
    from matchmaker_amd import ops
    dev = torch.device("cuda", torch.cuda.current_device())
    nq, C, Qt, Dt, Et = 64, 1000, 20, 200, 300
    B = nq * C
    g = torch.Generator(device=dev).manual_seed(1001)
    q = torch.randn(nq, Qt, Et, generator=g, device=dev)
    d = torch.randn(B, Dt, Et, generator=g, device=dev)
    q_len = torch.full((nq,), Qt, dtype=torch.int32, device=dev)
    d_len = torch.full((B,), Dt, dtype=torch.int32, device=dev)
    prm = [torch.tensor(MU, device=dev), torch.full((11,), 0.1, device=dev), torch.ones(11, device=dev),
           torch.linspace(-0.014, 0.014, 11, device=dev)]
    fn = lambda: ops.kernel_pool(q, d, q_len, d_len, *prm, pairs_per_query=C)
    ms = gpu_time_ms(fn, steps)
    by = B * Dt * Et * 4 + nq * Qt * Et * 4 + 4 * (B + nq) + 4 * B
    gbs = by / (ms * 1e-3) / 1e9
    out = {"workload": f"TK kernel pooling (ecai20_tk.py:105-124), {nq} queries x {C} candidates, Q={Qt}/D={Dt}/dim={Et}, "
                       f"all positions real, shared query tile, int32 lengths",
           "dtype": "fp32 (split-bf16 operands: x = hi + lo, 4 bf16 MFMAs, fp32 accumulation)", "ms": ms,
           "pairs_per_s": B / (ms * 1e-3), "algorithmic_bytes": by, "flop": B * (2 * Qt * Dt * Et + 2 * (Qt + Dt) * Et),
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS},
           "kernel": "kernel_pool_split_kernel", "profile": "profiles/r03_tk_pmc.json, profiles/r03_tk_trace.json"}
    del q, d
    torch.cuda.empty_cache()
    try:
        if not LEAN:
            out["exact_f32_mfma"] = tk_exact_f32_subprocess()
    except Exception as e:
        out["exact_f32_mfma"] = {"error": repr(e)}
    g = torch.Generator(device=dev).manual_seed(1001)
    q = torch.randn(nq, Qt, Et, generator=g, device=dev)
    d = torch.randn(200, Dt, Et, generator=g, device=dev)
    # --- CPU legs: scoring only (torch port of :105-124) and full forward (+ the 2-layer Transformer contextualiser)
    if cpu_budget > 0:
        from oracle import torch_port as TP
        from matchmaker_amd.tk import ECAI20_TK
        n = 200
        qc, dc = q[:1].cpu().expand(n, -1, -1).contiguous(), d[:n].cpu()
        qm, dm = torch.ones(n, Qt), torch.ones(n, Dt)
        mu, sg = prm[0].cpu().view(1, 1, 1, -1), prm[1].cpu().view(1, 1, 1, -1)
        al, w = prm[2].cpu().view(1, 1, -1), prm[3].cpu().view(1, -1)
        model = ECAI20_TK(Et, MU, [0.1] * 11, att_heads=10, att_layer=2, att_ff_dim=300, max_length=Dt,
                          use_diff_posencoding=True, mix_hybrid_context=True).eval()      # tk.yaml

        def scoring():
            with torch.no_grad():
                TP.tk_kernel_pool(qc, dc, qm, dm, mu, sg, al, w)

        def full():
            with torch.no_grad():
                qx = model.forward_representation(qc, qm, model.positional_features_q[:, :Qt])
                dx = model.forward_representation(dc, dm, model.positional_features_d[:, :Dt])
                TP.tk_kernel_pool(qx, dx, qm, dm, mu, sg, al, w)
        (ra, na, ta), (r1, n1, t1), threads = both_thread_settings(lambda: scoring, cpu_budget, n)
        (fa, _, _), (f1, _, _), _ = both_thread_settings(lambda: full, cpu_budget, n)
        out["cpu_baseline"] = {"value": ra, "unit": "pairs/s", "cores": threads, "kind": "port", "single_thread_value": r1,
                               "full_forward_value": fa, "full_forward_single_thread_value": f1,
                               "sample": f"{n}-pair forward calls, {na} timed calls in {ta:.1f} s ({threads} threads) / {n1} in "
                                         f"{t1:.1f} s (1 thread); scoring only = oracle/torch_port.tk_kernel_pool; full forward = "
                                         f"the drop-in's own PyTorch contextualiser (2 layers, 10 heads, tk.yaml) + that block"}
    return out


def extra_tkl(steps, cpu_budget):
    """BASELINE.json configs[2]: TKL, D = 2048, dim = 300, Q = 20, fp32; 256 documents of U{50..2048} tokens and
    queries of U{3..20} tokens, pre-contextualised packed chunks resident in HBM -> scores."""
    import torch
    from matchmaker_amd import ops
    from matchmaker_amd.tkl import TKL_sigir20, chunk_documents
    dev = torch.device("cuda", torch.cuda.current_device())
    B, Qt, Dt, Et = 256, 20, 2048, 300
    g = torch.Generator(device=dev).manual_seed(3003)
    m = TKL_sigir20(Et, MU, [0.1] * 11, 10, 2, 300, 2000, True, True, "embedding").to(dev).eval()     # tkl.yaml
    q = torch.randn(B, Qt, Et, generator=g, device=dev)
    d = torch.randn(B, Dt, Et, generator=g, device=dev)
    d_len = torch.randint(50, Dt + 1, (B,), generator=g, device=dev)
    q_len = torch.randint(3, Qt + 1, (B,), generator=g, device=dev)
    qm = (torch.arange(Qt, device=dev)[None] < q_len[:, None]).float()
    dm = (torch.arange(Dt, device=dev)[None] < d_len[:, None]).float()
    q_ctx = q * qm.unsqueeze(-1)
    chunks, cmask, slot, C = chunk_documents(d * dm.unsqueeze(-1), dm)       # stands in for the contextualised chunks
    params = m.pack_params()
    P = chunks.shape[0]
    fn = lambda: ops.tkl_score(q_ctx, chunks, cmask, slot, qm, params, B, C, 11, "embedding")
    ms = gpu_time_ms(fn, steps)
    by = P * 50 * Et * 4 + B * Qt * Et * 4 + P * 50 * 4 + 4 * B
    gbs = by / (ms * 1e-3) / 1e9
    out = {"workload": f"TKL scoring (sigir20_tkl.py:180-286), {B} documents x D={Dt} (lengths U{{50..{Dt}}}: {P} packed chunks "
                       f"of 50 tokens), dim={Et}, Q={Qt} (lengths U{{3..{Qt}}}), embedding saturation",
           "dtype": "fp32 (split-bf16 operands: x = hi + lo, 4 bf16 MFMAs, fp32 accumulation)", "ms": ms, "docs_per_s": B / (ms * 1e-3), "algorithmic_bytes": by,
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS},
           "kernel": "tkl_prep_kernel + tkl_stage1_run_kernel<cos> + tkl_window_kernel<cos> + tkl_region_kernel (whole mm_tkl_fwd call)",
           "profile": "profiles/r03_tkl_pmc.json, profiles/r03_tkl_trace.json (full documents: profiles/r03_tklfull_pmc.json)"}
    try:
        if not LEAN:
            out["exact_f32_mfma"] = tkl_exact_f32_subprocess()
    except Exception as e:
        out["exact_f32_mfma"] = {"error": repr(e)}
    try:      # the same length distributions at four times the batch: how much of the 256-document figure is fixed cost
        if LEAN:
            raise RuntimeError("skipped (--lean)")
        B4 = 4 * B
        g4 = torch.Generator(device=dev).manual_seed(3004)
        d4 = torch.randn(B4, Dt, Et, generator=g4, device=dev)
        q4 = torch.randn(B4, Qt, Et, generator=g4, device=dev)
        dl4 = torch.randint(50, Dt + 1, (B4,), generator=g4, device=dev)
        ql4 = torch.randint(3, Qt + 1, (B4,), generator=g4, device=dev)
        qm4 = (torch.arange(Qt, device=dev)[None] < ql4[:, None]).float()
        dm4 = (torch.arange(Dt, device=dev)[None] < dl4[:, None]).float()
        ch4, cm4, sl4, C4 = chunk_documents(d4 * dm4.unsqueeze(-1), dm4)
        del d4
        qc4 = q4 * qm4.unsqueeze(-1)
        ms4 = gpu_time_ms(lambda: ops.tkl_score(qc4, ch4, cm4, sl4, qm4, params, B4, C4, 11, "embedding"), steps)
        by4 = ch4.shape[0] * 50 * Et * 4 + B4 * Qt * Et * 4 + ch4.shape[0] * 50 * 4 + 4 * B4
        out["batch_1024_documents"] = {"ms": ms4, "docs_per_s": B4 / (ms4 * 1e-3), "algorithmic_bytes": by4,
                                       "frac": by4 / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del ch4, cm4, sl4, qc4, q4
        torch.cuda.empty_cache()
    except Exception as e:
        out["batch_1024_documents"] = {"error": repr(e)}
    if cpu_budget > 0:
        from oracle import torch_port as TP
        n = 4
        keep = (slot.long() // C) < n
        packed = torch.zeros(n * C, dtype=torch.bool)
        packed[slot[keep].long().cpu()] = True
        centre, cm = chunks[keep][:, 5:-5].cpu().contiguous(), cmask[keep][:, 5:-5].cpu().float().contiguous()
        prm = {"mu": m.mu, "sigma": m.sigma, "dense_w": m.dense.weight, "sat_w1": m.saturation_linear.weight,
               "sat_b1": m.saturation_linear.bias, "sat_w2": m.saturation_linear2.weight, "sat_b2": m.saturation_linear2.bias,
               "sat_w3": m.saturation_linear3.weight, "sat_b3": m.saturation_linear3.bias, "ln_w": m.sat_normer.weight,
               "ln_b": m.sat_normer.bias, "emb_reduce_w": m.sat_emb_reduce1.weight, "kernel_mult0": m.kernel_mult[0],
               "chunk_scoring": m.chunk_scoring}
        prm = {k: v.detach().float().cpu().reshape(-1) for k, v in prm.items()}
        qc, qmc = q_ctx[:n].cpu(), qm[:n].cpu()
        mc = TKL_sigir20(Et, MU, [0.1] * 11, 10, 2, 300, 2000, True, True, "embedding").eval()
        dc, dmc = d[:n].cpu() * dm[:n].cpu().unsqueeze(-1), dm[:n].cpu()

        def scoring():
            with torch.no_grad():
                TP.tkl_scoring(qc, centre, cm, packed, n, qmc, prm, "embedding")

        def full():
            with torch.no_grad():
                qx, _ = mc.forward_representation(qc, qmc, mc.positional_features_q[:, :Qt])
                ch, chm, sl, Cc = chunk_documents(dc, dmc)
                cx, _ = mc.forward_representation(ch, chm, mc.positional_features_d[:, :50])
                pk = torch.zeros(n * Cc, dtype=torch.bool)
                pk[sl.long()] = True
                TP.tkl_scoring(qx, cx[:, 5:-5].contiguous(), chm[:, 5:-5].float().contiguous(), pk, n, qmc, prm, "embedding")
        (ra, na, ta), (r1, n1, t1), threads = both_thread_settings(lambda: scoring, cpu_budget, n)
        (fa, _, _), (f1, _, _), _ = both_thread_settings(lambda: full, cpu_budget, n)
        out["cpu_baseline"] = {"value": ra, "unit": "docs/s", "cores": threads, "kind": "port", "single_thread_value": r1,
                               "full_forward_value": fa, "full_forward_single_thread_value": f1,
                               "sample": f"the first {n} documents of the workload per call, {na} timed calls in {ta:.1f} s "
                                         f"({threads} threads) / {n1} in {t1:.1f} s (1 thread); scoring only = "
                                         f"oracle/torch_port.tkl_scoring (sigir20_tkl.py:180-286); full forward adds chunking + the "
                                         f"drop-in's PyTorch chunk Transformer (tkl.yaml)"}
    return out


def tkl_exact_f32_subprocess():
    """TKL with stage 1 on the exact-f32 MFMA kernel (MM_KP_F32MFMA=1 is read once per process -> a child process):
    the reference's own operand precision (tkl.yaml use_fp16: False) beside the split-bf16 default."""
    env = dict(os.environ, MM_KP_F32MFMA="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_tkl.py"), "--steps", "5"],
                       capture_output=True, text=True, env=env, timeout=300)
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            j = json.loads(line)
            return {"kernel": "stage 1 = kernel_pool_stream_kernel<TKL> (v_mfma_f32_32x32x2_f32, exact fp32 operands), stages 2-3 unchanged",
                    "ms": j["ms"], "docs_per_s": j["docs_per_s"], "GBps": j["GBps_algorithmic"],
                    "frac": j["GBps_algorithmic"] / HBM_PEAK_GBS}
    return {"error": (r.stderr or r.stdout)[-300:]}


def extra_maxsim_fp32(steps, cpu_budget):
    """The "fp32 variant" of BASELINE.json configs[1]: ColBERT with use_fp16: False (colbert.py:60 without autocast) —
    fp32 token vectors, Q=32 / D=180 / dim=128, 1000 candidates per query, on the three-term split-bf16 kernel."""
    import torch
    from matchmaker_amd import ops, synth
    dev = torch.device("cuda", torch.cuda.current_device())
    nq = 64
    q, d, q_len, d_len = synth.colbert_batch(nq, CANDS, Q, D, E, torch.float32, dev, seed=3232, lengths="full")
    B = nq * CANDS
    fn = lambda: ops.maxsim(q, d, q_len, d_len, pairs_per_query=CANDS)
    ms = gpu_time_ms(fn, steps)
    by = B * D * E * 4 + nq * Q * E * 4 + 4 * (B + nq) + 4 * B
    gbs = by / (ms * 1e-3) / 1e9
    out = {"workload": f"ColBERT MaxSim, fp32 vectors (use_fp16: False), {nq} queries x {CANDS} candidates, Q={Q}/D={D}/dim={E}, "
                       f"shared query tile, int32 lengths, all positions real",
           "dtype": "fp32 (three-term split-bf16 operands x = hi + lo + c, 6 bf16 MFMAs per K step, fp32 accumulation)",
           "ms": ms, "pairs_per_s": B / (ms * 1e-3), "algorithmic_bytes": by,
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS},
           "kernel": "kernel_pool_split128_kernel<MX> (csrc/kernel_pool128.hip)", "profile": "profiles/r03_maxsim_fp32_trace.json"}
    if cpu_budget > 0:
        from oracle import torch_port as TP
        n = 1000
        qc = q[:1].cpu().expand(n, -1, -1).contiguous()
        dc = d[:n].cpu()
        qm, dm = torch.ones(n, Q, dtype=torch.long), torch.ones(n, D, dtype=torch.long)

        def run():
            with torch.no_grad():
                TP.maxsim_forward(qc, dc, qm, dm)
        (ra, na, ta), (r1, n1, t1), threads = both_thread_settings(lambda: run, cpu_budget, n)
        out["cpu_baseline"] = {"value": ra, "unit": "pairs/s", "cores": threads, "kind": "port", "single_thread_value": r1,
                               "sample": f"{n}-pair forward calls (oracle/torch_port.maxsim_forward, fp32), {na} calls in {ta:.1f} s"}
    return out


def extra_eval_batch(steps, cpu_budget):
    """eval.py-sized calls: `batch_size_eval: 512` pairs per model.forward (config/train/defaults.yaml:115, eval.py:108),
    pair-per-row, in the layouts the reference hands over.  Per shape: device time per call (HIP events), wall time per
    call completed (back-to-back calls, one sync at the end), host time per call ISSUED (the Python + ctypes + launch
    path, GPU running behind), and the HBM fraction of the completed rate.  The calls rotate through enough distinct
    batches to exceed the 256 MB Infinity Cache, so every call streams from HBM as in a real evaluation run."""
    import torch
    from matchmaker_amd import ops
    from matchmaker_amd.colbert import ColBERT
    dev = torch.device("cuda", torch.cuda.current_device())
    Bc = 512
    g = torch.Generator(device=dev).manual_seed(512)
    res = {}

    def run(name, n_batches, make, call, bytes_per_call, n_calls=400):
        batches = [make() for _ in range(n_batches)]
        for b in batches:
            call(b)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
        for i, (a, b) in enumerate(ev):
            a.record(); call(batches[i % n_batches]); b.record()
        torch.cuda.synchronize()
        dev_us = 1e3 * sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]
        t0 = time.perf_counter()
        for i in range(n_calls):
            call(batches[i % n_batches])
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_done = time.perf_counter() - t0
        us_done, us_host = 1e6 * t_done / n_calls, 1e6 * t_issue / n_calls
        gbs = bytes_per_call / (us_done * 1e-6) / 1e9
        gbs_dev = bytes_per_call / (dev_us * 1e-6) / 1e9
        res[name] = {"pairs_per_call": Bc, "distinct_batches": n_batches, "bytes_per_call": bytes_per_call,
                     "us_per_call_device": dev_us, "us_per_call_completed": us_done, "us_per_call_host_issue": us_host,
                     "pairs_per_s": Bc / (us_done * 1e-6),
                     "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                  "frac_device_time_only": gbs_dev / HBM_PEAK_GBS}}
        del batches
        torch.cuda.empty_cache()

    def colbert_batch_maker(Qc, Dc, Ec, dt):
        def make():
            q = (torch.randn(Bc, Qc, Ec, generator=g, device=dev) / Ec ** 0.5).to(dt)
            d = (torch.randn(Bc, Dc, Ec, generator=g, device=dev) / Ec ** 0.5).to(dt)
            return q, d, torch.ones(Bc, Qc, dtype=torch.long, device=dev), torch.ones(Bc, Dc, dtype=torch.long, device=dev)
        return make

    run("colbert_dim128_bf16", 16, colbert_batch_maker(Q, D, E, torch.bfloat16), lambda b: ColBERT._score(*b),
        Bc * ((D + Q) * E * 2 + 8 * (D + Q) + 4))
    run("colbert_published_dim768_fp16", 3, colbert_batch_maker(38, 200, 768, torch.float16), lambda b: ColBERT._score(*b),
        Bc * ((200 + 38) * 768 * 2 + 8 * (200 + 38) + 4))
    prm = [torch.tensor(MU, device=dev), torch.full((11,), 0.1, device=dev), torch.ones(11, device=dev),
           torch.linspace(-0.014, 0.014, 11, device=dev)]

    def tk_make():
        return (torch.randn(Bc, 20, 300, generator=g, device=dev), torch.randn(Bc, 200, 300, generator=g, device=dev),
                torch.ones(Bc, 20, device=dev), torch.ones(Bc, 200, device=dev))
    run("tk_dim300_fp32", 4, tk_make, lambda b: ops.kernel_pool(b[0], b[1], b[2], b[3], *prm, pairs_per_query=1),
        Bc * ((200 + 20) * 300 * 4 + 4 * (200 + 20) + 4))
    return {"workload": "512 pairs per call (defaults.yaml:115 batch_size_eval), pair-per-row, int64 HF masks (ColBERT) / float masks (TK); "
                        "config-2 shapes, the published checkpoint's shapes (Q=38 / D=200 / dim=768 fp16), TK Q=20 / D=200 / dim=300",
            "shapes": res}


def extra_published_checkpoint(steps, cpu_budget):
    """The configuration of the reference's PUBLISHED ColBERT checkpoint (config/huggingface_modelhub/sebastian-hofstaetter/
    colbert-distilbert-margin_mse-T2-msmarco.yaml: colbert_compression_dim 768, max_query_length 30 +
    query_augment_mask_number 8 -> Q = 38, max_doc_length 200, use_fp16) in eval.py's batch layout (ColBERT._score)."""
    import torch
    from matchmaker_amd.colbert import ColBERT
    dev = torch.device("cuda", torch.cuda.current_device())
    n, Qp, Dp, Ep = 16000, 38, 200, 768
    g = torch.Generator(device=dev).manual_seed(768)
    qp = (torch.randn(n, Qp, Ep, generator=g, device=dev) / Ep ** 0.5).half()
    dp = (torch.randn(n, Dp, Ep, generator=g, device=dev) / Ep ** 0.5).half()
    qm = torch.ones(n, Qp, dtype=torch.long, device=dev)
    dm = torch.ones(n, Dp, dtype=torch.long, device=dev)
    ms = gpu_time_ms(lambda: ColBERT._score(qp, dp, qm, dm), steps)
    by = n * ((Dp + Qp) * Ep * 2 + 8 * (Dp + Qp) + 4)
    gbs = by / (ms * 1e-3) / 1e9
    out = {"workload": f"{n} pairs in the reference's batch layout, Q={Qp} (30 + 8 [MASK]) / D={Dp} / dim={Ep}, fp16, int64 HF masks",
           "dtype": "f16", "ms": ms, "pairs_per_s": n / (ms * 1e-3), "algorithmic_bytes": by,
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS},
           "kernel": "pack_mask_kernel + maxsim_stream_kernel with two query tiles (NSL = 6, NQT = 2)",
           "profile": "profiles/r03_published_checkpoint_trace.json"}
    if cpu_budget > 0:
        from oracle import torch_port as TP
        m = 256
        qc, dc, qmc, dmc = qp[:m].float().cpu(), dp[:m].float().cpu(), qm[:m].cpu(), dm[:m].cpu()

        def run():
            with torch.no_grad():
                TP.maxsim_forward(qc, dc, qmc, dmc)
        (ra, na, ta), (r1, n1, t1), threads = both_thread_settings(lambda: run, cpu_budget, m)
        out["cpu_baseline"] = {"value": ra, "unit": "pairs/s", "cores": threads, "kind": "port",
                               "sample": f"{m}-pair batches (oracle/torch_port.maxsim_forward, fp32), {na} calls in {ta:.1f} s",
                               "one_thread": {"value": r1, "cores": 1}}
    return out


def extra_all_pairs(steps, cpu_budget):
    """forward_inbatch_aggregation (colbert.py:154-162) at a teacher batch far beyond the reference's 32 x 32:
    1024 queries x 1024 documents, Q=32 / D=180 / dim=128, bf16, the documents' own masks.  MFMA-shaped: every
    (query, document) pair is a 32 x 180 x 128 product and the documents (47 MB) stay in the Infinity Cache."""
    import torch
    from matchmaker_amd import ops
    dev = torch.device("cuda", torch.cuda.current_device())
    Bq = Bd = 1024
    g = torch.Generator(device=dev).manual_seed(1414)
    q = (torch.randn(Bq, Q, E, generator=g, device=dev) / E ** 0.5).bfloat16()
    d = (torch.randn(Bd, D, E, generator=g, device=dev) / E ** 0.5).bfloat16()
    qm = torch.ones(Bq, Q, dtype=torch.long, device=dev)
    dm = torch.ones(Bd, D, dtype=torch.long, device=dev)     # every position real: the flop count below is what runs
    fn = lambda: ops.maxsim_inbatch(q, qm, d, dm, bug_compatible=False)
    ms = gpu_time_ms(fn, steps)
    flop = 2.0 * Bq * Bd * Q * D * E
    t = ms * 1e-3
    out = {"workload": f"all-pairs MaxSim (colbert.py:154-162), {Bq} queries x {Bd} documents, Q={Q}/D={D}/dim={E}, bf16, "
                       f"all positions real, int64 masks",
           "dtype": "bf16", "ms": ms, "pairs_per_s": Bq * Bd / t, "flop": flop,
           "roofline": {"bound": "mfma", "achieved": flop / t / 1e12, "peak": MFMA_PEAK_16BIT / 1e12, "unit": "TFLOP/s",
                        "frac": flop / t / MFMA_PEAK_16BIT},
           "kernel": "maxsim_allpairs_wg_kernel: 4 wavefronts x 4 queries share one document ring", "profile": "profiles/r03_all_pairs_pmc.json"}
    if not LEAN:
        t0 = gpu_time_ms(lambda: ops.maxsim_inbatch(q[:32], qm[:32], d[:32], dm[:32], bug_compatible=True), steps)
        out["reference_batch_32x32"] = {"ms": t0, "note": "dynamic_teacher.py:245-276 calls it with batch_size_train = 32, bug-compatible masks"}
        # the vendor GEMM on this box: the same product with the [Bq Q, Bd D] score matrix WRITTEN (what colbert.py:154 does
        # before its max / sum passes) and a square 8192^3 (the matrix rate the board's power budget allows)
        out["vendor_gemm"] = vendor_gemm_tflops({"same_product_score_matrix_written_bf16": (Bq * Q, Bd * D, E, torch.bfloat16),
                                                 "square_8192_bf16": (8192, 8192, 8192, torch.bfloat16)})
        out["vendor_gemm"]["this_kernel_over_same_product"] = (flop / t / 1e12) / out["vendor_gemm"]["same_product_score_matrix_written_bf16"]["tflops"]
        out["vendor_gemm"]["this_kernel_over_square_8192"] = (flop / t / 1e12) / out["vendor_gemm"]["square_8192_bf16"]["tflops"]
    if cpu_budget > 0:
        from oracle import torch_port as TP
        n = 64
        qc, dc, qmc, dmc = q[:n].float().cpu(), d[:n].float().cpu(), qm[:n].cpu(), dm[:n].cpu()

        def run():
            with torch.no_grad():
                TP.maxsim_inbatch(qc, qmc, dc, dmc, bug_compatible=False)
        (ra, na, ta), (r1, n1, t1), threads = both_thread_settings(lambda: run, cpu_budget, n * n)
        out["cpu_baseline"] = {"value": ra, "unit": "pairs/s", "cores": threads, "kind": "port",
                               "sample": f"{n} x {n} all-pairs blocks (oracle/torch_port.py, fp32), {na} calls in {ta:.1f} s",
                               "one_thread": {"value": r1, "cores": 1}}
    return out


def extra_dot_topk(steps, cpu_budget):
    """BASELINE.json configs[4], ONE rank's shard: 8,841,823 / 8 passages x dim 768 fp16 against all 6,980 queries,
    exact top-1000 (faiss IndexFlatIP semantics)."""
    import torch
    from matchmaker_amd import ops
    dev = torch.device("cuda", torch.cuda.current_device())
    N, Ed, nq, k = 1105228, 768, 6980, 1000
    g = torch.Generator(device=dev).manual_seed(5005)
    c = torch.empty((N, Ed), dtype=torch.float16, device=dev)
    for s in range(0, N, 1 << 18):
        n = min(1 << 18, N - s)
        c[s:s + n] = torch.randn(n, Ed, generator=g, device=dev).half()
    q = torch.randn(nq, Ed, generator=g, device=dev).half()
    ops.dot_topk(q, c, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ops.dot_topk(q, c, k)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / steps
    flop = 2.0 * nq * N * Ed
    out = {"workload": f"brute-force inner-product top-{k} (faiss_indices.py:49-74, dense_retrieval.py:391): one rank's shard "
                       f"of configs[4] = {N} passages x dim {Ed} fp16, {nq} queries",
           "dtype": "fp16 (fp32 accumulation)", "ms": t * 1e3, "queries_per_s": nq / t, "flop": flop,
           "roofline": {"bound": "mfma", "achieved": flop / t / 1e12, "peak": MFMA_PEAK_16BIT / 1e12, "unit": "TFLOP/s",
                        "frac": flop / t / MFMA_PEAK_16BIT},
           "kernel": "dot_stream_kernel (sample + filter) + sample_tau_kernel + topk_rows_kernel (whole mm_dot_topk_fwd call, wall clock)",
           "profile": "profiles/r03_dot_topk_pmc.json, profiles/r03_dot_topk_trace.json"}
    if not LEAN:
        # what the vendor GEMM reaches on THIS box: the same product (one 262,144-passage slice, fp16 scores written, no
        # top-k) and a square 8192^3 — the matrix rate the board's power budget allows, next to the 2.5 PFLOP/s nominal peak
        out["vendor_gemm"] = vendor_gemm_tflops({"same_product_262144_passages_fp16": (nq, 1 << 18, Ed, torch.float16),
                                                 "square_8192_bf16": (8192, 8192, 8192, torch.bfloat16)})
        out["vendor_gemm"]["this_call_over_same_product"] = (flop / t / 1e12) / out["vendor_gemm"]["same_product_262144_passages_fp16"]["tflops"]
        out["vendor_gemm"]["this_call_over_square_8192"] = (flop / t / 1e12) / out["vendor_gemm"]["square_8192_bf16"]["tflops"]
    if cpu_budget > 0:
        nqc, nc = 64, 1 << 16
        qc, cc = q[:nqc].float().cpu(), c[:nc].float().cpu()

        def search():
            with torch.no_grad():
                torch.topk(qc @ cc.T, k, dim=1)
        (ra, na, ta), (r1, n1, t1), threads = both_thread_settings(lambda: search, cpu_budget, nqc * nc)
        out["cpu_baseline"] = {"value": ra, "unit": "query-passage inner products/s", "cores": threads, "kind": "port",
                               "single_thread_value": r1, "gpu_value_same_unit": nq * N / t,
                               "sample": f"{nqc} queries x {nc} passages per call (fp32 torch matmul + topk = IndexFlatIP.search's "
                                         f"definition; faiss is absent here), {na} calls in {ta:.1f} s ({threads} threads) / {n1} in "
                                         f"{t1:.1f} s (1 thread)"}
    return out


LEGS = (("eval_batch", extra_eval_batch), ("maxsim_fp32", extra_maxsim_fp32), ("published_checkpoint", extra_published_checkpoint),
        ("all_pairs", extra_all_pairs), ("tk", extra_tk), ("tkl", extra_tkl), ("dot_topk", extra_dot_topk))


# ------------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--queries", type=int, default=256,
                    help="queries per GPU per step (x1000 candidates; 256 -> 11.8 GB of bf16 token embeddings resident)")
    ap.add_argument("--lengths", default="full", choices=["full", "msmarco"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the drop-in-layout / TK / TKL / dot-top-k legs (N = 1 only)")
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"])
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"])
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the per-step all-gather even with one rank (exercises the "
                         "RCCL path on a single-GPU box)")
    ap.add_argument("--only", default=None,
                    choices=["headline", "dropin_forward", "published_checkpoint", "all_pairs", "tk", "tkl", "dot_topk",
                             "maxsim_fp32", "eval_batch"],
                    help="run ONE leg and print it alone (the command that is put under rocprofv3 --kernel-trace)")
    ap.add_argument("--lean", action="store_true",
                    help="with --only: the leg's main measurement alone (no sub-legs, no child processes): what gets profiled")
    ap.add_argument("--one-device", action="store_true",
                    help="every rank uses device 0 (RCCL test of the N > 1 path on a single-GPU box; RCCL may refuse it)")
    ap.add_argument("--dry", action="store_true",
                    help="launch + collective plumbing only (fabricated scores, value = null); for GPU-less machines")
    args = ap.parse_args()

    global LEAN
    LEAN = bool(args.lean)
    if args.only == "headline":
        args.no_extras = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the torch.distributed environment has WORLD_SIZE={world}")
    if args.device == "cpu" and not args.dry:
        raise SystemExit("--device cpu is only valid with --dry: there is no CPU scoring path")
    backend = args.backend or ("gloo" if args.device == "cpu" else "nccl")
    if args.device == "cuda":
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
        dev = torch.device("cuda", 0 if args.one_device else local_rank)
        torch.cuda.set_device(dev)
    else:
        dev = torch.device("cpu")
    if args.only and args.only != "headline":
        assert world == 1 and dev.type == "cuda", "--only legs are single-GPU measurements"
        from matchmaker_amd import _lib
        _lib.lib()
        cpu_b = 0.0 if args.no_cpu_baseline else 3.0
        legs = dict(LEGS)
        if args.only == "dropin_forward":
            from matchmaker_amd import synth
            q, d, q_len, d_len = synth.colbert_batch(args.queries, CANDS, Q, D, E, torch.bfloat16, dev, seed=4004, lengths=args.lengths)
            res = extra_dropin_forward(q, d, q_len, d_len, max(5, args.steps // 2))
        else:
            res = legs[args.only](3 if args.only == "dot_topk" else max(args.steps // 2, 5), cpu_b)
        print(json.dumps({"only": args.only, "result": res}), flush=True)
        return
    use_dist = world > 1 or args.force_dist
    if use_dist:
        # the GPU boxes export NCCL_DEBUG=VERSION and RCCL prints to STDOUT (its banner would follow the JSON line, once
        # per rank; WARN is noisier still): no RCCL logging unless MM_NCCL_DEBUG asks for it — the version is reported
        # in the line itself
        os.environ.pop("NCCL_DEBUG", None)
        if os.environ.get("MM_NCCL_DEBUG"):
            os.environ["NCCL_DEBUG"] = os.environ["MM_NCCL_DEBUG"]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        kw = {"device_id": dev} if dev.type == "cuda" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)

    nq = args.queries if not args.dry else min(args.queries, 2)
    B = nq * CANDS
    if args.dry:
        score_shard = lambda: torch.arange(B, dtype=torch.float32, device=dev) + rank * B     # rank-tagged, checkable
    else:
        from matchmaker_amd import ops, synth, _lib
        _lib.lib()   # fail loudly here if the HIP library is missing
        q, d, q_len, d_len = synth.colbert_batch(nq, CANDS, Q, D, E, torch.bfloat16, dev, seed=4004 + rank,
                                                 lengths=args.lengths)
        score_shard = lambda: ops.maxsim(q, d, q_len, d_len, pairs_per_query=CANDS)
    gathered = torch.empty(world * B, dtype=torch.float32, device=dev) if use_dist else None

    def step():
        s = score_shard()
        if use_dist:
            dist.all_gather_into_tensor(gathered, s)     # RCCL over xGMI: the ranking merge
        return s

    for _ in range(args.warmup):
        step()

    def barrier():
        sync()
        if use_dist:
            dist.barrier()
            sync()

    # kernel-only timing with HIP events on the launch stream (roofline numerator)
    use_ev = dev.type == "cuda"
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)] if use_ev else []
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if use_ev:
            ev[i][0].record()
        s = score_shard()
        if use_ev:
            ev[i][1].record()
        if use_dist:
            dist.all_gather_into_tensor(gathered, s)
    barrier()
    t = time.perf_counter() - t0
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev) if use_ev else None

    gather_matches = None
    if use_dist and not args.dry:      # the gathered tensor's slice of this rank must BE its local scores
        gather_matches = bool(torch.equal(gathered[rank * B:(rank + 1) * B], s))
    tt = torch.tensor([t], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    t = float(tt.item())
    gather_ok = None
    if use_dist and args.dry:
        want = torch.arange(world * B, dtype=torch.float32, device=dev)
        gather_ok = bool(torch.equal(gathered, want))

    if rank == 0:
        total_pairs = world * B * args.steps
        coll = None
        if use_dist:
            ver = None
            if backend == "nccl":
                try:
                    ver = ".".join(str(x) for x in torch.cuda.nccl.version())
                except Exception:
                    ver = "unknown"
            coll = {"backend": backend + (" (RCCL)" if backend == "nccl" else ""), "version": ver, "world_size": world,
                    "op": "all_gather_into_tensor of fp32 scores, once per step, inside the timed region",
                    "bytes_per_rank": 4 * B, "bytes_total": 4 * B * world,
                    "gathered_slice_equals_local_scores": gather_matches,
                    "devices": "all ranks on device 0 (--one-device)" if args.one_device else "one device per rank"}
        out = {
            "metric": "query-doc pairs scored/sec (ColBERT MaxSim, Q32/D180/dim128)",
            "value": None if args.dry else total_pairs / t, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * t / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "world_size": world, "collective": coll,
            "config": {"workload": f"BASELINE.json configs[1]: ColBERT MaxSim re-rank, dim=128, Q=32/D=180, "
                                   f"1000 candidates/query, bf16; {nq} queries x 1000 candidates resident per GPU "
                                   f"per step; doc lengths = {args.lengths}",
                       "queries_per_gpu": nq, "cands_per_query": CANDS, "Q": Q, "D": D, "E": E,
                       "parallelism": f"query-sharded x{world}" + (f", {backend} all-gather of scores" if world > 1 else "")},
        }
        if args.dry:
            out["dry"] = True
            out["note"] = "launch/collective plumbing check only: scores fabricated, nothing measured"
            out["all_gather_verified"] = gather_ok
        else:
            # the timed scores themselves, checked in-process against the oracle on one sampled query (checker only)
            try:
                import numpy as np
                from oracle import np_oracle as O
                from matchmaker_amd import synth as _synth
                qi = nq // 2
                got = s[qi * CANDS:(qi + 1) * CANDS].float().cpu().numpy()
                ref = O.maxsim_paired(np.repeat(q[qi:qi + 1].float().cpu().numpy(), CANDS, axis=0),
                                      d[qi * CANDS:(qi + 1) * CANDS].float().cpu().numpy(),
                                      _synth.len_to_mask(q_len[qi:qi + 1].cpu(), Q).numpy().repeat(CANDS, axis=0),
                                      _synth.len_to_mask(d_len[qi * CANDS:(qi + 1) * CANDS].cpu(), D).numpy())
                err = float(np.abs(got - ref).max())
                out["self_check"] = {"query": qi, "pairs": CANDS, "max_abs_err_vs_oracle": err, "tolerance": 1e-2,
                                     "ok": bool(err <= 1e-2),
                                     "same_order_as_oracle": bool((np.argsort(-got, kind="stable") == np.argsort(-ref, kind="stable")).mean() > 0.99)}
                if err > 1e-2:
                    out["value"] = None      # a fast kernel whose results differ from the reference's is not a measurement
            except Exception as e:
                out["self_check"] = {"error": repr(e)}
            ab = algorithmic_bytes(nq, CANDS)
            achieved = ab / (kern_ms * 1e-3) / 1e9
            flop = 2.0 * B * Q * D * E
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                               "kernel": "maxsim_stream_kernel", "kernel_ms": kern_ms, "algorithmic_bytes": ab,
                               "mfma_util": {"flop_per_launch": flop, "tflops": flop / (kern_ms * 1e-3) / 1e12,
                                             "frac_of_dense_bf16_peak": flop / (kern_ms * 1e-3) / MFMA_PEAK_16BIT,
                                             "peak_tflops": MFMA_PEAK_16BIT / 1e12}}
            tr = profile_summary("*maxsim*pmc*.json", "maxsim_stream_kernel", "_hbm_traffic_bytes_per_dispatch")
            if tr is not None and tr[2].get("queries") == nq and tr[2].get("cands") == CANDS and tr[2].get("lengths") == args.lengths:
                out["roofline"]["traffic"] = tr[0]
                out["roofline"]["traffic_source"] = f"profiles/{tr[1]} (rocprofv3 PMC passes of this workload)"
            busy = profile_summary("*maxsim*pmc*.json", "maxsim_stream_kernel", "_mfma_busy_frac")
            if busy is not None:
                out["roofline"]["mfma_util"]["pmc_busy_frac"] = busy[0]
                out["roofline"]["mfma_util"]["pmc_source"] = f"profiles/{busy[1]} (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), tools/summarize_rocprof.py)"
            if world == 1:      # CPU leg and extras are N = 1 figures; ranks > 0 would idle through them
                if not args.no_cpu_baseline:
                    nsamp = min(nq, 24)
                    out["cpu_baseline"] = cpu_baseline(q[:nsamp].cpu(), d[:nsamp * CANDS].cpu(), q_len[:nsamp].cpu(),
                                                       d_len[:nsamp * CANDS].cpu(), CANDS)
                if not args.no_extras:
                    extra = {}
                    cpu_b = 0.0 if args.no_cpu_baseline else 3.0
                    try:
                        extra["sustained"] = extra_sustained(score_shard, B)
                    except Exception as e:
                        extra["sustained"] = {"error": repr(e)}
                    try:
                        extra["dropin_forward"] = extra_dropin_forward(q, d, q_len, d_len, max(5, args.steps // 2))
                    except Exception as e:      # an extra must never take the headline line down with it
                        extra["dropin_forward"] = {"error": repr(e)}
                    del q, d
                    torch.cuda.empty_cache()
                    for name, fn in LEGS:
                        try:
                            extra[name] = fn(3 if name == "dot_topk" else 10, cpu_b)
                        except Exception as e:
                            extra[name] = {"error": repr(e)}
                        torch.cuda.empty_cache()
                    out["extra"] = extra
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
