#!/usr/bin/env python
"""bench.py — query-doc pairs scored / second, ColBERT MaxSim (Q32 / D180 / dim128, bf16,
1000 candidates per query; BASELINE.json config 2), on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path (mm_maxsim_fwd) over one resident batch of
`--queries` x 1000 synthetic (query, candidate) pairs per GPU.  Inputs are generated on the device
before the timed region (resident in HBM).  Every rank scores its own shard of queries (weak
scaling, no data-path collective); with N > 1 one RCCL all-gather of the fp32 scores per step is
part of the timed region (the ranking merge of SURVEY.md §8e).

Prints ONE JSON line (rank 0).  `roofline`: algorithmic bytes per launch / average
kernel time measured with HIP events on the launch stream; `cpu_baseline`: the torch CPU port of the
reference's ops (oracle/torch_port.py) timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

Q, D, E, CANDS = 32, 180, 128, 1000
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes(n_queries: int, cands: int) -> int:
    """B*D*E*s + Nq*Q*E*s + 4*(B + Nq) + 4*B  (shared-Q layout, int32 lengths)."""
    B = n_queries * cands
    return B * D * E * 2 + n_queries * Q * E * 2 + 4 * (B + n_queries) + 4 * B


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
    return {"value": rate, "unit": "pairs/s", "cores": threads, "kind": "port",
            "single_thread_value": rate1,
            "sample": f"{n} forward calls over whole queries ({nq} distinct) x {cands} candidates = {pairs} pairs of the "
                      f"bench workload in {t_total:.1f} s with {threads} torch threads (host has "
                      f"{os.cpu_count()} logical CPUs); fp32 torch CPU port of colbert.py:68-75 "
                      f"(oracle/torch_port.py: bmm, masked assign, max, sum); single-thread leg "
                      f"(OMP_NUM_THREADS=1 as in train.py:12): {pairs1} pairs in {t1:.1f} s"}


def measured_traffic(nq, cands, lengths):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of THIS workload
    (profiles/*_maxsim_pmc.json, written by tools/profile_maxsim.sh + tools/summarize_rocprof.py:
    separate --pmc passes, FETCH_SIZE KiB x1024 x2 (gfx950 half-count) + WRITE_SIZE KiB x1024)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*maxsim*pmc*.json"))):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        wl = j.get("workload", {})
        if wl.get("queries") == nq and wl.get("cands") == cands and wl.get("lengths") == lengths:
            for k, c in j.get("pmc", {}).items():
                if "maxsim_stream_kernel" in k and "_hbm_traffic_bytes_per_dispatch" in c:
                    best = (c["_hbm_traffic_bytes_per_dispatch"]["total"], os.path.basename(f))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--queries", type=int, default=256,
                    help="queries per GPU per step (x1000 candidates; 256 -> 11.8 GB of bf16 token embeddings resident)")
    ap.add_argument("--lengths", default="full", choices=["full", "msmarco"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from matchmaker_amd import ops, synth, _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    _lib.lib()   # fail loudly here if the HIP library is missing

    nq = args.queries
    q, d, q_len, d_len = synth.colbert_batch(nq, CANDS, Q, D, E, torch.bfloat16, dev, seed=4004 + rank,
                                             lengths=args.lengths)
    B = nq * CANDS
    gathered = torch.empty(world * B, dtype=torch.float32, device=dev) if world > 1 else None

    def step():
        s = ops.maxsim(q, d, q_len, d_len, pairs_per_query=CANDS)
        if world > 1:
            dist.all_gather_into_tensor(gathered, s)     # RCCL over xGMI: the ranking merge
        return s

    for _ in range(args.warmup):
        step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # kernel-only timing with HIP events on the launch stream (roofline numerator)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        s = ops.maxsim(q, d, q_len, d_len, pairs_per_query=CANDS)
        b.record()
        if world > 1:
            dist.all_gather_into_tensor(gathered, s)
    barrier()
    t = time.perf_counter() - t0
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)

    tt = torch.tensor([t], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    t = float(tt.item())

    if rank == 0:
        total_pairs = world * B * args.steps
        ab = algorithmic_bytes(nq, CANDS)
        achieved = ab / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "query-doc pairs scored/sec (ColBERT MaxSim, Q32/D180/dim128)",
            "value": total_pairs / t, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * t / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[1]: ColBERT MaxSim re-rank, dim=128, Q=32/D=180, "
                                   f"1000 candidates/query, bf16; {nq} queries x 1000 candidates resident per GPU "
                                   f"per step; doc lengths = {args.lengths}",
                       "queries_per_gpu": nq, "cands_per_query": CANDS, "Q": Q, "D": D, "E": E,
                       "parallelism": f"query-sharded x{world}" + (", RCCL all-gather of scores" if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "maxsim_stream_kernel", "kernel_ms": kern_ms, "algorithmic_bytes": ab},
        }
        tr = measured_traffic(nq, CANDS, args.lengths)
        if tr is not None:
            out["roofline"]["traffic"] = tr[0]
            out["roofline"]["traffic_source"] = f"profiles/{tr[1]} (rocprofv3 PMC passes of this workload)"
        if not args.no_cpu_baseline and world == 1:      # the CPU leg is an N = 1 figure; ranks > 0 would idle through it
            nsamp = min(nq, 24)
            out["cpu_baseline"] = cpu_baseline(q[:nsamp].cpu(), d[:nsamp * CANDS].cpu(), q_len[:nsamp].cpu(),
                                               d_len[:nsamp * CANDS].cpu(), CANDS)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
