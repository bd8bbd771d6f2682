#!/usr/bin/env python
"""MaxSim micro-benchmarks beyond the headline line: ragged MSMARCO-like lengths, the reference's
pair-per-row layout (query replicated per pair), HF int64 masks (packed on device), fp16, fp32 / E=768
(generic kernel), all-pairs.  Prints one JSON object per variant."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matchmaker_amd import ops, synth

dev = torch.device("cuda:0")
Q, D, E, C = 32, 180, 128, 1000


def timeit(fn, steps=10, warm=2):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]


def report(name, ms, pairs, bytes_):
    print(json.dumps({"variant": name, "ms": round(ms, 4), "Mpairs_per_s": round(pairs / ms / 1e3, 2),
                      "GBps": round(bytes_ / ms / 1e6, 1)}), flush=True)


nq = int(os.environ.get("NQ", 64))
which = sys.argv[1:] or ["full", "msmarco", "paired", "i64mask", "fp16", "fp32", "e768", "inbatch"]
q, d, q_len, d_len = synth.colbert_batch(nq, C, Q, D, E, torch.bfloat16, dev, lengths="full")
B = nq * C
if "full" in which:
    report("bf16 shared-Q full lengths", timeit(lambda: ops.maxsim(q, d, q_len, d_len, C)), B, B * D * E * 2)
if "msmarco" in which:
    g = torch.Generator(device=dev).manual_seed(7)
    dl = synth.msmarco_doc_lengths(B, D, g, dev)
    rd = int((((dl + 31) // 32) * 32).clamp(max=D).sum().item()) * E * 2
    report("bf16 shared-Q msmarco lengths (bytes = 32-token blocks actually read)",
           timeit(lambda: ops.maxsim(q, d, q_len, dl, C)), B, rd)
if "paired" in which:
    qp = q.repeat_interleave(C, 0)[: B // 4].contiguous()
    report("bf16 pair-per-row layout (reference batch layout, Q replicated)",
           timeit(lambda: ops.maxsim(qp, d[: B // 4], None, d_len[: B // 4], 1)), B // 4, (B // 4) * (D + Q) * E * 2)
if "dropin" in which:
    # exactly what ColBERT.forward hands over (eval.py:108 -> colbert.py:68-75): Q replicated per pair, int64 HF masks
    qp = q.repeat_interleave(C, 0).contiguous()
    dm = synth.len_to_mask(d_len, D, torch.int64)
    qm = synth.len_to_mask(q_len, Q, torch.int64).repeat_interleave(C, 0).contiguous()
    by = B * ((D + Q) * E * 2 + 8 * (D + Q) + 4)
    report("bf16 drop-in layout: pair-per-row Q + int64 masks (bytes incl. masks)",
           timeit(lambda: ops.maxsim(qp, d, qm, dm, 1)), B, by)
    report("bf16 pair-per-row Q + int32 lengths",
           timeit(lambda: ops.maxsim(qp, d, q_len.repeat_interleave(C), d_len, 1)), B, B * ((D + Q) * E * 2 + 12))
    del qp, dm, qm
if "dropin_msmarco" in which:
    g = torch.Generator(device=dev).manual_seed(7)
    dl = synth.msmarco_doc_lengths(B, D, g, dev)
    qp = q.repeat_interleave(C, 0).contiguous()
    dm = synth.len_to_mask(dl, D, torch.int64)
    qm = synth.len_to_mask(q_len, Q, torch.int64).repeat_interleave(C, 0).contiguous()
    rd = int((((dl + 31) // 32) * 32).clamp(max=D).sum().item()) * E * 2 + B * (Q * E * 2 + 8 * (D + Q) + 4)
    report("bf16 drop-in layout, msmarco document lengths (bytes = blocks actually read + query tiles + masks)",
           timeit(lambda: ops.maxsim(qp, d, qm, dm, 1)), B, rd)
    del qp, dm, qm
if "dropin768" in which:
    n = 16000
    q7 = torch.randn(n, 32, 768, device=dev).to(torch.float16)
    d7 = torch.randn(n, 200, 768, device=dev).to(torch.float16)
    qm7 = torch.ones(n, 32, dtype=torch.int64, device=dev); dm7 = torch.ones(n, 200, dtype=torch.int64, device=dev)
    report("fp16 drop-in layout at the reference's defaults (colbert_compression_dim 768, Q 32, D 200, autocast fp16)",
           timeit(lambda: ops.maxsim(q7, d7, qm7, dm7, 1)), n, n * ((200 + 32) * 768 * 2 + 8 * 232 + 4))
    del q7, d7, qm7, dm7
if "published768" in which:
    # the published ColBERT checkpoint's configuration (config/huggingface_modelhub/.../colbert-distilbert-margin_mse-T2-msmarco.yaml):
    # colbert_compression_dim 768, max_query_length 30 + query_augment_mask_number 8 -> Q = 38, D = 200, fp16
    n = 16000
    for QQ in (38, 40):
        q7 = torch.randn(n, QQ, 768, device=dev).to(torch.float16)
        d7 = torch.randn(n, 200, 768, device=dev).to(torch.float16)
        qm7 = torch.ones(n, QQ, dtype=torch.int64, device=dev); dm7 = torch.ones(n, 200, dtype=torch.int64, device=dev)
        report(f"fp16 drop-in layout, published ColBERT checkpoint config (dim 768, Q {QQ} = 30 + 8 [MASK], D 200)",
               timeit(lambda: ops.maxsim(q7, d7, qm7, dm7, 1)), n, n * ((200 + QQ) * 768 * 2 + 8 * (200 + QQ) + 4))
        qs = q7[: n // 1000].contiguous()
        report(f"fp16 shared query, same shapes (Q {QQ})",
               timeit(lambda: ops.maxsim(qs, d7, qm7[: n // 1000], dm7, 1000)), n, n * (200 * 768 * 2 + 8 * 200) )
        del q7, d7, qm7, dm7
if "longq128" in which:
    # [MASK]-augmented queries at the headline width: Q = 40, dim 128, D = 180, pair-per-row (the pair kernel holds one tile)
    n = 64000
    for QQ in (32, 40):
        q7 = torch.randn(n, QQ, 128, device=dev).to(torch.bfloat16)
        d7 = torch.randn(n, 180, 128, device=dev).to(torch.bfloat16)
        qm7 = torch.ones(n, QQ, dtype=torch.int64, device=dev); dm7 = torch.ones(n, 180, dtype=torch.int64, device=dev)
        report(f"bf16 drop-in layout, dim 128, Q {QQ}, D 180",
               timeit(lambda: ops.maxsim(q7, d7, qm7, dm7, 1)), n, n * ((180 + QQ) * 128 * 2 + 8 * (180 + QQ) + 4))
        del q7, d7, qm7, dm7
if "i64mask" in which:
    dm = synth.len_to_mask(d_len, D, torch.int64)
    qm = synth.len_to_mask(q_len, Q, torch.int64)
    report("bf16 shared-Q, HF int64 masks packed on device (bytes incl. masks)",
           timeit(lambda: ops.maxsim(q, d, qm, dm, C)), B, B * D * (E * 2 + 8))
if "fp16" in which:
    qh, dh = q.half(), d.half()
    report("fp16 shared-Q full lengths", timeit(lambda: ops.maxsim(qh, dh, q_len, d_len, C)), B, B * D * E * 2)
    del qh, dh
if "fp32" in which:
    n = B // 4
    qf, df = q[: nq // 4].float(), d[:n].float()
    report("fp32 shared-Q (split-bf16 streaming kernel)", timeit(lambda: ops.maxsim(qf, df, q_len[: nq // 4], d_len[:n], C)), n, n * D * E * 4)
    del qf, df
if "e768" in which:
    n = 8000
    q7 = torch.randn(8, Q, 768, device=dev).to(torch.bfloat16)
    d7 = torch.randn(n, D, 768, device=dev).to(torch.bfloat16)
    report("bf16 E=768 (reference default dim; generic kernel)", timeit(lambda: ops.maxsim(q7, d7, None, None, 1000)), n, n * D * 768 * 2)
    del q7, d7
if "inbatch" in which:
    qb, db = q[:32].contiguous(), d[:32].contiguous()
    qm = torch.ones(32, Q, dtype=torch.int64, device=dev); dm = torch.ones(32, D, dtype=torch.int64, device=dev)
    report("all-pairs 32x32 (dynamic teacher shape)", timeit(lambda: ops.maxsim_inbatch(qb, qm, db, dm, True)), 1024, 32 * D * E * 2)
if "fp32e768" in which:
    n = (nq // 8) * C
    q7 = torch.randn(max(nq // 8, 1), Q, 768, device=dev)
    d7 = torch.randn(n, D, 768, device=dev)
    report("fp32 E=768 (two-wave split-bf16 streaming kernel)", timeit(lambda: ops.maxsim(q7, d7, None, None, C)), n, n * D * 768 * 4)
