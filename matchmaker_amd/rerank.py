"""The re-ranking evaluation loop of matchmaker/eval.py around a drop-in model — only the part of
`evaluate_model` that touches the model (eval.py:82-108: per batch autocast + move to the device +
`model.forward(query_tokens, doc_tokens, use_fp16=..., output_secondary_output=...)`; eval.py:161-203:
scores back to the CPU in one piece, unrolled into `{query_id: [(doc_id, score), ...]}`) and the ranking
rule that follows it (utils/core_metrics.py:502-511).

The reference's own script cannot be imported without allennlp (eval.py:11-12); with allennlp installed,
`patch_matchmaker()` + the unchanged script is the route.  This caller exists so the
"eval.py calls forward() unchanged" path is exercised on the GPU box as eval.py drives it: pair-per-row
batches, HF int64 attention masks, fp16 autocast, one `.cpu()` per batch.
"""
from typing import Dict, Iterable, List, Tuple

import torch


def _to_device(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device, non_blocking=True)
    if isinstance(x, dict):
        return {k: _to_device(v, device) for k, v in x.items()}
    return x


def evaluate_batches(model, batches: Iterable[dict], use_fp16: bool = True, device=None,
                     output_secondary_output: bool = False) -> Dict[str, List[Tuple[str, float]]]:
    """batches: dicts with "query_tokens", "doc_tokens" (HF tokenizer dicts), "query_id", "doc_id" (lists),
    the fields eval.py reads.  Returns the unrolled results of eval.py:189-203."""
    if device is None:
        import itertools
        t = next(itertools.chain(model.parameters(), model.buffers()), None)
        if t is None:
            raise ValueError("evaluate_batches: the model has no parameters or buffers; pass device=")
        device = t.device
    validation_results: Dict[str, List[Tuple[str, float]]] = {}
    with torch.no_grad():
        for batch_orig in batches:
            with torch.autocast("cuda", dtype=torch.float16, enabled=use_fp16):          # eval.py:83
                batch = _to_device({k: batch_orig[k] for k in ("query_tokens", "doc_tokens")}, device)   # :89
                output = model.forward(batch["query_tokens"], batch["doc_tokens"],
                                       output_secondary_output=output_secondary_output, use_fp16=use_fp16)   # :108
                if output_secondary_output:
                    output, _ = output                                                   # :136
                output = output.cpu()                                                    # :161 — in one piece
            for i, qid in enumerate(batch_orig["query_id"]):                             # :169-190
                validation_results.setdefault(qid, []).append((batch_orig["doc_id"][i], float(output[i])))
    return validation_results


def unrolled_to_ranked_result(unrolled: Dict[str, List[Tuple[str, float]]]) -> Dict[str, List[str]]:
    """utils/core_metrics.py:502-511: per query, `sorted(..., key=score, reverse=True)` — stable, so ties
    keep arrival order."""
    return {qid: [doc for doc, _ in sorted(rows, key=lambda x: x[1], reverse=True)] for qid, rows in unrolled.items()}
