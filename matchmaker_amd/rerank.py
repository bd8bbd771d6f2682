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


class _GraphedForward:
    """One captured `model.forward` per batch shape (HIP graph): eval.py issues thousands of identical 512-pair steps
    (defaults.yaml:115), each of which costs ~10 us of Python + launches per kernel when issued eagerly — more than the
    4 us of HBM time a dim-128 ColBERT batch needs.  The tokenizer tensors of a batch are copied straight into the graph's
    static input buffers (the H2D copy eval.py:89 makes anyway) and the graph is replayed.
    Measured (bench.py extra.eval_batch.graph_replay): SLOWER than the eager path for dim-128 ColBERT batches (14.1 vs 10.5 us
    per call), so `graph=False` stays the default.  eval.py pads each batch to its longest sequence, so real runs see many
    (B, Lq, Ld) shapes: the cache is an LRU of at most `max_graphs` captures (each holds its static inputs and a private
    memory pool); shapes beyond it, and shapes whose capture fails, run eagerly."""

    def __init__(self, model, use_fp16, output_secondary_output, device, max_graphs: int = 16):
        from collections import OrderedDict
        self.model, self.use_fp16, self.sec, self.device = model, use_fp16, output_secondary_output, device
        self.entries = OrderedDict()
        self.failed = OrderedDict()      # shapes whose capture failed (bounded on its own: they do not push live graphs out)
        self.max_graphs = max_graphs
        self.eager_calls = 0

    def _run(self, static):
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.use_fp16):
            out = self.model.forward(static["query_tokens"], static["doc_tokens"],
                                     output_secondary_output=self.sec, use_fp16=self.use_fp16)
        return out[0] if self.sec else out

    def __call__(self, batch_orig):
        parts = {k: batch_orig[k] for k in ("query_tokens", "doc_tokens")}
        key = tuple((k, n, tuple(t.shape), t.dtype) for k in parts for n, t in sorted(parts[k].items()))
        if key in self.failed:                                 # a shape whose capture failed before: eager, every time
            self.eager_calls += 1
            return self._run(_to_device(parts, self.device))
        entry = self.entries.get(key, False)
        if entry is False:
            static = _to_device(parts, self.device)
            static = {k: {n: t.clone() for n, t in v.items()} for k, v in static.items()}
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):                      # warm-up outside capture (lazy initialisation, workspaces)
                eager_out = self._run(static)
            torch.cuda.current_stream(self.device).wait_stream(side)
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out = self._run(static)
            except RuntimeError as e:                          # not capturable (data-dependent shapes, a host sync): eager, every time
                if "out of memory" in str(e).lower():          # (an OOM is not a property of the shape: let the caller see it)
                    raise
                import warnings
                warnings.warn(f"rerank: HIP-graph capture failed for batch shape {key[0][2:]} ({str(e)[:120]}); this shape runs eagerly")
                self.failed[key] = True
                while len(self.failed) > 4 * self.max_graphs:
                    self.failed.popitem(last=False)
                self.eager_calls += 1
                torch.cuda.synchronize(self.device)            # leave no half-open capture state behind
                return self._run(static)                       # recomputed AFTER the failed capture (the warm-up result predates it)
            entry = self.entries[key] = (g, static, out)
            while len(self.entries) > self.max_graphs:         # least recently replayed shape goes (its pool is freed with it)
                self.entries.popitem(last=False)
        elif key in self.entries:
            self.entries.move_to_end(key)
        g, static, out = entry
        for k, v in parts.items():
            for n, t in v.items():
                static[k][n].copy_(t, non_blocking=True)
        g.replay()
        return out


def evaluate_batches(model, batches: Iterable[dict], use_fp16: bool = True, device=None,
                     output_secondary_output: bool = False, graph: bool = False, score_group: int = 1) -> Dict[str, List[Tuple[str, float]]]:
    """batches: dicts with "query_tokens", "doc_tokens" (HF tokenizer dicts), "query_id", "doc_id" (lists),
    the fields eval.py reads.  Returns the unrolled results of eval.py:189-203.
    score_group > 1 (models with forward_representation + score_batches: the ColBERT drop-in): the encoder runs batch by batch
    as in eval.py:108, the scoring block of `score_group` consecutive batches is ONE launch (mm_maxsim_fwd_batched) and their
    scores come back with one `.cpu()` — same scores, same result order.
    graph=True: every batch shape's forward is captured once in a HIP graph and replayed (_GraphedForward) — for models
    whose forward is capturable (no host synchronisation, no data-dependent shapes: ColBERT, TK; not TKL, whose chunk
    packing is data dependent as in the reference)."""
    if device is None:
        import itertools
        t = next(itertools.chain(model.parameters(), model.buffers()), None)
        if t is None:
            raise ValueError("evaluate_batches: the model has no parameters or buffers; pass device=")
        device = t.device
    validation_results: Dict[str, List[Tuple[str, float]]] = {}
    graphed = _GraphedForward(model, use_fp16, output_secondary_output, torch.device(device)) if graph else None
    grouped = score_group > 1 and graphed is None and not output_secondary_output \
        and hasattr(model, "score_batches") and hasattr(model, "forward_representation")
    if grouped:
        pend = []      # (batch_orig, query_vecs, document_vecs, query mask, document mask)

        def flush():
            if not pend:
                return
            with torch.autocast("cuda", dtype=torch.float16, enabled=use_fp16):
                scores = model.score_batches([p[1:] for p in pend])
            flat = torch.cat([s.float().reshape(-1) for s in scores]).cpu()          # :161 — one piece per group
            off = 0
            for (bo, *_), s in zip(pend, scores):
                for i, qid in enumerate(bo["query_id"]):
                    validation_results.setdefault(qid, []).append((bo["doc_id"][i], float(flat[off + i])))
                off += s.numel()
            pend.clear()
        with torch.no_grad():
            for batch_orig in batches:
                with torch.autocast("cuda", dtype=torch.float16, enabled=use_fp16):
                    batch = _to_device({k: batch_orig[k] for k in ("query_tokens", "doc_tokens")}, device)
                    qv = model.forward_representation(batch["query_tokens"])
                    dv = model.forward_representation(batch["doc_tokens"])
                if pend and (qv.shape[1:] != pend[0][1].shape[1:] or dv.shape[1:] != pend[0][2].shape[1:]):
                    flush()                                    # eval.py pads every batch to its own longest sequence
                pend.append((batch_orig, qv, dv, batch["query_tokens"]["attention_mask"], batch["doc_tokens"]["attention_mask"]))
                if len(pend) >= score_group:
                    flush()
            flush()
        return validation_results
    with torch.no_grad():
        for batch_orig in batches:
            if graphed is not None:
                output = graphed(batch_orig).cpu()                                       # :161 — in one piece
                for i, qid in enumerate(batch_orig["query_id"]):
                    validation_results.setdefault(qid, []).append((batch_orig["doc_id"][i], float(output[i])))
                continue
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
