"""Seeded synthetic "MSMARCO-shaped" inputs shared by bench.py, smoke() and tests.
Generated on whatever device is asked for; values are rounded to the compute dtype once so the
oracle can consume exactly the same numbers upcast to fp32."""
import torch


def msmarco_doc_lengths(n: int, D: int, gen: torch.Generator, device="cpu") -> torch.Tensor:
    """Passage-like token counts: N(70, 25) clipped to [8, D]."""
    x = torch.randn(n, generator=gen, device=device) * 25.0 + 70.0
    return x.round().clamp(8, D).to(torch.int32)


def colbert_batch(n_queries: int, cands: int, Q: int = 32, D: int = 180, E: int = 128,
                  dtype=torch.bfloat16, device="cpu", seed: int = 2002, lengths: str = "full"):
    """q [n_queries,Q,E], d [n_queries*cands,D,E] L2-normalised per token (dot = cosine),
    q_len [n_queries], d_len [n_queries*cands] int32.
    lengths: "full" (every position real: pure-roofline runs) | "msmarco" (ragged)."""
    gen = torch.Generator(device=device).manual_seed(seed)
    B = n_queries * cands
    q = torch.nn.functional.normalize(torch.randn(n_queries, Q, E, generator=gen, device=device), dim=-1).to(dtype)
    d = torch.empty(B, D, E, dtype=dtype, device=device)
    step = max(1, min(B, (1 << 28) // (D * E)))      # generate in slabs: fp32 temporaries stay small
    for s in range(0, B, step):
        n = min(step, B - s)
        d[s:s + n] = torch.nn.functional.normalize(
            torch.randn(n, D, E, generator=gen, device=device), dim=-1).to(dtype)
    if lengths == "full":
        q_len = torch.full((n_queries,), Q, dtype=torch.int32, device=device)
        d_len = torch.full((B,), D, dtype=torch.int32, device=device)
    elif lengths == "msmarco":
        q_len = torch.randint(4, Q + 1, (n_queries,), generator=gen, device=device).to(torch.int32)
        d_len = msmarco_doc_lengths(B, D, gen, device)
    else:
        raise ValueError(lengths)
    return q, d, q_len, d_len


def len_to_mask(lens: torch.Tensor, L: int, dtype=torch.int64) -> torch.Tensor:
    return (torch.arange(L, device=lens.device)[None, :] < lens[:, None]).to(dtype)
