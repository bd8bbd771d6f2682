"""TEST INFRASTRUCTURE — the TKL document score re-evaluated from the windows the region search picks, in differentiable
torch ops (sigir20_tkl.py:184-286 restricted to those 15 windows).  The native backward (mm_tkl_bwd) is tested against
autograd through this function, which is itself pinned on gradients of the REAL TKL_sigir20.forward
(tests/golden/grad_tkl_*.npz, tests/test_variants_cpu.py).  `self` is a matchmaker_amd.tkl.TKL_sigir20 module."""
import torch

from matchmaker_amd.tkl import CHUNK, OVERLAP, TOP_K, WINDOW


def selected_window_scores(self, query_ctx, chunks_ctx, chunk_mask, chunk_slot, query_mask, win, C):
    """Differentiable re-evaluation of the document score from the windows the region search picks
    (sigir20_tkl.py:184-286 restricted to those windows).  win [B, W]: the native window scores (0 = empty)."""
    B, W = win.shape
    dev = win.device
    # region search on the native window scores: three arg-max rounds with +-15 suppression (:262-272)
    s = torch.where(win == 0, win.new_full((), -9900.0), win)
    r = torch.arange(W, device=dev)
    picks = []
    for c in range(TOP_K):
        best = torch.argmax(s, dim=1)
        picks.append(best)
        s = torch.where((r.unsqueeze(0) - best.unsqueeze(1)).abs() < WINDOW / 2, s.new_full((), -10001.0 - c), s)
    top = torch.stack(picks, dim=1)
    idx = torch.cat([top, top - 1, top + 1, top - 2, top + 2], dim=1).clamp_(0, W - 1)          # :275-277 [B, 15]
    # the 30 positions of every selected window -> (packed chunk, row) of the contextualised chunks
    t = 2 * idx.unsqueeze(-1) + torch.arange(WINDOW, device=dev)                                # [B, 15, 30]
    in_doc = t < C * CHUNK                                   # documents shorter than a window are padded (:203-204)
    t = t.clamp(max=C * CHUNK - 1)
    slot = torch.arange(B, device=dev).view(B, 1, 1) * C + t // CHUNK
    slot2p = torch.full((B * C,), -1, dtype=torch.long, device=dev)
    slot2p[chunk_slot.long()] = torch.arange(chunk_slot.numel(), device=dev)
    p = slot2p[slot]
    present = (p >= 0) & in_doc
    p = p.clamp(min=0)
    row = t % CHUNK + OVERLAP
    if chunks_ctx.shape[0] == 0:
        return (win.sum(1) * 0.0) + 0.0 * self.chunk_scoring.sum()
    vec = chunks_ctx[p, row]                                                                     # [B, 15, 30, E]
    m = chunk_mask.to(vec.dtype)[p, row] * present.to(vec.dtype)
    qn = query_ctx / (query_ctx.norm(p=2, dim=-1, keepdim=True) + 1e-13)
    dn = vec / (vec.norm(p=2, dim=-1, keepdim=True) + 1e-13)
    cos = torch.einsum("bqe,bwte->bwqt", qn, dn)                                                 # :184
    act = torch.exp(-torch.pow(cos.unsqueeze(-1) - self.mu.view(1, 1, 1, 1, -1), 2) /
                    (2 * torch.pow(self.sigma.view(1, 1, 1, 1, -1), 2))) * m.unsqueeze(2).unsqueeze(-1)   # :192-194
    lengths = (act.sum(dim=-1) != 0).sum(dim=-1)                                                 # :210 [B, 15, Q]
    pkq = act.sum(dim=3)                                                                         # :211 [B, 15, Q, K]
    if self.use_embedding_sat:                                                                   # :224-235
        infl = torch.cat([self.sat_emb_reduce1(query_ctx).unsqueeze(1).expand(-1, idx.shape[1], -1, -1),
                          lengths.to(pkq.dtype).unsqueeze(-1)], dim=-1)
        infl = self.sat_normer(infl)
        sat = self.saturation_linear(infl) * (torch.clamp(pkq, min=1e-10) ** (1 / self.saturation_linear2(infl))) - \
            self.saturation_linear3(infl)
    else:                                                                                        # :246
        sat = torch.log(torch.clamp(pkq * self.kernel_mult[0].view(1, 1, 1, -1), min=1e-10))
    sat = sat * query_mask.to(sat.dtype).view(B, 1, -1, 1) * (lengths > 0).to(sat.dtype).unsqueeze(-1)   # :248
    wscore = self.dense(sat.sum(dim=2)).squeeze(-1)                                              # :249-252 [B, 15]
    wscore = torch.where(wscore == 0, torch.zeros_like(wscore), wscore)     # :257, :280: exact zeros are constants
    return (wscore * self.chunk_scoring).sum(dim=1)                                              # :284

