"""TEST INFRASTRUCTURE — torch port of the reference's scoring blocks, op by op (device-agnostic: CPU tensors for the
CPU baselines and the oracle pins; GPU tensors when bench.py / the GPU tests run the reference's statements eagerly ON THE
GPU — under torch.autocast for the fp16 mode — as the stated eager baseline / reference-arithmetic checker).

The reference IS a sequence of torch ops; this file repeats exactly those ops (same order, same
in-place masked assignments) on CPU tensors so that
  * tests can differentiate through them with autograd (the training path's gradients), and
  * bench.py's `cpu_baseline` leg times what the reference's CPU path would execute on the host
    cores (kind "port": /root/reference does not exist on the GPU box, so the real classes cannot be
    imported there; tests/test_oracle_golden.py pins these functions on the real reference's outputs).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import torch


def maxsim_forward(query_vecs, document_vecs, query_mask, document_mask):
    """ColBERT.forward scoring block — matchmaker/models/colbert.py:68-75."""
    score_per_term = torch.bmm(query_vecs, document_vecs.transpose(2, 1))                       # :68
    score_per_term[~(document_mask.bool()).unsqueeze(1).expand(-1, score_per_term.shape[1], -1)] = -1000  # :69
    score = score_per_term.max(-1).values                                                       # :71
    score[~(query_mask.bool())] = 0                                                             # :73
    return score.sum(-1)                                                                        # :75


def maxsim_aggregation(query_vecs, document_vecs):
    """ColBERT.forward_aggregation — matchmaker/models/colbert.py:100-112."""
    score = torch.bmm(query_vecs, document_vecs.transpose(2, 1))                                # :101
    score = score.max(-1).values                                                                # :104
    return score.sum(-1)                                                                        # :108


def maxsim_inbatch(query_vecs, query_mask, document_vecs, document_mask, bug_compatible=False):
    """ColBERT.forward_inbatch_aggregation — matchmaker/models/colbert.py:154-162: one matrix product over all
    (query token, document token) pairs, viewed as [Bq, Bd, Q, D].  bug_compatible=True masks score[i, j] with the
    mask row of index i as :158 does (it needs Bq == Bd); False masks with document j's own row."""
    Bq, Q, E = query_vecs.shape
    Bd, D, _ = document_vecs.shape
    score = torch.mm(query_vecs.reshape(Bq * Q, E), document_vecs.reshape(Bd * D, E).t())      # :154
    score = score.view(Bq, Q, Bd, D).transpose(1, 2)                                            # :155-156
    keep = document_mask.bool()
    if bug_compatible:
        keep = keep.unsqueeze(1).unsqueeze(1).expand(-1, Bd, Q, -1)                             # :158 (row i)
    else:
        keep = keep.unsqueeze(0).unsqueeze(2).expand(Bq, -1, Q, -1)
    score = score.masked_fill(~keep, -1000)                                                     # :158
    score = score.max(-1).values                                                                # :159
    score = score.masked_fill(~query_mask.bool().unsqueeze(1).expand(-1, Bd, -1), 0)            # :160
    return score.sum(-1)                                                                        # :161


def maxsim_forward_backward(q, d, query_mask, document_mask, grad_out):
    """Autograd through maxsim_forward: what loss.backward() (train.py:503-524) sends into the
    encoder outputs.  Returns (score, grad_q, grad_d) as float32 CPU tensors."""
    q = q.detach().float().clone().requires_grad_(True)
    d = d.detach().float().clone().requires_grad_(True)
    out = maxsim_forward(q, d, query_mask, document_mask)
    out.backward(grad_out.float())
    return out.detach(), q.grad, d.grad


def cosine_matrix(a, b):
    """allennlp 2.5.1 CosineMatrixAttention.forward (published source; call sites ecai20_tk.py:105,
    sigir20_tkl.py:184): x / (x.norm(p=2, dim=-1, keepdim=True) + tiny), then bmm.  tiny = 1e-13 (fp32)."""
    a_norm = a / (a.norm(p=2, dim=-1, keepdim=True) + 1e-13)
    b_norm = b / (b.norm(p=2, dim=-1, keepdim=True) + 1e-13)
    return torch.bmm(a_norm, b_norm.transpose(-1, -2))


def tk_kernel_pool(query_embeddings, document_embeddings, query_mask, document_mask, mu, sigma, alpha, weight):
    """ECAI20_TK.forward kernel-pooling block — matchmaker/models/published/ecai20_tk.py:105-124.
    mu, sigma [1,1,1,K]; alpha = kernel_alpha_scaler [1,1,K]; weight = kernel_bin_weights.weight [1,K]."""
    cosine_matrix_ = cosine_matrix(query_embeddings, document_embeddings).unsqueeze(-1)         # :105-110
    raw_kernel_results = torch.exp(- torch.pow(cosine_matrix_ - mu, 2) / (2 * torch.pow(sigma, 2)))   # :112
    kernel_results_masked = raw_kernel_results * document_mask.unsqueeze(1).unsqueeze(-1)       # :114
    per_kernel_query = torch.sum(kernel_results_masked, 2)                                      # :120
    log_per_kernel_query = torch.log(torch.clamp(per_kernel_query * alpha, min=1e-10))          # :121
    log_per_kernel_query_masked = log_per_kernel_query * query_mask.unsqueeze(-1)               # :122
    per_kernel = torch.sum(log_per_kernel_query_masked, 1)                                      # :123
    return torch.nn.functional.linear(per_kernel, weight).squeeze(1)                            # :124


def tk_sparse_kernel_pool(query_embeddings, document_embeddings, query_mask, document_mask, document_stop_words,
                          mu, sigma, alpha, weight):
    """CIKM20_TK_Sparse.forward scoring block — published/cikm20_tk_sparse.py:106-146.
    document_stop_words [B,1,D] (:133); mu, sigma [1,1,1,K]; alpha [1,1,K]; weight [1,K]."""
    query_by_doc_mask = torch.bmm(query_mask.unsqueeze(-1), document_mask.unsqueeze(-1).transpose(-1, -2))   # :106
    cosine_masked = cosine_matrix(query_embeddings, document_embeddings) * query_by_doc_mask                # :113-114
    raw_kernel_results = torch.exp(- torch.pow(cosine_masked.unsqueeze(-1) - mu, 2) / (2 * torch.pow(sigma, 2)))  # :122
    kernel_results_masked = raw_kernel_results * query_by_doc_mask.unsqueeze(-1) * document_stop_words.unsqueeze(-1)  # :135
    per_kernel_query = torch.sum(kernel_results_masked, 2)                                      # :141
    log_per_kernel_query = torch.log(torch.clamp(per_kernel_query * alpha, min=1e-10))          # :142
    per_kernel = torch.sum(log_per_kernel_query * query_mask.unsqueeze(-1), 1)                  # :143-144
    return torch.nn.functional.linear(per_kernel, weight).squeeze(1)                            # :145


def idcm_sampler_scores(query_ctx, document_ctx, query_mask, document_mask, mu, sigma, alpha, weight, bias):
    """IDCM passage-sampler score — published/sigir21_idcm.py:169-186 (query_ctx / document_ctx BEFORE the
    normalize of :169-170).  mu, sigma [1,1,1,K]; alpha [1,1,K]; weight [1,K], bias [1]."""
    query_ctx = torch.nn.functional.normalize(query_ctx, p=2, dim=-1)                           # :169
    document_ctx = torch.nn.functional.normalize(document_ctx, p=2, dim=-1)                     # :170
    cosine_matrix_ = torch.bmm(query_ctx, document_ctx.transpose(-1, -2)).unsqueeze(-1)         # :182
    kernel_activations = torch.exp(- torch.pow(cosine_matrix_ - mu, 2) / (2 * torch.pow(sigma, 2))) * \
        document_mask.unsqueeze(-1).unsqueeze(1)                                                # :184
    kernel_res = torch.log(torch.clamp(torch.sum(kernel_activations, 2) * alpha, min=1e-4)) * \
        query_mask.unsqueeze(-1)                                                                # :185
    return torch.nn.functional.linear(torch.sum(kernel_res, 1), weight, bias)                   # :186  [P,1]


def tkl_scoring(query_ctx, centre, centre_mask, packed_indices, batch_size, query_mask, p, saturation="embedding"):
    """TKL_sigir20.forward after the contextualiser — matchmaker/models/published/sigir20_tkl.py:180-286, the same
    torch ops on CPU tensors (bench.py's CPU leg for BASELINE configs[2]; pinned on tests/golden/tkl_*.npz).

    query_ctx [B,Q,E] (contextualised, times its mask, :306); centre [P,40,E] / centre_mask [P,40] = the packed chunks'
    centre tokens (:174-175); packed_indices [B*C] bool (:159); p: float tensors mu, sigma, dense_w [K], sat_w1..3 [2],
    sat_b1..3, ln_w, ln_b [2], emb_reduce_w [E], kernel_mult0 [K], chunk_scoring [15] (np_oracle.tkl_params_from_state).
    Returns (score [B], window scores [B,W] with 0 for empty windows)."""
    Q, K = query_ctx.shape[1], p["mu"].numel()
    chunk_pieces = packed_indices.shape[0] // batch_size
    packed_query = query_ctx.unsqueeze(1).expand(-1, chunk_pieces, -1, -1).reshape(-1, Q, query_ctx.shape[-1])[packed_indices]  # :180
    cos = cosine_matrix(packed_query, centre).unsqueeze(-1)                                                   # :184, :192
    raw = torch.exp(- torch.pow(cos - p["mu"].view(1, 1, 1, -1), 2) / (2 * torch.pow(p["sigma"].view(1, 1, 1, -1), 2)))  # :193
    masked = raw * centre_mask.unsqueeze(1).unsqueeze(-1)                                                     # :194
    act = torch.zeros((packed_indices.shape[0], Q, centre.shape[1], K), dtype=centre.dtype, device=centre.device)                   # :196
    act[packed_indices] = masked                                                                              # :197
    act = act.transpose(1, 2).reshape(batch_size, -1, Q, K).transpose(2, 1)                                   # :199
    if act.shape[2] < 30:                                                                                     # :206-207
        act = torch.nn.functional.pad(act, (0, 0, 0, 30 - act.shape[2]))
    unrolled = act.unfold(2, 30, 2).transpose(-1, -2)                                                         # :209
    lengths = torch.sum(unrolled.sum(dim=-1) != 0, dim=-1)                                                    # :210
    per_kernel_query = torch.sum(unrolled, -2)                                                                # :211
    lin = lambda x, w, b: torch.nn.functional.linear(x, w.view(1, 2), b.view(1))
    if saturation == "embedding":                                                                             # :224-234
        infl = torch.cat([torch.nn.functional.linear(query_ctx, p["emb_reduce_w"].view(1, -1)).expand_as(lengths).unsqueeze(-1),
                          lengths.float().unsqueeze(-1)], dim=-1)
        infl = torch.nn.functional.layer_norm(infl, (2,), p["ln_w"], p["ln_b"], 1e-5)
        sat = lin(infl, p["sat_w1"], p["sat_b1"]) * (torch.clamp(per_kernel_query, min=1e-10) ** (1 / lin(infl, p["sat_w2"], p["sat_b2"]))) \
            - lin(infl, p["sat_w3"], p["sat_b3"])
    else:                                                                                                     # :245-246
        sat = torch.log(torch.clamp(per_kernel_query * p["kernel_mult0"].view(1, 1, 1, -1), min=1e-10))
    sat = sat * query_mask.unsqueeze(-1).unsqueeze(-1) * (lengths > 0).float().unsqueeze(-1)                  # :248
    score = torch.nn.functional.linear(torch.sum(sat, 1), p["dense_w"].view(1, -1)).squeeze(-1)               # :249-252
    if score.shape[1] < 3:                                                                                    # :254-255
        score = torch.nn.functional.pad(score, (0, 3 - score.shape[1]))
    score[score == 0] = -9900                                                                                 # :257
    orig = score
    top = torch.zeros((orig.shape[0], 3), dtype=torch.long, device=orig.device)
    work = orig.clone()
    r = torch.arange(work.shape[1], device=work.device)
    for c in range(3):                                                                                        # :268-273
        best = torch.argmax(work, dim=1)
        top[:, c] = best
        work[torch.abs(r - best.unsqueeze(-1)) < 30 / 2] = -10001 - c
    nb = torch.cat([top, top - 1, top + 1, top - 2, top + 2], dim=1)                                          # :276
    nb[nb < 0] = 0
    nb[nb >= orig.shape[1]] = orig.shape[1] - 1
    flat = (nb + torch.arange(0, orig.shape[0] * orig.shape[1], orig.shape[1], device=orig.device).unsqueeze(-1)).view(-1)   # :280
    vals = orig.view(-1).index_select(0, flat).view(top.shape[0], -1)     # :281 (index_select, not gather: its backward keeps no
    #                                                                       reference to `orig`, which :284 rewrites in place)
    vals[vals <= -9900] = 0                                                                                   # :282
    orig[orig <= -9900] = 0                                                                                   # :284
    return (vals * p["chunk_scoring"].view(1, -1)).sum(dim=1), orig                                           # :286
