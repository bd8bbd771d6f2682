"""Generates tests/golden/*.npz by running the REAL reference (imported read-only from
/root/reference through oracle/ref_harness.py) on seeded synthetic inputs.

Run in the build container only:   python tests/golden/gen_golden.py
The .npz files are committed; the GPU box never needs /root/reference.

Inputs are stored (not just seeds) so the fixtures do not depend on RNG stability:
bf16-/fp16-representable fp32 values stored as 16-bit patterns to keep the files small.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness as R  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def bf16_bits(x: torch.Tensor) -> np.ndarray:
    return x.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


def fp16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.float16).to(torch.float32)


def prefix_mask(lens, L, dtype):
    return (torch.arange(L)[None, :] < lens[:, None]).to(dtype)


def gen_colbert():
    g = torch.Generator().manual_seed(2002)
    B, Q, D, E = 8, 32, 180, 128
    q = bf16_round(torch.nn.functional.normalize(torch.randn(B, Q, E, generator=g), dim=-1))
    d = bf16_round(torch.nn.functional.normalize(torch.randn(B, D, E, generator=g), dim=-1))
    q_len = torch.tensor([32, 32, 4, 17, 32, 1, 32, 9])
    d_len = torch.tensor([180, 0, 70, 33, 1, 179, 64, 96])     # full, empty, ragged, block edges
    qm = prefix_mask(q_len, Q, torch.int64)
    dm = prefix_mask(d_len, D, torch.int64)
    # one non-prefix ("holes") mask pair: reference semantics are per-position, not lengths
    dm[3, 5] = 0
    dm[3, 40] = 1
    qm[3, 2] = 0
    fwd = R.colbert_forward(q, d, qm, dm).numpy()                          # colbert.py:68-75
    agg = R.colbert_forward_aggregation(q, d).numpy()                      # colbert.py:100-112
    inb = R.colbert_forward_inbatch_aggregation(q, qm, d, dm).numpy()      # colbert.py:154-162 (bug incl.)
    np.savez_compressed(os.path.join(OUT, "colbert_q32_d180_e128.npz"),
                        q_bf16=bf16_bits(q), d_bf16=bf16_bits(d), q_mask=qm.numpy().astype(np.uint8),
                        d_mask=dm.numpy().astype(np.uint8), forward=fwd, forward_aggregation=agg,
                        forward_inbatch_aggregation=inb)

    # odd shapes: Q not multiple of 32, D not multiple of 32, E = 64 / 768 (reference default dim)
    for (B, Q, D, E, seed) in [(5, 13, 47, 64, 11), (3, 38, 200, 768, 12), (4, 70, 33, 32, 13)]:
        g = torch.Generator().manual_seed(seed)
        q = bf16_round(torch.randn(B, Q, E, generator=g) * 0.3)
        d = bf16_round(torch.randn(B, D, E, generator=g) * 0.3)
        q_len = torch.randint(1, Q + 1, (B,), generator=g)
        d_len = torch.randint(0, D + 1, (B,), generator=g)
        d_len[0] = D
        qm = prefix_mask(q_len, Q, torch.int64)
        dm = prefix_mask(d_len, D, torch.int64)
        np.savez_compressed(os.path.join(OUT, f"colbert_q{Q}_d{D}_e{E}.npz"),
                            q_bf16=bf16_bits(q), d_bf16=bf16_bits(d),
                            q_mask=qm.numpy().astype(np.uint8), d_mask=dm.numpy().astype(np.uint8),
                            forward=R.colbert_forward(q, d, qm, dm).numpy(),
                            forward_aggregation=R.colbert_forward_aggregation(q, d).numpy(),
                            forward_inbatch_aggregation=(
                                R.colbert_forward_inbatch_aggregation(q, qm, d, dm).numpy()))


def gen_colbert_16bit_flow():
    """The reference's 16-bit dtype flow, run by the REAL class on fp16 / bf16 CPU tensors (autocast off: bmm / mm, the
    -1000 fill, max and sum are all 16-bit ops — what the dynamic teacher's all-pairs call executes on the GPU,
    dynamic_teacher.py:245-246).  Pins np_oracle's sim_dtype / sum_dtype variants and the device's
    MM_SIM_ROUND | MM_SUM_ROUND arithmetic."""
    for (tag, dt, B, Q, D, E, seed, unit) in [("fp16", torch.float16, 16, 32, 180, 128, 2102, True),
                                              ("fp16", torch.float16, 6, 38, 200, 768, 2103, False),
                                              ("fp16", torch.float16, 5, 13, 47, 64, 2104, False),
                                              ("bf16", torch.bfloat16, 16, 32, 180, 128, 2105, True)]:
        g = torch.Generator().manual_seed(seed)
        q = torch.randn(B, Q, E, generator=g)
        d = torch.randn(B, D, E, generator=g)
        if unit:
            q, d = torch.nn.functional.normalize(q, dim=-1), torch.nn.functional.normalize(d, dim=-1)
        else:
            q, d = q * 0.3, d * 0.3
        q, d = q.to(dt), d.to(dt)
        q_len = torch.randint(1, Q + 1, (B,), generator=g)
        d_len = torch.randint(0, D + 1, (B,), generator=g)
        d_len[0], d_len[1] = D, 0
        qm = prefix_mask(q_len, Q, torch.int64)
        dm = prefix_mask(d_len, D, torch.int64)
        dm[2, min(3, D - 1)] = 0                                         # a hole
        bits = (lambda x: x.view(torch.int16).numpy().view(np.uint16))
        np.savez_compressed(os.path.join(OUT, f"flow16_{tag}_q{Q}_d{D}_e{E}.npz"),
                            dtype=tag, q_bits=bits(q), d_bits=bits(d),
                            q_mask=qm.numpy().astype(np.uint8), d_mask=dm.numpy().astype(np.uint8),
                            forward=R.colbert_forward_16bit(q, d, qm, dm).float().numpy(),
                            forward_aggregation=R.colbert_forward_aggregation(q, d).float().numpy(),
                            forward_inbatch_aggregation=(
                                R.colbert_forward_inbatch_aggregation(q, qm, d, dm).float().numpy()))


def gen_colbert_e2e():
    # the real ColBERT.forward end to end (encoder + compressor + scoring, colbert.py:54-98) on token ids, with
    # a tiny randomly initialised BERT: what eval.py:108 calls, minus the checkpoint download
    from transformers import BertConfig, BertModel
    torch.manual_seed(11)
    enc = BertModel(BertConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                               vocab_size=500, max_position_embeddings=256, hidden_dropout_prob=0.0,
                               attention_probs_dropout_prob=0.0))
    m = R.make_colbert_with_encoder(enc, 128)
    g = torch.Generator().manual_seed(12)
    B, Q, D = 7, 32, 180
    ql = torch.randint(3, Q + 1, (B,), generator=g)
    dl = torch.randint(8, D + 1, (B,), generator=g)
    mk = lambda L, n: {"input_ids": torch.randint(1, 500, (B, n), generator=g),
                       "attention_mask": (torch.arange(n)[None] < L[:, None]).long()}
    query, doc = mk(ql, Q), mk(dl, D)
    with torch.no_grad():
        score = m.forward(query, doc, use_fp16=False)
        qv = m.forward_representation(query, "query_encode")
        dv = m.forward_representation(doc, "doc_encode")
        agg = m.forward_aggregation(qv, dv)
    sd = {("param." + k): v.detach().numpy() for k, v in m.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, "e2e_colbert_tinybert.npz"),
                        q_ids=query["input_ids"].numpy(), q_mask=query["attention_mask"].numpy(),
                        d_ids=doc["input_ids"].numpy(), d_mask=doc["attention_mask"].numpy(),
                        forward=score.numpy(), forward_aggregation=agg.numpy(), **sd)


def gen_e2e_tk_tkl():
    # the real ECAI20_TK / TKL_sigir20 classes END TO END (positional encoding + Transformer contextualiser +
    # mixer + chunking + pooling), eval mode, random init: what NeuralIR_Encoder.forward calls
    # (neuralIR_encoder.py:86-87).  Small dims keep the fixture small.
    g = torch.Generator().manual_seed(1301)
    B, Q, D, E = 5, 12, 70, 60
    q = torch.randn(B, Q, E, generator=g)
    d = torch.randn(B, D, E, generator=g)
    qm = prefix_mask(torch.tensor([12, 3, 7, 12, 1]), Q, torch.float32)
    dm = prefix_mask(torch.tensor([70, 20, 5, 33, 64]), D, torch.float32)
    m = R.make_tk(E, bypass_contextualizer=False, att_heads=6, att_ff_dim=32, max_length=80, seed=21)
    with torch.no_grad():
        m.kernel_alpha_scaler.uniform_(0.5, 1.5, generator=g)
        m.kernel_bin_weights.weight.uniform_(-0.5, 0.5, generator=g)
    score = R.tk_forward(m, q, d, qm, dm)
    sd = {("param." + k): v.detach().numpy() for k, v in m.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, "e2e_tk_q12_d70_e60.npz"), q=q.numpy(), d=d.numpy(), q_mask=qm.numpy(),
                        d_mask=dm.numpy(), score=score.numpy(), **sd)

    B, Q, D, E = 3, 10, 333, 64
    q = torch.randn(B, Q, E, generator=g)
    d = torch.randn(B, D, E, generator=g)
    qm = prefix_mask(torch.tensor([10, 4, 7]), Q, torch.float32)
    dm = prefix_mask(torch.tensor([333, 120, 41]), D, torch.float32)
    m = R.make_tkl(E, bypass_contextualizer=False, att_heads=8, att_ff_dim=32, seed=22)
    score = R.tkl_forward(m, q, d, qm, dm)
    sd = {("param." + k): v.detach().numpy() for k, v in m.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, "e2e_tkl_q10_d333_e64.npz"), q=q.numpy(), d=d.numpy(), q_mask=qm.numpy(),
                        d_mask=dm.numpy(), score=score.numpy(), **sd)


def gen_tk():
    # BASELINE.json config 1 shapes (1 query x candidates, Q=20/D=200/E=300), B cut to 4 for size
    g = torch.Generator().manual_seed(1001)
    B, Q, D, E = 4, 20, 200, 300
    q1 = fp16_round(torch.randn(1, Q, E, generator=g))
    q = q1.expand(B, Q, E).contiguous()                        # same query replicated per pair
    d = fp16_round(torch.randn(B, D, E, generator=g))
    # make some doc tokens near-duplicates of query tokens so the mu=1.0 / 0.9 bins are populated
    d[0, 3] = q1[0, 1]
    d[1, 7] = fp16_round(q1[0, 2] + 0.05 * torch.randn(E, generator=g))
    d[2, 0] = 0.0                                              # zero vector -> cosine exactly 0
    q_len = torch.tensor([11, 11, 11, 11])
    d_len = torch.tensor([200, 10, 77, 133])
    qm = prefix_mask(q_len, Q, torch.float32)
    dm = prefix_mask(d_len, D, torch.float32)
    m = R.make_tk(E, seed=5)
    with torch.no_grad():
        m.kernel_alpha_scaler.uniform_(0.5, 1.5, generator=g)
    score, sec = R.tk_forward(m, q, d, qm, dm, secondary=True)             # ecai20_tk.py:105-129
    np.savez_compressed(os.path.join(OUT, "tk_q20_d200_e300.npz"),
                        q_fp16=q1.to(torch.float16).numpy(), d_fp16=d.to(torch.float16).numpy(),
                        q_mask=qm.numpy(), d_mask=dm.numpy(),
                        mu=m.mu.numpy().reshape(-1), sigma=m.sigma.numpy().reshape(-1),
                        alpha=m.kernel_alpha_scaler.detach().numpy().reshape(-1),
                        w=m.kernel_bin_weights.weight.detach().numpy().reshape(-1),
                        score=score.numpy(), per_kernel=sec["per_kernel"].numpy())


def gen_knrm():
    # KNRM (knrm.py:44-92): 11 kernels incl. the exact-match one (sigma 1e-4); word-embedding-like inputs
    g = torch.Generator().manual_seed(1101)
    B, Q, D, E = 5, 14, 60, 300
    q = fp16_round(torch.randn(B, Q, E, generator=g))
    d = fp16_round(torch.randn(B, D, E, generator=g))
    d[0, 3] = q[0, 1]                                          # exact matches feed the sigma = 1e-4 kernel
    d[1, 7] = q[1, 0]
    d[1, 8] = fp16_round(q[1, 2] + 0.05 * torch.randn(E, generator=g))
    q_len = torch.tensor([14, 3, 9, 1, 14])
    d_len = torch.tensor([60, 20, 1, 33, 0])
    qm = prefix_mask(q_len, Q, torch.float32)
    dm = prefix_mask(d_len, D, torch.float32)
    m = R.make_knrm(11, seed=7)
    q_in, d_in = m.forward_representation(q, qm), m.forward_representation(d, dm)     # knrm.py:94-95
    score, sec = R.knrm_forward(m, q_in, d_in, qm, dm, secondary=True)
    np.savez_compressed(os.path.join(OUT, "knrm_q14_d60_e300.npz"),
                        q_fp16=q_in.to(torch.float16).numpy(), d_fp16=d_in.to(torch.float16).numpy(),
                        q_mask=qm.numpy(), d_mask=dm.numpy(),
                        mu=m.mu.numpy().reshape(-1), sigma=m.sigma.numpy().reshape(-1),
                        w=m.dense.weight.detach().numpy().reshape(-1),
                        score=score.numpy(), per_kernel=sec["per_kernel"].numpy())


def gen_conv_knrm():
    # Conv-KNRM (conv_knrm.py:63-173): 3 n-gram convolutions (out dim 128) -> 9 match matrices x 11 kernels
    g = torch.Generator().manual_seed(1201)
    B, Q, D, E = 4, 12, 50, 64
    q_len = torch.tensor([12, 3, 7, 12])
    d_len = torch.tensor([50, 20, 5, 0])
    qm = prefix_mask(q_len, Q, torch.float32)
    dm = prefix_mask(d_len, D, torch.float32)
    q = fp16_round(torch.randn(B, Q, E, generator=g)) * qm.unsqueeze(-1)        # embeddings arrive masked
    d = fp16_round(torch.randn(B, D, E, generator=g)) * dm.unsqueeze(-1)
    d[0, 3] = q[0, 1]
    d[0, 4] = q[0, 2]                                          # a matching bigram
    m = R.make_conv_knrm(E, 3, 11, 128, seed=9)
    with torch.no_grad():
        score = m.forward(q, d, qm, dm)
    sd = {("param." + k): v.detach().numpy() for k, v in m.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, "conv_knrm_q12_d50_e64.npz"),
                        q_fp16=q.to(torch.float16).numpy(), d_fp16=d.to(torch.float16).numpy(),
                        q_mask=qm.numpy(), d_mask=dm.numpy(), score=score.numpy(), **sd)


def gen_tkl():
    for name, (B, Q, D, E, heads, seed, sat) in {
        "tkl_d333_e300_embedding": (2, 20, 333, 300, 10, 3003, "embedding"),
        "tkl_d2048_e64_embedding": (2, 20, 2048, 64, 8, 3004, "embedding"),
        "tkl_d2048_e64_log": (2, 20, 2048, 64, 8, 3005, "log"),
        "tkl_d20_e64_embedding": (3, 6, 20, 64, 8, 3006, "embedding"),     # single chunk, W small
        "tkl_d4_e64_embedding": (2, 5, 4, 64, 8, 3007, "embedding"),       # D <= overlap branch (:145)
    }.items():
        g = torch.Generator().manual_seed(seed)
        q = fp16_round(torch.randn(B, Q, E, generator=g))
        d = fp16_round(torch.randn(B, D, E, generator=g))
        q_len = torch.randint(1, Q + 1, (B,), generator=g)
        d_len = torch.randint(1, D + 1, (B,), generator=g)
        d_len[0] = D
        if D > 1000:
            d_len[1] = 90          # most chunks empty -> packed_indices drops them (:159-162)
        qm = prefix_mask(q_len, Q, torch.float32)
        dm = prefix_mask(d_len, D, torch.float32)
        m = R.make_tkl(E, saturation_type=sat, att_heads=heads, seed=seed)
        with torch.no_grad():
            m.chunk_scoring.uniform_(0.5, 1.5, generator=g)
            m.kernel_mult.uniform_(0.5, 1.5, generator=g)
            m.sat_normer.weight.uniform_(0.5, 1.5, generator=g)
            m.sat_normer.bias.uniform_(-0.5, 0.5, generator=g)
        if sat == "embedding":
            score, sec = R.tkl_forward(m, q, d, qm, dm, secondary=True)    # sigir20_tkl.py:128-294
            extra = dict(orig_score=sec["orig_score"].numpy(),            # window scores, -9900 -> 0 (:284)
                         top_idx=sec["top_non_overlapping_idx"].numpy(),
                         top_k_non_overlapping=sec["top_k_non_overlapping"].numpy(),           # :281-282
                         sat_influence_from_top_k=sec["sat_influence_from_top_k"].numpy())     # :290
        else:
            # secondary output raises UnboundLocalError for "log" in the reference (:290)
            score = R.tkl_forward(m, q, d, qm, dm)
            extra = {}
        sd = {k: v.detach().numpy() for k, v in m.state_dict().items()
              if not k.startswith("contextualizer") and not k.startswith("positional")}
        np.savez_compressed(os.path.join(OUT, name + ".npz"),
                            q_fp16=q.to(torch.float16).numpy(), d_fp16=d.to(torch.float16).numpy(),
                            q_mask=qm.numpy(), d_mask=dm.numpy(), saturation=np.array(sat),
                            score=score.numpy(), **extra,
                            **{"param." + k: v for k, v in sd.items()})


def gen_tk_sparse():
    # TK-Sparse scoring block (cikm20_tk_sparse.py:106-146) at TK's shapes with the contextualiser bypassed
    # (the stop-word MLP :132-133 is live), and the whole class end to end at small dims.
    g = torch.Generator().manual_seed(1401)
    B, Q, D, E = 4, 20, 200, 300
    q = fp16_round(torch.randn(B, Q, E, generator=g))
    d = fp16_round(torch.randn(B, D, E, generator=g))
    d[0, 3] = q[0, 1]
    d[1, 7] = fp16_round(q[1, 2] + 0.05 * torch.randn(E, generator=g))
    qm = prefix_mask(torch.tensor([20, 11, 3, 20]), Q, torch.float32)
    dm = prefix_mask(torch.tensor([200, 10, 77, 133]), D, torch.float32)
    m = R.make_tk_sparse(E, seed=31)
    with torch.no_grad():
        m.kernel_alpha_scaler.uniform_(0.5, 1.5, generator=g)
        m.kernel_bin_weights.weight.uniform_(-0.5, 0.5, generator=g)
        m.stop_word_reducer2.weight.uniform_(-0.3, 0.3, generator=g)   # a gate that is closed for about half the tokens
        m.stop_word_reducer2.bias.fill_(0.05)
        score, sec, stop = m.forward(q, d, qm, dm, True)
    assert 0.2 < (stop[0, 0] == 0).float().mean() < 0.8
    sd = {("param." + k): v.detach().numpy() for k, v in m.state_dict().items()
          if not k.startswith("contextualizer") and not k.startswith("positional")}
    np.savez_compressed(os.path.join(OUT, "sparse_tk_q20_d200_e300.npz"),
                        q_fp16=q.to(torch.float16).numpy(), d_fp16=d.to(torch.float16).numpy(),
                        q_mask=qm.numpy(), d_mask=dm.numpy(), score=score.numpy(),
                        per_kernel=sec["per_kernel"].numpy(), document_stop_words=stop.numpy(), **sd)

    B, Q, D, E = 5, 12, 70, 60
    q = torch.randn(B, Q, E, generator=g)
    d = torch.randn(B, D, E, generator=g)
    qm = prefix_mask(torch.tensor([12, 3, 7, 12, 1]), Q, torch.float32)
    dm = prefix_mask(torch.tensor([70, 20, 5, 33, 64]), D, torch.float32)
    m = R.make_tk_sparse(E, bypass_contextualizer=False, att_heads=6, att_ff_dim=32, max_length=80, seed=32)
    with torch.no_grad():
        m.kernel_alpha_scaler.uniform_(0.5, 1.5, generator=g)
        m.kernel_bin_weights.weight.uniform_(-0.5, 0.5, generator=g)
        m.stop_word_reducer2.weight.uniform_(-0.3, 0.3, generator=g)
        m.stop_word_reducer2.bias.fill_(0.05)
        score, stop = m.forward(q, d, qm, dm)
    sd = {("param." + k): v.detach().numpy() for k, v in m.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, "e2e_sparse_tk_q12_d70_e60.npz"), q=q.numpy(), d=d.numpy(), q_mask=qm.numpy(),
                        d_mask=dm.numpy(), score=score.numpy(), document_stop_words=stop.numpy(), **sd)


def gen_idcm():
    # the real IDCM class end to end (sigir21_idcm.py:111-274), eval mode, around a tiny random DistilBERT:
    # windowing, the kernel-pooling passage sampler (:167-186), selection, BERT passage scores, top-k combination
    from transformers import DistilBertConfig, DistilBertModel
    for ctx_kind, seed in (("ck", 41), ("ck-small", 42)):
        cfg = DistilBertConfig(vocab_size=200, dim=64, n_heads=4, hidden_dim=128, n_layers=2,
                               max_position_embeddings=128, dropout=0.0, attention_dropout=0.0)
        torch.manual_seed(seed)
        bert = DistilBertModel(cfg).eval()
        m = R.make_idcm(bert, sample_n=2, sample_context=ctx_kind, top_k_chunks=2, seed=seed + 100)
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            m.kernel_alpha_scaler.uniform_(0.5, 1.5, generator=g)
            m.sampling_binweights.weight.uniform_(-0.5, 0.5, generator=g)
            m.sampling_binweights.bias.fill_(0.3)
            m.top_k_scoring.uniform_(0.5, 1.5, generator=g)
        B, LQ, LD = 4, 12, 221
        q_len = torch.tensor([12, 7, 12, 3])
        d_len = torch.tensor([221, 60, 120, 8])
        q_mask = prefix_mask(q_len, LQ, torch.long)
        d_mask = prefix_mask(d_len, LD, torch.long)
        q_ids = torch.randint(1, 200, (B, LQ), generator=g) * q_mask
        d_ids = torch.randint(1, 200, (B, LD), generator=g) * d_mask
        d_ids[0, 60:70] = q_ids[0, 1:11]                           # a passage that repeats the query
        with torch.no_grad():
            score, bert_scores, sec, _, _ = m.forward({"input_ids": q_ids, "attention_mask": q_mask},
                                                      {"input_ids": d_ids, "attention_mask": d_mask},
                                                      use_fp16=False, output_secondary_output=True)
        sd = {("param." + k): v.detach().numpy() for k, v in m.state_dict().items()}
        np.savez_compressed(os.path.join(OUT, "idcm_" + ctx_kind.replace("-", "_") + ".npz"),
                            q_ids=q_ids.numpy(), q_mask=q_mask.numpy(), d_ids=d_ids.numpy(), d_mask=d_mask.numpy(),
                            score=score.numpy(), bert_scores=bert_scores.numpy(),
                            sampling_scores=sec["sampling_scores"].numpy(),
                            packed_indices=sec["packed_indices"].numpy(), **sd)


def gen_tkl_grad():
    # gradients of the REAL TKL_sigir20.forward (contextualiser bypassed as in gen_tkl) w.r.t. its inputs and every
    # trainable scoring parameter: what loss.backward() (train.py:503-524) sends through sigir20_tkl.py:180-286
    for sat, seed in (("embedding", 3101), ("log", 3102)):
        g = torch.Generator().manual_seed(seed)
        B, Q, D, E = 3, 8, 333, 64
        q = fp16_round(torch.randn(B, Q, E, generator=g))
        d = fp16_round(torch.randn(B, D, E, generator=g))
        d[0, 100:108] = q[0]                                       # a region that matches the query
        d[1, 40:44] = q[1, :4]
        qm = prefix_mask(torch.tensor([8, 5, 3]), Q, torch.float32)
        dm = prefix_mask(torch.tensor([333, 200, 61]), D, torch.float32)
        m = R.make_tkl(E, saturation_type=sat, att_heads=8, seed=seed)
        with torch.no_grad():
            m.chunk_scoring.uniform_(0.5, 1.5, generator=g)
            m.kernel_mult.uniform_(0.5, 1.5, generator=g)
            m.sat_normer.weight.uniform_(0.5, 1.5, generator=g)
            m.sat_normer.bias.uniform_(-0.5, 0.5, generator=g)
            m.dense.weight.uniform_(-0.5, 0.5, generator=g)
            m.sat_emb_reduce1.weight.uniform_(-0.2, 0.2, generator=g)
            for lin in (m.saturation_linear, m.saturation_linear2, m.saturation_linear3):
                lin.weight.uniform_(-0.3, 0.3, generator=g)
                lin.bias.uniform_(1.0, 3.0, generator=g)           # the init of 100 (:99-107) would hide every gradient
        m.train()
        q.requires_grad_(True)
        d.requires_grad_(True)
        go = torch.randn(B, generator=g)
        score, sec = m.forward(q, d, qm, dm, True) if sat == "embedding" else (m.forward(q, d, qm, dm), None)
        (score * go).sum().backward()
        grads = {"grad." + k: p.grad.numpy() for k, p in m.named_parameters()
                 if p.grad is not None and not k.startswith(("contextualizer", "positional"))}
        extra = {"orig_score": sec["orig_score"].detach().numpy()} if sec is not None else {}
        sd = {"param." + k: v.detach().numpy() for k, v in m.state_dict().items()
              if not k.startswith("contextualizer") and not k.startswith("positional")}
        np.savez_compressed(os.path.join(OUT, "grad_tkl_d333_e64_%s.npz" % sat),
                            q_fp16=q.detach().to(torch.float16).numpy(), d_fp16=d.detach().to(torch.float16).numpy(),
                            q_mask=qm.numpy(), d_mask=dm.numpy(), saturation=np.array(sat), grad_out=go.numpy(),
                            score=score.detach().numpy(), grad_q=q.grad.numpy(), grad_d=d.grad.numpy(),
                            **extra, **grads, **sd)


GENERATORS = {"colbert": gen_colbert, "colbert_16bit_flow": gen_colbert_16bit_flow, "colbert_e2e": gen_colbert_e2e, "e2e_tk_tkl": gen_e2e_tk_tkl, "tk": gen_tk,
              "knrm": gen_knrm, "conv_knrm": gen_conv_knrm, "tkl": gen_tkl, "tk_sparse": gen_tk_sparse, "idcm": gen_idcm, "tkl_grad": gen_tkl_grad}

if __name__ == "__main__":
    for name in (sys.argv[1:] or list(GENERATORS)):      # python gen_golden.py [generator ...]
        GENERATORS[name]()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")
