"""TEST INFRASTRUCTURE — CPU restatement (numpy) of matchmaker's interaction-scoring path.

This file is the parity ORACLE.  It is NOT product code: only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it; nothing under
matchmaker_amd/ does, and the product path raises if the HIP library is missing.

Every function restates one reference function op-by-op (same op order, same
intermediate dtypes) and cites the reference lines it follows.  Pinning status:
  * the reference holds NO golden vectors / tests for this path, so the
    oracle is pinned against outputs of the reference itself, run in the build container
    (oracle/ref_harness.py -> tests/golden/*.npz, generator tests/golden/gen_golden.py);
  * the cosine match matrix is third-party arithmetic (allennlp==2.5.1.dev20210625,
    pip-requirements.txt:1, not vendored): restated from its published formula, anchored on
    the reference call sites ecai20_tk.py:105 / sigir20_tkl.py:184.  No reference test pins
    that boundary -> for the cosine itself parity is "unpinned" beyond the shimmed reference run.

`dtype=np.float64` gives the accumulation-order-independent variant used to bound fp noise
in rank-order checks.
"""
import numpy as np

# ----------------------------------------------------------------------------- ColBERT
#
# The reference's DEFAULT precision mode (config/train/defaults.yaml:21 `use_fp16: True`, colbert.py:60) runs the block
# under torch.cuda.amp.autocast: `bmm` takes fp16 vectors and RETURNS an fp16 matrix (GPU GEMMs accumulate in fp32 and
# round each element once), the masked assignment and `max` are fp16 ops, and `sum` — on autocast's fp32 list — is
# promoted to fp32 (:68-75).  `sim_dtype` restates that: the similarity matrix is rounded to it (np.float16, or
# "bfloat16" for torch.bmm on bf16 tensors) right after the product; mask / max then act on representable values
# (-1000 is exact in both) exactly as the fp16 ops do.  `sum_dtype` rounds the pair's sum as well: 16-bit tensors OUTSIDE
# autocast (the dynamic teacher's all-pairs call, distillation/dynamic_teacher.py:245-246), where `sum` is a 16-bit op
# that accumulates in fp32 and rounds once.  Use dtype=np.float64 for the product to take the accumulation order out of
# the rounding decision (the inputs are fp16-valued: the fp64 product is exact to ~1e-16).


def _bf16_rne(x):
    """float32 -> bfloat16 (round to nearest even) -> float32, bit-exact with v_cvt_pk_bf16_f32."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << np.uint32(16)
    return r.view(np.float32)


def round_to(x, lowp):
    """x rounded (RNE) to the 16-bit type `lowp` (None = unchanged), returned in x's own dtype."""
    if lowp is None:
        return x
    x = np.asarray(x)
    if lowp == "bfloat16":
        return _bf16_rne(x.astype(np.float32)).astype(x.dtype)
    with np.errstate(over="ignore"):
        return x.astype(lowp).astype(x.dtype)


def maxsim_paired(q, d, q_mask, d_mask, dtype=np.float32, sim_dtype=None, sum_dtype=None):
    """ColBERT.forward scoring block — matchmaker/models/colbert.py:68-75.

    q [B,Q,E], d [B,D,E]; q_mask [B,Q], d_mask [B,D] (any dtype, nonzero = real token).
    score[b] = sum_{i: q_mask} max_j ( d_mask[b,j] ? <q_i, d_j> : -1000 )
    sim_dtype / sum_dtype: the 16-bit dtype flow of the autocast / all-fp16 modes (section comment above).
    """
    q = np.asarray(q, dtype=dtype)
    d = np.asarray(d, dtype=dtype)
    s = round_to(np.matmul(q, np.swapaxes(d, 1, 2)), sim_dtype)  # :68  bmm -> [B,Q,D] (fp16 under autocast)
    dm = np.asarray(d_mask) != 0
    s = np.where(dm[:, None, :], s, dtype(-1000))                # :69  doc pad -> -1000 (not -inf)
    m = s.max(-1)                                                # :71  max over D
    qm = np.asarray(q_mask) != 0
    m = np.where(qm, m, dtype(0))                                # :73  query pad -> 0
    return round_to(m.sum(-1, dtype=dtype), sum_dtype)           # :75  (fp32 under autocast)


def maxsim_unmasked(q, d, dtype=np.float32, sim_dtype=None, sum_dtype=None):
    """ColBERT.forward_aggregation — matchmaker/models/colbert.py:100-112 (no masks)."""
    q = np.asarray(q, dtype=dtype)
    d = np.asarray(d, dtype=dtype)
    s = round_to(np.matmul(q, np.swapaxes(d, 1, 2)), sim_dtype)  # :101
    return round_to(s.max(-1).sum(-1, dtype=dtype), sum_dtype)   # :104, :108


def maxsim_inbatch(q, q_mask, d, d_mask, bug_compatible=False, dtype=np.float32, sim_dtype=None, sum_dtype=None):
    """ColBERT.forward_inbatch_aggregation — matchmaker/models/colbert.py:154-162.

    q [Bq,Q,E], d [Bd,D,E] -> [Bq,Bd].  The reference (:158) expands the document mask as
    [Bd,1,1,D] -> (-1, Bd, Q, -1), i.e. score[i,j] is masked with doc *i*'s mask and the call
    needs Bq == Bd.  bug_compatible=True reproduces that; False masks with doc j's mask.
    """
    q = np.asarray(q, dtype=dtype)
    d = np.asarray(d, dtype=dtype)
    Bq, Q, E = q.shape
    Bd, D, _ = d.shape
    s = round_to(q.reshape(-1, E) @ d.reshape(-1, E).T, sim_dtype).reshape(Bq, Q, Bd, D)   # :154
    s = np.swapaxes(s, 1, 2)                                            # :156 [Bq,Bd,Q,D]
    dm = np.asarray(d_mask) != 0
    if bug_compatible:
        if Bq != Bd:
            raise ValueError("reference shape error: forward_inbatch_aggregation needs Bq == Bd")
        mask = dm[:, None, None, :]                                     # :158 (row i's doc)
    else:
        mask = dm[None, :, None, :]
    s = np.where(mask, s, dtype(-1000))
    m = s.max(-1)                                                       # :159
    qm = np.asarray(q_mask) != 0
    m = np.where(qm[:, None, :], m, dtype(0))                           # :160
    return round_to(m.sum(-1, dtype=dtype), sum_dtype)                  # :161


# ----------------------------------------------------------------------------- cosine


def cosine_matrix(a, b, dtype=np.float32):
    """allennlp CosineMatrixAttention.forward (allennlp 2.5.1, published source), as called at
    ecai20_tk.py:105 and sigir20_tkl.py:184:  x/(||x||_2 + 1e-13) on both sides, then bmm.
    tiny is added to the NORM (not the squared norm); an all-zero vector gives cosine 0."""
    a = np.asarray(a, dtype=dtype)
    b = np.asarray(b, dtype=dtype)
    tiny = dtype(1e-13)
    an = a / (np.sqrt((a * a).sum(-1, keepdims=True, dtype=dtype)) + tiny)
    bn = b / (np.sqrt((b * b).sum(-1, keepdims=True, dtype=dtype)) + tiny)
    return np.matmul(an, np.swapaxes(bn, -1, -2))


def cosine_matrix_split_bf16(a, b, lolo=False):
    """Emulation of the DEVICE arithmetic of the split-bf16 kernels: x = hi + lo with hi = bf16(x), lo = bf16(x - hi);
    dot = hi.hi + lo.hi + hi.lo — the THREE products the shipped TK pooling kernel and TKL's stage 1 compute
    (matchmaker_amd/csrc/kernel_pool.hip: MM_KP_LOLO and MM_TKL_LOLO both default to 0 since round 4).  lolo=True adds
    lo.lo: the four-product form of the 64n-wide kernels (kernel_pool128.hip: Conv-KNRM, IDCM sampler, fp32 MaxSim) and of
    the -DMM_KP_LOLO=1 / -DMM_TKL_LOLO=1 A/B builds.  (Products exact, fp32-class accumulation), norms from the
    fp32 values.  Not a reference restatement: it exists so the CPU suite can bound the device
    scheme's error against `cosine_matrix(..., float64)` without a GPU."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    ah, bh = _bf16_rne(a), _bf16_rne(b)
    al, bl = _bf16_rne(a - ah), _bf16_rne(b - bh)
    f = np.float64
    bt = lambda x: np.swapaxes(x.astype(f), -1, -2)
    dot = (ah.astype(f) @ bt(bh)).astype(np.float32) + ((al.astype(f) @ bt(bh)).astype(np.float32)
                                                        + (ah.astype(f) @ bt(bl) + (al.astype(f) @ bt(bl) if lolo else 0.0)).astype(np.float32))
    ra = np.float32(1) / (np.sqrt((a * a).sum(-1, keepdims=True, dtype=np.float32)) + np.float32(1e-13))
    rb = np.float32(1) / (np.sqrt((b * b).sum(-1, keepdims=True, dtype=np.float32)) + np.float32(1e-13))
    return (dot * ra) * np.swapaxes(rb, -1, -2)


def rbf_recurrence_fp32(c, mu, sigma):
    """Emulation of the DEVICE arithmetic of the pooling epilogue's recurrence form (matchmaker_amd/csrc/kp_device.h
    rbf_geo_one / detect_geo): the reference's kernel set — kernel 0 anything, kernels 1 .. 10 equally spaced (descending)
    with one width — evaluated from the two middle kernels outwards, exp2(-(t - j delta)^2) = exp2(-t^2) u^j g^(j (j - 1) / 2),
    every operation rounded to fp32 in the kernel's order; kernel 0 in the direct form.  c: cosines (any shape), masked
    positions as 1e5.  Returns [..., 11] float32 activations, or None when the kernel set is not one the device would run
    this way (it then evaluates exp2(-(c sq - mu sq)^2) per kernel: `rbf_direct_fp32`).  Not a reference restatement: it
    lets the CPU suite bound the recurrence's rounding against the exact kernel values without a GPU."""
    f = np.float32
    mu = np.asarray(mu, dtype=f)
    sg = np.asarray(sigma, dtype=f)
    if mu.shape[0] != 11:
        return None
    dmu = f(mu[5] - mu[6])
    ok = dmu > 0
    for k in range(1, 10):
        ok = ok and abs(f(f(mu[k] - mu[k + 1]) - dmu)) <= f(1e-6) and sg[k] == sg[k + 1]
    c2 = f(-1.4426950408889634) / (f(2.0) * sg * sg)
    sq = np.sqrt(-c2).astype(f)
    s = sq[1]
    delta = f(dmu * s)
    reach = f((f(1.0) + max(abs(mu[5]), abs(mu[6]))) * s)
    ok = ok and reach * reach < f(120.0) and f(2.0) * delta * f(13.0) < f(100.0)
    if not ok:
        return None
    c = np.asarray(c, dtype=f)
    t = np.stack([c * s + f(-mu[5] * s), c * s + f(-mu[6] * s)], -1).astype(f)
    t = np.clip(t, f(-13.0), f(13.0))
    e = np.exp2(-(t * t).astype(f)).astype(f)
    ud = np.exp2((t * np.array([f(2.0) * delta, f(-2.0) * delta], dtype=f) + f(-delta * delta)).astype(f)).astype(f)
    l2g = f(f(-2.0) * delta * delta)
    gs = [f(1), f(1), np.exp2(l2g).astype(f), np.exp2(f(3.0) * l2g).astype(f), np.exp2(f(6.0) * l2g).astype(f)]
    out = np.zeros(c.shape + (11,), dtype=f)
    out[..., 5], out[..., 6] = e[..., 0], e[..., 1]
    for j in range(2, 6):
        e = (e * ud).astype(f)
        out[..., 6 - j] = (e[..., 0] * gs[j - 1]).astype(f)     # (the device scales the pair's SUM once; same factor)
        out[..., 5 + j] = (e[..., 1] * gs[j - 1]).astype(f)
    sv = (c * sq[0] - f(mu[0] * sq[0])).astype(f)
    out[..., 0] = np.exp2(-(sv * sv).astype(f)).astype(f)
    return out


def rbf_direct_fp32(c, mu, sigma):
    """The direct device form, exp2(-(c sq - mu sq)^2) with sq = sqrt(log2(e) / (2 sigma^2)) (kp_device.h pack_rbf / rbf_block)."""
    f = np.float32
    mu = np.asarray(mu, dtype=f)
    sg = np.asarray(sigma, dtype=f)
    sq = np.sqrt(f(1.4426950408889634) / (f(2.0) * sg * sg)).astype(f)
    sv = (np.asarray(c, dtype=f)[..., None] * sq - (mu * sq).astype(f)).astype(f)
    return np.exp2(-(sv * sv).astype(f)).astype(f)


def maxsim_paired_split_bf16(q, d, q_mask, d_mask):
    """Emulation of the DEVICE arithmetic of the fp32 MaxSim path (kernel_pool128.hip, MX = true): every fp32
    operand x = hi + lo (two bf16), dot = hi.hi + lo.hi + hi.lo + lo.lo with exact products and fp32-class
    accumulation; masking / max / sum as colbert.py:69-75.  Not a reference restatement: it measures the
    rounding noise of the split scheme against `maxsim_paired(..., float64)` for the rank-order checks."""
    q = np.asarray(q, dtype=np.float32)
    d = np.asarray(d, dtype=np.float32)
    qh, dh = _bf16_rne(q), _bf16_rne(d)
    ql, dl = _bf16_rne(q - qh), _bf16_rne(d - dh)
    f = np.float64
    t = lambda x: np.swapaxes(x.astype(f), 1, 2)
    s = (np.matmul(qh.astype(f), t(dh)) + np.matmul(ql.astype(f), t(dh)) + np.matmul(qh.astype(f), t(dl))
         + np.matmul(ql.astype(f), t(dl))).astype(np.float32)
    s = np.where((np.asarray(d_mask) != 0)[:, None, :], s, np.float32(-1000))
    m = np.where(np.asarray(q_mask) != 0, s.max(-1), np.float32(0))
    return m.sum(-1, dtype=np.float32)


# ----------------------------------------------------------------------------- TK


def tk_kernel_pool(q, d, q_mask, d_mask, mu, sigma, alpha, w, dtype=np.float32,
                   return_per_kernel=False):
    """ECAI20_TK.forward kernel-pooling block — published/ecai20_tk.py:105-124.

    q [B,Q,E], d [B,D,E] are the *contextualised* embeddings (:95-96 stay in PyTorch).
    mu, sigma [K]; alpha = kernel_alpha_scaler [K]; w = kernel_bin_weights.weight[0] [K].
    """
    cos = cosine_matrix(q, d, dtype)                                             # :105
    mu = np.asarray(mu, dtype=dtype).reshape(1, 1, 1, -1)
    sigma = np.asarray(sigma, dtype=dtype).reshape(1, 1, 1, -1)
    raw = np.exp(-np.power(cos[..., None] - mu, 2) / (2 * np.power(sigma, 2)))   # :112
    dm = np.asarray(d_mask, dtype=dtype)
    masked = raw * dm[:, None, :, None]                                          # :114
    pkq = masked.sum(2, dtype=dtype)                                             # :120 [B,Q,K]
    alpha = np.asarray(alpha, dtype=dtype).reshape(1, 1, -1)
    lg = np.log(np.maximum(pkq * alpha, dtype(1e-10)))                           # :121
    qm = np.asarray(q_mask, dtype=dtype)
    lg = lg * qm[..., None]                                                      # :122
    per_kernel = lg.sum(1, dtype=dtype)                                          # :123 [B,K]
    score = per_kernel @ np.asarray(w, dtype=dtype).reshape(-1)                  # :124
    if return_per_kernel:
        return score, per_kernel
    return score


def knrm_kernel_pool(q, d, q_mask, d_mask, mu, sigma, w, dtype=np.float32, return_per_kernel=False):
    """KNRM.forward scoring block — matchmaker/models/knrm.py:55-84 (q, d already multiplied by their
    masks by forward_representation, :94-95).  mu, sigma [K]; w = dense.weight[0] [K]."""
    qm = np.asarray(q_mask, dtype=dtype)
    dm = np.asarray(d_mask, dtype=dtype)
    qd = qm[:, :, None] * dm[:, None, :]                                         # :55 query_by_doc_mask
    cos = cosine_matrix(q, d, dtype) * qd                                        # :62-63
    mu = np.asarray(mu, dtype=dtype).reshape(1, 1, 1, -1)
    sigma = np.asarray(sigma, dtype=dtype).reshape(1, 1, 1, -1)
    raw = np.exp(-np.power(cos[..., None] - mu, 2) / (2 * np.power(sigma, 2)))   # :71
    masked = raw * qd[..., None]                                                 # :72
    pkq = masked.sum(2, dtype=dtype)                                             # :74
    lg = np.log(np.maximum(pkq, dtype(1e-10))) * dtype(0.01)                     # :75
    lg = lg * qm[..., None]                                                      # :76
    per_kernel = lg.sum(1, dtype=dtype)                                          # :78
    score = per_kernel @ np.asarray(w, dtype=dtype).reshape(-1)                  # :84
    if return_per_kernel:
        return score, per_kernel
    return score


def tk_sparse_kernel_pool(q, d, q_mask, d_mask, stop_words, mu, sigma, alpha, w, dtype=np.float32,
                          return_per_kernel=False):
    """CIKM20_TK_Sparse.forward scoring block — published/cikm20_tk_sparse.py:106-146.

    q, d: the mixed (and masked, :170) contextualised embeddings; stop_words [B, D] = document_stop_words
    (:133, ReLU output already multiplied by the document mask) — the MLP that makes it stays PyTorch."""
    qm = np.asarray(q_mask, dtype=dtype)
    dm = np.asarray(d_mask, dtype=dtype)
    qd = qm[:, :, None] * dm[:, None, :]                                         # :106
    cos = cosine_matrix(q, d, dtype) * qd                                        # :113-114
    mu = np.asarray(mu, dtype=dtype).reshape(1, 1, 1, -1)
    sigma = np.asarray(sigma, dtype=dtype).reshape(1, 1, 1, -1)
    raw = np.exp(-np.power(cos[..., None] - mu, 2) / (2 * np.power(sigma, 2)))   # :122
    sw = np.asarray(stop_words, dtype=dtype)
    masked = raw * qd[..., None] * sw[:, None, :, None]                          # :135
    pkq = masked.sum(2, dtype=dtype)                                             # :141
    alpha = np.asarray(alpha, dtype=dtype).reshape(1, 1, -1)
    lg = np.log(np.maximum(pkq * alpha, dtype(1e-10))) * qm[..., None]           # :142-143
    per_kernel = lg.sum(1, dtype=dtype)                                          # :144
    score = per_kernel @ np.asarray(w, dtype=dtype).reshape(-1)                  # :145
    if return_per_kernel:
        return score, per_kernel
    return score


def idcm_sampler_scores(query_ctx, document_ctx, q_mask, d_mask, mu, sigma, alpha, w, bias, dtype=np.float32):
    """IDCM's fast passage-selection score (ESM) — published/sigir21_idcm.py:169-186.

    query_ctx [P,Q,E], document_ctx [P,D,E]: the sampler's contextualised vectors BEFORE
    torch.nn.functional.normalize (:169-170: x / max(|x|_2, 1e-12)); masks {0,1}; w [K], bias scalar =
    sampling_binweights (Linear(11, 1, bias=True), :102).  Returns packed_patch_scores [P]."""
    def normalize(x):
        x = np.asarray(x, dtype=dtype)
        n = np.sqrt((x * x).sum(-1, keepdims=True, dtype=dtype))
        return x / np.maximum(n, dtype(1e-12))
    cos = np.matmul(normalize(query_ctx), np.swapaxes(normalize(document_ctx), -1, -2))        # :182
    mu = np.asarray(mu, dtype=dtype).reshape(1, 1, 1, -1)
    sigma = np.asarray(sigma, dtype=dtype).reshape(1, 1, 1, -1)
    dm = np.asarray(d_mask, dtype=dtype)
    act = np.exp(-np.power(cos[..., None] - mu, 2) / (2 * np.power(sigma, 2))) * dm[:, None, :, None]   # :184
    alpha = np.asarray(alpha, dtype=dtype).reshape(1, 1, -1)
    qm = np.asarray(q_mask, dtype=dtype)
    res = np.log(np.maximum(act.sum(2, dtype=dtype) * alpha, dtype(1e-4))) * qm[..., None]     # :185
    return res.sum(1, dtype=dtype) @ np.asarray(w, dtype=dtype).reshape(-1) + dtype(bias)      # :186


# ----------------------------------------------------------------------------- TKL

TKL_CHUNK = 40       # sigir20_tkl.py:52
TKL_OVERLAP = 5      # :53
TKL_EXT = 50         # :54
TKL_WINDOW = 30      # :56
TKL_STRIDE = 2       # :209 unfold(2, 30, 2)
TKL_TOPK = 3         # :57


def tkl_chunk(d, d_mask):
    """Chunking in front of the hot path — sigir20_tkl.py:142-162 (stays host/PyTorch side in
    the product; restated here so the oracle can run end-to-end with a bypassed contextualiser).

    Returns chunks [B*C,50,E], chunk_mask [B*C,50], packed_indices [B*C] bool, C."""
    d = np.asarray(d)
    m = np.asarray(d_mask)
    B, D, E = d.shape
    if D > TKL_OVERLAP:
        need = TKL_EXT - ((D - TKL_OVERLAP) % TKL_CHUNK)                         # :143
    else:
        need = TKL_EXT - TKL_OVERLAP - D                                         # :145
    dp = np.pad(d, ((0, 0), (TKL_OVERLAP, need), (0, 0)))                        # :147
    mp = np.pad(m, ((0, 0), (TKL_OVERLAP, need)))                                # :148
    L = dp.shape[1]
    C = (L - TKL_EXT) // TKL_CHUNK + 1                                           # unfold(1,50,40)
    idx = (np.arange(C)[:, None] * TKL_CHUNK + np.arange(TKL_EXT)[None, :])      # [C,50]
    chunks = dp[:, idx, :].reshape(B * C, TKL_EXT, E)                            # :150,:156
    cmask = mp[:, idx].reshape(B * C, TKL_EXT)                                   # :151,:157
    packed = cmask[:, TKL_OVERLAP:-TKL_OVERLAP].sum(-1) != 0                     # :159
    return chunks, cmask, packed, C


def tkl_window_scores(q_ctx, q_mask, centre, centre_mask, packed_indices, B, params,
                      saturation="embedding", dtype=np.float32):
    """TKL match + windowed kernel pooling + saturation — sigir20_tkl.py:180-252.

    q_ctx [B,Q,E]        contextualised query, already multiplied by its mask (:306)
    centre [P,40,E]      documents_unique_again (:174), centre_mask [P,40] (:175)
    packed_indices [B*C] bool, P = packed_indices.sum()
    params: dict with mu, sigma [K]; dense_w [K]; sat_w{1,2,3} [2], sat_b{1,2,3} scalar;
            ln_w, ln_b [2]; emb_reduce_w [E]; kernel_mult0 [K] (log saturation).
    Returns score [B,W] *before* the ==0 -> -9900 rewrite of :257 (and W, for checks).
    """
    q_ctx = np.asarray(q_ctx, dtype=dtype)
    centre = np.asarray(centre, dtype=dtype)
    packed_indices = np.asarray(packed_indices, dtype=bool)
    Q = q_ctx.shape[1]
    K = len(params["mu"])
    BC = packed_indices.shape[0]
    C = BC // B
    # :180 query expanded to each kept chunk
    pq = np.repeat(q_ctx[:, None], C, axis=1).reshape(BC, Q, -1)[packed_indices]
    cos = cosine_matrix(pq, centre, dtype)                                       # :184 [P,Q,40]
    mu = np.asarray(params["mu"], dtype=dtype).reshape(1, 1, 1, -1)
    sigma = np.asarray(params["sigma"], dtype=dtype).reshape(1, 1, 1, -1)
    raw = np.exp(-np.power(cos[..., None] - mu, 2) / (2 * np.power(sigma, 2)))   # :193
    km = raw * np.asarray(centre_mask, dtype=dtype)[:, None, :, None]            # :194
    act = np.zeros((BC, Q, TKL_CHUNK, K), dtype=dtype)                           # :196
    act[packed_indices] = km                                                     # :197
    act = act.transpose(0, 2, 1, 3).reshape(B, C * TKL_CHUNK, Q, K).transpose(0, 2, 1, 3)  # :199
    Ptot = act.shape[2]
    if Ptot < TKL_WINDOW:                                                        # :206-207
        act = np.pad(act, ((0, 0), (0, 0), (0, TKL_WINDOW - Ptot), (0, 0)))
        Ptot = TKL_WINDOW
    W = (Ptot - TKL_WINDOW) // TKL_STRIDE + 1
    widx = np.arange(W)[:, None] * TKL_STRIDE + np.arange(TKL_WINDOW)[None, :]   # [W,30]
    unrolled = act[:, :, widx, :]                                                # :209 [B,Q,W,30,K]
    lengths = (unrolled.sum(-1, dtype=dtype) != 0).sum(-1)                       # :210 [B,Q,W]
    pkq = unrolled.sum(-2, dtype=dtype)                                          # :211 [B,Q,W,K]
    qm = np.asarray(q_mask, dtype=dtype)
    if saturation == "embedding":                                                # :224-234
        e = (q_ctx @ np.asarray(params["emb_reduce_w"], dtype=dtype).reshape(-1))  # [B,Q]
        infl = np.stack([np.broadcast_to(e[..., None], lengths.shape).astype(dtype),
                         lengths.astype(dtype)], axis=-1)                        # [B,Q,W,2]
        mean = infl.mean(-1, keepdims=True, dtype=dtype)
        var = ((infl - mean) ** 2).mean(-1, keepdims=True, dtype=dtype)
        infl = (infl - mean) / np.sqrt(var + dtype(1e-5))                        # LayerNorm(2) :228
        infl = infl * np.asarray(params["ln_w"], dtype=dtype) + np.asarray(params["ln_b"], dtype=dtype)
        lin = lambda w, b: (infl @ np.asarray(w, dtype=dtype).reshape(2, 1)) + dtype(b)
        sat1 = lin(params["sat_w1"], params["sat_b1"])                           # :230
        sat2 = dtype(1) / lin(params["sat_w2"], params["sat_b2"])                # :231
        sat3 = lin(params["sat_w3"], params["sat_b3"])                           # :232
        sat = sat1 * np.power(np.maximum(pkq, dtype(1e-10)), sat2) - sat3        # :234
    elif saturation == "log":                                                    # :245-246
        km0 = np.asarray(params["kernel_mult0"], dtype=dtype).reshape(1, 1, 1, -1)
        sat = np.log(np.maximum(pkq * km0, dtype(1e-10)))
    else:
        # "idf"/"linear" (:214-222, :236-243) read `query_idfs`, which is not an argument of
        # forward (:130 has it commented out): they raise NameError in the reference.
        raise NotImplementedError("saturation %r is dead code in the reference" % saturation)
    sat = sat * qm[:, :, None, None] * (lengths > 0).astype(dtype)[..., None]    # :248
    per_kernel = sat.sum(1, dtype=dtype)                                         # :249 [B,W,K]
    score = per_kernel @ np.asarray(params["dense_w"], dtype=dtype).reshape(-1)  # :251-252
    return score.astype(dtype)


def tkl_region_topk(score, chunk_scoring, dtype=np.float32):
    """TKL top-3 non-overlapping regions — sigir20_tkl.py:254-286.  score [B,W] -> [B]."""
    score = np.array(score, dtype=dtype, copy=True)
    B = score.shape[0]
    if score.shape[1] < TKL_TOPK:                                                # :254-255
        score = np.pad(score, ((0, 0), (0, TKL_TOPK - score.shape[1])))
    W = score.shape[1]
    score[score == 0] = dtype(-9900)                                             # :257
    work = score.copy()                                                          # :264
    r = np.arange(W)
    top = np.zeros((B, TKL_TOPK), dtype=np.int64)
    for c in range(TKL_TOPK):                                                    # :268-273
        best = work.argmax(1)            # first maximal index, like torch.argmax on CPU
        top[:, c] = best
        pool = np.abs(r[None, :] - best[:, None]) < TKL_WINDOW / 2
        work[pool] = dtype(-10001 - c)
    nb = np.concatenate([top, top - 1, top + 1, top - 2, top + 2], axis=1)       # :276
    nb = np.clip(nb, 0, W - 1)                                                   # :277-278
    vals = np.take_along_axis(score, nb, axis=1)                                 # :280-281
    vals = np.where(vals <= -9900, dtype(0), vals)                               # :282
    cs = np.asarray(chunk_scoring, dtype=dtype).reshape(1, -1)
    return (vals * cs).sum(1, dtype=dtype)                                       # :286


def tkl_forward_bypass(q, d, q_mask, d_mask, params, saturation="embedding", dtype=np.float32,
                       return_windows=False):
    """TKL_sigir20.forward (sigir20_tkl.py:128-294) with the contextualiser bypassed exactly as
    oracle/ref_harness.TKLBypass does: forward_representation(x, m) = x * m[...,None] (:306)."""
    q = np.asarray(q, dtype=dtype)
    d = np.asarray(d, dtype=dtype)
    qm = np.asarray(q_mask, dtype=dtype)
    B = q.shape[0]
    q_ctx = q * qm[..., None]                                                    # :139 / :306
    chunks, cmask, packed, C = tkl_chunk(d, np.asarray(d_mask, dtype=dtype))     # :142-162
    docs_packed = chunks[packed] * cmask[packed][..., None]                      # :172 / :306
    centre = docs_packed[:, TKL_OVERLAP:-TKL_OVERLAP, :]                         # :174
    centre_mask = cmask[packed][:, TKL_OVERLAP:-TKL_OVERLAP]                     # :175
    win = tkl_window_scores(q_ctx, qm, centre, centre_mask, packed, B, params, saturation, dtype)
    out = tkl_region_topk(win, params["chunk_scoring"], dtype)
    if return_windows:
        return out, win
    return out


def tkl_params_from_state(sd):
    """Collect the hot-path parameters from a TKL_sigir20 state_dict (numpy arrays / tensors)."""
    g = lambda k: np.asarray(sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else sd[k])
    return {
        "mu": g("mu").reshape(-1), "sigma": g("sigma").reshape(-1),
        "dense_w": g("dense.weight").reshape(-1),
        "sat_w1": g("saturation_linear.weight").reshape(-1), "sat_b1": float(g("saturation_linear.bias").reshape(-1)[0]),
        "sat_w2": g("saturation_linear2.weight").reshape(-1), "sat_b2": float(g("saturation_linear2.bias").reshape(-1)[0]),
        "sat_w3": g("saturation_linear3.weight").reshape(-1), "sat_b3": float(g("saturation_linear3.bias").reshape(-1)[0]),
        "ln_w": g("sat_normer.weight").reshape(-1), "ln_b": g("sat_normer.bias").reshape(-1),
        "emb_reduce_w": g("sat_emb_reduce1.weight").reshape(-1),
        "kernel_mult0": g("kernel_mult")[0].reshape(-1),
        "chunk_scoring": g("chunk_scoring").reshape(-1),
    }


# ----------------------------------------------------------------------------- dense retrieval


def dot_topk(queries, corpus, k, dtype=np.float32):
    """faiss IndexFlatIP.search as the reference uses it (retrieval/faiss_indices.py:29-36 on an
    IndexIDMap(IndexFlatIP), :65; score = BERT_DOT dot product, bert_dot.py:62): per query the k
    largest inner products, descending.  faiss (faiss-gpu==1.7.0, conda-requirements.txt:1) is a
    third-party dependency absent from the reference tree and from this image, so this restates its
    published semantics ("parity unpinned" beyond that); ties are broken by the lower row index, and
    rows are padded with (-inf, -1) when the collection has fewer than k vectors (faiss pads with -1).
    queries [nq,E], corpus [N,E] -> (scores [nq,k], idx [nq,k] int64)."""
    q = np.asarray(queries, dtype=dtype)
    c = np.asarray(corpus, dtype=dtype)
    s = q @ c.T
    order = np.argsort(-s, axis=1, kind="stable")[:, :k]
    top = np.take_along_axis(s, order, axis=1)
    if order.shape[1] < k:
        pad = k - order.shape[1]
        top = np.pad(top, ((0, 0), (0, pad)), constant_values=-np.inf)
        order = np.pad(order, ((0, 0), (0, pad)), constant_values=-1)
    return top.astype(dtype), order.astype(np.int64)


# ----------------------------------------------------------------------------- ranking


def rank_order(scores):
    """Per-query ranking rule of the reference: stable sort by score, descending, arrival order
    on ties — utils/core_metrics.py:502-511 (`sorted(..., key=score, reverse=True)` is stable).
    scores [C] -> indices [C]."""
    scores = np.asarray(scores)
    return np.argsort(-scores, kind="stable")
