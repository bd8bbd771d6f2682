"""TEST INFRASTRUCTURE — drives the *real* reference classes from /root/reference.

Only usable in the build container (the reference tree does not exist on the GPU
box).  Nothing under matchmaker_amd/ imports this.  It is used by
tests/golden/gen_golden.py to produce the committed golden vectors and by the
`not gpu` tests (when /root/reference is present) to pin oracle/np_oracle.py.

No reference source is copied: the modules are imported from where they lie.

What is shimmed (all non-arithmetic, or third-party arithmetic restated from its
published source):
  * allennlp ...cosine_matrix_attention.CosineMatrixAttention  — allennlp==2.5.1
    (pip-requirements.txt:1) is not installed; restated from the published 2.x source:
    x / (||x||_2 + tiny), tiny = 1e-13 (fp32/fp64) or 1e-4 (fp16); bmm.
  * allennlp TextFieldEmbedder / dot_product_matrix_attention — import-only stubs
    (sigir20_tkl.py:7-9 imports them, the hot path never calls them).
  * get_range_vector -> CPU branch (ecai20_tk.py:204, sigir20_tkl.py:373),
    torch.cuda.FloatTensor -> torch.FloatTensor for TKL's constructor on CPU
    (sigir20_tkl.py:68-69).
"""
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("MM_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "matchmaker", "models"))


class _CosineMatrixAttention(nn.Module):
    """allennlp.modules.matrix_attention.cosine_matrix_attention (2.x), restated."""

    def forward(self, matrix_1: torch.Tensor, matrix_2: torch.Tensor) -> torch.Tensor:
        def tiny(dtype):
            if dtype in (torch.float, torch.double):
                return 1e-13
            if dtype == torch.half:
                return 1e-4
            raise TypeError("Does not support dtype " + str(dtype))

        a_norm = matrix_1 / (matrix_1.norm(p=2, dim=-1, keepdim=True) + tiny(matrix_1.dtype))
        b_norm = matrix_2 / (matrix_2.norm(p=2, dim=-1, keepdim=True) + tiny(matrix_2.dtype))
        return torch.bmm(a_norm, b_norm.transpose(-1, -2))


_installed = False


def install_shims():
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not found at " + REFERENCE_ROOT)

    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    if "allennlp" not in sys.modules:
        mod("allennlp")
        mod("allennlp.modules")
        ma = mod("allennlp.modules.matrix_attention")
        cm = mod("allennlp.modules.matrix_attention.cosine_matrix_attention")
        cm.CosineMatrixAttention = _CosineMatrixAttention
        dp = mod("allennlp.modules.matrix_attention.dot_product_matrix_attention")
        dp.__all__ = []
        tf = mod("allennlp.modules.text_field_embedders")
        tf.TextFieldEmbedder = nn.Module
        ma.cosine_matrix_attention = cm
    if not torch.cuda.is_available():
        # sigir20_tkl.py:68-69 builds mu/sigma with torch.cuda.FloatTensor
        torch.cuda.FloatTensor = torch.FloatTensor  # type: ignore[attr-defined]
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


# --------------------------------------------------------------------------- ColBERT
class _StubEncoder(nn.Module):
    """Stands in for bert_model: returns the pre-made token vectors passed in as
    tokens["vecs"] (colbert.py:92 takes element [0] of the encoder output)."""

    def forward(self, vecs=None, attention_mask=None, **kw):
        return (vecs,)


def _make_colbert():
    install_shims()
    from matchmaker.models.colbert import ColBERT  # noqa

    m = ColBERT.__new__(ColBERT)  # constructor needs network weights (colbert.py:44)
    nn.Module.__init__(m)
    m.bert_model = _StubEncoder()
    m.compressor = nn.Identity()
    m.return_vecs = False
    m.eval()
    return m


def colbert_forward(q, d, q_mask, d_mask, grad=False):
    """Real ColBERT.forward (colbert.py:54-86) with identity encoder/compressor.
    q [B,Q,E], d [B,D,E] float32; masks int64 {0,1}.  Returns [B] float32.
    grad=True keeps the autograd graph (gradients of the real scoring block w.r.t. q / d)."""
    m = _make_colbert()
    with torch.set_grad_enabled(grad):
        return m.forward({"vecs": q, "attention_mask": q_mask},
                         {"vecs": d, "attention_mask": d_mask}, use_fp16=False)


def colbert_forward_16bit(q, d, q_mask, d_mask):
    """Real ColBERT.forward (colbert.py:54-86) on 16-bit CPU token vectors with autocast off: `bmm`, the masked assignment,
    `max` AND `sum` run as fp16 (bf16) ops — the dtype flow MM_SIM_ROUND | MM_SUM_ROUND restates.  (CUDA autocast, which
    promotes `sum` to fp32, is inert on the CPU: that mode is checked on the GPU box against oracle/torch_port under
    torch.autocast, tests/test_fp16_flow_gpu.py.)  Returns [B] in the vectors' dtype."""
    assert q.dtype in (torch.float16, torch.bfloat16)
    m = _make_colbert()
    with torch.no_grad():
        return m.forward({"vecs": q, "attention_mask": q_mask},
                         {"vecs": d, "attention_mask": d_mask}, use_fp16=False)


def make_colbert_with_encoder(encoder, compression_dim):
    """Real ColBERT class (colbert.py) around a given HF encoder: the constructor (:37-52) downloads weights by
    name and its config class does not validate under transformers >= 5, so the object is assembled the way
    the constructor does — bert_model, compressor = Linear(hidden, compression_dim) (:50), return_vecs — and
    every method that runs afterwards (forward :54-86, forward_representation :88-98) is the reference's own."""
    install_shims()
    from matchmaker.models.colbert import ColBERT  # noqa

    m = ColBERT.__new__(ColBERT)
    nn.Module.__init__(m)
    m.bert_model = encoder
    m.compressor = nn.Linear(encoder.config.hidden_size, compression_dim)
    m.return_vecs = False
    m.eval()
    return m


def colbert_forward_aggregation(q, d):
    """Real ColBERT.forward_aggregation (colbert.py:100-112); does not use self."""
    install_shims()
    from matchmaker.models.colbert import ColBERT

    with torch.no_grad():
        return ColBERT.forward_aggregation(None, q, d)


def colbert_forward_inbatch_aggregation(q, q_mask, d, d_mask):
    """Real ColBERT.forward_inbatch_aggregation (colbert.py:114-162), bug included."""
    install_shims()
    from matchmaker.models.colbert import ColBERT

    with torch.no_grad():
        return ColBERT.forward_inbatch_aggregation(None, q, q_mask, d, d_mask)


# --------------------------------------------------------------------------- TK
TK_MU = [1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9]   # config/train/models/tk.yaml:18
TK_SIGMA = [0.1] * 11                                                  # config/train/models/tk.yaml:19


def make_tk(embsize=300, mu=TK_MU, sigma=TK_SIGMA, bypass_contextualizer=True, seed=0,
            att_heads=10, att_layer=2, att_ff_dim=300, max_length=200):
    install_shims()
    from matchmaker.models.published.ecai20_tk import ECAI20_TK

    class TK(ECAI20_TK):
        def get_range_vector(self, size, device):  # CPU branch of ecai20_tk.py:196-204
            return torch.arange(0, size, dtype=torch.long)

    class TKBypass(TK):
        def forward_representation(self, emb, mask, positional_features=None):
            return emb

    torch.manual_seed(seed)
    cls = TKBypass if bypass_contextualizer else TK
    m = cls(embsize, mu, sigma, att_heads, att_layer, att_ff_dim, max_length, True, True)
    m.eval()
    return m


def tk_forward(model, q, d, q_mask, d_mask, secondary=False):
    with torch.no_grad():
        return model.forward(q, d, q_mask, d_mask, secondary)


# --------------------------------------------------------------------------- TK-Sparse
def make_tk_sparse(embsize=300, mu=TK_MU, sigma=TK_SIGMA, bypass_contextualizer=True, seed=0,
                   att_heads=10, att_layer=2, att_ff_dim=300, max_length=200):
    """Real CIKM20_TK_Sparse (published/cikm20_tk_sparse.py).  bypass: forward_representation returns
    (emb * mask, emb) — the mask multiply of :170 kept, positional encoding + Transformer dropped — so the
    scoring block :106-146 and the stop-word MLP :132-133 run on known inputs."""
    install_shims()
    from matchmaker.models.published.cikm20_tk_sparse import CIKM20_TK_Sparse

    class TKS(CIKM20_TK_Sparse):
        def get_range_vector(self, size, device):  # CPU branch of cikm20_tk_sparse.py:237-245
            return torch.arange(0, size, dtype=torch.long)

    class TKSBypass(TKS):
        def forward_representation(self, emb, mask, positional_features=None):
            return emb * mask.unsqueeze(-1), emb

    torch.manual_seed(seed)
    cls = TKSBypass if bypass_contextualizer else TKS
    m = cls(embsize, mu, sigma, att_heads, att_layer, 32, att_ff_dim, max_length, True)
    m.eval()
    return m


# --------------------------------------------------------------------------- IDCM
def make_idcm(bert, sample_n=2, sample_context="ck-small", top_k_chunks=2, chunk_size=50, overlap=7,
              sample_train_type="mseloss", seed=0):
    """Real IDCM (published/sigir21_idcm.py) around a given (randomly initialised) DistilBERT."""
    install_shims()
    from matchmaker.models.published.sigir21_idcm import IDCM

    torch.manual_seed(seed)
    m = IDCM(bert, sample_train_type=sample_train_type, sample_n=sample_n, sample_context=sample_context,
             top_k_chunks=top_k_chunks, chunk_size=chunk_size, overlap=overlap, padding_idx=0)
    m.eval()
    return m


# --------------------------------------------------------------------------- KNRM
def make_knrm(n_kernels=11, seed=0):
    """Real KNRM (matchmaker/models/knrm.py); its constructor builds mu/sigma with torch.cuda.FloatTensor
    (:31-32), aliased to the CPU type by install_shims()."""
    install_shims()
    from matchmaker.models.knrm import KNRM

    torch.manual_seed(seed)
    m = KNRM(n_kernels)
    m.eval()
    return m


def knrm_forward(model, q, d, q_mask, d_mask, secondary=False):
    with torch.no_grad():
        return model.forward(q, d, q_mask, d_mask, secondary)


def make_conv_knrm(embsize=300, n_grams=3, n_kernels=11, conv_out_dim=128, seed=0):
    """Real Conv_KNRM (matchmaker/models/conv_knrm.py)."""
    install_shims()
    from matchmaker.models.conv_knrm import Conv_KNRM

    torch.manual_seed(seed)
    m = Conv_KNRM(embsize, n_grams, n_kernels, conv_out_dim)
    m.eval()
    return m


# --------------------------------------------------------------------------- TKL
def make_tkl(embsize=300, mu=TK_MU, sigma=TK_SIGMA, saturation_type="embedding",
             bypass_contextualizer=True, seed=0, att_heads=10, att_layer=2, att_ff_dim=300,
             max_length=2000):
    install_shims()
    from matchmaker.models.published.sigir20_tkl import TKL_sigir20

    class TKL(TKL_sigir20):
        def get_range_vector(self, size, device):  # CPU branch of sigir20_tkl.py:365-373
            return torch.arange(0, size, dtype=torch.long)

    class TKLBypass(TKL):
        # keeps the mask multiply of sigir20_tkl.py:306, drops positional enc + Transformer
        def forward_representation(self, emb, mask, positional_features=None):
            return emb * mask.unsqueeze(-1), emb

    torch.manual_seed(seed)
    cls = TKLBypass if bypass_contextualizer else TKL
    m = cls(embsize, mu, sigma, att_heads, att_layer, att_ff_dim, max_length, True, True,
            saturation_type)
    m.eval()
    return m


def tkl_forward(model, q, d, q_mask, d_mask, secondary=False):
    with torch.no_grad():
        return model.forward(q, d, q_mask, d_mask, secondary)
