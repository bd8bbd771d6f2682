"""CPU tests of the host-side logic (no GPU, no compute through the HIP library): chunking,
positional features, drop-in constructors / state_dict keys vs the real reference classes, the C-ABI
library loading with every declared symbol, and loud failure on CPU tensors."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from oracle import ref_harness as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MU = [1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9]
SIGMA = [0.1] * 11


def test_library_builds_loads_and_exports_every_declared_symbol():
    from matchmaker_amd import build, _lib
    build.build()
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "mm_native.h")).read()
    declared = set(re.findall(r"\b(mm_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations found"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert hasattr(L, name), name
    assert L.mm_abi_version() == 4 == _lib.ABI_VERSION
    # size queries are pure host arithmetic: callable without a GPU
    assert L.mm_maxsim_workspace_bytes(10, 1, 32, 180, _lib.MASK_NONE, _lib.MASK_LEN_I32) == 0
    assert L.mm_maxsim_workspace_bytes(10, 1, 32, 180, _lib.MASK_I64, _lib.MASK_I64) > 0


def test_header_is_plain_c_and_a_c_client_links_the_library(tmp_path):
    """The drop-in boundary is a C ABI: include/mm_native.h must compile as C (gcc -std=c99, no C++ / torch types) and
    a C program must link libmm_native.so and call it — here the entry points that need no GPU: the ABI version, the
    host-side workspace arithmetic, and an argument error reported through mm_last_error()."""
    import shutil
    import subprocess
    from matchmaker_amd import build
    so = build.build()
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    src = tmp_path / "client.c"
    src.write_text(r"""
#include <stdio.h>
#include <string.h>
#include "mm_native.h"
int main(void) {
  if (mm_abi_version() != MM_ABI_VERSION || MM_ABI_VERSION != 4) return 1;
  if (mm_tkl_fwd_peaks(NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 4, 100, 52, 20, 300, 11, MM_TKL_SAT_EMBEDDING, NULL, 0, NULL) != MM_EINVAL) return 6;
  if (mm_maxsim_workspace_bytes(10, 1, 32, 180, MM_MASK_I64, MM_MASK_I64) == 0) return 2;
  if (mm_tkl_workspace_bytes(4, 100, 52, 20, 11) == 0) return 3;
  /* a null pointer is refused before anything touches the device */
  int e = mm_maxsim_fwd(NULL, NULL, NULL, MM_MASK_NONE, NULL, MM_MASK_NONE, NULL, 4, 1, 32, 180, 128, MM_BF16, MM_SIM_ROUND, NULL, 0, NULL);
  if (e != MM_EINVAL) return 4;
  if (strlen(mm_last_error()) == 0) return 5;
  printf("c client ok: %s\n", mm_last_error());
  return 0;
}
""")
    exe = tmp_path / "client"
    libdir = os.path.dirname(so)
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                        "-L", libdir, "-l:libmm_native.so", f"-Wl,-rpath,{libdir}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "c client ok" in r.stdout


def test_ops_reject_cpu_tensors_loudly():
    from matchmaker_amd import ops, NativeError
    q = torch.zeros(1, 4, 16)
    d = torch.zeros(2, 5, 16)
    with pytest.raises(NativeError):
        ops.maxsim(q, d, pairs_per_query=2)
    with pytest.raises(NativeError):
        ops.kernel_pool(q, d, None, None, torch.tensor(MU), torch.tensor(SIGMA), torch.ones(11), torch.ones(11), 2)


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "matchmaker_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+\.*oracle|import_module\(.*oracle|oracle[/.]np_oracle|libmm_oracle",
                                     src, re.M), f"{f} uses the oracle"
                assert "/root/reference" not in src, f


@pytest.mark.parametrize("D", [4, 5, 6, 20, 45, 46, 85, 333, 2048])
def test_chunk_documents_matches_oracle_chunking(D):
    from matchmaker_amd.tkl import chunk_documents
    g = torch.Generator().manual_seed(D)
    B, E = 3, 8
    d = torch.randn(B, D, E, generator=g)
    lens = torch.randint(1, D + 1, (B,), generator=g)
    lens[0] = D
    m = (torch.arange(D)[None] < lens[:, None]).float()
    chunks, cmask, slot, C = chunk_documents(d, m)
    rc, rm, rp, rC = O.tkl_chunk(d.numpy(), m.numpy())
    assert C == rC
    np.testing.assert_array_equal(slot.numpy(), np.nonzero(rp)[0])
    np.testing.assert_array_equal(chunks.numpy(), rc[rp])
    np.testing.assert_array_equal(cmask.numpy(), rm[rp])


def test_tkl_param_packing_layout():
    from matchmaker_amd.tkl import TKL_sigir20
    torch.manual_seed(0)
    m = TKL_sigir20(64, MU, SIGMA, 8, 1, 32, 2000, True, True, "embedding")
    with torch.no_grad():
        m.chunk_scoring.copy_(torch.arange(15.0).view(1, 15))
    v = m.pack_params()
    assert v.numel() == 4 * 11 + 13 + 15 + 64
    p = O.tkl_params_from_state(m.state_dict())
    np.testing.assert_array_equal(v[:11].numpy(), p["mu"])
    np.testing.assert_array_equal(v[22:33].numpy(), p["dense_w"])
    np.testing.assert_array_equal(v[44:46].numpy(), p["sat_w1"])
    assert float(v[46]) == p["sat_b1"] == 100.0
    np.testing.assert_array_equal(v[53:55].numpy(), p["ln_w"])
    np.testing.assert_array_equal(v[57:72].numpy(), np.arange(15.0))
    np.testing.assert_array_equal(v[72:].numpy(), p["emb_reduce_w"])
    v2 = m.pack_params()
    assert v2 is v                                   # cached
    with torch.no_grad():
        m.dense.weight.add_(1.0)
    assert m.pack_params() is not v                  # invalidated by an in-place update


@pytest.mark.skipif(not R.available(), reason="/root/reference not mounted (GPU box)")
def test_dropins_mirror_reference_state_dict_and_positions():
    from matchmaker_amd.tk import ECAI20_TK, sinusoid_positions
    from matchmaker_amd.tkl import TKL_sigir20
    ref_tk = R.make_tk(60, bypass_contextualizer=False, att_heads=6, att_ff_dim=32, max_length=50)
    mine = ECAI20_TK(60, MU, SIGMA, 6, 2, 32, 50, True, True)
    assert {k: tuple(v.shape) for k, v in ref_tk.state_dict().items()} == \
        {k: tuple(v.shape) for k, v in mine.state_dict().items()}
    torch.testing.assert_close(mine.positional_features_q, ref_tk.positional_features_q, rtol=0, atol=0)
    torch.testing.assert_close(mine.positional_features_d, ref_tk.positional_features_d, rtol=0, atol=0)
    mine.load_state_dict(ref_tk.state_dict())        # strict load works
    # odd dimension: zero column appended
    assert sinusoid_positions(7, 3).shape == (1, 3, 7) and float(sinusoid_positions(7, 3)[0, :, -1].abs().sum()) == 0

    ref_tkl = R.make_tkl(64, bypass_contextualizer=False, att_heads=8, att_ff_dim=32)
    mine_l = TKL_sigir20(64, MU, SIGMA, 8, 2, 32, 2000, True, True, "embedding")
    assert {k: tuple(v.shape) for k, v in ref_tkl.state_dict().items()} == \
        {k: tuple(v.shape) for k, v in mine_l.state_dict().items()}
    mine_l.load_state_dict(ref_tkl.state_dict())
    # the contextualiser half (stays PyTorch) is the same function as the reference's
    x = torch.randn(2, 9, 64)
    mk = torch.ones(2, 9)
    mk[1, 5:] = 0
    ref_tkl.eval(); mine_l.eval()
    with torch.no_grad():
        a, _ = ref_tkl.forward_representation(x, mk, ref_tkl.positional_features_q[:, :9, :])
        b, _ = mine_l.forward_representation(x, mk, mine_l.positional_features_q[:, :9, :])
    torch.testing.assert_close(a, b)


def test_colbert_dropin_constructs_offline_with_reference_keys():
    from transformers import BertConfig, BertModel
    from matchmaker_amd.colbert import ColBERT, ColBERTConfig
    enc = BertModel(BertConfig(hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64,
                               vocab_size=100))
    m = ColBERT(ColBERTConfig(bert_model="(injected)", compression_dim=16), bert_model=enc)
    keys = set(m.state_dict())
    assert "compressor.weight" in keys and "compressor.bias" in keys
    assert any(k.startswith("bert_model.") for k in keys)
    assert all(k.startswith(("bert_model.", "compressor.")) for k in keys)
    assert m.get_param_stats() == "ColBERT: / " and m.get_param_secondary() == {}
    toks = {"input_ids": torch.randint(1, 100, (2, 7)), "attention_mask": torch.ones(2, 7, dtype=torch.long)}
    with torch.no_grad():
        v = m.forward_representation(toks, sequence_type="doc_encode")
    assert v.shape == (2, 7, 16)


def test_split_bf16_numerics():
    """The pooling kernels feed the bf16 matrix pipe with fp32 operands split as hi + lo: three MFMAs per K step in the
    shipped TK / TKL kernels (no lo.lo: the emulation's default), four in the 64n-wide kernels (lolo=True).  Bound both
    schemes' cosine error against the fp64 oracle on the reference's golden TK inputs and on an adversarial batch of planted
    near-duplicates (cosine ~ 1, where the dropped lo.lo term biases the result): <= 2.5e-6 with lo.lo, <= 8e-6 without
    (the scores' 1e-3 tolerance is 2-3 orders above)."""
    from tests import util
    g = util.load("tk_q20_d200_e300.npz")
    B = g["d"].shape[0]
    q = np.repeat(g["q"], B, 0) if g["q"].shape[0] == 1 else g["q"]
    c64 = O.cosine_matrix(q, g["d"], np.float64)
    c32 = O.cosine_matrix(q, g["d"], np.float32)
    cs = O.cosine_matrix_split_bf16(q, g["d"], lolo=True)
    e32, es = np.abs(c32 - c64).max(), np.abs(cs - c64).max()
    assert es < 5e-7 and es < 4 * max(e32, 1e-7), (es, e32)
    rng = np.random.default_rng(0)
    q = rng.standard_normal((4, 20, 300)).astype(np.float32) * 3.0
    d = rng.standard_normal((4, 200, 300)).astype(np.float32)
    for b in range(4):
        for j in range(0, 200, 5):
            d[b, j] = q[b, j % 20] * (1 + 0.01 * j) + 0.02 * rng.standard_normal(300).astype(np.float32)
    es = np.abs(O.cosine_matrix_split_bf16(q, d, lolo=True) - O.cosine_matrix(q, d, np.float64)).max()
    assert es < 2.5e-6, es       # 2^-18 residual per operand; worst case = coherent near-duplicates
    # the TK pooling kernel's three-product form (no lo.lo): the missing term is a coherent ~2^-17^2 x E bias on exact
    # duplicates — a few 1e-6 on the cosine, three orders below what moves a score by the contract's 1e-3 (the widest RBF
    # kernel derivative is 1 / (sigma sqrt(e)) ~ 6 per unit cosine; the GPU suite checks every pair of 16 x 1000 at 1e-3)
    assert O.cosine_matrix_split_bf16.__defaults__ == (False,)      # the default models what ships (TK and TKL: three products)
    es3 = np.abs(O.cosine_matrix_split_bf16(q, d) - O.cosine_matrix(q, d, np.float64)).max()
    assert es3 < 8e-6, es3
    g3 = np.abs(O.cosine_matrix_split_bf16(np.repeat(g["q"], B, 0) if g["q"].shape[0] == 1 else g["q"], g["d"]) - c64).max()
    assert g3 < 5e-6, g3          # (the golden batch holds exact copies of query tokens: 3.3e-6 there, 5e-7 with lo.lo)
    # exact power-of-two scale invariance (what tests/test_kernel_pool_gpu.py asserts on the device)
    assert np.array_equal(O.cosine_matrix_split_bf16(q * 4, d * 0.5), O.cosine_matrix_split_bf16(q, d))


def test_token_store_reads_the_reference_layout(tmp_path):
    """TokenStore.load follows dense_retrieval.py:292-303: raw memmap blocks + doc_infos.npz; global row
    ranges must address exactly the rows the reference's `storage[file][start:end]` (:404) returns."""
    from matchmaker_amd.token_store import TokenStore, write_reference_store
    rng = np.random.default_rng(5)
    E, blk = 16, 64
    docs, ids = [], []
    for i in range(23):
        n = int(rng.integers(1, 20))
        x = rng.standard_normal((n + 2, E)).astype(np.float16)
        x[-2:] = 0                                   # padding rows, stripped by the writer (:244)
        docs.append(x)
        ids.append(f"doc{i}")
    write_reference_store(str(tmp_path), docs, ids, blk, "float16")
    st = TokenStore.load(str(tmp_path), E, "float16", blk, "cpu")
    # the reference's loader
    dfs = np.load(os.path.join(str(tmp_path), "doc_infos.npz"), allow_pickle=True)
    doc_infos = dfs.get("doc_infos")[()]
    filled = dfs.get("storage_filled_to_index")[()]
    storage = [np.memmap(os.path.join(str(tmp_path), f"token_reps_{f}.npy"), dtype=np.float16, mode="r",
                         shape=(blk, E))[: filled[f]] for f in range(len(filled))]
    assert len(storage) > 1                         # the synthetic store spans several files
    b, e = st.ranges(ids)
    for i, sid in enumerate(ids):
        f, a, z = doc_infos[sid]
        ref = np.asarray(storage[f][a:z])
        got = st.tokens[int(b[i]): int(e[i])].numpy()
        assert np.array_equal(ref, got) and np.array_equal(ref, docs[i][:-2])


def test_conv_knrm_block_decomposition_matches_the_real_class():
    """Conv_KNRM's concat + dense (conv_knrm.py:132-137) as a sum over (i, t) blocks of KNRM-style pooling with
    weight slices: checked on CPU with the drop-in's own (torch) convolutions + the numpy oracle against the
    real class's golden scores — the decomposition the native path (matchmaker_amd/conv_knrm.py) relies on."""
    from tests import util
    from matchmaker_amd.conv_knrm import Conv_KNRM
    g = util.load("conv_knrm_q12_d50_e64.npz")
    m = Conv_KNRM(64, 3, 11, 128)
    m.load_state_dict({k[len("param."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param.")})
    q, d = torch.from_numpy(g["q"]), torch.from_numpy(g["d"])
    with torch.no_grad():
        qg = [c(q.transpose(1, 2)).transpose(1, 2).numpy() for c in m.convolutions]
        dg = [c(d.transpose(1, 2)).transpose(1, 2).numpy() for c in m.convolutions]
    w = g["param.dense.weight"].reshape(-1)
    mu, sigma = m.mu.view(-1).numpy(), m.sigma.view(-1).numpy()
    assert abs(float(sigma[0]) - 1e-3) < 1e-9          # conv_knrm's exact-match sigma (KNRM: 1e-4)
    score = np.zeros(q.shape[0], np.float32)
    blk = 0
    for a in qg:
        for b in dg:
            score += O.knrm_kernel_pool(a, b, g["q_mask"], g["d_mask"], mu, sigma, w[blk * 11:(blk + 1) * 11])
            blk += 1
    np.testing.assert_allclose(score, g["score"], atol=2e-5, rtol=1e-4)


@pytest.mark.skipif(not R.available(), reason="needs the reference tree (config files + real classes)")
@pytest.mark.parametrize("model_name,model_yaml", [("TK", "tk.yaml"), ("TKL", "tkl.yaml"), ("TK_Sparse", "tk.yaml"),
                                                   ("knrm", None), ("conv_knrm", None)])
def test_from_config_on_the_reference_yaml_files_builds_the_same_parameters(model_name, model_yaml):
    """models/all.py:141-152 builds these models with Model.from_config(config, word_embedding_dim) from the merged
    YAML files under config/train/.  The drop-ins must accept the very same dictionaries and end up with the same
    parameters / buffers (names and shapes) as the real classes: checkpoints and param-group routing by name
    (train.py:119-137) keep working."""
    import os
    import yaml
    R.install_shims()
    cfg = {}
    files = ["defaults.yaml", "non-bert-defaults.yaml"] + ([os.path.join("models", model_yaml)] if model_yaml else [])
    for f in files:
        with open(os.path.join(R.REFERENCE_ROOT, "config", "train", f)) as fh:
            cfg.update(yaml.safe_load(fh) or {})
    cfg.setdefault("tk_att_proj_dim", 32)               # read by CIKM20_TK_Sparse.from_config, absent from the YAML files
    emb_dim = 300
    if model_name == "TK":
        from matchmaker.models.published.ecai20_tk import ECAI20_TK as Ref
        from matchmaker_amd.tk import ECAI20_TK as Mine
    elif model_name == "TKL":
        from matchmaker.models.published.sigir20_tkl import TKL_sigir20 as Ref
        from matchmaker_amd.tkl import TKL_sigir20 as Mine
    elif model_name == "TK_Sparse":
        from matchmaker.models.published.cikm20_tk_sparse import CIKM20_TK_Sparse as Ref
        from matchmaker_amd.tk_sparse import CIKM20_TK_Sparse as Mine
    elif model_name == "knrm":
        from matchmaker.models.knrm import KNRM as Ref
        from matchmaker_amd.knrm import KNRM as Mine
    else:
        from matchmaker.models.conv_knrm import Conv_KNRM as Ref
        from matchmaker_amd.conv_knrm import Conv_KNRM as Mine

    # the reference builds its range vectors with torch.cuda.LongTensor: use the CPU branch for the test
    saved = Ref.__dict__.get("get_range_vector")
    if saved is not None:
        Ref.get_range_vector = lambda self, size, device: torch.arange(0, size, dtype=torch.long)
    try:
        ref = Ref.from_config(cfg, emb_dim)
    finally:
        if saved is not None:
            Ref.get_range_vector = saved
    mine = Mine.from_config(cfg, emb_dim)
    shapes = lambda m: {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert shapes(ref) == shapes(mine)
    assert {k for k, _ in ref.named_parameters()} == {k for k, _ in mine.named_parameters()}
    mine.load_state_dict(ref.state_dict(), strict=True)


@pytest.mark.skipif(not R.available(), reason="needs the reference tree")
def test_patch_matchmaker_rebinds_the_model_classes():
    from matchmaker_amd import patch
    R.install_shims()
    import importlib
    saved = {}
    for ref_mod, ref_attr, _, _ in patch._TABLE:
        try:
            saved[(ref_mod, ref_attr)] = getattr(importlib.import_module(ref_mod), ref_attr)
        except Exception:
            pass
    try:
        done = patch.patch_matchmaker()
        assert "matchmaker.models.published.ecai20_tk.ECAI20_TK" in done and "matchmaker.models.colbert.ColBERT" in done
        import matchmaker.models.published.ecai20_tk as ref_tk
        import matchmaker.models.published.sigir20_tkl as ref_tkl
        from matchmaker_amd.tk import ECAI20_TK
        from matchmaker_amd.tkl import TKL_sigir20
        assert ref_tk.ECAI20_TK is ECAI20_TK and ref_tkl.TKL_sigir20 is TKL_sigir20
        m = ref_tk.ECAI20_TK.from_config({"tk_kernels_mu": MU, "tk_kernels_sigma": SIGMA, "tk_att_heads": 10, "tk_att_layer": 2,
                                          "tk_att_ff_dim": 100, "max_doc_length": 200, "tk_use_diff_posencoding": True,
                                          "tk_mix_hybrid_context": True}, 300)
        assert type(m).__module__ == "matchmaker_amd.tk"
    finally:                                             # other tests drive the real classes
        for (ref_mod, ref_attr), obj in saved.items():
            setattr(importlib.import_module(ref_mod), ref_attr, obj)


def test_rerank_helpers_follow_eval_py_and_core_metrics():
    """matchmaker_amd.rerank: evaluate_batches unrolls scores per query id exactly as eval.py:189-203 does (arrival order
    kept), and unrolled_to_ranked_result is the reference's ranking rule — compared with the REAL function of
    utils/core_metrics.py:502-511 when the reference tree is mounted (ties keep arrival order: Python's sort is stable)."""
    import torch
    from matchmaker_amd import rerank

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, query, document, use_fp16=True, output_secondary_output=False):
            s = document["input_ids"].float().sum(-1) - query["input_ids"].float().sum(-1)
            return (s, {}) if output_secondary_output else s

    g = torch.Generator().manual_seed(3)
    batches, want = [], {}
    for b in range(4):
        n = 5
        qid = [f"q{(b * n + i) // 7}" for i in range(n)]
        did = [f"d{b * n + i}" for i in range(n)]
        q = torch.randint(0, 3, (n, 4), generator=g)
        d = torch.randint(0, 3, (n, 6), generator=g)          # small range -> plenty of exact ties
        batches.append({"query_tokens": {"input_ids": q}, "doc_tokens": {"input_ids": d}, "query_id": qid, "doc_id": did})
        for i in range(n):
            want.setdefault(qid[i], []).append((did[i], float(d[i].sum() - q[i].sum())))
    got = rerank.evaluate_batches(Model(), batches, use_fp16=False, device="cpu")
    assert got == want
    got2 = rerank.evaluate_batches(Model(), batches, use_fp16=False, device="cpu", output_secondary_output=True)
    assert got2 == want
    ranked = rerank.unrolled_to_ranked_result(got)
    for qid, rows in want.items():
        assert ranked[qid] == [d for d, _ in sorted(rows, key=lambda x: x[1], reverse=True)]
    from oracle import ref_harness as R
    if R.available():
        import importlib.util, os
        spec = importlib.util.spec_from_file_location("ref_core_metrics", os.path.join(R.REFERENCE_ROOT, "matchmaker", "utils", "core_metrics.py"))
        mod = importlib.util.module_from_spec(spec)
        try:
            spec.loader.exec_module(mod)
        except Exception:
            return                                             # optional dependencies of the reference missing here
        assert mod.unrolled_to_ranked_result(got) == ranked


def test_grouped_evaluation_host_logic_groups_by_shape_and_keeps_arrival_order():
    """rerank.evaluate_batches(score_group=N) — the host side of the batched scoring entry (mm_maxsim_fwd_batched): the encoder is
    called batch by batch, score_batches once per group of <= N consecutive batches of ONE shape (a shape change closes the
    group early: eval.py pads every batch to its own longest sequence), the results are the batch-by-batch loop's, in arrival
    order; a model without score_batches, secondary output, or graph=True keep the per-batch route."""
    import torch
    from matchmaker_amd import rerank
    calls = []

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward_representation(self, tokens, sequence_type=None):
            return tokens["input_ids"].float().unsqueeze(-1)

        def score_batches(self, batches):
            calls.append([tuple(q.shape[1:]) + tuple(d.shape[1:]) for q, d, _, _ in batches])
            return [d.sum((1, 2)) - q.sum((1, 2)) for q, d, _, _ in batches]

        def forward(self, query, document, use_fp16=True, output_secondary_output=False):
            s = document["input_ids"].float().sum(-1) - query["input_ids"].float().sum(-1)
            return (s, {}) if output_secondary_output else s

    g = torch.Generator().manual_seed(9)
    batches = []
    for b, (n, Lq, Ld) in enumerate([(5, 4, 6)] * 5 + [(5, 3, 6)] + [(5, 4, 6)] * 2 + [(2, 4, 6)]):
        mk = lambda L: {"input_ids": torch.randint(0, 3, (n, L), generator=g), "attention_mask": torch.ones(n, L, dtype=torch.long)}
        batches.append({"query_tokens": mk(Lq), "doc_tokens": mk(Ld), "query_id": [f"q{(b * 5 + i) // 7}" for i in range(n)],
                        "doc_id": [f"d{b}_{i}" for i in range(n)]})
    want = rerank.evaluate_batches(Model(), batches, use_fp16=False, device="cpu")
    assert not calls
    got = rerank.evaluate_batches(Model(), batches, use_fp16=False, device="cpu", score_group=4)
    assert got == want
    # nine batches: four, then one alone (the fifth; the sixth has another shape), the odd shape alone, then the last three
    assert [len(c) for c in calls] == [4, 1, 1, 3]
    assert all(len(set(c)) == 1 for c in calls)
    calls.clear()
    assert rerank.evaluate_batches(Model(), batches, use_fp16=False, device="cpu", score_group=4, output_secondary_output=True) == want
    assert not calls                                         # secondary output: the per-batch route


@pytest.mark.parametrize("Bq,Bd,NQT", [(32, 32, 4), (1030, 37, 4), (3, 5, 2), (2, 1, 4), (1024, 1024, 4), (7, 20000, 4), (513, 9, 2),
                                       (130, 70, 4)])
def test_tiled_all_pairs_work_map_covers_every_pair_once(Bq, Bd, NQT):
    """Python mirror of launch_stream_inb_tiled / the INB == 2 prologue of maxsim_stream_body (csrc/maxsim.hip): workgroup
    (xcd, t, jg) takes document slice xcd * T + t of 8T and the query groups jg, jg + Gw, ...  Every (query, document)
    pair must be produced exactly once, whatever the sizes."""
    cus = 256
    G = (Bq + NQT - 1) // NQT
    target = cus * 4 // 8
    gw = min(G, target)
    T = max(1, min(target // gw, (Bd + 7) // 8))
    seen = np.zeros((Bq, Bd), dtype=np.int32)
    for bid in range(8 * T * gw):
        xcd, rest = bid & 7, bid >> 3
        t, g0 = rest % T, rest // T
        S, si = 8 * T, xcd * T + t
        d_first = Bd * si // S
        nd = Bd * (si + 1) // S - d_first
        ng = (G - g0 + gw - 1) // gw if g0 < G else 0
        for v in range(ng * nd):
            grp = g0 + (v // nd) * gw
            doc = d_first + v % nd
            for n in range(NQT):
                qq = grp * NQT + n
                if qq < Bq:
                    seen[qq, doc] += 1
    assert seen.min() == 1 and seen.max() == 1


@pytest.mark.parametrize("Bq,Bd,NQT", [(64, 64, 4), (1024, 1024, 4), (70, 90, 2), (1030, 137, 4), (65, 20000, 4), (4000, 64, 4)])
def test_shared_ring_all_pairs_work_map_covers_every_pair_once(Bq, Bd, NQT):
    """Python mirror of launch_inb_wg / maxsim_allpairs_wg_kernel (csrc/maxsim.hip): workgroup (xcd, t, jg) streams document
    slice xcd * T + t once per query group jg, jg + Gw, ...; wavefront w of the workgroup owns queries
    (4 g + w) * NQT .. + NQT - 1.  Every (query, document) pair exactly once."""
    cus = 256
    G = (Bq + 4 * NQT - 1) // (4 * NQT)
    target = cus * 2 // 8
    gw = min(G, target)
    T = max(1, min(target // gw, (Bd + 7) // 8))
    seen = np.zeros((Bq, Bd), dtype=np.int32)
    for bid in range(8 * T * gw):
        xcd, rest = bid & 7, bid >> 3
        t, jg = rest % T, rest // T
        S, si = 8 * T, xcd * T + t
        d_first = Bd * si // S
        nd = Bd * (si + 1) // S - d_first
        if nd <= 0 or jg >= G:
            continue
        for g in range(jg, G, gw):
            for w in range(4):
                for n in range(NQT):
                    qq = (g * 4 + w) * NQT + n
                    if qq < Bq:
                        seen[qq, d_first:d_first + nd] += 1
    assert seen.min() == 1 and seen.max() == 1


def test_store_burst_accounting_never_overcounts():
    """Model of the vmcnt bookkeeping of tkl_stage1_run_kernel / maxsim_allpairs_wg_kernel: a wavefront's loads (slices of
    `per` instructions) and its bursts of stores retire in issue order; the wait for the oldest slice is
    vmcnt(per * (inflight - 1) + nst) while `pre` > 0, else vmcnt(per * (inflight - 1)).  The value must never EXCEED the
    true number of younger operations (that would read a slice before it landed) and should equal it whenever at most one
    burst is in the queue."""
    rng = np.random.default_rng(5)
    for per, nbuf, ns in ((13, 3, 3), (13, 3, 1), (2, 3, 1), (2, 4, 2)):
        queue = []                       # issue-ordered: ("slice", id) x per | ("store",)
        inflight, pre, nst, next_id, oldest = 0, 0, 0, 0, 0
        exact = total = 0

        def top_up():
            nonlocal inflight, next_id
            while inflight < nbuf:
                queue.extend([("slice", next_id)] * per)
                next_id += 1
                inflight += 1
        top_up()
        for block in range(200):
            for s in range(ns):
                top_up()
                if pre > 0:
                    n = per * (inflight - 1) + nst
                    pre -= 1
                else:
                    nst = 0
                    n = per * (inflight - 1)
                # true number of operations younger than the last instruction of the oldest slice
                last = max(i for i, op in enumerate(queue) if op == ("slice", oldest))
                younger = len(queue) - 1 - last
                assert n <= younger, (per, nbuf, ns, block, s, n, younger)
                exact += n == younger
                total += 1
                queue = queue[last + 1:]                 # everything up to it has retired (in order)
                oldest += 1
                inflight -= 1
                top_up()                                 # early hand-back of the slot
            burst = int(rng.integers(0, 17))
            if burst:
                queue.extend([("store",)] * burst)
                nst, pre = burst, inflight
        if (per, nbuf, ns) == (13, 3, 3):   # TKL at E = 300: at most one burst in the queue -> the count is exact
            assert exact >= 0.9 * total, (per, nbuf, ns, exact, total)   # (elsewhere an older burst is left uncounted on purpose)


def test_tkl_backward_window_list_in_region_order():
    """Model of tkl_bwd_tiled_kernel's window bookkeeping (csrc/tkl_bwd.hip): the 15 windows of sigir20_tkl.py:275-277
    (j = peaks, -1, +1, -2, +2 around the three peaks, clamped to the document) are processed in REGION order — rank
    (j % 3) * 5 + j // 3 among the non-empty ones — d loss / d c is summed per position over a region's windows, and a
    region reads the chunk rows it adds to only when it shares a position with an EARLIER region.  Checked here: every
    non-empty window appears exactly once, regions are contiguous runs of the list, a window's positions lie inside its
    region's range (<= 38 positions = two blocks of 32), and the overlap flag is exactly 'intersects an earlier region'."""
    rng = np.random.default_rng(9)
    offs = [0, -1, 1, -2, 2]
    for trial in range(400):
        Wp = int(rng.integers(3, 200))
        # three peaks as the region search finds them: >= 15 windows apart while the document is long enough, anywhere else
        work = rng.random(Wp)
        top = []
        for c in range(3):
            b = int(np.argmax(work))
            top.append(b)
            work[np.abs(np.arange(Wp) - b) < 15] = -1.0 - c
        empty = rng.random(Wp) < 0.15                          # windows whose forward score is exactly 0 (:282)
        idx = [min(max(top[j % 3] + offs[j // 3], 0), Wp - 1) for j in range(15)]
        valid = [not empty[i] for i in idx]
        # device: compact list by rank, region ranges, overlap flags
        rank = [(j % 3) * 5 + j // 3 for j in range(15)]
        wlist = [j for _, j in sorted((rank[j], j) for j in range(15) if valid[j])]
        lo = [min([idx[r + 3 * o] for o in range(5) if valid[r + 3 * o]], default=None) for r in range(3)]
        hi = [max([idx[r + 3 * o] for o in range(5) if valid[r + 3 * o]], default=None) for r in range(3)]
        p0 = [2 * lo[r] if lo[r] is not None else 0 for r in range(3)]
        p1 = [2 * hi[r] + 30 if hi[r] is not None else 0 for r in range(3)]
        ov = [int(any(p1[o] > p0[o] and p1[r] > p0[r] and p0[o] < p1[r] and p0[r] < p1[o] for o in range(r))) for r in range(3)]
        # properties
        assert sorted(wlist) == [j for j in range(15) if valid[j]]
        regions = [j % 3 for j in wlist]
        assert regions == sorted(regions)                          # contiguous runs, region 0 first
        for j in wlist:
            r = j % 3
            assert p1[r] - p0[r] <= 38 and p0[r] <= 2 * idx[j] and 2 * idx[j] + 30 <= p1[r]
        touched = set()
        for r in range(3):
            mine = set(range(p0[r], p1[r]))
            assert ov[r] == int(bool(mine & touched)), (trial, r, p0, p1)
            touched |= mine


def test_flat_index_precision_follows_token_dtype():
    """base_index.py:14: use_fp16 = config["token_dtype"] == "float16"; an fp32 index is refused, not silently rounded."""
    from matchmaker_amd.retrieval import FlatIPIndexer
    from matchmaker_amd.ops import NativeError
    FlatIPIndexer({"token_dim": 128, "token_dtype": "float16"}, device="cpu", topk_fn=lambda *a: None, merge_fn=lambda *a: None)
    with pytest.raises(NativeError):
        FlatIPIndexer({"token_dim": 128, "token_dtype": "float32"}, device="cpu", topk_fn=lambda *a: None, merge_fn=lambda *a: None)
    with pytest.raises(NativeError):
        FlatIPIndexer({"token_dim": 128, "token_dtype": "float16", "faiss_use_fp16": False}, device="cpu",
                      topk_fn=lambda *a: None, merge_fn=lambda *a: None)


@pytest.mark.parametrize("sat", ["embedding", "log"])
def test_tkl_parameters_unused_by_the_saturation_mode_get_no_gradient_slot(sat):
    """The reference leaves .grad = None on parameters its active saturation never reads (kernel_mult under "embedding";
    the saturation layers, LayerNorm and sat_emb_reduce1 under "log"); the native backward's layout must say so too."""
    from matchmaker_amd.tkl import TKL_sigir20
    m = TKL_sigir20(64, [1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9], [0.1] * 11, 8, 1, 32, 2000, True, True, sat)
    scoring, sizes = m._pack_layout()
    live = {id(p) for p in m._scoring_parameters()}
    for t, n in zip(scoring, sizes):
        assert (n is not None) == (id(t) in live)
    assert sum(n is None for n in sizes) == (1 if sat == "embedding" else 9)


def test_stream_handle_is_read_once_and_the_public_path_is_the_fallback(monkeypatch):
    """ops._stream: the raw handle of the current stream through torch._C._cuda_getCurrentRawStream (one C call), the
    public torch.cuda.current_stream() when that symbol is missing; ops._workspace keys its cache by the handle it is given."""
    from matchmaker_amd import ops
    dev = torch.device("cuda", 0)
    calls = []
    monkeypatch.setattr(ops, "_RAW_STREAM", lambda idx: calls.append(idx) or 0x1234)
    assert ops._stream(dev) == 0x1234 and calls == [0]

    class _S:
        cuda_stream = 0x77
    monkeypatch.setattr(ops, "_RAW_STREAM", None)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda d=None: _S())
    assert ops._stream(dev) == 0x77
    # workspace cache: one buffer per (device, stream handle), reused while large enough
    made = []
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    monkeypatch.setattr(torch, "empty", lambda n, dtype=None, device=None: made.append(n) or torch.zeros(n, dtype=torch.uint8))
    monkeypatch.setattr(ops, "_WS", {})
    a = ops._workspace(dev, 100, 5)
    b = ops._workspace(dev, 4096, 5)
    c = ops._workspace(dev, 100, 6)
    assert a is b and c is not a and made == [1 << 16, 1 << 16]
    assert ops._workspace(dev, 0, 5) is None


@pytest.mark.parametrize("fname", [f for f in __import__("tests.util", fromlist=["x"]).golden_files("tkl_") if "embedding" in f])
def test_tkl_secondary_outputs_host_part_matches_the_real_class(fname):
    """sigir20_tkl.py:288-292 returns top_non_overlapping_idx / top_k_non_overlapping / sat_influence_from_top_k.  The kernels
    supply the window scores and the three peaks; everything else is host arithmetic (matchmaker_amd/tkl.py `_secondary`,
    `region_peaks`, `region_neighbors`) — checked here against the REAL class's outputs (tests/golden/gen_golden.py gen_tkl),
    feeding it the real class's own window scores in place of the kernels'."""
    from tests import util
    from matchmaker_amd.tkl import TKL_sigir20, region_peaks, region_neighbors, chunk_documents
    g = util.load(fname)
    state = {k[len("param."):]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("param.")}
    E = g["q"].shape[-1]
    m = TKL_sigir20(E, [1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9], [0.1] * 11, 8 if E % 10 else 10, 1, 32, 2000,
                    True, True, "embedding")
    missing, unexpected = m.load_state_dict(state, strict=False)
    assert not unexpected
    q, d = torch.from_numpy(g["q"]).float(), torch.from_numpy(g["d"]).float()
    qm, dm = torch.from_numpy(g["q_mask"]), torch.from_numpy(g["d_mask"])
    win = torch.from_numpy(g["orig_score"]).float()
    B = q.shape[0]
    _, _, _, C = chunk_documents(d, dm)
    W = (max(C * 40, 30) - 30) // 2 + 1
    assert win.shape[1] == max(W, 3)
    peaks = region_peaks(win)
    assert torch.equal(peaks, torch.from_numpy(g["top_idx"]))                         # :266-271, integer work: exact
    nb = region_neighbors(peaks, win.shape[1])
    assert int(nb.min()) >= 0 and int(nb.max()) <= win.shape[1] - 1
    q_ctx = q * qm.unsqueeze(-1)                                                     # the golden bypasses the contextualiser
    sec = m._secondary(torch.from_numpy(g["score"]), win, peaks, q_ctx, qm, dm, B, C, 0)
    assert torch.equal(sec["top_non_overlapping_idx"], torch.from_numpy(g["top_idx"]))
    np.testing.assert_array_equal(sec["top_k_non_overlapping"].numpy(), g["top_k_non_overlapping"])   # a gather: exact
    np.testing.assert_allclose(sec["sat_influence_from_top_k"].numpy(), g["sat_influence_from_top_k"], atol=2e-5, rtol=1e-5)
    assert sec["sat_influence_from_top_k"].shape == (B, 15, q.shape[1], 2)
    np.testing.assert_array_equal(sec["orig_doc_len"].numpy(), g["d_mask"].sum(-1))


def test_tkl_region_tie_policy_classifier():
    """tests/util.tkl_region_classify is what the GPU rank test trusts to tell "another region of equal score" from "a wrong
    region": pinned here on hand-made window rows and against the real class's own peaks (tkl goldens)."""
    from tests import util
    from matchmaker_amd.tkl import region_peaks
    for f in util.golden_files("tkl_"):
        g = util.load(f)
        if "top_idx" in g:
            for b in range(g["orig_score"].shape[0]):
                assert util.tkl_region_search(g["orig_score"][b]) == g["top_idx"][b].tolist()
    rng = np.random.default_rng(5)
    w = rng.uniform(0.1, 1.0, 200)
    w[[20, 90, 150]] = [5.0, 4.0, 3.0]
    assert util.tkl_region_search(w) == [20, 90, 150]
    assert util.tkl_region_search(w) == region_peaks(torch.from_numpy(w)[None])[0].tolist()
    assert util.tkl_region_classify(w, [20, 90, 150], 1e-5) == ("same", 0.0)
    w2 = w.copy()
    w2[60] = 4.0 - 3e-6                                   # a second candidate 3e-6 below the round-2 peak
    assert util.tkl_region_search(w2) == [20, 90, 60]
    cls, gap = util.tkl_region_classify(w2, [20, 60, 90], 1e-5)      # the device took 60 first, then 90: both within the noise
    assert cls == "tied" and abs(gap - 3e-6) < 1e-9
    cls, gap = util.tkl_region_classify(w2, [20, 60, 150], 1e-5)     # ... but skipping 90 (1.0 above 150) is wrong
    assert cls == "wrong" and abs(gap - 1.0) < 1e-6
    assert util.tkl_region_classify(w2, [20, 60, 90], 1e-6)[0] == "wrong"      # ... and so is a 3e-6 gap at 1e-6 of noise
    assert util.tkl_region_classify(w2, [20, 25, 90], 1e-5)[0] == "wrong"      # a suppressed window
    assert util.tkl_region_classify(w, [90, 20, 150], 1e-5)[0] == "wrong"      # same set, wrong order: weights differ (:286)
    # exact ties go to the lowest index on both sides; all-empty rows pick 0, 15, 30 (:257, :268-273)
    assert util.tkl_region_search(np.zeros(100)) == [0, 15, 30]
    assert util.tkl_region_search(np.zeros(20)) == [0, 15, 0]
    assert util.tkl_region_search(np.zeros(6)) == [0, 0, 0]
    z = torch.zeros(1, 20)
    assert region_peaks(z)[0].tolist() == [0, 15, 0]
    cs = np.arange(1, 16, dtype=np.float64)
    assert abs(util.tkl_score_at(w, [20, 90, 150], cs) -
               sum(cs[i] * w[j] for i, j in enumerate([20, 90, 150, 19, 89, 149, 21, 91, 151, 18, 88, 148, 22, 92, 152]))) < 1e-12


@pytest.mark.parametrize("groups,n_mq,n_md", [(1024, 3, 3), (2048, 3, 3), (7, 3, 3), (1, 1, 1), (33, 2, 4), (100, 4, 1), (2047, 3, 2)])
def test_flat_xcd_grouped_multi_launch_covers_every_combination_once(groups, n_mq, n_md):
    """Python mirror of kp_block_args' flat order (csrc/kp_device.h, round 5; grid from kp128_launch's flat_grid): workgroup g of
    the multi launch (Conv-KNRM's n_mq x n_md match matrices) -> (pair range x, combination y = i * n_md + t).  Every (x, y)
    exactly once; padding workgroups leave; and the n_mq workgroups that read the same pair range of the same DOCUMENT tensor
    have ids congruent modulo 8 (same XCD under the observed b % 8 placement) within one window of 8 * n_mq ids."""
    flat = (groups * n_md + 7) // 8 * 8
    n_wg = flat * n_mq
    period = 8 * n_mq
    seen = np.zeros((groups, n_mq * n_md), dtype=np.int32)
    readers = {}
    for g in range(n_wg):
        blk, r = divmod(g, period)
        fr = blk * 8 + (r & 7)
        x, t = divmod(fr, n_md)
        y = (r >> 3) * n_md + t
        if x >= groups:
            continue                                   # block_x = 0x3fffffff: p0 >= n_pairs, the workgroup returns
        seen[x, y] += 1
        readers.setdefault((x, t), []).append(g)
    assert seen.min() == 1 and seen.max() == 1
    for ids in readers.values():
        assert len(ids) == n_mq and len({g % 8 for g in ids}) == 1 and max(ids) - min(ids) < period


def test_host_extension_builds_and_loads_against_this_torch():
    """matchmaker_amd/csrc_host/_mm_autograd.so (the C++ autograd node, optional host plumbing): builds with g++ against the
    installed torch, imports, resolves the C-ABI entry points of the library the ctypes binding loaded, and refuses CPU tensors
    like every operator; MM_MAXSIM_PY_AUTOGRAD=1 switches it off (the Python node is the fallback)."""
    import subprocess
    import sys
    from matchmaker_amd import build
    so = build.build_host()
    assert so and os.path.exists(so)
    code = """
import torch
from matchmaker_amd import _fast
m = _fast.module()
print('MOD', m is not None)
try:
    m.maxsim_paired(torch.zeros(2, 3, 8), torch.zeros(2, 4, 8), None, None, 0)
except RuntimeError as e:
    print('REFUSED', 'HIP device' in str(e))
"""
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert "MOD True" in r.stdout and "REFUSED True" in r.stdout, (r.stdout, r.stderr[-2000:])
    r = subprocess.run([sys.executable, "-c", "from matchmaker_amd import _fast; print('MOD', _fast.module() is not None)"], cwd=ROOT,
                       env=dict(os.environ, MM_MAXSIM_PY_AUTOGRAD="1"), capture_output=True, text=True, timeout=300)
    assert "MOD False" in r.stdout, (r.stdout, r.stderr[-2000:])
