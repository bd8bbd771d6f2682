"""patch_matchmaker(): make an unmodified matchmaker checkout use the native scoring path.

matchmaker selects its models in matchmaker/models/all.py (:141-185) through names it imported from the model
modules.  Rebinding those names — in the defining modules and, if it is already imported, in models.all — is
all it takes: train.py / eval.py / dense_retrieval.py then build the drop-in classes from the same configs and
checkpoints.  Call it before `matchmaker.models.all.get_model(...)` runs.

    import matchmaker_amd.patch
    matchmaker_amd.patch.patch_matchmaker()
"""
import importlib
import sys

# (reference module, attribute) -> (our module, attribute)
_TABLE = [
    ("matchmaker.models.colbert", "ColBERT", "matchmaker_amd.colbert", "ColBERT"),
    ("matchmaker.models.colbert", "ColBERTConfig", "matchmaker_amd.colbert", "ColBERTConfig"),
    ("matchmaker.models.published.ecai20_tk", "ECAI20_TK", "matchmaker_amd.tk", "ECAI20_TK"),
    ("matchmaker.models.published.sigir20_tkl", "TKL_sigir20", "matchmaker_amd.tkl", "TKL_sigir20"),
    ("matchmaker.models.published.cikm20_tk_sparse", "CIKM20_TK_Sparse", "matchmaker_amd.tk_sparse", "CIKM20_TK_Sparse"),
    ("matchmaker.models.knrm", "KNRM", "matchmaker_amd.knrm", "KNRM"),
    ("matchmaker.models.conv_knrm", "Conv_KNRM", "matchmaker_amd.conv_knrm", "Conv_KNRM"),
    # IDCM is NOT rebound: only its passage sampler (sigir21_idcm.py:182-186) lies on the scoring path, the block is inline
    # in IDCM.forward (no module boundary to hook) and restating that forward here would be a copy of an out-of-scope method.
    # INTEGRATION.md shows the three-line edit a maintainer makes there to call matchmaker_amd.idcm.sampler_scores.
]


_idcm_note_given = False


def patch_matchmaker(strict: bool = False):
    """Rebinds the reference's model classes to the matchmaker_amd drop-ins.  Returns the list of
    "module.attribute" names that were rebound.  Modules that cannot be imported (optional dependencies of the
    reference that are not installed) are skipped unless strict=True.
    IDCM is not in the table (see above): when its module is importable a one-time warning says so — rounds 3-4 rebound it,
    and a caller relying on that would otherwise run the reference's eager sampler without notice."""
    global _idcm_note_given
    done = []
    for ref_mod, ref_attr, our_mod, our_attr in _TABLE:
        try:
            ref = importlib.import_module(ref_mod)
        except Exception:
            if strict:
                raise
            continue
        ours = getattr(importlib.import_module(our_mod), our_attr)
        setattr(ref, ref_attr, ours)
        done.append(ref_mod + "." + ref_attr)
        all_mod = sys.modules.get("matchmaker.models.all")      # `from ... import *` copies made earlier
        if all_mod is not None and hasattr(all_mod, ref_attr):
            setattr(all_mod, ref_attr, ours)
    if not _idcm_note_given:
        try:
            importlib.import_module("matchmaker.models.published.sigir21_idcm")
            import warnings
            warnings.warn("matchmaker_amd.patch_matchmaker: IDCM is NOT patched (its passage sampler is inline in IDCM.forward); "
                          "apply the three-line edit of INTEGRATION.md section 3 to call matchmaker_amd.idcm.sampler_scores",
                          stacklevel=2)
        except Exception:
            pass
        _idcm_note_given = True
    return done
