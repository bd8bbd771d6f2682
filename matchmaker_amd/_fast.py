"""Loader of the optional host extension matchmaker_amd/csrc_host/_mm_autograd.so (C++ torch::autograd::Function around
mm_maxsim_fwd / mm_maxsim_bwd: the training step's node without Python in its backward; csrc_host/mm_autograd.cpp says why).

`module()` returns the initialised extension or None: not built (python -m matchmaker_amd.build builds it), not importable
against this torch, or switched off with MM_MAXSIM_PY_AUTOGRAD=1 (A/B runs, tests).  None means the Python
autograd.Function (colbert._MaxSimFn) runs — the same two kernels either way."""
import importlib.util
import os

from . import _lib

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc_host", "_mm_autograd.so")
_mod = None
_tried = False


def module():
    global _mod, _tried
    if not _tried:
        _tried = True
        if os.environ.get("MM_MAXSIM_PY_AUTOGRAD", "0") not in ("", "0") or not os.path.exists(_PATH):
            return None
        try:
            import torch  # noqa: F401  (its shared libraries must be loaded first)
            spec = importlib.util.spec_from_file_location("_mm_autograd", _PATH)
            m = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(m)
            _lib.lib()                      # the scoring library itself must load (fails loudly if it is missing)
            # the artefact ships with the snapshot: under a torch other than the one it was compiled against its autograd-context
            # layout may differ — refuse it BEFORE the first call (matchmaker_amd.build rebuilds on the same mismatch)
            built = m.built_torch_version() if hasattr(m, "built_torch_version") else None
            if built != torch.__version__.split("+")[0]:
                raise RuntimeError(f"built against torch {built}, running under {torch.__version__}")
            m.init(_lib.LIB_PATH)
            _mod = m
        except Exception as e:              # built against another torch, ...: keep the Python node, say so once
            import warnings
            warnings.warn(f"matchmaker_amd: host extension {_PATH} not usable ({e!r}); using the Python autograd node")
            _mod = None
    return _mod
