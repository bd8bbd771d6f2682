"""Builds libmm_native.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m matchmaker_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the repo snapshot to
the GPU box.  No torch headers are involved: the library is a plain C-ABI shared object
(include/mm_native.h) bound from Python with ctypes (matchmaker_amd/_lib.py).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libmm_native.so")
SOURCES = ["common.hip", "maxsim.hip", "kernel_pool.hip", "kernel_pool_bwd.hip", "tkl.hip", "dot_topk.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-comment"]


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = _sources() + [os.path.join(CSRC, "mm_internal.h"),
                         os.path.join(HERE, "..", "include", "mm_native.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", LIB] + _sources()
    if verbose:
        print("[matchmaker_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
