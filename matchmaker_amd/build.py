"""Builds libmm_native.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m matchmaker_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the repo snapshot to
the GPU box.  No torch headers are involved: the library is a plain C-ABI shared object
(include/mm_native.h) bound from Python with ctypes (matchmaker_amd/_lib.py).
Each translation unit is compiled to an object file (in parallel, only when it or a header changed),
then linked.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libmm_native.so")
SOURCES = ["common.hip", "maxsim.hip", "maxsim_pair.hip", "kernel_pool.hip", "kernel_pool128.hip", "kernel_pool_bwd.hip", "kernel_pool_bwd_split.hip", "tkl.hip", "tkl_stage1_rows.hip", "tkl_stage1_ksplit.hip", "tkl_bwd.hip", "dot_topk.hip"]
HEADERS = [os.path.join(CSRC, "mm_internal.h"), os.path.join(CSRC, "kp_device.h"), os.path.join(CSRC, "maxsim_device.h"), os.path.join(CSRC, "kp_bwd.h"), os.path.join(HERE, "..", "include", "mm_native.h")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment"]


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _obj(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = _sources()
    os.makedirs(OBJ, exist_ok=True)
    todo = [s for s in srcs if force or _newer(_obj(s), [s] + HEADERS)]

    def compile_one(src):
        cmd = [hipcc] + CFLAGS + ["-c", "-o", _obj(src), src]
        if verbose:
            print("[matchmaker_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)

    if todo:
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 1)) as ex:
            list(ex.map(compile_one, todo))
    objs = [_obj(s) for s in srcs]
    relink = bool(force or todo or _newer(LIB, objs))
    # unambiguous in the build log: the objects / .so are git-ignored but ship with the snapshot, so whether a
    # call compiled anything depends on mtimes
    print(f"[matchmaker_amd.build] recompiled {len(todo)} of {len(srcs)} TUs"
          + (f" ({', '.join(os.path.basename(t) for t in todo)})" if todo and len(todo) < len(srcs) else "")
          + f"; {'relinked' if relink else 'kept'} {os.path.relpath(LIB, os.path.join(HERE, '..'))}"
          + (" [--force]" if force else ""), flush=True)
    if relink:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[matchmaker_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


# ---- host plumbing: the C++ autograd node around the C ABI (csrc_host/mm_autograd.cpp) ------------------------------------
HOST_SRC = os.path.join(HERE, "csrc_host", "mm_autograd.cpp")
HOST_LIB = os.path.join(HERE, "csrc_host", "_mm_autograd.so")


def build_host(force: bool = False, verbose: bool = True):
    """g++ -> matchmaker_amd/csrc_host/_mm_autograd.so (a torch extension: needs torch's headers, ~30 s).  Optional: the
    scoring path does not depend on it (matchmaker_amd/_fast.py falls back to the Python autograd.Function); returns the
    path, or None when it could not be built."""
    stamp = HOST_LIB + ".torch"
    try:
        import torch as _t
        cur = _t.__version__
    except Exception:
        cur = None
    same_torch = os.path.exists(stamp) and open(stamp).read().strip() == str(cur)
    if not force and same_torch and not _newer(HOST_LIB, [HOST_SRC, os.path.join(HERE, "..", "include", "mm_native.h")]):
        if verbose:
            print(f"[matchmaker_amd.build] kept {os.path.relpath(HOST_LIB, os.path.join(HERE, '..'))}", flush=True)
        return HOST_LIB
    try:
        import sysconfig
        import torch
        from torch.utils import cpp_extension as ce
        inc = ce.include_paths() + [sysconfig.get_paths()["include"], "/opt/rocm/include", os.path.join(HERE, "..", "include")]
        lib = ce.library_paths()[0]
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
               "-DTORCH_EXTENSION_NAME=_mm_autograd", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
               "-Wno-deprecated-declarations"] + [f"-I{p}" for p in inc] + \
              [HOST_SRC, "-o", HOST_LIB, f"-L{lib}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python",
               f"-Wl,-rpath,{lib}", "-ldl"]
        if verbose:
            print("[matchmaker_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(stamp, "w") as f:      # the torch it was compiled against (a snapshot may land beside another one: rebuild then)
            f.write(torch.__version__)
        return HOST_LIB
    except Exception as e:      # no torch headers / no compiler: the Python node keeps working
        print(f"[matchmaker_amd.build] host extension NOT built ({e!r}): the Python autograd.Function stays in use", flush=True)
        return None


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
    print(build_host(force="--force" in sys.argv))
