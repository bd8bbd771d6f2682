import os, subprocess, sys, numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
def run(lib, env, seed=5):
    path = f"/tmp/f_{os.getpid()}_{abs(hash((lib, tuple(sorted(env.items())))))}.npy"
    e = dict(os.environ, **env)
    if lib: e["MM_NATIVE_LIB"] = os.path.join(root, lib)
    r = subprocess.run([sys.executable, "-c", f"from tests.test_kernel_pool_gpu import _multi_run; _multi_run({path!r}, {seed})"], cwd=root, env=e, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(path)
for lib in ("variants/libmm_native_direct.so", None):
    a = run(lib, {"MM_KP_MULTI_LOOP": "1"}); b = run(lib, {"MM_KP_MULTI_LOOP": "0"}); c = run(lib, {"MM_KP_MULTI_LOOP": "0", "MM_KP_MULTI_WG": "1"})
    d = np.abs(a.astype(np.float64) - b)
    print(lib, "loop vs flat: equal", a.tobytes() == b.tobytes(), "max abs diff", d.max(), "n differing", int((a != b).sum()), "of", a.size, "| flat vs wg equal", b.tobytes() == c.tobytes())
