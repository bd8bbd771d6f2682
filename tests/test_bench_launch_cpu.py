"""CPU: `python bench.py --gpus N` starts its own ranks (north_star: "q-d pairs/sec reported at 1/2/4/8 GPUs"; the
reference starts all GPUs from one `python train.py`, matchmaker/train.py:194-202).  On a machine without GPUs only
the plumbing can run: `--dry --device cpu --backend gloo` fabricates rank-tagged scores, performs the per-step
all-gather and reports value = null."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env,
                       timeout=300, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


def test_gpus_2_launches_itself_and_gathers():
    r, lines = _run("--gpus", "2", "--dry", "--device", "cpu", "--backend", "gloo", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout          # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["world_size"] == 2 and j["dry"] is True and j["value"] is None
    assert j["all_gather_verified"] is True
    assert j["collective"]["backend"] == "gloo" and j["collective"]["world_size"] == 2
    assert j["collective"]["bytes_per_rank"] == 4 * j["config"]["queries_per_gpu"] * 1000
    assert j["steps"] == 3 and j["scaling"] == "weak"


def test_single_rank_dry_and_the_cpu_guard():
    r, lines = _run("--dry", "--device", "cpu", "--steps", "2", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["collective"] is None and j["dry"] is True
    # a CPU device without --dry is refused: there is no CPU scoring path to measure
    r, _ = _run("--device", "cpu")
    assert r.returncode != 0 and "only valid with --dry" in (r.stderr + r.stdout)
