"""GPU: the RCCL path of bench.py on the ONE GPU a test box has (north_star: "RCCL all-gather of scores over xGMI only
for the final ranking merge"; the reference starts all its GPUs from one command, train.py:194-202).

* world size 1 through `--force-dist`: backend "nccl" (= RCCL) is initialised, the per-step
  all_gather_into_tensor runs inside the timed region, and the gathered slice must equal the local scores;
* world size 2 on ONE device (`--one-device`): RCCL normally refuses two ranks on the same GPU — when it does, the
  test records RCCL's own message under gpurun_out/ and skips; when it accepts, the gathered tensors are verified.
The first real `--gpus 8` run then has no untried code path left but the xGMI transport itself."""
import json
import os
import signal
import subprocess
import sys

import pytest

from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
        return p.returncode, out, err
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)          # exactly the process group started above
        out, err = p.communicate()
        return -9, out, err + "\n[timeout]"


def _line(out):
    for ln in reversed(out.splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    return None


def test_rccl_all_gather_world_size_1():
    util.require_gpu()
    rc, out, err = _run([sys.executable, "bench.py", "--force-dist", "--queries", "8", "--steps", "3", "--warmup", "1",
                         "--no-extras", "--no-cpu-baseline"], 300)
    assert rc == 0, err[-2000:]
    j = _line(out)
    assert j is not None, out[-2000:]
    c = j["collective"]
    assert c["backend"].startswith("nccl") and c["version"] not in (None, "unknown"), c
    assert c["world_size"] == 1 and c["gathered_slice_equals_local_scores"] is True, c
    assert j["value"] and j["self_check"]["ok"], j.get("self_check")
    print("[rccl] world 1:", json.dumps(c))


def test_rccl_all_gather_two_ranks_on_one_device():
    util.require_gpu()
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    rc, out, err = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                         "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--one-device", "--queries", "4",
                         "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"], 240)
    j = _line(out) if rc == 0 else None
    rec = os.path.join(ROOT, "gpurun_out")
    if os.environ.get("GRAFT_REPO_ROOT") or os.path.isdir(rec):
        os.makedirs(rec, exist_ok=True)
        with open(os.path.join(rec, "rccl_two_ranks_one_device.txt"), "w") as f:
            f.write(f"rc={rc}\n--- stdout\n{out[-4000:]}\n--- stderr\n{err[-6000:]}\n")
    if j is None:
        msg = [ln for ln in (err + out).splitlines() if any(k in ln for k in ("NCCL", "RCCL", "Duplicate", "rror"))]
        pytest.skip("RCCL does not run two ranks on one device here: " + " | ".join(msg[-4:])[:600])
    c = j["collective"]
    assert c["world_size"] == 2 and c["gathered_slice_equals_local_scores"] is True, c
    print("[rccl] world 2 on one device:", json.dumps(c))


def test_config4_rehearsal_world_size_1_under_nccl():
    """BASELINE.json configs[3] as `bench.py --gpus 8` runs it, on one rank: this rank's eighth of the query list
    (sharding.shard_range), the PADDED all-gather of sharding.all_gather_scores under backend nccl, the final stable sort
    timed separately, a sampled query's ranking compared with the ranking of its local scores.  (The query total is
    shrunk so the shard is 2 GB instead of 40.)"""
    util.require_gpu()
    rc, out, err = _run([sys.executable, "bench.py", "--config4", "--total-queries", "349", "--force-dist", "--steps", "3",
                         "--warmup", "1", "--no-extras", "--no-cpu-baseline"], 400)
    assert rc == 0, err[-2000:]
    j = _line(out)
    assert j is not None, out[-2000:]
    c = j["collective"]
    assert "configs[3]" in j["config"]["workload"] and j["config"]["queries_per_gpu"] == 44      # shard_range(349, 8, 0)
    assert c["backend"].startswith("nccl") and c["gathered_slice_equals_local_scores"] is True, c
    assert c["final_sort"]["ranking_equals_single_rank"] is True and c["final_sort"]["rows"] == 44, c
    assert j["value"] and j["self_check"]["ok"], j.get("self_check")
    print("[rccl] config 4 rehearsal:", json.dumps(c))


def test_sharded_flat_index_world_size_1_under_nccl():
    """BASELINE.json configs[4]'s search path (retrieval.FlatIPIndexer.search_device: local exact top-1000, two RCCL
    all-gathers of the (score, id) lists, mm_topk_merge) on one rank under backend nccl — the collectives and the merge run
    even with a single rank (merge_single_rank).  The collection is shrunk to 1.6 M passages (200 k on this rank)."""
    util.require_gpu()
    rc, out, err = _run([sys.executable, "bench.py", "--only", "dot_topk", "--force-dist", "--dot-passages", "1600000",
                         "--steps", "2"], 400)
    assert rc == 0, err[-2000:]
    j = _line(out)
    assert j is not None and j.get("sharded") is True, out[-2000:]
    r = j["result"]
    assert r["backend"].startswith("nccl") and r["world_size"] == 1
    assert r["merged_lists_sorted_and_contain_the_local_top1"] is True, r
    print("[rccl] sharded flat index:", json.dumps({k: r[k] for k in ("ms", "queries_per_s", "version")}))
