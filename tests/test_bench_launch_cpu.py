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


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_last_stdout_line_is_a_compact_record_the_driver_can_keep():
    """Round 4's ONE line was 21.7 KB; the driver keeps ~9 KB of stdout, so value / ms_per_step / roofline were cut off and
    BENCH_r04.parsed was null.  The last line must stay small whatever the legs add, and still carry the contract's keys."""
    b = _bench_module()
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_n1.json")))
    assert len(json.dumps(full)) > 20000                      # the record that broke the driver's parse
    line = json.dumps(b.compact_record(full))
    assert len(line) < 4200, len(line)
    j = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "self_check"):
        assert k in j, k
    assert j["value"] == full["value"] and j["ms_per_step"] == full["ms_per_step"]
    assert j["roofline"]["frac"] == full["roofline"]["frac"] and j["roofline"]["traffic"] == full["roofline"]["traffic"]
    assert j["roofline"]["frac_of_calibrated"] == full["roofline"]["frac_of_calibrated"]
    assert j["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and j["cpu_baseline"]["cores"] == 128
    assert len(j["cpu_baseline"]["sample"]) <= 120
    assert j["config"]["workload"].startswith("BASELINE.json configs[1]")
    # every leg survives as numbers only
    for leg in ("tk", "tkl", "dot_topk", "variants", "train_step", "ragged_aggregate", "eval_batch"):
        assert leg in j["extra"], leg
    assert abs(j["extra"]["tk"]["frac"] - full["extra"]["tk"]["roofline"]["frac"]) < 1e-3
    assert not any(isinstance(v, str) and len(v) > 130 for v in _leaves(j["extra"]))
    # a pathological record (legs that balloon) is still bounded
    fat = json.loads(json.dumps(full))
    for i in range(40):
        fat["extra"][f"leg{i}"] = {"ms": 1.0, "roofline": {"frac": 0.5}, "sub": {f"s{k}": {"ms": 2.0, "frac": 0.1} for k in range(30)}}
    assert len(json.dumps(b.compact_record(fat))) <= 4000


def _leaves(o):
    if isinstance(o, dict):
        for v in o.values():
            yield from _leaves(v)
    else:
        yield o


def test_full_record_goes_to_an_earlier_line_and_to_a_file(tmp_path):
    r, lines = _run("--dry", "--device", "cpu", "--steps", "2", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout.strip().splitlines()
    assert out[-1].startswith("{") and len(out[-1]) < 4200            # the LAST line is the compact record
    assert any(l.startswith("FULL_RECORD {") for l in out[:-1])
    full = json.loads(next(l for l in out if l.startswith("FULL_RECORD "))[len("FULL_RECORD "):])
    assert full["n_gpus"] == json.loads(out[-1])["n_gpus"] == 1


def _hbm_fracs(o, path=""):
    """every roofline object of the record whose bound is HBM -> (path, frac)"""
    if isinstance(o, dict):
        r = o.get("roofline")
        if isinstance(r, dict) and r.get("bound") == "hbm" and isinstance(r.get("frac"), (int, float)):
            yield path, r["frac"]
        for k, v in o.items():
            if k != "roofline":
                yield from _hbm_fracs(v, path + "/" + k)


def test_no_hbm_bound_leg_claims_more_than_the_box_streams():
    """the variants leg divided PADDED bytes by the time of kernels that skip padded rows and printed
    0.92-0.96 of the spec peak on a box whose calibrated stream was 0.82 of it.  Over the newest committed full record of
    `python bench.py` (round 6 on): no HBM-bound `roofline.frac` may exceed calibrated_stream / 8000 by more than 3 %."""
    import glob
    import pytest
    recs = [f for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1.json")))
            if int(os.path.basename(f)[1:3]) >= 6]
    if not recs:
        pytest.skip("no round-6 full record committed yet")
    rec = json.load(open(recs[-1]))
    cal = rec["roofline"]["calibrated_stream_GBps"] / 8000.0
    fr = list(_hbm_fracs(rec.get("extra", {}), "extra")) + [("headline", rec["roofline"]["frac"])]
    assert len(fr) >= 8
    over = [(p, f) for p, f in fr if f > cal * 1.03]
    assert not over, (cal, over)
    v = rec["extra"]["variants"]
    for name in ("knrm", "tk_sparse", "idcm_sampler_ck", "idcm_sampler_ck_small", "conv_knrm_3x3"):
        r = v[name]["roofline"]
        assert r["needed_bytes"] <= v[name]["algorithmic_bytes_padded"]
        assert r["frac"] <= r["frac_padded_bytes"]


def test_variants_leg_prices_needed_bytes_in_the_round5_record():
    """the same rule applied to round 5's committed record with the corrected accounting: needed bytes / ms stays under the
    box's calibrated stream for every variant (the record's own `frac` did not)"""
    rec = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_n1.json")))
    cal = rec["roofline"]["calibrated_stream_GBps"]
    bad = 0
    for name, leg in rec["extra"]["variants"].items():
        if not isinstance(leg, dict) or "ms" not in leg:
            continue
        need = leg["bytes_of_rows_below_the_document_lengths"] + (leg["algorithmic_bytes_padded"] - leg["pairs"] * leg["shape_QDE"][1]
                                                                  * leg["shape_QDE"][2] * 4 * (3 if "conv" in name else 1))
        assert need / (leg["ms"] * 1e-3) / 1e9 <= cal * 1.03, name
        bad += leg["roofline"]["frac"] * 8000.0 > cal * 1.03
    assert bad >= 3      # what round 5 printed: three legs above the stream
