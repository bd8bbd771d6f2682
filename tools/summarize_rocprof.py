#!/usr/bin/env python
"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) runs into small committed text/JSON files.

    python tools/summarize_rocprof.py <dir with */*_results.db> <out.json> [kernel-substring]

For the kernel-trace DB: per-kernel calls / total / average duration (the `--stats` view).
For PMC DBs: per-kernel average counter value per dispatch.  HBM traffic is derived exactly as
MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE / WRITE_SIZE are KiB, collected in separate passes,
and on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read -> doubled.
"""
import glob
import json
import os
import sqlite3
import sys


def main():
    root, out = sys.argv[1], sys.argv[2]
    pat = sys.argv[3] if len(sys.argv) > 3 else "mm::"
    res = {"source_dir": os.path.basename(os.path.normpath(root)), "kernel_filter": pat, "kernel_trace": [], "pmc": {},
           "workload": {"queries": int(os.environ.get("MM_PROF_QUERIES", 256)), "cands": 1000,
                        "lengths": os.environ.get("MM_PROF_LENGTHS", "full"),
                        "command": os.environ.get("MM_PROF_COMMAND", "python bench.py --no-cpu-baseline (see tools/profile_maxsim.sh)")}}
    if "MM_PROF_COMMAND" in os.environ:      # not the bench workload: drop the bench-specific keys
        res["workload"] = {"command": os.environ["MM_PROF_COMMAND"]}
    for db in sorted(glob.glob(os.path.join(root, "*", "*_results.db"))):
        tag = os.path.basename(os.path.dirname(db))
        con = sqlite3.connect(db)
        cur = con.cursor()
        try:
            rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        except sqlite3.Error:
            rows = []
        ks = [{"name": r[0][:160], "calls": r[1], "total_us": r[2], "avg_us": r[3], "pct": r[4]}
              for r in rows if pat in r[0]]
        if tag == "trace" or tag.startswith("trace"):
            # per-dispatch durations (the `kernels` view): the median and the mean of the later half of the launches sit
            # beside the --stats average, which also counts a process's first, cold-clock launches
            try:
                per = {}
                for name, dur in cur.execute("select name, duration from kernels order by start"):
                    per.setdefault(name, []).append(dur)
                for k, r in zip(ks, [r for r in rows if pat in r[0]]):
                    d = per.get(r[0])
                    if d:
                        sd = sorted(d)
                        late = d[len(d) // 2:]
                        k["median_us"] = sd[len(sd) // 2] / 1e3
                        k["late_half_avg_us"] = sum(late) / len(late) / 1e3
                        k["min_us"] = sd[0] / 1e3
            except sqlite3.Error:
                pass
            res["kernel_trace"] += ks
            res["kernel_trace_all_top5"] = [{"name": r[0][:100], "calls": r[1], "avg_us": r[3], "pct": r[4]} for r in rows[:5]]
        try:
            q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                 "group by kernel_name, counter_name")
            for name, cname, n, v, dur in cur.execute(q):
                if pat in name:
                    res["pmc"].setdefault(name[:120], {})[cname] = {"dispatches": n, "avg_per_dispatch": v,
                                                                   "avg_dispatch_ns": dur, "pass": tag}
        except sqlite3.Error:
            pass
        con.close()
    for k, c in res["pmc"].items():
        if "FETCH_SIZE" in c:
            rd = c["FETCH_SIZE"]["avg_per_dispatch"] * 1024 * 2      # gfx950 correction: x2
            wr = c.get("WRITE_SIZE", {}).get("avg_per_dispatch", 0.0) * 1024
            c["_hbm_traffic_bytes_per_dispatch"] = {"read_corrected": rd, "write": wr, "total": rd + wr,
                                                    "note": "FETCH_SIZE KiB x1024 x2 (gfx950 half-count) + WRITE_SIZE KiB x1024"}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"]["avg_per_dispatch"] > 0:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; a busy cycle is counted per SIMD (4 x 256 of them)
            cyc = c["GRBM_GUI_ACTIVE"]["avg_per_dispatch"] / 8.0
            c["_mfma_busy_frac"] = {"total": c["SQ_VALU_MFMA_BUSY_CYCLES"]["avg_per_dispatch"] / (1024.0 * cyc),
                                    "kernel_cycles": cyc,
                                    "note": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)"}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1)[:3000])


if __name__ == "__main__":
    main()
