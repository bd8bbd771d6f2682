"""profiles/r04_train_step_trace.json (one traced process: `bench.py --only train_step --lean`) split per model, with the
per-size figures of the same leg from the full bench line beside the kernels:

    python tools/split_train_trace.py profiles/r04_train_step_trace.json profiles/r04_bench_n1.json profiles/r04

-> profiles/r04_{colbert,tk,tkl}_bwd_trace.json.  The trace mixes the 64 / 2,048 / 32,768-pair calls of a model under one
kernel name (avg / median are over all of them); the per-size times are bench.py's own (HIP events, steady state)."""
import json
import sys

MODELS = {
    "colbert": ("colbert_fp16_autocast_q32_d180_e128", ("maxsim_bwd_kernel", "maxsim_pair_kernel", "pack_mask_kernel<long>")),
    "tk": ("tk_pooling_q20_d200_e300", ("kernel_pool_bwd_tiled_kernel", "kp_bwd_split_kernel", "kp_bwd_combine_kernel", "kernel_pool_split_kernel", "pack_mask2_kernel", "pack_mask_kernel<float>")),
    "tkl": ("tkl_scoring_d2048_e300", ("tkl_bwd_tiled_kernel", "tkl_bwd_fill_kernel", "tkl_bwd_slot_kernel", "tkl_stage1_run_kernel", "tkl_stage1_rows_kernel",
                                         "tkl_window_kernel", "tkl_prep_kernel")),
}


def main():
    trace = json.load(open(sys.argv[1]))
    line = json.load(open(sys.argv[2]))
    legs = line["extra"]["train_step"]
    for short, (key, names) in MODELS.items():
        out = {"model": short, "source_trace": sys.argv[1], "command": trace.get("command"),
               "note": "kernel rows: all call sizes of the leg under one name; per-size figures: bench.py's own timing in the full line",
               "kernels": [k for k in trace["kernel_trace"] if any(n in k["name"] for n in names)],
               "per_size": {sz: {f: v[f] for f in ("pairs", "forward_us", "step_us", "backward_op_us", "backward_op_over_forward",
                                                      "algorithmic_bytes", "roofline", "eager_gpu_baseline") if f in v}
                            for sz, v in legs[key].items()}}
        with open(f"{sys.argv[3]}_{short}_bwd_trace.json", "w") as f:
            json.dump(out, f, indent=1)
        print(short, len(out["kernels"]), "kernels;", {sz: round(v["backward_op_over_forward"], 2) for sz, v in out["per_size"].items()})


if __name__ == "__main__":
    main()
