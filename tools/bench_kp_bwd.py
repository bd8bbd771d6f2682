"""TK pooling backward: time per launch and (with --check) agreement between the split-bf16 streaming kernel and the exact-f32
tiled kernel (MM_KP_BWD_F32=1 is read once per process -> each leg is a child process) plus fp64 autograd through the reference's ops.

    python tools/bench_kp_bwd.py [--pairs 2048,32768] [--shape 20,200,300] [--check] [--ragged] [--gate]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MU = [1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9]


def child(args):
    import torch
    from matchmaker_amd import ops
    dev = "cuda:0"
    Q, D, E = [int(x) for x in args.shape.split(",")]
    out = {}
    prm = [torch.tensor(MU, device=dev), torch.full((11,), 0.1, device=dev), torch.ones(11, device=dev),
           torch.linspace(-0.014, 0.014, 11, device=dev)]
    for n in [int(x) for x in args.pairs.split(",")]:
        g = torch.Generator(device=dev).manual_seed(7 + n)
        q = torch.randn(n, Q, E, device=dev, generator=g)
        d = torch.randn(n, D, E, device=dev, generator=g)
        if args.ragged:
            ql = torch.randint(3, Q + 1, (n,), device=dev, generator=g).to(torch.int32)
            dl = torch.randint(max(1, D // 8), D + 1, (n,), device=dev, generator=g).to(torch.int32)
        else:
            ql = torch.full((n,), Q, dtype=torch.int32, device=dev)
            dl = torch.full((n,), D, dtype=torch.int32, device=dev)
        gate = torch.relu(torch.randn(n, D, device=dev, generator=g)) if args.gate else None
        go = torch.randn(n, device=dev, generator=g)
        fwd = lambda: ops.kernel_pool(q, d, ql, dl, *prm, d_gate=gate)
        fwd_p = lambda: ops.kernel_pool(q, d, ql, dl, *prm, d_gate=gate, return_pooled=True)
        pooled = fwd_p()[1]
        bwd = lambda: ops.kernel_pool_bwd(q, d, ql, dl, *prm, go, d_gate=gate)
        bwd_p = lambda: ops.kernel_pool_bwd(q, d, ql, dl, *prm, go, d_gate=gate, pooled=pooled)

        def timed(fn, reps=12):
            for _ in range(3):
                fn()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a, b in ev:
                a.record(); fn(); b.record()
            torch.cuda.synchronize()
            return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2] * 1e3
        if args.phases:
            from matchmaker_amd import _lib
            L = _lib.lib()
            _, qp, qk = ops._mask(ql, n, Q, "q_mask")
            _, dp, dk = ops._mask(dl, n, D, "d_mask")
            gq, gd = torch.empty_like(q), torch.empty_like(d)
            ga, gw = torch.zeros(n, 11, device=dev), torch.zeros(n, 11, device=dev)
            wsb = L.mm_kernel_pool_bwd_workspace_bytes(n, Q, D, qk, dk)
            ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
            for _ in range(3):
                rc = L.mm_kernel_pool_ex_bwd(q.data_ptr(), d.data_ptr(), qp, qk, dp, dk, None, prm[0].data_ptr(), prm[1].data_ptr(),
                                             prm[2].data_ptr(), prm[3].data_ptr(), 1e-10, go.data_ptr(), gq.data_ptr(), gd.data_ptr(), None,
                                             ga.data_ptr(), gw.data_ptr(), n, Q, D, E, 11, ws.data_ptr(), wsb, ops._stream(torch.device("cuda", 0)))
                assert rc == 0
            torch.cuda.synchronize()
            names = ["prologue", "block head", "wait slice", "cosines+publish", "barrier 1", "G", "barrier 2", "grad_q pass", "epilogue", "grad_d pass", "LDS-DMA issue"]
            for w, row in (("wavefront 0", gw[n // 2]), ("wavefront 3", ga[n // 2])):
                v = row.tolist()
                print(f"PHASES {n} pairs, {w}: " + " | ".join(f"{nm} {x:.0f}" for nm, x in zip(names, v)) + f" | total {sum(v[:11]):.0f}")
            continue
        tf, tb, tfp, tbp = timed(fwd), timed(bwd), timed(fwd_p), timed(bwd_p)
        by = (2 * n * D * E + 2 * n * Q * E) * 4
        rec = {"fwd_us": tf, "bwd_us": tb, "bwd_over_fwd": tb / tf, "frac_hbm": by / (tb * 1e-6) / 8e12,
               "fwd_with_pooled_us": tfp, "bwd_with_pooled_us": tbp, "bwd_with_pooled_over_fwd": tbp / tf,
               "frac_hbm_with_pooled": by / (tbp * 1e-6) / 8e12}
        if args.check:
            m = min(n, 64)
            r = bwd()
            rp = bwd_p()
            rec["pooled_path_equals_prepass_path"] = bool(torch.equal(r[0], rp[0]) and torch.equal(r[1], rp[1]))
            rec["pooled_path_max_abs_diff"] = float(max((r[0] - rp[0]).abs().max(), (r[1] - rp[1]).abs().max()))
            qq, dd = q[:m].double().requires_grad_(True), d[:m].double().requires_grad_(True)
            qm = (torch.arange(Q, device=dev)[None] < ql[:m, None]).double()
            dm = (torch.arange(D, device=dev)[None] < dl[:m, None]).double()
            a_n = qq / (qq.norm(dim=-1, keepdim=True) + 1e-13)
            b_n = dd / (dd.norm(dim=-1, keepdim=True) + 1e-13)
            cos = torch.bmm(a_n, b_n.transpose(-1, -2))
            mu, sg = prm[0].double().view(1, 1, 1, -1), prm[1].double().view(1, 1, 1, -1)
            k = torch.exp(-(cos.unsqueeze(-1) - mu) ** 2 / (2 * sg ** 2)) * dm.view(m, 1, D, 1)
            if gate is not None:
                k = k * gate[:m].double().view(m, 1, D, 1)
            pkq = k.sum(2)
            lg = torch.log(torch.clamp(pkq * prm[2].double().view(1, 1, -1), min=1e-10)) * qm.unsqueeze(-1)
            sc = (lg.sum(1) * prm[3].double().view(1, -1)).sum(1)
            (sc * go[:m].double()).sum().backward()
            rel = lambda x, y: float((x.double() - y).abs().max() / y.abs().max().clamp_min(1e-30))
            rec["rel_err_grad_q_vs_fp64"] = rel(r[0][:m], qq.grad)
            rec["rel_err_grad_d_vs_fp64"] = rel(r[1][:m], dd.grad)
            rec["nan"] = bool(torch.isnan(r[0]).any() or torch.isnan(r[1]).any())
        out[str(n)] = rec
        del q, d
        torch.cuda.empty_cache()
    print("RESULT " + json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", default="64,2048,32768")
    ap.add_argument("--shape", default="20,200,300")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--gate", action="store_true")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--phases", action="store_true", help="print the phase clocks of a -DMM_KP_BWD_PHASE_TIMES=1 build (MM_NATIVE_LIB=variants/...)")
    args = ap.parse_args()
    if args.child:
        return child(args)
    for name, env in (("split_bf16", {}), ("exact_f32_tiled", {"MM_KP_BWD_F32": "1"})):
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + sys.argv[1:], env=e, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        print(name, line[0][7:] if line else ("FAILED: " + r.stderr[-600:]))


if __name__ == "__main__":
    main()
