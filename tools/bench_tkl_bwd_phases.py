"""Phase clocks of the tiled TKL backward kernel (tkl_bwd_tiled_kernel built with -DMM_TKL_BWD_PHASE_TIMES=1):

    tools/build_variant.sh phases tkl_bwd -DMM_TKL_BWD_PHASE_TIMES=1
    MM_NATIVE_LIB=variants/libmm_native_phases.so python tools/bench_tkl_bwd_phases.py [documents]

prints thread 0 / document 0's s_memtime ticks per phase (summed over its <= 15 windows), and — with any library — the
duration of the whole mm_tkl_bwd call and of its grad_chunks memset alone."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matchmaker_amd import _lib, ops, synth  # noqa: E402
from matchmaker_amd.tkl import TKL_sigir20, chunk_documents  # noqa: E402

NAMES = ["setup + window list + row tables", "window: commit + barrier", "window: norms + cosines", "window: cosine reduce",
         "window: pool + lengths", "window: saturation", "window: G -> region sum", "region block: fetch + commit", "region block: norms, td / sq",
         "region block: chunk-row gradients", "region block: query gradient", "final"]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    dev = torch.device("cuda", 0)
    Qt, Dt, Et = 20, 2048, 300
    g = torch.Generator(device=dev).manual_seed(5)
    mu = [1.0, 0.9, 0.7, 0.5, 0.3, 0.1, -0.1, -0.3, -0.5, -0.7, -0.9]
    m = TKL_sigir20(Et, mu, [0.1] * 11, 10, 2, 300, 2000, True, True, "embedding").to(dev)
    q_len = torch.randint(3, Qt + 1, (B,), generator=g, device=dev)
    d_len = torch.randint(50, Dt + 1, (B,), generator=g, device=dev)
    qm = synth.len_to_mask(q_len, Qt, torch.float32)
    dm = synth.len_to_mask(d_len, Dt, torch.float32)
    dd = torch.randn(B, Dt, Et, generator=g, device=dev) * dm.unsqueeze(-1)
    chunks, cmask, slot, C = chunk_documents(dd, dm)
    del dd
    q_ctx = torch.randn(B, Qt, Et, generator=g, device=dev) * qm.unsqueeze(-1)
    go = torch.randn(B, generator=g, device=dev)
    packed = m.pack_params()
    with torch.no_grad():
        win0 = ops.tkl_score(q_ctx, chunks, cmask, slot, qm, packed, B, C, 11, "embedding", return_windows=True, check_order=False)[1]
    L = _lib.lib()
    P, NP = chunks.shape[0], packed.numel()
    gq = torch.empty_like(q_ctx)
    gc = torch.empty_like(chunks)
    gp = torch.empty(B, NP, device=dev)
    cm = cmask.to(torch.float32).contiguous()
    sl = slot.to(torch.int32).contiguous()
    wsb = L.mm_tkl_bwd_workspace_bytes2(B, C, Qt, Et)       # (with the per-region shares when the batch is small: three workgroups per document)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for it in range(4):
        if it == 3:
            ev[0].record()
        rc = L.mm_tkl_bwd(q_ctx.data_ptr(), chunks.data_ptr(), cm.data_ptr(), sl.data_ptr(), qm.data_ptr(), packed.data_ptr(),
                          win0.data_ptr(), go.data_ptr(), gq.data_ptr(), gc.data_ptr(), gp.data_ptr(), B, P, C, Qt, Et, 11,
                          _lib.TKL_SAT_EMBEDDING, ws.data_ptr(), wsb, ops._stream(dev))
        assert rc == 0, rc
    ev[1].record()
    ev[2].record()
    gc.zero_()
    ev[3].record()
    torch.cuda.synchronize()
    print(f"{B} documents, {P} packed chunks: mm_tkl_bwd {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us; zero-filling grad_chunks alone "
          f"({gc.numel() * 4 / 1e9:.2f} GB) {ev[2].elapsed_time(ev[3]) * 1e3:.1f} us")
    if "phases" in os.environ.get("MM_NATIVE_LIB", ""):
        t = gp[0, :12].tolist()
        tot = sum(t)
        for name, v in zip(NAMES, t):
            print(f"{name:34s} {v:10.0f} ticks  {100 * v / tot:5.1f} %")
        print(f"{'total':34s} {tot:10.0f} ticks")


if __name__ == "__main__":
    main()
