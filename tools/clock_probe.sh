#!/bin/bash
# Shader clock / power while a kernel family runs back to back (run on the GPU box from the repo root):
#   bash tools/clock_probe.sh     -> prints rocm-smi samples taken during ~6 s loops of MaxSim, TK pooling and dot top-k
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
probe() {
  python - "$1" <<'PY' &
import sys, time, torch
sys.path.insert(0, ".")
from matchmaker_amd import ops, synth
dev = torch.device("cuda:0"); which = sys.argv[1]
g = torch.Generator(device=dev).manual_seed(1)
if which == "maxsim":
    q, d, ql, dl = synth.colbert_batch(256, 1000, dtype=torch.bfloat16, device=dev)
    fn = lambda: ops.maxsim(q, d, ql, dl, pairs_per_query=1000)
elif which == "tk":
    q = torch.randn(64, 20, 300, generator=g, device=dev); d = torch.randn(64000, 200, 300, generator=g, device=dev)
    p = [torch.tensor([1.0, .9, .7, .5, .3, .1, -.1, -.3, -.5, -.7, -.9], device=dev), torch.full((11,), .1, device=dev), torch.ones(11, device=dev), torch.ones(11, device=dev)]
    fn = lambda: ops.kernel_pool(q, d, None, None, *p, pairs_per_query=1000)
else:
    c = torch.randn(1105228, 768, generator=g, device=dev).half(); q = torch.randn(6980, 768, generator=g, device=dev).half()
    fn = lambda: ops.dot_topk(q, c, 1000)
fn(); torch.cuda.synchronize(); t0 = time.time()
while time.time() - t0 < 7:
    for _ in range(20 if which != "dot" else 2): fn()
    torch.cuda.synchronize()
PY
  PID=$!
  sleep 4.5
  echo "== $1"
  rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|mclk|power" | head -6
  sleep 1.0
  rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|power" | head -3
  wait $PID
}
probe maxsim; probe tk; probe dot
