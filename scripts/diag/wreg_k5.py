import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ssl_audio_amd import ops
dev = torch.device("cuda:0")
M, N, K = int(sys.argv[1]) if len(sys.argv) > 1 else 4133, 768, 192
g = torch.Generator().manual_seed(1)
A = torch.randn(M, K, generator=g).to(torch.bfloat16); W = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16)
aux = torch.randn(M, N, generator=g).to(torch.bfloat16)
acc = A.double() @ W.double().t()
ref = acc * aux.double()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
cs = torch.zeros(N, device=dev)
ops.gemm(A.to(dev), W.to(dev), act=4, aux_in=aux.to(dev), out_bf16=out, colsum_out=cs)
torch.cuda.synchronize()
d = (out.double().cpu() - ref).abs()
badmask = d > 0.02 * ref.abs() + 1e-3
print("bad elements", int(badmask.sum()), "of", M * N)
rows = badmask.any(1).nonzero().flatten(); cols = badmask.any(0).nonzero().flatten()
print("bad rows", rows[:40].tolist(), "n", len(rows)); print("bad cols", cols[:40].tolist(), "n", len(cols))
if len(rows):
    r = int(rows[0]); c = int(cols[0])
    print("sample", r, c, float(out[r, c]), float(ref[r, c]), "acc", float(acc[r, c]), "aux", float(aux[r, c]))
    # is out = acc * aux[some other row]?
    for dr in (-48, -32, -16, 16, 32, 48, 64):
        rr = r + dr
        if 0 <= rr < M: print("  vs aux row", rr, float(acc[r, c] * aux[rr, c].double()))
print("colsum rel err", float((cs.double().cpu() - ref.sum(0)).norm() / ref.sum(0).norm()))
idx = badmask.nonzero()
import collections
pat = collections.Counter((int(r) % 64, int(c) % 96) for r, c in idx[:20000].tolist())
print("distinct (row%64, col%96) patterns", len(pat)); print(sorted(pat.items())[:60])
auxd = aux.double()
for r, c in idx[:12].tolist():
    implied = float(out[r, c]) / float(acc[r, c])
    # where does aux hold this value (to bf16 precision) in the same stage?
    s0 = r // 64 * 64
    blk = auxd[s0:s0 + 64]
    hits = ((blk - implied).abs() < 0.01 * abs(implied) + 1e-3).nonzero()[:6].tolist()
    print(f"bad ({r},{c}) out {float(out[r,c]):.4f} ref {float(ref[r,c]):.4f} implied aux {implied:.4f} true {float(aux[r,c]):.4f} found at (row-in-stage, col): {hits}")
