cd $GRAFT_REPO_ROOT
python - <<'PY' &
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from ssl_audio_amd import ops
dev = torch.device("cuda:0")
S, H, N = 256, 12, 249
C = 64 * H
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(S * N, 3 * C, device=dev, generator=g).to(torch.bfloat16)
out = torch.empty(S * N, C, device=dev, dtype=torch.bfloat16); lse = torch.empty(S * H, N, device=dev)
dout = torch.randn(S * N, C, device=dev, generator=g).to(torch.bfloat16); dqkv = torch.empty_like(qkv)
ops.attention_fwd(qkv, H, N, 0.125, out, lse)
A = torch.randn(64256, 768, device=dev).bfloat16(); W = torch.randn(3072, 768, device=dev).bfloat16(); o2 = torch.empty(64256, 3072, device=dev, dtype=torch.bfloat16)
t0 = time.time()
print("phase attention_bwd", flush=True)
while time.time() - t0 < 6:
    for _ in range(50): ops.attention_bwd(qkv, H, N, 0.125, out, dout, lse, dqkv)
    torch.cuda.synchronize()
print("phase gemm", flush=True)
t0 = time.time()
while time.time() - t0 < 6:
    for _ in range(50): ops.gemm(A, W, out_bf16=o2)
    torch.cuda.synchronize()
print("done", flush=True)
PY
sleep 4
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|fclk\|mclk" | head -3 | tr '\n' ' '; rocm-smi --showpower 2>/dev/null | grep -i "power" | head -1; sleep 1; done
wait
