import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl_audio_amd import ops
dev = torch.device("cuda:0")
S, H = 256, 12
N = int(sys.argv[1]) if len(sys.argv) > 1 else 249        # 249: 16 x 16 patches at 10 s; 501: 16 x 8 patches
C = 64 * H
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(S * N, 3 * C, device=dev, generator=g).to(torch.bfloat16)
out = torch.empty(S * N, C, device=dev, dtype=torch.bfloat16); lse = torch.empty(S * H, N, device=dev)
dout = torch.randn(S * N, C, device=dev, generator=g).to(torch.bfloat16); dqkv = torch.empty_like(qkv)
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fw = t(lambda: ops.attention_fwd(qkv, H, N, 0.125, out, lse))
bw = t(lambda: ops.attention_bwd(qkv, H, N, 0.125, out, dout, lse, dqkv))
fl = 4.0 * S * H * N * N * 64
print(f"attention fwd {fw:.1f} us ({fl/fw/1e6:.1f} TFLOP/s)   bwd {bw:.1f} us ({2.5*fl/bw/1e6:.1f} TFLOP/s algorithmic)")

