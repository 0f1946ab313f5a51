"""What does the memory system give a WRITE-heavy stream?  Thin products (K = 192) write 3 - 8 x the bytes they read; this times stock
streaming kernels with the same read : write mixes (fill, copy, 1 : 4 expand) at the ViT-T activation sizes.
   python scripts/diag/write_mix_rate.py"""
import torch
dev = torch.device("cuda:0")
M = 512 * 249
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
bufs = [torch.empty(M, 768, device=dev, dtype=torch.bfloat16) for _ in range(6)]     # 196 MB each: rotate so the Infinity Cache does not hold the target
x192 = [torch.randn(M, 192, device=dev).to(torch.bfloat16) for _ in range(6)]
i = [0]
def nxt():
    i[0] = (i[0] + 1) % 6
    return i[0]
def fill():
    bufs[nxt()].fill_(1.0)
def copy():
    k = nxt(); bufs[k].copy_(bufs[(k + 3) % 6])
def expand():
    k = nxt(); bufs[k].view(M, 4, 192).copy_(x192[k].view(M, 1, 192).expand(M, 4, 192))
for name, fn, nbytes in [("fill 196 MB bf16", fill, M * 768 * 2), ("copy 196 -> 196 MB", copy, 2 * M * 768 * 2), ("expand 49 -> 196 MB (1 : 4)", expand, M * 960 * 2)]:
    us = timeit(fn)
    print(f"{name:32s} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s")
f32 = [torch.empty(M, 768, device=dev) for _ in range(3)]
def fill32():
    f32[nxt() % 3].fill_(1.0)
us = timeit(fill32); print(f"{'fill 392 MB f32':32s} {us:8.1f} us  {M * 768 * 4 / us / 1e6:6.2f} TB/s")
