"""Per-phase cycle stamps of the persistent 256x256 GEMM (SA_GEMM_DBG=8): where does one wave spend a K-step?"""
import os, sys, ctypes
os.environ.setdefault("SA_GEMM_DBG", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl_audio_amd import ops
from ssl_audio_amd._lib import lib

M = 2 * 128 * 251
shapes = {"qkv": (2304, 768), "proj": (768, 768), "fc1": (3072, 768), "fc2": (768, 3072)}
dev = "cuda"
for name, (N, K) in shapes.items():
    a = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16() * 0.02
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, w, b_kmajor=True, bias=(None if os.environ.get("SA_PROFILE_NO_BIAS") else bias), out_bf16=out)
    buf = (ctypes.c_uint64 * 16)()
    assert lib().sa_gemm_debug_counters(buf) == 0
    for wg in range(2):
        steps, mfma, vm, bar, epi, total, top, ebar = [buf[wg * 8 + i] for i in range(8)]
        if steps == 0:
            continue
        print(f"{name:5s} wg{wg}: steps={steps:4d} total={total:9d} clk | per K-step: frag+mfma {mfma/steps:7.0f}  vmcnt-wait {vm/steps:7.0f}  "
              f"barrier {bar/steps:6.0f} | per tile: epilogue {epi/(steps/(K//64)):8.0f} (of which pre-epilogue barrier {ebar/(steps/(K//64)):7.0f})  tile-top wait {top/(steps/(K//64)):7.0f}")
