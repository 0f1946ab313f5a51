"""GEMM micro-benchmark at the ViT-B step's shapes AND epilogues (M = 256 sequences x 249 tokens): the 12 launches one transformer
block's forward + backward makes, exactly as engine.block_forward / block_backward call them.  Random operands (rule 25).
   python scripts/bench_gemm.py [filter] [reps]          (SA_GEMM_TILE / SA_GEMM_WGRAD_PHASE select kernels, read once per process)
   SA_BENCH_D / SA_BENCH_SEQS: model width and sequences per step (default 768 / 256 = ViT-B at 128 clips; 192 / 512 = ViT-T at 256)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl_audio_amd import ops, engine
dev = torch.device("cuda:0")
d = int(os.environ.get("SA_BENCH_D", "768"))
M = int(os.environ.get("SA_BENCH_SEQS", "256")) * 249
g = torch.Generator(device=dev).manual_seed(0)
def rb(*s): return torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
def rf(*s): return torch.randn(*s, device=dev, generator=g)
def e16(*s): return torch.empty(*s, device=dev, dtype=torch.bfloat16)
x16, h4 = rb(M, d), rb(M, 4 * d)
q16 = rb(M, 3 * d)
res = rf(M, d)
W = {"qkv": rb(3 * d, d) * 0.05, "proj": rb(d, d) * 0.05, "fc1": rb(4 * d, d) * 0.05, "fc2": rb(d, 4 * d) * 0.05}
bias = {"qkv": rf(3 * d), "proj": rf(d), "fc1": rf(4 * d), "fc2": rf(d)}
aux = rb(M, 4 * d)
NT = os.environ.get("SA_DGRAD_NT", "1") != "0"
DG = "NT" if NT else "NN"
WD = {k: (ops.transpose_bf16(v.contiguous()) if NT else v) for k, v in W.items()}
cases = [
    ("fwd  NT qkv  bias->bf16", lambda o=e16(M, 3 * d): ops.gemm(x16, W["qkv"], bias=bias["qkv"], out_bf16=o), 2.0 * M * 3 * d * d),
    ("fwd  NT proj bias+res->f32", lambda o=torch.empty(M, d, device=dev): ops.gemm(x16, W["proj"], bias=bias["proj"], residual=res, out_f32=o), 2.0 * M * d * d),
    ("fwd  NT fc1  gelu pair", lambda o=e16(M, 4 * d), a=e16(M, 4 * d): ops.gemm(x16, W["fc1"], bias=bias["fc1"], act=3, aux_out=a, out_bf16=o), 2.0 * M * 4 * d * d),
    ("fwd  NT fc2  bias+res->f32", lambda o=torch.empty(M, d, device=dev): ops.gemm(h4, W["fc2"], bias=bias["fc2"], residual=res, out_f32=o), 2.0 * M * 4 * d * d),
    # data gradients as the engine runs them: against the transposed bf16 weight copy, forward operand layout (SA_DGRAD_NT=0 / "NN" below: k-strided W)
    ("dgrad %s fc2 *gelu' +colsum" % DG, lambda o=e16(M, 4 * d), cs=torch.zeros(4 * d, device=dev): ops.gemm(x16, WD["fc2"], b_kmajor=NT, act=4, aux_in=aux, out_bf16=o, colsum_out=cs), 2.0 * M * 4 * d * d),
    ("dgrad %s fc1 ->bf16" % DG, lambda o=e16(M, d): ops.gemm(h4, WD["fc1"], b_kmajor=NT, out_bf16=o), 2.0 * M * 4 * d * d),
    ("dgrad %s proj ->bf16" % DG, lambda o=e16(M, d): ops.gemm(x16, WD["proj"], b_kmajor=NT, out_bf16=o), 2.0 * M * d * d),
    ("dgrad %s qkv ->bf16" % DG, lambda o=e16(M, d): ops.gemm(q16, WD["qkv"], b_kmajor=NT, out_bf16=o), 2.0 * M * 3 * d * d),
]
for name, N, K, dy, xx in [("qkv", 3 * d, d, q16, x16), ("proj", d, d, x16, x16), ("fc1", 4 * d, d, h4, x16), ("fc2", d, 4 * d, x16, h4)]:
    out = torch.zeros(N, K, device=dev)
    use256 = ((N + 255) // 256) * ((K + 255) // 256) >= int(os.environ.get("SA_WGRAD256_MIN", "9"))
    if engine.stream_wgrad(N, K, M) or os.environ.get("SA_BENCH_STREAM") == "1":
        use256 = 2
    sk = ops.pick_split_k(N, K, M, tile={0: 128, 1: 256, 2: 192}[int(use256)])
    if os.environ.get("SA_BENCH_SPLIT"):
        sk = int(os.environ["SA_BENCH_SPLIT"])
    cases.append((f"wgrad TN {name} split{sk}{['', ' 256', ' 192 stream'][int(use256)]}",
                  lambda out=out, dy=dy, xx=xx, sk=sk, use256=use256: ops.gemm(dy, xx, a_kmajor=False, b_kmajor=False, out_f32=out, split_k=sk, tile256=int(use256)), 2.0 * M * N * K))
_grp = [(dy, xx, torch.zeros(N, K, device=dev)) for name, N, K, dy, xx in [("qkv", 3 * d, d, q16, x16), ("proj", d, d, x16, x16), ("fc1", 4 * d, d, h4, x16), ("fc2", d, 4 * d, x16, h4)]]
if all(engine.stream_wgrad(o.shape[0], o.shape[1], M) for _, _, o in _grp):
    # the block's four weight gradients as ONE launch pair (engine.WgradGroup)
    _tiles = sum(((o.shape[0] + 191) // 192) * ((o.shape[1] + 191) // 192) for _, _, o in _grp)
    _sk = int(os.environ.get("SA_BENCH_SPLIT", "0")) or ops.pick_split_k(0, 0, M, tile=192, tiles=_tiles)
    cases.append((f"wgrad TN block group of 4 split{_sk}", lambda: ops.gemm_wgrad_group([j[0] for j in _grp], [j[1] for j in _grp], [j[2] for j in _grp], _sk),
                  sum(2.0 * M * o.shape[0] * o.shape[1] for _, _, o in _grp)))
only = sys.argv[1] if len(sys.argv) > 1 else None
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
tot_t = tot_f = 0.0
for name, fn, fl in cases:
    if only and only not in name:
        continue
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    if "group" not in name:                                # (the block total keeps its twelve launches)
        tot_t += ms; tot_f += fl
    print(f"{name:30s} {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s")
if tot_t > 0:
    print(f"ALL (one block's fwd+bwd GEMMs, tile={os.environ.get('SA_GEMM_TILE', 'default')}): {tot_t:.3f} ms  {tot_f/tot_t/1e9:.1f} TFLOP/s")
