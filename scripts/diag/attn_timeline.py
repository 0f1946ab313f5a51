"""Per-workgroup phase timeline of the attention backward (look at the two workgroups a CU hosts side by side).
   bash scripts/ab_build.sh attndbg -DSA_ATTN_DBG=1
   SA_HIP_LIB=$PWD/scripts/microbench/bin/libssl_audio_hip_attndbg.so python scripts/diag/attn_timeline.py [S] [H] [N]
Stamps (shader-clock cycles, waves 0 and 7 of every workgroup): 0 entry, 1 loads issued, 2 loads landed + barrier, 3 pass A done (dQ stores
issued), 4 hand-over done, 5 pass B done (dK / dV stores issued), 6 stores drained; plus HW_ID / XCC_ID of the CU."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssl_audio_amd import ops, _lib
dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = int(sys.argv[2]) if len(sys.argv) > 2 else 12
N = int(sys.argv[3]) if len(sys.argv) > 3 else 249
FWD = len(sys.argv) > 4 and sys.argv[4] == "fwd"          # the forward kernel's timeline instead (stamps 0 entry, 1 loads issued, 2 landed, 5 done, 6 drained)
C = 64 * H
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(S * N, 3 * C, device=dev, generator=g)).to(torch.bfloat16)
out = torch.empty(S * N, C, device=dev, dtype=torch.bfloat16)
lse = torch.empty(S * H, N, device=dev)
ops.attention_fwd(qkv, H, N, 0.125, out, lse)
dout = torch.randn(S * N, C, device=dev, generator=g).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
h = _lib.lib()
buf = torch.zeros(S * H * 2 * 8, dtype=torch.int64, device=dev)
h.sa_attn_dbg_set.argtypes = [ctypes.c_void_p]; h.sa_attn_dbg_set.restype = ctypes.c_int
run = (lambda: ops.attention_fwd(qkv, H, N, 0.125, out, lse)) if FWD else (lambda: ops.attention_bwd(qkv, H, N, 0.125, out, dout, lse, dqkv))
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
t_plain = e0.elapsed_time(e1) * 1e3
assert h.sa_attn_dbg_set(ctypes.c_void_p(buf.data_ptr())) == 0
e0.record(); run(); e1.record(); torch.cuda.synchronize()
t_stamped = e0.elapsed_time(e1) * 1e3
h.sa_attn_dbg_set(ctypes.c_void_p(0))
st = buf.cpu().numpy().astype(np.uint64).reshape(S * H, 2, 8)
hw = st[:, 0, 7]
xcc, se, sh, cu = (hw >> np.uint64(32)) & np.uint64(15), (hw >> np.uint64(13)) & np.uint64(7), (hw >> np.uint64(12)) & np.uint64(1), (hw >> np.uint64(8)) & np.uint64(15)
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
t = st[:, :, :7].astype(np.int64)
print(f"attention {'forward' if FWD else 'backward'} S={S} H={H} N={N}: {t_plain:.1f} us ({t_stamped:.1f} us with stamps); {len(np.unique(cuid))} distinct CUs host {S * H} workgroups")
names = ["issue loads", "loads land (wait + barrier)", "pass A (dQ)", "hand-over", "pass B (dK, dV)", "store drain"]
d = np.diff(t[:, 0, :], axis=1).astype(np.float64)
life = (t[:, 0, 6] - t[:, 0, 0]).astype(np.float64)
print("wave 0, cycles per phase (mean / p10 / p90) and share of the workgroup's life:")
for k, n in enumerate(names):
    print(f"  {n:30s} {d[:, k].mean():8.0f} {np.percentile(d[:, k], 10):8.0f} {np.percentile(d[:, k], 90):8.0f}   {d[:, k].sum() / life.sum():6.1%}")
print(f"  workgroup life {life.mean():.0f} cycles (p10 {np.percentile(life, 10):.0f}, p90 {np.percentile(life, 90):.0f}); wave 7 vs wave 0 end: "
      f"{(t[:, 1, 6] - t[:, 0, 6]).astype(np.float64).mean():+.0f} cycles")
# per CU: overlap of the phases of co-resident workgroups (shader clock of one CU: stamps of its workgroups are comparable)
LOAD, COMP = 0, 1
tot = dict(span=0.0, both_load=0.0, both_comp=0.0, mixed=0.0, one_load=0.0, one_comp=0.0, idle=0.0, resid=0.0)
first_cu = None
for cid in np.unique(cuid):
    idx = np.where(cuid == cid)[0]
    ev = []
    for i in idx:
        a = t[i, 0]
        ev += [(a[0], i, "L+"), (a[2], i, "L-"), (a[2], i, "C+"), (a[5], i, "C-"), (a[5], i, "D+"), (a[6], i, "D-")]
    ev.sort(key=lambda e: e[0])
    nL = nC = nD = 0
    prev = ev[0][0]
    for tt, i, kind in ev:
        dt = float(tt - prev)
        if dt > 0:
            tot["span"] += dt
            tot["resid"] += dt * (nL + nC + nD)
            if nL >= 2 and nC == 0: tot["both_load"] += dt
            elif nC >= 2 and nL == 0: tot["both_comp"] += dt
            elif nL >= 1 and nC >= 1: tot["mixed"] += dt
            elif nL == 1: tot["one_load"] += dt
            elif nC == 1: tot["one_comp"] += dt
            else: tot["idle"] += dt
        prev = tt
        if kind == "L+": nL += 1
        elif kind == "L-": nL -= 1
        elif kind == "C+": nC += 1
        elif kind == "C-": nC -= 1
        elif kind == "D+": nD += 1
        else: nD -= 1
    if first_cu is None and len(idx) >= 8:
        first_cu = idx
sp = tot["span"]
print(f"per-CU timeline, summed over CUs (load = entry .. operands landed, compute = landed .. last stores issued); mean resident workgroups {tot['resid'] / sp:.2f}:")
for k in ("both_load", "mixed", "both_comp", "one_load", "one_comp", "idle"):
    print(f"  {k:10s} {tot[k] / sp:6.1%}")
if first_cu is not None:
    i0 = first_cu[np.argsort(t[first_cu, 0, 0])]
    base = t[i0[0], 0, 0]
    print("one CU, its workgroups in start order (cycles since the first one's entry: entry, landed, passA end, hand-over end, passB end, drained):")
    for i in i0:
        a = t[i, 0] - base
        print(f"  wg {i:5d}  " + " ".join(f"{int(x):8d}" for x in (a[0], a[2], a[3], a[4], a[5], a[6])))
