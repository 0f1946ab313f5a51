import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ssl_audio_amd import ops
from oracle import augment as oaug
g = dict(np.load("tests/golden/augment.npz"))
dev = torch.device("cuda:0")
for tag in ["t1001", "t1001b"]:
    x = torch.from_numpy(g[f"rrc_{tag}_x"]); p = [int(v) for v in g[f"rrc_{tag}_params"]]
    out = torch.empty(1, 1, 64, 1001, device=dev)
    params = torch.tensor([[0.0, *p, 0, 0, 0]], dtype=torch.float32, device=dev)
    ops.augment_views(x.to(dev), 64 * 1001, torch.zeros(1, dtype=torch.int32, device=dev), None, params, out, 64, 1001, (64, 1501), 1501 / 1000, False)
    y = out.cpu().numpy()[0]
    ref = g[f"rrc_{tag}_y"]; orc = oaug.rrc_apply(x.numpy(), p, (64, 1001))
    d = np.abs(y - ref)[0]; f, t = np.unravel_index(d.argmax(), d.shape)
    print(tag, "params", p, "max|hip-golden|", d.max(), "at f,t", f, t, "| max|hip-oracle|", np.abs(y - orc).max(), "| max|oracle-golden|", np.abs(orc - ref).max())
    print("   per-column max err (every 100):", [float(d[:, k].max()) for k in range(0, 1001, 100)])
    print("   per-row max err:", [round(float(v), 6) for v in d.max(1)[:8]])
