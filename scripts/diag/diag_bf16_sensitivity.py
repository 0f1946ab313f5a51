"""How much do first-step gradients of the golden 'step' fixtures move when only the matmul operands are rounded to
bf16 (CPU autocast), everything else fp32?  Sets the expectation for the HIP path's bf16 error on these fixtures."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import step as ostep, heads

def load(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(np.asarray(v)).clone() for k, v in g.items() if k.startswith(prefix)}

for tag, stop, pred_on in [("byol", True, True), ("plain", False, False)]:
    g = dict(np.load(f"tests/golden/step_{tag}.npz"))
    views = [torch.from_numpy(g["view0"]), torch.from_numpy(g["view1"])]
    res = {}
    for mode in ("fp32", "bf16"):
        online, pred = load(g, "online_sd."), load(g, "pred_sd.")
        target = {k: v.clone() for k, v in online.items()}
        opt = ostep.AdamW(float(g["lr"]), float(g["wd"]))
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if mode == "bf16" else torch.autocast("cpu", enabled=False)
        with ctx:
            l, grads = ostep.bt_byol_step(online, target, pred, views, 2, (4, 6), opt, stop, pred_on)
        res[mode] = (l, grads)
    print(tag, "loss fp32 %.4f bf16 %.4f" % (res["fp32"][0], res["bf16"][0]))
    for k in [k for k in g if k.startswith("grad0.")]:
        n = k[len("grad0."):]
        a, b = res["bf16"][1][n].double(), res["fp32"][1][n].double()
        print("   %-50s rel diff %.4f" % (n, float((a - b).norm() / b.norm())))
