"""Diagnostic: isolate where the byol-mode gradient error arises (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ssl_audio_amd import model, hyperparameters as hp
from ssl_audio_amd.loss import BarlowTwinsLoss
from oracle import heads as oh

dev = torch.device("cuda:0")
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))

torch.manual_seed(0)
B, D = 8, 64
cfg = hp.make_args(model_type="vit_tiny", projector_hidden_dim=192, projector_out_dim=D)
pred = model.BarlowTwinsPredictor(D, use=True).to(dev)
crit = BarlowTwinsLoss(cfg, ncrops=2).to(dev)
zo = torch.randn(2 * B, D, device=dev).requires_grad_(True)
zt = (zo.detach() * 0.7 + 0.5 * torch.randn(2 * B, D, device=dev))
for use_pred in (False, True):
    zo.grad = None
    o = pred(zo, ncrops=1) if use_pred else zo
    loss = crit(o, zt, ngcrops_each=2)
    loss.backward()
    # oracle in fp64 on the same inputs
    sd = {k: v.detach().double().cpu() for k, v in pred.state_dict().items()}
    zo64 = zo.detach().double().cpu().requires_grad_(True)
    o64 = oh.predictor_forward(zo64, sd, ncrops=1)[0] if use_pred else zo64
    l64, _ = oh.bt_forward(o64, zt.double().cpu(), ncrops=2, ngcrops_each=2)
    l64.backward()
    print(f"use_pred={use_pred}: loss {float(loss):.6f} vs {float(l64):.6f}; d zo rel err {rel(zo.grad, zo64.grad):.4e}")
    if use_pred:
        for n, p in pred.named_parameters():
            pass
