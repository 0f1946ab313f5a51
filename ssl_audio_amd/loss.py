"""BarlowTwinsLoss with the reference's interface (utils/loss.py:8-48): BarlowTwinsLoss(cfg, ncrops)(student, teacher,
ngcrops_each).  forward_loss is one fused HIP schedule (functional.BTLossFn)."""
import torch
import torch.nn as nn

from . import functional as Fn


class BarlowTwinsLoss(nn.Module):
    def __init__(self, cfg, ncrops, literal_ddp=False):
        super().__init__()
        self.cfg = cfg
        self.ncrops = ncrops
        self.literal_ddp = literal_ddp   # True: reproduce utils/loss.py:19-21 under DDP verbatim
        self.bn = nn.BatchNorm1d(cfg.projector_out_dim, affine=False)   # buffer holder (running stats are checkpointed)

    def forward_loss(self, z1, z2):
        loss = Fn.BTLossFn.apply(z1, z2, float(self.cfg.alpha), float(self.cfg.lmbda), bool(self.cfg.HSIC),
                                 self.bn.running_mean, self.bn.running_var, self.literal_ddp)
        with torch.no_grad():
            self.bn.num_batches_tracked += 2
        return loss

    def forward(self, student_output, teacher_output, ngcrops_each=1):
        student_out = student_output.chunk(self.ncrops - (2 - ngcrops_each))
        teacher_out = teacher_output.chunk(ngcrops_each)
        total_loss = 0
        n_loss_terms = 0
        for q in range(len(teacher_out)):
            for v in range(len(student_out)):
                if len(teacher_out) > 1:
                    if q == v:
                        continue
                total_loss = total_loss + self.forward_loss(teacher_out[q], student_out[v])
                n_loss_terms += 1
        return total_loss / n_loss_terms
