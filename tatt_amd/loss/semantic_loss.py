"""SemanticLoss (text-prior distillation, L_TP) on the HIP kernels.  Reference: loss/semantic_loss.py:10-38.
Same constructor and call: `SemanticLoss()(pred_vec, gt_vec) -> scalar` with
loss = mean|gt - pred| + KLDivLoss()(log(pred + 1e-20), gt + 1e-20); gradients flow to `pred_vec` (the student) only."""
import torch
from torch import nn

from ..lib import lib


class _SemanticLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt):
        if not pred.is_cuda:
            raise RuntimeError("tatt_amd SemanticLoss runs on the GPU only")
        p, g = pred.detach().contiguous().float(), gt.detach().contiguous().float()
        out = torch.empty(1, device=p.device)
        ws = torch.empty((p.numel() + 255) // 256, device=p.device)
        lib().semantic_loss_probs(p, g, out, None, 0.0, p.numel(), ws)
        ctx.save_for_backward(p, g)
        return out[0]

    @staticmethod
    def backward(ctx, dout):
        p, g = ctx.saved_tensors
        dp = torch.empty_like(p)
        out = torch.empty(1, device=p.device)
        ws = torch.empty((p.numel() + 255) // 256, device=p.device)
        lib().semantic_loss_probs(p, g, out, dp, float(dout), p.numel(), ws)     # dloss is a host float in the C ABI
        return dp, None


class SemanticLoss(nn.Module):
    def __init__(self, margin=0.1):
        super().__init__()
        self.margin = margin
        self.lambda1 = 1.0
        self.lambda2 = 1.0

    def forward(self, pred_vec, gt_vec):
        return _SemanticLossFn.apply(pred_vec, gt_vec)
