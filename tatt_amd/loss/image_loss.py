"""ImageLoss (MSE + gradient-prior L1) on the HIP kernels.  Reference: loss/image_loss.py:10-56.
Same constructor (`gradient`, `loss_weight`) and call `ImageLoss(...)(out_images, target_images) -> [B]`."""
import torch
from torch import nn

from ..lib import lib


class _ImageLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sr, hr, w0, w1):
        if not sr.is_cuda:
            raise RuntimeError("tatt_amd ImageLoss runs on the GPU only")
        sr, hr = sr.contiguous().float(), hr.detach().contiguous().float()
        B, C, H, W = sr.shape
        per = torch.empty(B, device=sr.device, dtype=torch.float32)
        lib().image_loss_fwd(sr, hr, per, B, C, H, W, w0, w1)
        ctx.save_for_backward(sr, hr)
        ctx.w = (w0, w1)
        return per

    @staticmethod
    def backward(ctx, dper):
        sr, hr = ctx.saved_tensors
        B, C, H, W = sr.shape
        dsr = torch.empty_like(sr)
        lib().image_loss_bwd(sr, hr, dper.contiguous().float(), 0.0, dsr, B, C, H, W, *ctx.w)
        return dsr, None, None, None


class ImageLoss(nn.Module):
    def __init__(self, gradient=True, loss_weight=[20, 1e-4]):
        super().__init__()
        self.gradient = gradient
        self.loss_weight = loss_weight

    def forward(self, out_images, target_images, grad_mask=None):
        w1 = float(self.loss_weight[1]) if self.gradient else 0.0
        return _ImageLossFn.apply(out_images, target_images, float(self.loss_weight[0]), w1)
