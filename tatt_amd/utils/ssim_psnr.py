"""TRI_SSIM / SSIM / calculate_psnr on the HIP kernels.  Reference: utils/ssim_psnr.py:9-15 (calculate_psnr),
:76-96 + :200-225 (SSIM, channels 0..2), :99-128 + :229-254 (TRI_SSIM, all channels).  Same class names, constructor
arguments and call signatures; `size_average=True` (the only mode the reference trainer uses) returns a 0-d tensor."""
import torch
from torch import nn

from ..lib import lib


def _check(*imgs):
    for t in imgs:
        if not t.is_cuda:
            raise RuntimeError("tatt_amd SSIM/PSNR run on the GPU only")
    B, C, H, W = imgs[0].shape
    if H * W > 4096:
        raise RuntimeError(f"tatt_amd SSIM kernels hold one plane in LDS: H*W <= 4096 (got {H}x{W})")
    return B, C, H, W


class _TriSSIMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        B, C, H, W = _check(a, b, c)
        a, b, c = (t.detach().contiguous().float() for t in (a, b, c))
        out = torch.empty(1, device=a.device, dtype=torch.float32)
        ws = torch.empty(B * C, device=a.device, dtype=torch.float32)
        lib().tri_ssim_fwd(a, b, c, out, B, C, H, W, ws)
        ctx.save_for_backward(a, b, c)
        return out[0]

    @staticmethod
    def backward(ctx, dout):
        a, b, c = ctx.saved_tensors
        B, C, H, W = a.shape
        da, db = torch.empty_like(a), torch.empty_like(b)
        # dmean is a host float in the C ABI; a 0-d upstream gradient costs one sync here (the trainer's fused step
        # calls the kernel directly with the constant -10)
        lib().tri_ssim_bwd(a, b, c, float(dout), da, db, 0, B, C, H, W)
        return da, db, None


class TRI_SSIM(nn.Module):
    def __init__(self, window_size=11, size_average=True):
        super().__init__()
        if window_size != 11 or not size_average:
            raise NotImplementedError("tatt_amd TRI_SSIM: window_size=11, size_average=True (the reference's use)")
        self.window_size, self.size_average = window_size, size_average

    def forward(self, img1, img2, img3):
        return _TriSSIMFn.apply(img1, img2, img3)


class SSIM(nn.Module):
    """Evaluation metric (no gradient): SSIM over channels 0..2."""

    def __init__(self, window_size=11, size_average=True):
        super().__init__()
        if window_size != 11 or not size_average:
            raise NotImplementedError("tatt_amd SSIM: window_size=11, size_average=True (the reference's use)")

    def forward(self, img1, img2):
        B, C, H, W = _check(img1, img2)
        a, b = img1.detach().contiguous().float(), img2.detach().contiguous().float()
        out = torch.empty(1, device=a.device, dtype=torch.float32)
        ws = torch.empty(B * 3, device=a.device, dtype=torch.float32)
        lib().ssim_fwd(a, b, out, B, C, min(3, C), H, W, ws)
        return out[0]


def calculate_psnr(img1, img2):
    """img1 and img2 have range [0, 1]; channels 0..2 (utils/ssim_psnr.py:9-15)."""
    B, C, H, W = img1.shape
    if not img1.is_cuda:
        raise RuntimeError("tatt_amd SSIM/PSNR run on the GPU only")
    a, b = img1.detach().contiguous().float(), img2.detach().contiguous().float()
    out = torch.empty(1, device=a.device, dtype=torch.float32)
    ws = torch.empty(B * 3, device=a.device, dtype=torch.float32)
    lib().psnr(a, b, out, B, C, min(3, C), H, W, ws)
    return out[0]
