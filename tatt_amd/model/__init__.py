from . import tsrn, transformer_v2, stn_head, tps_spatial_transformer, crnn  # noqa: F401

import torch as _torch

from ..lib import lib as _lib


class _RotateFn(_torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, arc):
        if not img.is_cuda:
            raise RuntimeError("tatt_amd torch_distortion runs on the GPU only")
        x = img.detach().contiguous().float()
        arc = arc.detach().to(device=x.device, dtype=_torch.float32).contiguous()
        B, C, H, W = x.shape
        y = _torch.empty_like(x)
        _lib().rotate_fwd(x, arc, y, B, C, H, W)
        ctx.save_for_backward(arc)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arc,) = ctx.saved_tensors
        dy = dy.contiguous().float()
        B, C, H, W = dy.shape
        dx = _torch.zeros_like(dy)                       # the kernel scatters (+=) into dx
        _lib().rotate_bwd(arc, dy, dx, B, C, H, W)
        return dx, None


def torch_distortion(torch_image_batches, arc_batches, rand_offs=None, off_range=0.2):
    """Rotation used by the TSC loss: reference model/__init__.py:4-30 and TextSR.torch_rotate_img
    (interfaces/super_resolution_eng.py:50-82).  theta = [[cos, sin*r, 0], [-sin/r, cos, 0]], r = H/W, affine_grid +
    bilinear grid_sample with the torch defaults.  The trainer always calls it with rand_offs=None; the jittered ratio
    is not built."""
    if rand_offs is not None:
        raise NotImplementedError("tatt_amd torch_distortion: rand_offs is not used by the TATT trainer and is not built")
    return _RotateFn.apply(torch_image_batches, arc_batches)
