"""TPSSpatialTransformer buffers (reference: model/tps_spatial_transformer.py:54-112).

Holds the same four registered buffers; sampling itself is the fused HIP kernel tatt_tps_sample_fwd/bwd.
"""
import numpy as np
import torch
from torch import nn


def _partial_repr(a, b):
    """phi(r) = r^2 log r with 0 at r = 0 (model/tps_spatial_transformer.py:22-34); 0.5*d2*log(d2)."""
    d = a.view(-1, 1, 2) - b.view(1, -1, 2)
    d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]
    r = 0.5 * d2 * torch.log(d2)
    return torch.where(torch.isnan(r), torch.zeros_like(r), r)


class TPSSpatialTransformer(nn.Module):
    def __init__(self, output_image_size=None, num_control_points=None, margins=None):
        super().__init__()
        self.output_image_size, self.num_control_points, self.margins = output_image_size, num_control_points, margins
        self.target_height, self.target_width = H, W = output_image_size
        N, side = num_control_points, num_control_points // 2
        xs = np.linspace(margins[0], 1.0 - margins[0], side)
        ctrl = torch.Tensor(np.concatenate([np.stack([xs, np.full(side, margins[1])], 1),
                                            np.stack([xs, np.full(side, 1.0 - margins[1])], 1)], 0))
        fk = torch.zeros(N + 3, N + 3)
        fk[:N, :N] = _partial_repr(ctrl, ctrl)
        fk[:N, -3] = 1
        fk[-3, :N] = 1
        fk[:N, -2:] = ctrl
        fk[-2:, :N] = ctrl.t()
        ys, xs_ = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        coord = torch.stack([xs_.reshape(-1) / (W - 1), ys.reshape(-1) / (H - 1)], 1)
        rep = torch.cat([_partial_repr(coord, ctrl), torch.ones(H * W, 1), coord], 1)
        self.register_buffer('inverse_kernel', torch.inverse(fk).contiguous())
        self.register_buffer('padding_matrix', torch.zeros(3, 2))
        self.register_buffer('target_coordinate_repr', rep)
        self.register_buffer('target_control_points', ctrl)

    def forward(self, input, source_control_points):
        raise NotImplementedError("TPS runs inside TSRN_TL_TRANS.forward (HIP engine)")
