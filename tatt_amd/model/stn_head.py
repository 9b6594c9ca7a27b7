"""STNHead parameter container (reference: model/stn_head.py:25-106).

Same constructor, parameter names/shapes and initialisation as the reference; the arithmetic runs in the
HIP engine (tatt_amd/engine.py), not in these modules.
"""
import math

import numpy as np
import torch
from torch import nn


def conv3x3_block(in_planes, out_planes):
    """conv3x3 + BN + ReLU triple (reference model/stn_head.py:13-22)."""
    return nn.Sequential(nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=1, padding=1),
                         nn.BatchNorm2d(out_planes), nn.ReLU(inplace=True))


class STNHead(nn.Module):
    # (Cin, Cout, pool) per stage — model/stn_head.py:32-45
    STAGES = [(None, 32, (2, 2)), (32, 64, (2, 2)), (64, 128, (2, 2)), (128, 256, (2, 2)), (256, 256, (1, 2)), (256, 256, None)]

    def __init__(self, in_planes, num_ctrlpoints, activation='none', input_size=(16, 64)):
        super().__init__()
        assert activation == 'none', "the hot path uses activation='none' (model/tsrn.py:641)"
        self.in_planes, self.num_ctrlpoints, self.activation = in_planes, num_ctrlpoints, activation
        layers = []
        cin = in_planes
        for _, cout, pool in self.STAGES:
            layers.append(conv3x3_block(cin, cout))
            if pool is not None:
                layers.append(nn.MaxPool2d(kernel_size=pool, stride=pool))
            cin = cout
        self.stn_convnet = nn.Sequential(*layers)
        self.stn_fc1 = nn.Sequential(nn.Linear(512, 512), nn.BatchNorm1d(512), nn.ReLU(inplace=True))
        self.stn_fc2 = nn.Linear(512, num_ctrlpoints * 2)
        self._init_weights(self.stn_convnet)
        self._init_weights(self.stn_fc1)
        self._init_stn()

    @staticmethod
    def _init_weights(module):          # model/stn_head.py:59-71
        for m in module.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
                m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                m.weight.data.normal_(0, 0.001)
                m.bias.data.zero_()

    def _init_stn(self):                # model/stn_head.py:73-90 : identity control points
        margin, side = 0.01, self.num_ctrlpoints // 2
        xs = np.linspace(margin, 1. - margin, side)
        pts = np.concatenate([np.stack([xs, np.full(side, margin)], 1), np.stack([xs, np.full(side, 1 - margin)], 1)], 0)
        self.stn_fc2.weight.data.zero_()
        self.stn_fc2.bias.data = torch.tensor(pts.astype(np.float32)).view(-1)

    def forward(self, x):
        raise NotImplementedError("STNHead runs inside TSRN_TL_TRANS.forward (HIP engine)")
